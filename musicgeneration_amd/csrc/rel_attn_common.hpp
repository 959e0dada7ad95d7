// Shared pieces of the relative-attention kernels (forward + the three backward kernels).
//
// Geometry (all kernels): MFMA v_mfma_f32_32x32x16_bf16, dh = 64, tiles of 32 queries x 32 keys.
//   A operand lane l: row (l&31), k = 8*(l>>5)+j, j=0..7  -> 16 contiguous bytes of a row-major row
//   B operand lane l: col (l&31), k = 8*(l>>5)+j          -> 16 contiguous bytes of a row-major row
//   C/D       lane l: col (l&31), row(r) = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15
//
// Relative term.  With delta = i - j >= 0 the reference's skewed Srel[i,j] = q_i . Er[delta] where
// Er[delta] = E[M-1-delta]  (layers.py:89-92,111-133).  Er is cut into 32-row "chunks"
// c = delta>>5.  For a 32x32 tile with D = i0-j0 (a multiple of 32) the tile needs
// delta in [D-31, D+31], i.e. chunks D/32-1 and D/32.  One product Q_tile . Er_chunk^T (same cost
// as Q K^T) therefore serves TWO consecutive key tiles: it is written once into a per-wave LDS
// "band" buffer band[a][ (delta & 63) ] (a = query row in tile) and each tile reads
// band[a][(D + a - b) & 63] -- the skew is a conflict-free diagonal LDS read, never a tensor.
#pragma once
#include "mgx_common.hpp"

namespace relattn {

constexpr int DH = 64;
constexpr int TILE_BYTES = 32 * DH * 2;   // one 32-row x 64-col bf16 tile = 4 KiB
constexpr float LOG2E = 1.4426950408889634f;
constexpr float PAD_NEG = -1.0e9f;        // additive mask value of the reference (layers.py:100)

typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4_ptr;
typedef __attribute__((address_space(3))) float* lds_f32_ptr;
typedef __attribute__((address_space(3))) char* lds_char_ptr;
// LDS byte address of a __shared__ object as an integer, and a store through such an integer.  Address arithmetic done
// on the integer (XOR included) stays ONE VALU per store: through a generic pointer the compiler re-adds the (link-time)
// base of the dynamic LDS segment after every XOR.
MGX_DEV uint32_t lds_addr_of(const char* p) { return (uint32_t)(uintptr_t)(lds_char_ptr)p; }
MGX_DEV void lds_store_f32(uint32_t addr, float v) { *(lds_f32_ptr)(uintptr_t)addr = v; }

// row index (within the 32-row tile) held by accumulator register r of a lane in half hh
MGX_DEV int crow(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// LDS image "R" (row-fragment reads, ds_read_b128): 32 rows x 128 B, 16-byte chunk index XORed with
// (row>>1)&7 so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-B slots.
MGX_DEV int imgR_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// LDS image "T" (transposed reads, ds_read_b64_tr_b16): chunk XORed with ((row>>1)&1)<<2 so that
// rows q and q+2 of a 4-row block land in different 128-B halves of the 256-B bank row.
MGX_DEV int imgT_off(int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); }

// A/B row-fragment (8 bf16 at k = 16*ks + 8*hh ..) of row `row` from an image-R tile
MGX_DEV bf16x8 frag_R(const char* tile, int row, int hh, int ks) {
    return *(const bf16x8*)(tile + imgR_off(row, 2 * ks + hh));
}

// Transposed fragment from an image-T tile holding X[32 rows][64 cols]:
// returns, for this lane, X[kappa(j)][col0 + (lane&31)], j = 0..7, with
// kappa(j) = 16*s + 8*(j>>2) + 4*hh + (j&3)  -- the k order of an accumulator tile used as operand.
MGX_DEV bf16x8 frag_T(const char* tile, int lane, int s, int ct) {
    const int i = lane & 15, g = lane >> 4, hh = lane >> 5;
    const int rq = i >> 2;                               // row within the 4-row block this lane addresses
    const int chunk = 4 * ct + 2 * (g & 1) + ((i & 3) >> 1);
    const int byte_in = 8 * (i & 1);
    bf16x8 out;
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        const int row = 16 * s + 8 * jq + 4 * hh + rq;
        const char* p = tile + imgT_off(row, chunk) + byte_in;
        bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)p);
        out[4 * jq + 0] = t[0]; out[4 * jq + 1] = t[1]; out[4 * jq + 2] = t[2]; out[4 * jq + 3] = t[3];
    }
    return out;
}

// transposed fragment read from an image-R tile (2-way bank conflict, saves a second LDS image):
// X[kappa(j)][32*ct + (lane&31)], kappa(j) = 16*s + 8*(j>>2) + 4*hh + (j&3)
MGX_DEV bf16x8 frag_T_onR(const char* tile, int lane, int s, int ct) {
    const int i = lane & 15, g = lane >> 4, hh = lane >> 5;
    const int rq = i >> 2;
    const int chunk = 4 * ct + 2 * (g & 1) + ((i & 3) >> 1);
    const int byte_in = 8 * (i & 1);
    bf16x8 out;
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        const int row = 16 * s + 8 * jq + 4 * hh + rq;
        bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + imgR_off(row, chunk) + byte_in));
        out[4 * jq + 0] = t[0]; out[4 * jq + 1] = t[1]; out[4 * jq + 2] = t[2]; out[4 * jq + 3] = t[3];
    }
    return out;
}

// LDS image "B" (both): the chunk XORed with a permutation of (row>>1)&7 whose bit 2 is bit 1 of the row -- the eight same-parity rows a
// ds_read_b128 lane group touches still get eight different values (conflict-free row fragments, as on image R), and rows q and q+2 of a
// 4-row block differ in bit 2 (conflict-free ds_read_b64_tr_b16, as on image T): ONE image serves both kinds of fragment reads.
MGX_DEV int imgB_swz(int row) { return (((row >> 1) & 1) << 2) | (((row >> 3) & 1) << 1) | ((row >> 2) & 1); }
MGX_DEV int imgB_off(int row, int chunk) { return row * 128 + ((chunk ^ imgB_swz(row)) << 4); }
MGX_DEV bf16x8 frag_B(const char* tile, int row, int hh, int ks) {
    return *(const bf16x8*)(tile + imgB_off(row, 2 * ks + hh));
}
MGX_DEV bf16x8 frag_T_onB(const char* tile, int lane, int s, int ct) {      // as frag_T_onR
    const int i = lane & 15, g = lane >> 4, hh = lane >> 5;
    const int rq = i >> 2;
    const int chunk = 4 * ct + 2 * (g & 1) + ((i & 3) >> 1);
    const int byte_in = 8 * (i & 1);
    bf16x8 out;
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        const int row = 16 * s + 8 * jq + 4 * hh + rq;
        bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + imgB_off(row, chunk) + byte_in));
        out[4 * jq + 0] = t[0]; out[4 * jq + 1] = t[1]; out[4 * jq + 2] = t[2]; out[4 * jq + 3] = t[3];
    }
    return out;
}

MGX_DEV u32x4 scale8(const u32x4& raw, float sc) {
    float f[8];
    unpack8(raw, f);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] *= sc;
    return pack8(f);
}

// accumulator registers 8s..8s+7 -> bf16x8 operand fragment for k-step s (k order kappa, see frag_T)
MGX_DEV bf16x8 acc_to_frag(const f32x16& c, int s) {
    u32x4 w;
    w.x = pack_bf16x2(c[8 * s + 0], c[8 * s + 1]);
    w.y = pack_bf16x2(c[8 * s + 2], c[8 * s + 3]);
    w.z = pack_bf16x2(c[8 * s + 4], c[8 * s + 5]);
    w.w = pack_bf16x2(c[8 * s + 6], c[8 * s + 7]);
    return __builtin_bit_cast(bf16x8, w);
}

MGX_DEV f32x16 mfma(const bf16x8& a, const bf16x8& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

MGX_DEV f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[r] = 0.f;
    return z;
}

// ---- rotated band (skew buffer) ---------------------------------------------------------------------
// One band row per query a of the tile, 64 fp32 columns + 4 pad = 272 B (17 16-byte slots: the 16
// lanes of a ds_read_b128 group land on 16 distinct slots).  Relative distance delta lives at column
//     col(a, delta) = (a - delta) & 63.
// * swapped orientation (keys on registers, queries on lanes): the 4 registers of group g4 hold keys
//   b = 8*g4 + 4*hh + j, delta = D + a - b  =>  col = (b - D) & 63: four ascending, 16-byte aligned
//   columns that never wrap => ONE ds_read_b128 at  a*272 + 16*hh + 32*g4 (+128 when D/32 is odd).
// * natural orientation (queries on registers, keys on lanes): col = (bl - D) & 63 = bl (+32) =>
//   ds_read_b32 at  a_r*272 + 4*bl (+128): one base register + immediates.
// * QE[a_r][t] (t on lanes) of chunk q is written at col (a_r - t - 32q) & 63: per-lane wrap, so the 16
//   byte offsets are precomputed once for even q (wa0) and odd q (wa1 = column ^ 32).
constexpr int BAND_STRIDE = 272;
constexpr int BAND_BYTES = 32 * BAND_STRIDE;   // 8,704 B per wave
MGX_DEV int band_off(int row, int col) { return row * BAND_STRIDE + col * 4; }
// Row placement variant used by the dQ kernel: row a sits at slot ((a&3)|((a>>3)<<2)) of region (a>>2)&1.
// For the row crow(r,hh) held by accumulator register r this is simply slot r of region hh, and the
// region size (4,352 B = 17*256) keeps the low 8 address bits equal to the column byte offset, so ONE
// precomputed per-register offset serves both chunk parities: XOR byte-address bit 7 flips column bit 5.
constexpr int BAND_REGION = 16 * BAND_STRIDE;
MGX_DEV int band_rowoff(int a) { return ((a & 3) | ((a >> 3) << 2)) * BAND_STRIDE + ((a >> 2) & 1) * BAND_REGION; }
// store an accumulator tile through precomputed offsets; `odd` must be wave-uniform
MGX_DEV void band_store(char* band, const int (&wa0)[16], const int (&wa1)[16], int odd, const f32x16& v) {
    if (odd) {
#pragma unroll
        for (int r = 0; r < 16; ++r) *(float*)(band + wa1[r]) = v[r];
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) *(float*)(band + wa0[r]) = v[r];
    }
}

// LDS-DMA: one wave instruction moves 64 x 16 (or 4) bytes global -> LDS without passing through VGPRs.  The LDS destination is
// (wave-uniform base, in M0) + lane * size, so an image's bank swizzle is applied on the SOURCE side: lane l fetches the bytes
// that belong in slot l.  Issued from inline asm, in the (SGPR base + 32-bit lane offset) address form:
//  * through the builtin hipcc treats the instruction as a FLAT access with LDS side effects: while one is in flight every wait
//    for an ordinary load becomes s_waitcnt vmcnt(0) (no counted waits: the tile prefetch would be waited for at the top of the
//    same step), every ds_read_b64_tr_b16 builtin gets a vmcnt(0) in front (possible alias of the DMA's destination) and
//    __syncthreads() a vmcnt(0) for its release fence;
//  * the compiler therefore does NOT know that these instructions write LDS or occupy vmcnt: the caller separates them from the
//    LDS reads of the same bytes by an explicit counted s_waitcnt vmcnt(N) + barrier (N = VMEM operations issued after them).
//    A compiler-generated vmcnt(N') for an older ordinary load only ever waits longer than needed, never shorter.
//  * hazards inside the string are the author's: the s_nop 0 is the wait state an LDS-DMA needs after the SALU write of M0 (hipcc pads
//    nothing inside an asm statement; M0 is written in the same statement that uses it because the compiler does not preserve it).
MGX_DEV void dma16(const char* sbase /* wave-uniform */, uint32_t voff, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave_base) : "memory", "m0");
}
MGX_DEV void dma4(const char* sbase /* wave-uniform */, uint32_t voff, uint32_t lds_wave_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %0" ::"s"(sbase), "v"(voff), "s"(lds_wave_base) : "memory", "m0");
}

// order LDS traffic of one wave (same-wave DS ops execute in order; this only pins the compiler)
MGX_DEV void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Store a transposed accumulator pair T^T[c][x] (c = 32*ct + crow(r,hh) on registers, x = row on lanes) as rows
// x = 0..31 of 64 consecutive bf16 each, scaled by sc (per lane = per row).  Written straight from the registers a
// wave instruction would touch 16 bytes of 32 different rows; parked in a wave-private LDS patch (>= 4,608 B, 144-byte
// rows) and read back row-major, every global store instruction writes 8 whole 128-byte rows instead.
MGX_DEV void store_rows_lds(uint16_t* dst, size_t row_stride, const f32x16& t0, const f32x16& t1, int lane, float sc,
                            char* patch) {
    const int x = lane & 31, hh = lane >> 5;
    wave_lds_fence();
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        *(u32x2*)(patch + x * 144 + (8 * g4 + 4 * hh) * 2) =
            u32x2{pack_bf16x2(t0[4 * g4] * sc, t0[4 * g4 + 1] * sc), pack_bf16x2(t0[4 * g4 + 2] * sc, t0[4 * g4 + 3] * sc)};
        *(u32x2*)(patch + x * 144 + (32 + 8 * g4 + 4 * hh) * 2) =
            u32x2{pack_bf16x2(t1[4 * g4] * sc, t1[4 * g4 + 1] * sc), pack_bf16x2(t1[4 * g4 + 2] * sc, t1[4 * g4 + 3] * sc)};
    }
    wave_lds_fence();
    const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rr + 8 * i;
        *(u32x4*)(dst + (size_t)row * row_stride + ch * 8) = *(const u32x4*)(patch + row * 144 + ch * 16);
    }
    wave_lds_fence();
}

// ---- fragment-ordered copies of Er ------------------------------------------------------------------------------
// The MFMA operands built from E are 16 bytes per lane of 32 DIFFERENT rows of E: read from E's natural
// [delta][64] layout one wave load touches 32 cache lines for 1 KB of data.  A tiny pre-pass (256 KB at L = 2048,
// once per attention call) rewrites Er[delta] = E[M-1-delta] in the order the lanes consume it, so every wave load
// is 1 KB contiguous:
//   EfA[(q*4 + ks)*64 + lane]            (u32x4) = Er[32q + (lane&31)][16ks + 8(lane>>5) + j],        j = 0..7
//       row fragment ks of chunk q: B operand of Q.Er^T (forward, dQ, dK/dV kernels)
//   EfT[((q*2 + ks)*2 + ct)*64 + lane]   (u32x4) = Er[32q + 16ks + 8(lane>>5) + j][32ct + (lane&31)], j = 0..7
//       fragment of the transposed chunk: A operand of dq^T += Er^T . dQE^T (dQ kernel)
inline size_t er_frag_bytes(int L) { return (((size_t)L * 64 * 2) + 255) / 256 * 256; }

static __global__ __launch_bounds__(256) void er_frag_kernel(const uint16_t* __restrict__ Er, u32x4* __restrict__ EfA,
                                                             u32x4* __restrict__ EfT, int L) {
    const int n = (L >> 5) * 4 * 64;                      // u32x4 units per buffer
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < n) {
        const int lane = gid & 63, ks = (gid >> 6) & 3, q = gid >> 8;
        const int delta = 32 * q + (lane & 31);
        EfA[gid] = *(const u32x4*)(Er + (size_t)(L - 1 - delta) * 64 + 16 * ks + 8 * (lane >> 5));
    } else if (gid < 2 * n && EfT) {
        const int g = gid - n;
        const int lane = g & 63, ct = (g >> 6) & 1, ks = (g >> 7) & 1, q = g >> 8;
        const int col = 32 * ct + (lane & 31);
        uint16_t v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int delta = 32 * q + 16 * ks + 8 * (lane >> 5) + j;
            v[j] = Er[(size_t)(L - 1 - delta) * 64 + col];
        }
        u32x4 o;
        o.x = v[0] | ((uint32_t)v[1] << 16); o.y = v[2] | ((uint32_t)v[3] << 16);
        o.z = v[4] | ((uint32_t)v[5] << 16); o.w = v[6] | ((uint32_t)v[7] << 16);
        EfT[g] = o;
    }
}

static inline void launch_er_frag(const uint16_t* Er, u32x4* EfA, u32x4* EfT, int L, hipStream_t s) {
    const int n = (L >> 5) * 4 * 64 * (EfT ? 2 : 1);
    hipLaunchKernelGGL(er_frag_kernel, dim3((n + 255) / 256), dim3(256), 0, s, Er, EfA, EfT, L);
}

// rel_attn_dkv64.hip: dK / dV with 64 keys per wave and the generated asm main loop (L % 128 == 0)
int dkv64_launch(const uint16_t* qkv, const void* EfA, const uint32_t* padbits, const uint16_t* dctx, const float* nlse2,
                 const float* ndelta, uint16_t* dqkv, uint16_t* dst, int B, int L, int d, int bg, void* stream);
// experiment builds only (tools/experiments/rel_attn_fwd64.hip): forward with 64 query rows per wave and a generated asm sweep
int fwd64a_launch(const uint16_t* qkv, const void* Ef, const uint32_t* padbits, uint16_t* ctx, float* lse, int B, int L, int d, int bg,
                  void* stream);
// experiment builds only (tools/experiments/rel_attn_fwd2.hip, rel_attn_fwd3.hip; MGX_EXPERIMENTS)
int fwd64_launch(const uint16_t* qkv, const void* EfA, const uint32_t* padbits, uint16_t* ctx, float* lse, int B, int L, int d,
                 void* stream);
int fwdpp_launch(const uint16_t* qkv, const void* EfA, const uint32_t* padbits, uint16_t* ctx, float* lse, int B, int L, int d,
                 void* stream);

}  // namespace relattn
