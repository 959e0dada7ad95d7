// Fused relative global attention, backward: the software-pipelined dQ kernel for L % 256 == 0 (cfg2, cfg4).
// (autograd of layers.py:86-106 of the reference; rel_attn_bwd.hip holds the math, the other kernels and every other shape)
//
// Same ownership as rel_attn_dq_kernel (query rows own dq, sweep over key tiles, dS exported un-skewed for the streamed dE
// kernel), different wave structure -- the one rel_attn_fwd2.hip tried on the forward, which suits the backward better: a dQ
// tile carries 20 MFMAs for ~55 VALU + ~40 LDS instructions, so ONE wave per SIMD (one instruction issued per ~4 cycles) can
// keep the matrix pipe fed if its stream always has an MFMA group beside its VALU / LDS work.  A wave owns TWO adjacent
// 32-row query blocks (A, B) with the whole 512-entry register file; K / V / K^T / Er / ErT fragments are read once for both.
//
//   iteration s (key tile s; both blocks X):
//     G1  S_X^T   = K(s) Q_X^T + Srel_X^T        8 MFMA | band stores of Q_X.Er^T for step s+1 (half), staging, V / K^T fragments
//     G2  dP_X^T  = V(s) dO_X^T                  8 MFMA | P = exp2(S log2e - lse)        ; band stores (other half)
//     G3  QE_X    = Q_X Er_chunk^T  (step s+2)   8 MFMA | dS = P (dP - delta), bf16 packing
//     G4  dq_X^T += K(s)^T dS_X^T                8 MFMA | un-skew stores of dS into the (query, distance) band ; its fragment reads
//         workgroup barrier (K/V tile s+1 visible)
//     G5  dq_X^T += ErT_chunk dQE_X^T            8 MFMA | band loads for step s+1, K fragments of tile s+1, dS export, Er loads
//
// Band chunks are computed TWO steps ahead (G3) so their 32 LDS stores spread over G1/G2 of the next iteration: the LDS store
// path (64 B/clk per CU) is the second budget of this kernel -- 64 band stores per wave and step are 1,024 of a step's
// ~1,280 MFMA cycles per CU when four waves store at once.
// A wave runs the body unmasked up to its own diagonal, masked for its two diagonal steps (every step when the batch row has
// padded keys), then only keeps staging K/V tiles for the waves below it.
#include <type_traits>
#include "rel_attn_common.hpp"

using namespace relattn;

namespace q2 {
constexpr int WAVES = 4;
constexpr int OFF_KR = 0;                                       // 2 x 4 KiB  K image R (row + transposed reads)
constexpr int OFF_VR = OFF_KR + 2 * TILE_BYTES;                 // 2 x 4 KiB  V image R
constexpr int OFF_BAND = OFF_VR + 2 * TILE_BYTES;               // 8 x 8,704 B fp32 rotated bands (wave w block X at 2w+X)
constexpr int DB_BYTES = 32 * 128;                              // bf16 band of dS by (query, delta & 63): 128-byte rows, 16-byte
                                                                // chunks XOR-swizzled by (row >> 1) & 7 (a lane only touches its own row)
constexpr int OFF_DBAND = OFF_BAND + 2 * WAVES * BAND_BYTES;    // 8 dbands
constexpr int OFF_DO = OFF_DBAND + 2 * WAVES * DB_BYTES;        // 8 x 4 KiB  dO rows of (wave, block), image R: read every step (saves 32 VGPRs)
constexpr int OFF_PAD = OFF_DO + 2 * WAVES * TILE_BYTES;        // key-padding words of this batch row (first 256)
constexpr int OFF_FLAG = OFF_PAD + 1024;
constexpr int LDS_BYTES = OFF_FLAG + 16;                        // 152,592 B -> 1 workgroup per CU
}  // namespace q2

// MGX_B2_STAMP (diagnostic build only, `_build.py --variant stamp -DMGX_B2_STAMP`): s_memtime stamps at the region
// boundaries of a step; lane 0 of each wave writes its sums over block A's delta rows (tools/attn64_stamp.py reads them).
#ifdef MGX_B2_STAMP
#define STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define STAMP_ADD(acc, t1, t0) acc += (t1) - (t0)
#else
#define STAMP(var)
#define STAMP_ADD(acc, t1, t0)
#endif

#ifndef MGX_B2_SCHED
#define MGX_B2_SCHED 1
#endif
#if MGX_B2_SCHED
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#else
#define SGB(mask, n)
#endif
// sched_group_barrier masks
#define SG_VALU 0x002
#define SG_MFMA 0x008
#define SG_VMEM 0x010
#define SG_DSR 0x100
#define SG_DSW 0x200

__global__ __launch_bounds__(256, 1) void rel_attn_dq64_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ EfA, const u32x4* __restrict__ EfT,
    const uint32_t* __restrict__ padbits, const uint16_t* __restrict__ dctx, const float* __restrict__ lse,
    float* __restrict__ delta, uint16_t* __restrict__ dqkv, uint16_t* __restrict__ dsrel, const uint16_t* __restrict__ ctx,
    int L, int d, int bgroup) {
    using namespace q2;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    const int nqb = L >> 8;
    const int b = (blockIdx.y / nqb) * bgroup + blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qb = nqb - 1 - (blockIdx.y % nqb);         // heaviest (latest) query blocks first
    const int Q0 = qb * 8;
    const int qA = Q0 + 2 * w;                           // block A's diagonal tile (even); block B's is qA + 1
    const int nchunk = L >> 5;
    const int ntw = Q0 + 8;                              // key tiles this workgroup visits
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;
#ifdef MGX_B2_STAMP
    unsigned long long st_r1 = 0, st_bar = 0, st_r2 = 0, st_s1 = 0, st_s2 = 0, st_s3 = 0;
#endif
    STAMP(st_begin);

    const int srow = tid >> 3, sch = tid & 7;
    const int st_offR = imgR_off(srow, sch);
    const uint16_t* kg = qkv_b + (size_t)srow * ld + d + hd * 64 + sch * 8;
    const uint16_t* vg = kg + d;
    const size_t tile_stride = (size_t)32 * ld;
    // fragment-ordered copies of Er (er_frag_kernel): every load is unconditional with a clamped index
    auto e_frag = [&](int q, int ks) { return __builtin_bit_cast(bf16x8, EfA[(size_t)(max(q, 0) * 4 + ks) * 64 + lane]); };
    auto et_frag = [&](int q, int i) {                   // i = 2*ks + ct
        return __builtin_bit_cast(bf16x8, EfT[(size_t)(min(max(q, 0), nchunk - 1) * 4 + i) * 64 + lane]);
    };

    // ---- prologue ------------------------------------------------------------------------------------------------------
    *(u32x4*)(smem + OFF_KR + st_offR) = *(const u32x4*)kg;
    *(u32x4*)(smem + OFF_VR + st_offR) = *(const u32x4*)vg;
    // zero the dS bands (their never-written halves must read as 0 on the first step)
    for (int o = tid * 16; o < 2 * WAVES * DB_BYTES; o += 256 * 16) *(u32x4*)(smem + OFF_DBAND + o) = u32x4{0, 0, 0, 0};
    int anypad = 0;
    if (padbits) {
        if (tid == 0) *(volatile uint32_t*)(smem + OFF_FLAG) = 0u;
        __syncthreads();
        uint32_t acc = 0;
#pragma unroll 1
        for (int t = tid; t < ntw; t += 256) {
            const uint32_t pwv = padbits[(size_t)b * nchunk + t];
            if (t < 256) *(uint32_t*)(smem + OFF_PAD + 4 * t) = pwv;
            acc |= pwv;
        }
        if (acc) *(volatile uint32_t*)(smem + OFF_FLAG) = 1u;
        __syncthreads();
        anypad = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)(smem + OFF_FLAG));
    }
    auto padword = [&](int kt) -> uint32_t {             // wave-uniform
        if (!anypad) return 0u;
        uint32_t v = *(const uint32_t*)(smem + OFF_PAD + 4 * min(kt, 255));
        if (kt >= 256) v = padbits[(size_t)b * nchunk + kt];
        return __builtin_amdgcn_readfirstlane(v);
    };
    bf16x8 qf[2][4];
    float lse2[2], dlt[2];
    char* dotile = smem + OFF_DO + 2 * w * TILE_BYTES;   // this wave's dO rows (block A; block B one tile further), image R
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const int i0 = (qA + X) * 32;
        const uint16_t* qp = qkv_b + (size_t)(i0 + a) * ld + hd * 64 + hh * 8;
        const uint16_t* dp = dctx + ((size_t)b * L + i0 + a) * d + hd * 64 + hh * 8;
        const uint16_t* op = ctx + ((size_t)b * L + i0 + a) * d + hd * 64 + hh * 8;
        float acc_d = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[X][ks] = __builtin_bit_cast(bf16x8, scale8(*(const u32x4*)(qp + ks * 16), 0.125f));
            const u32x4 draw = *(const u32x4*)(dp + ks * 16);
            *(u32x4*)(dotile + X * TILE_BYTES + imgR_off(a, 2 * ks + hh)) = draw;      // read back as frag_R(.., a, hh, ks)
            float o8[8], g8[8];
            unpack8(*(const u32x4*)(op + ks * 16), o8);
            unpack8(draw, g8);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc_d += o8[k] * g8[k];
        }
        const size_t si = ((size_t)b * heads + hd) * L + i0 + a;
        lse2[X] = lse[si] * LOG2E;
        // delta_i = sum_c dO[i][c] O[i][c]: this wave owns the row; published for the dK/dV kernel, which runs after this one
        dlt[X] = acc_d + __shfl_xor(acc_d, 32, 64);
        if (hh == 0) delta[si] = dlt[X];
    }
    u32x4 kreg = *(const u32x4*)(kg + tile_stride);      // ntw >= 8: tiles 1 and 2 exist
    u32x4 vreg = *(const u32x4*)(vg + tile_stride);
    size_t koff = 2 * tile_stride;                       // element offset of the tile the NEXT prefetch fetches
    int tnext = 2;
    auto prefetch_next = [&]() {
        kreg = *(const u32x4*)(kg + koff);
        vreg = *(const u32x4*)(vg + koff);
        koff += (tnext + 1 < ntw) ? tile_stride : 0;
        ++tnext;
    };
    __syncthreads();

    // fp32 band of Q.Er^T (rel_attn_common.hpp), block A's of this wave; block B's is BAND_BYTES further (an immediate)
    const int band_base = OFF_BAND + 2 * w * BAND_BYTES;
    uint32_t wc[16];                                     // even chunk parity; odd = ^128 (one VALU per store of the odd block)
#pragma unroll
    for (int r = 0; r < 16; ++r) wc[r] = lds_addr_of(smem) + band_base + hh * BAND_REGION + (((crow(r, hh) - a) & 63) << 2);
    const int rbase = band_base + band_rowoff(a) + 16 * hh;
    auto band_put = [&](const f32x16& v, int X, int par) {
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_store_f32((par ? (wc[r] ^ 128u) : wc[r]) + r * BAND_STRIDE + X * BAND_BYTES, v[r]);
    };
    auto band_get = [&](int X, int par) {
        const char* rb = smem + rbase + X * BAND_BYTES + (par << 7);
        f32x16 c;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *(const f32x4*)(rb + 32 * g4);
            c[4 * g4] = v.x; c[4 * g4 + 1] = v.y; c[4 * g4 + 2] = v.z; c[4 * g4 + 3] = v.w;
        }
        return c;
    };
    auto qe_prod = [&](int X, const bf16x8 (&e)[4]) {
        f32x16 c = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma(qf[X][ks], e[ks], c);
        return c;
    };
    // bf16 band of dS by (query a, delta & 63): a lane writes and reads only ITS OWN row (128 bytes, 16-byte chunk ch stored
    // at ch ^ ((a >> 1) & 7): the 16 rows of a ds_read_b128 group land on 16 distinct 16-byte slots).  dwa[r] = LDS byte address
    // of key crow(r,hh) of a tile whose D/32 is even (block A's band; block B's is DB_BYTES further); odd = ^64 (column bit 5).
    const int dsw = (a >> 1) & 7;
    uint32_t dwa[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = (a - crow(r, hh)) & 63;
        dwa[r] = lds_addr_of(smem) + OFF_DBAND + 2 * w * DB_BYTES + a * 128 + (((c >> 3) ^ dsw) << 4) + (c & 7) * 2;
    }
    // fragment reads of a completed chunk: columns 32 par + 16 ks + 8 hh .. +7 = chunk 4 par + 2 ks + hh
    uint32_t gqa[2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            gqa[p][ks] = lds_addr_of(smem) + OFF_DBAND + 2 * w * DB_BYTES + a * 128 + (((4 * p + 2 * ks + hh) ^ dsw) << 4);

    // ---- band prologue: "old" and "new" chunks of step 0, pending chunks of step 1; K fragments of tile 0 ------------------
    bf16x8 e[2][4], et[2][4], kf[4];
    f32x16 cS[2], qe[2];
    {
        bf16x8 c3[4], c2[4], c1[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            c3[ks] = e_frag(qA + 1, ks); c2[ks] = e_frag(qA, ks); c1[ks] = e_frag(qA - 1, ks);
            e[0][ks] = e_frag(qA - 2, ks); e[1][ks] = e_frag(qA - 3, ks);
            et[0][ks] = et_frag(qA + 1, ks); et[1][ks] = et_frag(qA, ks);
        }
        band_put(qe_prod(0, c2), 0, 0);                  // OLD_A(0) = chunk qA   (parity 0)
        band_put(qe_prod(1, c3), 1, 1);                  // OLD_B(0) = chunk qA+1 (parity 1)
        band_put(qe_prod(1, c2), 1, 0);                  // NEW_B(0) = chunk qA
        band_put(qe_prod(0, c1), 0, 1);                  // NEW_A(0) = chunk qA-1
        qe[1] = qe_prod(1, c1);                          // NEW_B(1) = chunk qA-1, stored in iteration 0
        qe[0] = qe_prod(0, e[0]);                        // NEW_A(1) = chunk qA-2
        wave_lds_fence();
        cS[0] = band_get(0, 0);
        cS[1] = band_get(1, 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_R(smem + OFF_KR, a, hh, ks);
    }
    f32x16 dq[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
    // dS by (query, relative distance) for the dE kernel: tile (b,h, I, q) of the packed causal grid (rel_attn_bwd.hip)
    const size_t ntri = (size_t)nchunk * (nchunk + 1) / 2;
    uint16_t* dsp[2];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const size_t I = qA + X;
        dsp[X] = dsrel + (((size_t)b * heads + hd) * ntri + I * (I + 1) / 2) * 1024 + a * 16 + hh * 8;
    }
    const int am = a - 4 * hh;                           // key crow(r,hh) is in the future of query a  <=>  crow(r,0) > am

    auto step = [&](int s, auto par_tag, auto masked_tag) {
        constexpr int PAR = decltype(par_tag)::value;     // = s & 1
        constexpr bool MASKED = decltype(masked_tag)::value;
        STAMP(st_a);
        // ---- G1: staging of tile s+1, prefetch of tile s+2, V / K^T fragments of tile s, S^T, first half of the band stores
        *(u32x4*)(smem + OFF_KR + (PAR ^ 1) * TILE_BYTES + st_offR) = kreg;
        *(u32x4*)(smem + OFF_VR + (PAR ^ 1) * TILE_BYTES + st_offR) = vreg;
        prefetch_next();
        const char* kt = smem + OFF_KR + PAR * TILE_BYTES;
        const char* vt = smem + OFF_VR + PAR * TILE_BYTES;
        bf16x8 vf[4], ktf[2][2], dof[2][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) vf[ks] = frag_R(vt, a, hh, ks);
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dof[X][ks] = frag_R(dotile + X * TILE_BYTES, a, hh, ks);
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) { ktf[ss][0] = frag_T_onR(kt, lane, ss, 0); ktf[ss][1] = frag_T_onR(kt, lane, ss, 1); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cS[0] = mfma(kf[ks], qf[0][ks], cS[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cS[1] = mfma(kf[ks], qf[1][ks], cS[1]);
        band_put(qe[0], 0, PAR);                          // NEW_A(s+1) = chunk qA-s-2
        band_put(qe[1], 1, PAR ^ 1);                      // NEW_B(s+1) = chunk qA-s-1
        if (MASKED) {
            // key crow(r,hh) of tile s is in the future of query a of block X  <=>  crow(r,0) > am + 32 (qA + X - s)
            const uint32_t pwl = padword(s) >> (4 * hh);
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const int thr = am + 32 * (qA + X - s);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    cS[X][r] = ((crow(r, 0) > thr) || (pwl & (1u << crow(r, 0)))) ? -INFINITY : cS[X][r];
            }
        }
#ifdef MGX_B2_STAMP
        __builtin_amdgcn_sched_barrier(0); STAMP(st_g1); __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- G2: dP^T = V dO^T beside P = exp2(S log2e - lse) -------------------------------------------------------------
        f32x16 dp[2] = {zero16(), zero16()};
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dp[X] = mfma(vf[ks], dof[X][ks], dp[X]);
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int r = 0; r < 16; ++r) cS[X][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(cS[X][r], LOG2E, -lse2[X]));
#ifdef MGX_B2_STAMP
        __builtin_amdgcn_sched_barrier(0); STAMP(st_g2); __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- G3: Q.Er^T of step s+2 beside dS = P (dP - delta) and its bf16 packing ------------------------------------------
        qe[1] = qe_prod(1, e[PAR]);                       // NEW_B(s+2) = chunk qA-s-2
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[PAR][ks] = e_frag(qA - s - 4, ks);      // block A's chunk of the next iteration
        qe[0] = qe_prod(0, e[PAR ^ 1]);                   // NEW_A(s+2) = chunk qA-s-3
        bf16x8 dsf[2][2];
#pragma unroll
        for (int X = 0; X < 2; ++X) {
#pragma unroll
            for (int r = 0; r < 16; ++r) cS[X][r] = cS[X][r] * (dp[X][r] - dlt[X]);
            dsf[X][0] = acc_to_frag(cS[X], 0);
            dsf[X][1] = acc_to_frag(cS[X], 1);
        }
#ifdef MGX_B2_STAMP
        __builtin_amdgcn_sched_barrier(0); STAMP(st_g3); __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- G4: dq^T += K^T dS^T beside the un-skew stores; then the fragment reads of the completed chunk -----------------
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                dq[X][0] = mfma(ktf[ss][0], dsf[X][ss], dq[X][0]);
                dq[X][1] = mfma(ktf[ss][1], dsf[X][ss], dq[X][1]);
            }
        bf16x8 gq[2][2];
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int pX = X == 0 ? PAR : (PAR ^ 1);      // parity of D/32 = qA + X - s
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const u32x4 wv = __builtin_bit_cast(u32x4, dsf[X][ss]);       // word j: keys 8ss+2j (lo), 8ss+2j+1 (hi)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t a0 = dwa[8 * ss + 2 * j], a1 = dwa[8 * ss + 2 * j + 1];
                    *(__attribute__((address_space(3))) uint16_t*)(uintptr_t)((pX ? (a0 ^ 64u) : a0) + X * DB_BYTES) = (uint16_t)wv[j];
                    *(__attribute__((address_space(3))) uint16_t*)(uintptr_t)((pX ? (a1 ^ 64u) : a1) + X * DB_BYTES) = (uint16_t)(wv[j] >> 16);
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                gq[X][ks] = *(const __attribute__((address_space(3))) bf16x8*)(uintptr_t)(gqa[pX][ks] + X * DB_BYTES);
        }
#if MGX_B2_SCHED
        SGB(SG_DSW, 2);
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(SG_MFMA, 1); SGB(SG_DSR, 3); SGB(SG_DSW, 2); }                    // G1
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(SG_MFMA, 1); SGB(SG_VALU, 8); SGB(SG_DSW, 2); }                   // G2
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(SG_MFMA, 1); SGB(SG_VALU, 10); }                                  // G3
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(SG_MFMA, 1); SGB(SG_DSW, 4); }                                    // G4
        SGB(SG_DSR, 4);
#endif
        STAMP(st_b);
        __syncthreads();                                  // tile s+1 visible; every wave is done reading tile s-1's buffer
        STAMP(st_c);
        // ---- G5: dq^T += ErT dQE^T ; dS export ; band loads and K fragments for step s+1 ------------------------------------
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                  // block B: chunk qA-s+1 (block A's of the step before)
            dq[1][0] = mfma(et[PAR][2 * ks], gq[1][ks], dq[1][0]);
            dq[1][1] = mfma(et[PAR][2 * ks + 1], gq[1][ks], dq[1][1]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) et[PAR][i] = et_frag(qA - s - 1, i);         // block A's chunk of the next iteration
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                  // block A: chunk qA-s
            dq[0][0] = mfma(et[PAR ^ 1][2 * ks], gq[0][ks], dq[0][0]);
            dq[0][1] = mfma(et[PAR ^ 1][2 * ks + 1], gq[0][ks], dq[0][1]);
        }
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int dqx = qA + X - s;                   // the chunk just completed (block beyond its diagonal: none)
            if (!MASKED || dqx >= 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, gq[X][ks]), (u32x4*)(dsp[X] + (size_t)dqx * 1024 + 512 * ks));
            }
        }
        cS[0] = band_get(0, PAR ^ 1);                     // dq = qA-s-1
        cS[1] = band_get(1, PAR);                         // dq = qA-s
        const char* ktn = smem + OFF_KR + (PAR ^ 1) * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_R(ktn, a, hh, ks);
#if MGX_B2_SCHED
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(SG_MFMA, 1); SGB(SG_DSR, 2); SGB(SG_VMEM, 1); }
#endif
        __builtin_amdgcn_sched_barrier(0);
        STAMP(st_d);
        if (!MASKED) { STAMP_ADD(st_r1, st_b, st_a); STAMP_ADD(st_bar, st_c, st_b); STAMP_ADD(st_r2, st_d, st_c); }
#ifdef MGX_B2_STAMP
        if (!MASKED) { st_s1 += st_g1 - st_a; st_s2 += st_g2 - st_g1; st_s3 += st_g3 - st_g2; }
#endif
    };

    const int nfull = anypad ? 0 : qA;                    // even; every tile s < qA is full for both blocks of this wave
    int s = 0;
    // Nothing may be in flight where a loop is entered: hipcc merges the VMEM scoreboards of the preheader and of the back
    // edge conservatively, and with prologue loads still pending it asked for vmcnt(0) INSIDE the loop -- i.e. it waited for the
    // dS export stores of the step before (store acknowledgements take microseconds) on every step.
    __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0)
    STAMP(st_loop);
    for (; s < nfull; s += 2) {
        step(s, std::integral_constant<int, 0>{}, std::false_type{});
        step(s + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
    STAMP(st_loop_end);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (; s < qA + 2; s += 2) {
        step(s, std::integral_constant<int, 0>{}, std::true_type{});
        step(s + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    for (; s < ntw; ++s) {
        if (s + 1 < ntw) {
            *(u32x4*)(smem + OFF_KR + ((s & 1) ^ 1) * TILE_BYTES + st_offR) = kreg;
            *(u32x4*)(smem + OFF_VR + ((s & 1) ^ 1) * TILE_BYTES + st_offR) = vreg;
        }
        prefetch_next();
        __syncthreads();
    }
    STAMP(st_epi);
#pragma unroll
    for (int X = 0; X < 2; ++X)
        store_rows_lds(dqkv + ((size_t)b * L + (qA + X) * 32) * ld + hd * 64, ld, dq[X][0], dq[X][1], lane, 0.125f,
                       smem + band_base + X * BAND_BYTES);
#ifdef MGX_B2_STAMP
    {
        STAMP(st_end);
        float* o = delta + ((size_t)b * heads + hd) * L + qA * 32;
        if (lane == 0) {
            o[0] = (float)(st_end - st_begin); o[1] = (float)(st_loop - st_begin); o[2] = (float)(st_loop_end - st_loop);
            o[3] = (float)(st_r1 + st_bar); o[4] = (float)st_bar; o[5] = (float)st_r2; o[6] = (float)(st_epi - st_loop_end);
            o[7] = (float)(st_end - st_epi); o[8] = (float)nfull; o[9] = (float)qb; o[10] = (float)w; o[11] = (float)st_s1; o[12] = (float)st_s2; o[13] = (float)st_s3;
        }
    }
#endif
}

static int bwd64_batch_group(int B, int L, int d) {
    const double per_row = (double)L * d * 2 * 5;
    int g = B;
    while (g > 1 && (g * per_row > 110e6 || B % g != 0)) --g;
    return g;
}

// launched by mgx_rel_attn_bwd_parts (rel_attn_bwd.hip) for L % 256 == 0 in place of rel_attn_dq_kernel<true, true>
int relattn::dq64_launch(const uint16_t* qkv, const void* EfA, const void* EfT, const uint32_t* padbits, const uint16_t* dctx,
                         const float* lse, float* delta, uint16_t* dqkv, uint16_t* dsrel, const uint16_t* ctx, int B, int L,
                         int d, void* stream) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)rel_attn_dq64_kernel,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, q2::LDS_BYTES);
    (void)attr;
    const int bg = bwd64_batch_group(B, L, d);
    MGX_REQUIRE((long)(L / 256) * (B / bg) <= 65535, MGX_ERR_SHAPE, "mgx_rel_attn_bwd: L/256 * batch groups too large");
    dim3 grid(bg * (d / 64), (L / 256) * (B / bg));
    hipLaunchKernelGGL(rel_attn_dq64_kernel, grid, dim3(256), q2::LDS_BYTES, (hipStream_t)stream, qkv, (const u32x4*)EfA,
                       (const u32x4*)EfT, padbits, dctx, lse, delta, dqkv, dsrel, ctx, L, d, bg);
    MGX_CHECK_LAUNCH("mgx_rel_attn_bwd(dQ, 64-row waves)");
    return MGX_OK;
}
