// Projection / FFN / vocabulary GEMMs of the hot path (layers.py:71-84,108,157-158; network.py:39)
// and their backward, all on v_mfma_f32_32x32x16_bf16 with 128 x 128 output tiles, 4 waves (2 x 2),
// each wave 64 x 64 = 2 x 2 MFMA tiles, XOR-swizzled LDS images, register-staged prefetch of the next
// reduction tile while the current one is multiplied (one barrier per tile), XCD-aware tile order.
//
//   forward  C  = act(A W^T + b)   "NT": A [M,K], W [N,K] both K-contiguous -> plain row fragments
//   dX       dX = dY W (o relu')   "NN": dY [M,N] row fragments; W [N,K] is read as [k=n][col] through
//                                        ds_read_b64_tr_b16 on a row-major LDS image (no W^T copy)
//   dW       gW += dY^T X          "TN": both operands have the reduction index (rows m) outermost:
//                                        both fragments come from transposed LDS reads; the M range is
//                                        split over workgroups and partial tiles are added with fp32
//                                        atomics (128-byte row segments) straight into the flat grad
//                                        buffer -- gradient accumulation across micro-batches for free.
//   bias     gb += column sums of dY, folded into the dW kernel (its first k-tile column stages those rows anyway).
//
// Two families: the 128 x 128 kernels described above (any shape), and the 256 x 256 "ring" kernels further down that the host
// entry points pick for the big projections of an encoder block -- LDS-DMA staging into a ring of stages, persistent workgroups --
// in two generations: eight waves with 128 x 64 wave tiles written in HIP (linear_ring_kernel, linear_dw_ring_kernel) and, round 5,
// four waves with 128 x 128 wave tiles whose main loops are generated gfx950 assembly owning all 256 accumulators of a wave
// (linear_ring4_kernel, linear_dw_ring4_kernel; gen_gemm_asm.py).  All produce bit-identical forward / dX results.
// A/B knobs of the GEMMs: gemm_knob() below.
#include <stdlib.h>
#include <type_traits>
#include "rel_attn_common.hpp"
#include "mgx.h"

// A/B knobs (MGX_GEMM_RING, MGX_RING4, MGX_DW_RING4, MGX_GEMM_SINGLE_BUF, MGX_DW_TARGET_WGS, MGX_DW_GROUP_WGS) exist in experiment builds only
// (`_build.py --variant NAME --experiments`, -DMGX_EXPERIMENTS=1): the product library reads no environment variable.
// tests/test_gpu_ring.py builds such a variant to run the ring and the 128 x 128 kernels on the same inputs.
#ifndef MGX_EXPERIMENTS
#define MGX_EXPERIMENTS 0
#endif
static inline int gemm_knob(const char* name, int unset) {
#if MGX_EXPERIMENTS
    const char* e = getenv(name);
    return e ? atoi(e) : unset;
#else
    (void)name;
    return unset;
#endif
}


using namespace relattn;

namespace {
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int IMG = BM * BK * 2;             // 16 KiB: 128 rows x 64 k (image R)  or  4 sub-tiles of 32 x 64 (image T)
constexpr int LDS_BYTES = 4 * IMG;           // two operands, double buffered = 64 KiB -> 2 workgroups / CU

MGX_DEV int xcd_remap(int bid, int nwg) {    // bijective: one XCD walks a contiguous run of tiles
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// natural-k transposed fragment (see rel_attn_bwd.hip): X[16*ks + 8*hh + j][32*ct + (lane&31)]
MGX_DEV bf16x8 fragTn(const char* tile, int lane, int ks, int ct) {
    const int i = lane & 15, g = lane >> 4, hh = lane >> 5;
    const int rq = i >> 2;
    const int chunk = 4 * ct + 2 * (g & 1) + ((i & 3) >> 1);
    const int byte_in = 8 * (i & 1);
    bf16x8 out;
#pragma unroll
    for (int jq = 0; jq < 2; ++jq) {
        const int row = 16 * ks + 8 * hh + 4 * jq + rq;
        bf16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + imgT_off(row, chunk) + byte_in));
        out[4 * jq + 0] = t[0]; out[4 * jq + 1] = t[1]; out[4 * jq + 2] = t[2]; out[4 * jq + 3] = t[3];
    }
    return out;
}

// Epilogue of a wave's 64 x 64 output block held as C^T tiles: acc[rt][ct][r] = C[mb + 32*rt + lane&31]
// [nb + 32*ct + crow(r,hh)].  Registers 4*g4 .. 4*g4+3 are 4 consecutive columns -> one 8-byte store.
// Optional fused bias (per column), ReLU, and ReLU-backward mask (zero where relu_y <= 0).  N % 4 == 0.
MGX_DEV void store_tileT(uint16_t* __restrict__ C, const uint16_t* __restrict__ relu_y, const f32x16 (&acc)[2][2],
                         const float* __restrict__ bias, int act, int mb, int nb, int M, int N, int l31, int hh,
                         const uint16_t* __restrict__ addend = nullptr) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int m = mb + 32 * rt + l31;
        if (m >= M) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = nb + 32 * ct + 8 * g4 + 4 * hh;
                if (n >= N) continue;
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = acc[rt][ct][4 * g4 + k];
                if (bias) {
                    const f32x4 bv = *(const f32x4*)(bias + n);
                    v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
                }
                if (act == 1) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                if (relu_y) {
                    const u32x2 y = *(const u32x2*)(relu_y + (size_t)m * N + n);
                    if (!(bf16lo(y.x) > 0.f)) v[0] = 0.f;
                    if (!(bf16hi(y.x) > 0.f)) v[1] = 0.f;
                    if (!(bf16lo(y.y) > 0.f)) v[2] = 0.f;
                    if (!(bf16hi(y.y) > 0.f)) v[3] = 0.f;
                }
                if (addend) {                  // residual-branch gradient joins here (saves an elementwise pass)
                    const u32x2 a = *(const u32x2*)(addend + (size_t)m * N + n);
                    v[0] += bf16lo(a.x); v[1] += bf16hi(a.x); v[2] += bf16lo(a.y); v[3] += bf16hi(a.y);
                }
                u32x2 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *(u32x2*)(C + (size_t)m * N + n) = o;
            }
        }
    }
}

// Same epilogue through LDS, for N % 8 == 0: the direct form above writes 16-byte pieces of 32 different rows per
// wave instruction (8x more L2 write requests than lines; measured 14 us of a 37 us 32768x512x512 projection).
// Here the wave parks 32 output rows at a time in its own 4.6 KB LDS patch (144-byte rows: conflict-free 8-byte
// writes from the accumulator layout) and reads them back row-major, so each global store instruction writes
// 8 full 128-byte row segments, and the ReLU mask / residual addend are fetched the same coalesced way.
// `patch` must not alias LDS another wave may still read: callers pass a barrier first.
constexpr int EPI_STRIDE = 144;
constexpr int EPI_PATCH = 32 * EPI_STRIDE;                  // 4,608 B per wave
MGX_DEV void store_tile_lds(uint16_t* __restrict__ C, const uint16_t* __restrict__ relu_y,
                            const uint16_t* __restrict__ addend, const f32x16 (&acc)[2][2],
                            const float* __restrict__ bias, int act, int mb, int nb, int M, int N, int lane,
                            char* patch) {
    const int l31 = lane & 31, hh = lane >> 5;
    const int rr = lane >> 3, ch = lane & 7;                // read-back: rows rr + 8 i, 16-byte chunk ch
    const int n = nb + ch * 8;
    float bv[2][4][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int nn = nb + 32 * ct + 8 * g4 + 4 * hh;
            f32x4 b = {0.f, 0.f, 0.f, 0.f};
            if (bias && nn < N) b = *(const f32x4*)(bias + nn);
            bv[ct][g4][0] = b.x; bv[ct][g4][1] = b.y; bv[ct][g4][2] = b.z; bv[ct][g4][3] = b.w;
        }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    v[k] = acc[rt][ct][4 * g4 + k] + bv[ct][g4][k];
                    if (act == 1) v[k] = fmaxf(v[k], 0.f);
                }
                *(u32x2*)(patch + l31 * EPI_STRIDE + (32 * ct + 8 * g4 + 4 * hh) * 2) =
                    u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rr + 8 * i;
            const int m = mb + 32 * rt + row;
            u32x4 o = *(const u32x4*)(patch + row * EPI_STRIDE + ch * 16);
            if (m < M && n < N) {
                const size_t off = (size_t)m * N + n;
                if (relu_y || addend) {
                    float f[8];
                    unpack8(o, f);
                    if (relu_y) {
                        float y[8];
                        unpack8(*(const u32x4*)(relu_y + off), y);
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] = (y[k] > 0.f) ? f[k] : 0.f;
                    }
                    if (addend) {
                        float a[8];
                        unpack8(*(const u32x4*)(addend + off), a);
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] += a[k];
                    }
                    o = pack8(f);
                }
                *(u32x4*)(C + off) = o;
            }
        }
        wave_lds_fence();
    }
}

MGX_DEV void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero16();
}
}  // namespace

// =================================================================================================
// forward (NT)
// =================================================================================================
template <bool DBUF>
__global__ __launch_bounds__(256, DBUF ? 2 : 3) void linear_fwd_kernel(const uint16_t* __restrict__ A,
                                                            const uint16_t* __restrict__ W,
                                                            const float* __restrict__ bias,
                                                            uint16_t* __restrict__ C, int M, int N, int K, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, ntm * ntn);
    const int tn = bid % ntn, tm = bid / ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const int srow = tid >> 3, sch = tid & 7;
    u32x4 areg[4], wreg[4];
    // Loads are unconditional: rows beyond M / N are clamped into range (their products land in accumulator rows / columns
    // that the epilogue never stores).  A load under a per-lane condition costs an exec-mask branch and a zero-fill per
    // load and makes the compiler wait for the whole VMEM queue where the paths rejoin.
    const uint16_t* ap[4];
    const uint16_t* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = srow + 32 * i;
        ap[i] = A + (size_t)min(m0 + row, M - 1) * K + sch * 8;
        wp[i] = W + (size_t)min(n0 + row, N - 1) * K + sch * 8;
    }
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            areg[i] = *(const u32x4*)(ap[i] + k0);
            wreg[i] = *(const u32x4*)(wp[i] + k0);
        }
    };
    auto store_tiles = [&](int buf) {
        char* at = smem + buf * 2 * IMG;
        char* wt = at + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            *(u32x4*)(at + imgR_off(row, sch)) = areg[i];
            *(u32x4*)(wt + imgR_off(row, sch)) = wreg[i];
        }
    };
    f32x16 acc[2][2];
    zero_acc(acc);
    auto multiply = [&](int cur) {
        const char* at = smem + cur * 2 * IMG;
        const char* wt = at + IMG;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a0 = frag_R(at, 64 * wm + l31, hh, ks);
            const bf16x8 a1 = frag_R(at, 64 * wm + 32 + l31, hh, ks);
            const bf16x8 b0 = frag_R(wt, 64 * wn + l31, hh, ks);
            const bf16x8 b1 = frag_R(wt, 64 * wn + 32 + l31, hh, ks);
            // swapped operands: acc[rt][ct] holds C^T (rows = n on registers, column = m on the lane), so
            // 4 consecutive registers are 4 consecutive n of one output row -> 8-byte stores
            acc[0][0] = mfma(b0, a0, acc[0][0]);
            acc[0][1] = mfma(b1, a0, acc[0][1]);
            acc[1][0] = mfma(b0, a1, acc[1][0]);
            acc[1][1] = mfma(b1, a1, acc[1][1]);
        }
    };
    const int nk = K / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    // all reduction tiles but the last: branch-free body (prefetch the next tile, multiply the current one, publish)
    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = DBUF ? (kt & 1) : 0;
        load_tiles((kt + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch AHEAD of the MFMAs (the scheduler sinks it to the barrier)
        multiply(cur);
        if (!DBUF) __syncthreads();                 // single buffer: everyone has read the tile before it is replaced
        store_tiles(DBUF ? (cur ^ 1) : 0);
        __syncthreads();
    }
    multiply(DBUF ? ((nk - 1) & 1) : 0);
    __syncthreads();
    // (the loop's last barrier has passed: no wave reads the tile buffers any more)
    if ((N & 7) == 0)
        store_tile_lds(C, nullptr, nullptr, acc, bias, act, m0 + 64 * wm, n0 + 64 * wn, M, N, lane, smem + w * EPI_PATCH);
    else
        store_tileT(C, nullptr, acc, bias, act, m0 + 64 * wm, n0 + 64 * wn, M, N, l31, hh);
}

// =================================================================================================
// Ring kernel: C[M,NO] = A[M,R] . B^T  for the big projections (forward: B = W [NO,R]; dX: B = W [R,NO], BTRANS).
//   * 256 x 256 output tile per workgroup, 8 waves (2 x 4), wave tile 128 x 64 (8 accumulator tiles): half the operand
//     bytes per flop of the 128 x 128 kernels above.  Measured on the QKV projection (M = 65,536): those kernels spend
//     108 of their 145 us just moving 1.6 GB of operand tiles from L2 into LDS.
//   * operands go global -> LDS by DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no VGPR staging, no ds_write).
//     The LDS destination of a DMA instruction is linear in the lane, so the bank swizzle of an image is applied on the
//     SOURCE side: lane l of piece p fetches the 16 bytes that belong in slot 64 p + l.
//   * reduction steps of 32, a ring of 4 stages (A 256 rows x 64 B + B 16 KB = 32 KB each): the request for step g+4 is
//     made when step g's stage is released and is waited for three steps later with a COUNTED s_waitcnt (never 0 in the
//     steady state) -- with two 64-wide stages the request had one step to land and the waves waited 1,400 cycles per step.
//   * persistent workgroups (one per CU) walk their tiles as ONE stream of reduction steps: the DMA ring runs across tile
//     boundaries, so a tile's epilogue overlaps the next tile's first requests.
//   * inside a step every MFMA is followed by one fragment read of the NEXT block or one DMA piece (a wave issues in
//     order: a group of reads or DMA issues ahead of the MFMAs holds them back for ~100-300 cycles per block).
//   LDS: 4 x 32 KB stages + 8 x 4 KB epilogue patches = 160 KB.
// =================================================================================================
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;
MGX_DEV void glds16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)lds_wave_base, 16, 0, 0);
}
MGX_DEV void glds4(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((glb_void_ptr)g, (lds_void_ptr)lds_wave_base, 4, 0, 0);
}
template <int N> MGX_DEV void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int RG_STAGE = 32768, RG_NST = 4, RG_PATCH = 4096;
constexpr int RG_LDS = RG_NST * RG_STAGE + 8 * RG_PATCH;
// image H (64-byte rows): 16-byte chunk c of row r at r*64 + ((c ^ ((r >> 2) & 3)) << 4): conflict-free ds_read_b128 of
// (row = lane & 31, chunk = 2 ks + hh), and one DMA wave instruction = 16 whole rows.
MGX_DEV int imgH_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// Epilogue of one wave: the C^T accumulators (n on registers, m on lanes) of its 128 (m) x 64 (n) block -> row-major bf16
// through a 4 KB swizzled patch (32 rows x 128 B, 16-byte chunk c of row r at chunk c ^ (r & 7)): every global store
// instruction writes 8 full 128-byte row segments.  bias / ReLU on the accumulator side; ReLU-backward mask and residual
// addend on the row-major side (coalesced loads).  The block lies inside the matrix: 16 unconditional stores.
typedef short s16x2 __attribute__((ext_vector_type(2)));
// max(x, 0) on a packed bf16 pair: as 16-bit integers a negative bf16 (-0 included) is negative and a positive one keeps its order
// (v_pk_max_i16) -- the same result as v_max_f32 before the conversion, one instruction per pair instead of two per element
MGX_DEV uint32_t relu_bf16x2(uint32_t p) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), (s16x2)(0)));
}
// FWD: bias / ReLU (forward projection); !FWD: ReLU-backward mask / residual addend (dX).  The variants a kernel cannot take are
// compiled out and `act` selects between two straight-line bodies: the epilogue used to be ~1,500 instructions per wave (per-element
// v_max + v_cndmask on the runtime `act`, both operand paths) -- 5.5 K cycles per tile with the MFMA pipe idle, 17 % of a K = 512
// tile and 30 % of a K = 256 one (tools/ring_stamp.py)
template <bool FWD, int PRE = 0>     // PRE (dX only): 0 plain, 1 ReLU-backward mask, 2 residual addend -- straight-line variants: across a
                                     // runtime branch hipcc's wait for the prefetched rows becomes vmcnt(0) again
MGX_DEV void store_wave_block(uint16_t* __restrict__ C, const uint16_t* __restrict__ relu_y,
                              const uint16_t* __restrict__ addend, f32x16 (&acc)[4][2], bool bias, int act, int mb,
                              int nb, int N, int lane, char* patch) {
    const int l31 = lane & 31, hh = lane >> 5;
    const int rr = lane >> 3, ch = lane & 7;
    // bias: the wave's 64 values were put into its patch by DMA a tile ago (a vector load here would be waited for with
    // the whole DMA ring ahead of it in the in-order VMEM queue); added in place before the patch is reused
    if (FWD && bias) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 b = *(const f32x4*)(patch + (32 * ct + 8 * g4 + 4 * hh) * 4);
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) {
                    acc[rt][ct][4 * g4 + 0] += b.x; acc[rt][ct][4 * g4 + 1] += b.y;
                    acc[rt][ct][4 * g4 + 2] += b.z; acc[rt][ct][4 * g4 + 3] += b.w;
                }
            }
        wave_lds_fence();
    }
    char* wr = patch + l31 * 128 + 8 * hh;
    const int sw = l31 & 7;
    auto park = [&](int rt, auto relu_tag) {                 // 32 rows of the block -> the patch, row-major
        constexpr bool RELU = decltype(relu_tag)::value;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint32_t p0 = pack_bf16x2(acc[rt][ct][4 * g4 + 0], acc[rt][ct][4 * g4 + 1]);
                uint32_t p1 = pack_bf16x2(acc[rt][ct][4 * g4 + 2], acc[rt][ct][4 * g4 + 3]);
                if (RELU) { p0 = relu_bf16x2(p0); p1 = relu_bf16x2(p1); }
                *(u32x2*)(wr + (((4 * ct + g4) ^ sw) << 4)) = u32x2{p0, p1};
            }
    };
    const bool relu = FWD && act == 1;
    uint16_t* crow = C + (size_t)(mb + rr) * N + nb + ch * 8;
    if constexpr (!FWD && PRE != 0) {
        // dX with a mask / addend.  Loaded where they are used, the rows of a 32-row slice were waited for with vmcnt(0) -- behind the
        // previous slice's four stores, a full store round trip per slice: 19-32 K cycles per tile instead of 4 K (tools/ring_stamp.py),
        // +27 % on the dX of QKV, +80 % on the dX of FFN_pre.  Now all sixteen rows of the tile are requested at once, after the
        // accumulators have been packed to bf16 (64 registers instead of 128: what makes room for 64 registers of rows in flight),
        // and waited for once: 11-17 K cycles.  What is left is bandwidth, not latency: every workgroup reaches its epilogue at the
        // same time, and 32 MB of rows in + 32 MB of tile out per round of tiles is ~12 us of HBM on its own (with the operand
        // these K <= 512 GEMMs sit at 1.4-1.5 x their HBM floors).  (The host sends a call with BOTH operands to the 128 x 128
        // kernel; the training step never makes one.)
        const uint16_t* prow = (PRE == 2 ? addend : relu_y) + (size_t)(mb + rr) * N + nb + ch * 8;
        u32x4 pre[4][4];
        u32x2 pk[4][8];
        auto fetch = [&](int rt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[rt][i] = *(const u32x4*)(prow + (size_t)(32 * rt + 8 * i) * N);
        };
        __builtin_amdgcn_sched_barrier(0);
        fetch(0); fetch(1);                                  // (hipcc moves these below the packing whatever is put between them)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    pk[rt][4 * ct + g4] = u32x2{pack_bf16x2(acc[rt][ct][4 * g4 + 0], acc[rt][ct][4 * g4 + 1]),
                                                pack_bf16x2(acc[rt][ct][4 * g4 + 2], acc[rt][ct][4 * g4 + 3])};
        __builtin_amdgcn_sched_barrier(0);
        fetch(2); fetch(3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) *(u32x2*)(wr + ((c8 ^ sw) << 4)) = pk[rt][c8];
            wave_lds_fence();
            u32x4 o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + 8 * i;
                o[i] = *(const u32x4*)(patch + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float f[8], a[8];
                unpack8(o[i], f);
                unpack8(pre[rt][i], a);
                if (PRE == 1) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] = (a[k] > 0.f) ? f[k] : 0.f;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) f[k] += a[k];
                }
                o[i] = pack8(f);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(crow + (size_t)(32 * rt + 8 * i) * N) = o[i];
            wave_lds_fence();
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            if (relu) park(rt, std::true_type{}); else park(rt, std::false_type{});
            wave_lds_fence();
            u32x4 o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + 8 * i;
                o[i] = *(const u32x4*)(patch + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#ifdef MGX_RING_PEEL_STORE            // diagnostic: the stores of the epilogue left out (the kernel's result is then garbage)
            if (o[0].x == 0x12345678u && o[1].y == 0x9abcdef0u)
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)(crow + (size_t)(32 * rt + 8 * i) * N) = o[i];
            wave_lds_fence();
        }
    }
}

#define MGX_SB() __builtin_amdgcn_sched_barrier(0)
#ifndef MGX_RING_ALWAYS
#define MGX_RING_ALWAYS 1
#endif
#ifndef MGX_RING_PEEL
// diagnostic builds only (results are garbage): 1 no barrier in the loop, 2 no DMA in the loop, 4 no fragment reads.  Round 4, per step of
// cfg2's forward GEMMs at batch 64 (tools/ab_gemm.sh): product 2.71 ms; no barrier 2.70; no DMA 2.31; no fragment reads 2.33; neither
// 1.83; all three 1.64 (= the MFMAs, the epilogue and the loop: 1.25 PF).  The barrier is free; the DMA pieces and the fragment reads
// cost 15 % each and add up -- the waves' in-order issue behind the LDS pipe, not its bandwidth (12 reads + 4 pieces per wave and step)
#define MGX_RING_PEEL 0
#endif
template <bool BTRANS>
__global__ __launch_bounds__(512, 1) void linear_ring_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                            const float* __restrict__ bias,
                                                            const uint16_t* __restrict__ relu_y,
                                                            const uint16_t* __restrict__ addend,
                                                            uint16_t* __restrict__ C, int M, int NO, int R, int act) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ntn = (NO + 255) / 256, ntm = (M + 255) / 256, ntiles = ntm * ntn;
    const int nh = R / 32;                                   // reduction steps per tile
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = my_tiles * nh;                             // steps of this workgroup, over all its tiles
    char* patch = smem + RG_NST * RG_STAGE + w * RG_PATCH;
    if (G <= 0) return;
    auto tile_origin = [&](int i, int& m0, int& n0) {        // i-th tile of this workgroup
        const int t = min(xcd_remap((int)blockIdx.x + i * (int)gridDim.x, ntiles), ntiles - 1);   // (a request past the last tile re-reads it)
        m0 = (t / ntn) * 256; n0 = (t % ntn) * 256;
    };

    // ---- DMA stream: wave w stages pieces 2w, 2w+1 of the A image and of the B image of every step ----
    //  A piece p: rows 16p .. 16p+15 of the tile (64 bytes each).
    //  B piece p, !BTRANS: the same for the rows of B;  BTRANS: the step's B tile is [32 r][256 n] = 4 sub-tiles [32][64]
    //  (image T, 128-byte rows): piece p = rows 8 (p & 3) .. +7 of sub-tile p >> 2.
    const uint16_t* ap[2];
    const uint16_t* bp[2];
    int d_i = -1, d_h = 0, d_st = 0, d_k0 = 0;
    char* d_at = nullptr;
    auto dma_begin = [&]() {                                 // addresses of the next request
        if (d_h == 0) {
            ++d_i;
            int m0, n0;
            tile_origin(d_i, m0, n0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = 2 * w + j;
                const int row = 16 * p + (lane >> 2);
                ap[j] = A + (size_t)min(m0 + row, M - 1) * R + ((lane & 3) ^ ((row >> 2) & 3)) * 8;
                if (!BTRANS) {
                    bp[j] = B + (size_t)min(n0 + row, NO - 1) * R + ((lane & 3) ^ ((row >> 2) & 3)) * 8;
                } else {
                    const int r = 8 * (p & 3) + (lane >> 3);
                    const int chunk = (lane & 7) ^ (((r >> 1) & 1) << 2);
                    bp[j] = B + (size_t)r * NO + min(n0 + 64 * (p >> 2) + chunk * 8, NO - 8);
                }
            }
        }
        d_at = smem + d_st * RG_STAGE + (2 * w) * 1024;
        d_k0 = d_h * 32;
        d_h = (d_h + 1 == nh) ? 0 : d_h + 1;
        d_st = (d_st + 1) & 3;
    };
    auto b_src = [&](int j) { return BTRANS ? bp[j] + (size_t)d_k0 * NO : bp[j] + d_k0; };
    auto dma_all = [&]() {
        dma_begin();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            glds16(ap[j] + d_k0, d_at + j * 1024);
            glds16(b_src(j), d_at + 16384 + j * 1024);
        }
    };
    bf16x8 fa[2][4], fb[2][2];
    auto rd_a = [&](int stg, int ks, int i) {
        return *(const bf16x8*)(smem + stg * RG_STAGE + imgH_off(128 * wm + 32 * i + l31, 2 * ks + hh));
    };
    auto rd_b = [&](int stg, int ks, int i) {      // !BTRANS
        return *(const bf16x8*)(smem + stg * RG_STAGE + 16384 + imgH_off(64 * wn + 32 * i + l31, 2 * ks + hh));
    };
    // BTRANS: the B fragments are transposed reads (ds_read_b64_tr_b16 x 2) of sub-tile wn of the step's [32 r][256 n] tile.
    // Through the builtin the compiler puts s_waitcnt vmcnt(0) in front of every such read while a DMA is in flight (it
    // cannot tell the read from the DMA's LDS destination), which drains the ring twice per block: the reads are issued
    // from inline asm instead.  The compiler does not count them, so (a) every block ends with an explicit
    // s_waitcnt lgkmcnt(0) -- before any control flow, where register copies could be placed -- and (b) the two 64-bit halves
    // are only joined into an operand after that wait.  tb[ct]: the lane's byte address of fragTn(sub-tile wn, ks = 0,
    // column half ct) in stage 0 (see fragTn: row = 16 ks + 8 hh + 4 jq + rq; ks and jq are the immediate offset).
    uint32_t tb[2] = {0u, 0u};
    u32x2 hb[2][2][2];                                        // [set][ct][jq]
    if (BTRANS) {
        const int i15 = lane & 15, gq = lane >> 4, rq = i15 >> 2;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int chunk = 4 * ct + 2 * (gq & 1) + ((i15 & 3) >> 1);
            tb[ct] = lds_addr_of(smem) + 16384 + wn * TILE_BYTES + (8 * hh + rq) * 128 +
                     ((chunk ^ (((rq >> 1) & 1) << 2)) << 4) + 8 * (i15 & 1);
        }
    }
    auto rd_bt = [&](int stg, auto ks_tag, int ct, u32x2 (&h)[2]) {
        constexpr int KS = decltype(ks_tag)::value;
        const uint32_t addr = tb[ct] + stg * RG_STAGE;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h[0]) : "v"(addr), "n"(2048 * KS));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h[1]) : "v"(addr), "n"(2048 * KS + 512));
    };
    auto join = [&](const u32x2 (&h)[2]) { return __builtin_bit_cast(bf16x8, u32x4{h[0].x, h[0].y, h[1].x, h[1].y}); };
    // one block: the 8 MFMAs of fragment set `cur`; each gap carries one fragment read of the NEXT block (into set cur ^ 1)
    // or one DMA piece
    auto block = [&](f32x16 (&acc)[4][2], auto cur_tag, int nstg, auto nks_tag, const uint16_t* g0, char* l0, const uint16_t* g1, char* l1, bool on, auto first_tag) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1, NKS = decltype(nks_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value;   // a tile's first block: C = 0 (an inline constant) instead of 128 v_mov per tile
        const bf16x8(&a)[4] = fa[CUR];
        bf16x8(&na)[4] = fa[NXT];
        bf16x8 b[2];
        if (BTRANS) { b[0] = join(hb[CUR][0]); b[1] = join(hb[CUR][1]); }
        else { b[0] = fb[CUR][0]; b[1] = fb[CUR][1]; }
        MGX_SB();
        // reads in the first three gaps (two per gap), DMA pieces in gaps 3 and 5: by the end of the block the reads have had
        // five MFMAs to return
        // (round 4: which gaps carry the pieces -- 1/3, 3/5, 5/7 -- makes no difference, and neither did spreading the eight waves'
        //  pieces over all eight gaps, which only cost the scalar branches)
#define MGX_GAP(i) do { MGX_SB(); if (!(MGX_RING_PEEL & 2) && (MGX_RING_ALWAYS || on) && (i) == 3) glds16(g0, l0); if (!(MGX_RING_PEEL & 2) && (MGX_RING_ALWAYS || on) && (i) == 5) glds16(g1, l1); MGX_SB(); } while (0)
        acc[0][0] = mfma(b[0], a[0], FIRST ? zero16() : acc[0][0]); MGX_SB();
        if (!(MGX_RING_PEEL & 4)) {
            na[0] = rd_a(nstg, NKS, 0);
            if (BTRANS) rd_bt(nstg, nks_tag, 0, hb[NXT][0]); else fb[NXT][0] = rd_b(nstg, NKS, 0);
        }
        MGX_GAP(0);
        acc[0][1] = mfma(b[1], a[0], FIRST ? zero16() : acc[0][1]); MGX_SB();
        if (!(MGX_RING_PEEL & 4)) {
            na[1] = rd_a(nstg, NKS, 1);
            if (BTRANS) rd_bt(nstg, nks_tag, 1, hb[NXT][1]); else fb[NXT][1] = rd_b(nstg, NKS, 1);
        }
        MGX_GAP(1);
        acc[1][0] = mfma(b[0], a[1], FIRST ? zero16() : acc[1][0]); MGX_SB();
        if (!(MGX_RING_PEEL & 4)) {
            na[2] = rd_a(nstg, NKS, 2);
            na[3] = rd_a(nstg, NKS, 3);
        }
        MGX_GAP(2);
        acc[1][1] = mfma(b[1], a[1], FIRST ? zero16() : acc[1][1]); MGX_GAP(3);
        acc[2][0] = mfma(b[0], a[2], FIRST ? zero16() : acc[2][0]); MGX_GAP(4);
        acc[2][1] = mfma(b[1], a[2], FIRST ? zero16() : acc[2][1]); MGX_GAP(5);
        acc[3][0] = mfma(b[0], a[3], FIRST ? zero16() : acc[3][0]); MGX_GAP(6);
        acc[3][1] = mfma(b[1], a[3], FIRST ? zero16() : acc[3][1]); MGX_GAP(7);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every LDS read of the block has returned (asm reads included)
        MGX_SB();
    };
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;

    // bias of tile i -> the wave's patch (64 floats): one more entry in the in-order VMEM queue, issued when the patch is
    // free (right after the previous tile's epilogue); the counted waits below then leave at most one operation fewer
    // outstanding than they could, which is always safe
    auto dma_bias = [&](int i) {
        int m0, n0;
        tile_origin(i, m0, n0);
        glds4(bias + min(n0 + 64 * wn + lane, NO - 1), patch);
    };
    // prologue: requests 0..3 (a stream shorter than that simply waits for everything)
    if (bias) dma_bias(0);
    dma_all();
    if (G > 1) dma_all();
    if (G > 2) dma_all();
    if (G > 3) dma_all();
    if (G > 3) wait_vmcnt<12>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[0][i] = rd_a(0, 0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (BTRANS) rd_bt(0, T0{}, i, hb[0][i]); else fb[0][i] = rd_b(0, 0, i);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MGX_SB();
    int g = 0, since_epi = 2, cs = 0;
    bool pend = false;                                       // the B pieces of the newest request are still to be issued
    // (ALWAYS: block 1 of step 0 would issue B pieces of a request that was never begun -- the prologue's fourth, whose pieces are
    //  complete: d_at / b_src still describe it, so the pieces are fetched once more into the same place)
#ifdef MGX_RING_STAMP
    // diagnostic build only (tools/ring_stamp.py): s_memtime sums per phase, left by lane 0 of every wave in the first bytes of C
    unsigned long long st_acc[5] = {0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_t0 = st_last, st_r0 = __builtin_amdgcn_s_memrealtime();
#define RING_STAMP(i) do { MGX_SB(); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); MGX_SB(); st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define RING_STAMP(i)
#endif
    // one reduction step of the stream (g counts them over all tiles of the workgroup)
    auto step = [&](f32x16 (&acc)[4][2], auto first_tag) {
        const int ns = (cs + 1) & 3;
        // block 1: multiply (stage cs, k 0..15); its gaps read (cs, k 16..31) and issue the B pieces of the request made at
        // the last barrier
        block(acc, T0{}, cs, T1{}, b_src(0), d_at + 16384, b_src(1), d_at + 16384 + 1024, pend, first_tag);
        RING_STAMP(0);
        // (the block ended with lgkmcnt(0): this wave has read everything it needs from stage cs)
        // step g+1 has landed once at most the younger operations are outstanding: requests g+2 and g+3 (4 each) and,
        // for two steps after a tile's epilogue, its 16 stores
        if (g + 3 < G) { if (since_epi < 2) wait_vmcnt<24>(); else wait_vmcnt<8>(); }
        else wait_vmcnt<0>();
        RING_STAMP(1);
        if (!(MGX_RING_PEEL & 1)) __builtin_amdgcn_s_barrier();
        RING_STAMP(2);
        pend = MGX_RING_ALWAYS || (g + 4 < G);                // (ALWAYS: past the end of the stream the pieces re-read the last tile into a
        if (pend) dma_begin();                                //  stage nobody reads again, instead of four scalar branches per step)
        // block 2: multiply (cs, k 16..31); its gaps read (ns, k 0..15) and issue the A pieces of the new request
        block(acc, T1{}, ns, T0{}, ap[0] + d_k0, d_at, ap[1] + d_k0, d_at + 1024, pend, std::false_type{});
        RING_STAMP(3);
        ++since_epi;
        ++g;
        cs = ns;
    };
    // The accumulators live inside the tile loop: after the epilogue has read them they are dead, and the compiler knows it (as
    // one flat loop over steps with a runtime "first step of a tile" test it kept all 128 alive across the epilogue).
    for (int ti = 0; ti < my_tiles; ++ti) {
        f32x16 acc[4][2];
        step(acc, std::true_type{});                         // the tile's first block multiplies with C = 0
        for (int h = 1; h < nh; ++h) step(acc, std::false_type{});
        int m0, n0;
        tile_origin(ti, m0, n0);
        // (the host only takes this kernel for M % 256 == 0 and NO % 256 == 0: every tile is whole, 16 unconditional
        //  stores per wave -- the count the waits above rely on)
        if (BTRANS && addend) store_wave_block<!BTRANS, 2>(C, relu_y, addend, acc, false, 0, m0 + 128 * wm, n0 + 64 * wn, NO, lane, patch);
        else if (BTRANS && relu_y) store_wave_block<!BTRANS, 1>(C, relu_y, addend, acc, false, 0, m0 + 128 * wm, n0 + 64 * wn, NO, lane, patch);
        else store_wave_block<!BTRANS, 0>(C, relu_y, addend, acc, bias != nullptr, act, m0 + 128 * wm, n0 + 64 * wn, NO, lane, patch);
        since_epi = 0;
        if (bias && ti + 1 < my_tiles) dma_bias(ti + 1);
        // the fragments block 2 has just prefetched for the next step are read AGAIN here instead of being kept across the
        // epilogue (48 registers the epilogue's prefetch of the mask / addend rows needs; ~150 cycles per tile)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[0][i] = rd_a(cs, 0, i);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (BTRANS) rd_bt(cs, T0{}, i, hb[0][i]); else fb[0][i] = rd_b(cs, 0, i);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RING_STAMP(4);
    }
    if (MGX_RING_ALWAYS) wait_vmcnt<0>();                    // the pieces requested past the end land before the workgroup's LDS is released
#ifdef MGX_RING_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
        float* rec = (float*)C + ((size_t)blockIdx.x * 8 + w) * 16;
        for (int i = 0; i < 5; ++i) rec[i] = (float)st_acc[i];
        rec[5] = (float)G; rec[6] = (float)my_tiles; rec[7] = (float)(__builtin_amdgcn_s_memtime() - st_t0);
        rec[8] = (float)(__builtin_amdgcn_s_memrealtime() - st_r0); rec[9] = (float)w;
    }
#endif
}

// =================================================================================================
// The same ring with FOUR waves, one per SIMD, each holding a 128 x 128 block of the tile = 16 accumulator tiles = all 256 AGPRs
// (round 5).  Per 16-column k-step a wave reads 8 operand fragments for 16 MFMAs instead of 6 for 8: two thirds of the LDS bytes per
// MFMA.  hipcc cannot keep 256 accumulators in place (rounds 3-4), so one TILE's stages are one generated asm statement that owns them
// (gen_gemm_asm.py: ring_tile -> linear_ring4_loop.inc); what stays HIP is the tile loop, the wave's parameter block in LDS (source
// pointers of this and the next tile, the ring's state between two statements) and the epilogue.  The DMA ring runs on across the
// statement's end: the first three stages of the next tile are in flight during the epilogue, whose 32 global stores per wave
// (MGX_RING4_EPI_STORES) the statement's first counted waits allow for.  Same images, same MFMA operand order as the eight-wave kernel:
// bit-identical results.  Stages of 64 reduction columns in two 64 KB slots, so that every DMA instruction fetches whole 128-byte lines
// (gen_gemm_asm.py).  LDS: 2 x 64 KB stages + 4 x 4 KB patches + 4 x 4 KB parameter blocks (bias at + 512) = 160 KB.
// =================================================================================================
#if defined(MGX_GEMM_DIAG) && MGX_GEMM_DIAG
#include "linear_ring4_loop_diag.inc"      // timing-only loops of a diagnostic build (gen_gemm_asm.py with MGX_RING4_DIAG / MGX_DW4_NO*)
#else
#include "linear_ring4_loop.inc"
#endif
#ifdef MGX_DW4_TIMES
extern __device__ unsigned long long mgx_dw4_times_buf[8 * 1024];
#endif
template <bool FWD, int PRE>
MGX_DEV void store_wave_block4(uint16_t* __restrict__ C, const uint16_t* __restrict__ relu_y, const uint16_t* __restrict__ addend,
                               f32x16 (&acc)[4][4], const char* bias_lds, int act, int mb, int nb, int N, int lane, char* patch) {
    const int l31 = lane & 31, hh = lane >> 5;
    const int rr = lane >> 3, ch = lane & 7;
    char* wr = patch + l31 * 128 + 8 * hh;
    const int sw = l31 & 7;
    const bool relu = FWD && act == 1;
#pragma unroll
    for (int half = 0; half < 2; ++half) {                   // 64 columns at a time: the patch holds 32 rows x 64 columns
        // byte offsets from the (uniform) matrix bases in 32 bits (host: the matrix is smaller than 4 GB): sixteen 64-bit row pointers
        // per operand cost 64 registers and spilled
        const uint32_t off0 = (uint32_t)(((size_t)(mb + rr) * N + nb + 64 * half + ch * 8) * 2), rowb = (uint32_t)N * 16u;   // 8 rows
        u32x4 pre[4][4];
        if constexpr (!FWD && PRE != 0) {
            // all sixteen rows of the half-block at once, waited for once (store_wave_block)
            const char* pbase = (const char*)(PRE == 2 ? addend : relu_y);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int i = 0; i < 4; ++i) pre[rt][i] = *(const u32x4*)(pbase + (off0 + (uint32_t)(4 * rt + i) * rowb));
            __builtin_amdgcn_sched_barrier(0);
        }
        f32x4 b[2][4];                                       // bias of the lane's columns (parameter block + 512: the statement's DMA)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) b[ct][g4] = FWD ? *(const f32x4*)(bias_lds + (64 * half + 32 * ct + 8 * g4 + 4 * hh) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            // (packed one 32-row slice at a time: the accumulators stay where they are -- AGPRs -- until they are read here.  ReLU as a
            //  straight-line variant: on the runtime flag hipcc computed both and selected, 4 more instructions per 4 values)
            auto park = [&](auto relu_tag) {
                constexpr bool RELU = decltype(relu_tag)::value;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const f32x16& a = acc[rt][2 * half + ct];
                        uint32_t p0 = pack_bf16x2(a[4 * g4 + 0] + b[ct][g4].x, a[4 * g4 + 1] + b[ct][g4].y);
                        uint32_t p1 = pack_bf16x2(a[4 * g4 + 2] + b[ct][g4].z, a[4 * g4 + 3] + b[ct][g4].w);
                        if (RELU) { p0 = relu_bf16x2(p0); p1 = relu_bf16x2(p1); }
                        *(u32x2*)(wr + (((4 * ct + g4) ^ sw) << 4)) = u32x2{p0, p1};
                    }
            };
            if (relu) park(std::true_type{}); else park(std::false_type{});
            wave_lds_fence();
            u32x4 o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + 8 * i;
                o[i] = *(const u32x4*)(patch + row * 128 + ((ch ^ (row & 7)) << 4));
            }
            if constexpr (!FWD && PRE != 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float f[8], a[8];
                    unpack8(o[i], f);
                    unpack8(pre[rt][i], a);
                    if (PRE == 1) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] = (a[k] > 0.f) ? f[k] : 0.f;
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) f[k] += a[k];
                    }
                    o[i] = pack8(f);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) *(u32x4*)((char*)C + (off0 + (uint32_t)(4 * rt + i) * rowb)) = o[i];
            wave_lds_fence();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <bool BTRANS, int PRE>      // PRE (dX): 0 plain, 1 ReLU-backward mask, 2 residual addend -- ONE epilogue per kernel: with the three behind
                                     // runtime branches hipcc moved accumulator tiles between AGPR tuples after the statement and spilled
__global__ __launch_bounds__(256, 1) void linear_ring4_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                             const float* __restrict__ bias, const uint16_t* __restrict__ relu_y,
                                                             const uint16_t* __restrict__ addend, uint16_t* __restrict__ C, int M, int NO,
                                                             int R, int act) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ntn = NO / 256, ntm = M / 256, ntiles = ntm * ntn;           // whole tiles (host)
    const int nd = R / 64;                                   // 64-column stages per tile: even, >= 4 (host)
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    if (my_tiles <= 0) return;
    char* patch = smem + 2 * 65536 + w * RG_PATCH;
    char* pb = smem + 2 * 65536 + 4 * RG_PATCH + w * 4096;
    auto tile_origin = [&](int i, int& m0, int& n0) {        // i-th tile of this workgroup (past the last one: the last one again)
        const int t = min(xcd_remap((int)blockIdx.x + min(i, my_tiles - 1) * (int)gridDim.x, ntiles), ntiles - 1);
        m0 = (t / ntn) * 256; n0 = (t % ntn) * 256;
    };
    auto a_base = [&](int m0) { return (uint64_t)(uintptr_t)(A + (size_t)m0 * R); };
    auto b_base = [&](int n0) { return (uint64_t)(uintptr_t)(BTRANS ? B + n0 : B + (size_t)n0 * R); };
    // ---- the lane's table (layout: gen_gemm_asm.py, ring_tile): DMA source offsets of the wave's pieces 0 and 1, fragment addresses ----
    {
        uint32_t* lt = (uint32_t*)(pb + 1024) + lane;
        // image R (128-byte rows): piece p = rows 8 p .. 8 p + 7; the wave fetches pieces 8 w + j; physical chunk lane & 7 of row
        // r holds logical chunk (lane & 7) ^ ((r >> 1) & 7)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 64 * w + 8 * j + (lane >> 3);
            const uint32_t offR = (uint32_t)(((size_t)row * R + ((lane & 7) ^ ((row >> 1) & 7)) * 8) * 2);
            lt[64 * j] = offR;
            if (!BTRANS) lt[64 * (2 + j)] = offR;
        }
        if (BTRANS) {                                        // image T: piece 0 of the wave = rows 0 .. 7 of 64-column sub-tile w
            const int r = lane >> 3;
            const int chunk = (lane & 7) ^ (((r >> 1) & 1) << 2);
            lt[64 * 2] = (uint32_t)(((size_t)r * NO + 64 * w + chunk * 8) * 2);
            lt[64 * 3] = 0u;
        }
        lt[64 * 4] = lds_addr_of(smem) + imgR_off(128 * wm + l31, hh);
        if (!BTRANS) { lt[64 * 5] = lds_addr_of(smem) + 32768 + imgR_off(128 * wn + l31, hh); lt[64 * 6] = 0u; }
        else {
            const int i15 = lane & 15, gq = lane >> 4, rq = i15 >> 2;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int chunk = 4 * ct + 2 * (gq & 1) + ((i15 & 3) >> 1);
                lt[64 * (5 + ct)] = lds_addr_of(smem) + 32768 + 2 * wn * TILE_BYTES + (8 * hh + rq) * 128 +
                                    ((chunk ^ (((rq >> 1) & 1) << 2)) << 4) + 8 * (i15 & 1);
            }
        }
    }
    if (!BTRANS) {                                           // no bias: the epilogue adds these zeros (the statement's bias DMA fetches nothing)
        float* bl = (float*)(pb + 512);
        bl[lane] = 0.f;
        bl[64 + lane] = 0.f;
    }
    const uint32_t pba = __builtin_amdgcn_readfirstlane(lds_addr_of(pb));
    for (int ti = 0; ti < my_tiles; ++ti) {
        int m0, n0, m1, n1;
        tile_origin(ti, m0, n0);
        tile_origin(ti + 1, m1, n1);
        if (lane == 0) {
            uint64_t* p64 = (uint64_t*)pb;
            uint32_t* p32 = (uint32_t*)pb;
            if (ti == 0) {
                p64[0] = a_base(m0); p64[1] = b_base(n0);
                p32[11] = p32[19] = (uint32_t)nd;            // requests of A / of B left in the tile that operand's pointer stands in
            }
            p64[2] = a_base(m1); p64[3] = b_base(n1);
            p32[8] = 128u;                                   // bytes per stage: 64 columns of A
            p32[9] = BTRANS ? (uint32_t)(64 * NO * 2) : 128u;
            p32[10] = (uint32_t)nd;
            p32[12] = lds_addr_of(smem);
            p32[13] = (uint32_t)w;
            p32[14] = ti == 0 ? 1u : 0u;
            p32[15] = (uint32_t)(8 * R * 2);                 // 8 rows of A
            p64[8] = (uint64_t)(uintptr_t)(bias ? bias + n0 + 128 * wn : nullptr);      // this tile's bias (0: none -- the zeros below stay)
            p32[18] = BTRANS ? (uint32_t)(8 * NO * 2) : (uint32_t)(8 * R * 2);      // 8 rows of B
        }
        f32x16 acc[4][4];
#ifdef MGX_DW4_TIMES
        const unsigned long long tq0 = __builtin_amdgcn_s_memtime();
#endif
#define MGX_RING4_OPERANDS                                                                                                                  \
    : "=a"(acc[0][0]), "=a"(acc[0][1]), "=a"(acc[0][2]), "=a"(acc[0][3]), "=a"(acc[1][0]), "=a"(acc[1][1]), "=a"(acc[1][2]),                \
      "=a"(acc[1][3]), "=a"(acc[2][0]), "=a"(acc[2][1]), "=a"(acc[2][2]), "=a"(acc[2][3]), "=a"(acc[3][0]), "=a"(acc[3][1]),                \
      "=a"(acc[3][2]), "=a"(acc[3][3])                                                                                                      \
    : "s"(pba)                                                                                                                              \
    : MGX_RING4_CLOBBERS
        if constexpr (BTRANS) asm volatile(MGX_RING4_NN_ASM MGX_RING4_OPERANDS);
        else asm volatile(MGX_RING4_NT_ASM MGX_RING4_OPERANDS);
#undef MGX_RING4_OPERANDS
#ifdef MGX_DW4_TIMES
        const unsigned long long tq1 = __builtin_amdgcn_s_memtime();
#endif
        store_wave_block4<!BTRANS, PRE>(C, relu_y, addend, acc, pb + 512, act, m0 + 128 * wm, n0 + 128 * wn, NO, lane, patch);
#ifdef MGX_DW4_TIMES
        if (tid == 0) {          // per workgroup: [0] statement cycles, [1] epilogue cycles (summed over its tiles), [2] tiles, [3] real time
            const unsigned long long tq2 = __builtin_amdgcn_s_memtime();
            unsigned long long* rec = mgx_dw4_times_buf + 8 * blockIdx.x;
            if (ti == 0) { rec[0] = rec[1] = rec[2] = 0; rec[3] = __builtin_amdgcn_s_memrealtime(); }
            rec[0] += tq1 - tq0; rec[1] += tq2 - tq1; rec[2] += 1;
            rec[4] = __builtin_amdgcn_s_memrealtime() - rec[3];
        }
#endif
    }
    wait_vmcnt<0>();                                         // the requests past the last tile land before the workgroup's LDS is released
}

// =================================================================================================
// dX = dY W   (NN; optional epilogue: dX *= (relu_y > 0), the backward of a fused ReLU; then dX += addend)
//   tile: 128 rows m x 128 cols k', reduction over n in steps of 64
//   LDS:  dY tile [128 m][64 n] image R;  W tile [64 n][128 k'] as 4 sub-tiles (2 n-blocks x 2 col halves)
//         of [32][64] image T
// =================================================================================================
template <bool DBUF, bool EXACT>
__global__ __launch_bounds__(256, DBUF ? 2 : 3) void linear_dx_kernel(const uint16_t* __restrict__ dY,
                                                           const uint16_t* __restrict__ W,
                                                           const uint16_t* __restrict__ relu_y,
                                                           const uint16_t* __restrict__ addend,
                                                           uint16_t* __restrict__ dX, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ntk = (K + BN - 1) / BN, ntm = (M + BM - 1) / BM;
    const int bid = xcd_remap(blockIdx.x, ntm * ntk);
    const int tk = bid % ntk, tm = bid / ntk;
    const int m0 = tm * BM, k0 = tk * BN;

    // staging: dY tile as in the forward (row = srow + 32 i, 16-byte chunk sch of 64 n);
    //          W tile: 64 rows n x 256 B; thread -> (n rows (tid >> 4) + 16 i, chunk tid & 15 of 16)
    const int srow = tid >> 3, sch = tid & 7;
    const int wrow = tid >> 4, wch = tid & 15;           // W rows wrow + 16 i, 16 lanes per 256-byte row (see dW)
    u32x4 areg[4], wreg[4];
    // EXACT (N % 64 == 0, the case of every model shape): every load is unconditional -- rows beyond M and columns beyond
    // K are clamped into range (they only feed accumulator entries that are never stored), and no reduction tile is
    // partial.  Otherwise the reduction tail must be zero-filled: guarded loads.
    const uint16_t* ap[4];
    const uint16_t* wp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ap[i] = dY + (size_t)min(m0 + srow + 32 * i, M - 1) * N + sch * 8;
        wp[i] = W + (size_t)(wrow + 16 * i) * K + min(k0 + wch * 8, K - 8);
    }
    auto load_tiles = [&](int n0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (EXACT) {
                areg[i] = *(const u32x4*)(ap[i] + n0);
                wreg[i] = *(const u32x4*)(wp[i] + (size_t)n0 * K);
            } else {
                const int gm = m0 + srow + 32 * i;
                const int gn = n0 + sch * 8;                                  // N % 8 == 0 (host-checked)
                areg[i] = (gm < M && gn < N) ? *(const u32x4*)(dY + (size_t)gm * N + gn) : u32x4{0, 0, 0, 0};
                const int wn_ = n0 + wrow + 16 * i, wk = k0 + wch * 8;
                wreg[i] = (wn_ < N && wk < K) ? *(const u32x4*)(W + (size_t)wn_ * K + wk) : u32x4{0, 0, 0, 0};
            }
        }
    };
    auto store_tiles = [&](int buf) {
        char* at = smem + buf * 2 * IMG;
        char* wt = at + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(u32x4*)(at + imgR_off(srow + 32 * i, sch)) = areg[i];
            const int row = wrow + 16 * i;
            const int sub = (row >> 5) * 2 + (wch >> 3);                  // (n block, column half)
            *(u32x4*)(wt + sub * TILE_BYTES + imgT_off(row & 31, wch & 7)) = wreg[i];
        }
    };
    f32x16 acc[2][2];
    zero_acc(acc);
    auto multiply = [&](int cur) {
        const char* at = smem + cur * 2 * IMG;
        const char* wt = at + IMG;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a0 = frag_R(at, 64 * wm + l31, hh, ks);
            const bf16x8 a1 = frag_R(at, 64 * wm + 32 + l31, hh, ks);
            const char* sub = wt + ((ks >> 1) * 2 + wn) * TILE_BYTES;
            const bf16x8 b0 = fragTn(sub, lane, ks & 1, 0);
            const bf16x8 b1 = fragTn(sub, lane, ks & 1, 1);
            acc[0][0] = mfma(b0, a0, acc[0][0]);      // C^T tiles (see the forward kernel)
            acc[0][1] = mfma(b1, a0, acc[0][1]);
            acc[1][0] = mfma(b0, a1, acc[1][0]);
            acc[1][1] = mfma(b1, a1, acc[1][1]);
        }
    };
    const int nn = (N + BK - 1) / BK;
    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int nt = 0; nt + 1 < nn; ++nt) {
        const int cur = DBUF ? (nt & 1) : 0;
        load_tiles((nt + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);          // keep the prefetch AHEAD of the MFMAs (the scheduler sinks it to the barrier)
        multiply(cur);
        if (!DBUF) __syncthreads();                 // single buffer: everyone has read the tile before it is replaced
        store_tiles(DBUF ? (cur ^ 1) : 0);
        __syncthreads();
    }
    multiply(DBUF ? ((nn - 1) & 1) : 0);
    __syncthreads();
    store_tile_lds(dX, relu_y, addend, acc, nullptr, 0, m0 + 64 * wm, k0 + 64 * wn, M, K, lane, smem + w * EPI_PATCH);   // K % 8 == 0
}

// =================================================================================================
// gW[N,K] += dY^T X   (TN), M split over gridDim.y workgroups; partial tiles added with fp32 atomics
//   tile: 128 rows n x 128 cols k, reduction over m in steps of 64
//   LDS:  dY tile [64 m][128 n] and X tile [64 m][128 k], each as 4 sub-tiles [32][64] image T
// =================================================================================================
// EXACT: the row range [mbeg, mend) is a whole number of 64-row reduction tiles (every model shape): all loads are
// unconditional, columns beyond N / K clamped into range (they only feed accumulator entries that are never added).
template <bool EXACT>
MGX_DEV void dw_tile(const uint16_t* __restrict__ dY, const uint16_t* __restrict__ X, float* __restrict__ gW,
                     float* __restrict__ gb, int M, int N, int K, int tile, int mbeg, int mend, char* smem,
                     long long* __restrict__ detW = nullptr, long long* __restrict__ detb = nullptr) {
    // detW / detb (deterministic mode): fixed-point images of this launch's updates of gW / gb; the M-splits add integers
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    const int ntk = (K + BN - 1) / BN;
    const int tk = tile % ntk, tn = tile / ntk;
    const int n0 = tn * BM, k0 = tk * BN;

    // staging: 16 consecutive lanes cover one 256-byte row (16 chunks) -> the 8 lanes of a ds_write_b128 group hit
    // 8 distinct 16-byte slots of one sub-tile row (conflict-free), and global reads are 256-byte segments
    const int wrow = tid >> 4, ch = tid & 15;            // rows wrow + 16 i
    const int sub_c = ch >> 3, slot = ch & 7;
    // Register staging, DEPTH tiles deep (EXACT path): a reduction step is 32 KB of operands for 512 MFMA cycles, and a load
    // takes ~3,500 cycles to come back from beyond L2 with the chip streaming; with one tile in flight per workgroup and two
    // workgroups per CU the step time WAS the load latency (3,650 cycles per step measured = 14 % MFMA-busy per workgroup).
    // Three register sets keep three tiles in flight per workgroup while a fourth is multiplied out of LDS.
    constexpr int DEPTH = EXACT ? 3 : 1;
    u32x4 areg[DEPTH][4], breg[DEPTH][4];
    const uint16_t* ap = dY + (size_t)wrow * N + min(n0 + ch * 8, N - 8);
    const uint16_t* bp = X + (size_t)wrow * K + min(k0 + ch * 8, K - 8);
    auto load_tiles = [&](int mm, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        const int gn = n0 + ch * 8, gk = k0 + ch * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (EXACT) {
                areg[S][i] = *(const u32x4*)(ap + (size_t)(mm + 16 * i) * N);
                breg[S][i] = *(const u32x4*)(bp + (size_t)(mm + 16 * i) * K);
            } else {
                const int gm = mm + wrow + 16 * i;
                areg[S][i] = (gm < mend && gn < N) ? *(const u32x4*)(dY + (size_t)gm * N + gn) : u32x4{0, 0, 0, 0};
                breg[S][i] = (gm < mend && gk < K) ? *(const u32x4*)(X + (size_t)gm * K + gk) : u32x4{0, 0, 0, 0};
            }
        }
    };
    // bias gradient gb[n] += sum_m dY[m][n]: the workgroups of the first k-tile column add up the dY rows they stage anyway
    const bool do_bias = (gb != nullptr) && (tk == 0) && (!EXACT || n0 + ch * 8 < N);
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto store_tiles = [&](int buf, auto set_tag) {
        constexpr int S = decltype(set_tag)::value;
        char* at = smem + buf * 2 * IMG;
        char* bt = at + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wrow + 16 * i;
            const int off = ((row >> 5) * 2 + sub_c) * TILE_BYTES + imgT_off(row & 31, slot);
            *(u32x4*)(at + off) = areg[S][i];
            *(u32x4*)(bt + off) = breg[S][i];
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float f[8];
                unpack8(areg[S][i], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum[k] += f[k];
            }
        }
    };
    f32x16 acc[2][2];
    zero_acc(acc);
    const int nm = (mend - mbeg + BK - 1) / BK;
    auto multiply = [&](int cur) {
        const char* at = smem + cur * 2 * IMG;
        const char* bt = at + IMG;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const char* asub = at + ((ks >> 1) * 2 + wm) * TILE_BYTES;
            const char* bsub = bt + ((ks >> 1) * 2 + wn) * TILE_BYTES;
            const bf16x8 a0 = fragTn(asub, lane, ks & 1, 0);
            const bf16x8 a1 = fragTn(asub, lane, ks & 1, 1);
            const bf16x8 b0 = fragTn(bsub, lane, ks & 1, 0);
            const bf16x8 b1 = fragTn(bsub, lane, ks & 1, 1);
            acc[0][0] = mfma(a0, b0, acc[0][0]);
            acc[0][1] = mfma(a0, b1, acc[0][1]);
            acc[1][0] = mfma(a1, b0, acc[1][0]);
            acc[1][1] = mfma(a1, b1, acc[1][1]);
        }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, (DEPTH > 1 ? 1 : 0)>;
    using S2 = std::integral_constant<int, (DEPTH > 2 ? 2 : 0)>;
    if (EXACT) {
        // tile t lives in register set t % 3 until it is written to LDS buffer t & 1.  Loads are clamped to the last tile
        // (the surplus ones are never stored).
        auto tile_row = [&](int t) { return mbeg + min(t, nm - 1) * BK; };
        if (nm > 0) {
            load_tiles(tile_row(0), S0{});
            load_tiles(tile_row(1), S1{});
            load_tiles(tile_row(2), S2{});
            store_tiles(0, S0{});
            load_tiles(tile_row(3), S0{});
        }
        __syncthreads();
        // one reduction step: multiply tile t, publish tile t+1 (its set has arrived: two younger tiles stay in flight),
        // refill that set with tile t+4
        auto step = [&](int t, auto set_tag) {
            multiply(t & 1);
            store_tiles((t + 1) & 1, set_tag);
            __syncthreads();
            load_tiles(tile_row(t + 4), set_tag);
        };
        int t = 0;
        for (; t + 3 <= nm - 1; t += 3) {                // branch-free: the set of tile t+1 is (t+1) % 3 = 1, 2, 0
            step(t, S1{});
            step(t + 1, S2{});
            step(t + 2, S0{});
        }
        if (t < nm - 1) {
            step(t, S1{});
            ++t;
            if (t < nm - 1) { step(t, S2{}); ++t; }
        }
        if (nm > 0) multiply((nm - 1) & 1);
    } else {
        if (nm > 0) {
            load_tiles(mbeg, S0{});
            store_tiles(0, S0{});
        }
        __syncthreads();
        for (int mt = 0; mt + 1 < nm; ++mt) {
            const int cur = mt & 1;
            load_tiles(mbeg + (mt + 1) * BK, S0{});
            __builtin_amdgcn_sched_barrier(0);               // keep the prefetch AHEAD of the MFMAs
            multiply(cur);
            store_tiles(cur ^ 1, S0{});
            __syncthreads();
        }
        if (nm > 0) multiply((nm - 1) & 1);
    }
    if (do_bias) {      // lanes with equal (tid & 15) hold the same 8 columns: fold lane bits 4,5, then one atomic per wave
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bsum[k] += __shfl_xor(bsum[k], 16, 64);
            bsum[k] += __shfl_xor(bsum[k], 32, 64);
        }
        if (lane < 16 && n0 + ch * 8 < N) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (detb) det_add(detb + n0 + ch * 8 + k, bsum[k]);
                else atomicAdd(gb + n0 + ch * 8 + k, bsum[k]);
            }
        }
    }
    // D[n][k]: k on the lane -> one register = two 128-byte row segments per wave-instruction
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int k = k0 + 64 * wn + 32 * ct + l31;
        if (k >= K) continue;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + 64 * wm + 32 * rt + crow(r, hh);
                if (n < N) {
                    if (detW) det_add(detW + (size_t)n * K + k, acc[rt][ct][r]);
                    else atomicAdd(gW + (size_t)n * K + k, acc[rt][ct][r]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void linear_dw_kernel(const uint16_t* __restrict__ dY,
                                                           const uint16_t* __restrict__ X,
                                                           float* __restrict__ gW, float* __restrict__ gb, int M, int N,
                                                           int K, int mchunk, int tiles, long long* __restrict__ detW,
                                                           long long* __restrict__ detb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 1-D grid of tiles x splits units, unit = split * tiles + tile; each XCD walks a contiguous run of units, i.e.
    // (mostly) the tiles of ONE row chunk: the dY / X blocks those tiles share are fetched into that XCD's L2 once
    const int u = xcd_remap(blockIdx.x, gridDim.x);
    const int mbeg = (u / tiles) * mchunk;
    if (M % mchunk == 0 && mchunk % BK == 0) dw_tile<true>(dY, X, gW, gb, M, N, K, u % tiles, mbeg, mbeg + mchunk, smem, detW, detb);
    else dw_tile<false>(dY, X, gW, gb, M, N, K, u % tiles, mbeg, min(M, mbeg + mchunk), smem, detW, detb);
}

// Several weight gradients that share the row count M (one encoder block's QKV / fc / FFN projections) in ONE launch:
// with all their 128 x 128 tiles in the grid, far fewer M-splits fill the chip, and the fp32 atomic traffic -- one
// partial tile per split, ~1.3 TB/s chip-wide -- drops from 75 MB to ~30 MB per block at cfg2.
struct DwGroup {
    const uint16_t* dY[MGX_DW_MAX_GROUP];
    const uint16_t* X[MGX_DW_MAX_GROUP];
    float* gW[MGX_DW_MAX_GROUP];
    float* gb[MGX_DW_MAX_GROUP];
    int N[MGX_DW_MAX_GROUP], K[MGX_DW_MAX_GROUP];
    int first_tile[MGX_DW_MAX_GROUP + 1];                  // prefix sums of the tile counts
    int n;
};

__global__ __launch_bounds__(256, 2) void linear_dw_grouped_kernel(const DwGroup g, int M, int mchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles = g.first_tile[g.n];
    const int u = xcd_remap(blockIdx.x, gridDim.x);        // see linear_dw_kernel
    const int t = u % tiles;
    int p = 0;
    while (p + 1 < g.n && t >= g.first_tile[p + 1]) ++p;
    const int mbeg = (u / tiles) * mchunk;
    if (M % mchunk == 0 && mchunk % BK == 0)
        dw_tile<true>(g.dY[p], g.X[p], g.gW[p], g.gb[p], M, g.N[p], g.K[p], t - g.first_tile[p], mbeg, mbeg + mchunk, smem);
    else
        dw_tile<false>(g.dY[p], g.X[p], g.gW[p], g.gb[p], M, g.N[p], g.K[p], t - g.first_tile[p], mbeg, min(M, mbeg + mchunk), smem);
}

// =================================================================================================
// Ring kernel for the weight gradients of one encoder block:  gW[N,K] += dY^T X  (TN), same structure as
// linear_ring_kernel (256 x 256 tile, 8 waves, 4-stage DMA ring, reduction steps of 32 rows m), one (tile, M-split) unit per
// workgroup.  Both operand tiles of a step are [32 m][256 cols] = 4 sub-tiles [32][64] (image T), and every fragment
// is a transposed read (ds_read_b64_tr_b16 x 2, issued from inline asm: see linear_ring_kernel).
// A unit leaves its 256 x 256 fp32 partial in the workspace with plain stores (row-major, through the wave's LDS patch);
// dw_fixup_kernel then adds the splits of a tile into gW.  (fp32 atomics run at ~1.3 TB/s chip-wide and stall the issuing
// waves: 63 MB of partials per block would cost ~48 us of every CU's time, plain stores + the fix-up pass ~20.)
// Bias gradient gb[n] += sum_m dY[m][n]: the waves of the first k-tile column with wn == 0 add up the dY fragments they
// hold anyway (v_dot2c_f32_bf16 against (1, 1)).
// =================================================================================================
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
struct DwRing {
    const uint16_t* dY[MGX_DW_MAX_GROUP];
    const uint16_t* X[MGX_DW_MAX_GROUP];
    float* gW[MGX_DW_MAX_GROUP];
    float* gb[MGX_DW_MAX_GROUP];
    long long* detb[MGX_DW_MAX_GROUP];                     // deterministic mode: fixed-point images of the bias-gradient updates (else NULL)
    int N[MGX_DW_MAX_GROUP], K[MGX_DW_MAX_GROUP];
    int first_tile[MGX_DW_MAX_GROUP + 1];                  // prefix sums of the 256 x 256 tile counts
    int n, splits, steps_per_split;
    int ragged;                                            // some weight does not tile into whole 256 x 256 tiles (four-wave kernel only)
};

__global__ __launch_bounds__(512, 1) void linear_dw_ring_kernel(const DwRing g, int M, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 2, wn = w & 3;
    const int l31 = lane & 31, hh = lane >> 5;
    // unit order: one XCD runs a contiguous range of (split, tile) pairs, split-major -- the tiles of one M-split read the
    // same rows of dY / X (tiles of one row / column of a weight share an operand tile), so they meet in that XCD's L2
    const int tiles_all = g.first_tile[g.n];
    const int u = xcd_remap(blockIdx.x, gridDim.x);
    const int sp = u / tiles_all, t = u - sp * tiles_all;
    const int unit = t * g.splits + sp;                      // position of the partial tile in the workspace
    int p = 0;
    while (p + 1 < g.n && t >= g.first_tile[p + 1]) ++p;
    const int N = g.N[p], K = g.K[p];
    const int ntk = K >> 8, tl = t - g.first_tile[p];
    const int n0 = (tl / ntk) << 8, k0 = (tl % ntk) << 8;
    const int total = M >> 5;
    const int s0 = sp * g.steps_per_split;
    const int G = min(total, s0 + g.steps_per_split) - s0;   // >= 1 (host)
    char* patch = smem + RG_NST * RG_STAGE + w * RG_PATCH;

    // ---- DMA stream: piece q = 2w + j of an operand image = rows 8 (q & 3) .. +7 of sub-tile q >> 2 ----
    const uint16_t* ap[2];
    const uint16_t* bp[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = 2 * w + j;
        const int r = 8 * (q & 3) + (lane >> 3);
        const int chunk = (lane & 7) ^ (((r >> 1) & 1) << 2);
        ap[j] = g.dY[p] + (size_t)(s0 * 32 + r) * N + n0 + 64 * (q >> 2) + chunk * 8;
        bp[j] = g.X[p] + (size_t)(s0 * 32 + r) * K + k0 + 64 * (q >> 2) + chunk * 8;
    }
    int d_st = 0;
    char* d_at = nullptr;
    const uint16_t* da[2] = {nullptr, nullptr};
    const uint16_t* db[2] = {nullptr, nullptr};
    auto dma_begin = [&]() {                                 // addresses of the next request, pointers move one step on
        d_at = smem + d_st * RG_STAGE + (2 * w) * 1024;
        d_st = (d_st + 1) & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            da[j] = ap[j]; db[j] = bp[j];
            ap[j] += (size_t)32 * N; bp[j] += (size_t)32 * K;
        }
    };
    auto dma_all = [&]() {
        dma_begin();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            glds16(da[j], d_at + j * 1024);
            glds16(db[j], d_at + 16384 + j * 1024);
        }
    };
    // ---- fragments: transposed reads; ta[ct] / tbb[ct] = the lane's address of fragTn(first sub-tile of the wave, ks = 0,
    //      column half ct) in stage 0; the immediate offset adds ks, jq and (for the A operand) the second sub-tile ----
    uint32_t ta[2], tbb[2];
    {
        const int i15 = lane & 15, gq = lane >> 4, rq = i15 >> 2;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int chunk = 4 * ct + 2 * (gq & 1) + ((i15 & 3) >> 1);
            const uint32_t in_tile = (8 * hh + rq) * 128 + ((chunk ^ (((rq >> 1) & 1) << 2)) << 4) + 8 * (i15 & 1);
            ta[ct] = lds_addr_of(smem) + 2 * wm * TILE_BYTES + in_tile;
            tbb[ct] = lds_addr_of(smem) + 16384 + wn * TILE_BYTES + in_tile;
        }
    }
    u32x2 ha[2][4][2], hb[2][2][2];                           // [set][fragment][jq]
    auto rd_at = [&](int stg, auto ks_tag, auto i_tag, u32x2 (&h)[2]) {
        constexpr int KS = decltype(ks_tag)::value, I = decltype(i_tag)::value;
        const uint32_t addr = ta[I & 1] + stg * RG_STAGE;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h[0]) : "v"(addr), "n"(2048 * KS + 4096 * (I >> 1)));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h[1]) : "v"(addr), "n"(2048 * KS + 4096 * (I >> 1) + 512));
    };
    auto rd_bt = [&](int stg, auto ks_tag, int ct, u32x2 (&h)[2]) {
        constexpr int KS = decltype(ks_tag)::value;
        const uint32_t addr = tbb[ct] + stg * RG_STAGE;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h[0]) : "v"(addr), "n"(2048 * KS));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h[1]) : "v"(addr), "n"(2048 * KS + 512));
    };
    auto join = [&](const u32x2 (&h)[2]) { return __builtin_bit_cast(bf16x8, u32x4{h[0].x, h[0].y, h[1].x, h[1].y}); };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i][0] = zero16(); acc[i][1] = zero16(); }
    const bool do_bias = __builtin_amdgcn_readfirstlane((g.gb[p] != nullptr) && k0 == 0 && wn == 0);
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    auto block = [&](auto cur_tag, int nstg, auto nks_tag, const uint16_t* g0, char* l0, const uint16_t* g1, char* l1, bool on) {
        constexpr int CUR = decltype(cur_tag)::value, NXT = CUR ^ 1;
        bf16x8 a[4], b[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = join(ha[CUR][i]);
        b[0] = join(hb[CUR][0]); b[1] = join(hb[CUR][1]);
        MGX_SB();
        acc[0][0] = mfma(b[0], a[0], acc[0][0]); MGX_SB();
        rd_at(nstg, nks_tag, T0{}, ha[NXT][0]);
        rd_bt(nstg, nks_tag, 0, hb[NXT][0]);
        MGX_SB();
        acc[0][1] = mfma(b[1], a[0], acc[0][1]); MGX_SB();
        rd_at(nstg, nks_tag, T1{}, ha[NXT][1]);
        rd_bt(nstg, nks_tag, 1, hb[NXT][1]);
        MGX_SB();
        acc[1][0] = mfma(b[0], a[1], acc[1][0]); MGX_SB();
        rd_at(nstg, nks_tag, T2{}, ha[NXT][2]);
        rd_at(nstg, nks_tag, T3{}, ha[NXT][3]);
        MGX_SB();
        acc[1][1] = mfma(b[1], a[1], acc[1][1]); MGX_SB();
        if (on) glds16(g0, l0);
        MGX_SB();
        acc[2][0] = mfma(b[0], a[2], acc[2][0]); MGX_SB();
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                // (element pairs by shufflevector: indexing a u32x4 view of the fragment inside an unrolled loop made
                //  hipcc 7.2 feed the FIRST dword to all four dot products)
                const bf16x2_t one = {(__bf16)1.0f, (__bf16)1.0f};
                gsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 0, 1), one, gsum[i], false);
                gsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 2, 3), one, gsum[i], false);
                gsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 4, 5), one, gsum[i], false);
                gsum[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(a[i], a[i], 6, 7), one, gsum[i], false);
            }
        }
        MGX_SB();
        acc[2][1] = mfma(b[1], a[2], acc[2][1]); MGX_SB();
        if (on) glds16(g1, l1);
        MGX_SB();
        acc[3][0] = mfma(b[0], a[3], acc[3][0]); MGX_SB();
        acc[3][1] = mfma(b[1], a[3], acc[3][1]); MGX_SB();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every transposed read of the block has returned
        MGX_SB();
    };

    dma_all();
    if (G > 1) dma_all();
    if (G > 2) dma_all();
    if (G > 3) dma_all();
    if (G > 3) wait_vmcnt<12>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    rd_at(0, T0{}, T0{}, ha[0][0]); rd_at(0, T0{}, T1{}, ha[0][1]); rd_at(0, T0{}, T2{}, ha[0][2]); rd_at(0, T0{}, T3{}, ha[0][3]);
    rd_bt(0, T0{}, 0, hb[0][0]); rd_bt(0, T0{}, 1, hb[0][1]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    MGX_SB();
    int cs = 0;
    bool pend = false;
    for (int s = 0; s < G; ++s) {
        const int ns = (cs + 1) & 3;
        block(T0{}, cs, T1{}, db[0], d_at + 16384, db[1], d_at + 16384 + 1024, pend);
        if (s + 3 < G) wait_vmcnt<8>(); else wait_vmcnt<0>();   // step s+1 has landed (requests s+2, s+3 may be outstanding)
        __builtin_amdgcn_s_barrier();
        pend = (s + 4 < G);
        if (pend) dma_begin();
        block(T1{}, ns, T0{}, da[0], d_at, da[1], d_at + 1024, pend);
        cs = ns;
    }

    // ---- epilogue: fp32 partial tile -> workspace, row-major [n][k], 128-byte row segments per 8 lanes ----
    float* wsu = ws + (size_t)unit * 65536;
    const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *(f32x4*)(patch + l31 * 128 + (((2 * g4 + hh) ^ (l31 & 7)) << 4)) =
                    f32x4{acc[rt][ct][4 * g4], acc[rt][ct][4 * g4 + 1], acc[rt][ct][4 * g4 + 2], acc[rt][ct][4 * g4 + 3]};
            wave_lds_fence();
            f32x4 o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + 8 * i;
                o[i] = *(const f32x4*)(patch + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(f32x4*)(wsu + (size_t)(128 * wm + 32 * rt + rr + 8 * i) * 256 + 64 * wn + 32 * ct + 4 * ch) = o[i];
            wave_lds_fence();
        }
    if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = gsum[i] + __shfl_xor(gsum[i], 32, 64);
            if (hh == 0) {
                if (g.detb[p]) det_add(g.detb[p] + n0 + 128 * wm + 32 * i + l31, v);
                else atomicAdd(g.gb[p] + n0 + 128 * wm + 32 * i + l31, v);
            }
        }
    }
}

// ---- the same unit with FOUR waves, one per SIMD, each holding a 128 x 128 output tile = 16 accumulator tiles = all 256 AGPRs ----
// The eight-wave kernel above reads 6 operand fragments from LDS for 8 MFMAs per k-step and wave: 96 KB of transposing reads + 32 KB
// of DMA writes per stage and workgroup against 1024 MFMA cycles per SIMD -- the LDS (128 B/clk) is as busy as the MFMA pipe, and the
// kernel sat at ~49 % MFMA-busy.  128 x 128 wave tiles read 8 fragments for 16 MFMAs: 64 + 32 KB per stage.  hipcc cannot keep 256
// accumulators in place for one wave (rounds 3-4: it shuffles them between the register files), so the main loop is one generated asm
// statement that owns them (gen_gemm_asm.py -> linear_dw_ring4_loop.inc; parameters through an LDS block as in rel_attn_dkv64.hip);
// unit decoding, the parameter block and the epilogue stay HIP.  Same images, same stage order, same MFMA operand order as the
// eight-wave kernel: the partial tiles are bit-identical to its.
#if defined(MGX_GEMM_DIAG) && MGX_GEMM_DIAG
#include "linear_dw_ring4_loop_diag.inc"
#else
#include "linear_dw_ring4_loop.inc"
#endif
#ifdef MGX_DW4_TIMES
// diagnostic builds: s_memrealtime (100 MHz) at a unit's start, loop start, loop end and end, per workgroup (tools/dw4_times.py)
__device__ unsigned long long mgx_dw4_times_buf[8 * 1024];     // [workgroup][4 real-time stamps, 4 shader-clock stamps]
extern "C" int mgx_debug_dw4_times(unsigned long long* out, int n) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mgx_dw4_times_buf), sizeof(unsigned long long) * n);
}
#define DW4_TIME(k) do { if (tid == 0) { mgx_dw4_times_buf[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); \
                                        mgx_dw4_times_buf[8 * blockIdx.x + 4 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define DW4_TIME(k) do { } while (0)
#endif
__global__ __launch_bounds__(256, 1) void linear_dw_ring4_kernel(const DwRing g, int M, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, hh = lane >> 5;
    DW4_TIME(0);
    const int tiles_all = g.first_tile[g.n];
    const int u = xcd_remap(blockIdx.x, gridDim.x);          // unit order: see linear_dw_ring_kernel
    const int sp = u / tiles_all, t = u - sp * tiles_all;
    const int unit = t * g.splits + sp;
    int p = 0;
    while (p + 1 < g.n && t >= g.first_tile[p + 1]) ++p;
    const int N = g.N[p], K = g.K[p];
    const int ntk = (K + 255) >> 8, tl = t - g.first_tile[p];
    const int n0 = (tl / ntk) << 8, k0 = (tl % ntk) << 8;     // a weight's last tile row / column may be ragged (N, K % 8 == 0)
    const int total = M >> 5;
    const int s0 = sp * g.steps_per_split;
    const int G = min(total, s0 + g.steps_per_split) - s0;   // >= 1 (host)
    char* patch = smem + RG_NST * RG_STAGE + w * RG_PATCH;
    // bias gradient gb[n] += sum_m dY[m][n]: the 2 ntk waves that hold the same dY fragments (wn = 0, 1 of every k-tile of the tile row)
    // share the sums -- one fragment each when there are four or more of them, two each when there are two (gen_gemm_asm.py:
    // BIAS_VARIANTS; 32 v_dot2c per stage in one wave of a workgroup slowed the whole workgroup by a fifth)
    int bias_mode = 0, bias_mask = 0;
    if (g.gb[p] != nullptr) {
        const int j = 2 * (k0 >> 8) + wn;
        if (ntk >= 2) { if (j < 4) { bias_mode = 4 + j; bias_mask = 1 << j; } }
        else { bias_mode = 2 + j; bias_mask = 3 << (2 * j); }
    }
    bias_mode = __builtin_amdgcn_readfirstlane(bias_mode);
    bias_mask = __builtin_amdgcn_readfirstlane(bias_mask);

    // ---- parameter block (layout: gen_gemm_asm.py, prologue), in the wave's epilogue patch ----
    {
        uint32_t* lt = (uint32_t*)(patch + 256) + lane;
        // DMA: this wave fetches pieces q = 4 w + j of both images = rows 8 j .. 8 j + 7 of 64-column sub-tile w
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * j + (lane >> 3);
            const int chunk = (lane & 7) ^ (((r >> 1) & 1) << 2);
            // columns past a ragged edge: fetch the row's last 8 columns instead (in bounds; output element (n, k) depends on dY
            // column n and X column k alone, and the fix-up pass never reads the rows / columns past the edge)
            const int cy = min(64 * w + chunk * 8, N - 8 - n0), cx = min(64 * w + chunk * 8, K - 8 - k0);
            lt[64 * j] = (uint32_t)(((size_t)r * N + cy) * 2);
            lt[64 * (4 + j)] = (uint32_t)(((size_t)r * K + cx) * 2);
        }
        const int i15 = lane & 15, gq = lane >> 4, rq = i15 >> 2;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int chunk = 4 * ct + 2 * (gq & 1) + ((i15 & 3) >> 1);
            const uint32_t in_tile = (8 * hh + rq) * 128 + ((chunk ^ (((rq >> 1) & 1) << 2)) << 4) + 8 * (i15 & 1);
            lt[64 * (8 + ct)] = lds_addr_of(smem) + 2 * wm * TILE_BYTES + in_tile;
            lt[64 * (10 + ct)] = lds_addr_of(smem) + 16384 + 2 * wn * TILE_BYTES + in_tile;
        }
        if (lane == 0) {
            uint64_t* p64 = (uint64_t*)patch;
            p64[0] = (uint64_t)(uintptr_t)(g.dY[p] + (size_t)(s0 * 32) * N + n0);
            p64[1] = (uint64_t)(uintptr_t)(g.X[p] + (size_t)(s0 * 32) * K + k0);
            uint32_t* p32 = (uint32_t*)patch;
            p32[4] = (uint32_t)(32 * N * 2);                 // bytes per stage
            p32[5] = (uint32_t)(32 * K * 2);
            p32[6] = (uint32_t)G;
            p32[7] = lds_addr_of(smem);
            p32[8] = (uint32_t)w;
            p32[9] = (uint32_t)bias_mode;
            p32[10] = p32[11] = 0u;
        }
    }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = zero16();
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};
    const uint32_t pba = __builtin_amdgcn_readfirstlane(lds_addr_of(patch));
    DW4_TIME(1);
    asm volatile(MGX_DW4_LOOP_ASM
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]),
                   "+a"(acc[1][3]), "+a"(acc[2][0]), "+a"(acc[2][1]), "+a"(acc[2][2]), "+a"(acc[2][3]), "+a"(acc[3][0]), "+a"(acc[3][1]),
                   "+a"(acc[3][2]), "+a"(acc[3][3]), "+v"(gsum[0]), "+v"(gsum[1]), "+v"(gsum[2]), "+v"(gsum[3])
                 : "s"(pba)
                 : MGX_DW4_LOOP_CLOBBERS);
    DW4_TIME(2);

    // ---- epilogue: fp32 partial tile -> workspace, row-major [n][k], 128-byte row segments per 8 lanes ----
    float* wsu = ws + (size_t)unit * 65536;
    const int rr = lane >> 3, ch = lane & 7;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *(f32x4*)(patch + l31 * 128 + (((2 * g4 + hh) ^ (l31 & 7)) << 4)) =
                    f32x4{acc[rt][ct][4 * g4], acc[rt][ct][4 * g4 + 1], acc[rt][ct][4 * g4 + 2], acc[rt][ct][4 * g4 + 3]};
            wave_lds_fence();
            f32x4 o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = rr + 8 * i;
                o[i] = *(const f32x4*)(patch + row * 128 + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *(f32x4*)(wsu + (size_t)(128 * wm + 32 * rt + rr + 8 * i) * 256 + 128 * wn + 32 * ct + 4 * ch) = o[i];
            wave_lds_fence();
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if ((bias_mask >> i) & 1) {
            const float v = gsum[i] + __shfl_xor(gsum[i], 32, 64);
            if (hh == 0 && n0 + 128 * wm + 32 * i + l31 < N) {
                if (g.detb[p]) det_add(g.detb[p] + n0 + 128 * wm + 32 * i + l31, v);
                else atomicAdd(g.gb[p] + n0 + 128 * wm + 32 * i + l31, v);
            }
        }
    }
    DW4_TIME(3);
}

// gW tile += sum over the M-splits of its partial tiles (fp32, 16 bytes per thread, fully coalesced)
__global__ __launch_bounds__(256) void dw_fixup_kernel(const DwRing g, const float* __restrict__ ws) {
    const int t = blockIdx.y;
    int p = 0;
    while (p + 1 < g.n && t >= g.first_tile[p + 1]) ++p;
    const int K = g.K[p], ntk = (K + 255) >> 8, tl = t - g.first_tile[p];
    const int n0 = (tl / ntk) << 8, k0 = (tl % ntk) << 8;
    const int e4 = blockIdx.x * 256 + threadIdx.x;           // float4 index inside the tile: 0 .. 16383
    if (n0 + (e4 >> 6) >= g.N[p] || k0 + 4 * (e4 & 63) >= K) return;      // past a ragged edge
    const float* src = ws + (size_t)t * g.splits * 65536 + (size_t)e4 * 4;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < g.splits; ++s) {
        const f32x4 v = *(const f32x4*)(src + (size_t)s * 65536);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
    }
    float* dst = g.gW[p] + (size_t)(n0 + (e4 >> 6)) * K + k0 + 4 * (e4 & 63);
    f32x4 o = *(f32x4*)dst;
    o.x += sum.x; o.y += sum.y; o.z += sum.z; o.w += sum.w;
    *(f32x4*)dst = o;
}

// =================================================================================================
// skinny forward (M <= 32: the decode path's projections).  The weights are streamed exactly once:
// workgroup = 32 output columns, its 4 waves split K; W rows and x rows go straight from global/L2 into
// MFMA fragments (no LDS staging, no barriers in the loop); the four partial tiles are combined in LDS.
//   D[n][m] = sum_k W[n][k] x[m][k]   (A = W rows, B = x^T)
// =================================================================================================
// FRAG: W is in MFMA fragment order (mgx.h: unit ((nt*K/16 + ks)*64 + lane) = W[32 nt + lane%32][16 ks + 8 (lane/32) ..+7], rows
// padded with zeros to a multiple of 32): a wave load is 1 KB contiguous instead of 32 B of 32 different rows.
template <bool FRAG>
__global__ __launch_bounds__(256) void linear_skinny_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                            const float* __restrict__ bias, uint16_t* __restrict__ C,
                                                            int M, int N, int K, int act) {
    __shared__ float part[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kq = K >> 2;                                   // K per wave (multiple of 16)
    const int nrow = n0 + l31, mrow = l31;
    const bool nv = FRAG || nrow < N, mv = mrow < M;
    const uint16_t* wp = FRAG ? W + (((size_t)blockIdx.x * (K >> 4) + (size_t)(w * kq >> 4)) * 64 + lane) * 8
                              : W + (size_t)(nv ? nrow : 0) * K + w * kq + hh * 8;
    const int wstep = FRAG ? 512 : 16;                       // elements between consecutive k-steps
    const uint16_t* xp = A + (size_t)(mv ? mrow : 0) * K + w * kq + hh * 8;
    f32x16 acc = zero16();
    for (int k0 = 0; k0 < kq; k0 += 128) {                   // 8 k-steps per trip: all 16 loads of a K <= 512 projection at once
        u32x4 wf[8], xf[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bool in = k0 + 16 * ks < kq;
            wf[ks] = (nv && in) ? *(const u32x4*)(wp + (size_t)((k0 >> 4) + ks) * wstep) : u32x4{0, 0, 0, 0};
            xf[ks] = (mv && in) ? *(const u32x4*)(xp + k0 + 16 * ks) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            acc = mfma(__builtin_bit_cast(bf16x8, wf[ks]), __builtin_bit_cast(bf16x8, xf[ks]), acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[w][crow(r, hh)][l31] = acc[r];      // [n][m]
    __syncthreads();
    // thread -> (n = tid >> 3, 4 consecutive m)
    const int n = tid >> 3, m4 = (tid & 7) * 4;
    if (n0 + n < N) {
        const float bv = bias ? bias[n0 + n] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m4 + k;
            if (m < M) {
                float v = part[0][n][m] + part[1][n][m] + part[2][n][m] + part[3][n][m] + bv;
                if (act == 1) v = fmaxf(v, 0.f);
                C[(size_t)m * N + n0 + n] = f32_to_bf16(v);
            }
        }
    }
}

// =================================================================================================
// =================================================================================================
// skinny forward with a LayerNorm prologue (decode path): Z = LN(X + RES) (layers.py:154-155,159-160, eps 1e-6, no
// dropout in eval) and C = act(Z W^T + b) in ONE launch.  Every workgroup owns 32 output columns and, like the kernel
// above, reads all M <= 32 rows of its operand anyway, so it normalises them itself (row statistics reduced across
// its 4 k-slices through LDS); workgroup 0 also writes Z, which the next LayerNorm needs as its residual.  Removes
// the 12 LayerNorm launches of a decode step (each ~4.6 us at the launch floor).  K <= 1024.
// =================================================================================================
constexpr int SKLN_MAXF = 16;                              // 16-column fragments per wave: K/4/16 <= 16
template <bool FRAG>
__global__ __launch_bounds__(256) void linear_skinny_ln_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ RES,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, const uint16_t* __restrict__ W,
                                                               const float* __restrict__ bias, uint16_t* __restrict__ C,
                                                               uint16_t* __restrict__ Z, int M, int N, int K, int act) {
    __shared__ float part[4][32][33];
    __shared__ float stat[2][4][32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kq = K >> 2, nf = kq >> 4;                     // fragments of 16 columns per wave
    const int nrow = n0 + l31, mrow = l31;
    const bool nv = FRAG || nrow < N, mv = mrow < M;
    const size_t xoff = (size_t)(mv ? mrow : 0) * K + w * kq + hh * 8;
    float z[SKLN_MAXF][8];
    float s1 = 0.f, s2 = 0.f;
    // the weight fragments are requested first: their latency hides under the statistics
    const uint16_t* wp = FRAG ? W + (((size_t)blockIdx.x * (K >> 4) + (size_t)(w * kq >> 4)) * 64 + lane) * 8
                              : W + (size_t)(nv ? nrow : 0) * K + w * kq + hh * 8;
    const int wstep = FRAG ? 512 : 16;
    u32x4 wf[SKLN_MAXF];
#pragma unroll
    for (int f = 0; f < SKLN_MAXF; ++f)
        if (f < nf) wf[f] = nv ? *(const u32x4*)(wp + (size_t)f * wstep) : u32x4{0, 0, 0, 0};
#pragma unroll
    for (int f = 0; f < SKLN_MAXF; ++f) {
        if (f < nf) {
            float a[8], r[8];
            unpack8(mv ? *(const u32x4*)(X + xoff + 16 * f) : u32x4{0, 0, 0, 0}, a);
            unpack8(mv ? *(const u32x4*)(RES + xoff + 16 * f) : u32x4{0, 0, 0, 0}, r);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                z[f][k] = a[k] + r[k];
                s1 += z[f][k];
                s2 += z[f][k] * z[f][k];
            }
        }
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (hh == 0) { stat[0][w][l31] = s1; stat[1][w][l31] = s2; }
    __syncthreads();
    const float t1 = stat[0][0][l31] + stat[0][1][l31] + stat[0][2][l31] + stat[0][3][l31];
    const float t2 = stat[1][0][l31] + stat[1][1][l31] + stat[1][2][l31] + stat[1][3][l31];
    const float mean = t1 / (float)K;
    const float rstd = rsqrtf(fmaxf(t2 / (float)K - mean * mean, 0.f) + eps);
    f32x16 acc = zero16();
#pragma unroll
    for (int f = 0; f < SKLN_MAXF; ++f) {
        if (f < nf) {
            const int kc = w * kq + hh * 8 + 16 * f;
            const f32x4 g0 = *(const f32x4*)(gamma + kc), g1 = *(const f32x4*)(gamma + kc + 4);
            const f32x4 b0 = *(const f32x4*)(beta + kc), b1 = *(const f32x4*)(beta + kc + 4);
            const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
            float y[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) y[k] = (z[f][k] - mean) * rstd * gg[k] + bb[k];
            const u32x4 yf = pack8(y);
            if (blockIdx.x == 0 && mv) *(u32x4*)(Z + xoff + 16 * f) = yf;
            acc = mfma(__builtin_bit_cast(bf16x8, wf[f]), __builtin_bit_cast(bf16x8, mv ? yf : u32x4{0, 0, 0, 0}), acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[w][crow(r, hh)][l31] = acc[r];      // [n][m]
    __syncthreads();
    const int n = tid >> 3, m4 = (tid & 7) * 4;
    if (n0 + n < N) {
        const float bv = bias ? bias[n0 + n] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m4 + k;
            if (m < M) {
                float v = part[0][n][m] + part[1][n][m] + part[2][n][m] + part[3][n][m] + bv;
                if (act == 1) v = fmaxf(v, 0.f);
                C[(size_t)m * N + n0 + n] = f32_to_bf16(v);
            }
        }
    }
}


// =================================================================================================
// Decode-step fusion (M <= 32 rows = the decode batch): linear_skinny_embed_kernel computes H = emb[tok] sqrt(d) + PE[t]
// (layers.py:226-229) inside the first QKV projection (8.4 us against 4.7 + 5.2 us for the two launches).
// Two further fusions were built and measured in round 3 and are NOT kept (profiles/README.md): the split-K attention merge
// inside the output projection (16.7 us against 4.8 + 5.2: the fp32 partials are 8x the bytes of the bf16 context row and
// every workgroup re-merges them) and LN1 + FFN_pre + ReLU + FFN_suf in one launch with the hidden layer recomputed per
// workgroup (47 us against 9.4 + 5.2: 16 workgroups each stream all of W1 behind a 32-row operand with ~8 KB in flight per
// wave -- the chain is bound by dependent L2 round trips, and recomputation multiplies them).
// =================================================================================================
template <bool FRAG>
__global__ __launch_bounds__(256) void linear_skinny_embed_kernel(const int32_t* __restrict__ tok, const float* __restrict__ table,
                                                                  const float* __restrict__ pe, const int32_t* __restrict__ pos_dev,
                                                                  const uint16_t* __restrict__ W, const float* __restrict__ bias,
                                                                  uint16_t* __restrict__ C, uint16_t* __restrict__ H, int M, int N,
                                                                  int K, int V, float scale) {
    __shared__ float part[4][32][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kq = K >> 2;
    const int nrow = n0 + l31, mrow = l31;
    const bool nv = FRAG || nrow < N, mv = mrow < M;
    int t = tok[mv ? mrow : 0];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const int pos = pos_dev[0];
    const float* tp = table + (size_t)t * K + w * kq + hh * 8;
    const float* pp = pe + (size_t)pos * K + w * kq + hh * 8;
    const uint16_t* wp = FRAG ? W + (((size_t)blockIdx.x * (K >> 4) + (size_t)(w * kq >> 4)) * 64 + lane) * 8
                              : W + (size_t)(nv ? nrow : 0) * K + w * kq + hh * 8;
    const int wstep = FRAG ? 32 : 1;                         // elements per unit of k
    f32x16 acc = zero16();
    for (int k0 = 0; k0 < kq; k0 += 16) {
        const u32x4 wf = nv ? *(const u32x4*)(wp + (size_t)k0 * wstep) : u32x4{0, 0, 0, 0};
        const f32x4 a0 = *(const f32x4*)(tp + k0), a1 = *(const f32x4*)(tp + k0 + 4);
        const f32x4 p0 = *(const f32x4*)(pp + k0), p1 = *(const f32x4*)(pp + k0 + 4);
        const float f[8] = {a0.x * scale + p0.x, a0.y * scale + p0.y, a0.z * scale + p0.z, a0.w * scale + p0.w,
                            a1.x * scale + p1.x, a1.y * scale + p1.y, a1.z * scale + p1.z, a1.w * scale + p1.w};
        const u32x4 xf = mv ? pack8(f) : u32x4{0, 0, 0, 0};
        if (blockIdx.x == 0 && mv) *(u32x4*)(H + (size_t)mrow * K + w * kq + hh * 8 + k0) = xf;
        acc = mfma(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf), acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) part[w][crow(r, hh)][l31] = acc[r];
    __syncthreads();
    const int n = tid >> 3, m4 = (tid & 7) * 4;
    if (n0 + n < N) {
        const float bv = bias ? bias[n0 + n] : 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = m4 + k;
            if (m < M) C[(size_t)m * N + n0 + n] = f32_to_bf16(part[0][n][m] + part[1][n][m] + part[2][n][m] + part[3][n][m] + bv);
        }
    }
}

static bool g_attr_set = false;
static void set_attrs() {
    if (g_attr_set) return;
    hipFuncSetAttribute((const void*)linear_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_ring_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_ring_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_dw_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_dw_ring4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_ring4_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_ring4_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_ring4_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_ring4_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, RG_LDS);
    hipFuncSetAttribute((const void*)linear_dx_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_dx_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_dw_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    g_attr_set = true;
}

// The ring kernel pays when its 256 x 256 tiles fill the chip (one persistent workgroup per CU) without much padding.
// MGX_GEMM_RING=0 / 1 forces it off / on where the shape allows (A/B timing; experiment builds only).
static int ring_grid(int M, int NO, int R, void* stream) {
    static int env = -2;
    if (env == -2) env = gemm_knob("MGX_GEMM_RING", -1);
    const int cus = mgx_stream_cu_count(stream);            // a CU-masked stream: one persistent workgroup per CU it may use
    if (env == 0 || R % 32 != 0 || R < 128 || M % 256 != 0 || NO % 256 != 0) return 0;     // whole tiles only
    const long ntm = (M + 255) / 256, ntn = (NO + 255) / 256;
    const long ntiles = ntm * ntn;
    if (env != 1) {
        if (ntiles * 4 < (long)cus * 3) return 0;                                    // < 3/4 of the CUs busy
    }
    return (int)(ntiles < cus ? ntiles : cus);
}

// the four-wave kernel's tile statement runs whole rounds of its two-slot ring, at least two (reduction % 128 == 0, >= 256);
// MGX_RING4=0 keeps the eight-wave kernel (A/B; experiment builds only)
// ... and it does not win everywhere.  Measured per call site inside the training step (profiles/r05_ring4_in_step.txt, cfg2; cfg4 with
// MGX_RING4=0/1/2): it wins 3-7 % on the QKV and output projections, forward and dX (8-24 stages of 64 columns per tile, output 512
// columns or wider; cfg4's 768 x 768 projection with ONE tile per workgroup included), and loses 10-14 % on the FFN's GEMMs: tiles of
// only four stages (reduction 256) pay the statement's fixed costs (parameter block, fragment addresses, first reads: ~1.5 K cycles)
// per 11 K, and with a 256-column output every workgroup streams its own A rows against the same B tile -- two 64 KB requests in
// flight keep the HBM less busy than the eight-wave kernel's four 32 KB ones.  Those stay on the eight-wave kernel.
static bool ring4_shape(int R, int NO, long long out_elems) {
    static int env = -2;
    if (env == -2) env = gemm_knob("MGX_RING4", 1);
    if (env == 0 || R % 128 != 0 || R < 256 || out_elems * 2 >= (1ll << 32)) return false;     // (the epilogue addresses the output with 32-bit offsets)
    return env == 2 || (R >= 512 && NO >= 512);              // MGX_RING4=2 (experiment builds): wherever the shape allows
}

extern "C" int mgx_linear_fwd(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* C, int M, int N,
                              int K, int act, void* stream) {
    MGX_REQUIRE(A && W && C, MGX_ERR_NULL, "mgx_linear_fwd: NULL pointer");
    MGX_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0, MGX_ERR_SHAPE,
                "mgx_linear_fwd: need K%%64==0 and N%%4==0 (got M=%d N=%d K=%d)", M, N, K);
    MGX_REQUIRE(act == 0 || act == 1, MGX_ERR_SHAPE, "mgx_linear_fwd: act must be 0 (none) or 1 (ReLU)");
    set_attrs();
    if (M <= 32) {      // decode-size batches: weight-streaming skinny kernel
        hipLaunchKernelGGL(linear_skinny_kernel<false>, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, A, W, bias, C, M, N,
                           K, act);
        MGX_CHECK_LAUNCH("mgx_linear_fwd");
        return MGX_OK;
    }
    if (const int rg = ring_grid(M, N, K, stream)) {
        if (ring4_shape(K, N, (long long)M * N))
            hipLaunchKernelGGL((linear_ring4_kernel<false, 0>), dim3(rg), dim3(256), RG_LDS, (hipStream_t)stream, A, W, bias,
                               (const uint16_t*)nullptr, (const uint16_t*)nullptr, C, M, N, K, act);
        else
        hipLaunchKernelGGL(linear_ring_kernel<false>, dim3(rg), dim3(512), RG_LDS, (hipStream_t)stream, A, W, bias,
                           (const uint16_t*)nullptr, (const uint16_t*)nullptr, C, M, N, K, act);
        MGX_CHECK_LAUNCH("mgx_linear_fwd");
        return MGX_OK;
    }
    const int nwg = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // Large grids (>= 3 workgroups per CU) run the single-LDS-buffer variant: 32 KiB -> 3 workgroups/CU
    // (+8 % on the QKV projection); small grids keep the double-buffered one (one barrier per step).
    static int sbuf_env = -2;
    if (sbuf_env == -2) sbuf_env = gemm_knob("MGX_GEMM_SINGLE_BUF", -1);
    const bool sbuf = sbuf_env >= 0 ? (sbuf_env != 0) : (nwg >= 768);
    if (sbuf)
        hipLaunchKernelGGL(linear_fwd_kernel<false>, dim3(nwg), dim3(256), LDS_BYTES / 2, (hipStream_t)stream, A, W, bias, C,
                           M, N, K, act);
    else
        hipLaunchKernelGGL(linear_fwd_kernel<true>, dim3(nwg), dim3(256), LDS_BYTES, (hipStream_t)stream, A, W, bias, C, M,
                           N, K, act);
    MGX_CHECK_LAUNCH("mgx_linear_fwd");
    return MGX_OK;
}

extern "C" int mgx_linear_dx(const uint16_t* dY, const uint16_t* W, const uint16_t* relu_y, const uint16_t* addend,
                             uint16_t* dX, int M, int N, int K, void* stream) {
    MGX_REQUIRE(dY && W && dX, MGX_ERR_NULL, "mgx_linear_dx: NULL pointer");
    MGX_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, MGX_ERR_SHAPE,
                "mgx_linear_dx: need N%%8==0 and K%%8==0 (got M=%d N=%d K=%d)", M, N, K);
    set_attrs();
    if (const int rg = (relu_y && addend) ? 0 : ring_grid(M, K, N, stream)) {   // dX [M,K] = dY [M,N] . W [N,K]: reduction over N, W read transposed
        // the ring kernels have ONE straight-line epilogue per operand (addend, else mask): a call with both never gets here (rg = 0
        // above) and takes the 128 x 128 kernel.  relu_y and addend are [M,K] like dX: the epilogue's 32-bit offsets (bounded by
        // ring4_shape's out_elems check) address all three
#define MGX_RING4_DX(PRE) hipLaunchKernelGGL((linear_ring4_kernel<true, PRE>), dim3(rg), dim3(256), RG_LDS, (hipStream_t)stream, dY, W, \
                                             (const float*)nullptr, relu_y, addend, dX, M, K, N, 0)
        if (ring4_shape(N, K, (long long)M * K)) { if (addend) MGX_RING4_DX(2); else if (relu_y) MGX_RING4_DX(1); else MGX_RING4_DX(0); }
#undef MGX_RING4_DX
        else
        hipLaunchKernelGGL(linear_ring_kernel<true>, dim3(rg), dim3(512), RG_LDS, (hipStream_t)stream, dY, W,
                           (const float*)nullptr, relu_y, addend, dX, M, K, N, 0);
        MGX_CHECK_LAUNCH("mgx_linear_dx");
        return MGX_OK;
    }
    const int nwg = ((M + BM - 1) / BM) * ((K + BN - 1) / BN);
    static int sbuf_env = -2;
    if (sbuf_env == -2) sbuf_env = gemm_knob("MGX_GEMM_SINGLE_BUF", -1);
    const bool sbuf = sbuf_env >= 0 ? (sbuf_env != 0) : (nwg >= 768);      // as in the forward: 3 workgroups / CU for big grids
    const bool exact = (N % BK == 0);          // no partial reduction tile: the branch-free load path
#define MGX_DX_LAUNCH(DB, EX, LDS) hipLaunchKernelGGL((linear_dx_kernel<DB, EX>), dim3(nwg), dim3(256), LDS, (hipStream_t)stream, \
                                                      dY, W, relu_y, addend, dX, M, N, K)
    if (sbuf) { if (exact) MGX_DX_LAUNCH(false, true, LDS_BYTES / 2); else MGX_DX_LAUNCH(false, false, LDS_BYTES / 2); }
    else      { if (exact) MGX_DX_LAUNCH(true, true, LDS_BYTES); else MGX_DX_LAUNCH(true, false, LDS_BYTES); }
#undef MGX_DX_LAUNCH
    MGX_CHECK_LAUNCH("mgx_linear_dx");
    return MGX_OK;
}

// Which kernel family a forward / dX call of this shape takes on `stream` (mgx.h: mgx_linear_kernel_id; tests assert that the
// bench-shape calls they check really run the ring kernels of THIS binary).  Mirrors the dispatch of the two entry points above.
extern "C" int mgx_linear_kernel_id(int kind, int M, int N, int K, void* stream) {
    MGX_REQUIRE(kind >= 0 && kind <= 3 && M > 0 && N > 0 && K > 0, MGX_ERR_SHAPE, "mgx_linear_kernel_id: kind 0..3, positive sizes");
    if (kind == 0) {                                       // forward: C [M,N] = A [M,K] . W [N,K]^T
        if (M <= 32) return MGX_GEMM_SKINNY;
        if (!ring_grid(M, N, K, stream)) return MGX_GEMM_TILE128;
        return ring4_shape(K, N, (long long)M * N) ? MGX_GEMM_RING4 : MGX_GEMM_RING8;
    }
    // dX [M,K] = dY [M,N] . W [N,K]; kind 1: no epilogue operand, 2: ReLU mask, 3: residual addend (both: the 128 x 128 kernel)
    if (!ring_grid(M, K, N, stream)) return MGX_GEMM_TILE128;
    return ring4_shape(N, K, (long long)M * K) ? MGX_GEMM_RING4 : MGX_GEMM_RING8;
}

extern "C" int mgx_linear_dw(const uint16_t* dY, const uint16_t* X, float* gW, float* gb, int M, int N, int K,
                             void* stream) {
    MGX_REQUIRE(dY && X && gW, MGX_ERR_NULL, "mgx_linear_dw: NULL pointer");
    MGX_REQUIRE(M > 0 && N > 0 && K > 0 && N % 8 == 0 && K % 8 == 0, MGX_ERR_SHAPE,
                "mgx_linear_dw: need N%%8==0 and K%%8==0 (got M=%d N=%d K=%d)", M, N, K);
    set_attrs();
    const int tiles = ((N + BM - 1) / BM) * ((K + BN - 1) / BN);
    // split M so that the grid has ~target workgroups; chunk is a multiple of 64 rows.  Each split adds
    // one 64 KiB partial tile with fp32 atomics (~1.3 TB/s chip-wide), so fewer, longer splits are better
    // as long as the grid still covers the CUs.
    static int target = -1;
    if (target < 0) {
        target = gemm_knob("MGX_DW_TARGET_WGS", 0);
    }
    // measured on MI355X at M=16384 (tools/gemm_bench.py): ~384 workgroups for many-tile weights (QKV),
    // ~256 for the small ones
    const int tgt = target > 0 ? target : (tiles >= 32 ? 384 : 256);
    int splits = (tgt + tiles - 1) / tiles;
    int mchunk = ((M + splits - 1) / splits + 63) / 64 * 64;
    if (mchunk < 64) mchunk = 64;
    splits = (M + mchunk - 1) / mchunk;
    int rc;
    long long* det = mgx_det_scratch((size_t)N * K + N, stream, &rc);      // deterministic mode: integer atomics + fold
    if (rc != MGX_OK) return rc;
    hipLaunchKernelGGL(linear_dw_kernel, dim3(tiles * splits), dim3(256), LDS_BYTES, (hipStream_t)stream, dY, X, gW, gb, M,
                       N, K, mchunk, tiles, det, det ? det + (size_t)N * K : nullptr);
    if (det) {
        launch_det_fold(det, gW, (size_t)N * K, 1.f, 1, (hipStream_t)stream);
        if (gb) launch_det_fold(det + (size_t)N * K, gb, (size_t)N, 1.f, 1, (hipStream_t)stream);
    }
    MGX_CHECK_LAUNCH("mgx_linear_dw");
    return MGX_OK;
}

// The ring kernel takes the weights that fill their 256 x 256 tiles at least to 60 % (N, K multiples of 8, at least 64): every
// encoder-block projection tiles exactly; the vocabulary projection (448 x 512: 87.5 %) and cfg4's FFN weights (384 x 768: 75 %)
// have a ragged last tile row / column, whose DMA offsets the kernel clamps and whose outer part the fix-up pass skips -- the wasted
// MFMA work costs less than the 128 x 128 kernel's atomics and its half-idle MFMA pipe (round 5: 148 -> ~75 us for the vocabulary
// projection at M = 131072).
static bool dw_ring_shape(int N, int K) {
    if (N % 8 != 0 || K % 8 != 0 || N < 64 || K < 64) return false;
    const long long tiles = (long long)((N + 255) / 256) * ((K + 255) / 256);
    return (long long)N * K * 10 >= tiles * 65536 * 6;
}
// plan: number of M-splits so that tiles x splits fills the CUs once; every split gets at least one 32-row step.
static bool dw_ring_plan(const mgx_dw_problem* problems, int count, int M, DwRing* out, int cus) {
    static int env = -2;
    if (env == -2) env = gemm_knob("MGX_GEMM_RING", -1);
    if (env == 0 || M % 32 != 0 || M < 4096) return false;
    DwRing g;
    g.n = count;
    g.ragged = 0;
    g.first_tile[0] = 0;
    for (int i = 0; i < count; ++i) {
        const mgx_dw_problem& q = problems[i];
        if (!dw_ring_shape(q.N, q.K)) return false;
        g.dY[i] = q.dY; g.X[i] = q.X; g.gW[i] = q.gW; g.gb[i] = q.gb; g.detb[i] = nullptr; g.N[i] = q.N; g.K[i] = q.K;
        g.ragged |= (q.N % 256 != 0 || q.K % 256 != 0);
        g.first_tile[i + 1] = g.first_tile[i] + ((q.N + 255) / 256) * ((q.K + 255) / 256);
    }
    const int tiles = g.first_tile[count];
    if (tiles > cus) return false;
    const int total = M / 32;
    int splits = cus / tiles;
    if (splits > total) splits = total;
    g.steps_per_split = (total + splits - 1) / splits;
    g.splits = (total + g.steps_per_split - 1) / g.steps_per_split;
    if (out) *out = g;
    return true;
}

// The problems of a group whose weights fill their 256 x 256 tiles (dw_ring_shape) go to the ring kernel, the others to the
// 128 x 128 grouped kernel -- until round 4 one such weight sent the whole block there (cfg4: 219 us per block, 12 % of the step).
static int dw_split(const mgx_dw_problem* problems, int count, mgx_dw_problem* ring, int* nring, mgx_dw_problem* rest, int* nrest) {
    *nring = *nrest = 0;
    for (int i = 0; i < count; ++i) {
        if (dw_ring_shape(problems[i].N, problems[i].K)) ring[(*nring)++] = problems[i];
        else rest[(*nrest)++] = problems[i];
    }
    return *nring;
}

extern "C" size_t mgx_linear_dw_grouped_workspace(const mgx_dw_problem* problems, int count, int M) {
    DwRing g;
    if (!problems || count <= 0 || count > MGX_DW_MAX_GROUP) return 0;
    mgx_dw_problem ring[MGX_DW_MAX_GROUP], rest[MGX_DW_MAX_GROUP];
    int nr, ns;
    // (no stream here: planned for the whole device -- a CU-masked stream runs fewer M-splits, so this is an upper bound for it)
    if (!dw_split(problems, count, ring, &nr, rest, &ns) || !dw_ring_plan(ring, nr, M, &g, mgx_stream_cu_count(nullptr))) return 0;
    return (size_t)g.first_tile[g.n] * g.splits * 65536 * sizeof(float);
}

extern "C" int mgx_linear_dw_grouped(const mgx_dw_problem* problems_in, int count_in, int M, void* workspace, size_t ws_bytes,
                                     void* stream) {
    MGX_REQUIRE(problems_in && count_in > 0 && count_in <= MGX_DW_MAX_GROUP && M > 0, MGX_ERR_SHAPE,
                "mgx_linear_dw_grouped: need 1..%d problems and M > 0 (got %d, M=%d)", MGX_DW_MAX_GROUP, count_in, M);
    // a mixed group: its ring-shaped weights first (one recursive call on that sub-group), the others below
    mgx_dw_problem ring_p[MGX_DW_MAX_GROUP], rest_p[MGX_DW_MAX_GROUP];
    int nring, nrest;
    const mgx_dw_problem* problems = problems_in;
    int count = count_in;
    if (dw_split(problems_in, count_in, ring_p, &nring, rest_p, &nrest) && nrest > 0 && dw_ring_plan(ring_p, nring, M, nullptr, mgx_stream_cu_count(mgx_deterministic() ? nullptr : stream))) {
        if (int rc = mgx_linear_dw_grouped(ring_p, nring, M, workspace, ws_bytes, stream)) return rc;
        problems = rest_p;
        count = nrest;
    }
    DwGroup g;
    g.n = count;
    g.first_tile[0] = 0;
    for (int i = 0; i < count; ++i) {
        const mgx_dw_problem& q = problems[i];
        MGX_REQUIRE(q.dY && q.X && q.gW, MGX_ERR_NULL, "mgx_linear_dw_grouped: NULL pointer in problem %d", i);
        MGX_REQUIRE(q.N > 0 && q.K > 0 && q.N % 8 == 0 && q.K % 8 == 0, MGX_ERR_SHAPE,
                    "mgx_linear_dw_grouped: need N%%8==0 and K%%8==0 (problem %d: N=%d K=%d)", i, q.N, q.K);
        g.dY[i] = q.dY; g.X[i] = q.X; g.gW[i] = q.gW; g.gb[i] = q.gb; g.N[i] = q.N; g.K[i] = q.K;
        g.first_tile[i + 1] = g.first_tile[i] + ((q.N + BM - 1) / BM) * ((q.K + BN - 1) / BN);
    }
    set_attrs();
    DwRing rg;
    // the M-splits are planned for the CUs of the stream the call is issued on -- except in deterministic mode: the partial tiles
    // are added in split order, so the number of splits is part of the result's bits, and a run with the weight gradients on a
    // CU-masked side stream must equal the one-stream run bit for bit (tests/test_gpu_dp.py): there the plan is the whole device's
    if (dw_ring_plan(problems, count, M, &rg, mgx_stream_cu_count(mgx_deterministic() ? nullptr : stream))) {
        const size_t need = (size_t)rg.first_tile[rg.n] * rg.splits * 65536 * sizeof(float);
        MGX_REQUIRE(workspace && ws_bytes >= need && ((uintptr_t)workspace & 15) == 0, MGX_ERR_SHAPE,
                    "mgx_linear_dw_grouped: workspace must be 16-byte aligned and >= mgx_linear_dw_grouped_workspace() = %zu bytes "
                    "(got %zu)", need, ws_bytes);
        const int tiles = rg.first_tile[rg.n];
        // deterministic mode: the weight tiles already are (partial tiles in the workspace, added in split order by the fix-up
        // pass); the bias gradients, which the M-splits add with atomics, go through the fixed-point scratch
        size_t nb = 0;
        for (int i = 0; i < count; ++i) nb += problems[i].gb ? (size_t)problems[i].N : 0;
        int rc;
        long long* det = nb ? mgx_det_scratch(nb, stream, &rc) : (rc = MGX_OK, nullptr);
        if (rc != MGX_OK) return rc;
        if (det) {
            size_t o = 0;
            for (int i = 0; i < count; ++i)
                if (problems[i].gb) { rg.detb[i] = det + o; o += (size_t)problems[i].N; }
        }
        static int four = -1;                              // 0: the eight-wave HIP kernel (A/B, experiment builds)
        if (four < 0) four = gemm_knob("MGX_DW_RING4", 1);
        if (four || rg.ragged)
            hipLaunchKernelGGL(linear_dw_ring4_kernel, dim3(tiles * rg.splits), dim3(256), RG_LDS, (hipStream_t)stream, rg, M,
                               (float*)workspace);
        else
            hipLaunchKernelGGL(linear_dw_ring_kernel, dim3(tiles * rg.splits), dim3(512), RG_LDS, (hipStream_t)stream, rg, M,
                               (float*)workspace);
        hipLaunchKernelGGL(dw_fixup_kernel, dim3(64, tiles), dim3(256), 0, (hipStream_t)stream, rg, (const float*)workspace);
        if (det)
            for (int i = 0; i < count; ++i)
                if (rg.detb[i]) launch_det_fold(rg.detb[i], problems[i].gb, (size_t)problems[i].N, 1.f, 1, (hipStream_t)stream);
        MGX_CHECK_LAUNCH("mgx_linear_dw_grouped");
        return MGX_OK;
    }
    if (mgx_deterministic()) {          // no ring plan for these shapes: one deterministic launch per weight
        for (int i = 0; i < count; ++i)
            if (int rc = mgx_linear_dw(problems[i].dY, problems[i].X, problems[i].gW, problems[i].gb, M, problems[i].N, problems[i].K, stream))
                return rc;
        return MGX_OK;
    }
    const int tiles = g.first_tile[count];
    static int target = -1;
    if (target < 0) {
        target = gemm_knob("MGX_DW_GROUP_WGS", 480);       // ~2 workgroups on every CU
    }
    int splits = (target + tiles - 1) / tiles;
    int mchunk = ((M + splits - 1) / splits + 63) / 64 * 64;
    if (mchunk < 64) mchunk = 64;
    splits = (M + mchunk - 1) / mchunk;
    hipLaunchKernelGGL(linear_dw_grouped_kernel, dim3(tiles * splits), dim3(256), LDS_BYTES, (hipStream_t)stream, g, M, mchunk);
    MGX_CHECK_LAUNCH("mgx_linear_dw_grouped");
    return MGX_OK;
}

extern "C" int mgx_linear_ln_fwd(const uint16_t* X, const uint16_t* RES, const float* gamma, const float* beta, float eps,
                                 const uint16_t* W, const float* bias, uint16_t* C, uint16_t* Z, int M, int N, int K, int act,
                                 void* stream) {
    MGX_REQUIRE(X && RES && gamma && beta && W && C && Z, MGX_ERR_NULL, "mgx_linear_ln_fwd: NULL pointer");
    MGX_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0 && K % 64 == 0 && K <= 64 * SKLN_MAXF, MGX_ERR_SHAPE,
                "mgx_linear_ln_fwd: need 0<M<=32, K%%64==0, K<=%d (got M=%d N=%d K=%d)", 64 * SKLN_MAXF, M, N, K);
    MGX_REQUIRE(act == 0 || act == 1, MGX_ERR_SHAPE, "mgx_linear_ln_fwd: act must be 0 (none) or 1 (ReLU)");
    hipLaunchKernelGGL(linear_skinny_ln_kernel<false>, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, X, RES, gamma, beta, eps, W,
                       bias, C, Z, M, N, K, act);
    MGX_CHECK_LAUNCH("mgx_linear_ln_fwd");
    return MGX_OK;
}

// ---- decode-size projections with the weight in MFMA fragment order (mgx.h; rows zero-padded to a multiple of 32) --------------
extern "C" int mgx_skinny_fwd_frag(const uint16_t* A, const uint16_t* Wf, const float* bias, uint16_t* C, int M, int N, int K,
                                   int act, void* stream) {
    MGX_REQUIRE(A && Wf && C, MGX_ERR_NULL, "mgx_skinny_fwd_frag: NULL pointer");
    MGX_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0 && K % 64 == 0, MGX_ERR_SHAPE,
                "mgx_skinny_fwd_frag: need 0<M<=32, K%%64==0 (got M=%d N=%d K=%d)", M, N, K);
    MGX_REQUIRE(act == 0 || act == 1, MGX_ERR_SHAPE, "mgx_skinny_fwd_frag: act must be 0 (none) or 1 (ReLU)");
    hipLaunchKernelGGL(linear_skinny_kernel<true>, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, A, Wf, bias, C, M, N, K, act);
    MGX_CHECK_LAUNCH("mgx_skinny_fwd_frag");
    return MGX_OK;
}

extern "C" int mgx_linear_ln_fwd_frag(const uint16_t* X, const uint16_t* RES, const float* gamma, const float* beta, float eps,
                                      const uint16_t* Wf, const float* bias, uint16_t* C, uint16_t* Z, int M, int N, int K, int act,
                                      void* stream) {
    MGX_REQUIRE(X && RES && gamma && beta && Wf && C && Z, MGX_ERR_NULL, "mgx_linear_ln_fwd_frag: NULL pointer");
    MGX_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0 && K % 64 == 0 && K <= 64 * SKLN_MAXF, MGX_ERR_SHAPE,
                "mgx_linear_ln_fwd_frag: need 0<M<=32, K%%64==0, K<=%d (got M=%d N=%d K=%d)", 64 * SKLN_MAXF, M, N, K);
    MGX_REQUIRE(act == 0 || act == 1, MGX_ERR_SHAPE, "mgx_linear_ln_fwd_frag: act must be 0 (none) or 1 (ReLU)");
    hipLaunchKernelGGL(linear_skinny_ln_kernel<true>, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, X, RES, gamma, beta, eps,
                       Wf, bias, C, Z, M, N, K, act);
    MGX_CHECK_LAUNCH("mgx_linear_ln_fwd_frag");
    return MGX_OK;
}


extern "C" int mgx_decode_embed_linear(const int32_t* tok, const float* table, const float* pe, const int32_t* pos_dev,
                                       const uint16_t* W, const float* bias, uint16_t* C, uint16_t* H, int M, int N, int K,
                                       int V, void* stream) {
    MGX_REQUIRE(tok && table && pe && pos_dev && W && C && H, MGX_ERR_NULL, "mgx_decode_embed_linear: NULL pointer");
    MGX_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0 && K % 64 == 0 && V > 0, MGX_ERR_SHAPE,
                "mgx_decode_embed_linear: need 0<M<=32, K%%64==0 (got M=%d N=%d K=%d)", M, N, K);
    hipLaunchKernelGGL(linear_skinny_embed_kernel<false>, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, tok, table, pe, pos_dev,
                       W, bias, C, H, M, N, K, V, sqrtf((float)K));
    MGX_CHECK_LAUNCH("mgx_decode_embed_linear");
    return MGX_OK;
}

extern "C" int mgx_decode_embed_linear_frag(const int32_t* tok, const float* table, const float* pe, const int32_t* pos_dev,
                                            const uint16_t* Wf, const float* bias, uint16_t* C, uint16_t* H, int M, int N, int K,
                                            int V, void* stream) {
    MGX_REQUIRE(tok && table && pe && pos_dev && Wf && C && H, MGX_ERR_NULL, "mgx_decode_embed_linear_frag: NULL pointer");
    MGX_REQUIRE(M > 0 && M <= 32 && N > 0 && K > 0 && K % 64 == 0 && V > 0, MGX_ERR_SHAPE,
                "mgx_decode_embed_linear_frag: need 0<M<=32, K%%64==0 (got M=%d N=%d K=%d)", M, N, K);
    hipLaunchKernelGGL(linear_skinny_embed_kernel<true>, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, tok, table, pe,
                       pos_dev, Wf, bias, C, H, M, N, K, V, sqrtf((float)K));
    MGX_CHECK_LAUNCH("mgx_decode_embed_linear_frag");
    return MGX_OK;
}
