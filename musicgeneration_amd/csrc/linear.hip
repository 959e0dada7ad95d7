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
#include <stdlib.h>
#include "rel_attn_common.hpp"
#include "mgx.h"

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
                     float* __restrict__ gb, int M, int N, int K, int tile, int mbeg, int mend, char* smem) {
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
    u32x4 areg[4], breg[4];
    const uint16_t* ap = dY + (size_t)wrow * N + min(n0 + ch * 8, N - 8);
    const uint16_t* bp = X + (size_t)wrow * K + min(k0 + ch * 8, K - 8);
    auto load_tiles = [&](int mm) {
        const int gn = n0 + ch * 8, gk = k0 + ch * 8;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (EXACT) {
                areg[i] = *(const u32x4*)(ap + (size_t)(mm + 16 * i) * N);
                breg[i] = *(const u32x4*)(bp + (size_t)(mm + 16 * i) * K);
            } else {
                const int gm = mm + wrow + 16 * i;
                areg[i] = (gm < mend && gn < N) ? *(const u32x4*)(dY + (size_t)gm * N + gn) : u32x4{0, 0, 0, 0};
                breg[i] = (gm < mend && gk < K) ? *(const u32x4*)(X + (size_t)gm * K + gk) : u32x4{0, 0, 0, 0};
            }
        }
    };
    // bias gradient gb[n] += sum_m dY[m][n]: the workgroups of the first k-tile column add up the dY rows they stage anyway
    const bool do_bias = (gb != nullptr) && (tk == 0) && (!EXACT || n0 + ch * 8 < N);
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto store_tiles = [&](int buf) {
        char* at = smem + buf * 2 * IMG;
        char* bt = at + IMG;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wrow + 16 * i;
            const int off = ((row >> 5) * 2 + sub_c) * TILE_BYTES + imgT_off(row & 31, slot);
            *(u32x4*)(at + off) = areg[i];
            *(u32x4*)(bt + off) = breg[i];
        }
        if (do_bias) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float f[8];
                unpack8(areg[i], f);
#pragma unroll
                for (int k = 0; k < 8; ++k) bsum[k] += f[k];
            }
        }
    };
    f32x16 acc[2][2];
    zero_acc(acc);
    const int nm = (mend - mbeg + BK - 1) / BK;
    if (nm > 0) {
        load_tiles(mbeg);
        store_tiles(0);
    }
    __syncthreads();
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
    for (int mt = 0; mt + 1 < nm; ++mt) {                // all reduction tiles but the last: branch-free body
        const int cur = mt & 1;
        load_tiles(mbeg + (mt + 1) * BK);
        __builtin_amdgcn_sched_barrier(0);               // keep the prefetch AHEAD of the MFMAs
        multiply(cur);
        store_tiles(cur ^ 1);
        __syncthreads();
    }
    if (nm > 0) multiply((nm - 1) & 1);
    if (do_bias) {      // lanes with equal (tid & 15) hold the same 8 columns: fold lane bits 4,5, then one atomic per wave
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bsum[k] += __shfl_xor(bsum[k], 16, 64);
            bsum[k] += __shfl_xor(bsum[k], 32, 64);
        }
        if (lane < 16 && n0 + ch * 8 < N) {
#pragma unroll
            for (int k = 0; k < 8; ++k) atomicAdd(gb + n0 + ch * 8 + k, bsum[k]);
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
                if (n < N) atomicAdd(gW + (size_t)n * K + k, acc[rt][ct][r]);
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void linear_dw_kernel(const uint16_t* __restrict__ dY,
                                                           const uint16_t* __restrict__ X,
                                                           float* __restrict__ gW, float* __restrict__ gb, int M, int N,
                                                           int K, int mchunk, int tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // 1-D grid of tiles x splits units, unit = split * tiles + tile; each XCD walks a contiguous run of units, i.e.
    // (mostly) the tiles of ONE row chunk: the dY / X blocks those tiles share are fetched into that XCD's L2 once
    const int u = xcd_remap(blockIdx.x, gridDim.x);
    const int mbeg = (u / tiles) * mchunk;
    if (M % mchunk == 0 && mchunk % BK == 0) dw_tile<true>(dY, X, gW, gb, M, N, K, u % tiles, mbeg, mbeg + mchunk, smem);
    else dw_tile<false>(dY, X, gW, gb, M, N, K, u % tiles, mbeg, min(M, mbeg + mchunk), smem);
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
// skinny forward (M <= 32: the decode path's projections).  The weights are streamed exactly once:
// workgroup = 32 output columns, its 4 waves split K; W rows and x rows go straight from global/L2 into
// MFMA fragments (no LDS staging, no barriers in the loop); the four partial tiles are combined in LDS.
//   D[n][m] = sum_k W[n][k] x[m][k]   (A = W rows, B = x^T)
// =================================================================================================
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
    const bool nv = nrow < N, mv = mrow < M;
    const uint16_t* wp = W + (size_t)(nv ? nrow : 0) * K + w * kq + hh * 8;
    const uint16_t* xp = A + (size_t)(mv ? mrow : 0) * K + w * kq + hh * 8;
    f32x16 acc = zero16();
    for (int k0 = 0; k0 < kq; k0 += 64) {                    // 4 k-steps per iteration, 8 loads in flight
        u32x4 wf[4], xf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bool in = k0 + 16 * ks < kq;
            wf[ks] = (nv && in) ? *(const u32x4*)(wp + k0 + 16 * ks) : u32x4{0, 0, 0, 0};
            xf[ks] = (mv && in) ? *(const u32x4*)(xp + k0 + 16 * ks) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
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
    const bool nv = nrow < N, mv = mrow < M;
    const size_t xoff = (size_t)(mv ? mrow : 0) * K + w * kq + hh * 8;
    float z[SKLN_MAXF][8];
    float s1 = 0.f, s2 = 0.f;
    // the weight fragments are requested first: their latency hides under the statistics
    const uint16_t* wp = W + (size_t)(nv ? nrow : 0) * K + w * kq + hh * 8;
    u32x4 wf[SKLN_MAXF];
#pragma unroll
    for (int f = 0; f < SKLN_MAXF; ++f)
        if (f < nf) wf[f] = nv ? *(const u32x4*)(wp + 16 * f) : u32x4{0, 0, 0, 0};
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

static bool g_attr_set = false;
static void set_attrs() {
    if (g_attr_set) return;
    hipFuncSetAttribute((const void*)linear_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_dx_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_dx_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)linear_dw_grouped_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    g_attr_set = true;
}

extern "C" int mgx_linear_fwd(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* C, int M, int N,
                              int K, int act, void* stream) {
    MGX_REQUIRE(A && W && C, MGX_ERR_NULL, "mgx_linear_fwd: NULL pointer");
    MGX_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0, MGX_ERR_SHAPE,
                "mgx_linear_fwd: need K%%64==0 and N%%4==0 (got M=%d N=%d K=%d)", M, N, K);
    MGX_REQUIRE(act == 0 || act == 1, MGX_ERR_SHAPE, "mgx_linear_fwd: act must be 0 (none) or 1 (ReLU)");
    set_attrs();
    if (M <= 32) {      // decode-size batches: weight-streaming skinny kernel
        hipLaunchKernelGGL(linear_skinny_kernel, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, A, W, bias, C, M, N,
                           K, act);
        MGX_CHECK_LAUNCH("mgx_linear_fwd");
        return MGX_OK;
    }
    const int nwg = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    // Large grids (>= 3 workgroups per CU) run the single-LDS-buffer variant: 32 KiB -> 3 workgroups/CU
    // (+8 % on the QKV projection); small grids keep the double-buffered one (one barrier per step).
    static int sbuf_env = -2;
    if (sbuf_env == -2) { const char* e = getenv("MGX_GEMM_SINGLE_BUF"); sbuf_env = e ? atoi(e) : -1; }
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
    const int nwg = ((M + BM - 1) / BM) * ((K + BN - 1) / BN);
    static int sbuf_env = -2;
    if (sbuf_env == -2) { const char* e = getenv("MGX_GEMM_SINGLE_BUF"); sbuf_env = e ? atoi(e) : -1; }
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
        const char* e = getenv("MGX_DW_TARGET_WGS");
        target = e ? atoi(e) : 0;
    }
    // measured on MI355X at M=16384 (tools/gemm_bench.py): ~384 workgroups for many-tile weights (QKV),
    // ~256 for the small ones
    const int tgt = target > 0 ? target : (tiles >= 32 ? 384 : 256);
    int splits = (tgt + tiles - 1) / tiles;
    int mchunk = ((M + splits - 1) / splits + 63) / 64 * 64;
    if (mchunk < 64) mchunk = 64;
    splits = (M + mchunk - 1) / mchunk;
    hipLaunchKernelGGL(linear_dw_kernel, dim3(tiles * splits), dim3(256), LDS_BYTES, (hipStream_t)stream, dY, X, gW, gb, M,
                       N, K, mchunk, tiles);
    MGX_CHECK_LAUNCH("mgx_linear_dw");
    return MGX_OK;
}

extern "C" int mgx_linear_dw_grouped(const mgx_dw_problem* problems, int count, int M, void* stream) {
    MGX_REQUIRE(problems && count > 0 && count <= MGX_DW_MAX_GROUP && M > 0, MGX_ERR_SHAPE,
                "mgx_linear_dw_grouped: need 1..%d problems and M > 0 (got %d, M=%d)", MGX_DW_MAX_GROUP, count, M);
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
    const int tiles = g.first_tile[count];
    static int target = -1;
    if (target < 0) {
        const char* e = getenv("MGX_DW_GROUP_WGS");
        target = e ? atoi(e) : 480;                        // ~2 workgroups on every CU
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
    hipLaunchKernelGGL(linear_skinny_ln_kernel, dim3((N + 31) / 32), dim3(256), 0, (hipStream_t)stream, X, RES, gamma, beta, eps, W,
                       bias, C, Z, M, N, K, act);
    MGX_CHECK_LAUNCH("mgx_linear_ln_fwd");
    return MGX_OK;
}

