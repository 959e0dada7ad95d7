// HBM-bound satellites of the hot path: embedding+PE, residual+LayerNorm, smoothed CE, Adam.
// All are streaming kernels: 16-byte vector loads/stores, one wave per row for the row ops,
// wave-shuffle reductions, fp32 statistics.  Roofline: HBM (algorithmic bytes in DESIGN.md).
#include <type_traits>
#include "mgx_common.hpp"

// =================================================================================================
// K1  out[r, :] = dropout(table[tok[r], :] * sqrt(d) + pe[r % L, :])        layers.py:226-229
// =================================================================================================
__global__ __launch_bounds__(256) void embed_pe_fwd_kernel(
    const int32_t* __restrict__ tok, const float* __restrict__ table, const float* __restrict__ pe,
    uint16_t* __restrict__ out, int rows, int L, int d, int V, float scale, DropCfg dc) {
    const int gpr = d >> 3;   // 8-element groups per row
    const long total = (long)rows * gpr;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
        const int r = (int)(g / gpr), c = (int)(g % gpr) * 8;
        int t = tok[r];
        t = t < 0 ? 0 : (t >= V ? V - 1 : t);   // host validates; clamp keeps the load in bounds
        const f32x4* tp = (const f32x4*)(table + (size_t)t * d + c);
        const f32x4* pp = (const f32x4*)(pe + (size_t)(r % L) * d + c);
        f32x4 a0 = tp[0], a1 = tp[1], p0 = pp[0], p1 = pp[1];
        float f[8] = {a0.x * scale + p0.x, a0.y * scale + p0.y, a0.z * scale + p0.z, a0.w * scale + p0.w,
                      a1.x * scale + p1.x, a1.y * scale + p1.y, a1.z * scale + p1.z, a1.w * scale + p1.w};
        if (dc.thr16) {
            float m[8];
            drop_mult8(dc, (uint32_t)g, m);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] *= m[k];
        }
        *(u32x4*)(out + (size_t)r * d + c) = pack8(f);
    }
}

// dtable[v, :] += sqrt(d) * sum_{r: tok[r]==v} dropmask * dout[r, :]     block (v, s) = vocab row v, token range s of
// gridDim.y (the tokens are re-scanned from L2 by every block of the range).  The range is scanned in chunks of 4096 into
// an LDS hit list, then the block's waves gather the hit rows in parallel (wave = one hit row at a time, lane = 8 columns;
// d <= 512 per pass), the four partial sums are folded through LDS and the block adds its sum with one fp32 atomic per
// element -- gridDim.y adds per element, not one per token.  (One block per vocabulary row, as until round 3, is a chain of
// dependent scan / gather phases over the whole batch: 235 us at cfg2 / batch 64 for a 67 MB read.)
constexpr int EB_CHUNK = 4096;
constexpr long long EB_BAD_MARK = (long long)0x8000000000000000ull;   // deterministic mode: "this wave saw a value with no fixed-point image"
// DET (deterministic mode): the order of the LDS hit list depends on the arrival order of the LDS atomics and the token
// ranges' sums arrive in any order, so every addend (one bf16 gradient element, times its dropout multiplier) is accumulated as
// a 64-bit fixed-point integer, in the thread, across the waves and across the ranges: the sum is independent of all three
// orders.  `det` is the [V, d] int64 image of dtable's update (folded in, times sqrt(d), by the caller).
template <bool DET>
__global__ __launch_bounds__(256) void embed_bwd_kernel(
    const int32_t* __restrict__ tok, const uint16_t* __restrict__ dout, float* __restrict__ dtable,
    int rows, int d, float scale, DropCfg dc, long long* __restrict__ det) {
    const int v = blockIdx.x;
    const int gpr = d >> 3;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int per = ((rows + (int)gridDim.y - 1) / (int)gridDim.y + EB_CHUNK - 1) / EB_CHUNK * EB_CHUNK;
    const int r_lo = blockIdx.y * per, r_hi = min(rows, r_lo + per);
    __shared__ int hits[EB_CHUNK];
    __shared__ int nhit;
    using acc_t = typename std::conditional<DET, long long, float>::type;
    __shared__ acc_t part[3][64][8];
    for (int g0 = 0; g0 < gpr; g0 += 64) {                  // column pass (one for d <= 512)
        const int gi = g0 + lane;
        acc_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        unsigned bad = 0;                                  // DET: bit q = a contribution to column q had no fixed-point image (NaN / Inf / huge)
        for (int base = r_lo; base < r_hi; base += EB_CHUNK) {
            if (tid == 0) nhit = 0;
            __syncthreads();
            {   // the thread's 16 tokens of the chunk are requested together (as a loop of load-compare-branch they were 16 dependent
                // L2 round trips per chunk: most of this kernel's time)
                int tk[EB_CHUNK / 256];
#pragma unroll
                for (int i = 0; i < EB_CHUNK / 256; ++i) {
                    const int r = base + tid + 256 * i;
                    tk[i] = r < r_hi ? tok[r] : -1;
                }
#pragma unroll
                for (int i = 0; i < EB_CHUNK / 256; ++i)
                    if (tk[i] == v) hits[atomicAdd(&nhit, 1)] = base + tid + 256 * i;
            }
            __syncthreads();
            const int n = nhit;
            if (gi < gpr) {
                // four hit rows in flight per wave (every row is a dependent 1 KB gather from HBM: one at a time, a block's
                // ~24 rows were six serial round trips per wave)
                for (int k0 = w; k0 < n; k0 += 16) {
                    u32x4 raw[4];
                    int rrs[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int k = k0 + 4 * j;
                        rrs[j] = hits[min(k, n - 1)];
                        raw[j] = *(const u32x4*)(dout + (size_t)rrs[j] * d + gi * 8);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (k0 + 4 * j >= n) break;
                        float f[8];
                        unpack8(raw[j], f);
                        if (dc.thr16) {
                            float m[8];
                            drop_mult8(dc, (uint32_t)((size_t)rrs[j] * gpr + gi), m);
#pragma unroll
                            for (int q = 0; q < 8; ++q) f[q] *= m[q];
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            if (DET) {
                                if (det_representable(f[q])) acc[q] += (acc_t)__float2ll_rn(f[q] * MGX_DET_SCALE);
                                else bad |= 1u << q;
                            } else {
                                acc[q] += (acc_t)f[q];
                            }
                        }
                    }
                }
            }
            __syncthreads();
        }
        if (w > 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) part[w - 1][lane][q] = (DET && ((bad >> q) & 1u)) ? (acc_t)EB_BAD_MARK : acc[q];
        }
        __syncthreads();
        if (w == 0 && gi < gpr) {
            float* dp = dtable + (size_t)v * d + gi * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (DET) {
                    long long* dq = det + (size_t)v * d + gi * 8 + q;
                    const long long p0 = (long long)part[0][lane][q], p1 = (long long)part[1][lane][q], p2 = (long long)part[2][lane][q];
                    if (((bad >> q) & 1u) || p0 == EB_BAD_MARK || p1 == EB_BAD_MARK || p2 == EB_BAD_MARK) { det_poison(dq); continue; }
                    const long long isum = (long long)acc[q] + p0 + p1 + p2;
                    if (isum != 0) atomicAdd((unsigned long long*)dq, (unsigned long long)isum);
                } else {
                    const float sum = (float)(acc[q] + part[0][lane][q] + part[1][lane][q] + part[2][lane][q]) * scale;
                    if (gridDim.y == 1) dp[q] += sum;
                    else if (sum != 0.f) atomicAdd(dp + q, sum);
                }
            }
        }
        __syncthreads();
    }
}
// NOTE on determinism: hits[] order depends on LDS atomic arrival order and the ranges' sums arrive in any order, so the
// fp32 sum order can vary between launches by rounding only; values are identical to ~1 ulp of the fp32 sum.

extern "C" int mgx_embed_pe_fwd(const int32_t* tok, const float* table, const float* pe, uint16_t* out,
                                int B, int L, int d, int V, float p_drop, uint64_t seed, void* stream) {
    MGX_REQUIRE(tok && table && pe && out, MGX_ERR_NULL, "mgx_embed_pe_fwd: NULL pointer");
    MGX_REQUIRE(B > 0 && L > 0 && V > 0 && d > 0 && d % 8 == 0, MGX_ERR_SHAPE,
                "mgx_embed_pe_fwd: need B,L,V>0 and d%%8==0 (got B=%d L=%d d=%d V=%d)", B, L, d, V);
    const long total = (long)B * L * (d / 8);
    int grid = (int)((total + 255) / 256);
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, tok, table, pe, out,
                       B * L, L, d, V, sqrtf((float)d), make_drop(p_drop, seed));
    MGX_CHECK_LAUNCH("mgx_embed_pe_fwd");
    return MGX_OK;
}

extern "C" int mgx_embed_bwd(const int32_t* tok, const uint16_t* dout, float* dtable, int B, int L, int d, int V,
                             float p_drop, uint64_t seed, void* stream) {
    MGX_REQUIRE(tok && dout && dtable, MGX_ERR_NULL, "mgx_embed_bwd: NULL pointer");
    MGX_REQUIRE(B > 0 && L > 0 && V > 0 && d > 0 && d % 8 == 0, MGX_ERR_SHAPE,
                "mgx_embed_bwd: need B,L,V>0 and d%%8==0 (got B=%d L=%d d=%d V=%d)", B, L, d, V);
    // token ranges: enough blocks to fill the chip several times over, at least two chunks of tokens each
    const long rows = (long)B * L;
    int split = (int)std::min<long>(16, std::max<long>(1, rows / (2 * EB_CHUNK)));
    while (split > 1 && (long)V * split > 8192) --split;
    int rc;
    long long* det = mgx_det_scratch((size_t)V * d, stream, &rc);
    if (rc != MGX_OK) return rc;
    if (det) {
        hipLaunchKernelGGL(embed_bwd_kernel<true>, dim3(V, split), dim3(256), 0, (hipStream_t)stream, tok, dout, dtable, B * L, d,
                           sqrtf((float)d), make_drop(p_drop, seed), det);
        launch_det_fold(det, dtable, (size_t)V * d, sqrtf((float)d), 1, (hipStream_t)stream);
    } else {
        hipLaunchKernelGGL(embed_bwd_kernel<false>, dim3(V, split), dim3(256), 0, (hipStream_t)stream, tok, dout, dtable, B * L, d,
                           sqrtf((float)d), make_drop(p_drop, seed), (long long*)nullptr);
    }
    MGX_CHECK_LAUNCH("mgx_embed_bwd");
    return MGX_OK;
}

// =================================================================================================
// A3  key-padding bitmap                                                     utils.py:73-77
// =================================================================================================
__global__ void pad_bitmap_kernel(const int32_t* __restrict__ tok, uint32_t* __restrict__ bits, uint32_t* __restrict__ flag,
                                  int total, int L, int pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // one thread per token, 64 | blockDim
    const bool p = (i < total) && (tok[i] == pad);
    const unsigned long long m = __ballot(p);
    const int lane = threadIdx.x & 63;
    if (i < total) {
        if (lane == 0) bits[i >> 5] = (uint32_t)m;
        if (lane == 32) bits[i >> 5] = (uint32_t)(m >> 32);
    }
    // LEADING padding is the one input class outside the parity contract: a row that starts with a pad has
    // queries whose every visible key is masked (the reference's result there is a rounding artefact of -1e9 + x).  A real token in
    // such a row raises the sticky flag; trailing and interior pads (some real key j <= i for every real query i) do not -- the
    // attention kernels mask those keys exactly as the reference's look-ahead mask does (utils.py:58-83).  An all-pad row has no
    // real query and stays silent.
    if (flag && i < total && !p && tok[i - (i % L)] == pad) atomicOr(flag, 1u);
}
extern "C" int mgx_pad_bitmap(const int32_t* tok, uint32_t* bits, uint32_t* flag, int B, int L, int pad, void* stream) {
    MGX_REQUIRE(tok && bits, MGX_ERR_NULL, "mgx_pad_bitmap: NULL pointer");
    MGX_REQUIRE(B > 0 && L > 0 && L % 32 == 0, MGX_ERR_SHAPE, "mgx_pad_bitmap: need L%%32==0 (got L=%d)", L);
    const int total = B * L;
    hipLaunchKernelGGL(pad_bitmap_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, tok, bits, flag,
                       total, L, pad);
    MGX_CHECK_LAUNCH("mgx_pad_bitmap");
    return MGX_OK;
}

// =================================================================================================
// K6  out = LN(dropout(x) + res)      one wave per row, row held in registers (d <= 2048)
// =================================================================================================
constexpr int LN_MAXC = 4;   // chunks of 512 columns per wave
// streaming hints (bit 0: nontemporal loads, bit 1: nontemporal stores), per kernel; round 4, tools/ln_bench.py at 131,072 x 512: the forward
// 68.6 us plain, 65.3 loads, 63.6 stores, 65.3 both; the backward 134.6 plain, 123.9 loads, 130 stores, 130.5 both
#ifndef MGX_LNF_NT
#define MGX_LNF_NT 2
#endif
#ifndef MGX_LNB_NT
#define MGX_LNB_NT 1
#endif
#define LN_LD(M, p) (((M) & 1) ? __builtin_nontemporal_load(p) : *(p))
#define LN_ST(M, v, p) do { if ((M) & 2) __builtin_nontemporal_store(v, p); else *(p) = (v); } while (0)
#ifndef MGX_LNF_GRID
#define MGX_LNF_GRID 1048576  // cap on the workgroups of the forward (4 waves = 4 rows each): uncapped, one row per wave -- with the cap of 4096 used until
                              // round 4 a wave looped over 8 rows at batch 64: 78 against 70 us (tools/ln_bench.py)
#endif
#ifndef MGX_LNB_BLOCKS
#define MGX_LNB_BLOCKS 512    // workgroups of the backward (= rows of its column-partial buffer): 512-768 are level (132-134 us at batch 64), 1024 140, 256 154
#endif

template <int NC>
__global__ __launch_bounds__(256) void add_ln_fwd_kernel(
    const uint16_t* __restrict__ x, const uint16_t* __restrict__ res, const float* __restrict__ gamma,
    const float* __restrict__ beta, uint16_t* __restrict__ out, float* __restrict__ mean_o,
    float* __restrict__ rstd_o, int rows, int d, float eps, DropCfg dc) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const int gpr = d >> 3;
    const float inv_d = 1.f / (float)d;
    for (int r = wave; r < rows; r += nwave) {
        float z[NC][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 512 + lane * 8;
            if (col < d) {
                float a[8], b[8];
                unpack8(LN_LD(MGX_LNF_NT, (const u32x4*)(x + (size_t)r * d + col)), a);
                unpack8(LN_LD(MGX_LNF_NT, (const u32x4*)(res + (size_t)r * d + col)), b);
                if (dc.thr16) {
                    float m[8];
                    drop_mult8(dc, (uint32_t)((size_t)r * gpr + (col >> 3)), m);
#pragma unroll
                    for (int k = 0; k < 8; ++k) a[k] *= m[k];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) { z[c][k] = a[k] + b[k]; s += z[c][k]; }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) z[c][k] = 0.f;
            }
        }
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 512 + lane * 8;
            if (col < d) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { const float t = z[c][k] - mean; q += t * t; }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
        if (lane == 0) { mean_o[r] = mean; rstd_o[r] = rstd; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 512 + lane * 8;
            if (col < d) {
                const f32x4* gp = (const f32x4*)(gamma + col);
                const f32x4* bp = (const f32x4*)(beta + col);
                f32x4 g0 = gp[0], g1 = gp[1], b0 = bp[0], b1 = bp[1];
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float o[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) o[k] = (z[c][k] - mean) * rstd * gg[k] + bb[k];
                LN_ST(MGX_LNF_NT, pack8(o), (u32x4*)(out + (size_t)r * d + col));
            }
        }
    }
}

// backward.  z recomputed from x,res; dz = rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma.
// Column reductions (dgamma, dbeta and -- optionally -- the column sums of dx, i.e. the bias gradient of
// the projection that produced x): per-wave register partials over the rows the wave owns -> LDS -> ONE
// plain store per column per block into a [gridDim.x][3*d] partial buffer; ln_finish_kernel then adds the
// partials to the gradients (no same-address atomics: 512 blocks x 1,024 atomics on the same 4 KB ran
// 14x below the atomic rate and dominated this kernel).
template <int NC>
__global__ __launch_bounds__(256) void add_ln_bwd_kernel(
    const uint16_t* __restrict__ dout, const uint16_t* __restrict__ x, const uint16_t* __restrict__ res,
    const float* __restrict__ gamma, const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
    uint16_t* __restrict__ dx, uint16_t* __restrict__ dres, float* __restrict__ partial, int rows, int d,
    DropCfg dc, int want_dxsum) {
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const int gpr = d >> 3;
    const float inv_d = 1.f / (float)d;
    float ag[NC][8], ab[NC][8], ax[NC][8], gg[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * 512 + lane * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            ag[c][k] = 0.f; ab[c][k] = 0.f; ax[c][k] = 0.f;
            gg[c][k] = (col < d) ? gamma[col + k] : 0.f;
        }
    }
    // The rows of a wave are a dependent chain of (load 3 x 16 B per lane, two wave reductions, store): with one row in
    // flight per wave the kernel ran at half the HBM rate.  The next row's operands are requested before the current row
    // is reduced (clamped index, unconditional: a load under a branch would be waited for at the join).
    u32x4 ra[NC], rb[NC], rd[NC];
    float mean = 0.f, rstd = 0.f;
    auto fetch = [&](int r, u32x4 (&a)[NC], u32x4 (&b)[NC], u32x4 (&dy)[NC], float& mn, float& rs) {
        const int rc = min(r, rows - 1);
        mn = mean_i[rc]; rs = rstd_i[rc];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = min(c * 512 + lane * 8, d - 8);
            a[c] = LN_LD(MGX_LNB_NT, (const u32x4*)(x + (size_t)rc * d + col));
            b[c] = LN_LD(MGX_LNB_NT, (const u32x4*)(res + (size_t)rc * d + col));
            dy[c] = LN_LD(MGX_LNB_NT, (const u32x4*)(dout + (size_t)rc * d + col));
        }
    };
    if (wave < rows) fetch(wave, ra, rb, rd, mean, rstd);
    for (int r = wave; r < rows; r += nwave) {
        u32x4 na[NC], nb[NC], nd[NC];
        float nmean, nrstd;
        fetch(r + nwave, na, nb, nd, nmean, nrstd);
        float xh[NC][8], g[NC][8], mult[NC][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 512 + lane * 8;
            if (col < d) {
                float a[8], b[8], dy[8];
                unpack8(ra[c], a);
                unpack8(rb[c], b);
                unpack8(rd[c], dy);
                if (dc.thr16) drop_mult8(dc, (uint32_t)((size_t)r * gpr + (col >> 3)), mult[c]);
                else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) mult[c][k] = 1.f;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float z = a[k] * mult[c][k] + b[k];
                    xh[c][k] = (z - mean) * rstd;
                    g[c][k] = dy[k] * gg[c][k];
                    s1 += g[c][k];
                    s2 += g[c][k] * xh[c][k];
                    ag[c][k] += dy[k] * xh[c][k];
                    ab[c][k] += dy[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) { xh[c][k] = 0.f; g[c][k] = 0.f; mult[c][k] = 0.f; }
            }
        }
        const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * 512 + lane * 8;
            if (col < d) {
                float dz[8], dxa[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    dz[k] = rstd * (g[c][k] - c1 - xh[c][k] * c2);
                    dxa[k] = dz[k] * mult[c][k];
                }
                const u32x4 pz = pack8(dz), px = pack8(dxa);
                LN_ST(MGX_LNB_NT, pz, (u32x4*)(dres + (size_t)r * d + col));
                if (dc.thr16 || dx != dres) LN_ST(MGX_LNB_NT, px, (u32x4*)(dx + (size_t)r * d + col));
                if (want_dxsum) {       // sum what the consumer will read: the bf16-rounded dx
                    float q[8];
                    unpack8(px, q);
#pragma unroll
                    for (int k = 0; k < 8; ++k) ax[c][k] += q[k];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) { ra[c] = na[c]; rb[c] = nb[c]; rd[c] = nd[c]; }
        mean = nmean; rstd = nrstd;
    }
    // block reduce of the column partials over the 4 waves, one plain store per column
    __shared__ float red[4][512];
    float* prow = partial + (size_t)blockIdx.x * 3 * d;
    const int npass = want_dxsum ? 3 : 2;
    for (int pass = 0; pass < npass; ++pass) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) red[wid][lane * 8 + k] = pass == 0 ? ag[c][k] : (pass == 1 ? ab[c][k] : ax[c][k]);
            __syncthreads();
            for (int j = threadIdx.x; j < 512; j += 256) {
                const int col = c * 512 + j;
                if (col < d) prow[pass * d + col] = red[0][j] + red[1][j] + red[2][j] + red[3][j];
            }
        }
    }
}

// out[k*d + col] += sum_b partial[b][k*d + col], k = 0 (dgamma), 1 (dbeta), 2 (dxsum)
// block = 16 columns x 16 block-groups; every thread keeps 8 independent loads in flight.
__global__ __launch_bounds__(256) void ln_finish_kernel(const float* __restrict__ partial, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ dxsum, int nblk,
                                                        int d) {
    const int cl = threadIdx.x & 15, part = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + cl;                       // column within [0, 3d)
    const int ncol = dxsum ? 3 * d : 2 * d;
    __shared__ float red[16][17];
    float s = 0.f;
    if (idx < ncol) {
        int bb = part;
        for (; bb + 7 * 16 < nblk; bb += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(bb + 16 * u) * 3 * d + idx];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; bb < nblk; bb += 16) s += partial[(size_t)bb * 3 * d + idx];
    }
    red[part][cl] = s;
    __syncthreads();
    if (threadIdx.x < 16 && idx < ncol) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][threadIdx.x];
        float* dst = idx < d ? dgamma + idx : (idx < 2 * d ? dbeta + (idx - d) : dxsum + (idx - 2 * d));
        *dst += t;
    }
}

template <int NC>
static void launch_ln_fwd(const uint16_t* x, const uint16_t* res, const float* gamma, const float* beta,
                          uint16_t* out, float* mean, float* rstd, int rows, int d, float eps, DropCfg dc,
                          hipStream_t s) {
    int grid = (rows + 3) / 4;
    if (grid > MGX_LNF_GRID) grid = MGX_LNF_GRID;
    hipLaunchKernelGGL(add_ln_fwd_kernel<NC>, dim3(grid), dim3(256), 0, s, x, res, gamma, beta, out, mean, rstd,
                       rows, d, eps, dc);
}
constexpr int LN_BWD_BLOCKS = MGX_LNB_BLOCKS;

template <int NC>
static void launch_ln_bwd(const uint16_t* dout, const uint16_t* x, const uint16_t* res, const float* gamma,
                          const float* mean, const float* rstd, uint16_t* dx, uint16_t* dres, float* partial,
                          int grid, int rows, int d, DropCfg dc, int want_dxsum, hipStream_t s) {
    hipLaunchKernelGGL(add_ln_bwd_kernel<NC>, dim3(grid), dim3(256), 0, s, dout, x, res, gamma, mean, rstd, dx,
                       dres, partial, rows, d, dc, want_dxsum);
}

extern "C" int mgx_add_ln_fwd(const uint16_t* x, const uint16_t* res, const float* gamma, const float* beta,
                              uint16_t* out, float* mean, float* rstd, int rows, int d, float eps, float p_drop,
                              uint64_t seed, void* stream) {
    MGX_REQUIRE(x && res && gamma && beta && out && mean && rstd, MGX_ERR_NULL, "mgx_add_ln_fwd: NULL pointer");
    MGX_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= 512 * LN_MAXC, MGX_ERR_SHAPE,
                "mgx_add_ln_fwd: need d%%8==0 and d<=%d (got rows=%d d=%d)", 512 * LN_MAXC, rows, d);
    const DropCfg dc = make_drop(p_drop, seed);
    const int nc = (d + 511) / 512;
    hipStream_t s = (hipStream_t)stream;
    switch (nc) {
        case 1: launch_ln_fwd<1>(x, res, gamma, beta, out, mean, rstd, rows, d, eps, dc, s); break;
        case 2: launch_ln_fwd<2>(x, res, gamma, beta, out, mean, rstd, rows, d, eps, dc, s); break;
        case 3: launch_ln_fwd<3>(x, res, gamma, beta, out, mean, rstd, rows, d, eps, dc, s); break;
        default: launch_ln_fwd<4>(x, res, gamma, beta, out, mean, rstd, rows, d, eps, dc, s); break;
    }
    MGX_CHECK_LAUNCH("mgx_add_ln_fwd");
    return MGX_OK;
}

extern "C" size_t mgx_add_ln_bwd_workspace(int rows, int d) {
    (void)rows;
    return d > 0 ? (size_t)LN_BWD_BLOCKS * 3 * d * sizeof(float) : 0;
}

extern "C" int mgx_add_ln_bwd(const uint16_t* dout, const uint16_t* x, const uint16_t* res, const float* gamma,
                              const float* mean, const float* rstd, uint16_t* dx, uint16_t* dres, float* dgamma,
                              float* dbeta, float* dxsum, void* workspace, size_t ws_bytes, int rows, int d,
                              float p_drop, uint64_t seed, void* stream) {
    MGX_REQUIRE(dout && x && res && gamma && mean && rstd && dx && dres && dgamma && dbeta && workspace, MGX_ERR_NULL,
                "mgx_add_ln_bwd: NULL pointer");
    MGX_REQUIRE(rows > 0 && d > 0 && d % 8 == 0 && d <= 512 * LN_MAXC, MGX_ERR_SHAPE,
                "mgx_add_ln_bwd: need d%%8==0 and d<=%d (got rows=%d d=%d)", 512 * LN_MAXC, rows, d);
    MGX_REQUIRE(p_drop <= 0.f || dx != dres, MGX_ERR_SHAPE, "mgx_add_ln_bwd: dx may alias dres only when p_drop==0");
    MGX_REQUIRE(ws_bytes >= mgx_add_ln_bwd_workspace(rows, d), MGX_ERR_SHAPE,
                "mgx_add_ln_bwd: workspace must hold mgx_add_ln_bwd_workspace() = %zu bytes (got %zu)",
                mgx_add_ln_bwd_workspace(rows, d), ws_bytes);
    const DropCfg dc = make_drop(p_drop, seed);
    const int nc = (d + 511) / 512;
    hipStream_t s = (hipStream_t)stream;
    int grid = (rows + 3) / 4;
    if (grid > LN_BWD_BLOCKS) grid = LN_BWD_BLOCKS;
    float* partial = (float*)workspace;
    const int wd = dxsum ? 1 : 0;
    switch (nc) {
        case 1: launch_ln_bwd<1>(dout, x, res, gamma, mean, rstd, dx, dres, partial, grid, rows, d, dc, wd, s); break;
        case 2: launch_ln_bwd<2>(dout, x, res, gamma, mean, rstd, dx, dres, partial, grid, rows, d, dc, wd, s); break;
        case 3: launch_ln_bwd<3>(dout, x, res, gamma, mean, rstd, dx, dres, partial, grid, rows, d, dc, wd, s); break;
        default: launch_ln_bwd<4>(dout, x, res, gamma, mean, rstd, dx, dres, partial, grid, rows, d, dc, wd, s); break;
    }
    hipLaunchKernelGGL(ln_finish_kernel, dim3((3 * d + 15) / 16), dim3(256), 0, s, partial, dgamma, dbeta, dxsum, grid, d);
    MGX_CHECK_LAUNCH("mgx_add_ln_bwd");
    return MGX_OK;
}

// =================================================================================================
// K9+K10  label-smoothed CE + accuracy + argmax, one wave per row     criterion.py:51-67
//   loss_r = lse - (1-eps) x_t - (eps/V) sum_v x_v      (closed form of -sum q' log softmax)
// =================================================================================================
// NC > 0: rows of ld % 8 == 0 elements, 16-byte aligned, V <= 512 NC (the model's logits: vocabulary rows padded to the GEMM tile):
//   lane l holds columns 512 c + 8 l .. + 7 of chunk c in registers -- one 16-byte load per chunk instead of eight 2-byte ones, and
//   the row is read ONCE (the scalar path, NC = 0, reads it for the maximum and again for the exponentials).
#ifndef MGX_CE_GRID
#define MGX_CE_GRID 512
#endif
// Blocks of 16 waves: every block ends with four atomics on the SAME 16 bytes of `stats`, which the L2 serialises at ~12 ns each --
// with 2,048 blocks of 4 waves those 8,192 atomics, not the 100 MB of logits, were the kernel's time (62 us at cfg2 / batch 64).
template <int NC>
__global__ __launch_bounds__(1024) void smooth_ce_fwd_kernel(
    const uint16_t* __restrict__ logits, const int32_t* __restrict__ target, float* __restrict__ stats,
    int32_t* __restrict__ argmax_o, float* __restrict__ row_lse, int rows, int V, int ld, float eps_ls, int pad,
    long long* __restrict__ det /* deterministic mode: the four sums as fixed-point integers, folded into stats afterwards */) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    float loss_acc = 0.f, cnt_acc = 0.f, hit_acc = 0.f, row_acc = 0.f;
    for (int r = wave; r < rows; r += nwave) {
        const uint16_t* lp = logits + (size_t)r * ld;
        const int t = target[r];                               // requested with the row, not after its reductions
        float mx = -INFINITY, sx = 0.f, xt_part = 0.f;
        int am = 0x7fffffff;
        float xr[NC > 0 ? NC : 1][8];
        if (NC > 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int col = 512 * c + 8 * lane;
                if (col < V) unpack8(*(const u32x4*)(lp + col), xr[c]);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (col + k < V) {                         // ascending columns: the first maximum of a lane is its lowest index
                        sx += xr[c][k];
                        if (xr[c][k] > mx) { mx = xr[c][k]; am = col + k; }
                        if (col + k == t) xt_part = xr[c][k];  // the target's logit sits in some lane's registers: no second, dependent
                                                               // global load of lp[t] per row (it was a third of this kernel's time)
                    } else {
                        xr[c][k] = -INFINITY;                  // exp(-inf - mx) = 0 below
                    }
                }
            }
        } else {
            for (int v = lane; v < V; v += 64) {
                const float xv = bf16_to_f32(lp[v]);
                sx += xv;
                if (xv > mx) { mx = xv; am = v; }
            }
        }
        // wave arg-max: max value, lowest index among ties (torch.argmax picks the first)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float omx = __shfl_xor(mx, o, 64);
            const int oam = __shfl_xor(am, o, 64);
            if (omx > mx || (omx == mx && oam < am)) { mx = omx; am = oam; }
        }
        float se = 0.f;
        if (NC > 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int k = 0; k < 8; ++k) se += __expf(xr[c][k] - mx);
        } else {
            for (int v = lane; v < V; v += 64) se += __expf(bf16_to_f32(lp[v]) - mx);
        }
        se = wave_sum(se);
        sx = wave_sum(sx);
        const float lse = mx + __logf(se);
        const float xt_vec = NC > 0 ? wave_sum(xt_part) : 0.f;
        if (lane == 0) {
            const float xt = NC > 0 ? xt_vec : ((t >= 0 && t < V) ? bf16_to_f32(lp[t]) : 0.f);
            row_lse[r] = lse;
            argmax_o[r] = am;
            if (t != pad) {
                loss_acc += lse - (1.f - eps_ls) * xt - (eps_ls / (float)V) * sx;
                cnt_acc += 1.f;
            }
            hit_acc += (am == t) ? 1.f : 0.f;
            row_acc += 1.f;
        }
    }
    __shared__ float red[16][4];
    if (lane == 0) { red[wid][0] = loss_acc; red[wid][1] = cnt_acc; red[wid][2] = hit_acc; red[wid][3] = row_acc; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float v = 0.f;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) v += red[i][threadIdx.x];
        if (det) det_add(det + threadIdx.x, v);
        else atomicAdd(stats + threadIdx.x, v);
    }
}

template <bool VEC>     // VEC: ld % 8 == 0, 16-byte aligned rows: 16-byte loads and stores (8 columns per lane and pass)
__global__ __launch_bounds__(256) void smooth_ce_bwd_kernel(
    const uint16_t* __restrict__ logits, const int32_t* __restrict__ target, const float* __restrict__ stats,
    const float* __restrict__ row_lse, uint16_t* __restrict__ dlogits, int rows, int V, int ld, float eps_ls,
    int pad, float gscale, const float* __restrict__ gscale_dev) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwave = (gridDim.x * blockDim.x) >> 6;
    const float cnt = stats[1];
    const float sc = (gscale_dev ? gscale * gscale_dev[0] : gscale) / cnt;
    const float u = eps_ls / (float)V;
    for (int r = wave; r < rows; r += nwave) {
        const uint16_t* lp = logits + (size_t)r * ld;
        uint16_t* dp = dlogits + (size_t)r * ld;
        const int t = target[r];
        const float lse = row_lse[r];
        const bool keep = (t != pad);
        if (VEC) {
            for (int col = 8 * lane; col < ld; col += 512) {
                float x[8], gv[8];
                if (keep && col < V) unpack8(*(const u32x4*)(lp + col), x);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int v = col + k;
                    gv[k] = (keep && v < V) ? sc * (__expf(x[k] - lse) - u - ((v == t) ? (1.f - eps_ls) : 0.f)) : 0.f;
                }
                *(u32x4*)(dp + col) = pack8(gv);
            }
            continue;
        }
        for (int v = lane; v < ld; v += 64) {
            float gval = 0.f;
            if (keep && v < V) {
                const float p = __expf(bf16_to_f32(lp[v]) - lse);
                gval = sc * (p - u - ((v == t) ? (1.f - eps_ls) : 0.f));
            }
            dp[v] = f32_to_bf16(gval);
        }
    }
}

extern "C" int mgx_smooth_ce_fwd(const uint16_t* logits, const int32_t* target, float* stats, int32_t* argmax,
                                 float* row_lse, int rows, int V, int ld, float eps_ls, int pad, void* stream) {
    MGX_REQUIRE(logits && target && stats && argmax && row_lse, MGX_ERR_NULL, "mgx_smooth_ce_fwd: NULL pointer");
    MGX_REQUIRE(rows > 0 && V > 0 && ld >= V, MGX_ERR_SHAPE, "mgx_smooth_ce_fwd: need ld>=V (rows=%d V=%d ld=%d)",
                rows, V, ld);
    int grid = (rows + 15) / 16;                             // 16 waves = 16 rows in flight per block
    if (grid > MGX_CE_GRID) grid = MGX_CE_GRID;
    int rc;
    long long* det = mgx_det_scratch(4, stream, &rc);
    if (rc != MGX_OK) return rc;
    const bool vec = ld % 8 == 0 && ((uintptr_t)logits & 15) == 0 && V <= 2048;
    const int nc = vec ? (V + 511) / 512 : 0;
#define MGX_CE_FWD(NC) hipLaunchKernelGGL(smooth_ce_fwd_kernel<NC>, dim3(grid), dim3(1024), 0, (hipStream_t)stream, logits, target, stats, \
                                          argmax, row_lse, rows, V, ld, eps_ls, pad, det)
    switch (nc) {
        case 1: MGX_CE_FWD(1); break;
        case 2: MGX_CE_FWD(2); break;
        case 3: MGX_CE_FWD(3); break;
        case 4: MGX_CE_FWD(4); break;
        default: MGX_CE_FWD(0); break;
    }
#undef MGX_CE_FWD
    if (det) launch_det_fold(det, stats, 4, 1.f, 1, (hipStream_t)stream);
    MGX_CHECK_LAUNCH("mgx_smooth_ce_fwd");
    return MGX_OK;
}
extern "C" int mgx_smooth_ce_bwd(const uint16_t* logits, const int32_t* target, const float* stats,
                                 const float* row_lse, uint16_t* dlogits, int rows, int V, int ld, float eps_ls,
                                 int pad, float gscale, const float* gscale_dev, void* stream) {
    MGX_REQUIRE(logits && target && stats && row_lse && dlogits, MGX_ERR_NULL, "mgx_smooth_ce_bwd: NULL pointer");
    MGX_REQUIRE(rows > 0 && V > 0 && ld >= V, MGX_ERR_SHAPE, "mgx_smooth_ce_bwd: need ld>=V (rows=%d V=%d ld=%d)",
                rows, V, ld);
    int grid = (rows + 3) / 4;
    if (grid > 2048) grid = 2048;
    if (ld % 8 == 0 && (((uintptr_t)logits | (uintptr_t)dlogits) & 15) == 0)
        hipLaunchKernelGGL(smooth_ce_bwd_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, stats,
                           row_lse, dlogits, rows, V, ld, eps_ls, pad, gscale, gscale_dev);
    else
        hipLaunchKernelGGL(smooth_ce_bwd_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, logits, target, stats,
                           row_lse, dlogits, rows, V, ld, eps_ls, pad, gscale, gscale_dev);
    MGX_CHECK_LAUNCH("mgx_smooth_ce_bwd");
    return MGX_OK;
}

// =================================================================================================
// K11  Adam on one flat buffer + bf16 shadow                                train.py:143
// =================================================================================================
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   uint16_t* __restrict__ shadow, size_t n, float lr, float b1,
                                                   float b2, float eps, float bc1, float bc2s, float gscale) {
    // torch.optim.Adam: p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
    const size_t n4 = n >> 2;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pp = ((f32x4*)p)[i], gg = ((const f32x4*)g)[i], mm = ((f32x4*)m)[i], vv = ((f32x4*)v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = gg[k] * gscale;
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            pp[k] -= (lr / bc1) * mm[k] / (sqrtf(vv[k]) / bc2s + eps);
        }
        ((f32x4*)p)[i] = pp; ((f32x4*)m)[i] = mm; ((f32x4*)v)[i] = vv;
        if (shadow) {
            u32x2 w = {pack_bf16x2(pp.x, pp.y), pack_bf16x2(pp.z, pp.w)};
            ((u32x2*)shadow)[i] = w;
        }
    }
    // tail (n % 4)
    const size_t t0 = n4 << 2;
    const size_t i = t0 + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float gk = g[i] * gscale;
        const float mk = b1 * m[i] + (1.f - b1) * gk, vk = b2 * v[i] + (1.f - b2) * gk * gk;
        m[i] = mk; v[i] = vk;
        const float pk = p[i] - (lr / bc1) * mk / (sqrtf(vk) / bc2s + eps);
        p[i] = pk;
        if (shadow) shadow[i] = f32_to_bf16(pk);
    }
}
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ p, uint16_t* __restrict__ s, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s[i] = f32_to_bf16(p[i]);
}

extern "C" int mgx_adam_step(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t n, float lr,
                             float beta1, float beta2, float eps, int step, float gscale, void* stream) {
    MGX_REQUIRE(p && g && m && v, MGX_ERR_NULL, "mgx_adam_step: NULL pointer");
    MGX_REQUIRE(n > 0 && step >= 1, MGX_ERR_SHAPE, "mgx_adam_step: need n>0 and step>=1 (step=%d)", step);
    MGX_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0 &&
                    ((uintptr_t)shadow & 7) == 0,
                MGX_ERR_SHAPE, "mgx_adam_step: buffers must be 16-byte aligned");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, shadow, n,
                       lr, beta1, beta2, eps, bc1, bc2s, gscale);
    MGX_CHECK_LAUNCH("mgx_adam_step");
    return MGX_OK;
}
extern "C" int mgx_cast_bf16(const float* p, uint16_t* shadow, size_t n, void* stream) {
    MGX_REQUIRE(p && shadow, MGX_ERR_NULL, "mgx_cast_bf16: NULL pointer");
    size_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, shadow, n);
    MGX_CHECK_LAUNCH("mgx_cast_bf16");
    return MGX_OK;
}
