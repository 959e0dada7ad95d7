// Fused relative global attention, forward (replaces layers.py:86-106 + 111-133 of the reference).
//
// Work decomposition: workgroup = 4 waves = 128 consecutive query rows of one (batch, head); each wave
// owns 32 query rows and keeps O^T (64 x 32 fp32), a softmax reference m and the running sum l in
// registers.  The workgroup sweeps key tiles from j0 = 0 up to its own diagonal, NSUB 32-key tiles per
// step; K and V tiles are staged once per workgroup in LDS (shared by the 4 waves), the next step's
// tiles prefetched into registers while the current ones are computed (one barrier per step).
//
// Per 32-key tile and wave (all MFMA 32x32x16 bf16, fp32 accumulate):
//   QE   = Q_tile . Er_chunk^T            4 MFMA   -> LDS band (skew buffer, see rel_attn_common.hpp)
//   S^T  = K_tile . Q_tile^T + Srel^T     4 MFMA   (Srel^T read from the band as the C operand)
//   P^T  = exp2(S^T*log2e - m)            in registers: keys on registers, queries on lanes
//   O^T += V_tile^T . P^T                 4 MFMA   (P^T accumulators are the B operand directly;
//                                                   V^T fragments by ds_read_b64_tr_b16)
// Q is pre-scaled by 1/8 (exact in bf16), so S is already logit = (qk + srel)/sqrt(64).
//
// * The Er chunk operands come from a FRAGMENT-ORDERED copy of E (er_frag_kernel, rel_attn_common.hpp):
//   one wave load instruction reads 1 KB contiguous.  (Read from E's natural [delta][64] layout the same
//   instruction touched 32 B of 32 different rows; a build without those loads ran 19 % faster.)
// * Lazy softmax reference: m is NOT the running maximum.  A tile is exponentiated against the current m
//   straight away (no max chain, no cross-half shuffle, no rescale of O); only when a lane's partial sum
//   shows that some exponent left the safe range (first tile; a score that jumps by > 40 nats) the tile
//   is redone against the true maximum and O, l are rescaled once.  exp2(S - m) with m <= true maximum is
//   exact to fp32 / bf16 relative precision however far m lags, so the result is the same softmax.
// * Steps whose tiles are all full (no diagonal, no padded key) run a branch-free body in which the
//   NSUB tiles' MFMA chains, band traffic and exponentials are independent instruction streams.
//
// Algorithmic FLOPs per (b,h): 3 products x 2*64 x L(L+1)/2 (causal half) -- DESIGN.md.
#include "rel_attn_common.hpp"

using namespace relattn;

namespace {
constexpr int WAVES = 4;
template <int NSUB>
struct FwdCfg {
    static constexpr int KT = NSUB * TILE_BYTES;                    // one step's K (or V) tiles
    static constexpr int OFF_K = 0;                                 // 2 x KT  image R
    static constexpr int OFF_V = 2 * KT;                            // 2 x KT  image T
    static constexpr int OFF_BAND = 4 * KT;                         // 4 x (32 rows x 272 B) fp32 rotated band
    static constexpr int OFF_PAD = OFF_BAND + WAVES * BAND_BYTES;   // key-padding words of this batch row (<= 256)
    static constexpr int LDS_BYTES = OFF_PAD + 1024;
};
constexpr float M_INIT = -1.0e37f;      // "no reference yet": finite, so exp2(-inf - m) is 0 and not NaN
constexpr float L_SAFE = 1.0e24f;       // a lane's partial sum of one step above this => redo against the true max
}  // namespace

// WRITE_W = false: the training/inference forward (ctx + lse).
// WRITE_W = true : debug/eval output of the reference (layers.py:102,109): the same sweep recomputes S and
//                  writes weights[b,h,i,j] = exp(S - lse_i) (fp32, caller pre-zeroes the future triangle).
template <int NSUB, bool WRITE_W>
__global__ __launch_bounds__(256, NSUB == 1 ? 3 : 2) void rel_attn_fwd_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ Ef /* fragment-ordered Er, see er_frag_kernel */,
    const uint32_t* __restrict__ padbits, uint16_t* __restrict__ ctx, float* __restrict__ lse_out,
    const float* __restrict__ lse_in, float* __restrict__ weights, int L, int d) {
    using C = FwdCfg<NSUB>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    // grid: x = (batch, head) [fast], y = query-block rank [slow], heaviest (latest) blocks first, so the
    // whole grid is dispatched longest-job-first; blocks of one (b,h) share an XCD when B*h % 8 == 0.
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qb = gridDim.y - 1 - blockIdx.y;
    const int I0 = qb * 128, Q0 = I0 >> 5;
    const int i0 = I0 + w * 32;
    const int nchunk = L >> 5;                           // number of 32-row chunks / key tiles
    const bool wave_on = i0 < L;
    const int ntw = min(Q0 + 4, nchunk);                 // key tiles this workgroup visits
    const int nsteps = (ntw + NSUB - 1) / NSUB;
    const size_t ld = (size_t)3 * d;                     // qkv row stride (elements)
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;

    // ---- staging roles: thread -> (row, 16-byte chunk) of a 32x64 tile -------------------------
    const int srow = tid >> 3, sch = tid & 7;
    const int st_offR = imgR_off(srow, sch), st_offT = imgT_off(srow, sch);
    const uint16_t* kg = qkv_b + (size_t)srow * ld + d + hd * 64 + sch * 8;       // + 32*tile*ld
    const uint16_t* vg = kg + d;
    // every load below is unconditional with a clamped index (a load inside a branch makes the compiler drain
    // the whole VMEM queue where the branch rejoins); data of clamped tiles / chunks is never used
    auto tile_off = [&](int kt) { return (size_t)min(kt, nchunk - 1) * 32 * ld; };
    auto ef = [&](int q, int ks) { return __builtin_bit_cast(bf16x8, Ef[(size_t)(max(q, 0) * 4 + ks) * 64 + lane]); };

    // ---- prologue: K/V tiles of step 0, key-padding words ------------------------------------------
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        *(u32x4*)(smem + C::OFF_K + u * TILE_BYTES + st_offR) = *(const u32x4*)(kg + tile_off(u));
        *(u32x4*)(smem + C::OFF_V + u * TILE_BYTES + st_offT) = *(const u32x4*)(vg + tile_off(u));
    }
    int anypad = 0;
    if (padbits) {
        uint32_t acc = 0;
#pragma unroll 1
        for (int t = tid; t < ntw; t += 256) {
            const uint32_t pwv = padbits[(size_t)b * nchunk + t];
            if (t < 256) *(uint32_t*)(smem + C::OFF_PAD + 4 * t) = pwv;
            acc |= pwv;
        }
        anypad = __syncthreads_or(acc != 0);
    }
    const uint32_t* padrow = padbits + (size_t)b * nchunk;
    auto padword = [&](int kt) -> uint32_t {             // wave-uniform
        if (!anypad || kt >= ntw) return 0u;
        const uint32_t v = (kt < 256) ? *(const uint32_t*)(smem + C::OFF_PAD + 4 * kt) : padrow[kt];
        return __builtin_amdgcn_readfirstlane(v);
    };
    // Q fragments (A operand of QE, B operand of S^T), pre-scaled by 1/8
    bf16x8 qf[4], e[NSUB][4];
    const int q0 = Q0 + w;                               // the wave's diagonal tile / first "hi" chunk
    if (wave_on) {
        const uint16_t* qp = qkv_b + (size_t)(i0 + a) * ld + hd * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 raw = *(const u32x4*)(qp + ks * 16);
            float f[8];
            unpack8(raw, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] *= 0.125f;
            qf[ks] = __builtin_bit_cast(bf16x8, pack8(f));
            e[0][ks] = ef(q0, ks);
        }
    }
    __syncthreads();

    // band addressing (rel_attn_common.hpp): register r writes row slot r of region hh at the precomputed
    // absolute LDS offset wcl[r] (XOR bit 7 for odd chunks: OFF_BAND and BAND_BYTES are multiples of 256, so
    // the low 8 bits are the column byte offset); a lane reads its own row with four ds_read_b128 at
    // rbase + 32*g4 (+128 when D/32 is odd)
    const int band_base = C::OFF_BAND + w * BAND_BYTES;
    int wcl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wcl[r] = band_base + hh * BAND_REGION + (((crow(r, hh) - a) & 63) << 2);
    const int rbase = band_base + band_rowoff(a) + 16 * hh;
    auto band_put = [&](const f32x16& v, int q) {        // chunk q of Q.Er^T -> band
        const int tog = (q & 1) << 7;
#pragma unroll
        for (int r = 0; r < 16; ++r) *(float*)(smem + r * BAND_STRIDE + (wcl[r] ^ tog)) = v[r];
    };
    auto band_get = [&](int dq) {                        // Srel^T of the tile with D/32 = dq
        const char* rb = smem + rbase + ((dq & 1) << 7);
        f32x16 c;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *(const f32x4*)(rb + 32 * g4);
            c[4 * g4] = v.x; c[4 * g4 + 1] = v.y; c[4 * g4 + 2] = v.z; c[4 * g4 + 3] = v.w;
        }
        return c;
    };
    if (wave_on) {
        f32x16 qe = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], e[0][ks], qe);
        band_put(qe, q0);
#pragma unroll
        for (int u = 0; u < NSUB; ++u)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) e[u][ks] = ef(q0 - u - 1, ks);      // new chunks of step 0
    }

    f32x16 o0 = zero16(), o1 = zero16();
    float m_ref = M_INIT, l_run = 0.f;
    float lse2w = 0.f;
    if (WRITE_W && wave_on) lse2w = lse_in[((size_t)b * heads + hd) * L + i0 + a] * LOG2E;

    // ---- softmax of the step's tiles against the lazy reference, then O^T += V^T P^T ---------------
    // P^T goes straight into the bf16 operand fragments of the O^T product (k order kappa, see acc_to_frag)
    auto exp_tiles = [&](const f32x16 (&c)[NSUB], const bool (&act)[NSUB], float mneg, bf16x8 (&pf)[NSUB][2]) {
        float lsum = 0.f;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            if (!act[u]) continue;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                u32x4 wv;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(c[u][8 * ss + 2 * jj], LOG2E, mneg));
                    const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(c[u][8 * ss + 2 * jj + 1], LOG2E, mneg));
                    lsum += p0;
                    lsum += p1;
                    wv[jj] = pack_bf16x2(p0, p1);
                }
                pf[u][ss] = __builtin_bit_cast(bf16x8, wv);
            }
        }
        return lsum;
    };
    auto softmax_pv = [&](const f32x16 (&c)[NSUB], const bool (&act)[NSUB], int cur) {
        bf16x8 pf[NSUB][2];
        float lsum = exp_tiles(c, act, -m_ref * LOG2E, pf);
        if (__builtin_expect(__any(!(lsum <= L_SAFE)), 0)) {
            // redo against the true maximum (both lane halves of a query row must agree on m)
            float tmax = -INFINITY;
#pragma unroll
            for (int u = 0; u < NSUB; ++u) {
                if (!act[u]) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, c[u][r]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_ref, tmax);       // finite: every visited tile has a key j <= i (or PAD_NEG)
            const float alpha = __builtin_amdgcn_exp2f((m_ref - m_new) * LOG2E);
            lsum = exp_tiles(c, act, -m_new * LOG2E, pf);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            m_ref = m_new;
        }
        l_run += lsum;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            if (!act[u]) continue;
            const char* vt = smem + C::OFF_V + cur * C::KT + u * TILE_BYTES;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                o0 = mfma(frag_T(vt, lane, ss, 0), pf[u][ss], o0);
                o1 = mfma(frag_T(vt, lane, ss, 1), pf[u][ss], o1);
            }
        }
    };
    const int am = a - 4 * hh;                           // key crow(r,hh) is in the future of query a  <=>  crow(r,0) > am

    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        // ---- prefetch the next step's tiles into registers -------------------------------------------
        u32x4 kreg[NSUB], vreg[NSUB];
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            kreg[u] = *(const u32x4*)(kg + tile_off(NSUB * (s + 1) + u));
            vreg[u] = *(const u32x4*)(vg + tile_off(NSUB * (s + 1) + u));
        }
        const int dq0 = q0 - NSUB * s;                   // D/32 of the step's first tile for this wave
        if (wave_on && dq0 >= 0) {
            uint32_t pw[NSUB];
            bool full = !WRITE_W && (dq0 - (NSUB - 1) >= 1);
#pragma unroll
            for (int u = 0; u < NSUB; ++u) {
                pw[u] = padword(NSUB * s + u);
                full = full && (pw[u] == 0);
            }
            const char* kt = smem + C::OFF_K + cur * C::KT;
            if (full) {
                // ---- branch-free body: every tile of the step is full (below the diagonal, no padded key) --
                f32x16 c[NSUB];
#pragma unroll
                for (int u = 0; u < NSUB; ++u) {
                    c[u] = zero16();
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) c[u] = mfma(qf[ks], e[u][ks], c[u]);
                }
#pragma unroll
                for (int u = 0; u < NSUB; ++u) {          // LDS operations of a wave execute in order
                    band_put(c[u], dq0 - u - 1);
                    wave_lds_fence();
                    c[u] = band_get(dq0 - u);
                    wave_lds_fence();
                }
#pragma unroll
                for (int u = 0; u < NSUB; ++u)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) e[u][ks] = ef(dq0 - u - 1 - NSUB, ks);    // chunks of the next step
#pragma unroll
                for (int u = 0; u < NSUB; ++u)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) c[u] = mfma(frag_R(kt + u * TILE_BYTES, a, hh, ks), qf[ks], c[u]);
                bool all_on[NSUB];
#pragma unroll
                for (int u = 0; u < NSUB; ++u) all_on[u] = true;
                softmax_pv(c, all_on, cur);
            } else {
                // ---- general body: diagonal tile, padded keys, tiles beyond the diagonal, weights output ----
                f32x16 c[NSUB];
                bool act[NSUB];
#pragma unroll
                for (int u = 0; u < NSUB; ++u) {
                    const int dq = dq0 - u;
                    act[u] = dq >= 0;
                    c[u] = zero16();
                    if (dq >= 1) {
                        f32x16 qe = zero16();
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], e[u][ks], qe);
                        band_put(qe, dq - 1);
                    }
                    wave_lds_fence();
                    if (act[u]) {
                        c[u] = band_get(dq);
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) c[u] = mfma(frag_R(kt + u * TILE_BYTES, a, hh, ks), qf[ks], c[u]);
                        if (dq == 0) {                    // diagonal tile: key b > query a is the future
#pragma unroll
                            for (int r = 0; r < 16; ++r) c[u][r] = (crow(r, 0) > am) ? -INFINITY : c[u][r];
                        }
                        if (pw[u]) {                      // padded key: the reference's additive -1e9 (future keys stay -inf)
                            const uint32_t pwl = pw[u] >> (4 * hh);
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                c[u][r] = (pwl & (1u << crow(r, 0))) ? fminf(c[u][r], PAD_NEG) : c[u][r];
                        }
                    }
                    wave_lds_fence();
                }
#pragma unroll
                for (int u = 0; u < NSUB; ++u)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) e[u][ks] = ef(dq0 - u - 1 - NSUB, ks);
                if (WRITE_W) {
                    // weights[b,h,i0+a, j0 + 8*g4 + 4*hh + k] = exp(S - lse); masked entries are exactly 0
#pragma unroll
                    for (int u = 0; u < NSUB; ++u) {
                        if (!act[u]) continue;
                        float* wrow = weights + (((size_t)b * heads + hd) * L + i0 + a) * L + 32 * (NSUB * s + u) + 4 * hh;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            f32x4 v;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const float sv = c[u][4 * g4 + k];
                                v[k] = (sv <= PAD_NEG) ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(sv, LOG2E, -lse2w));
                            }
                            *(f32x4*)(wrow + 8 * g4) = v;
                        }
                    }
                } else {
                    softmax_pv(c, act, cur);
                }
            }
        }
        // ---- publish the prefetched tiles into the other buffers ---------------------------------
        if (s + 1 < nsteps) {
#pragma unroll
            for (int u = 0; u < NSUB; ++u) {
                *(u32x4*)(smem + C::OFF_K + (cur ^ 1) * C::KT + u * TILE_BYTES + st_offR) = kreg[u];
                *(u32x4*)(smem + C::OFF_V + (cur ^ 1) * C::KT + u * TILE_BYTES + st_offT) = vreg[u];
            }
        }
        __syncthreads();
    }

    // ---- epilogue: ctx[b, i0+a, hd*64 + c] = O^T[c][a] / l ; lse = m + ln l ---------------------
    if (!WRITE_W && wave_on) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.f / l_tot;
        store_rows_lds(ctx + ((size_t)b * L + i0) * d + hd * 64, (size_t)d, o0, o1, lane, inv, smem + band_base);
        if (hh == 0) lse_out[((size_t)b * heads + hd) * L + i0 + a] = m_ref + __logf(l_tot);
    }
}

// NSUB: 32-key tiles per step.  2 => 64-key steps, 67.6 KB LDS, 2 workgroups per CU with up to 256 VGPRs (two
// independent tile streams per wave); 1 => 32-key steps, 52.2 KB, 3 workgroups per CU.
#ifndef MGX_FWD_NSUB
#define MGX_FWD_NSUB 1
#endif

static void set_fwd_attrs() {
    static bool attr_set = false;
    if (attr_set) return;
    hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<MGX_FWD_NSUB, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                        FwdCfg<MGX_FWD_NSUB>::LDS_BYTES);
    hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, FwdCfg<1>::LDS_BYTES);
    attr_set = true;
}

extern "C" size_t mgx_rel_attn_fwd_workspace(int L) { return L > 0 ? er_frag_bytes(L) : 0; }

static int fwd_common_checks(const char* who, const void* ws, size_t ws_bytes, int B, int L, int d, int M) {
    MGX_REQUIRE(B > 0 && L > 0 && d > 0 && d % 64 == 0 && L % 32 == 0 && M >= L, MGX_ERR_SHAPE,
                "%s: need d%%64==0, L%%32==0, M>=L (got B=%d L=%d d=%d M=%d)", who, B, L, d, M);
    MGX_REQUIRE((L + 127) / 128 <= 65535, MGX_ERR_SHAPE, "%s: L too large", who);
    MGX_REQUIRE(ws && ws_bytes >= er_frag_bytes(L) && ((uintptr_t)ws & 255) == 0, MGX_ERR_SHAPE,
                "%s: workspace must be 256-byte aligned and >= mgx_rel_attn_fwd_workspace(L) = %zu bytes (got %zu)", who,
                er_frag_bytes(L), ws_bytes);
    return MGX_OK;
}

extern "C" int mgx_rel_attn_fwd(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits, uint16_t* ctx,
                                float* lse, void* workspace, size_t ws_bytes, int B, int L, int d, int M, void* stream) {
    MGX_REQUIRE(qkv && E && ctx && lse, MGX_ERR_NULL, "mgx_rel_attn_fwd: NULL pointer");
    if (int rc = fwd_common_checks("mgx_rel_attn_fwd", workspace, ws_bytes, B, L, d, M)) return rc;
    set_fwd_attrs();
    launch_er_frag(E + (size_t)(M - L) * 64, (u32x4*)workspace, nullptr, L, (hipStream_t)stream);
    dim3 grid(B * (d / 64), (L + 127) / 128);
    hipLaunchKernelGGL((rel_attn_fwd_kernel<MGX_FWD_NSUB, false>), grid, dim3(256), FwdCfg<MGX_FWD_NSUB>::LDS_BYTES,
                       (hipStream_t)stream, qkv, (const u32x4*)workspace, padbits, ctx, lse, (const float*)nullptr,
                       (float*)nullptr, L, d);
    MGX_CHECK_LAUNCH("mgx_rel_attn_fwd");
    return MGX_OK;
}

extern "C" int mgx_rel_attn_weights(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits, const float* lse,
                                    float* weights, void* workspace, size_t ws_bytes, int B, int L, int d, int M,
                                    void* stream) {
    MGX_REQUIRE(qkv && E && lse && weights, MGX_ERR_NULL, "mgx_rel_attn_weights: NULL pointer");
    if (int rc = fwd_common_checks("mgx_rel_attn_weights", workspace, ws_bytes, B, L, d, M)) return rc;
    set_fwd_attrs();
    launch_er_frag(E + (size_t)(M - L) * 64, (u32x4*)workspace, nullptr, L, (hipStream_t)stream);
    dim3 grid(B * (d / 64), (L + 127) / 128);
    hipLaunchKernelGGL((rel_attn_fwd_kernel<1, true>), grid, dim3(256), FwdCfg<1>::LDS_BYTES, (hipStream_t)stream, qkv,
                       (const u32x4*)workspace, padbits, (uint16_t*)nullptr, (float*)nullptr, lse, weights, L, d);
    MGX_CHECK_LAUNCH("mgx_rel_attn_weights");
    return MGX_OK;
}
