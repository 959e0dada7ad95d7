// Fused relative global attention, forward (replaces layers.py:86-106 + 111-133 of the reference).
//
// Work decomposition: workgroup = 4 waves = 128 consecutive query rows of one (batch, head); each wave
// owns 32 query rows and keeps O^T (64 x 32 fp32), the running max m and the running sum l in
// registers.  The workgroup sweeps key tiles of 32 keys from j0 = 0 up to its own diagonal; K, V tiles
// and the E chunks are staged once per workgroup in LDS (shared by the 4 waves), next tile
// prefetched into registers while the current one is computed (one barrier per step).
//
// Per step and wave (all MFMA 32x32x16 bf16, fp32 accumulate):
//   QE   = Q_tile . Er_chunk^T            4 MFMA   -> LDS band (skew buffer, see rel_attn_common.hpp)
//   S^T  = K_tile . Q_tile^T + Srel^T     4 MFMA   (Srel^T read from the band as the C operand)
//   P^T  = exp2(S^T*log2e - m)            in registers: keys on registers, queries on lanes
//   O^T += V_tile^T . P^T                 4 MFMA   (P^T accumulators are the B operand directly;
//                                                   V^T fragments by ds_read_b64_tr_b16)
// Q is pre-scaled by 1/8 (exact in bf16), so S is already logit = (qk + srel)/sqrt(64).
//
// Algorithmic FLOPs per (b,h): 3 products x 2*64 x L(L+1)/2 (causal half) -- DESIGN.md.
#include "rel_attn_common.hpp"

using namespace relattn;

namespace {
constexpr int WAVES = 4;
constexpr int OFF_K = 0;                               // 2 x 4 KiB   image R
constexpr int OFF_V = OFF_K + 2 * TILE_BYTES;          // 2 x 4 KiB   image T
constexpr int OFF_BAND = OFF_V + 2 * TILE_BYTES;       // 4 x (32 rows x 272 B) fp32 rotated band
constexpr int LDS_BYTES = OFF_BAND + WAVES * BAND_BYTES;    // 51,200 B -> 3 workgroups per CU
}  // namespace

// WRITE_W = false: the training/inference forward (ctx + lse).
// WRITE_W = true : debug/eval output of the reference (layers.py:102,109): the same sweep recomputes S and
//                  writes weights[b,h,i,j] = exp(S - lse_i) (fp32, caller pre-zeroes the future triangle).
template <bool WRITE_W>
__global__ __launch_bounds__(256, 3) void rel_attn_fwd_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ Er /* = E + (M-L)*64 */,
    const uint32_t* __restrict__ padbits, uint16_t* __restrict__ ctx, float* __restrict__ lse_out,
    const float* __restrict__ lse_in, float* __restrict__ weights, int L, int d) {
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
    const int nsteps = min(Q0 + 4, nchunk);
    const size_t ld = (size_t)3 * d;                     // qkv row stride (elements)
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;

    // ---- staging roles: thread -> (row, 16-byte chunk) of a 32x64 tile -------------------------
    const int srow = tid >> 3, sch = tid & 7;
    const int st_offR = imgR_off(srow, sch), st_offT = imgT_off(srow, sch);
    const uint16_t* kg = qkv_b + (size_t)srow * ld + d + hd * 64 + sch * 8;       // + j0*ld
    const uint16_t* vg = kg + d;
    // Er row fragment ks of chunk q (B operand of Q.Er^T: column t = lane&31, delta = 32q + t), from L2
    auto e_frag = [&](int q, int ks) {
        return __builtin_bit_cast(bf16x8, *(const u32x4*)(Er + (size_t)(L - 1 - 32 * q - a) * 64 + hh * 8 + ks * 16));
    };

    // ---- prologue: K/V tile 0 -------------------------------------------------------------------
    *(u32x4*)(smem + OFF_K + st_offR) = *(const u32x4*)kg;
    *(u32x4*)(smem + OFF_V + st_offT) = *(const u32x4*)vg;
    // Q fragments (A operand of QE, B operand of S^T), pre-scaled by 1/8; first "hi" E chunk
    bf16x8 qf[4], ecur[4];
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
            ecur[ks] = e_frag(Q0 + w, ks);
        }
    }
    __syncthreads();

    char* band = smem + OFF_BAND + w * BAND_BYTES;
    // band addressing (rel_attn_common.hpp): register r writes row slot r of region hh at the
    // precomputed column offset wcl[r] (XOR bit 7 for odd chunks); a lane reads its own row with
    // four ds_read_b128 at rbase + 32*g4 (+128 when D/32 is odd)
    int wcl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) wcl[r] = hh * BAND_REGION + (((crow(r, hh) - a) & 63) << 2);
    const int rbase = band_rowoff(a) + 16 * hh;
    // the wave's first "hi" chunk (delta in [i0, i0+31])
    if (wave_on) {
        const int q = Q0 + w;
        f32x16 qe = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], ecur[ks], qe);
        const int tog = (q & 1) << 7;
#pragma unroll
        for (int r = 0; r < 16; ++r) *(float*)(band + r * BAND_STRIDE + (wcl[r] ^ tog)) = qe[r];
        if (q >= 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ecur[ks] = e_frag(q - 1, ks);      // new chunk of step 0
        }
    }

    f32x16 o0 = zero16(), o1 = zero16();
    float m_run = -INFINITY, l_run = 0.f;
    float lse2w = 0.f;
    if (WRITE_W && wave_on) lse2w = lse_in[((size_t)b * heads + hd) * L + i0 + a] * LOG2E;

    for (int s = 0; s < nsteps; ++s) {
        const int cur = s & 1;
        // ---- prefetch next step's tiles into registers ------------------------------------------
        u32x4 kreg, vreg;
        const bool have_next = (s + 1 < nsteps);
        if (have_next) {
            kreg = *(const u32x4*)(kg + (size_t)(s + 1) * 32 * ld);
            vreg = *(const u32x4*)(vg + (size_t)(s + 1) * 32 * ld);
        }

        const int dq = Q0 + w - s;                       // D/32 for this wave; active iff dq >= 0
        if (wave_on && dq >= 0) {
            // ---- new chunk dq-1 (delta in [D-32, D-1]); nothing to do on the diagonal -----------
            if (dq >= 1) {
                f32x16 qe = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], ecur[ks], qe);
                const int tog = ((dq - 1) & 1) << 7;
#pragma unroll
                for (int r = 0; r < 16; ++r) *(float*)(band + r * BAND_STRIDE + (wcl[r] ^ tog)) = qe[r];
            }
            if (dq >= 2) {                                // chunk of the next step: latency hidden by this step
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) ecur[ks] = e_frag(dq - 2, ks);
            }
            wave_lds_fence();
            // ---- S^T = K Q^T + Srel^T ------------------------------------------------------------
            f32x16 c;
            {
                const char* rb = band + rbase + ((dq & 1) << 7);
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const f32x4 v = *(const f32x4*)(rb + 32 * g4);
                    c[4 * g4] = v.x; c[4 * g4 + 1] = v.y; c[4 * g4 + 2] = v.z; c[4 * g4 + 3] = v.w;
                }
            }
            const char* kt = smem + OFF_K + cur * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = mfma(frag_R(kt, a, hh, ks), qf[ks], c);
            // ---- masks ---------------------------------------------------------------------------
            if (dq == 0) {                                // diagonal tile: key b > query a is the future
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (crow(r, hh) > a) ? -INFINITY : c[r];
            }
            if (padbits) {
                const uint32_t pw = padbits[(size_t)b * nchunk + s];   // wave-uniform
                if (pw) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        c[r] = ((pw >> crow(r, hh)) & 1u) ? ((c[r] == -INFINITY) ? c[r] : PAD_NEG) : c[r];
                }
            }
            if (WRITE_W) {
                // weights[b,h,i0+a, j0 + 8*g4 + 4*hh + k] = exp(S - lse); masked entries are exactly 0
                float* wrow = weights + (((size_t)b * heads + hd) * L + i0 + a) * L + 32 * s + 4 * hh;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    f32x4 v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float sv = c[4 * g4 + k];
                        v[k] = (sv <= PAD_NEG) ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(sv, LOG2E, -lse2w));
                    }
                    *(f32x4*)(wrow + 8 * g4) = v;
                }
            } else {
            // ---- online softmax (keys on registers + lane half, queries on lanes) ----------------
            float tmax = c[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, c[r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax);       // finite: every visited tile has key j0 <= i
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
            const float mneg = -m_new * LOG2E;
            float lsum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(c[r], LOG2E, mneg));
                lsum += c[r];
            }
            l_run = l_run * alpha + lsum;
            m_run = m_new;
            if (!__all(alpha == 1.f)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            }
            // ---- O^T += V^T P^T -------------------------------------------------------------------
            const char* vt = smem + OFF_V + cur * TILE_BYTES;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const bf16x8 pf = acc_to_frag(c, ss);
                o0 = mfma(frag_T(vt, lane, ss, 0), pf, o0);
                o1 = mfma(frag_T(vt, lane, ss, 1), pf, o1);
            }
            }   // !WRITE_W
        }
        // ---- publish the prefetched tiles into the other buffers ---------------------------------
        if (have_next) {
            *(u32x4*)(smem + OFF_K + (cur ^ 1) * TILE_BYTES + st_offR) = kreg;
            *(u32x4*)(smem + OFF_V + (cur ^ 1) * TILE_BYTES + st_offT) = vreg;
        }
        __syncthreads();
    }

    // ---- epilogue: ctx[b, i0+a, hd*64 + c] = O^T[c][a] / l ; lse = m + ln l ---------------------
    if (!WRITE_W && wave_on) {
        const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
        const float inv = 1.f / l_tot;
        store_rows_lds(ctx + ((size_t)b * L + i0) * d + hd * 64, (size_t)d, o0, o1, lane, inv, band);
        if (hh == 0) lse_out[((size_t)b * heads + hd) * L + i0 + a] = m_run + __logf(l_tot);
    }
}

static void set_fwd_attrs() {
    static bool attr_set = false;
    if (attr_set) return;
    hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
}

extern "C" int mgx_rel_attn_fwd(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits, uint16_t* ctx,
                                float* lse, int B, int L, int d, int M, void* stream) {
    MGX_REQUIRE(qkv && E && ctx && lse, MGX_ERR_NULL, "mgx_rel_attn_fwd: NULL pointer");
    MGX_REQUIRE(B > 0 && L > 0 && d > 0 && d % 64 == 0 && L % 32 == 0 && M >= L, MGX_ERR_SHAPE,
                "mgx_rel_attn_fwd: need d%%64==0, L%%32==0, M>=L (got B=%d L=%d d=%d M=%d)", B, L, d, M);
    MGX_REQUIRE((L + 127) / 128 <= 65535, MGX_ERR_SHAPE, "mgx_rel_attn_fwd: L too large");
    set_fwd_attrs();
    dim3 grid(B * (d / 64), (L + 127) / 128);
    hipLaunchKernelGGL(rel_attn_fwd_kernel<false>, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, qkv,
                       E + (size_t)(M - L) * 64, padbits, ctx, lse, (const float*)nullptr, (float*)nullptr, L, d);
    MGX_CHECK_LAUNCH("mgx_rel_attn_fwd");
    return MGX_OK;
}

extern "C" int mgx_rel_attn_weights(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits, const float* lse,
                                    float* weights, int B, int L, int d, int M, void* stream) {
    MGX_REQUIRE(qkv && E && lse && weights, MGX_ERR_NULL, "mgx_rel_attn_weights: NULL pointer");
    MGX_REQUIRE(B > 0 && L > 0 && d > 0 && d % 64 == 0 && L % 32 == 0 && M >= L, MGX_ERR_SHAPE,
                "mgx_rel_attn_weights: need d%%64==0, L%%32==0, M>=L (got B=%d L=%d d=%d M=%d)", B, L, d, M);
    set_fwd_attrs();
    dim3 grid(B * (d / 64), (L + 127) / 128);
    hipLaunchKernelGGL(rel_attn_fwd_kernel<true>, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, qkv,
                       E + (size_t)(M - L) * 64, padbits, (uint16_t*)nullptr, (float*)nullptr, lse, weights, L, d);
    MGX_CHECK_LAUNCH("mgx_rel_attn_weights");
    return MGX_OK;
}
