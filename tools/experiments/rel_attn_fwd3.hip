// Fused relative global attention, forward -- "ping-pong" kernel for L % 256 == 0 (cfg2, cfg4, cfg5 prefill).
// (replaces layers.py:86-106 + 111-133 of the reference; rel_attn_fwd.hip keeps every other shape and the weights output)
//
// What the measurements of this round say about the CU (profiles/README.md, round 3):
//   * one wave issues at most one instruction per ~4 cycles, whatever the unit: a wave ALONE on its SIMD cannot keep the
//     matrix pipe (32 cycles per MFMA) busy when the tile needs ~7-10 other instructions per MFMA (rel_attn_fwd2.hip);
//   * several waves that all run the same serial chain (MFMA -> LDS -> MFMA -> exp -> MFMA) overlap poorly: the SIMD's time
//     is close to the SUM of its MFMA and VALU time (rel_attn_fwd.hip, 34 % MFMA-busy at 2.6 waves per SIMD).
// So the two waves of a SIMD are put in COMPLEMENTARY phases by construction.  A workgroup is 8 waves = 256 query rows
// (wave w owns rows 32(Q0+w)..+31); every wave's step is split in two segments separated by workgroup barriers:
//
//   M segment (matrix pipe only, operands already in registers):
//        QE   = Q~ . Er_chunk^T   (chunk of step s+1)        4 MFMA
//        S^T  = K(s) . Q~^T + Srel^T                         4 MFMA   (band values preloaded as the C operand)
//        O^T += V(s-1)^T . P(s-1)^T                          4 MFMA
//   V segment (VALU / LDS / memory only):
//        exp2 / row sums / bf16 packing of S(s); lazy-softmax check;  band stores of QE, band loads for step s+1;
//        K fragments of tile s+1, V^T fragments of tile s; staging of tile s+2 into the LDS ring, global loads of tile s+3.
//
// Waves 0-3 (one per SIMD) start with an M segment, waves 4-7 one barrier later: while a wave runs its 12 MFMAs its SIMD
// partner runs its ~95 VALU / LDS instructions.  A segment contains no dependency between the two pipes, so the compiler's
// schedule inside it does not matter much; the overlap is produced by the hardware arbitration between the two waves.
// K / V tiles live in a ring of 4 LDS buffers (a tile is read during three half-steps by the two wave groups).
#include <type_traits>
#include "rel_attn_common.hpp"

using namespace relattn;

namespace f3 {
constexpr int WAVES = 8;
constexpr int NBUF = 8;
constexpr int OFF_K = 0;                                        // 4 x 4 KiB   image R
constexpr int OFF_V = OFF_K + NBUF * TILE_BYTES;                // 4 x 4 KiB   image T
constexpr int OFF_BAND = OFF_V + NBUF * TILE_BYTES;             // 8 x (32 rows x 272 B) fp32 rotated bands
constexpr int OFF_PAD = OFF_BAND + WAVES * BAND_BYTES;          // key-padding words of this batch row (first 256)
constexpr int OFF_FLAG = OFF_PAD + 1024;
constexpr int LDS_BYTES = OFF_FLAG + 16;                        // 136,208 B -> 1 workgroup (8 waves) per CU
constexpr float M_INIT = -1.0e37f;
constexpr float L_SAFE = 1.0e24f;
}  // namespace f3

// RIGID: two barriers per step (every M segment faces a V segment of the SIMD partner and nothing else).
// !RIGID: ONE barrier per step -- waves 0-3 pass it after their V segment, waves 4-7 between their M and V segments, so the
//         partners still run the two segments in opposite order but flow freely inside a barrier interval.
template <bool RIGID>
__global__ __launch_bounds__(512, 2) void rel_attn_fwd_pp_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ Ef, const uint32_t* __restrict__ padbits,
    uint16_t* __restrict__ ctx, float* __restrict__ lse_out, int L, int d, int bgroup) {
    using namespace f3;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int late = w >> 2;                             // waves 4-7 run half a step behind waves 0-3
    const int a = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    const int nqb = L >> 8;
    const int b = (blockIdx.y / nqb) * bgroup + blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qb = nqb - 1 - (blockIdx.y % nqb);         // heaviest (latest) query blocks first
    const int Q0 = qb * 8;                               // first 32-row chunk of the workgroup (even)
    const int q0 = Q0 + w;                               // the wave's diagonal tile
    const int nchunk = L >> 5;
    const int ntile = Q0 + 8;                            // key tiles this workgroup visits
    const int nstep = ntile + 1;                         // + one step whose M segment holds the last O^T += V^T P^T
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;

    // staging roles: threads 0..255 stage the K tile (image R), threads 256..511 the V tile (image T)
    const int srow = (tid & 255) >> 3, sch = tid & 7;
    const int st_off = (tid < 256) ? OFF_K + imgR_off(srow, sch) : OFF_V + imgT_off(srow, sch);
    const uint16_t* sg = qkv_b + (size_t)srow * ld + d + ((tid < 256) ? 0 : d) + hd * 64 + sch * 8;     // + 32*tile*ld
    const size_t tile_stride = (size_t)32 * ld;
    auto ef = [&](int q, int ks) { return __builtin_bit_cast(bf16x8, Ef[(size_t)(max(q, 0) * 4 + ks) * 64 + lane]); };

    // ---- prologue: tiles 0, 1, 2 into the ring, key-padding words --------------------------------------------------------
    *(u32x4*)(smem + st_off) = *(const u32x4*)sg;
    *(u32x4*)(smem + st_off + TILE_BYTES) = *(const u32x4*)(sg + tile_stride);
    *(u32x4*)(smem + st_off + 2 * TILE_BYTES) = *(const u32x4*)(sg + 2 * tile_stride);          // ntile >= 8
    int anypad = 0;
    if (padbits) {
        if (tid == 0) *(volatile uint32_t*)(smem + OFF_FLAG) = 0u;
        __syncthreads();
        uint32_t acc = 0;
#pragma unroll 1
        for (int t = tid; t < ntile; t += 512) {
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
        if (kt >= 256) v = padbits[(size_t)b * nchunk + min(kt, nchunk - 1)];
        return __builtin_amdgcn_readfirstlane(v);
    };
    bf16x8 qf[4];
    {
        const uint16_t* qp = qkv_b + (size_t)(q0 * 32 + a) * ld + hd * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = __builtin_bit_cast(bf16x8, scale8(*(const u32x4*)(qp + ks * 16), 0.125f));
    }
    // staging registers: tile 3 now, tile t+4 during step t.  Loads are clamped to the last tile (never used beyond it).
    u32x4 sreg = *(const u32x4*)(sg + (size_t)3 * tile_stride);
    __syncthreads();

    // band addressing (rel_attn_common.hpp).  PHYSICAL chunk parity = (chunk - q0) & 1 (each wave has its own band, so the
    // assignment is free): the chunk stored in step s then has parity s & 1 for every wave -- a compile-time constant in the
    // two-step loop body -- and the tile read for step s+1 has parity (s+1) & 1.
    const int band_base = OFF_BAND + w * BAND_BYTES;
    uint32_t wc[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        wc[0][r] = lds_addr_of(smem) + band_base + hh * BAND_REGION + (((crow(r, hh) - a) & 63) << 2);
        wc[1][r] = wc[0][r] ^ 128u;
    }
    const int rbase = band_base + band_rowoff(a) + 16 * hh;
    auto band_put = [&](const f32x16& v, int par) {
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_store_f32(wc[par][r] + r * BAND_STRIDE, v[r]);
    };
    auto band_get = [&](int par) {
        const char* rb = smem + rbase + (par << 7);
        f32x16 c;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *(const f32x4*)(rb + 32 * g4);
            c[4 * g4] = v.x; c[4 * g4 + 1] = v.y; c[4 * g4 + 2] = v.z; c[4 * g4 + 3] = v.w;
        }
        return c;
    };
    auto qe_prod = [&](const bf16x8 (&e)[4]) {
        f32x16 c = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma(qf[ks], e[ks], c);
        return c;
    };

    f32x16 o0 = zero16(), o1 = zero16();
    float m_ref = M_INIT, l_run = 0.f;
    auto exp_tile = [&](const f32x16& c, float mneg, bf16x8 (&pf)[2]) {
        float lsum = 0.f;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            u32x4 wv;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(c[8 * ss + 2 * jj], LOG2E, mneg));
                const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(c[8 * ss + 2 * jj + 1], LOG2E, mneg));
                lsum += p0;
                lsum += p1;
                wv[jj] = pack_bf16x2(p0, p1);
            }
            pf[ss] = __builtin_bit_cast(bf16x8, wv);
        }
        return lsum;
    };

    // ---- band prologue: "old" chunk of step 0 (chunk q0, parity 0) and its "new" chunk (q0-1, parity 1); C operand and K
    //      fragments of step 0; Er fragments for the M segments of steps 0 and 1 ----------------------------------------------
    bf16x8 e[2][4], kf[4], vfr[2][2], pf[2];
    f32x16 cS;
    {
        bf16x8 c0[4], c1[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            c0[ks] = ef(q0, ks); c1[ks] = ef(q0 - 1, ks);
            e[0][ks] = ef(q0 - 2, ks);                   // chunk of step 1, multiplied in M segment 0
            e[1][ks] = ef(q0 - 3, ks);                   // chunk of step 2, multiplied in M segment 1
        }
        band_put(qe_prod(c0), 0);
        band_put(qe_prod(c1), 1);
        wave_lds_fence();
        cS = band_get(0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_R(smem + OFF_K, a, hh, ks);
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) { pf[ss] = bf16x8{}; vfr[ss][0] = bf16x8{}; vfr[ss][1] = bf16x8{}; }    // P(-1) = 0
    }
    const int am = a - 4 * hh;                           // key crow(r,hh) is in the future of query a  <=>  crow(r,0) > am
    const int last = q0 + 1;                             // the wave's last step (M segment only: O^T += V(q0)^T P(q0)^T)

    // one step: M segment, barrier, V segment, barrier.  PAR = s & 1.  A wave beyond its last step only stages and syncs.
    auto step = [&](int s, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;
        f32x16 qe;
        if (s <= last) {
            // ---- M segment ----
            qe = qe_prod(e[PAR]);                         // chunk q0-s-2 (the "new" chunk of step s+1)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) cS = mfma(kf[ks], qf[ks], cS);
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                o0 = mfma(vfr[ss][0], pf[ss], o0);
                o1 = mfma(vfr[ss][1], pf[ss], o1);
            }
        }
        if (RIGID || late) __syncthreads();
        // ---- V segment ----
        // ring: tile s+3 -> buffer (s+3) & 7 (complete two barriers before its first read in V segment s+2, whichever group
        // wrote which half; it replaces tile s-5); registers <- tile s+4
        *(u32x4*)(smem + st_off + ((s + 3) & 7) * TILE_BYTES) = sreg;
        sreg = *(const u32x4*)(sg + (size_t)min(s + 4, ntile - 1) * tile_stride);
        if (s <= last) {
            const int dq = q0 - s;                        // >= -1
            if (dq <= 0 || anypad) {
                // key crow(r,hh) of tile s is in the future of query a  <=>  crow(r,0) > am + 32 dq  (dq = -1: every key);
                // padded keys get the reference's additive -1e9 (future keys stay -inf)
                const uint32_t pwl = padword(s) >> (4 * hh);
                const int thr = am + 32 * dq;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (pwl & (1u << crow(r, 0))) ? fminf(cS[r], PAD_NEG) : cS[r];
                    cS[r] = (crow(r, 0) > thr) ? -INFINITY : v;
                }
            }
            float lsum = exp_tile(cS, -m_ref * LOG2E, pf);
            if (__builtin_expect(__any(!(lsum <= L_SAFE)), 0)) {
                // redo against the true maximum and rescale O, l once (first tile; a score jumping by > 55 nats)
                float tmax = cS[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, cS[r]);
                tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
                const float m_new = fmaxf(m_ref, tmax);
                const float alpha = __builtin_amdgcn_exp2f((m_ref - m_new) * LOG2E);
                lsum = exp_tile(cS, -m_new * LOG2E, pf);
                l_run *= alpha;
#pragma unroll
                for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
                m_ref = m_new;
            }
            l_run += lsum;
            band_put(qe, PAR);                            // chunk q0-s-2
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) e[PAR][ks] = ef(q0 - s - 4, ks);      // multiplied in M segment s+2
            cS = band_get(PAR ^ 1);                       // Srel^T of tile s+1
            const char* kt = smem + OFF_K + ((s + 1) & 7) * TILE_BYTES;
            const char* vt = smem + OFF_V + (s & 7) * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_R(kt, a, hh, ks);
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) { vfr[ss][0] = frag_T(vt, lane, ss, 0); vfr[ss][1] = frag_T(vt, lane, ss, 1); }
        }
        if (RIGID || !late) __syncthreads();
    };

    if (RIGID && late) __syncthreads();                   // half a step behind: this group's M segments face the other's V segments
    for (int s = 0; s < nstep; s += 2) {                  // nstep = Q0 + 9 is odd: the second call of the last trip is s = nstep
        step(s, std::integral_constant<int, 0>{});
        if (s + 1 < nstep) step(s + 1, std::integral_constant<int, 1>{});
    }
    if (RIGID && !late) __syncthreads();                  // same number of barriers for both groups

    // ---- epilogue: ctx[b, i0+a, hd*64 + c] = O^T[c][a] / l ; lse = m + ln l ---------------------
    const int i0 = q0 * 32;
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.f / l_tot;
    store_rows_lds(ctx + ((size_t)b * L + i0) * d + hd * 64, (size_t)d, o0, o1, lane, inv, smem + band_base);
    if (hh == 0) lse_out[((size_t)b * heads + hd) * L + i0 + a] = m_ref + __logf(l_tot);
}

static int fwdpp_batch_group(int B, int L, int d) {
    const double per_row = (double)L * d * 2 * 4;
    int g = B;
    while (g > 1 && (g * per_row > 110e6 || B % g != 0)) --g;
    return g;
}

// launched by mgx_rel_attn_fwd (rel_attn_fwd.hip) when L % 256 == 0; the workspace already holds the fragment-ordered Er
int relattn::fwdpp_launch(const uint16_t* qkv, const void* EfA, const uint32_t* padbits, uint16_t* ctx, float* lse, int B,
                          int L, int d, void* stream) {
    static const hipError_t attr0 = hipFuncSetAttribute((const void*)rel_attn_fwd_pp_kernel<true>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, f3::LDS_BYTES);
    static const hipError_t attr1 = hipFuncSetAttribute((const void*)rel_attn_fwd_pp_kernel<false>,
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, f3::LDS_BYTES);
    (void)attr0; (void)attr1;
    const int bg = fwdpp_batch_group(B, L, d);
    MGX_REQUIRE((long)(L / 256) * (B / bg) <= 65535, MGX_ERR_SHAPE, "mgx_rel_attn_fwd: L/256 * batch groups too large");
    dim3 grid(bg * (d / 64), (L / 256) * (B / bg));
    if (env_is_one("MGX_ATTN_PP_RIGID"))
        hipLaunchKernelGGL(rel_attn_fwd_pp_kernel<true>, grid, dim3(512), f3::LDS_BYTES, (hipStream_t)stream, qkv,
                           (const u32x4*)EfA, padbits, ctx, lse, L, d, bg);
    else
        hipLaunchKernelGGL(rel_attn_fwd_pp_kernel<false>, grid, dim3(512), f3::LDS_BYTES, (hipStream_t)stream, qkv,
                           (const u32x4*)EfA, padbits, ctx, lse, L, d, bg);
    MGX_CHECK_LAUNCH("mgx_rel_attn_fwd(ping-pong)");
    return MGX_OK;
}
