// Fused relative global attention, forward -- the software-pipelined kernel for L % 256 == 0 (cfg2, cfg4, cfg5 prefill).
// (replaces layers.py:86-106 + 111-133 of the reference; rel_attn_fwd.hip keeps every other shape and the weights output)
//
// Why a second structure.  In rel_attn_fwd_kernel a wave's instruction stream is one serial chain per key tile
//   Q.Er^T (MFMA) -> band stores -> band loads -> K.Q^T (MFMA) -> exp/pack (48 VALU) -> V^T.P^T (MFMA)
// so a wave never has an MFMA and a VALU instruction in flight together; overlap comes only from the 3 waves that share a
// SIMD, and the profile shows that the SIMD's time is close to the SUM of its MFMA and VALU issue time (34 % MFMA-busy).
// Here a wave owns TWO adjacent 32-row query blocks (A: rows i0.., B: rows i0+32..) and ONE wave runs per SIMD with the
// whole register file, and the loop is software-pipelined so that every MFMA group has independent VALU / LDS work beside it:
//
//   iteration s (key tile s; both blocks X = A, B):
//     P1  S_X(s)  += K(s) . Q_X^T         8 MFMA   | band stores of Q_X.Er^T for step s+1, K/V staging, V^T fragment reads
//     P2  QE_X(s+2) = Q_X . Er_chunk^T    8 MFMA   | exp2 / row sums / bf16 packing of S_A(s), S_B(s)
//         one wave-uniform check (lazy softmax reference, see rel_attn_fwd.hip) ; workgroup barrier
//     P3  O_X    += V(s)^T . P_X(s)^T     8 MFMA   | band loads for step s+1 (C operand of the next S), K fragments of tile s+1
//
// * K / V / Er fragments are shared by the two blocks: half the LDS fragment reads and Er loads per MFMA.  Block B's new
//   Er chunk of a step is block A's new chunk of the step before, so two fragment sets alternate (two steps per loop trip;
//   band parities and LDS buffers are compile-time constants as well).
// * Each wave runs the pipelined loop up to ITS OWN diagonal (qA = Q0 + 2w full tiles), then the same pipelined body with
//   the masks applied for its two diagonal steps (for every step when the batch row has padded keys); waves that are done
//   keep staging K/V tiles and meeting the barrier.
// * 256 query rows per workgroup, 4 waves, 1 workgroup per CU (LDS: 16 KB of K/V double buffers + 8 fp32 bands of 8.5 KB).
#include <type_traits>
#include "rel_attn_common.hpp"

using namespace relattn;

namespace f2 {
constexpr int WAVES = 4;
constexpr int OFF_K = 0;                                        // 2 x 4 KiB   image R
constexpr int OFF_V = OFF_K + 2 * TILE_BYTES;                   // 2 x 4 KiB   image T
constexpr int OFF_BAND = OFF_V + 2 * TILE_BYTES;                // 8 x (32 rows x 272 B) fp32 rotated bands: wave w block X at 2w+X
constexpr int OFF_PAD = OFF_BAND + 2 * WAVES * BAND_BYTES;      // key-padding words of this batch row (first 256)
constexpr int OFF_FLAG = OFF_PAD + 1024;
constexpr int LDS_BYTES = OFF_FLAG + 16;                        // 87,056 B -> 1 workgroup per CU
constexpr float M_INIT = -1.0e37f;
constexpr float L_SAFE = 1.0e24f;
}  // namespace f2

// MGX_F2_STAMP (diagnostic build only, `_build.py --variant stamp -DMGX_F2_STAMP`): s_memtime stamps at the phase-group
// boundaries that exist anyway; lane 0 of each wave writes its sums over block A's lse rows (tools/fwd64_stamp.py reads them).
#ifdef MGX_F2_STAMP
#define STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define STAMP_ADD(acc, t1, t0) acc += (t1) - (t0)
#else
#define STAMP(var)
#define STAMP_ADD(acc, t1, t0)
#endif

#ifndef MGX_F2_SCHED
#define MGX_F2_SCHED 1
#endif
#if MGX_F2_SCHED
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#else
#define SGB(mask, n)
#endif

__global__ __launch_bounds__(256, 1) void rel_attn_fwd64_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ Ef, const uint32_t* __restrict__ padbits,
    uint16_t* __restrict__ ctx, float* __restrict__ lse_out, int L, int d, int bgroup) {
    using namespace f2;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    const int nqb = L >> 8;
#ifdef MGX_F2_STAMP
    unsigned long long st_p12 = 0, st_bar = 0, st_p3 = 0, st_tail = 0, st_p1 = 0;
#endif
    STAMP(st_begin);
    const int b = (blockIdx.y / nqb) * bgroup + blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qb = nqb - 1 - (blockIdx.y % nqb);         // heaviest (latest) query blocks first
    const int Q0 = qb * 8;                               // first 32-row chunk of the workgroup
    const int qA = Q0 + 2 * w;                           // block A's diagonal tile (even); block B's is qA + 1
    const int nchunk = L >> 5;
    const int ntw = Q0 + 8;                              // key tiles this workgroup visits
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;

    const int srow = tid >> 3, sch = tid & 7;
    const int st_offR = imgR_off(srow, sch), st_offT = imgT_off(srow, sch);
    const uint16_t* kg = qkv_b + (size_t)srow * ld + d + hd * 64 + sch * 8;
    const uint16_t* vg = kg + d;
    const size_t tile_stride = (size_t)32 * ld;
    auto ef = [&](int q, int ks) { return __builtin_bit_cast(bf16x8, Ef[(size_t)(max(q, 0) * 4 + ks) * 64 + lane]); };

    // ---- prologue: K/V tile 0, key-padding words ---------------------------------------------------
    *(u32x4*)(smem + OFF_K + st_offR) = *(const u32x4*)kg;
    *(u32x4*)(smem + OFF_V + st_offT) = *(const u32x4*)vg;
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
    // Q fragments of both blocks, pre-scaled by 1/8 (exact in bf16)
    bf16x8 qf[2][4];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const uint16_t* qp = qkv_b + (size_t)((qA + X) * 32 + a) * ld + hd * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float f[8];
            unpack8(*(const u32x4*)(qp + ks * 16), f);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] *= 0.125f;
            qf[X][ks] = __builtin_bit_cast(bf16x8, pack8(f));
        }
    }
    // next K/V tile in registers; koff = element offset of the tile the NEXT prefetch fetches
    u32x4 kreg = *(const u32x4*)(kg + tile_stride);      // ntw >= 8: tile 1 exists
    u32x4 vreg = *(const u32x4*)(vg + tile_stride);
    size_t koff = 2 * tile_stride;                       // ... and so does tile 2
    int tnext = 2;
    auto prefetch_next = [&]() {
        kreg = *(const u32x4*)(kg + koff);
        vreg = *(const u32x4*)(vg + koff);
        koff += (tnext + 1 < ntw) ? tile_stride : 0;
        ++tnext;
    };
    __syncthreads();

    // band addressing (rel_attn_common.hpp): block A's band of this wave; block B's is BAND_BYTES further (an immediate).
    // wc[p][r] = absolute LDS address of (band + region + column byte offset) for chunk parity p; the row slot r*272 is the
    // instruction's immediate offset.  A lane reads its own row with four ds_read_b128 at rbase + 32*g4 (+128 when D/32 is odd).
    const int band_base = OFF_BAND + 2 * w * BAND_BYTES;
    uint32_t wc[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        wc[0][r] = lds_addr_of(smem) + band_base + hh * BAND_REGION + (((crow(r, hh) - a) & 63) << 2);
        wc[1][r] = wc[0][r] ^ 128u;
    }
    const int rbase = band_base + band_rowoff(a) + 16 * hh;
    auto band_put_c = [&](const f32x16& v, int X, int par) {          // par compile-time after inlining
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_store_f32(wc[par][r] + r * BAND_STRIDE + X * BAND_BYTES, v[r]);
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

    f32x16 o[2][2] = {{zero16(), zero16()}, {zero16(), zero16()}};
    float m_ref[2] = {M_INIT, M_INIT}, l_run[2] = {0.f, 0.f};

    // exp2 of one tile against reference mneg = -m*log2e: bf16 operand fragments of O^T += V^T P^T and the lane's partial row sum
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
    // redo a tile against the true maximum and rescale O, l once (rare: first tile, or a score jumping by > 55 nats)
    auto redo_tile = [&](int X, const f32x16& c, bf16x8 (&pf)[2]) {
        float tmax = c[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, c[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_ref[X], tmax);
        const float alpha = __builtin_amdgcn_exp2f((m_ref[X] - m_new) * LOG2E);
        const float lsum = exp_tile(c, -m_new * LOG2E, pf);
        l_run[X] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[X][0][r] *= alpha; o[X][1][r] *= alpha; }
        m_ref[X] = m_new;
        return lsum;
    };

    // ---- band: the "old" and "new" chunks of step 0 for both blocks, the pending chunks of step 1, K fragments of tile 0 ----
    const int nmain = anypad ? 0 : qA;                    // even; every tile s < qA is full for both blocks of this wave
    bf16x8 e[2][4], kf[4];
    f32x16 cS[2], qe[2];
    {
        bf16x8 c3[4], c2[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { c3[ks] = ef(qA + 1, ks); c2[ks] = ef(qA, ks); }
        band_put_c(qe_prod(0, c2), 0, 0);                // OLD_A(0) = chunk qA   (parity 0)
        band_put_c(qe_prod(1, c3), 1, 1);                // OLD_B(0) = chunk qA+1 (parity 1)
        {
            bf16x8 c1[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { c1[ks] = ef(qA - 1, ks); e[0][ks] = ef(qA - 2, ks); e[1][ks] = ef(qA - 3, ks); }
            band_put_c(qe_prod(1, c2), 1, 0);            // NEW_B(0) = chunk qA
            band_put_c(qe_prod(0, c1), 0, 1);            // NEW_A(0) = chunk qA-1
            qe[1] = qe_prod(1, c1);                      // NEW_B(1) = chunk qA-1, stored in iteration 0
            qe[0] = qe_prod(0, e[0]);                    // NEW_A(1) = chunk qA-2
            wave_lds_fence();
            cS[0] = band_get(0, 0);
            cS[1] = band_get(1, 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_R(smem + OFF_K, a, hh, ks);
        }
    }

    // ---- pipelined main loop: two steps per trip ------------------------------------------------------------------------
    const int am = a - 4 * hh;                           // key crow(r,hh) is in the future of query a  <=>  crow(r,0) > am
    auto step = [&](int s, auto par_tag, auto masked_tag) {
        constexpr int PAR = decltype(par_tag)::value;     // = s & 1 : LDS buffer of tile s, chunk parities
        constexpr bool MASKED = decltype(masked_tag)::value;   // the wave's two diagonal steps (s = qA, qA + 1)
        STAMP(st_a);
        // P1: staging of tile s+1, prefetch of tile s+2, V^T fragments of tile s, S MFMAs, band stores for step s+1
        *(u32x4*)(smem + OFF_K + (PAR ^ 1) * TILE_BYTES + st_offR) = kreg;
        *(u32x4*)(smem + OFF_V + (PAR ^ 1) * TILE_BYTES + st_offT) = vreg;
        prefetch_next();
        const char* vt = smem + OFF_V + PAR * TILE_BYTES;
        bf16x8 vfr[2][2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) { vfr[ss][0] = frag_T(vt, lane, ss, 0); vfr[ss][1] = frag_T(vt, lane, ss, 1); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cS[0] = mfma(kf[ks], qf[0][ks], cS[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cS[1] = mfma(kf[ks], qf[1][ks], cS[1]);
        band_put_c(qe[0], 0, PAR);                        // NEW_A(s+1) = chunk qA-s-2
        band_put_c(qe[1], 1, PAR ^ 1);                    // NEW_B(s+1) = chunk qA-s-1
#if MGX_F2_SCHED
        // 8 MFMA, each followed by 4 band stores; the 2 staging stores + 8 transposed reads ride on the first gaps
        SGB(0x200, 2); SGB(0x100, 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(0x008, 1); SGB(0x200, 4); }
#endif
#ifdef MGX_F2_STAMP
        __builtin_amdgcn_sched_barrier(0);
        STAMP(st_a2);
        __builtin_amdgcn_sched_barrier(0);
        if (!MASKED) STAMP_ADD(st_p1, st_a2, st_a);
#endif
        if (MASKED) {
            // key crow(r,hh) of tile s is in the future of query a of block X  <=>  crow(r,0) > am + 32 (qA + X - s):
            // s = qA: block A on its diagonal, block B full;  s = qA + 1: block A beyond its diagonal (every key masked:
            // P = 0, nothing accumulates), block B on its diagonal
            // padded keys of tile s (batch rows with padding): the reference's additive -1e9 (future keys stay -inf)
            const uint32_t pwl = padword(s) >> (4 * hh);
#pragma unroll
            for (int X = 0; X < 2; ++X) {
                const int thr = am + 32 * (qA + X - s);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = (pwl & (1u << crow(r, 0))) ? fminf(cS[X][r], PAD_NEG) : cS[X][r];
                    cS[X][r] = (crow(r, 0) > thr) ? -INFINITY : v;
                }
            }
        }
        // P2: Q.Er^T of step s+2 beside the exponentials of step s
        qe[1] = qe_prod(1, e[PAR]);                       // NEW_B(s+2) = chunk qA-s-2
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[PAR][ks] = ef(qA - s - 4, ks);      // block A's chunk of the next iteration
        qe[0] = qe_prod(0, e[PAR ^ 1]);                   // NEW_A(s+2) = chunk qA-s-3
        bf16x8 pf[2][2];
        float lsum[2];
        lsum[0] = exp_tile(cS[0], -m_ref[0] * LOG2E, pf[0]);
        lsum[1] = exp_tile(cS[1], -m_ref[1] * LOG2E, pf[1]);
#if MGX_F2_SCHED
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(0x008, 1); SGB(0x002, 14); }
#endif
        if (__builtin_expect(__any(!(lsum[0] <= L_SAFE) || !(lsum[1] <= L_SAFE)), 0)) {
            lsum[0] = redo_tile(0, cS[0], pf[0]);
            lsum[1] = redo_tile(1, cS[1], pf[1]);
        }
        l_run[0] += lsum[0];
        l_run[1] += lsum[1];
        STAMP(st_b);
        __syncthreads();                                  // tile s+1 visible; every wave is done reading tile s-1's buffer
        STAMP(st_c);
        // P3: O^T += V^T P^T ; band loads and K fragments for step s+1
#pragma unroll
        for (int X = 0; X < 2; ++X)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                o[X][0] = mfma(vfr[ss][0], pf[X][ss], o[X][0]);
                o[X][1] = mfma(vfr[ss][1], pf[X][ss], o[X][1]);
            }
        cS[0] = band_get(0, PAR ^ 1);                     // dq = qA-s-1
        cS[1] = band_get(1, PAR);                         // dq = qA-s
        const char* ktn = smem + OFF_K + (PAR ^ 1) * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = frag_R(ktn, a, hh, ks);
#if MGX_F2_SCHED
        SGB(0x100, 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) { SGB(0x008, 1); SGB(0x100, 1); }
#endif
        __builtin_amdgcn_sched_barrier(0);                // one scheduling region per phase group: P3 does not mix with the next P1
        STAMP(st_d);
        if (!MASKED) { STAMP_ADD(st_p12, st_b, st_a); STAMP_ADD(st_bar, st_c, st_b); STAMP_ADD(st_p3, st_d, st_c); }
    };
    STAMP(st_loop);
    int s = 0;
    for (; s < nmain; s += 2) {
        step(s, std::integral_constant<int, 0>{}, std::false_type{});
        step(s + 1, std::integral_constant<int, 1>{}, std::false_type{});
    }
    STAMP(st_loop_end);
    // the wave's two diagonal steps -- and every step of a batch row with padded keys -- run the same pipelined body with
    // the masks applied; after its diagonal the wave only keeps staging K/V tiles for the waves below it (one barrier per
    // step, like every other step)
    for (; s < qA + 2; s += 2) {
        step(s, std::integral_constant<int, 0>{}, std::true_type{});
        step(s + 1, std::integral_constant<int, 1>{}, std::true_type{});
    }
    for (; s < ntw; ++s) {
        if (s + 1 < ntw) {
            *(u32x4*)(smem + OFF_K + ((s & 1) ^ 1) * TILE_BYTES + st_offR) = kreg;
            *(u32x4*)(smem + OFF_V + ((s & 1) ^ 1) * TILE_BYTES + st_offT) = vreg;
        }
        prefetch_next();
        __syncthreads();
    }

    // ---- epilogue: ctx[b, i0+a, hd*64 + c] = O^T[c][a] / l ; lse = m + ln l ---------------------
    STAMP(st_epi);
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const int i0 = (qA + X) * 32;
        const float l_tot = l_run[X] + __shfl_xor(l_run[X], 32, 64);
        const float inv = 1.f / l_tot;
        store_rows_lds(ctx + ((size_t)b * L + i0) * d + hd * 64, (size_t)d, o[X][0], o[X][1], lane, inv,
                       smem + band_base + X * BAND_BYTES);
#ifdef MGX_F2_STAMP
        if (X == 0) continue;                            // block A's lse rows carry the stamps below
#endif
        if (hh == 0) lse_out[((size_t)b * heads + hd) * L + i0 + a] = m_ref[X] + __logf(l_tot);
    }
#ifdef MGX_F2_STAMP
    {
        STAMP(st_end);
        st_tail = st_epi - st_loop_end;
        float* o = lse_out + ((size_t)b * heads + hd) * L + qA * 32;
        if (lane == 0) {
            o[0] = (float)(st_end - st_begin); o[1] = (float)(st_loop - st_begin); o[2] = (float)(st_loop_end - st_loop);
            o[3] = (float)st_p12; o[4] = (float)st_bar; o[5] = (float)st_p3; o[6] = (float)st_tail;
            o[7] = (float)(st_end - st_epi); o[8] = (float)nmain; o[9] = (float)qb; o[10] = (float)w; o[11] = (float)st_p1;
        }
    }
#endif
}

static int fwd64_batch_group(int B, int L, int d) {
    const double per_row = (double)L * d * 2 * 4;
    int g = B;
    while (g > 1 && (g * per_row > 110e6 || B % g != 0)) --g;
    return g;
}

// launched by mgx_rel_attn_fwd (rel_attn_fwd.hip) when L % 256 == 0; the workspace already holds the fragment-ordered Er
int relattn::fwd64_launch(const uint16_t* qkv, const void* EfA, const uint32_t* padbits, uint16_t* ctx, float* lse, int B,
                          int L, int d, void* stream) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)rel_attn_fwd64_kernel,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, f2::LDS_BYTES);
    (void)attr;
    const int bg = fwd64_batch_group(B, L, d);
    MGX_REQUIRE((long)(L / 256) * (B / bg) <= 65535, MGX_ERR_SHAPE, "mgx_rel_attn_fwd: L/256 * batch groups too large");
    dim3 grid(bg * (d / 64), (L / 256) * (B / bg));
    hipLaunchKernelGGL(rel_attn_fwd64_kernel, grid, dim3(256), f2::LDS_BYTES, (hipStream_t)stream, qkv, (const u32x4*)EfA,
                       padbits, ctx, lse, L, d, bg);
    MGX_CHECK_LAUNCH("mgx_rel_attn_fwd(64-row waves)");
    return MGX_OK;
}
