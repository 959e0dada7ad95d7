// dK / dV of the fused relative attention with 64 KEYS PER WAVE (round 4; see rel_attn_bwd.hip for the math and for the
// 32-key kernel this one is measured against).
//
// Why: per 32 x 32 tile the 32-key wave reads every q / dO fragment (24 LDS reads), both statistics (8) and computes BOTH
// Q.Er^T chunk products of the tile (8 MFMA) for ONE key tile.  A wave that owns two adjacent key tiles J, J+1 shares all of
// that: the fragments and statistics of query tile I serve both, and the two tiles need the chunks I-J, I-J-1 and I-J-1, I-J-2,
// i.e. THREE products instead of four.  Per pair of tiles: 44 MFMA instead of 48, 64 LDS reads instead of 96, one E chunk
// load instead of two, one barrier instead of two.  The price is registers: 128 accumulators (dK^T, dV^T of 64 keys) + 64 for
// the K / V fragments + three E chunks -- one wave per SIMD with the whole 512-entry register file.
//
// RESULT (round 4, MI355X, cfg2 at batch 64; profiles/r04_dkv64.txt): bit-identical dk, dv and dS tiles, and SLOWER than the
// 32-key kernel -- 1.44-1.47 ms against 1.11-1.26, in both of its builds:
//  (a) plain: with more than 256 registers hipcc builds every MFMA in its AGPR form, so each of the seven 16-register score tiles
//      of a step that VALU code touches (three chunk products, two S, two dP) is copied between the register files: 144
//      v_accvgpr moves, 502 instructions per step of two tiles (two steps of the 32-key kernel: 447);
//  (b) MGX_K64_ASMACC (how _build.py --experiments builds it): -amdgpu-mfma-vgpr-form=1 for this file, with the 128 accumulators
//      kept in AGPRs by inline-asm MFMAs ("+a" operands, s_nop 1 in front for their just-packed operands) -- without that the flag
//      does not compile this kernel.  No accumulator copy is left in the main loop, 381 instructions per step ... and the SAME time.
// So the instruction count was not it: a wave that is alone on its SIMD has nobody to fill the latencies of its own dependent
// chain (LDS fragment read -> MFMA chain -> 12 wait states -> VALU -> pack -> MFMA ...), and hipcc does not software-pipeline the
// two sub-tiles against each other (pinning the sub-tile order with sched_barrier makes it worse, 1.65 ms).  The 44 MFMAs of a step
// need 1,408 cycles; the step takes about 3,700.  What it would take is the schedule written by hand -- the MFMAs of one sub-tile
// issued every ~8 instructions between the VALU / LDS work of the other and of the next step's chunk products -- i.e. a main loop in
// assembly.  Kept here, with its parity check (check_dkv64.py), as the starting point for that.
//
// Work decomposition: workgroup = 2 waves = 128 keys of one (batch, head) (the same 128-key blocks, grid and dispatch order
// as the 32-key kernel, so two workgroups share a CU = one wave per SIMD); wave w owns key tiles 2w, 2w+1 of the block.
// Everything else -- LDS images, DMA staging, the lane-permutation skew, the stored dS tiles and their layout -- is the
// 32-key kernel's.
#include <type_traits>
#include "rel_attn_common.hpp"

using namespace relattn;

#ifndef MGX_K64_SB
#define MGX_K64_SB 0      // 1: pin the sub-tile phase order with sched_barrier (measured: 1.65 ms against 1.47 without)
#endif
#if MGX_K64_SB
#define K64_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define K64_SB()
#endif
#ifndef MGX_K64_ASMACC
#define MGX_K64_ASMACC 0    // 1 (with -mllvm -amdgpu-mfma-vgpr-form=1 for this file): the dK / dV accumulators live in AGPRs through
#endif                      //    inline-asm MFMAs, every other MFMA is the builtin in its VGPR form (no v_accvgpr copies of score tiles)
// acc (AGPRs) += a * b.  The operands may have just been written by VALU code (the bf16 packs): s_nop 1 = the two wait states an MFMA
// operand needs after a VALU write (cdna_hip_programming.md 5.7 item 2).  The accumulators are only read in the epilogue.
MGX_DEV void mfma_acc(f32x16& acc, const bf16x8& a, const bf16x8& b) {
#if MGX_K64_ASMACC
    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
#else
    acc = mfma(a, b, acc);
#endif
}
namespace k64 {   // (LDS layout)
constexpr int WAVES = 2;
constexpr int OFF_QR = 0;                                  // 2 x 4K  q image R
constexpr int OFF_QT = OFF_QR + 2 * TILE_BYTES;            // 2 x 4K  q image T
constexpr int OFF_OR = OFF_QT + 2 * TILE_BYTES;            // 2 x 4K  dO image R
constexpr int OFF_OT = OFF_OR + 2 * TILE_BYTES;            // 2 x 4K  dO image T
constexpr int ST_BYTES = 512;                              // per buffer: 2 waves x (-lse2[32], -delta[32])
constexpr int OFF_ST = OFF_OT + 2 * TILE_BYTES;
constexpr int PATCH_BYTES = 4608;                          // per wave: the epilogue's row-major store patch
constexpr int OFF_PATCH = OFF_ST + 2 * ST_BYTES;
constexpr int OFF_FLAG = OFF_PATCH + WAVES * PATCH_BYTES;
constexpr int LDS_BYTES = OFF_FLAG + 16;                   // 43,024 B
}  // namespace k64

__global__ __launch_bounds__(128, 1) void rel_attn_dkv64_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ EfA, const uint32_t* __restrict__ padbits,
    const uint16_t* __restrict__ dctx, const float* __restrict__ nlse2 /* -lse log2(e) */, const float* __restrict__ ndelta /* -delta */,
    uint16_t* __restrict__ dqkv, uint16_t* __restrict__ dst, int L, int d, int bgroup) {
    using namespace k64;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bl = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    const int nkb = L >> 7;                                // L % 128 == 0 (host check)
    const int b = (blockIdx.y / nkb) * bgroup + blockIdx.x / heads, hd = blockIdx.x % heads;
    const int J0 = (blockIdx.y % nkb) * 128;               // small J0 = longest sweep = dispatched first
    const int nchunk = L >> 5;
    const int nT = (L - J0) >> 5;                          // query tiles i0 = J0 + 32 t, t = 0 .. nT-1  (nT >= 4)
    const int wk = 2 * w;                                  // the wave's first key tile inside the block: dq0 = t - wk, dq1 = dq0 - 1
    const int j0 = J0 + wk * 32;
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;
    const size_t stat_base = ((size_t)b * heads + hd) * L;

    // ---- staging by LDS-DMA (rel_attn_common.hpp): thread tid owns the 16-byte slots tid and tid + 128 of every 4 KB image ----
    const char* q_base = (const char*)(qkv_b + (size_t)J0 * ld + hd * 64);
    const char* o_base = (const char*)(dctx + ((size_t)b * L + J0) * d + hd * 64);
    uint32_t q_voffR[2], q_voffT[2], o_voffR[2], o_voffT[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tid + 128 * i, srow = s >> 3, spc = s & 7;
        const int lcR = spc ^ ((srow >> 1) & 7), lcT = spc ^ (((srow >> 1) & 1) << 2);
        q_voffR[i] = (uint32_t)((srow * ld + lcR * 8) * 2); q_voffT[i] = (uint32_t)((srow * ld + lcT * 8) * 2);
        o_voffR[i] = (uint32_t)((srow * d + lcR * 8) * 2);  o_voffT[i] = (uint32_t)((srow * d + lcT * 8) * 2);
    }
    const uint32_t q_step = (uint32_t)(32 * ld * 2), o_step = (uint32_t)(32 * d * 2);
    const uint32_t st_voff = (uint32_t)(((lane & 32) ? (const char*)ndelta - (const char*)nlse2 : 0) + (lane & 31) * 4);
    const char* st_base = (const char*)(nlse2 + stat_base + J0);
    const uint32_t lds0 = lds_addr_of(smem);
    auto stage = [&](int t, int buf) {                     // query tile t (clamped) -> LDS buffers `buf`
        const int tn = min(t, nT - 1);
        const char* qb = q_base + (size_t)tn * q_step;
        const char* ob = o_base + (size_t)tn * o_step;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t dstw = lds0 + buf * TILE_BYTES + (w + 2 * i) * 1024;      // slots 64 (w + 2i) .. + 63
            dma16(qb, q_voffR[i], dstw + OFF_QR);
            dma16(qb, q_voffT[i], dstw + OFF_QT);
            dma16(ob, o_voffR[i], dstw + OFF_OR);
            dma16(ob, o_voffT[i], dstw + OFF_OT);
        }
        dma4(st_base + (size_t)tn * 128, st_voff, lds0 + OFF_ST + buf * ST_BYTES + w * 256);
    };
    stage(0, 0);
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto e_frag = [&](int q, int ks) {
        return __builtin_bit_cast(bf16x8, *(const u32x4*)((const char*)EfA + (size_t)min(max(q, 0), nchunk - 1) * 4096 + ks * 1024 + lane16));
    };

    // K / V row fragments of the wave's two key tiles; three E chunk slots
    bf16x8 kf[2][4], vf[2][4], e[3][4];
    uint32_t padlane[2] = {0, 0};
    int wgpad = 0;
    {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint16_t* kp = qkv_b + (size_t)(j0 + 32 * u + bl) * ld + d + hd * 64 + hh * 8;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[u][ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(kp + ks * 16));
                vf[u][ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(kp + d + ks * 16));
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { e[0][ks] = e_frag(0, ks); e[1][ks] = e[0][ks]; e[2][ks] = e[0][ks]; }
        if (padbits) {
            uint32_t any = 0;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint32_t pwv = padbits[(size_t)b * nchunk + (j0 >> 5) + u];
                padlane[u] = (pwv >> bl) & 1u;
                any |= pwv;
            }
            if (tid == 0) *(volatile uint32_t*)(smem + OFF_FLAG) = 0u;
            __syncthreads();
            if (any) *(volatile uint32_t*)(smem + OFF_FLAG) = 1u;
            __syncthreads();
            wgpad = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)(smem + OFF_FLAG));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    uint32_t rd[16];                                        // ds_bpermute source-lane addresses of the skew (rel_attn_bwd.hip)
#pragma unroll
    for (int r = 0; r < 16; ++r) rd[r] = (uint32_t)((hh * 32 + ((crow(r, hh) - bl) & 31)) << 2);
    f32x16 dk[2][2], dv[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) { dk[u][ct] = zero16(); dv[u][ct] = zero16(); }
    const size_t ntri = (size_t)nchunk * (nchunk + 1) / 2;
    char* ds_col = (char*)(dst + (((size_t)b * heads + hd) * ntri + (size_t)(j0 >> 5)) * 1024) + lane16;
    auto ds_tile = [&](int t) {                           // tile (I = J0/32 + t, J = j0/32); the wave's second tile is 2 KB further
        const size_t I = (size_t)(J0 >> 5) + t;
        return ds_col + (I * (I + 1) / 2) * 2048;
    };

    // ---- one query tile against the wave's two key tiles.  SH: E slot of chunk dq0 (chunk dq0-1 in slot (SH+2)%3, chunk dq0-2 in
    //      (SH+1)%3, which then receives chunk dq0+1); MASKED: diagonal / padded-key / not-yet-started masks -------------------------
    auto tile2 = [&](int dq0, int cur, auto sh_tag, auto masked_tag, char* dsp, int tnext) {
        constexpr int SH = decltype(sh_tag)::value, SM = (SH + 2) % 3, SL = (SH + 1) % 3;
        constexpr bool MASKED = decltype(masked_tag)::value;
        const char* qr = smem + OFF_QR + cur * TILE_BYTES;
        bf16x8 qa[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qa[ks] = frag_R(qr, bl, hh, ks);
        f32x16 t0 = zero16(), t1 = zero16(), t2 = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) t0 = mfma(qa[ks], e[SH][ks], t0);
        if (!MASKED || dq0 >= 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) t1 = mfma(qa[ks], e[SM][ks], t1);
        }
        if (!MASKED || dq0 >= 2) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) t2 = mfma(qa[ks], e[SL][ks], t2);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[SL][ks] = e_frag(dq0 + 1, ks);      // the next step's chunk dq0
        stage(tnext, cur ^ 1);                             // after the E loads (see rel_attn_bwd.hip)
        if (!MASKED) __builtin_amdgcn_sched_barrier(0x78F);
        // The rest of the step is written sub-tile by sub-tile, in the order a lone wave should issue it -- with one wave per SIMD
        // nobody else fills the gaps --: MFMA groups of one sub-tile next to the VALU / LDS work of the other, and every 16-register
        // tile of scores packed to its 8-register bf16 operand form as early as possible (the register file is full: 128
        // accumulators, 64 K / V fragments, 48 E fragments).  sched_barrier pins the phase order, not the order inside a phase.
        const char* st = smem + OFF_ST + cur * ST_BYTES + w * 256;
        const char* orr = smem + OFF_OR + cur * TILE_BYTES;
        f32x16 nl, nd;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 d4 = *(const f32x4*)(st + 128 + (8 * g4 + 4 * hh) * 4);
            const f32x4 l4 = *(const f32x4*)(st + (8 * g4 + 4 * hh) * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { nd[4 * g4 + k] = d4[k]; nl[4 * g4 + k] = l4[k]; }
        }
        bf16x8 of[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) of[ks] = frag_R(orr, bl, hh, ks);
        // sub-tile 0: merge (hi chunk for keys <= query, lo chunk beyond), skew (lane permutation inside each half-wave), S, dP
        f32x16 c0, c1, dp0 = nd, dp1 = nd;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m0 = (bl <= crow(r, hh)) ? t0[r] : t1[r];
            c0[r] = __int_as_float(__builtin_amdgcn_ds_bpermute((int)rd[r], __float_as_int(m0)));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c0 = mfma(qa[ks], kf[0][ks], c0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp0 = mfma(of[ks], vf[0][ks], dp0);
        K64_SB();
        // sub-tile 1: the same, under sub-tile 0's MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m1 = (bl <= crow(r, hh)) ? t1[r] : t2[r];
            c1[r] = __int_as_float(__builtin_amdgcn_ds_bpermute((int)rd[r], __float_as_int(m1)));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c1 = mfma(qa[ks], kf[1][ks], c1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp1 = mfma(of[ks], vf[1][ks], dp1);
        K64_SB();
        // softmax / dS of sub-tile 0 (under sub-tile 1's MFMAs), straight into the bf16 operand fragments
        bf16x8 pf0[2], df0[2], pf1[2], df1[2];
        if (MASKED) {                                      // future keys of a diagonal tile, padded keys, a sub-tile that has not started
            const int dq1 = dq0 - 1;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const bool fut = bl > crow(r, hh);
                if ((dq0 == 0 && fut) || padlane[0]) c0[r] = -INFINITY;
                if (dq1 < 0 || (dq1 == 0 && fut) || padlane[1]) c1[r] = -INFINITY;
            }
        }
        auto soft = [&](f32x16& c, const f32x16& dp, bf16x8 (&pf)[2], bf16x8 (&df)[2]) {
            f32x16 ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(c[r], 0.125f * LOG2E, nl[r]));     // c = 8 S
                c[r] = p;
                ds[r] = p * dp[r];
            }
            pf[0] = acc_to_frag(c, 0); pf[1] = acc_to_frag(c, 1);
            df[0] = acc_to_frag(ds, 0); df[1] = acc_to_frag(ds, 1);
        };
        soft(c0, dp0, pf0, df0);
        K64_SB();
        const char* ot = smem + OFF_OT + cur * TILE_BYTES;
        const char* qt = smem + OFF_QT + cur * TILE_BYTES;
        // dV / dK of sub-tile 0 (its MFMAs run under the softmax of sub-tile 1), then of sub-tile 1; each transposed fragment is read
        // once per sub-tile (reading it once for both would hold 32 more registers across the softmax)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                mfma_acc(dv[0][ct], frag_T(ot, lane, ss, ct), pf0[ss]);
                mfma_acc(dk[0][ct], frag_T(qt, lane, ss, ct), df0[ss]);
            }
        soft(c1, dp1, pf1, df1);
        K64_SB();
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                mfma_acc(dv[1][ct], frag_T(ot, lane, ss, ct), pf1[ss]);
                mfma_acc(dk[1][ct], frag_T(qt, lane, ss, ct), df1[ss]);
            }
        u32x4 dfx[2][2];
        dfx[0][0] = __builtin_bit_cast(u32x4, df0[0]); dfx[0][1] = __builtin_bit_cast(u32x4, df0[1]);
        dfx[1][0] = __builtin_bit_cast(u32x4, df1[0]); dfx[1][1] = __builtin_bit_cast(u32x4, df1[1]);
        __builtin_nontemporal_store(dfx[0][0], (u32x4*)dsp);
        __builtin_nontemporal_store(dfx[0][1], (u32x4*)(dsp + 1024));
        if (!MASKED || dq0 >= 1) {                          // the second key tile exists in the causal half only for J+1 <= I
            __builtin_nontemporal_store(dfx[1][0], (u32x4*)(dsp + 2048));
            __builtin_nontemporal_store(dfx[1][1], (u32x4*)(dsp + 3072));
        }
    };
    using S0_ = std::integral_constant<int, 0>; using S1_ = std::integral_constant<int, 1>; using S2_ = std::integral_constant<int, 2>;
    // general step: canonical slots = those of SH 0 (chunk dq0 in e[0], dq0-1 in e[2], dq0-2 in e[1]), restored by register moves
    auto general_step = [&](int t) {
        const int dq0 = t - wk;
        if (dq0 < 0) {
            stage(t + 1, (t & 1) ^ 1);
        } else {
            tile2(dq0, t & 1, S0_{}, std::true_type{}, ds_tile(t), t + 1);      // leaves chunk dq0+1 in e[1], dq0 in e[0], dq0-1 in e[2]
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { const bf16x8 nw = e[1][ks]; e[1][ks] = e[2][ks]; e[2][ks] = e[0][ks]; e[0][ks] = nw; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // after the E loads and the DMA a wave issues its four dS stores: the DMA has landed at vmcnt(4)
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); };

    int t = 0;
    const int nhead = wgpad ? nT : min(4, nT);
    for (; t < nhead; ++t) general_step(t);
    // ---- main loop (t >= 4: every sub-tile of both waves is full): six steps per trip, so that the LDS buffer (t & 1) and the E
    //      slot of chunk dq0 (t % 3, the same for both waves) of every step are compile-time constants.  Entry at t = 4: slot
    //      4 % 3 = 1 must hold chunk dq0, slot 0 chunk dq0-1, slot 2 chunk dq0-2 (canonical: 0, 2, 1): one rotation of the slots.
    if (t + 6 <= nT) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { const bf16x8 x = e[1][ks]; e[1][ks] = e[0][ks]; e[0][ks] = e[2][ks]; e[2][ks] = x; }
        for (; t + 6 <= nT; t += 6) {
            tile2(t - wk, 0, S1_{}, std::false_type{}, ds_tile(t), t + 1);         landed(); __syncthreads();
            tile2(t + 1 - wk, 1, S2_{}, std::false_type{}, ds_tile(t + 1), t + 2); landed(); __syncthreads();
            tile2(t + 2 - wk, 0, S0_{}, std::false_type{}, ds_tile(t + 2), t + 3); landed(); __syncthreads();
            tile2(t + 3 - wk, 1, S1_{}, std::false_type{}, ds_tile(t + 3), t + 4); landed(); __syncthreads();
            tile2(t + 4 - wk, 0, S2_{}, std::false_type{}, ds_tile(t + 4), t + 5); landed(); __syncthreads();
            tile2(t + 5 - wk, 1, S0_{}, std::false_type{}, ds_tile(t + 5), t + 6); landed(); __syncthreads();
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { const bf16x8 x = e[0][ks]; e[0][ks] = e[1][ks]; e[1][ks] = e[2][ks]; e[2][ks] = x; }
    }
    for (; t < nT; ++t) general_step(t);

    char* patch = smem + OFF_PATCH + w * PATCH_BYTES;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        uint16_t* row0 = dqkv + ((size_t)b * L + j0 + 32 * u) * ld + hd * 64;
        store_rows_lds(row0 + d, ld, dk[u][0], dk[u][1], lane, 0.125f, patch);       // dk = dS^T (q/8)
        store_rows_lds(row0 + 2 * d, ld, dv[u][0], dv[u][1], lane, 1.f, patch);
    }
}

namespace relattn {
int dkv64_launch(const uint16_t* qkv, const void* EfA, const uint32_t* padbits, const uint16_t* dctx, const float* nlse2,
                 const float* ndelta, uint16_t* dqkv, uint16_t* dst, int B, int L, int d, int bg, void* stream) {
    static const bool once = [] {
        hipFuncSetAttribute((const void*)rel_attn_dkv64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, k64::LDS_BYTES);
        return true;
    }();
    (void)once;
    const int heads = d / 64;
    const dim3 grid(bg * heads, (L / 128) * (B / bg));
    hipLaunchKernelGGL(rel_attn_dkv64_kernel, grid, dim3(128), k64::LDS_BYTES, (hipStream_t)stream, qkv, (const u32x4*)EfA, padbits, dctx,
                       nlse2, ndelta, dqkv, dst, L, d, bg);
    return 0;
}
}  // namespace relattn
