// Fused relative global attention, forward (replaces layers.py:86-106 + 111-133 of the reference).
//
// Work decomposition: workgroup = 4 waves = 128 consecutive query rows of one (batch, head); each wave
// owns 32 query rows and keeps O^T (64 x 32 fp32), a softmax reference m and the running sum l in
// registers.  The workgroup sweeps 32-key tiles from j0 = 0 up to its own diagonal; K and V tiles are staged
// once per workgroup in LDS (shared by the 4 waves), the next tile prefetched into registers while the
// current one is computed (one barrier per step).
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
//   instruction touched 32 B of 32 different rows.)
// * Lazy softmax reference: m is NOT the running maximum.  A tile is exponentiated against the current m
//   straight away (no max chain, no cross-half shuffle, no rescale of O); only when a lane's partial sum
//   shows that some exponent left the safe range (first tile; a score that jumps by > 40 nats) the tile
//   is redone against the true maximum and O, l are rescaled once.  exp2(S - m) with m <= true maximum is
//   exact to fp32 / bf16 relative precision however far m lags, so the result is the same softmax.
// * Loop structure: a workgroup's key tiles 0 .. Q0-1 lie strictly below the diagonal of ALL its four waves.
//   That main loop has one body without a single branch (every tile full, every load unconditional), so the
//   compiler keeps loads in flight across iterations and needs no register copies where paths would rejoin
//   (a build with one loop and a full/general branch inside spent ~45 v_mov per step on such copies and
//   waited for the next step's E loads at the end of every step).  The last (up to) four tiles -- the
//   128 x 128 diagonal block, where a wave is full, on its diagonal, or already done -- and batch rows with
//   padded keys run a general body.
//
// Algorithmic FLOPs per (b,h): 3 products x 2*64 x L(L+1)/2 (causal half) -- DESIGN.md.
#include "rel_attn_common.hpp"

using namespace relattn;

#ifndef MGX_EXPERIMENTS
#define MGX_EXPERIMENTS 0   // 1 (experiment builds only): environment knobs that change which kernel runs / its residency, and the two
#endif                      // alternative forward kernels of tools/experiments/.  The product library reads no environment variable here.
// (round 3 experiment, removed in round 4: the Q.Er^T product one step ahead of its tile -- band read at the top of the step, the
//  product's MFMAs under the softmax -- ran 0.567-0.569 ms against 0.562-0.563 at cfg2 batch 64: with three waves per SIMD the
//  shorter per-wave chain buys nothing; profiles/README.md)
#ifndef MGX_FWD_PEEL
#define MGX_FWD_PEEL 0      // timing experiments only (tools/peel_fwd.sh): 1 no E-fragment loads in the main loop | 2 no band round trip |
#endif                      // 4 no exponentials | 8 K / V prefetch re-reads tile 0 | 16 no parity XOR of the band-store addresses | 32 no row-sum
                            // adds | 64 no redo branch anywhere (main loop AND general body) | 128 no general steps after a main loop | 256 no fma in front of the exponentials; results are then wrong
namespace {
constexpr int WAVES = 4;
constexpr int OFF_K = 0;                                        // 2 x 4 KiB   image R
constexpr int OFF_V = OFF_K + 2 * TILE_BYTES;                   // 2 x 4 KiB   image T
constexpr int OFF_BAND = OFF_V + 2 * TILE_BYTES;                // 4 x (32 rows x 272 B) fp32 rotated band
constexpr int OFF_PAD = OFF_BAND + WAVES * BAND_BYTES;          // key-padding words of this batch row (first 256)
constexpr int OFF_FLAG = OFF_PAD + 1024;                        // "this batch row has padded keys" flag
constexpr int LDS_BYTES = OFF_FLAG + 16;                        // 52,240 B -> 3 workgroups per CU
constexpr float M_INIT = -1.0e37f;      // "no reference yet": finite, so exp2(-inf - m) is 0 and not NaN
constexpr float L_SAFE = 1.0e24f;       // a lane's partial sum of one tile above this => redo against the true max
}  // namespace

// WRITE_W = false: the training/inference forward (ctx + lse).
// WRITE_W = true : debug/eval output of the reference (layers.py:102,109): the same sweep recomputes S and
//                  writes weights[b,h,i,j] = exp(S - lse_i) (fp32, caller pre-zeroes the future triangle).
// CAUSAL = false: the reference's generate() call, Decoder(x, mask=None) (network.py:60-62): every query attends to EVERY key
//                  j < Lk (no look-ahead, no padding mask) while the relative term stays what _qe_masking + _skewing leave
//                  of it -- q_i.E[M-1-(i-j)] for j <= i, zero for j > i (layers.py:111-133).  Inference only; all tiles run
//                  the general body.
template <bool WRITE_W, bool CAUSAL = true>
__global__ __launch_bounds__(256, 3) void rel_attn_fwd_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ Ef /* fragment-ordered Er, see er_frag_kernel */,
    const uint32_t* __restrict__ padbits, uint16_t* __restrict__ ctx, float* __restrict__ lse_out,
    const float* __restrict__ lse_in, float* __restrict__ weights, int L, int d, int bgroup, int Lk = 0) {
    extern __shared__ __attribute__((aligned(256))) char smem[];     // 256: the band stores XOR bit 7 of absolute LDS addresses
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    // grid: x = (batch, head) [fast], y = query-block rank [slow], heaviest (latest) blocks first, so the
    // whole grid is dispatched longest-job-first; blocks of one (b,h) share an XCD when B*h % 8 == 0.
    // Batch groups: the grid's slow axis is (group of `bgroup` batch rows, query-block rank): one group's q, k, v and ctx
    // (~100 MB) stay inside the 256 MB Infinity Cache while its workgroups run; with the whole cfg2 batch of 32 in one
    // sweep (268 MB) the K/V re-reads of the 16 query blocks of a (b,h) fall through to HBM.
    const int nqb = (L + 127) >> 7;
    const int b = (blockIdx.y / nqb) * bgroup + blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qb = nqb - 1 - (blockIdx.y % nqb);
    const int I0 = qb * 128, Q0 = I0 >> 5;
    const int nchunk = L >> 5;                           // number of 32-row chunks / key tiles
    const bool wave_on = I0 + w * 32 < L;
    // a wave beyond the end of the sequence (L % 128 != 0) shadows the last valid 32-row block and stores nothing
    const int q0 = wave_on ? Q0 + w : nchunk - 1;        // the wave's diagonal tile / first "hi" chunk
    const int i0 = q0 * 32;
    const int ntw = CAUSAL ? min(Q0 + 4, nchunk) : (Lk + 31) >> 5;      // key tiles this workgroup visits
    const size_t ld = (size_t)3 * d;                     // qkv row stride (elements)
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;

    // ---- staging: K and V tiles go global -> LDS by DMA (rel_attn_common.hpp: dma16).  Thread tid owns the 16-byte slot tid of
    // both images -- row tid >> 3, PHYSICAL chunk tid & 7 -- and fetches the logical chunk the image's swizzle puts there: a tile
    // costs a wave two DMA instructions, no registers and no ds_write (until round 4: two loads parked in registers for a step,
    // then two stores).  Every global address of the sweep is (wave-uniform base in SGPRs) + (32-bit per-lane offset), every
    // request unconditional with a clamped index.
    const int srow = tid >> 3, spc = tid & 7;
    const char* kv_base = (const char*)(qkv_b + d + hd * 64);                 // K columns of this head; V is d elements further
    const uint32_t k_voff = (uint32_t)((srow * ld + (spc ^ ((srow >> 1) & 7)) * 8) * 2);              // image R (imgR_off inverted)
    const uint32_t v_voff = (uint32_t)((srow * ld + (spc ^ (((srow >> 1) & 1) << 2)) * 8) * 2 + d * 2);   // image T (imgT_off)
    const uint32_t tile_bytes = (uint32_t)(32 * ld * 2);                       // one 32-row step of qkv
    const uint32_t lds_w = lds_addr_of(smem) + w * 1024;                        // this wave's 1 KB of a 4 KB image
    auto stage_kv = [&](int t, int buf) {                                      // key tile t -> LDS buffers `buf`
        const char* tb = kv_base + (size_t)t * tile_bytes;
        dma16(tb, k_voff, lds_w + OFF_K + buf * TILE_BYTES);
        dma16(tb, v_voff, lds_w + OFF_V + buf * TILE_BYTES);
    };
    // the DMA of a step is the last VMEM operation the wave issues in it: everything has landed at vmcnt(0)
    auto landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto ef = [&](int q, int ks) {
        return __builtin_bit_cast(bf16x8, *(const u32x4*)((const char*)Ef + (size_t)max(q, 0) * 4096 + ks * 1024 + lane16));
    };

    // ---- prologue: K/V tile 0, key-padding words ---------------------------------------------------
    stage_kv(0, 0);
    // (no __syncthreads_or: it allocates static LDS, which moves the dynamic base off 0 and costs one v_add per band store)
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
        if (acc) *(volatile uint32_t*)(smem + OFF_FLAG) = 1u;      // every writer stores the same value
        __syncthreads();
        anypad = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)(smem + OFF_FLAG));
    }
    auto padword = [&](int kt) -> uint32_t {             // wave-uniform
        if (!anypad) return 0u;
        uint32_t v = *(const uint32_t*)(smem + OFF_PAD + 4 * min(kt, 255));      // unconditional LDS read (no pointer select)
        if (kt >= 256) v = padbits[(size_t)b * nchunk + kt];                     // L > 8192 only
        return __builtin_amdgcn_readfirstlane(v);
    };
    // Q fragments (A operand of QE, B operand of S^T), pre-scaled by 1/8
    bf16x8 qf[4], e[4];
    {
        const uint16_t* qp = qkv_b + (size_t)(i0 + a) * ld + hd * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 raw = *(const u32x4*)(qp + ks * 16);
            float f[8];
            unpack8(raw, f);
#pragma unroll
            for (int k = 0; k < 8; ++k) f[k] *= 0.125f;
            qf[ks] = __builtin_bit_cast(bf16x8, pack8(f));
            e[ks] = ef(q0, ks);
        }
    }
    landed();
    __syncthreads();

    // band addressing (rel_attn_common.hpp): register r writes row slot r of region hh.  wcl[r] = absolute LDS address of
    // (wave band + region + column byte offset): every term but the column is a multiple of 256, so XOR-ing bit 7 of the
    // whole value flips the chunk parity; the row slot r*272 is the instruction's immediate offset.  A lane reads its own
    // row with four ds_read_b128 at rbase + 32*g4 (+128 when D/32 is odd).
    const int band_base = OFF_BAND + w * BAND_BYTES;
    uint32_t wcl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        wcl[r] = lds_addr_of(smem) + band_base + hh * BAND_REGION + (((crow(r, hh) - a) & 63) << 2);
    const int rbase = band_base + band_rowoff(a) + 16 * hh;
    auto band_put = [&](const f32x16& v, int q) {        // chunk q of Q.Er^T -> band
        const uint32_t tog = (q & 1) << 7;
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_store_f32(((MGX_FWD_PEEL & 16) ? wcl[r] : (wcl[r] ^ tog)) + r * BAND_STRIDE, v[r]);
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
    {
        f32x16 qe = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], e[ks], qe);
        band_put(qe, q0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[ks] = ef(q0 - 1, ks);      // new chunk of step 0
    }

    f32x16 o0 = zero16(), o1 = zero16();
    float m_ref = M_INIT, l_run = 0.f;
    float lse2w = 0.f;
    if (WRITE_W) lse2w = lse_in[((size_t)b * heads + hd) * L + i0 + a] * LOG2E;

    // ---- softmax of one tile against the lazy reference, then O^T += V^T P^T ------------------------
    // P^T goes straight into the bf16 operand fragments of the O^T product (k order kappa, see acc_to_frag)
    auto exp_tile = [&](const f32x16& c, float mneg, bf16x8 (&pf)[2]) {
        // the tile's partial row sum is formed as four interleaved chains s_j = p[j] + p[j+4] + p[j+8] + p[j+12], then (s0 + s1) + (s2 + s3):
        // the order the hand-scheduled 64-row kernel (rel_attn_fwd64.hip) uses -- a single chain of 15 dependent adds stalls a lone
        // wave -- so the two kernels give the same bits
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // (peel 256, round 6, timing only: the exponent's argument taken as it is -- what the 16 v_fma of a tile cost, i.e. what folding
            //  log2(e) / 8 into the q operand and -m into the chunk product's initial accumulator could win at most)
            const float ar = (MGX_FWD_PEEL & 256) ? c[r] : __builtin_fmaf(c[r], LOG2E, mneg);
            p[r] = (MGX_FWD_PEEL & 4) ? ar * 1e-9f : __builtin_amdgcn_exp2f(ar);
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            u32x4 wv;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) wv[jj] = pack_bf16x2(p[8 * ss + 2 * jj], p[8 * ss + 2 * jj + 1]);
            pf[ss] = __builtin_bit_cast(bf16x8, wv);
        }
        if (MGX_FWD_PEEL & 32) return 0.f;
        float sj[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sj[j] = ((p[j] + p[j + 4]) + p[j + 8]) + p[j + 12];
        return (sj[0] + sj[1]) + (sj[2] + sj[3]);
    };
    auto softmax_pv = [&](const f32x16& c, int cur) {
        bf16x8 pf[2];
        float lsum = exp_tile(c, -m_ref * LOG2E, pf);
        if (!(MGX_FWD_PEEL & 64) && __builtin_expect(__any(!(lsum <= L_SAFE)), 0)) {
            // redo against the true maximum (both lane halves of a query row must agree on m)
            float tmax = c[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, c[r]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_ref, tmax);       // finite: every visited tile has a key j <= i (or PAD_NEG)
            const float alpha = __builtin_amdgcn_exp2f((m_ref - m_new) * LOG2E);
            lsum = exp_tile(c, -m_new * LOG2E, pf);
            l_run *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            m_ref = m_new;
        }
        l_run += lsum;
        const char* vt = smem + OFF_V + cur * TILE_BYTES;
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            o0 = mfma(frag_T(vt, lane, ss, 0), pf[ss], o0);
            o1 = mfma(frag_T(vt, lane, ss, 1), pf[ss], o1);
        }
    };
    const int am = a - 4 * hh;                           // key crow(r,hh) is in the future of query a  <=>  crow(r,0) > am

    // ---- main loop: tiles strictly below every wave's diagonal, no padded keys: ONE branch-free body -------------
    const int nmain = (WRITE_W || anypad || !CAUSAL) ? 0 : Q0;       // Q0 <= ntw - 1: a next tile always exists inside this loop
    int s = 0;
    for (; s < nmain; ++s) {
        const int cur = s & 1;
        const int tn = (MGX_FWD_PEEL & 8) ? 0 : min(s + 1, ntw - 1);
        const int dq = q0 - s;                            // >= 1
        f32x16 c = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma(qf[ks], e[ks], c);
        if (!(MGX_FWD_PEEL & 2)) {
            band_put(c, dq - 1);
            wave_lds_fence();
            c = band_get(dq);
            wave_lds_fence();
        }
        if (!(MGX_FWD_PEEL & 1)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) e[ks] = ef(dq - 2, ks);          // chunk of the next step
        }
        // the next key tile, into the buffers every wave left at the last barrier.  AFTER the E loads: the compiler's wait for
        // those at the top of the next step (it does not know about the DMA) then is the vmcnt(0) this step ends with anyway
        stage_kv(tn, cur ^ 1);
        __builtin_amdgcn_sched_barrier(0x78F);             // VMEM may not sink below: the fragments are needed at the top of the next step
        const char* kt = smem + OFF_K + cur * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma(frag_R(kt, a, hh, ks), qf[ks], c);
        softmax_pv(c, cur);
        landed();
        __syncthreads();
    }

    // ---- general body: the diagonal 128 x 128 block (a wave is full / on its diagonal / done), padded keys,
    //      weights output ------------------------------------------------------------------------------------------
    if ((MGX_FWD_PEEL & 128) && nmain > 0) s = ntw;       // peel: no general steps after a main loop (what the diagonal block costs)
    for (; s < ntw; ++s) {
        const int cur = s & 1;
        const int tn = min(s + 1, ntw - 1);
        stage_kv(tn, cur ^ 1);
        const int dq = q0 - s;                            // causal: wave active iff dq >= 0
        if (!CAUSAL || dq >= 0) {
            const uint32_t pw = CAUSAL ? padword(s) : 0u;
            if (dq >= 1) {
                f32x16 qe = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], e[ks], qe);
                band_put(qe, dq - 1);
            }
            wave_lds_fence();
            f32x16 c = (CAUSAL || dq >= 0) ? band_get(dq) : zero16();        // tiles beyond the diagonal have no relative term
            wave_lds_fence();
            if (!CAUSAL && dq == 0) {                     // ... and on the diagonal tile only the keys b <= a have one
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (crow(r, 0) > am) ? 0.f : c[r];
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) e[ks] = ef(dq - 2, ks);
            const char* kt = smem + OFF_K + cur * TILE_BYTES;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) c = mfma(frag_R(kt, a, hh, ks), qf[ks], c);
            if (CAUSAL && dq == 0) {                      // diagonal tile: key b > query a is the future
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (crow(r, 0) > am) ? -INFINITY : c[r];
            }
            if (!CAUSAL && 32 * s + 32 > Lk) {            // last tile of a window that is not a multiple of 32: keys >= Lk do not exist
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (32 * s + crow(r, hh) >= Lk) ? -INFINITY : c[r];
            }
            if (pw) {                                     // padded key: the reference's additive -1e9 (future keys stay -inf)
                const uint32_t pwl = pw >> (4 * hh);
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (pwl & (1u << crow(r, 0))) ? fminf(c[r], PAD_NEG) : c[r];
            }
            if (WRITE_W) {
                // weights[b,h,i0+a, j0 + 8*g4 + 4*hh + k] = exp(S - lse); masked entries are exactly 0
                if (wave_on) {
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
                }
            } else {
                softmax_pv(c, cur);
            }
        }
        landed();
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

static void set_fwd_attrs() {
    // function-local static: initialised exactly once, thread-safe (C++11), as mgx.h promises for the whole library
    static const bool once = [] {
        hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        return true;
    }();
    (void)once;
}

extern "C" size_t mgx_rel_attn_fwd_workspace(int L) { return L > 0 ? er_frag_bytes(L) : 0; }

// batch rows per grid group: the largest divisor of B whose q/k/v/ctx bytes stay near 100 MB (see the kernel)
static int batch_group(int B, int L, int d) {
    const double per_row = (double)L * d * 2 * 4;
    int g = B;
#if MGX_EXPERIMENTS
    static const int forced = [] { const char* e = getenv("MGX_ATTN_BGROUP"); return e ? atoi(e) : 0; }();   // experiment knob
    if (forced > 0 && B % forced == 0) return forced;
#endif
    while (g > 1 && (g * per_row > 110e6 || B % g != 0)) --g;
    return g;
}

static int fwd_common_checks(const char* who, const void* ws, size_t ws_bytes, int B, int L, int d, int M) {
    MGX_REQUIRE(B > 0 && L > 0 && d > 0 && d % 64 == 0 && L % 32 == 0 && M >= L, MGX_ERR_SHAPE,
                "%s: need d%%64==0, L%%32==0, M>=L (got B=%d L=%d d=%d M=%d)", who, B, L, d, M);
    MGX_REQUIRE((long)((L + 127) / 128) * B <= 65535, MGX_ERR_SHAPE, "%s: L/128 * B too large for the grid", who);
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
    const int bg = batch_group(B, L, d);
#if MGX_EXPERIMENTS
    // experiment builds only (`_build.py --variant NAME --experiments`, tools/experiments/), all measured slower: MGX_ATTN_FWD64 = 3 the
    // 64-rows-per-wave kernel with the generated asm sweep (rel_attn_fwd64.hip; L % 128 == 0, L <= 8192), 2 the ping-pong kernel
    // (rel_attn_fwd3.hip), 1 the 64-rows-per-wave HIP kernel (rel_attn_fwd2.hip) (both L % 256 == 0)
    {
        const int mode = env_digit("MGX_ATTN_FWD64", 0);
        if (mode == 3 && L % 128 == 0 && L <= 8192) return fwd64a_launch(qkv, workspace, padbits, ctx, lse, B, L, d, bg, stream);
        if (mode == 2 && L % 256 == 0) return fwdpp_launch(qkv, workspace, padbits, ctx, lse, B, L, d, stream);
        if (mode == 1 && L % 256 == 0) return fwd64_launch(qkv, workspace, padbits, ctx, lse, B, L, d, stream);
    }
#endif
    dim3 grid(bg * (d / 64), ((L + 127) / 128) * (B / bg));
#if MGX_EXPERIMENTS
    static const int occ_lds = [] {        // experiment: MGX_FWD_LDS pads the dynamic LDS to lower the residency (timing only)
        const char* e = getenv("MGX_FWD_LDS");
        const int v = e ? atoi(e) : 0;
        if (v > LDS_BYTES) hipFuncSetAttribute((const void*)rel_attn_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, v);
        return v > LDS_BYTES ? v : LDS_BYTES;
    }();
#else
    constexpr int occ_lds = LDS_BYTES;
#endif
    hipLaunchKernelGGL(rel_attn_fwd_kernel<false>, grid, dim3(256), occ_lds, (hipStream_t)stream, qkv,
                       (const u32x4*)workspace, padbits, ctx, lse, (const float*)nullptr, (float*)nullptr, L, d, bg);
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
    hipLaunchKernelGGL(rel_attn_fwd_kernel<true>, grid, dim3(256), LDS_BYTES, (hipStream_t)stream, qkv,
                       (const u32x4*)workspace, padbits, (uint16_t*)nullptr, (float*)nullptr, lse, weights, L, d, B);
    MGX_CHECK_LAUNCH("mgx_rel_attn_weights");
    return MGX_OK;
}

// The reference's sampling call Decoder(x, mask=None) (network.py:60-62): bidirectional over the Lk <= L real positions of
// the window (rows / keys Lk..L-1 are padding up to the kernels' multiple of 32: never attended to, their own outputs are
// don't-cares), relative term for j <= i only.  Inference only (no lse / backward contract beyond the forward's).
extern "C" int mgx_rel_attn_fwd_nomask(const uint16_t* qkv, const uint16_t* E, uint16_t* ctx, float* lse, void* workspace,
                                       size_t ws_bytes, int B, int L, int Lk, int d, int M, void* stream) {
    MGX_REQUIRE(qkv && E && ctx && lse, MGX_ERR_NULL, "mgx_rel_attn_fwd_nomask: NULL pointer");
    if (int rc = fwd_common_checks("mgx_rel_attn_fwd_nomask", workspace, ws_bytes, B, L, d, M)) return rc;
    MGX_REQUIRE(Lk > 0 && Lk <= L, MGX_ERR_SHAPE, "mgx_rel_attn_fwd_nomask: need 0 < Lk <= L (got Lk=%d L=%d)", Lk, L);
    set_fwd_attrs();
    launch_er_frag(E + (size_t)(M - L) * 64, (u32x4*)workspace, nullptr, L, (hipStream_t)stream);
    dim3 grid(B * (d / 64), (L + 127) / 128);
    hipLaunchKernelGGL((rel_attn_fwd_kernel<false, false>), grid, dim3(256), LDS_BYTES, (hipStream_t)stream, qkv,
                       (const u32x4*)workspace, (const uint32_t*)nullptr, ctx, lse, (const float*)nullptr, (float*)nullptr, L, d, B, Lk);
    MGX_CHECK_LAUNCH("mgx_rel_attn_fwd_nomask");
    return MGX_OK;
}
