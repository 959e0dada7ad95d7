// Fused relative global attention, backward (autograd of layers.py:86-106 of the reference).
//
// With qs = q/8 (exact pre-scale), delta = i-j, Er[delta] = E[M-1-delta]:
//     S[i,j]  = qs_i.k_j + qs_i.Er[i-j]            P = exp(S - lse_i)   (0 where masked)
//     dP[i,j] = dO_i.v_j                           dS = P o (dP - rowsum(dO o O)_i)
//     dqs_i   = sum_j dS[i,j] (k_j + Er[i-j])      dq = dqs/8
//     dk_j    = sum_i dS[i,j] qs_i                 dv_j = sum_i P[i,j] dO_i
//     dEr[dl] = sum_{b,h} sum_i dS[i,i-dl] qs_i
//
// Every output has a different "owner" axis (query row / key row / relative distance), and summing a non-owned output
// across workgroups with float atomics would cost several GB of atomic traffic per layer at cfg2.  So ONE kernel
// recomputes P and dS flash-style -- the dK/dV kernel, whose outputs need both -- and stores every bf16 dS tile (the operand
// registers of its own dK product) in the workspace; dQ and dE only need dS and are computed from the stored tiles:
//   K2  dkv_kernel   : workgroup = 128 keys, sweeps query tiles (24 MFMA / 32x32 tile); stores dS by (query tile, key tile)
//   K1L dq_lite      : workgroup = 128 query rows, sweeps the stored tiles of its rows (8 MFMA / tile, HBM-bound)
//   K3t de_tiles     : workgroup = 4 tile diagonals, un-skews the stored tiles in LDS (8 MFMA / 64 rows, HBM-bound)
//   K1  dq_kernel    : dQ by full recomputation (20 MFMA / tile) -- the round-2 kernel, kept as an independent cross-check
//   K3  de_kernel    : dE by full recomputation (24 MFMA / tile) -- cross-check (parts bit 4)
// Until round 3 dQ and dK/dV each recomputed P (two kernels x ~0.6 ms per layer at cfg2); reading dS back costs the dQ side
// 0.26 ms instead (profiles/r03_bwd_pipeline_ab.txt).  The skew between (i,j) tiles and (i,delta) chunks is done through LDS
// (rel_attn_common.hpp); the only L x L object that ever exists is the bf16 dS workspace (causal half, tile-blocked),
// written once and read twice per layer.
#include <type_traits>
#include "rel_attn_common.hpp"

using namespace relattn;

#ifndef MGX_EXPERIMENTS
#define MGX_EXPERIMENTS 0   // see rel_attn_fwd.hip
#endif
namespace {
constexpr int WAVES = 4;
}  // namespace

// ================================================================================================
// delta[b,h,i] = sum_c dctx[b,i,h*64+c] * ctx[b,i,h*64+c]         (8 lanes per (row, head))
// ================================================================================================
__global__ __launch_bounds__(256) void attn_delta_kernel(const uint16_t* __restrict__ ctx,
                                                         const uint16_t* __restrict__ dctx, const float* __restrict__ lse,
                                                         float* __restrict__ delta, float* __restrict__ nlse2,
                                                         float* __restrict__ ndelta, int B, int L, int d) {
    const int heads = d >> 6;
    const long total = (long)B * L * heads * 8;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= total) return;   // total is a multiple of 8 and blockDim of 64: whole 8-lane groups exit together
    const int sub = (int)(gid & 7);
    // group g = ((b * heads + hd) * L + i): consecutive 8-lane groups are consecutive rows i of ONE head, so the per-row results
    // are written (and lse is read) contiguously -- with the head on the fast axis (rounds 1-3) every 4-byte store went to its own
    // cache line.  The 128-byte reads of ctx / dctx are whole lines either way.
    const long grp = gid >> 3;
    const int i = (int)(grp % L);
    const int hd = (int)((grp / L) % heads);
    const long row = (grp / ((long)L * heads)) * L + i;      // b*L + i
    const size_t off = (size_t)row * d + hd * 64 + sub * 8;
    float a[8], g[8];
    unpack8(*(const u32x4*)(ctx + off), a);
    unpack8(*(const u32x4*)(dctx + off), g);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += a[k] * g[k];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if (sub == 0) {
        const size_t si = (size_t)grp;
        delta[si] = s;
        // the dK/dV kernel's copies, in the form it consumes them (the initial accumulators of its S and dP products), so
        // that its LDS-DMA staging needs no arithmetic on the way
        ndelta[si] = -s;
        nlse2[si] = -lse[si] * LOG2E;
    }
}

// natural-k transposed fragment: X[16*ks + 8*hh + j][32*ct + (lane&31)], j = 0..7, from an image-T tile
MGX_DEV bf16x8 frag_Tn(const char* tile, int lane, int ks, int ct) {
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

// ================================================================================================
// K1: dQ by recomputation (cross-check of K1L; parts bit 5).  Same sweep as the forward (query-block owner, key tiles
//   0..diagonal).
//   orientation: keys on registers, queries on lanes (S^T, P^T, dP^T, dS^T), dqs^T[c][a] accumulators.
//   E never touches LDS here: the Er row fragments (B operand of Q.Er^T) and the fragments of the
//   transposed copy ErT[c][delta] (A operand of dqs^T += ErT . dQE^T) are loaded from global/L2.
//   K has ONE LDS image (R) that serves both the row reads (S^T) and the transposed reads (dq).
// ================================================================================================
namespace k1 {
constexpr int OFF_KR = 0;                                  // 2 x 4K  K image R (row + transposed reads)
constexpr int OFF_VR = OFF_KR + 2 * TILE_BYTES;            // 2 x 4K  V image R (row frags for dP^T)
constexpr int OFF_BAND = OFF_VR + 2 * TILE_BYTES;          // 4 x 8,704 B fp32 rotated band (see common.hpp)
constexpr int DB_STRIDE = 144;                             // bytes per dband row (64 bf16 + pad)
constexpr int OFF_DBAND = OFF_BAND + WAVES * BAND_BYTES;   // 4 x 4,608 B bf16 [32][72]: dS by (query, delta&63)
constexpr int OFF_PAD = OFF_DBAND + WAVES * 32 * DB_STRIDE; // key-padding words of this batch row (first 256)
constexpr int OFF_FLAG = OFF_PAD + 1024;                   // "this batch row has padded keys" flag
constexpr int LDS_BYTES = OFF_FLAG + 16;                   // 70,672 B -> 2 workgroups per CU
}  // namespace k1

__global__ __launch_bounds__(256, 2) void rel_attn_dq_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ EfA, const u32x4* __restrict__ EfT,
    const uint32_t* __restrict__ padbits, const uint16_t* __restrict__ dctx, const float* __restrict__ lse,
    const float* __restrict__ delta, uint16_t* __restrict__ dqkv, int L, int d, int bgroup) {
    using namespace k1;
    extern __shared__ __attribute__((aligned(256))) char smem[];     // 256: the band stores XOR bit 7 of absolute LDS addresses
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    // x = (b,h) of one batch group [fast], y = (batch group, heaviness rank) [slow]: one group's tensors (~100 MB) stay
    // inside the Infinity Cache while its workgroups run (see rel_attn_fwd.hip)
    const int nqb = (L + 127) >> 7;
    const int b = (blockIdx.y / nqb) * bgroup + blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qb = nqb - 1 - (blockIdx.y % nqb);
    const int I0 = qb * 128, Q0 = I0 >> 5;
    const int nchunk = L >> 5;
    const bool wave_on = I0 + w * 32 < L;
    // a wave beyond the end of the sequence (L % 128 != 0) shadows the last valid 32-row block: it recomputes that
    // block's values (its duplicate stores of delta / dS carry identical data) and skips the dq store
    const int q0 = wave_on ? Q0 + w : nchunk - 1;
    const int i0 = q0 * 32;
    const int ntw = min(Q0 + 4, nchunk);                 // key tiles this workgroup visits
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;

    const int srow = tid >> 3, sch = tid & 7;
    const int st_offR = imgR_off(srow, sch);
    // Every global address of the sweep is (wave-uniform base in SGPRs) + (32-bit per-lane offset) + immediate, so a load
    // costs no vector address arithmetic (a 64-bit per-lane pointer bumped per step cost 2-3 VALU + SALU per load).
    const char* kv_base = (const char*)(qkv_b + d + hd * 64);                 // K columns of this head; V is d elements further
    const uint32_t kv_voff = (uint32_t)((srow * ld + sch * 8) * 2);            // bytes
    const uint32_t tile_bytes = (uint32_t)(32 * ld * 2);                       // one 32-row step of qkv
    auto k_tile = [&](int t) { return *(const u32x4*)(kv_base + (size_t)t * tile_bytes + kv_voff); };
    auto v_tile = [&](int t) { return *(const u32x4*)(kv_base + (size_t)t * tile_bytes + (size_t)d * 2 + kv_voff); };
    // fragment-ordered copies of Er (er_frag_kernel, rel_attn_common.hpp): 1 KB contiguous per wave load.  Every load
    // of the sweep is unconditional with a clamped index (a load inside a branch makes the compiler drain the whole
    // VMEM queue where the branch rejoins); data of clamped tiles / chunks is never used.
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // Er row fragment ks of chunk q (row t = lane&31 of the chunk, i.e. delta = 32q + t)
    auto e_frag = [&](int q, int ks) {
        return __builtin_bit_cast(bf16x8, *(const u32x4*)((const char*)EfA + (size_t)max(q, 0) * 4096 + ks * 1024 + lane16));
    };
    // ErT fragment: row c = 32*ct + (lane&31), k = t = 16*ks + 8*hh + j of chunk q
    auto et_frag = [&](int q, int ks, int ct) {
        return __builtin_bit_cast(bf16x8, *(const u32x4*)((const char*)EfT + (size_t)max(q, 0) * 4096 + (2 * ks + ct) * 1024 + lane16));
    };

    {   // prologue staging
        *(u32x4*)(smem + OFF_KR + st_offR) = k_tile(0);
        *(u32x4*)(smem + OFF_VR + st_offR) = v_tile(0);
        // zero the dS band (its never-written half must read as 0 on the first step)
        for (int o = tid * 16; o < WAVES * 32 * DB_STRIDE; o += 256 * 16) *(u32x4*)(smem + OFF_DBAND + o) = u32x4{0, 0, 0, 0};
    }
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
    bf16x8 qf[4], dof[4], e[4];
    float lse2 = 0.f, dlt = 0.f;
    {
        const uint16_t* qp = qkv_b + (size_t)(i0 + a) * ld + hd * 64 + hh * 8;
        const uint16_t* dp = dctx + ((size_t)b * L + i0 + a) * d + hd * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = __builtin_bit_cast(bf16x8, scale8(*(const u32x4*)(qp + ks * 16), 0.125f));
            dof[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(dp + ks * 16));
            e[ks] = e_frag(q0, ks);                        // the wave's first "hi" chunk
        }
        const size_t si = ((size_t)b * heads + hd) * L + i0 + a;
        lse2 = lse[si] * LOG2E;
        dlt = delta[si];
    }
    __syncthreads();

    // band addressing (rotated band, rows placed so that the register index r is the row slot and the lane half hh
    // selects a 256-byte-aligned region).  wcl[r] = ABSOLUTE LDS address of (wave band + region + column byte offset): every
    // term but the column is a multiple of 256, so XOR-ing bit 7 of the whole value flips the chunk parity (one VALU per
    // store); the row slot r*272 is the instruction's immediate offset.
    const int band_base = OFF_BAND + w * BAND_BYTES;
    char* dband = smem + OFF_DBAND + w * (32 * DB_STRIDE);
    uint32_t wcl[16];
#pragma unroll
    for (int r = 0; r < 16; ++r)
        wcl[r] = lds_addr_of(smem) + band_base + hh * BAND_REGION + (((crow(r, hh) - a) & 63) << 2);
    const int rbase = band_base + band_rowoff(a) + 16 * hh;
    // PHYSICAL chunk parity = (chunk - q0) & 1: each wave has its own bands, so the assignment is free, and with it the
    // chunk stored in step s has parity (s + 1) & 1 and the tile read in step s parity s & 1 for EVERY wave -- compile-time
    // constants in the two-step main loop (with the chunk's own parity every band store paid a v_bitop3 and every dS store
    // a v_cndmask to select the address at run time).
    auto band_put = [&](const f32x16& v, int par) {      // a chunk of Q.Er^T -> band
        const uint32_t tog = (uint32_t)par << 7;
#pragma unroll
        for (int r = 0; r < 16; ++r) lds_store_f32((wcl[r] ^ tog) + r * BAND_STRIDE, v[r]);
    };
    auto band_get = [&](int par) {                       // Srel^T of a tile
        const char* rb = smem + rbase + (par << 7);
        f32x16 c;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 v = *(const f32x4*)(rb + 32 * g4);
            c[4 * g4] = v.x; c[4 * g4 + 1] = v.y; c[4 * g4 + 2] = v.z; c[4 * g4 + 3] = v.w;
        }
        return c;
    };
    // dband (unrotated, [a][delta&63] bf16): write offsets for D/32 even; odd flips column bit 5
    int dwa0[16], dwa1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dwa0[r] = a * DB_STRIDE + (((a - crow(r, hh)) & 63) << 1);
        dwa1[r] = a * DB_STRIDE + (((a - crow(r, hh) + 32) & 63) << 1);
    }
    {      // first "hi" chunk -> band
        f32x16 qe = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], e[ks], qe);
        band_put(qe, 0);                                 // chunk q0: physical parity 0
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[ks] = e_frag(q0 - 1, ks);    // new chunk of step 0
    }
    f32x16 dq0 = zero16(), dq1 = zero16();
    const int am = a - 4 * hh;                           // key crow(r,hh) is in the future of query a  <=>  crow(r,0) > am

    // ---- one tile, from S^T (band term already in c) to the dq accumulators ---------------------------------------
    // MASKED: apply the diagonal / key-padding masks (general body only)
    auto tile_tail = [&](f32x16& c, int dq, int p, int cur, uint32_t pw, auto masked_tag, const bf16x8 (&et)[4]) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        const char* kt = smem + OFF_KR + cur * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma(frag_R(kt, a, hh, ks), qf[ks], c);
        if (MASKED) {
            if (dq == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (crow(r, 0) > am) ? -INFINITY : c[r];
            }
            if (pw) {
                const uint32_t pwl = pw >> (4 * hh);
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (pwl & (1u << crow(r, 0))) ? -INFINITY : c[r];
            }
        }
        // P^T
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(c[r], LOG2E, -lse2));
        // dP^T = V dO^T
        f32x16 dp = zero16();
        const char* vt = smem + OFF_VR + cur * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp = mfma(frag_R(vt, a, hh, ks), dof[ks], dp);
        // dS^T
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = c[r] * (dp[r] - dlt);
        // dqs^T += K^T dS^T
        bf16x8 df[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            df[ss] = acc_to_frag(c, ss);
            dq0 = mfma(frag_T_onR(kt, lane, ss, 0), df[ss], dq0);
            dq1 = mfma(frag_T_onR(kt, lane, ss, 1), df[ss], dq1);
        }
        // un-skew dS into the (query, delta) band -- the bf16 pairs packed for the product above are stored as their low
        // and high halves (ds_write_b16 / ds_write_b16_d16_hi: no second conversion) --, then the completed chunk feeds dq_rel
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const u32x4 wv = __builtin_bit_cast(u32x4, df[ss]);               // word j: keys 8ss+2j (low), 8ss+2j+1 (high)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r0 = 8 * ss + 2 * j;
                *(uint16_t*)(dband + (p ? dwa1[r0] : dwa0[r0])) = (uint16_t)wv[j];
                *(uint16_t*)(dband + (p ? dwa1[r0 + 1] : dwa0[r0 + 1])) = (uint16_t)(wv[j] >> 16);
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 gq = *(const bf16x8*)(dband + a * DB_STRIDE + (p * 32 + 16 * ks + 8 * hh) * 2);
            dq0 = mfma(et[2 * ks], gq, dq0);
            dq1 = mfma(et[2 * ks + 1], gq, dq1);
        }
    };
    // ---- main loop: tiles strictly below every wave's diagonal, no padded keys: branch-free, two steps per trip so that the
    //      LDS buffers and the band parities of a step are compile-time constants (nmain = Q0 is a multiple of 4) -------------
    const int nmain = anypad ? 0 : Q0;                    // Q0 <= ntw - 1: a next tile always exists inside this loop
    auto main_step = [&](int s, auto par_tag) {
        constexpr int PAR = decltype(par_tag)::value;     // = s & 1: LDS buffer of tile s, physical parity of chunk dq
        const int tn = min(s + 1, ntw - 1);
        const u32x4 kreg = k_tile(tn);
        const u32x4 vreg = v_tile(tn);
        const int dq = q0 - s;                            // >= 1
        // fragments of ErT for chunk dq (used at the end of this step)
        bf16x8 et[4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { et[2 * ks] = et_frag(dq, ks, 0); et[2 * ks + 1] = et_frag(dq, ks, 1); }
        f32x16 c = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma(qf[ks], e[ks], c);
        band_put(c, PAR ^ 1);                             // chunk dq-1
        wave_lds_fence();
        c = band_get(PAR);
        wave_lds_fence();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) e[ks] = e_frag(dq - 2, ks);      // Er chunk of the next step
        __builtin_amdgcn_sched_barrier(0x78F);             // VMEM may not sink below: the fragments are needed at the top of the next step
        tile_tail(c, dq, PAR, PAR, 0u, std::false_type{}, et);
        *(u32x4*)(smem + OFF_KR + (PAR ^ 1) * TILE_BYTES + st_offR) = kreg;
        *(u32x4*)(smem + OFF_VR + (PAR ^ 1) * TILE_BYTES + st_offR) = vreg;
        __syncthreads();
    };
    int s = 0;
    for (; s < nmain; s += 2) {
        main_step(s, std::integral_constant<int, 0>{});
        main_step(s + 1, std::integral_constant<int, 1>{});
    }

    // ---- general body: the diagonal 128 x 128 block (a wave is full / on its diagonal / done), padded keys ------------
    for (; s < ntw; ++s) {
        const int cur = s & 1;
        const int tn = min(s + 1, ntw - 1);
        const u32x4 kreg = k_tile(tn);
        const u32x4 vreg = v_tile(tn);
        const int dq = q0 - s;
        if (dq >= 0) {
            const uint32_t pw = padword(s);
            bf16x8 et[4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) { et[2 * ks] = et_frag(dq, ks, 0); et[2 * ks + 1] = et_frag(dq, ks, 1); }
            if (dq >= 1) {
                f32x16 qe = zero16();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) qe = mfma(qf[ks], e[ks], qe);
                band_put(qe, cur ^ 1);
            }
            wave_lds_fence();
            f32x16 c = band_get(cur);
            wave_lds_fence();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) e[ks] = e_frag(dq - 2, ks);
            tile_tail(c, dq, cur, cur, pw, std::true_type{}, et);
        }
        if (s + 1 < ntw) {
            *(u32x4*)(smem + OFF_KR + (cur ^ 1) * TILE_BYTES + st_offR) = kreg;
            *(u32x4*)(smem + OFF_VR + (cur ^ 1) * TILE_BYTES + st_offR) = vreg;
        }
        __syncthreads();
    }
    if (wave_on) store_rows_lds(dqkv + ((size_t)b * L + i0) * ld + hd * 64, ld, dq0, dq1, lane, 0.125f, smem + band_base);
}

// ================================================================================================
// K1L: dQ from STORED dS.  When dK/dV and dQ are both wanted, the dK/dV kernel runs first and leaves every dS tile in the
// workspace (bf16, its own operand registers); dq = dS (K + Er-band) / 8 then needs no S / Q.Er^T / exp / dP at all: 8 MFMA
// per 32x32 tile instead of 28, fed by a 2 KB tile load.  Same sweep and ownership as K1 (workgroup = 128 query rows, wave =
// 32 rows, key tiles 0..diagonal, K staged through LDS); HBM-bound (the causal half of dS, read once).
//   stored tile: lane (j = lane&31, hh) holds dS[i = crow(8ss+k,hh)][j]: queries on registers, keys on lanes.
//   * dqs^T[c][i] += K^T[c][j] dS^T[j][i]: the tile is parked row-major [j][i] in a wave-private image-T patch and read back
//     with transposing LDS reads as the B operand (k order kappa, matching frag_T_onR of the K tile);
//   * the same registers are scattered into the (query, distance & 63) band exactly as K1 does; the completed chunk is the
//     B operand of dqs^T += ErT . dS_rel^T.
// ================================================================================================
#ifndef MGX_DQL_PEEL
#define MGX_DQL_PEEL 0      // timing experiments only (tools/peel_dq_lite.sh): bits drop parts of the dq_lite step, results are then wrong
#endif                    // 1 dS^T patch stores | 2 three quarters of the band stores | 4 half of dS K | 8 half of dS_rel ErT | 16 K / ErT ring refills
#ifndef MGX_DQL_KT
#define MGX_DQL_KT 1        // K staged as image T (conflict-free transposing reads); 0 (A/B builds): image R, 2-way conflicts (rounds 1-3)
#endif
namespace k1l {
constexpr int OFF_KR = 0;                                  // 2 x 4K  K image (tile t in slot t & 1)
constexpr int OFF_ET = OFF_KR + 2 * TILE_BYTES;            // 8 x 4K  ErT chunk fragments, ring: chunk Q0 - k in slot k & 7
constexpr int XROW = 72;                                   // bytes per row of the dS^T patch: 32 queries + pad (lane-per-row writes and the
                                                           // transposing reads both hit distinct 8-byte bank groups)
constexpr int OFF_X = OFF_ET + 8 * 4096;                   // 4 x 2,304 B  dS^T tile [32 j][32 i]
constexpr int DB_STRIDE = 144;
constexpr int OFF_DBAND = OFF_X + WAVES * 32 * XROW;       // 4 x 4,608 B bf16 [32][72]: dS by (query, delta&63)
constexpr int LDS_BYTES = OFF_DBAND + WAVES * 32 * DB_STRIDE;   // 68,608 B -> 2 workgroups per CU
constexpr int DEPTH = 4;                                   // dS tiles in flight per wave (2 KB each)
}  // namespace k1l

__global__ __launch_bounds__(256, 2) void rel_attn_dq_lite_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ EfT, const uint16_t* __restrict__ dst,
    uint16_t* __restrict__ dqkv, int L, int d, int bgroup) {
    using namespace k1l;
    extern __shared__ __attribute__((aligned(256))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int a = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    const int nqb = (L + 127) >> 7;
    const int b = (blockIdx.y / nqb) * bgroup + blockIdx.x / heads, hd = blockIdx.x % heads;
    const int qb = nqb - 1 - (blockIdx.y % nqb);           // heaviest query blocks first
    const int I0 = qb * 128, Q0 = I0 >> 5;
    const int nchunk = L >> 5;
    const bool wave_on = I0 + w * 32 < L;
    const int q0 = wave_on ? Q0 + w : nchunk - 1;          // a wave beyond the end runs on clamped data and stores nothing
    const int i0 = q0 * 32;
    const int ntw = min(Q0 + 4, nchunk);
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;

    // Per step the workgroup fetches ONE K tile and ONE ErT chunk (4 KB each, 16 bytes per thread) for its four waves -- wave w
    // multiplies with chunk Q0 + w - s, i.e. the chunk wave 0 used w steps earlier -- and each wave its own 2 KB dS tile.  Item k
    // (K tile k / chunk Q0 - k) is requested at the end of step k - 3, parked in registers for two steps, written to LDS at the
    // end of step k - 1 and read from step k on.
    const int srow = tid >> 3, sch = tid & 7;
    // K is only ever read transposed here (A operand of dq^T += K^T dS^T): image T, whose ds_read_b64_tr_b16 are conflict-free (on
    // image R -- the layout the recompute dQ kernel shares with its row reads -- they are 2-way conflicted: 27 M of this kernel's
    // 151 M LDS cycles at cfg2 / batch 64, r03_pmc_attn_b64.json)
    const int st_offR = MGX_DQL_KT ? imgT_off(srow, sch) : imgR_off(srow, sch);
    const char* k_base = (const char*)(qkv_b + d + hd * 64);
    const uint32_t k_voff = (uint32_t)((srow * ld + sch * 8) * 2);
    const uint32_t tile_bytes = (uint32_t)(32 * ld * 2);
    auto k_tile = [&](int t) { return *(const u32x4*)(k_base + (size_t)min(t, ntw - 1) * tile_bytes + k_voff); };
    const uint32_t tid16 = (uint32_t)tid * 16u, lane16 = (uint32_t)lane * 16u;
    auto e_item = [&](int k) {                             // this thread's 16 bytes of chunk Q0 - k (fragment-ordered copy of ErT)
        return *(const u32x4*)((const char*)EfT + (size_t)min(max(Q0 - k, 0), nchunk - 1) * 4096 + tid16);
    };
    const size_t ntri = (size_t)nchunk * (nchunk + 1) / 2;
    const size_t row_tiles = ((size_t)b * heads + hd) * ntri + (size_t)q0 * (q0 + 1) / 2;      // tile (b,h, I = q0, 0)
    const char* ds_row = (const char*)(dst + row_tiles * 1024);
    // read once: streamed past L2 (K / ErT stay); index clamped to the wave's diagonal, clamped tiles are never used
    auto ds_load = [&](int J, int ss) {
        return __builtin_nontemporal_load((const u32x4*)(ds_row + (size_t)min(J, q0) * 2048 + ss * 1024 + lane16));
    };

    u32x4 dsr[DEPTH][2];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) { dsr[j][0] = ds_load(j, 0); dsr[j][1] = ds_load(j, 1); }
    *(u32x4*)(smem + OFF_KR + st_offR) = k_tile(0);
#pragma unroll
    for (int k = -3; k <= 0; ++k) *(u32x4*)(smem + OFF_ET + (k & 7) * 4096 + tid16) = e_item(k);
    u32x4 kq[2] = {k_tile(1), k_tile(2)}, eq[2] = {e_item(1), e_item(2)};
    for (int o = tid * 16; o < WAVES * 32 * DB_STRIDE; o += 256 * 16) *(u32x4*)(smem + OFF_DBAND + o) = u32x4{0, 0, 0, 0};
    __syncthreads();

    char* xt = smem + OFF_X + w * (32 * XROW);
    char* dband = smem + OFF_DBAND + w * (32 * DB_STRIDE);
    const int xw0 = a * XROW + 8 * hh;                    // + 16 * (2ss + jq): the 8-byte piece (ss, jq) of this lane's row
    const int xi = lane & 15, xg = lane >> 4;
    const int xr0 = ((xi >> 2) + 4 * hh) * XROW + 32 * (xg & 1) + 8 * (xi & 3);     // transposing read, + (16s + 8jq) * XROW
    int dwa0[16], dwa1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        dwa0[r] = crow(r, hh) * DB_STRIDE + (((crow(r, hh) - a) & 63) << 1);
        dwa1[r] = crow(r, hh) * DB_STRIDE + (((crow(r, hh) - a + 32) & 63) << 1);
    }
    f32x16 dq0 = zero16(), dq1 = zero16();

    // dS^T patch -> B operand (k order kappa): X[16s + 8jq + 4hh + rq][lane&31]
    auto frag_X = [&](int s) {
        bf16x8 out;
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            const bf16x4 t4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(xt + xr0 + (16 * s + 8 * jq) * XROW));
            out[4 * jq + 0] = t4[0]; out[4 * jq + 1] = t4[1]; out[4 * jq + 2] = t4[2]; out[4 * jq + 3] = t4[3];
        }
        return out;
    };
    auto compute = [&](int p, const char* kt, const char* ec, const u32x4 (&t)[2]) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            if (!(MGX_DQL_PEEL & 1)) {
                *(u32x2*)(xt + xw0 + 16 * (2 * ss)) = u32x2{t[ss].x, t[ss].y};
                *(u32x2*)(xt + xw0 + 16 * (2 * ss + 1)) = u32x2{t[ss].z, t[ss].w};
            }
#pragma unroll
            for (int j = 0; j < ((MGX_DQL_PEEL & 2) ? 1 : 4); ++j) {
                const int r0 = 8 * ss + 2 * j;
                *(uint16_t*)(dband + (p ? dwa1[r0] : dwa0[r0])) = (uint16_t)t[ss][j];
                *(uint16_t*)(dband + (p ? dwa1[r0 + 1] : dwa0[r0 + 1])) = (uint16_t)(t[ss][j] >> 16);
            }
        }
        wave_lds_fence();
#pragma unroll
        for (int ss = 0; ss < ((MGX_DQL_PEEL & 4) ? 1 : 2); ++ss) {
            const bf16x8 df = frag_X(ss);
            dq0 = mfma(MGX_DQL_KT ? frag_T(kt, lane, ss, 0) : frag_T_onR(kt, lane, ss, 0), df, dq0);
            dq1 = mfma(MGX_DQL_KT ? frag_T(kt, lane, ss, 1) : frag_T_onR(kt, lane, ss, 1), df, dq1);
        }
#pragma unroll
        for (int ks = 0; ks < ((MGX_DQL_PEEL & 8) ? 1 : 2); ++ks) {
            const bf16x8 gq = *(const bf16x8*)(dband + a * DB_STRIDE + (p * 32 + 16 * ks + 8 * hh) * 2);
            dq0 = mfma(*(const bf16x8*)(ec + (2 * ks) * 1024 + lane16), gq, dq0);
            dq1 = mfma(*(const bf16x8*)(ec + (2 * ks + 1) * 1024 + lane16), gq, dq1);
        }
        wave_lds_fence();                                  // the patch and the band half are rewritten by the next step
    };
    // one step: SLOT = s & 3 (registers of the dS tile), PAR = s & 1 (K slot, physical parity of the completed chunk)
    auto step = [&](int s, auto slot_tag, bool active) {
        constexpr int SLOT = decltype(slot_tag)::value, PAR = SLOT & 1;
        const u32x4 t[2] = {dsr[SLOT][0], dsr[SLOT][1]};
        dsr[SLOT][0] = ds_load(s + DEPTH, 0);
        dsr[SLOT][1] = ds_load(s + DEPTH, 1);
        if (active) compute(PAR, smem + OFF_KR + PAR * TILE_BYTES, smem + OFF_ET + ((s - w) & 7) * 4096, t);
        // items s+1 (requested two steps ago): the K slot was last read in step s-1, the chunk slot in step s-4
        if (!(MGX_DQL_PEEL & 16)) {
            *(u32x4*)(smem + OFF_KR + (PAR ^ 1) * TILE_BYTES + st_offR) = kq[PAR];
            *(u32x4*)(smem + OFF_ET + ((s + 1) & 7) * 4096 + tid16) = eq[PAR];
            // the freed registers take items s+3 (no register rotation: a move of a register with a load in flight is a wait)
            kq[PAR] = k_tile(s + 3);
            eq[PAR] = e_item(s + 3);
        }
        __syncthreads();
    };
    using S0_ = std::integral_constant<int, 0>; using S1_ = std::integral_constant<int, 1>;
    using S2_ = std::integral_constant<int, 2>; using S3_ = std::integral_constant<int, 3>;
    static_assert(DEPTH == 4, "the loops below are unrolled by DEPTH");
    // entered with loads in flight the loop gets an s_waitcnt vmcnt(0) at its top (the compiler merges the unknown entry state
    // into every trip): drain once here, the loop then keeps its own four steps of requests outstanding
    __builtin_amdgcn_s_waitcnt(0x0F70);
    int s = 0;
    for (; s < Q0; s += 4) {                               // tiles strictly below every wave's diagonal (Q0 is a multiple of 4)
        step(s, S0_{}, true);
        step(s + 1, S1_{}, true);
        step(s + 2, S2_{}, true);
        step(s + 3, S3_{}, true);
    }
    // the diagonal 128 x 128 block (s = Q0 here): a wave is full / on its diagonal / done
    step(s, S0_{}, q0 - s >= 0);
    if (s + 1 < ntw) step(s + 1, S1_{}, q0 - s - 1 >= 0);
    if (s + 2 < ntw) step(s + 2, S2_{}, q0 - s - 2 >= 0);
    if (s + 3 < ntw) step(s + 3, S3_{}, q0 - s - 3 >= 0);
    if (wave_on) store_rows_lds(dqkv + ((size_t)b * L + i0) * ld + hd * 64, ld, dq0, dq1, lane, 0.125f, dband);
}

// ================================================================================================
// K2: dK, dV.  workgroup = 128 keys (wave = 32 keys, K/V row fragments in registers), sweeps query
// tiles i0 = J0, J0+32, ...  orientation: queries on registers, keys on lanes (S, P, dP, dS);
// accumulators dK^T[c][b], dV^T[c][b].
// ================================================================================================
#ifndef MGX_DKV_PEEL
#define MGX_DKV_PEEL 0      // timing experiments only (tools/peel_dkv.sh): 1 no E loads in the sweep | 2 no dS stores | 4 no skew (bpermute)
#endif                      // | 8 no exponentials | 16 no q / dO tile prefetch+publish (the first tile is reused) | 32 no lse / delta reads
                            // (constants); results are then wrong
// MGX_DKV_STAMP (diagnostic build only, `_build.py --variant dkvstamp -DMGX_DKV_STAMP`; tools/dkv_stamp.py): s_memtime stamps at five
// points of a main-loop step; lane 0 of every wave leaves its sums in its first dk row (the results are then garbage).  Reading
// a stamp waits for lgkmcnt(0), i.e. for the wave's outstanding LDS operations: the stamped kernel is a little slower.
#ifdef MGX_DKV_STAMP
#define DKV_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                          __builtin_amdgcn_sched_barrier(0); if (!MASKED) st_acc[i] += t_ - st_last; st_last = t_; } while (0)
#else
#define DKV_STAMP(i)
#endif
// (round 4 experiment, removed: log2(e)/8 folded into K and into a scaled copy of the Er fragments, -lse log2(e) as the initial
//  accumulator of the Q.Er^T products, so that S arrives as the exponent's argument -- 16 fewer VALU per tile: 1.258 ms against
//  1.262 at cfg2 / batch 64, nothing; and the backward's P would no longer equal the forward's bit for bit.)
#ifndef MGX_DKV_IMGB
#define MGX_DKV_IMGB 0      // 1 (A/B builds): the one image of q / dO is image B (rel_attn_common.hpp: conflict-free for the row AND the transposed
#endif                      //    reads; the 37 M conflict cycles of the kernel's 211 M LDS cycles are gone): 1.262-1.267 against 1.255-1.259 ms with
                            //    image R and its 2-way conflicted transposed reads -- the conflicts cost nothing, the longer swizzle a little
#ifndef MGX_DKV_ONEIMG
#define MGX_DKV_ONEIMG 1    // 1: q and dO staged as ONE LDS image each (image R), the transposed fragments read from it with 2-way bank conflicts --
#endif                      //    two DMA instructions fewer per wave and tile: 1.227 against 1.252 ms at cfg2 / batch 64 (a VMEM instruction costs the
                            //    issuing wave ~50 cycles here, tools/dkv_stamp.py).  0 (A/B builds): separate images R and T, conflict-free reads
#ifndef MGX_DKV_WAVES
#define MGX_DKV_WAVES 4     // waves (= 32-key tiles) per workgroup of the dK/dV kernel: 4 (128 keys) or 2 (64 keys, A/B builds: a shorter diagonal
#endif                      // block and twice the workgroups, but every wave stages twice as much: 1.315 against 1.246 ms at cfg2 / batch 64)
#if 0
#endif
namespace k2 {
constexpr int KW = MGX_DKV_WAVES;
constexpr int OFF_QR = 0;                                  // 2 x 4K  qs image R
constexpr int OFF_QT = OFF_QR + 2 * TILE_BYTES;            // 2 x 4K  qs image T
constexpr int OFF_OR = OFF_QT + 2 * TILE_BYTES;            // 2 x 4K  dO image R
constexpr int OFF_OT = OFF_OR + 2 * TILE_BYTES;            // 2 x 4K  dO image T
constexpr int ST_BYTES = 256 * KW;                         // per buffer: KW waves x (-lse2[32], -delta[32]): every wave stages and reads its own copy
constexpr int OFF_ST = OFF_OT + 2 * TILE_BYTES;            // 2 x 1 KB
constexpr int PATCH_BYTES = 4608;                          // per wave: 32 rows x 144 B, the epilogue's row-major store patch
constexpr int OFF_BAND = KW == 4 ? OFF_ST + 2 * ST_BYTES : 0;      // (64-key workgroups: the patches reuse the image buffers after the sweep)
constexpr int OFF_FLAG = KW == 4 ? OFF_BAND + KW * PATCH_BYTES : OFF_ST + 2 * ST_BYTES;   // "a key of this workgroup is padded" flag
constexpr int LDS_BYTES = OFF_FLAG + 16;                   // 51,728 B (the 256 VGPRs limit the kernel to 2 waves per SIMD)
// The Er chunks (B operand of Q.Er^T: column t = lane&31, 16 contiguous bytes of row L-1-32q-t) are
// loaded straight from global/L2 into registers, one new chunk per step (the previous "hi" chunk is
// the next "lo" chunk), so E needs no LDS here.
}  // namespace k2

template <bool EXPORT_DS>     // always true (one instantiation): as a plain function hipcc builds a 36 % longer main loop from the same source
__global__ __launch_bounds__(64 * k2::KW, 2) void rel_attn_dkv_kernel(
    const uint16_t* __restrict__ qkv, const u32x4* __restrict__ EfA, const uint32_t* __restrict__ padbits,
    const uint16_t* __restrict__ dctx, const float* __restrict__ nlse2 /* -lse log2(e) */, const float* __restrict__ ndelta /* -delta */,
    uint16_t* __restrict__ dqkv, uint16_t* __restrict__ dst, int L, int d, int bgroup) {
    using namespace k2;
    extern __shared__ __attribute__((aligned(256))) char smem[];     // 256: the band reads XOR bit 7 of absolute LDS addresses
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bl = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    const int nkb = (L + 32 * KW - 1) / (32 * KW);         // y = (batch group, key block): groups as in the dQ kernel
#ifndef MGX_DKV_XCD
#define MGX_DKV_XCD 0       // 1 (A/B builds): the (b,h) of a batch group dealt to the XCDs, see below
#endif
    int bh_l = blockIdx.x, kbr = blockIdx.y % nkb;         // (b,h) inside the batch group, key-block rank (0 = longest sweep, dispatched first)
    if (MGX_DKV_XCD && (gridDim.x & 7) == 0) {
        // workgroups r and r + 8 share an XCD (MI355X_MICROARCH.md): XCD x = r & 7 works through the (b,h) with bh % 8 == x only, key
        // blocks in the same heaviest-first order, so the 16 key-block workgroups of a (b,h) -- which all read its q / dO rows --
        // share one L2 (8 (b,h) x 2 MB of q + dO per XCD at cfg2)
        const int r = kbr * gridDim.x + bh_l, per = gridDim.x >> 3, m = r >> 3;
        kbr = m / per;
        bh_l = (m % per) * 8 + (r & 7);
    }
    const int b = (blockIdx.y / nkb) * bgroup + bh_l / heads, hd = bh_l % heads;
    const int J0 = kbr * 32 * KW;
    const int nchunk = L >> 5;
    const int nT = (L - J0) >> 5;                          // query tiles i0 = J0 + 32 t
    const bool wave_on = J0 + w * 32 < L;
    // a wave beyond the end of the sequence (L % 128 != 0) shadows the last valid key block and stores nothing
    const int wk = wave_on ? w : nT - 1;                   // the wave's key tile inside the workgroup; D/32 = t - wk
    const int j0 = J0 + wk * 32;
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;
    const size_t stat_base = ((size_t)b * heads + hd) * L;

    // Staging of a query tile (q and dO, two LDS images each, and the two statistics of its 32 rows) is LDS-DMA: thread tid owns
    // the 16-byte slot tid of every image -- row tid >> 3, PHYSICAL chunk tid & 7 -- and fetches the logical chunk the image's
    // swizzle puts there (rel_attn_common.hpp: dma16), so a tile costs a wave five DMA instructions and neither registers nor
    // ds_write (round 3: 2 loads into registers, then 5 stores).  Every global address of the sweep is (wave-uniform base in
    // SGPRs) + (32-bit per-lane offset).
    // (a workgroup of KW waves covers the 256 slots of an image in NS = 4 / KW rounds: slot = tid + 64 KW i)
    constexpr int NS = 4 / KW;
    const char* q_base = (const char*)(qkv_b + (size_t)J0 * ld + hd * 64);                      // + t * 32 rows
    const char* o_base = (const char*)(dctx + ((size_t)b * L + J0) * d + hd * 64);
    uint32_t q_voffR[NS], q_voffT[NS], o_voffR[NS], o_voffT[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        const int slot = tid + 64 * KW * i, srow = slot >> 3, spc = slot & 7;
        const int lcR = spc ^ (MGX_DKV_IMGB ? imgB_swz(srow) : ((srow >> 1) & 7)), lcT = spc ^ (((srow >> 1) & 1) << 2);   // logical chunks: imgR_off (imgB_off) / imgT_off inverted
        q_voffR[i] = (uint32_t)((srow * ld + lcR * 8) * 2); q_voffT[i] = (uint32_t)((srow * ld + lcT * 8) * 2);
        o_voffR[i] = (uint32_t)((srow * d + lcR * 8) * 2);  o_voffT[i] = (uint32_t)((srow * d + lcT * 8) * 2);
    }
    const uint32_t q_step = (uint32_t)(32 * ld * 2), o_step = (uint32_t)(32 * d * 2);
    // fragment ks of Er chunk q for this lane (fragment-ordered copy: 1 KB contiguous per wave load).  Every load of the
    // sweep is unconditional with a clamped index; data of clamped tiles / chunks is never used.
    const uint32_t lane16 = (uint32_t)lane * 16u;
    auto e_frag = [&](int q, int ks) {
        return __builtin_bit_cast(bf16x8, *(const u32x4*)((const char*)EfA + (size_t)min(max(q, 0), nchunk - 1) * 4096 + ks * 1024 + lane16));
    };
    // -lse log2(e) (lanes 0..31) / -delta (lanes 32..63) of row (lane & 31) of a query tile, from the pre-pass's copies, in the form
    // the kernel consumes them -- the addend of the exponent's fma and the INITIAL ACCUMULATOR of dP = dO V^T (rows = queries) --,
    // so the DMA needs no arithmetic on the way.  Every wave stages (and reads) its own 256-byte copy: no statistic crosses waves.
    const uint32_t st_voff = (uint32_t)(((lane & 32) ? (const char*)ndelta - (const char*)nlse2 : 0) + (lane & 31) * 4);   // |offset| < 2^31: same allocation
    const char* st_base = (const char*)(nlse2 + stat_base + J0);
    const uint32_t lds_w = lds_addr_of(smem) + w * 1024;   // this wave's 1 KB of every 4 KB image; + OFF_ST: its 256 B of statistics
    auto stage = [&](int t, int buf) {                     // tile t (clamped) -> LDS buffers `buf`
        const int tn = (MGX_DKV_PEEL & 16) ? 0 : min(t, nT - 1);
        const char* qb = q_base + (size_t)tn * q_step;
        const char* ob = o_base + (size_t)tn * o_step;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const uint32_t dw = lds_w + KW * 1024 * i + buf * TILE_BYTES;       // slots 64 (w + KW i) .. + 63
            dma16(qb, q_voffR[i], dw + OFF_QR);
            if (!MGX_DKV_ONEIMG) dma16(qb, q_voffT[i], dw + OFF_QT);
            dma16(ob, o_voffR[i], dw + OFF_OR);
            if (!MGX_DKV_ONEIMG) dma16(ob, o_voffT[i], dw + OFF_OT);
        }
        dma4(st_base + (size_t)tn * 128, st_voff, lds_addr_of(smem) + OFF_ST + buf * ST_BYTES + w * 256);
    };
    stage(0, 0);
    // E chunk fragments: a step's "hi" chunk (t - wk) sits in e[PAR], the "lo" chunk (t - wk - 1) in e[PAR^1]; the slot of
    // the lo chunk receives chunk t - wk + 1 once it has been used, which is the next step's hi chunk.  The main loop
    // alternates PAR = 0, 1 (two steps per trip); the general body always uses PAR = 0 and swaps the slots afterwards.
    bf16x8 kf[4], vf[4], e[2][4];
    uint32_t padlane = 0;
    int wgpad = 0;
    {
        const uint16_t* kp = qkv_b + (size_t)(j0 + bl) * ld + d + hd * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(kp + ks * 16));
            vf[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(kp + d + ks * 16));
            e[0][ks] = e_frag(0, ks);                       // the wave's first step (t = wk) is its diagonal: hi chunk 0
            e[1][ks] = e[0][ks];
        }
        if (padbits) {
            const uint32_t pwv = padbits[(size_t)b * nchunk + (j0 >> 5)];
            padlane = (pwv >> bl) & 1u;
            // any padded key in this workgroup's 128 keys?  (no __syncthreads_or: it allocates static LDS)
            if (tid == 0) *(volatile uint32_t*)(smem + OFF_FLAG) = 0u;
            __syncthreads();
            if (pwv) *(volatile uint32_t*)(smem + OFF_FLAG) = 1u;
            __syncthreads();
            wgpad = __builtin_amdgcn_readfirstlane(*(volatile uint32_t*)(smem + OFF_FLAG));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile 0 has landed (a DMA has no register the compiler could wait on)
    __syncthreads();
    char* band = smem + OFF_BAND + w * PATCH_BYTES;          // the epilogue's store patch
    // rd[r] = byte address (source lane * 4) of the ds_bpermute that skews register r (see `tile`)
    uint32_t rd[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rd[r] = (uint32_t)((hh * 32 + ((crow(r, hh) - bl) & 31)) << 2);
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
    // Every dS tile goes to the workspace as the operand registers this wave multiplies with q (bf16, the dK
    // product's own rounding): tile (b,h, I = query tile, J = key tile <= I) is 2 KB at ((bh*T + I(I+1)/2 + J)*1024 elements,
    // T = nchunk(nchunk+1)/2 (causal half); inside a tile unit (ss, lane) = 16 bytes at ss*512 + lane*8 elements holds
    // dS[i = crow(8ss+k, hh)][j = lane&31], k = 0..7 -- one wave store instruction writes 1 KB contiguously.  The dQ kernel
    // (dq_lite) and the dE kernel read these tiles instead of recomputing S / P / dP.
    // (wave-uniform base: tile (b,h, I = 0, J = j0/32); a wave beyond the end of the sequence rewrites the last key block's
    // tiles with identical data)
    char* ds_col = nullptr;
    if (EXPORT_DS) {
        const size_t ntri = (size_t)nchunk * (nchunk + 1) / 2;
        ds_col = (char*)(dst + (((size_t)b * heads + hd) * ntri + (size_t)(j0 >> 5)) * 1024) + lane16;
    }
    auto ds_tile = [&](int t) {                           // query tile I = J0/32 + t
        const size_t I = (size_t)(J0 >> 5) + t;
        return ds_col + (I * (I + 1) / 2) * 2048;
    };

    // ---- one query tile.  cur = t & 1 (LDS buffers), PAR = E slot of the hi chunk; MASKED: diagonal / padded-key masks ----
#ifdef MGX_DKV_STAMP
    unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_last = 0;
    unsigned st_steps = 0;
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();   // realtime: constant 100 MHz
#endif
    auto tile = [&](int dq, int cur, auto par_tag, auto masked_tag, char* dsp, int tnext) {
        constexpr int PAR = decltype(par_tag)::value;
        constexpr bool MASKED = decltype(masked_tag)::value;
        DKV_STAMP(5);                                     // [5] from the previous stamp (after the barrier) to here: prefetch issue
        const char* qr = smem + OFF_QR + cur * TILE_BYTES;
        bf16x8 qa[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qa[ks] = MGX_DKV_IMGB ? frag_B(qr, bl, hh, ks) : frag_R(qr, bl, hh, ks);
        // Q.Er^T for chunks dq ("hi": keys bl <= query, t = a - bl) and dq-1 ("lo": keys bl > query, t = 32 + a - bl); rows =
        // query a, columns = t.  A tile reads column (a - bl) & 31 of row a and needs the hi chunk there for t <= a and the lo
        // chunk for t > a: the two products are MERGED in registers (one v_cndmask per element) and stored once -- 16 band
        // stores per tile instead of 32, a 4 KB band per wave instead of an 8 KB ring, and no parity in any address (this
        // kernel computes both chunks for every tile anyway: unlike the forward / dQ kernels nothing is reused by the next tile).
        const char* st = smem + OFF_ST + cur * ST_BYTES + w * 256;
        f32x16 nl;                                        // -lse2 of the accumulator's query rows
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 l4 = (MGX_DKV_PEEL & 32) ? f32x4{-9.f, -9.f, -9.f, -9.f} : *(const f32x4*)(st + (8 * g4 + 4 * hh) * 4);
            nl[4 * g4] = l4.x; nl[4 * g4 + 1] = l4.y; nl[4 * g4 + 2] = l4.z; nl[4 * g4 + 3] = l4.w;
        }
        f32x16 qe = zero16();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qe = mfma(qa[ks], e[PAR][ks], qe);
        if (!MASKED || dq >= 1) {
            f32x16 ql = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ql = mfma(qa[ks], e[PAR ^ 1][ks], ql);
#pragma unroll
            for (int r = 0; r < 16; ++r) qe[r] = (bl <= crow(r, hh)) ? qe[r] : ql[r];
        }
        // the lo slot is free now: fetch the next step's hi chunk into it
        if (!(MGX_DKV_PEEL & 1)) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) e[PAR ^ 1][ks] = e_frag(dq + 1, ks);
        }
        // ... and request the next query tile into the other LDS buffers (every wave is past the barrier that ended their last
        // use).  AFTER the E loads: the compiler's wait for those at the top of the next step counts the operations it knows to
        // be younger (the two dS stores) and so also covers these five -- which have landed by then anyway (see `landed`)
        stage(tnext, cur ^ 1);
        if (!MASKED) __builtin_amdgcn_sched_barrier(0x78F);    // VMEM may not sink below: needed at the top of the next step
        DKV_STAMP(0);                                     // [0] q fragments, 8 Q.Er^T MFMAs, merge
        // The skew is a LANE permutation inside each half-wave: the tile's element (row a = crow(r,hh), key bl) is the merged value
        // merged[a][t = (a - bl) & 31], which lane t of the same half holds in the SAME register r -- one ds_bpermute_b32 per
        // register and no LDS memory (until round 3 the merged tile went through a 4 KB band: 16 stores + 16 loads per tile).
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = qe[r];            // (a __builtin_bit_cast of the vector ELEMENT expression itself reads element 0)
            c[r] = (MGX_DKV_PEEL & 4) ? v : __int_as_float(__builtin_amdgcn_ds_bpermute((int)rd[r], __float_as_int(v)));
        }
        DKV_STAMP(1);                                     // [1] 16 ds_bpermute and their results
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) c = mfma(qa[ks], kf[ks], c);
        if (MASKED) {
            if (dq == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = (bl > crow(r, hh)) ? -INFINITY : c[r];
            }
            if (padlane) {
#pragma unroll
                for (int r = 0; r < 16; ++r) c[r] = -INFINITY;
            }
        }
        f32x16 dp;                                        // initial accumulator: -delta of the accumulator's query rows
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 d4 = (MGX_DKV_PEEL & 32) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(st + 128 + (8 * g4 + 4 * hh) * 4);
            dp[4 * g4] = d4.x; dp[4 * g4 + 1] = d4.y; dp[4 * g4 + 2] = d4.z; dp[4 * g4 + 3] = d4.w;
        }
        const char* orr = smem + OFF_OR + cur * TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) dp = mfma(MGX_DKV_IMGB ? frag_B(orr, bl, hh, ks) : frag_R(orr, bl, hh, ks), vf[ks], dp);
        f32x16 ds;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // c = 8 S: q is staged unscaled, 1/8 (exact) rides in this multiplier and in the final scale of dK
            const float p = (MGX_DKV_PEEL & 8) ? c[r] * 1e-9f : __builtin_amdgcn_exp2f(__builtin_fmaf(c[r], 0.125f * LOG2E, nl[r]));
            c[r] = p;
            ds[r] = p * dp[r];
        }
        DKV_STAMP(2);                                     // [2] S, dP MFMAs, statistics, exponentials, dS
        const char* ot = smem + OFF_OT + cur * TILE_BYTES;
        const char* qt = smem + OFF_QT + cur * TILE_BYTES;
        u32x4 dfx[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 pf = acc_to_frag(c, ss);
            const bf16x8 df = acc_to_frag(ds, ss);
            dv0 = mfma(MGX_DKV_ONEIMG ? (MGX_DKV_IMGB ? frag_T_onB(orr, lane, ss, 0) : frag_T_onR(orr, lane, ss, 0)) : frag_T(ot, lane, ss, 0), pf, dv0);
            dv1 = mfma(MGX_DKV_ONEIMG ? (MGX_DKV_IMGB ? frag_T_onB(orr, lane, ss, 1) : frag_T_onR(orr, lane, ss, 1)) : frag_T(ot, lane, ss, 1), pf, dv1);
            dk0 = mfma(MGX_DKV_ONEIMG ? (MGX_DKV_IMGB ? frag_T_onB(qr, lane, ss, 0) : frag_T_onR(qr, lane, ss, 0)) : frag_T(qt, lane, ss, 0), df, dk0);
            dk1 = mfma(MGX_DKV_ONEIMG ? (MGX_DKV_IMGB ? frag_T_onB(qr, lane, ss, 1) : frag_T_onR(qr, lane, ss, 1)) : frag_T(qt, lane, ss, 1), df, dk1);
            dfx[ss] = __builtin_bit_cast(u32x4, df);
        }
        // streamed (read back from HBM by two later kernels): costs this kernel 55-100 us of its 600 at cfg2 (tools/peel_dkv.sh);
        // issuing them before the dV / dK products instead of after changes nothing
        if (EXPORT_DS && !(MGX_DKV_PEEL & 2)) {           // == DS_STORES below (`landed`)
            __builtin_nontemporal_store(dfx[0], (u32x4*)dsp);
            __builtin_nontemporal_store(dfx[1], (u32x4*)(dsp + 1024));
        }
        DKV_STAMP(3);                                     // [3] packs, transposed fragments, 8 dV / dK MFMAs (issue), dS stores
    };
    // The next query tile's DMA (issued inside `tile`, after the E loads) must have landed before the barrier that ends the
    // step; the only VMEM operations a wave issues after it are the two dS stores of its tile: a COUNTED wait, vmcnt(2).
    // (the count follows the condition under which the stores are compiled: a peel build without them must wait for vmcnt(0))
    constexpr bool DS_STORES = EXPORT_DS && !(MGX_DKV_PEEL & 2);
    auto landed = [&]() {
        if (DS_STORES) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // ---- general body: the diagonal 128 x 128 block (t < 4: a wave is not started / on its diagonal / full), every
    //      step when a key of this workgroup is padded, and an odd last step ------------------------------------------------
    auto general_step = [&](int t) {
        const int dq = t - wk;
        if (dq >= 0) {
            tile(dq, t & 1, std::integral_constant<int, 0>{}, std::true_type{}, ds_tile(t), t + 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) { const bf16x8 x = e[0][ks]; e[0][ks] = e[1][ks]; e[1][ks] = x; }
            landed();                                     // counted, as in the main loop: the two dS stores may stay in flight
        } else {
            stage(t + 1, (t & 1) ^ 1);                    // (a wave that has a tile stages from inside it)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // a wave that skipped its tile issued nothing after the DMA
        }
        __syncthreads();
    };
    // (round 4: the four steps of the diagonal block specialised at compile time -- the wave's first tile through the masked body,
    //  later ones through the main loop's branch-free body with its E-slot protocol, instead of the general body below -- made the
    //  kernel SLOWER, 1.31 against 1.265 ms at cfg2 / batch 64: eight more inlined tile bodies, 68 spilled registers outside the main
    //  loop and a 15 K-instruction kernel.  The general body costs 4.2-5.0 K cycles per step against 3.1 K in the main loop,
    //  15 % of a workgroup's time: tools/dkv_stamp.py.)
    int t = 0;
    const int nhead = wgpad ? nT : min(KW, nT);           // (KW is even: the main loop starts on an even step)
#ifdef MGX_DKV_STAMP
    const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();      // end of the prologue
#endif
    for (; t < nhead; ++t) general_step(t);
#ifdef MGX_DKV_STAMP
    const unsigned long long st_t2 = __builtin_amdgcn_s_memtime();      // end of the diagonal block's general steps
#endif
    // ---- main loop (t >= 4 is even here): every wave's tile is full, no masks: branch-free bodies, two steps per trip so
    //      that the LDS buffer and the E slot of each step are compile-time constants -----------------------------------------
    for (; t + 1 < nT; t += 2) {
        tile(t - wk, 0, std::integral_constant<int, 0>{}, std::false_type{}, ds_tile(t), t + 1);
        landed();
        __syncthreads();
#ifdef MGX_DKV_STAMP
        { constexpr bool MASKED = false; DKV_STAMP(4); st_steps += 2; }     // [4] publish + barrier
#endif
        tile(t + 1 - wk, 1, std::integral_constant<int, 1>{}, std::false_type{}, ds_tile(t + 1), t + 2);
        landed();
        __syncthreads();
#ifdef MGX_DKV_STAMP
        { constexpr bool MASKED = false; DKV_STAMP(4); }
#endif
    }
#ifdef MGX_DKV_STAMP
    const unsigned long long st_t3 = __builtin_amdgcn_s_memtime();      // end of the main loop
#endif
    for (; t < nT; ++t) general_step(t);

    if (KW != 4) __syncthreads();                         // the patches reuse the image buffers: every wave is done reading them
    if (wave_on) {
        uint16_t* row0 = dqkv + ((size_t)b * L + j0) * ld + hd * 64;
        store_rows_lds(row0 + d, ld, dk0, dk1, lane, 0.125f, band);      // dk = dS^T (q/8)
        store_rows_lds(row0 + 2 * d, ld, dv0, dv1, lane, 1.f, band);
#ifdef MGX_DKV_STAMP
        if (lane == 0) {
            float* rec = (float*)(row0 + d);
            for (int i = 0; i < 6; ++i) rec[i] = (float)st_acc[i];
            rec[6] = (float)st_steps; rec[7] = (float)(J0 >> 7); rec[8] = (float)w;
            rec[9] = (float)(__builtin_amdgcn_s_memtime() - st_t0); rec[10] = (float)(__builtin_amdgcn_s_memrealtime() - st_r0);
            rec[11] = (float)(st_t1 - st_t0); rec[12] = (float)(st_t2 - st_t1); rec[13] = (float)(st_t3 - st_t2);
        }
#endif
    }
}

// ================================================================================================
// K3: dE.  workgroup = 8 consecutive chunks of 32 relative distances (wave = chunk c, Er chunk
// fragments + dEr[32][64] accumulators in registers).  For query tile i0 the band of chunk c
// covers the lower triangle (b<=a) of key tile u = i0/32 - c and the upper triangle (b>a) of key
// tile u-1: both tiles are computed and merged element-wise before exp/dS.
// ================================================================================================
namespace k3 {
constexpr int W3 = 8;                                      // waves (= distance chunks) per workgroup
constexpr int KV_SLOTS = 10;                               // live key tiles [t-8, t] + the incoming one
constexpr int OFF_KR = 0;                                  // 10 x 4K K image R ring (slot = tile % 10)
constexpr int OFF_VR = OFF_KR + KV_SLOTS * TILE_BYTES;     // 10 x 4K V image R ring
constexpr int OFF_QR = OFF_VR + KV_SLOTS * TILE_BYTES;     // 2 x 4K qs image R
constexpr int OFF_QT = OFF_QR + 2 * TILE_BYTES;            // 2 x 4K qs image T
constexpr int OFF_OR = OFF_QT + 2 * TILE_BYTES;            // 2 x 4K dO image R
constexpr int OFF_ST = OFF_OR + 2 * TILE_BYTES;            // 2 x 256 B
constexpr int OFF_BAND = OFF_ST + 2 * 256;                 // 8 x 4K fp32 [32][32] (QE, then dS)
constexpr int LDS_BYTES = OFF_BAND + W3 * 4096;            // 139,776 B: one 8-wave workgroup per CU
}  // namespace k3

__global__ __launch_bounds__(512, 2) void rel_attn_de_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ Er, const uint32_t* __restrict__ padbits,
    const uint16_t* __restrict__ dctx, const float* __restrict__ lse, const float* __restrict__ delta,
    float* __restrict__ dEr /* = dE + (M-L)*64 */, int L, int d) {
    using namespace k3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bl = lane & 31, hh = lane >> 5;
    const int heads = d >> 6;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int nchunk = L >> 5;
    const int C0 = blockIdx.y * W3;                        // small C0 = longest sweep = dispatched first
    const int cw = C0 + w;
    const bool wave_on = cw < nchunk;
    const int nT = nchunk - C0;
    const size_t ld = (size_t)3 * d;
    const uint16_t* qkv_b = qkv + (size_t)b * L * ld;
    const size_t stat_base = ((size_t)b * heads + hd) * L;
    const uint32_t* pb = padbits ? padbits + (size_t)b * nchunk : nullptr;

    // staging roles: threads 0..255 stage the K and qs tiles, threads 256..511 the V and dO tiles
    const int half = tid >> 8;
    const int srow = (tid & 255) >> 3, sch = tid & 7;
    const int st_offR = imgR_off(srow, sch), st_offT = imgT_off(srow, sch);
    const uint16_t* kg = qkv_b + (size_t)srow * ld + d + half * d + hd * 64 + sch * 8;           // K or V, + u*32*ld
    const uint16_t* qg = qkv_b + (size_t)(32 * C0 + srow) * ld + hd * 64 + sch * 8;              // + t*32*ld
    const uint16_t* og = dctx + ((size_t)b * L + 32 * C0 + srow) * d + hd * 64 + sch * 8;        // + t*32*d
    auto stat_src = [&](int t) {
        const int i = 32 * (C0 + t) + (tid & 31);
        return (tid < 32) ? lse[stat_base + i] * LOG2E : delta[stat_base + i];
    };
    {
        *(u32x4*)(smem + (half ? OFF_VR : OFF_KR) + st_offR) = *(const u32x4*)kg;      // key tile 0 -> slot 0
        if (half == 0) {
            const u32x4 qq = scale8(*(const u32x4*)qg, 0.125f);
            *(u32x4*)(smem + OFF_QR + st_offR) = qq;
            *(u32x4*)(smem + OFF_QT + st_offT) = qq;
        } else {
            *(u32x4*)(smem + OFF_OR + st_offR) = *(const u32x4*)og;
        }
        if (tid < 64) *(float*)(smem + OFF_ST + tid * 4) = stat_src(0);
    }
    bf16x8 ef[4];
    if (wave_on) {
        const uint16_t* ep = Er + (size_t)(L - 1 - 32 * cw - bl) * 64 + hh * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ef[ks] = __builtin_bit_cast(bf16x8, *(const u32x4*)(ep + ks * 16));
    }
    __syncthreads();
    char* band = smem + OFF_BAND + w * 4096;
    f32x16 de0 = zero16(), de1 = zero16();

    for (int t = 0; t < nT; ++t) {
        const int cur = t & 1;
        u32x4 kreg, qreg;      // K (or V) tile and qs (or dO) tile of the next step, by staging half
        float streg = 0.f;
        const bool have_next = (t + 1 < nT);
        if (have_next) {
            kreg = *(const u32x4*)(kg + (size_t)(t + 1) * 32 * ld);      // key tile t+1 <= nT-1 < nchunk
            qreg = half ? *(const u32x4*)(og + (size_t)(t + 1) * 32 * d) : *(const u32x4*)(qg + (size_t)(t + 1) * 32 * ld);
            if (tid < 64) streg = stat_src(t + 1);
        }
        const int u = t - w;                              // lower key tile; upper = u-1
        if (wave_on && u >= 0) {
            const char* qr = smem + OFF_QR + cur * TILE_BYTES;
            bf16x8 qa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qa[ks] = frag_R(qr, bl, hh, ks);
            f32x16 qe = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) qe = mfma(qa[ks], ef[ks], qe);
#pragma unroll
            for (int r = 0; r < 16; ++r) *(float*)(band + (crow(r, hh) * 32 + bl) * 4) = qe[r];
            wave_lds_fence();
            f32x16 srel;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ar = crow(r, hh);
                srel[r] = *(const float*)(band + (ar * 32 + ((ar - bl) & 31)) * 4);
            }
            const bool has_up = (u >= 1);
            const char* klo = smem + OFF_KR + (u % KV_SLOTS) * TILE_BYTES;
            const char* vlo = smem + OFF_VR + (u % KV_SLOTS) * TILE_BYTES;
            const char* kup = smem + OFF_KR + ((u + KV_SLOTS - 1) % KV_SLOTS) * TILE_BYTES;
            const char* vup = smem + OFF_VR + ((u + KV_SLOTS - 1) % KV_SLOTS) * TILE_BYTES;
            f32x16 slo = srel, sup = srel;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) slo = mfma(qa[ks], frag_R(klo, bl, hh, ks), slo);
            if (has_up) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) sup = mfma(qa[ks], frag_R(kup, bl, hh, ks), sup);
            }
            bool plo = false, pup = !has_up;              // "masked" flags of this lane's key in each tile
            if (pb) {
                plo = (pb[u] >> bl) & 1u;
                if (has_up) pup = (pb[u - 1] >> bl) & 1u;
            }
            const char* orr = smem + OFF_OR + cur * TILE_BYTES;
            bf16x8 oa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) oa[ks] = frag_R(orr, bl, hh, ks);
            f32x16 dlo = zero16(), dup = zero16();
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) dlo = mfma(oa[ks], frag_R(vlo, bl, hh, ks), dlo);
            if (has_up) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) dup = mfma(oa[ks], frag_R(vup, bl, hh, ks), dup);
            }
            const char* st = smem + OFF_ST + cur * 256;
            f32x16 ds;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 l4 = (MGX_DKV_PEEL & 32) ? f32x4{9.f, 9.f, 9.f, 9.f} : *(const f32x4*)(st + (8 * g4 + 4 * hh) * 4);
                const f32x4 d4 = (MGX_DKV_PEEL & 32) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(st + 128 + (8 * g4 + 4 * hh) * 4);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r = 4 * g4 + k;
                    const bool lower = (bl <= crow(r, hh));
                    const bool masked = lower ? plo : pup;
                    const float sv = lower ? slo[r] : sup[r];
                    const float dv = lower ? dlo[r] : dup[r];
                    const float p = masked ? 0.f : __builtin_amdgcn_exp2f(__builtin_fmaf(sv, LOG2E, -l4[k]));
                    ds[r] = p * (dv - d4[k]);
                }
            }
            // un-skew: dQE[a][t] = dS[a][b] with t = (a-b)&31, through the same band buffer
            wave_lds_fence();
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ar = crow(r, hh);
                *(float*)(band + (ar * 32 + ((ar - bl) & 31)) * 4) = ds[r];
            }
            wave_lds_fence();
            f32x16 x;
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = *(const float*)(band + (crow(r, hh) * 32 + bl) * 4);
            wave_lds_fence();
            const char* qt = smem + OFF_QT + cur * TILE_BYTES;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const bf16x8 xf = acc_to_frag(x, ss);
                de0 = mfma(xf, frag_T(qt, lane, ss, 0), de0);
                de1 = mfma(xf, frag_T(qt, lane, ss, 1), de1);
            }
        }
        if (have_next) {
            *(u32x4*)(smem + (half ? OFF_VR : OFF_KR) + ((t + 1) % KV_SLOTS) * TILE_BYTES + st_offR) = kreg;
            if (half == 0) {
                const u32x4 qq = scale8(qreg, 0.125f);
                *(u32x4*)(smem + OFF_QR + (cur ^ 1) * TILE_BYTES + st_offR) = qq;
                *(u32x4*)(smem + OFF_QT + (cur ^ 1) * TILE_BYTES + st_offT) = qq;
            } else {
                *(u32x4*)(smem + OFF_OR + (cur ^ 1) * TILE_BYTES + st_offR) = qreg;
            }
            if (tid < 64) *(float*)(smem + OFF_ST + (cur ^ 1) * 256 + tid * 4) = streg;
        }
        __syncthreads();
    }
    // flush: dEr[delta = 32*cw + t][cc] += de[t][cc]; Er row index = L-1-delta.  One register of the
    // accumulator = two 128-byte row segments per wave instruction (full-rate atomic shape).
    if (wave_on) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dl = 32 * cw + crow(r, hh);
            float* row = dEr + (size_t)(L - 1 - dl) * 64;
            atomicAdd(row + bl, de0[r]);
            atomicAdd(row + 32 + bl, de1[r]);
        }
    }
}

// ================================================================================================
// K3t: dE from the dS tiles the dK/dV kernel stored (by query tile x key tile, NOT yet un-skewed):
//     dEr[delta][c] = 1/8 sum_{b,h} sum_{i >= delta} dS[b,h][i][i-delta] q[b,i,h,c]
// A workgroup owns four tile DIAGONALS I - J = c0 + m, m = 0..3, for a fixed number of 64-row steps of the flattened
// (b, h, i-block) sweep, so every stored tile is read exactly once by this kernel.  Tile m holds the distances
// 32(c0+m) - 31 .. + 31: the four diagonals touch FIVE chunks of 32 distances, c0-1 .. c0+3 (the first and the last only
// through one triangle of their tiles; the neighbouring workgroups add the other triangles -- dE is summed with atomics
// anyway).  The skew happens while a step's tiles are parked in LDS: element (i, j) of tile m goes to row i, column
// 32(m+1) + i - j of a [64 rows][160 distances] image -- every write lands inside the row (no predication, no wrap);
// positions no tile writes are zeroed once and stay zero.  Wave w (of 5) multiplies columns 32w .. 32w+31 = chunk c0-1+w.
// ================================================================================================
#ifndef MGX_DET_PEEL
#define MGX_DET_PEEL 0      // timing experiments only (tools/peel_de_tiles.sh): 1 three quarters of the scatter stores | 2 no products |
#endif                      // 4 q tile re-read from row block 0 (L2-resident) | 8 no atomic flush of the chunk sums; results are then wrong
namespace k3t {
#ifndef MGX_DET_STEPS
#define MGX_DET_STEPS 32
#endif
constexpr int DIAGS = 4, RS = 64, STEPS = MGX_DET_STEPS, NW = 5;
constexpr int AROW = 352;                                  // bytes per image row: 160 bf16 + pad (4 consecutive rows -> 4 bank groups)
constexpr int OFF_A = 0;                                   // [64 i][160 distance columns]
constexpr int OFF_Q = RS * AROW;                           // q tile [64 i][64 c]: 2 sub-tiles image T
constexpr int LDS_BYTES = OFF_Q + 2 * TILE_BYTES;          // 30,720 B
constexpr int SLOTS = 2;                                   // tiles per wave and step: 8 tiles on waves 0..3
}  // namespace k3t

__global__ __launch_bounds__(320, 4) void rel_attn_de_tiles_kernel(
    const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dst, float* __restrict__ dEr /* = dE + (M-L)*64 */,
    int bgroup, int ngroups, int wg_per_group, int L, int d,
    long long* __restrict__ det /* deterministic mode: [L][64] fixed-point image of this launch's dEr */) {
    using namespace k3t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int heads = d >> 6, nbh = bgroup * heads;
    const size_t ld = (size_t)3 * d;
    const int nchunk = L >> 5;
    // workgroup -> (batch group, diagonal group t, slice of its flattened (bh, row-block) sweep); inside a batch group the
    // diagonal groups are laid out longest sweep first
    int t = 0, first = 0, ns = 0;
    // Batch groups are dealt to the XCDs (workgroups b and b + 8 share one, MI355X_MICROARCH.md): workgroup 8 k + x belongs to group
    // 8 (k / wg_per_group) + x.  A group's workgroups -- all the diagonal groups that re-read the same q rows -- then share an L2,
    // and with groups of ONE batch row (2 MB of q at cfg2) the re-reads are L2 hits instead of fabric reads.  Speed only.
    // (ngroups % 8 == 0 on this path; otherwise -- e.g. cfg4's four batch rows per GPU -- the host passes ngroups < 0 and the
    // groups are laid out one after the other: a round of eight with idle XCDs would leave part of the chip without work)
    const bool dealt = ngroups > 0;
    const int xk = dealt ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int grp = dealt ? (xk / wg_per_group) * 8 + (int)(blockIdx.x & 7) : xk / wg_per_group;
    {
        int rest = xk % wg_per_group;
        const int ntile = (nchunk + DIAGS - 1) / DIAGS;
        for (t = 0; t < ntile; ++t) {
            ns = (L - t * DIAGS * 32 + RS - 1) / RS;       // row blocks i0 = 32 c0, +64, ... < L (query tiles I >= c0)
            const int nwg = (nbh * ns + STEPS - 1) / STEPS;
            if (rest < nwg) break;
            rest -= nwg;
        }
        if (t == ntile) return;
        first = rest * STEPS;
    }
    const int total = nbh * ns;
    const int last = min(total, first + STEPS);
    const int c0 = t * DIAGS, d0 = c0 * 32;
    const size_t ntri = (size_t)nchunk * (nchunk + 1) / 2;
    const int qrow = tid >> 3, qch = tid & 7;              // threads 0..255 stage q
    const uint32_t lane16 = (uint32_t)lane * 16u;
    // slot k of wave w < 4 = tile idx = w + 4k of the step: query tile rb = idx / 4 (of 2), diagonal m = idx % 4 = w
    const int wm = w & 3;
    // scatter address of register r (query crow(r,hh), key l31): row crow*AROW, column 32(m+1) + crow - l31
    const int sc_lane = hh * 4 * (AROW + 2) - 2 * l31;      // + crow(r,0) * (AROW + 2) as the immediate
    u32x4 areg[SLOTS][2], qreg[2];
    bool a_ok[SLOTS], q_ok[2];
    auto load_tiles = [&](int g) {
        const int bhl = g / ns, i0 = d0 + (g - bhl * ns) * RS;
        const int bh = grp * nbh + bhl;
        const int bb = bh / heads, hd = bh - bb * heads;
        const int I0 = i0 >> 5;
        const char* tp = (const char*)(dst + (size_t)bh * ntri * 1024) + lane16;
#pragma unroll
        for (int k = 0; k < SLOTS; ++k) {                  // slot k = query tile I0 + k; wave 4 loads (clamped) data it never uses
            const int I = I0 + k, J = I - c0 - wm;
            a_ok[k] = I < nchunk && J >= 0;
            const size_t Ic = (size_t)min(I, nchunk - 1), Jc = (size_t)max(J, 0);      // clamped: a valid address either way
            const char* p = tp + (Ic * (Ic + 1) / 2 + min(Jc, Ic)) * 2048;
            areg[k][0] = __builtin_nontemporal_load((const u32x4*)p);
            areg[k][1] = __builtin_nontemporal_load((const u32x4*)(p + 1024));
        }
        const uint16_t* qp = qkv + ((size_t)bb * L + ((MGX_DET_PEEL & 4) ? 0 : i0)) * ld + hd * 64 + (qch & 7) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = (qrow & 31) + 32 * i;
            q_ok[i] = i0 + r < L;
            qreg[i] = *(const u32x4*)(qp + (size_t)min(r, L - 1 - i0) * ld);
        }
    };
    auto store_tiles = [&]() {
        const u32x4 zero = {0, 0, 0, 0};
        if (w < 4) {                                       // wave-uniform
#pragma unroll
            for (int k = 0; k < SLOTS; ++k) {
                char* base = smem + OFF_A + k * 32 * AROW + 64 * (wm + 1) + sc_lane;
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) {
                    const u32x4 v = a_ok[k] ? areg[k][ss] : zero;
#pragma unroll
                    for (int j = 0; j < ((MGX_DET_PEEL & 1) ? 1 : 4); ++j) {
                        const int r0 = 8 * ss + 2 * j;
                        *(uint16_t*)(base + crow(r0, 0) * (AROW + 2)) = (uint16_t)v[j];
                        *(uint16_t*)(base + crow(r0 + 1, 0) * (AROW + 2)) = (uint16_t)(v[j] >> 16);
                    }
                }
            }
            char* qt = smem + OFF_Q;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = qrow + 32 * i;
                *(u32x4*)(qt + (r >> 5) * TILE_BYTES + imgT_off(r & 31, qch)) = q_ok[i] ? qreg[i] : zero;
            }
        }
    };
    f32x16 de0 = zero16(), de1 = zero16();
    // A fragment: A[m = distance column 32w + (lane&31)][k = i = 16ks + 8hh + j] from the [i][distance] image (transposing reads)
    const int fa_i = lane & 15, fa_g = lane >> 4;
    const int fa_off = (fa_i >> 2) * AROW + (32 * w) * 2 + (2 * (fa_g & 1) + ((fa_i & 3) >> 1)) * 16 + 8 * (fa_i & 1) + 8 * hh * AROW;
    auto multiply = [&]() {
        if (MGX_DET_PEEL & 2) return;
        const char* at = smem + OFF_A + fa_off;
        const char* qt = smem + OFF_Q;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af;
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                const bf16x4 tq = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(at + (16 * ks + 4 * jq) * AROW));
                af[4 * jq + 0] = tq[0]; af[4 * jq + 1] = tq[1]; af[4 * jq + 2] = tq[2]; af[4 * jq + 3] = tq[3];
            }
            const char* qs = qt + (ks >> 1) * TILE_BYTES;
            de0 = mfma(af, frag_Tn(qs, lane, ks & 1, 0), de0);
            de1 = mfma(af, frag_Tn(qs, lane, ks & 1, 1), de1);
        }
    };
    // image positions that no tile writes (the triangles that belong to the neighbouring diagonal groups) must read as zero
    for (int o = tid * 16; o < RS * AROW; o += NW * 64 * 16) *(u32x4*)(smem + OFF_A + o) = u32x4{0, 0, 0, 0};
    __syncthreads();
    if (first < last) {
        load_tiles(first);
        store_tiles();
    }
    __syncthreads();
    for (int g = first; g + 1 < last; ++g) {
        load_tiles(g + 1);
        __builtin_amdgcn_sched_barrier(0);              // keep the prefetch ahead of the products
        multiply();
        __syncthreads();                                // every wave has read the image
        store_tiles();
        __syncthreads();
    }
    if (first < last) multiply();
    // flush: rows = distances 32(c0-1+w) + crow(r,hh), columns on lanes; q was not pre-scaled -> 1/8 here
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int dl = d0 - 32 + 32 * w + crow(r, hh);
        if ((MGX_DET_PEEL & 8) && de0[r] + de1[r] != 12345.f) continue;      // peel: no flush (timing only)
        if (dl >= 0 && dl < L) {
            if (det) {
                long long* drow = det + (size_t)(L - 1 - dl) * 64;
                det_add(drow + l31, 0.125f * de0[r]);
                det_add(drow + 32 + l31, 0.125f * de1[r]);
                continue;
            }
            float* row = dEr + (size_t)(L - 1 - dl) * 64;
            atomicAdd(row + l31, 0.125f * de0[r]);
            atomicAdd(row + 32 + l31, 0.125f * de1[r]);
        }
    }
}

// batch rows per grid group: the largest divisor of B whose q/k/v/dO/ctx bytes stay near 100 MB (rel_attn_fwd.hip)
static int bwd_batch_group(int B, int L, int d) {
    const double per_row = (double)L * d * 2 * 5;
    int g = B;
#if MGX_EXPERIMENTS
    static const int forced = [] { const char* e = getenv("MGX_ATTN_BGROUP"); return e ? atoi(e) : 0; }();   // experiment knob
    if (forced > 0 && B % forced == 0) return forced;
#endif
    while (g > 1 && (g * per_row > 110e6 || B % g != 0)) --g;
    return g;
}

static size_t ws_stat_bytes(int B, int L, int d) { return (((size_t)B * (d / 64) * L * 4) + 255) / 256 * 256; }      // one f32 [B,h,L]
static size_t ws_delta_bytes(int B, int L, int d) { return 3 * ws_stat_bytes(B, L, d); }    // delta | -lse log2e | -delta

static size_t ws_ert_bytes(int L) { return 2 * er_frag_bytes(L); }   // EfA | EfT

extern "C" size_t mgx_rel_attn_bwd_workspace(int B, int L, int d) {
    if (B <= 0 || L <= 0 || d <= 0) return 0;
    // delta, -lse log2e, -delta f32 [B,h,L] each | fragment-ordered Er (EfA, EfT) | causal half of dS by (query tile, key tile) bf16:
    // B*h*T tiles of 2 KB
    const size_t nchunk = (size_t)L / 32;
    return ws_delta_bytes(B, L, d) + ws_ert_bytes(L) + (size_t)B * (d / 64) * (nchunk * (nchunk + 1) / 2) * 2048;
}

// parts: 1 pre-pass (delta, E re-layout) | 4 dK/dV (stores the dS tiles) | 2 dQ from the stored tiles | 8 dE from the stored
// tiles | 16 dE by recomputation | 32 dQ by recomputation (16, 32: independent of the stored tiles; cross-checks / A-B
// timing) | 64 dK/dV by the 32-key kernel whatever the shape (instead of 4; cross-check of the 64-key kernel).  Launch order inside
// one call: 1, 4 (or 64), 2 (or 32), 8, 16.
extern "C" int mgx_rel_attn_bwd_parts(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits,
                                      const uint16_t* ctx, const uint16_t* dctx, const float* lse, uint16_t* dqkv,
                                      float* dE, void* workspace, size_t ws_bytes, int B, int L, int d, int M,
                                      int parts, void* stream) {
    MGX_REQUIRE(qkv && E && ctx && dctx && lse && dqkv && dE && workspace, MGX_ERR_NULL, "mgx_rel_attn_bwd: NULL pointer");
    MGX_REQUIRE(B > 0 && L > 0 && d > 0 && d % 64 == 0 && L % 32 == 0 && M >= L, MGX_ERR_SHAPE,
                "mgx_rel_attn_bwd: need d%%64==0, L%%32==0, M>=L (got B=%d L=%d d=%d M=%d)", B, L, d, M);
    MGX_REQUIRE(ws_bytes >= mgx_rel_attn_bwd_workspace(B, L, d) && ((uintptr_t)workspace & 255) == 0, MGX_ERR_SHAPE,
                "mgx_rel_attn_bwd: workspace must be 256-byte aligned and >= mgx_rel_attn_bwd_workspace() = %zu bytes (got %zu)",
                mgx_rel_attn_bwd_workspace(B, L, d), ws_bytes);
    MGX_REQUIRE(!((parts & 2) && (parts & 32)), MGX_ERR_SHAPE, "mgx_rel_attn_bwd: parts 2 and 32 both write dq");
    static const bool attr_once = [] {                  // thread-safe one-time init (C++11 function-local static)
        hipFuncSetAttribute((const void*)rel_attn_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, k1::LDS_BYTES);
        hipFuncSetAttribute((const void*)rel_attn_dq_lite_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, k1l::LDS_BYTES);
        hipFuncSetAttribute((const void*)rel_attn_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, k2::LDS_BYTES);
        hipFuncSetAttribute((const void*)rel_attn_de_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, k3::LDS_BYTES);
        hipFuncSetAttribute((const void*)rel_attn_de_tiles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, k3t::LDS_BYTES);
        return true;
    }();
    (void)attr_once;
    hipStream_t s = (hipStream_t)stream;
    const int heads = d / 64;
    const uint16_t* Er = E + (size_t)(M - L) * 64;
    float* delta = (float*)workspace;
    float* nlse2 = (float*)((char*)workspace + ws_stat_bytes(B, L, d));
    float* ndelta = (float*)((char*)workspace + 2 * ws_stat_bytes(B, L, d));
    u32x4* EfA = (u32x4*)((char*)workspace + ws_delta_bytes(B, L, d));
    u32x4* EfT = (u32x4*)((char*)EfA + er_frag_bytes(L));
    uint16_t* dst = (uint16_t*)((char*)EfA + ws_ert_bytes(L));
    if (parts & 1) {
        // delta = rowsum(dO o O) for the kernels that form dS (dK/dV and the two recompute cross-checks)
        const long total = (long)B * L * heads * 8;
        hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, ctx, dctx, lse, delta, nlse2, ndelta, B, L, d);
        launch_er_frag(Er, EfA, EfT, L, s);
    }
    const int bg = bwd_batch_group(B, L, d);
    MGX_REQUIRE((long)((L + 127) / 128) * (B / bg) <= 65535, MGX_ERR_SHAPE, "mgx_rel_attn_bwd: L/128 * batch groups too large");
    const dim3 gq(bg * heads, ((L + 127) / 128) * (B / bg));
    MGX_REQUIRE(!((parts & 4) && (parts & 64)), MGX_ERR_SHAPE, "mgx_rel_attn_bwd: parts 4 and 64 both write dk / dv and the dS tiles");
#ifndef MGX_DKV64_DEFAULT
#define MGX_DKV64_DEFAULT 1   // 0 (A/B builds): bit 2 launches the 32-key kernel for every shape
#endif
    // bit 2: the 64-keys-per-wave kernel with the hand-scheduled main loop (rel_attn_dkv64.hip) where the sequence is whole 128-key
    // blocks, the 32-key kernel otherwise; bit 6: the 32-key kernel whatever the shape (cross-check: both give the same bits)
    if ((parts & 4) && MGX_DKV64_DEFAULT && L % 128 == 0) {
        const int rc64 = dkv64_launch(qkv, EfA, padbits, dctx, nlse2, ndelta, dqkv, dst, B, L, d, bg, stream);
        if (rc64 != MGX_OK) return rc64;
    } else if (parts & (4 | 64)) {
#if MGX_EXPERIMENTS
        static const int dkv_lds = [] {     // experiment: MGX_DKV_LDS pads the dynamic LDS to lower the residency (timing only)
            const char* e = getenv("MGX_DKV_LDS");
            const int v = e ? atoi(e) : 0;
            if (v > k2::LDS_BYTES) hipFuncSetAttribute((const void*)rel_attn_dkv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, v);
            return v > k2::LDS_BYTES ? v : k2::LDS_BYTES;
        }();
#else
        constexpr int dkv_lds = k2::LDS_BYTES;
#endif
        const dim3 gk(bg * heads, ((L + 32 * k2::KW - 1) / (32 * k2::KW)) * (B / bg));
        MGX_REQUIRE(gk.y <= 65535, MGX_ERR_SHAPE, "mgx_rel_attn_bwd: too many key blocks for the grid");
        hipLaunchKernelGGL(rel_attn_dkv_kernel<true>, gk, dim3(64 * k2::KW), dkv_lds, s, qkv, EfA, padbits, dctx, nlse2, ndelta, dqkv, dst, L, d, bg);
    }
    if (parts & 2)
        hipLaunchKernelGGL(rel_attn_dq_lite_kernel, gq, dim3(256), k1l::LDS_BYTES, s, qkv, EfT, dst, dqkv, L, d, bg);
    if (parts & 32)
        hipLaunchKernelGGL(rel_attn_dq_kernel, gq, dim3(256), k1::LDS_BYTES, s, qkv, EfA, EfT, padbits, dctx, lse, delta, dqkv, L, d, bg);
    if (parts & 8) {
#ifndef MGX_DET_BG
#define MGX_DET_BG 1        // batch rows per group of the dE kernel (A/B builds: the attention kernels' bgroup is 8 at cfg2 / batch 64)
#endif
        const bool dealt = B % (8 * MGX_DET_BG) == 0;   // whole rounds of eight groups of MGX_DET_BG rows: one group per XCD
        const int bgd = dealt ? MGX_DET_BG : bg, ngr = B / bgd;
        long nwg = 0;                                   // workgroups of ONE batch group
        for (int t = 0; t < (L / 32 + k3t::DIAGS - 1) / k3t::DIAGS; ++t)
            nwg += ((long)bgd * heads * ((L - t * k3t::DIAGS * 32 + k3t::RS - 1) / k3t::RS) + k3t::STEPS - 1) / k3t::STEPS;
        const long grid = nwg * ngr;
        MGX_REQUIRE(grid < (1L << 31), MGX_ERR_SHAPE, "mgx_rel_attn_bwd: grid too large");
        int rc;
        long long* det = mgx_det_scratch((size_t)L * 64, stream, &rc);      // deterministic mode: integer atomics + fold
        if (rc != MGX_OK) return rc;
        hipLaunchKernelGGL(rel_attn_de_tiles_kernel, dim3((unsigned)grid), dim3(64 * k3t::NW), k3t::LDS_BYTES, s, qkv, dst,
                           dE + (size_t)(M - L) * 64, bgd, dealt ? ngr : -ngr, (int)nwg, L, d, det);
        if (det) launch_det_fold(det, dE + (size_t)(M - L) * 64, (size_t)L * 64, 1.f, 1, s);
    }
    if (parts & 16) {
        const dim3 ge(B * heads, ((L >> 5) + k3::W3 - 1) / k3::W3);
        hipLaunchKernelGGL(rel_attn_de_kernel, ge, dim3(64 * k3::W3), k3::LDS_BYTES, s, qkv, Er, padbits, dctx, lse, delta,
                           dE + (size_t)(M - L) * 64, L, d);
    }
    MGX_CHECK_LAUNCH("mgx_rel_attn_bwd");
    return MGX_OK;
}

extern "C" int mgx_rel_attn_bwd(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits, const uint16_t* ctx,
                                const uint16_t* dctx, const float* lse, uint16_t* dqkv, float* dE, void* workspace,
                                size_t ws_bytes, int B, int L, int d, int M, void* stream) {
    return mgx_rel_attn_bwd_parts(qkv, E, padbits, ctx, dctx, lse, dqkv, dE, workspace, ws_bytes, B, L, d, M, 15, stream);
}
