// Autoregressive decode path (replaces the O(W^2)-per-token recompute of network.py:52-77).
//
// One generated token per step and sequence.  Everything the step needs is device resident -- the
// current position is read from a device counter -- so a whole step (all layers + sampling) can be
// captured once in a hipGraph and replayed per token.
//
//   mgx_decode_embed      h = emb[tok]*sqrt(d) + PE[pos]                       (layers.py:226-229)
//   mgx_rel_attn_decode   append k_t, v_t to the cache, then for the single query q_t:
//                         logit_j = (q.k_j + q.E[M-1-(t-j)])/8, j <= t; softmax; ctx = sum p_j v_j
//                         -- HBM-bound: streams K and V of the (b,h) once (2*(t+1)*128 B), E from L2.
//   mgx_sample_topk_topp  temperature -> softmax -> top-k -> top-p -> categorical draw, one wave per row;
//                         top_k = 0 and top_p = 1 reproduce the reference's full-softmax categorical
//                         (network.py:73-74).  Thresholds are exact (bisection on the float bit pattern).
#include "mgx_common.hpp"

namespace {
constexpr float LOG2E = 1.4426950408889634f;
}

__global__ __launch_bounds__(256) void decode_embed_kernel(const int32_t* __restrict__ tok, const float* __restrict__ table,
                                                           const float* __restrict__ pe, const int32_t* __restrict__ pos_dev,
                                                           uint16_t* __restrict__ out, int B, int d, int V, float scale) {
    const int gpr = d >> 3;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * gpr) return;
    const int r = g / gpr, c = (g % gpr) * 8;
    int t = tok[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    const int pos = pos_dev[0];
    const f32x4* tp = (const f32x4*)(table + (size_t)t * d + c);
    const f32x4* pp = (const f32x4*)(pe + (size_t)pos * d + c);
    const f32x4 a0 = tp[0], a1 = tp[1], p0 = pp[0], p1 = pp[1];
    float f[8] = {a0.x * scale + p0.x, a0.y * scale + p0.y, a0.z * scale + p0.z, a0.w * scale + p0.w,
                  a1.x * scale + p1.x, a1.y * scale + p1.y, a1.z * scale + p1.z, a1.w * scale + p1.w};
    *(u32x4*)(out + (size_t)r * d + c) = pack8(f);
}

// One workgroup (8 waves) per (batch, head, key split).  lane = (key slot ks = lane>>3, dim group dg = lane&7): a wave
// handles 8 keys per iteration, each lane 8 of the 64 dims (16-byte loads: a key row is one 128-byte line).
// Every (wave, key slot) runs its own online softmax stream; the 64 streams are merged at the end.
// Split-K: with one workgroup per (b,h) a batch-32, 8-head decode step has 256 workgroups = 8 waves per CU, and the
// bytes those waves keep in flight bound the cache stream at ~4.3 TB/s.  NSPLIT workgroups per (b,h) each take a
// contiguous, 64-key aligned share of the keys 0..t (balanced from the device-side position), write an unnormalised
// partial (m, l, acc[64]) and a tiny second kernel merges them by their maxima.  The new row t is taken from qkv_new by
// whichever split covers it (another workgroup appends it to the cache in the same launch).
constexpr int DEC_WAVES = 8;
constexpr int DEC_PART = 68;                                   // floats per partial: acc[64], m, l, 2 pad (16-byte aligned rows)
#ifndef MGX_DEC_NT
#define MGX_DEC_NT 1
#endif
__global__ __launch_bounds__(64 * DEC_WAVES) void rel_attn_decode_kernel(
    const uint16_t* __restrict__ qkv_new, uint16_t* __restrict__ kcache, uint16_t* __restrict__ vcache,
    const uint16_t* __restrict__ E, const int32_t* __restrict__ pos_dev, uint16_t* __restrict__ ctx,
    float* __restrict__ partial, int Lmax, int d, int M) {
    const int heads = d >> 6;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads;
    const int nsplit = gridDim.y, sp = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ks = lane >> 3, dg = lane & 7;
    const int t = pos_dev[0];                                   // current position; keys 0..t
    const uint16_t* qrow = qkv_new + (size_t)b * 3 * d + hd * 64;
    // caches are head-major [B, h, Lmax, 64]: a workgroup streams one contiguous run of 128-byte rows
    uint16_t* kc = kcache + ((size_t)b * heads + hd) * Lmax * 64;
    uint16_t* vc = vcache + ((size_t)b * heads + hd) * Lmax * 64;
    // append this step's key/value for the following steps (split 0 of each head owns the head's 64 columns); in THIS
    // step row t is read from qkv_new, so no workgroup depends on another one's store
    if (sp == 0) {
        if (tid < 8) *(u32x4*)(kc + (size_t)t * 64 + tid * 8) = *(const u32x4*)(qrow + d + tid * 8);
        else if (tid < 16) *(u32x4*)(vc + (size_t)t * 64 + (tid - 8) * 8) = *(const u32x4*)(qrow + 2 * d + (tid - 8) * 8);
    }
    float q[8];
    unpack8(*(const u32x4*)(qrow + dg * 8), q);
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] *= 0.125f * LOG2E;         // logits in log2 units
    // this split's keys: [lo, hi), shares of ceil((t+1)/nsplit) keys rounded up to the workgroup's 64-key stride
    const int share = ((t + nsplit) / nsplit + 63) & ~63;
    const int lo = sp * share, hi = min(t + 1, lo + share);

    float m = -INFINITY, l = 0.f, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint16_t* Eb = E + (size_t)(M - 1 - t) * 64;           // E row of key j is Eb + j*64
    for (int j0 = lo + (w * 8); j0 < hi; j0 += DEC_WAVES * 8) {
        const int j = j0 + ks;
        const bool valid = j < hi;
        const int jc = valid ? j : hi - 1;
        const uint16_t* kp = (jc == t) ? qrow + d : kc + (size_t)jc * 64;
        const uint16_t* vp = (jc == t) ? qrow + 2 * d : vc + (size_t)jc * 64;
        float kf[8], ef[8], vf[8];
        // the K / V caches are streamed once per token (3.2 GB per step at cfg5's end): nontemporal; E rows are shared by every (b, h): cached
        unpack8(MGX_DEC_NT ? __builtin_nontemporal_load((const u32x4*)(kp + dg * 8)) : *(const u32x4*)(kp + dg * 8), kf);
        unpack8(*(const u32x4*)(Eb + (size_t)jc * 64 + dg * 8), ef);
        unpack8(MGX_DEC_NT ? __builtin_nontemporal_load((const u32x4*)(vp + dg * 8)) : *(const u32x4*)(vp + dg * 8), vf);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += q[k] * (kf[k] + ef[k]);
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (!valid) s = -INFINITY;
        const float mn = fmaxf(m, s);
        const float alpha = (mn == -INFINITY) ? 1.f : __builtin_amdgcn_exp2f(m - mn);
        const float p = (mn == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(s - mn);
        l = l * alpha + p;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] * alpha + p * vf[k];
        m = mn;
    }
    // merge the 8 key slots of the wave (lanes with equal dg), then the 8 waves through LDS
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        const float mo = __shfl_xor(m, o, 64), lo2 = __shfl_xor(l, o, 64);
        const float mn = fmaxf(m, mo);
        const float a0 = (m == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(m - mn);
        const float a1 = (mo == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(mo - mn);
        l = l * a0 + lo2 * a1;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = acc[k] * a0 + __shfl_xor(acc[k], o, 64) * a1;
        m = mn;
    }
    __shared__ float sm[DEC_WAVES], sl[DEC_WAVES], sacc[DEC_WAVES][64];
    if (ks == 0) {
        if (dg == 0) { sm[w] = m; sl[w] = l; }
#pragma unroll
        for (int k = 0; k < 8; ++k) sacc[w][dg * 8 + k] = acc[k];
    }
    __syncthreads();
    if (tid < 64) {
        float mm = -INFINITY;
#pragma unroll
        for (int i = 0; i < DEC_WAVES; ++i) mm = fmaxf(mm, sm[i]);
        float ll = 0.f, o = 0.f;
#pragma unroll
        for (int i = 0; i < DEC_WAVES; ++i) {
            const float a = (sm[i] == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(sm[i] - mm);
            ll += sl[i] * a;
            o += sacc[i][tid] * a;
        }
        if (nsplit == 1) {
            ctx[(size_t)b * d + hd * 64 + tid] = f32_to_bf16(o / ll);
        } else {                                                // an empty split leaves (m = -inf, l = 0, acc = 0)
            float* pp = partial + ((size_t)blockIdx.x * nsplit + sp) * DEC_PART;
            if (tid == 0) { pp[64] = mm; pp[65] = ll; }
            pp[tid] = o;
        }
    }
}

// ctx[b, hd*64 + c] = sum_s acc_s[c] 2^(m_s - m) / sum_s l_s 2^(m_s - m): one wave per (b,h)
__global__ __launch_bounds__(64) void rel_attn_decode_merge_kernel(const float* __restrict__ partial, uint16_t* __restrict__ ctx,
                                                                   int nsplit, int d) {
    const int heads = d >> 6;
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, c = threadIdx.x;
    const float* pp = partial + (size_t)blockIdx.x * nsplit * DEC_PART;
    float mm = -INFINITY;
    for (int s = 0; s < nsplit; ++s) mm = fmaxf(mm, pp[s * DEC_PART + 64]);
    float ll = 0.f, o = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = pp[s * DEC_PART + 64];
        const float a = (ms == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(ms - mm);
        ll += pp[s * DEC_PART + 65] * a;
        o += pp[s * DEC_PART + c] * a;
    }
    ctx[(size_t)b * d + hd * 64 + c] = f32_to_bf16(o / ll);
}

// ---------------------------------------------------------------------------------------------------
// sampling: one wave per row, V <= 64 * 16
// ---------------------------------------------------------------------------------------------------
constexpr int SMP_PER_LANE = 16;
MGX_DEV float u01(uint64_t seed, uint32_t step, uint32_t row) {
    uint32_t x = hash32((uint32_t)seed ^ hash32(step * 0x9e3779b9u + 0x7f4a7c15u) ^ hash32(row + 0x85ebca6bu) ^
                        hash32((uint32_t)(seed >> 32) + 0xc2b2ae35u));
    return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0,1)
}

__global__ __launch_bounds__(64) void sample_kernel(const uint16_t* __restrict__ logits, int V, int ld, float inv_temp,
                                                    int top_k, float top_p, uint64_t seed, int32_t* __restrict__ pos_dev,
                                                    int32_t* __restrict__ next_tok, int32_t* __restrict__ out_tokens,
                                                    int out_ld, float* __restrict__ probs_out, int row0,
                                                    const uint32_t* __restrict__ allow_table) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const uint16_t* lp = logits + (size_t)row * ld;
    float p[SMP_PER_LANE];
    float mx = -INFINITY;
    // grammar mask: row `prev` of allow_table (bit v = token v may follow token prev); the previous
    // token is what next_tok still holds.  A row that allows nothing is ignored.
    const uint32_t* arow = nullptr;
    if (allow_table) {
        int prev = next_tok[row];
        prev = prev < 0 ? 0 : (prev >= V ? V - 1 : prev);
        arow = allow_table + (size_t)prev * ((V + 31) >> 5);
    }
#pragma unroll
    for (int i = 0; i < SMP_PER_LANE; ++i) {
        const int v = lane + 64 * i;
        p[i] = (v < V) ? bf16_to_f32(lp[v]) * inv_temp : -INFINITY;
        if (arow && v < V && !((arow[v >> 5] >> (v & 31)) & 1u)) p[i] = -INFINITY;
        mx = fmaxf(mx, p[i]);
    }
    mx = wave_max(mx);
    if (arow && mx == -INFINITY) {                       // empty row: fall back to the unmasked distribution
#pragma unroll
        for (int i = 0; i < SMP_PER_LANE; ++i) {
            const int v = lane + 64 * i;
            p[i] = (v < V) ? bf16_to_f32(lp[v]) * inv_temp : -INFINITY;
            mx = fmaxf(mx, p[i]);
        }
        mx = wave_max(mx);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SMP_PER_LANE; ++i) { p[i] = (lane + 64 * i < V) ? __expf(p[i] - mx) : 0.f; sum += p[i]; }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < SMP_PER_LANE; ++i) p[i] *= inv;
    if (probs_out) {
#pragma unroll
        for (int i = 0; i < SMP_PER_LANE; ++i)
            if (lane + 64 * i < V) probs_out[(size_t)row * V + lane + 64 * i] = p[i];
    }
    // threshold tau: keep {p_i >= tau}.  top-k: largest tau with count(p >= tau) >= k; top-p: largest tau with
    // mass(p >= tau) >= top_p.  Positive floats order like their bit patterns -> exact bisection on the bits.
    uint32_t tau_bits = 0;
    if (top_k > 0 && top_k < V) {
        uint32_t lo = 0, hi = 0x3f800001u;          // (p >= lo) holds for all; hi = just above 1.0
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            const float tv = __builtin_bit_cast(float, mid);
            float c = 0.f;
#pragma unroll
            for (int i = 0; i < SMP_PER_LANE; ++i) c += (p[i] >= tv && lane + 64 * i < V) ? 1.f : 0.f;
            c = wave_sum(c);
            if (c >= (float)top_k) lo = mid; else hi = mid;
        }
        tau_bits = lo;
    }
    if (top_p < 1.f) {
        uint32_t lo = tau_bits, hi = 0x3f800001u;
        float mass_all = 0.f;
        {
            const float tv = __builtin_bit_cast(float, lo);
#pragma unroll
            for (int i = 0; i < SMP_PER_LANE; ++i) mass_all += (p[i] >= tv) ? p[i] : 0.f;
            mass_all = wave_sum(mass_all);
        }
        const float need = top_p * mass_all;
        while (hi - lo > 1) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            const float tv = __builtin_bit_cast(float, mid);
            float ms = 0.f;
#pragma unroll
            for (int i = 0; i < SMP_PER_LANE; ++i) ms += (p[i] >= tv) ? p[i] : 0.f;
            ms = wave_sum(ms);
            if (ms >= need) lo = mid; else hi = mid;
        }
        tau_bits = lo;
    }
    const float tau = __builtin_bit_cast(float, tau_bits);
    // categorical draw over the kept set by inverse CDF in index order
    float kept = 0.f;
#pragma unroll
    for (int i = 0; i < SMP_PER_LANE; ++i) { if (!(p[i] >= tau)) p[i] = 0.f; kept += p[i]; }
    const float total = wave_sum(kept);
    const int step = pos_dev[0];
    const float target = u01(seed, (uint32_t)step, (uint32_t)(row0 + row)) * total;
    int choice = -1;
    float base = 0.f;
    // index order: v = lane + 64*i  ->  iterate i outer (blocks of 64 consecutive ids), prefix over lanes inner
#pragma unroll
    for (int i = 0; i < SMP_PER_LANE; ++i) {
        float incl = p[i];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        const float blk = __shfl(incl, 63, 64);
        const bool hit = (choice < 0) && (p[i] > 0.f) && (base + incl >= target);
        const unsigned long long mask = __ballot(hit);
        if (choice < 0 && mask) choice = 64 * i + (int)__builtin_ctzll(mask);
        base += blk;
    }
    if (choice < 0) {   // rounding at the top end: take the last kept id
#pragma unroll
        for (int i = SMP_PER_LANE - 1; i >= 0; --i) {
            const unsigned long long mask = __ballot(p[i] > 0.f);
            if (choice < 0 && mask) choice = 64 * i + 63 - (int)__builtin_clzll(mask);
        }
    }
    if (lane == 0) {
        next_tok[row] = choice;
        if (out_tokens) out_tokens[(size_t)row * out_ld + step + 1] = choice;
    }
}
__global__ void advance_pos_kernel(int32_t* pos_dev) { pos_dev[0] += 1; }

extern "C" int mgx_decode_embed(const int32_t* tok, const float* table, const float* pe, const int32_t* pos_dev,
                                uint16_t* out, int B, int d, int V, void* stream) {
    MGX_REQUIRE(tok && table && pe && pos_dev && out, MGX_ERR_NULL, "mgx_decode_embed: NULL pointer");
    MGX_REQUIRE(B > 0 && d > 0 && d % 8 == 0 && V > 0, MGX_ERR_SHAPE, "mgx_decode_embed: need d%%8==0 (B=%d d=%d)", B, d);
    const int total = B * (d / 8);
    hipLaunchKernelGGL(decode_embed_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, tok, table, pe,
                       pos_dev, out, B, d, V, sqrtf((float)d));
    MGX_CHECK_LAUNCH("mgx_decode_embed");
    return MGX_OK;
}

// key splits per (b,h): enough workgroups to keep ~4 per CU (32 waves) streaming once the cache is long; short caches
// keep one workgroup (the merge launch would cost more than it saves)
static int decode_splits(int B, int Lmax, int d) {
    const int wgs = B * (d / 64);
    if (Lmax < 1024) return 1;
    int s = 1;
    while (s < 8 && wgs * s < 1024 && Lmax / (2 * s) >= 512) s *= 2;
    return s;
}

extern "C" int mgx_rel_attn_decode_splits(int B, int Lmax, int d) {
    return (B <= 0 || d <= 0 || Lmax <= 0) ? 0 : decode_splits(B, Lmax, d);
}

extern "C" size_t mgx_rel_attn_decode_workspace(int B, int Lmax, int d) {
    if (B <= 0 || d <= 0 || Lmax <= 0) return 0;
    const int s = decode_splits(B, Lmax, d);
    return s == 1 ? 0 : (size_t)B * (d / 64) * s * DEC_PART * sizeof(float);
}

extern "C" int mgx_rel_attn_decode(const uint16_t* qkv_new, uint16_t* kcache, uint16_t* vcache, const uint16_t* E,
                                   const int32_t* pos_dev, uint16_t* ctx, void* workspace, size_t ws_bytes, int B, int Lmax,
                                   int d, int M, void* stream) {
    MGX_REQUIRE(qkv_new && kcache && vcache && E && pos_dev && ctx, MGX_ERR_NULL, "mgx_rel_attn_decode: NULL pointer");
    MGX_REQUIRE(B > 0 && d > 0 && d % 64 == 0 && Lmax > 0 && M >= Lmax, MGX_ERR_SHAPE,
                "mgx_rel_attn_decode: need d%%64==0 and M>=Lmax (B=%d Lmax=%d d=%d M=%d)", B, Lmax, d, M);
    const int ns = decode_splits(B, Lmax, d);
    MGX_REQUIRE(ns == 1 || (workspace && ws_bytes >= mgx_rel_attn_decode_workspace(B, Lmax, d)), MGX_ERR_SHAPE,
                "mgx_rel_attn_decode: workspace must hold mgx_rel_attn_decode_workspace() = %zu bytes (got %zu)",
                mgx_rel_attn_decode_workspace(B, Lmax, d), ws_bytes);
    hipLaunchKernelGGL(rel_attn_decode_kernel, dim3(B * (d / 64), ns), dim3(64 * DEC_WAVES), 0, (hipStream_t)stream, qkv_new,
                       kcache, vcache, E, pos_dev, ctx, (float*)workspace, Lmax, d, M);
    if (ns > 1)
        hipLaunchKernelGGL(rel_attn_decode_merge_kernel, dim3(B * (d / 64)), dim3(64), 0, (hipStream_t)stream,
                           (const float*)workspace, ctx, ns, d);
    MGX_CHECK_LAUNCH("mgx_rel_attn_decode");
    return MGX_OK;
}

extern "C" int mgx_sample_topk_topp_rows(const uint16_t* logits, int V, int ld, float temperature, int top_k, float top_p,
                                         uint64_t seed, int32_t* pos_dev, int32_t* next_tok, int32_t* out_tokens, int out_ld,
                                         float* probs_out, int B, int row0, int advance, const uint32_t* allow_table,
                                         void* stream) {
    MGX_REQUIRE(logits && pos_dev && next_tok, MGX_ERR_NULL, "mgx_sample_topk_topp: NULL pointer");
    MGX_REQUIRE(B > 0 && V > 0 && V <= 64 * SMP_PER_LANE && ld >= V && temperature > 0.f && top_p > 0.f && row0 >= 0, MGX_ERR_SHAPE,
                "mgx_sample_topk_topp: need 0<V<=%d, ld>=V, temperature>0, top_p>0, row0>=0 (V=%d ld=%d)", 64 * SMP_PER_LANE, V, ld);
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, V, ld, 1.f / temperature, top_k,
                       top_p, seed, pos_dev, next_tok, out_tokens, out_ld, probs_out, row0, allow_table);
    if (advance) hipLaunchKernelGGL(advance_pos_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, pos_dev);
    MGX_CHECK_LAUNCH("mgx_sample_topk_topp");
    return MGX_OK;
}

extern "C" int mgx_sample_topk_topp(const uint16_t* logits, int V, int ld, float temperature, int top_k, float top_p,
                                    uint64_t seed, int32_t* pos_dev, int32_t* next_tok, int32_t* out_tokens, int out_ld,
                                    float* probs_out, int B, int advance, const uint32_t* allow_table, void* stream) {
    return mgx_sample_topk_topp_rows(logits, V, ld, temperature, top_k, top_p, seed, pos_dev, next_tok, out_tokens, out_ld,
                                     probs_out, B, 0, advance, allow_table, stream);
}

// ---------------------------------------------------------------------------------------------------
// K13: Event_Melody_RNN step pieces (Event_MelodyRNN/network.py:51-61): row gather + fused GRU gates
// ---------------------------------------------------------------------------------------------------
// out bf16 [B, ld] = table bf16 [V, ld][tok]        (ld = embedding width padded to the GEMM's K % 64)
__global__ __launch_bounds__(256) void gather_rows_kernel(const int32_t* __restrict__ tok, const uint16_t* __restrict__ table,
                                                          uint16_t* __restrict__ out, int B, int ld, int V) {
    const int gpr = ld >> 3;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= B * gpr) return;
    const int r = g / gpr, c = (g % gpr) * 8;
    int t = tok[r];
    t = t < 0 ? 0 : (t >= V ? V - 1 : t);
    *(u32x4*)(out + (size_t)r * ld + c) = *(const u32x4*)(table + (size_t)t * ld + c);
}
// torch.nn.GRU cell (gate order r,z,n):  r = s(gi_r+gh_r), z = s(gi_z+gh_z), n = tanh(gi_n + r*gh_n),
// h' = (1-z)*n + z*h.   gi, gh bf16 [B,3H] (biases already added by the GEMM epilogue), h f32 [B,H] in/out,
// h_bf16 [B,H] = bf16(h') for the next GEMM.
__global__ __launch_bounds__(256) void gru_gates_kernel(const uint16_t* __restrict__ gi, const uint16_t* __restrict__ gh,
                                                        float* __restrict__ h, uint16_t* __restrict__ h_bf16, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, k = i % H;
    const size_t o = (size_t)b * 3 * H + k;
    const float ir = bf16_to_f32(gi[o]), iz = bf16_to_f32(gi[o + H]), in_ = bf16_to_f32(gi[o + 2 * H]);
    const float hr = bf16_to_f32(gh[o]), hz = bf16_to_f32(gh[o + H]), hn = bf16_to_f32(gh[o + 2 * H]);
    const float r = 1.f / (1.f + __expf(-(ir + hr)));
    const float z = 1.f / (1.f + __expf(-(iz + hz)));
    const float n = tanhf(in_ + r * hn);
    const float hv = (1.f - z) * n + z * h[i];
    h[i] = hv;
    h_bf16[i] = f32_to_bf16(hv);
}

extern "C" int mgx_gather_rows(const int32_t* tok, const uint16_t* table, uint16_t* out, int B, int ld, int V, void* stream) {
    MGX_REQUIRE(tok && table && out, MGX_ERR_NULL, "mgx_gather_rows: NULL pointer");
    MGX_REQUIRE(B > 0 && V > 0 && ld > 0 && ld % 8 == 0, MGX_ERR_SHAPE, "mgx_gather_rows: need ld%%8==0 (ld=%d)", ld);
    const int total = B * (ld / 8);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, tok, table, out, B, ld, V);
    MGX_CHECK_LAUNCH("mgx_gather_rows");
    return MGX_OK;
}
extern "C" int mgx_gru_gates(const uint16_t* gi, const uint16_t* gh, float* h, uint16_t* h_bf16, int B, int H, void* stream) {
    MGX_REQUIRE(gi && gh && h && h_bf16, MGX_ERR_NULL, "mgx_gru_gates: NULL pointer");
    MGX_REQUIRE(B > 0 && H > 0, MGX_ERR_SHAPE, "mgx_gru_gates: bad shape");
    hipLaunchKernelGGL(gru_gates_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, gi, gh, h, h_bf16, B, H);
    MGX_CHECK_LAUNCH("mgx_gru_gates");
    return MGX_OK;
}
