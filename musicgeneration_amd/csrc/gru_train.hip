// Training path of Event_Melody_RNN (Event_MelodyRNN/network.py:63-84,109-116: SeqForward / Train; the cell is
// torch.nn.GRU, gate order r,z,n).  The input projections of a whole sequence are ONE GEMM per layer
// (mgx_linear_fwd on [T*B, in]); per time step only the recurrent projection (mgx_linear_fwd, M = B) and the
// element-wise cell below remain.  Backward through time mirrors it: per step one cell-backward kernel and one
// recurrent dX GEMM; all weight gradients are batched GEMMs over the whole sequence (mgx_linear_dw).
//
//   r = s(gi_r + gh_r)   z = s(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   h' = (1-z) n + z h
#include "mgx_common.hpp"
#include "mgx.h"

namespace {
MGX_DEV float sigm(float x) { return 1.f / (1.f + __expf(-x)); }
}  // namespace

// h_next f32, y bf16 [B,H] = cell(gi, gh bf16 [B,3H] (biases included), h_prev f32 [B,H])
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const uint16_t* __restrict__ gi, const uint16_t* __restrict__ gh,
                                                           const float* __restrict__ h_prev, float* __restrict__ h_next,
                                                           uint16_t* __restrict__ y, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, k = i % H;
    const size_t o = (size_t)b * 3 * H + k;
    const float r = sigm(bf16_to_f32(gi[o]) + bf16_to_f32(gh[o]));
    const float z = sigm(bf16_to_f32(gi[o + H]) + bf16_to_f32(gh[o + H]));
    const float n = tanhf(bf16_to_f32(gi[o + 2 * H]) + r * bf16_to_f32(gh[o + 2 * H]));
    const float hv = (1.f - z) * n + z * h_prev[i];
    h_next[i] = hv;
    y[i] = f32_to_bf16(hv);
}

// dh = dh_direct (f32, from step t+1's z path) + d_rec (bf16, step t+1's dgh . W_hh) + dy (bf16, from the layer above),
// each optional.  Recomputes the gates from gi/gh (nothing but the pre-activations is saved) and emits
//   dgi, dgh bf16 [B,3H]  (gradients of the two projections' outputs)   and   dh_prev_direct f32 [B,H] = dh * z.
__global__ __launch_bounds__(256) void gru_cell_bwd_kernel(const uint16_t* __restrict__ gi, const uint16_t* __restrict__ gh,
                                                           const float* __restrict__ h_prev,
                                                           const float* __restrict__ dh_direct,
                                                           const uint16_t* __restrict__ d_rec, const uint16_t* __restrict__ dy,
                                                           uint16_t* __restrict__ dgi, uint16_t* __restrict__ dgh,
                                                           float* __restrict__ dh_prev_direct, int B, int H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * H) return;
    const int b = i / H, k = i % H;
    const size_t o = (size_t)b * 3 * H + k;
    const float hn = bf16_to_f32(gh[o + 2 * H]);
    const float r = sigm(bf16_to_f32(gi[o]) + bf16_to_f32(gh[o]));
    const float z = sigm(bf16_to_f32(gi[o + H]) + bf16_to_f32(gh[o + H]));
    const float n = tanhf(bf16_to_f32(gi[o + 2 * H]) + r * hn);
    float dh = 0.f;
    if (dh_direct) dh += dh_direct[i];
    if (d_rec) dh += bf16_to_f32(d_rec[i]);
    if (dy) dh += bf16_to_f32(dy[i]);
    const float dn_pre = dh * (1.f - z) * (1.f - n * n);
    const float dz_pre = dh * (h_prev[i] - n) * z * (1.f - z);
    const float dr_pre = dn_pre * hn * r * (1.f - r);
    dgi[o] = f32_to_bf16(dr_pre);
    dgi[o + H] = f32_to_bf16(dz_pre);
    dgi[o + 2 * H] = f32_to_bf16(dn_pre);
    dgh[o] = f32_to_bf16(dr_pre);
    dgh[o + H] = f32_to_bf16(dz_pre);
    dgh[o + 2 * H] = f32_to_bf16(dn_pre * r);
    dh_prev_direct[i] = dh * z;
}

// =================================================================================================
// Fused time step (round 3).  The per-step pair "recurrent GEMM kernel + cell kernel" (9.5 + 2.2 us forward, 18.6 + 4.1 us
// backward at the reference configuration B = 100, H = 512: the generic 128 x 128 GEMM kernels fill 12 / 4 CUs and pay
// their prologue per launch) becomes ONE kernel each way, built like the decode path's skinny projections:
//   forward : workgroup = 32 hidden units (their 96 gate rows of W_hh) x 32 batch rows; its 4 waves split the reduction
//             (K = H), every fragment of W_hh and h_{t-1} goes straight from global/L2 into MFMA operands with ALL of a
//             wave's loads requested up front (the step is one L2 round trip, not a chain of them); the four partial
//             tiles are summed through LDS in a fixed order, rounded to bf16 exactly as the two-kernel path stores gh,
//             and consumed by the cell in a row-major thread mapping (every global access a contiguous row segment).
//   backward: the same workgroup first forms d_rec = dgh_{t+1} W_hh for its 32 units (A = rows of the transposed copy
//             W_hh^T, reduction over the 3H gates split over 8 waves), then runs the cell backward of step t for those
//             units and writes their slices of dgi_t / dgh_t / dh_direct_t; the next launch (step t-1) reads dgh_t of ALL
//             units, the kernel boundary is the only synchronisation.
// A persistent kernel over the whole sequence was measured out: a device-wide barrier between 16 workgroups costs
// 6 us on MI355X (agent-scope atomics, tools/experiments/grid_barrier_bench.hip), more than a fused step.  The T launches
// of a layer are replayed from a hipGraph by the host side (melody_rnn.py).
// =================================================================================================
namespace {
typedef float f32x16_ __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
MGX_DEV int crow_(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }
MGX_DEV f32x16_ mfma_(const u32x4& a, const u32x4& b, const f32x16_& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_, a), __builtin_bit_cast(bf16x8_, b), c, 0, 0, 0);
}
constexpr int GRU_CH = 8;        // k-steps (of 16) whose loads a wave requests at once
}  // namespace

__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const uint16_t* __restrict__ gi, const uint16_t* __restrict__ hp_bf,
                                                           const float* __restrict__ h_prev, const uint16_t* __restrict__ Whh,
                                                           const float* __restrict__ bhh, float* __restrict__ h_next,
                                                           uint16_t* __restrict__ y, uint16_t* __restrict__ gh_out, int B, int H) {
    __shared__ float part[4][96][33];                        // [k-quarter][gate * 32 + unit][row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int u0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int kq = H >> 2;                                   // K per wave: a multiple of 16 (H % 64 == 0)
    const int mrow = m0 + l31;
    const bool mv = mrow < B;
    const uint16_t* xp = hp_bf + (size_t)(mv ? mrow : 0) * H + w * kq + hh * 8;
    // W_hh in FRAGMENT order (mgx.h): the 16 bytes lane l feeds the MFMA for row tile nt, k-step ks sit at ((nt*K/16 + ks)*64 + l)*16,
    // so a wave load is 1 KB contiguous (from the row-major matrix it touched 32 B of 32 different rows)
    const uint16_t* wp = Whh + (((size_t)(u0 >> 5) * (H >> 4) + (size_t)(w * kq >> 4)) * 64 + lane) * 8;   // gate g: + g * gstride
    const size_t gstride = (size_t)H * H;                    // H/32 row tiles of H/16 k-steps of 512 elements
    // the cell's own inputs (row tid >> 3, 4 units) are requested first: they arrive under the projection's round trip
    const int mr = tid >> 3, u4 = (tid & 7) * 4;
    const int m = m0 + mr, mc = m < B ? m : B - 1;
    const size_t go = (size_t)mc * 3 * H + u0 + u4, ho = (size_t)mc * H + u0 + u4;
    u32x2 gi2[3];
    float bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        gi2[g] = *(const u32x2*)(gi + go + g * H);
        const f32x4 b4 = *(const f32x4*)(bhh + g * H + u0 + u4);
        bh[g][0] = b4.x; bh[g][1] = b4.y; bh[g][2] = b4.z; bh[g][3] = b4.w;
    }
    const f32x4 hp4 = *(const f32x4*)(h_prev + ho);
    f32x16_ acc[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
    for (int k0 = 0; k0 < kq; k0 += 16 * GRU_CH) {
        u32x4 xf[GRU_CH], wf[3][GRU_CH];
#pragma unroll
        for (int f = 0; f < GRU_CH; ++f) {
            const bool in = k0 + 16 * f < kq;
            xf[f] = (mv && in) ? *(const u32x4*)(xp + k0 + 16 * f) : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 3; ++g) wf[g][f] = in ? *(const u32x4*)(wp + g * gstride + (size_t)(k0 + 16 * f) * 32) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int f = 0; f < GRU_CH; ++f)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = mfma_(wf[g][f], xf[f], acc[g]);
    }
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) part[w][g * 32 + crow_(r, hh)][l31] = acc[g][r];
    __syncthreads();
    // the cell, row-major: thread -> (row tid >> 3, 4 consecutive units)
    if (m >= B) return;
    float ghv[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = g * 32 + u4 + k;
            // bf16: the value the backward reads back (and the two-kernel path's rounding point)
            ghv[g][k] = bf16_to_f32(f32_to_bf16(part[0][n][mr] + part[1][n][mr] + part[2][n][mr] + part[3][n][mr] + bh[g][k]));
        }
    float giv[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const u32x2 v = gi2[g];
        giv[g][0] = bf16_to_f32((uint16_t)(v.x & 0xFFFF)); giv[g][1] = bf16_to_f32((uint16_t)(v.x >> 16));
        giv[g][2] = bf16_to_f32((uint16_t)(v.y & 0xFFFF)); giv[g][3] = bf16_to_f32((uint16_t)(v.y >> 16));
    }
    const float hpv[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
    float hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = sigm(giv[0][k] + ghv[0][k]);
        const float z = sigm(giv[1][k] + ghv[1][k]);
        const float n = tanhf(giv[2][k] + r * ghv[2][k]);
        hv[k] = (1.f - z) * n + z * hpv[k];
    }
    *(f32x4*)(h_next + ho) = f32x4{hv[0], hv[1], hv[2], hv[3]};
    *(u32x2*)(y + ho) = u32x2{pack_bf16x2(hv[0], hv[1]), pack_bf16x2(hv[2], hv[3])};
#pragma unroll
    for (int g = 0; g < 3; ++g)
        *(u32x2*)(gh_out + go + g * H) = u32x2{pack_bf16x2(ghv[g][0], ghv[g][1]), pack_bf16x2(ghv[g][2], ghv[g][3])};
}

// Sampling step of one layer (Event_MelodyRNN/network.py:144-149 via gen_forward): BOTH projections and the cell in one launch --
// gi = x W_ih^T + b_ih and gh = h_{t-1} W_hh^T + b_hh (each rounded to bf16 as the three-kernel path stores them), then the cell.
// Same decomposition as the training step above; all fragment loads of both projections are requested before the first MFMA
// (one L2 round trip), the two reductions go through the same LDS buffer one after the other.  h is NOT updated in place:
// other workgroups still read h_{t-1} as their operand, the host alternates two state buffers.
__global__ __launch_bounds__(256) void gru_step_x_fwd_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ Wih,
                                                             const float* __restrict__ bih, int Kx,
                                                             const uint16_t* __restrict__ hp_bf, const float* __restrict__ h_prev,
                                                             const uint16_t* __restrict__ Whh, const float* __restrict__ bhh,
                                                             float* __restrict__ h_next, uint16_t* __restrict__ y, int B, int H) {
    __shared__ float part[4][96][33];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int u0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int kq = H >> 2, kx = Kx >> 2;                     // K per wave of the two projections (multiples of 16)
    const int mrow = m0 + l31;
    const bool mv = mrow < B;
    const int mr = tid >> 3, u4 = (tid & 7) * 4;
    const int m = m0 + mr, mc = m < B ? m : B - 1;
    const size_t ho = (size_t)mc * H + u0 + u4;
    float bi[3][4], bh[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        const f32x4 a4 = *(const f32x4*)(bih + g * H + u0 + u4), b4 = *(const f32x4*)(bhh + g * H + u0 + u4);
        bi[g][0] = a4.x; bi[g][1] = a4.y; bi[g][2] = a4.z; bi[g][3] = a4.w;
        bh[g][0] = b4.x; bh[g][1] = b4.y; bh[g][2] = b4.z; bh[g][3] = b4.w;
    }
    const f32x4 hp4 = *(const f32x4*)(h_prev + ho);
    // both projections: every fragment of the first chunks (8 k-steps = 128 columns per wave: all of them for Kx, H <= 512) is
    // requested before the first MFMA, so the step costs one L2 round trip, not two
    const uint16_t* xr = x + (size_t)(mv ? mrow : 0) * Kx + w * kx + hh * 8;
    const uint16_t* wir = Wih + (((size_t)(u0 >> 5) * (Kx >> 4) + (size_t)(w * kx >> 4)) * 64 + lane) * 8;      // fragment order
    const uint16_t* hr = hp_bf + (size_t)(mv ? mrow : 0) * H + w * kq + hh * 8;
    const uint16_t* whr = Whh + (((size_t)(u0 >> 5) * (H >> 4) + (size_t)(w * kq >> 4)) * 64 + lane) * 8;
    const size_t gsi = (size_t)H * Kx, gsh = (size_t)H * H;
    f32x16_ ai[3], ah[3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ai[g][r] = 0.f; ah[g][r] = 0.f; }
    {
        u32x4 xf[GRU_CH], wf[3][GRU_CH], hf[GRU_CH], vf[3][GRU_CH];
#pragma unroll
        for (int f = 0; f < GRU_CH; ++f) {
            const bool ix = 16 * f < kx, ih = 16 * f < kq;
            xf[f] = (mv && ix) ? *(const u32x4*)(xr + 16 * f) : u32x4{0, 0, 0, 0};
            hf[f] = (mv && ih) ? *(const u32x4*)(hr + 16 * f) : u32x4{0, 0, 0, 0};
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                wf[g][f] = ix ? *(const u32x4*)(wir + g * gsi + (size_t)f * 512) : u32x4{0, 0, 0, 0};
                vf[g][f] = ih ? *(const u32x4*)(whr + g * gsh + (size_t)f * 512) : u32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int f = 0; f < GRU_CH; ++f)
#pragma unroll
            for (int g = 0; g < 3; ++g) ai[g] = mfma_(wf[g][f], xf[f], ai[g]);
#pragma unroll
        for (int f = 0; f < GRU_CH; ++f)
#pragma unroll
            for (int g = 0; g < 3; ++g) ah[g] = mfma_(vf[g][f], hf[f], ah[g]);
    }
    auto rest = [&](const uint16_t* xrow, const uint16_t* wrow, size_t gstride, int kw, f32x16_ (&acc)[3]) {     // K > 512 only
        for (int k0 = 16 * GRU_CH; k0 < kw; k0 += 16 * GRU_CH) {
            u32x4 xf[GRU_CH], wf[3][GRU_CH];
#pragma unroll
            for (int f = 0; f < GRU_CH; ++f) {
                const bool in = k0 + 16 * f < kw;
                xf[f] = (mv && in) ? *(const u32x4*)(xrow + k0 + 16 * f) : u32x4{0, 0, 0, 0};
#pragma unroll
                for (int g = 0; g < 3; ++g) wf[g][f] = in ? *(const u32x4*)(wrow + g * gstride + (size_t)(k0 + 16 * f) * 32) : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int f = 0; f < GRU_CH; ++f)
#pragma unroll
                for (int g = 0; g < 3; ++g) acc[g] = mfma_(wf[g][f], xf[f], acc[g]);
        }
    };
    rest(xr, wir, gsi, kx, ai);
    rest(hr, whr, gsh, kq, ah);
    float giv[3][4], ghv[3][4];
    auto reduce = [&](const f32x16_ (&acc)[3], const float (&bias)[3][4], float (&out)[3][4]) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) part[w][g * 32 + crow_(r, hh)][l31] = acc[g][r];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = g * 32 + u4 + k;
                out[g][k] = bf16_to_f32(f32_to_bf16(part[0][n][mr] + part[1][n][mr] + part[2][n][mr] + part[3][n][mr] + bias[g][k]));
            }
        __syncthreads();
    };
    reduce(ai, bi, giv);
    reduce(ah, bh, ghv);
    if (m >= B) return;
    const float hpv[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
    float hv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = sigm(giv[0][k] + ghv[0][k]);
        const float z = sigm(giv[1][k] + ghv[1][k]);
        const float n = tanhf(giv[2][k] + r * ghv[2][k]);
        hv[k] = (1.f - z) * n + z * hpv[k];
    }
    *(f32x4*)(h_next + ho) = f32x4{hv[0], hv[1], hv[2], hv[3]};
    *(u32x2*)(y + ho) = u32x2{pack_bf16x2(hv[0], hv[1]), pack_bf16x2(hv[2], hv[3])};
}

// NS waves split the reduction over the 3H gates.  final != 0: no cell, dh_out = dh_direct + dgh_next W_hh (the gradient of
// the layer's initial state)
template <int NS>
__global__ __launch_bounds__(64 * NS) void gru_step_bwd_kernel(const uint16_t* __restrict__ gi, const uint16_t* __restrict__ gh,
                                                               const float* __restrict__ h_prev, const float* __restrict__ dh_direct,
                                                               const uint16_t* __restrict__ dgh_next, const uint16_t* __restrict__ WhhT,
                                                               const uint16_t* __restrict__ dy, uint16_t* __restrict__ dgi,
                                                               uint16_t* __restrict__ dgh, float* __restrict__ dh_out, int B, int H,
                                                               int final) {
    __shared__ float part[NS][32][33];                       // [k-slice][unit][row]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int u0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int G3 = 3 * H;
    // the cell's own inputs (threads 0..255: row tid >> 3, 4 units) are requested first
    const int mr = (tid >> 3) & 31, u4 = (tid & 7) * 4;
    const int m = m0 + mr, mc = m < B ? m : B - 1;
    const size_t go = (size_t)mc * G3 + u0 + u4, ho = (size_t)mc * H + u0 + u4;
    u32x2 gi2[3] = {}, gh2[3] = {}, dy2 = {};
    f32x4 hp4 = {}, dd4 = {};
    if (tid < 256) {
        if (!final) {
#pragma unroll
            for (int g = 0; g < 3; ++g) { gi2[g] = *(const u32x2*)(gi + go + g * H); gh2[g] = *(const u32x2*)(gh + go + g * H); }
            hp4 = *(const f32x4*)(h_prev + ho);
            if (dy) dy2 = *(const u32x2*)(dy + ho);
        }
        if (dh_direct) dd4 = *(const f32x4*)(dh_direct + ho);
    }
    if (dgh_next) {                                           // kernel-uniform
        // d_rec^T[u][m] = sum_g W_hh^T[u][g] dgh_next[m][g]
        const int kq = G3 / NS;                              // a multiple of 16 (checked by the host)
        const int mrow = m0 + l31;
        const bool mv = mrow < B;
        const uint16_t* xp = dgh_next + (size_t)(mv ? mrow : 0) * G3 + w * kq + hh * 8;
        const uint16_t* wp = WhhT + (((size_t)(u0 >> 5) * (G3 >> 4) + (size_t)(w * kq >> 4)) * 64 + lane) * 8;   // fragment order
        f32x16_ acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < kq; k0 += 16 * 12) {            // 24 loads requested at once
            u32x4 xf[12], wf[12];
#pragma unroll
            for (int f = 0; f < 12; ++f) {
                const bool in = k0 + 16 * f < kq;
                xf[f] = (mv && in) ? *(const u32x4*)(xp + k0 + 16 * f) : u32x4{0, 0, 0, 0};
                wf[f] = in ? *(const u32x4*)(wp + (size_t)(k0 + 16 * f) * 32) : u32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int f = 0; f < 12; ++f) acc = mfma_(wf[f], xf[f], acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) part[w][crow_(r, hh)][l31] = acc[r];
        __syncthreads();
    }
    if (tid >= 256 || m >= B) return;
    float dh[4] = {dd4.x, dd4.y, dd4.z, dd4.w};              // zero without dh_direct
    if (dgh_next) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < NS; ++q) sum += part[q][u4 + k][mr];
            dh[k] += bf16_to_f32(f32_to_bf16(sum));         // rounded like the output of the GEMM kernel this replaces
        }
    }
    if (final) {
        *(f32x4*)(dh_out + ho) = f32x4{dh[0], dh[1], dh[2], dh[3]};
        return;
    }
    auto up4 = [](const u32x2& v, float* f) {
        f[0] = bf16_to_f32((uint16_t)(v.x & 0xFFFF)); f[1] = bf16_to_f32((uint16_t)(v.x >> 16));
        f[2] = bf16_to_f32((uint16_t)(v.y & 0xFFFF)); f[3] = bf16_to_f32((uint16_t)(v.y >> 16));
    };
    auto pk4 = [](const float* f) { return u32x2{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3])}; };
    if (dy) {
        float t4[4];
        up4(dy2, t4);
#pragma unroll
        for (int k = 0; k < 4; ++k) dh[k] += t4[k];
    }
    float giv[3][4], ghv[3][4];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
        up4(gi2[g], giv[g]);
        up4(gh2[g], ghv[g]);
    }
    const float hpv[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
    float dr[4], dz[4], dn[4], dnr[4], dd[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float r = sigm(giv[0][k] + ghv[0][k]);
        const float z = sigm(giv[1][k] + ghv[1][k]);
        const float n = tanhf(giv[2][k] + r * ghv[2][k]);
        dn[k] = dh[k] * (1.f - z) * (1.f - n * n);
        dz[k] = dh[k] * (hpv[k] - n) * z * (1.f - z);
        dr[k] = dn[k] * ghv[2][k] * r * (1.f - r);
        dnr[k] = dn[k] * r;
        dd[k] = dh[k] * z;
    }
    *(u32x2*)(dgi + go) = pk4(dr);
    *(u32x2*)(dgi + go + H) = pk4(dz);
    *(u32x2*)(dgi + go + 2 * H) = pk4(dn);
    *(u32x2*)(dgh + go) = pk4(dr);
    *(u32x2*)(dgh + go + H) = pk4(dz);
    *(u32x2*)(dgh + go + 2 * H) = pk4(dnr);
    *(f32x4*)(dh_out + ho) = f32x4{dd[0], dd[1], dd[2], dd[3]};
}

// out = dropout(x): keep/drop is a pure function of (seed, element index) -- the same call on the gradient is the
// backward.  8 elements per thread (n % 8 == 0).
__global__ __launch_bounds__(256) void dropout_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out, size_t n8,
                                                           DropCfg dc) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n8) return;
    float f[8], m[8];
    unpack8(*(const u32x4*)(x + g * 8), f);
    drop_mult8(dc, (uint32_t)g, m);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] *= m[k];
    *(u32x4*)(out + g * 8) = pack8(f);
}

// dst f32 [V,cols] row idx[r] += src bf16 [n,ld] row r   (embedding gradient; float atomics, 128-byte segments per wave)
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const int32_t* __restrict__ idx, const uint16_t* __restrict__ src,
                                                               float* __restrict__ dst, int n, int ld, int cols, int V) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(g / cols), c = (int)(g % cols);
    if (r >= n) return;
    const int t = idx[r];
    if (t < 0 || t >= V) return;
    atomicAdd(dst + (size_t)t * cols + c, bf16_to_f32(src[(size_t)r * ld + c]));
}

extern "C" int mgx_gru_cell_fwd(const uint16_t* gi, const uint16_t* gh, const float* h_prev, float* h_next, uint16_t* y,
                                int B, int H, void* stream) {
    MGX_REQUIRE(gi && gh && h_prev && h_next && y, MGX_ERR_NULL, "mgx_gru_cell_fwd: NULL pointer");
    MGX_REQUIRE(B > 0 && H > 0, MGX_ERR_SHAPE, "mgx_gru_cell_fwd: bad shape");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, gi, gh, h_prev, h_next, y, B, H);
    MGX_CHECK_LAUNCH("mgx_gru_cell_fwd");
    return MGX_OK;
}

extern "C" int mgx_gru_cell_bwd(const uint16_t* gi, const uint16_t* gh, const float* h_prev, const float* dh_direct,
                                const uint16_t* d_rec, const uint16_t* dy, uint16_t* dgi, uint16_t* dgh,
                                float* dh_prev_direct, int B, int H, void* stream) {
    MGX_REQUIRE(gi && gh && h_prev && dgi && dgh && dh_prev_direct, MGX_ERR_NULL, "mgx_gru_cell_bwd: NULL pointer");
    MGX_REQUIRE(B > 0 && H > 0, MGX_ERR_SHAPE, "mgx_gru_cell_bwd: bad shape");
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3((B * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, gi, gh, h_prev, dh_direct,
                       d_rec, dy, dgi, dgh, dh_prev_direct, B, H);
    MGX_CHECK_LAUNCH("mgx_gru_cell_bwd");
    return MGX_OK;
}

extern "C" int mgx_dropout_bf16(const uint16_t* x, uint16_t* out, size_t n, float p_drop, uint64_t seed, void* stream) {
    MGX_REQUIRE(x && out, MGX_ERR_NULL, "mgx_dropout_bf16: NULL pointer");
    MGX_REQUIRE(n > 0 && n % 8 == 0 && p_drop >= 0.f && p_drop < 1.f, MGX_ERR_SHAPE,
                "mgx_dropout_bf16: need n %% 8 == 0 and 0 <= p < 1");
    hipLaunchKernelGGL(dropout_bf16_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, out,
                       n / 8, make_drop(p_drop, seed));
    MGX_CHECK_LAUNCH("mgx_dropout_bf16");
    return MGX_OK;
}

extern "C" int mgx_scatter_add_rows(const int32_t* idx, const uint16_t* src, float* dst, int n, int ld, int cols, int V,
                                    void* stream) {
    MGX_REQUIRE(idx && src && dst, MGX_ERR_NULL, "mgx_scatter_add_rows: NULL pointer");
    MGX_REQUIRE(n > 0 && cols > 0 && ld >= cols && V > 0, MGX_ERR_SHAPE, "mgx_scatter_add_rows: bad shape");
    const size_t total = (size_t)n * cols;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, src,
                       dst, n, ld, cols, V);
    MGX_CHECK_LAUNCH("mgx_scatter_add_rows");
    return MGX_OK;
}

extern "C" int mgx_gru_step_fwd(const uint16_t* gi, const uint16_t* h_prev_bf, const float* h_prev, const uint16_t* Whh,
                                const float* bhh, float* h_next, uint16_t* y, uint16_t* gh_out, int B, int H, void* stream) {
    MGX_REQUIRE(gi && h_prev_bf && h_prev && Whh && bhh && h_next && y && gh_out, MGX_ERR_NULL, "mgx_gru_step_fwd: NULL pointer");
    MGX_REQUIRE(B > 0 && H > 0 && H % 64 == 0, MGX_ERR_SHAPE, "mgx_gru_step_fwd: need H %% 64 == 0 (got B=%d H=%d)", B, H);
    hipLaunchKernelGGL(gru_step_fwd_kernel, dim3(H / 32, (B + 31) / 32), dim3(256), 0, (hipStream_t)stream, gi, h_prev_bf, h_prev,
                       Whh, bhh, h_next, y, gh_out, B, H);
    MGX_CHECK_LAUNCH("mgx_gru_step_fwd");
    return MGX_OK;
}

extern "C" int mgx_gru_step_bwd(const uint16_t* gi, const uint16_t* gh, const float* h_prev, const float* dh_direct,
                                const uint16_t* dgh_next, const uint16_t* WhhT, const uint16_t* dy, uint16_t* dgi, uint16_t* dgh,
                                float* dh_out, int B, int H, int final, void* stream) {
    MGX_REQUIRE(dh_out && (final || (gi && gh && h_prev && dgi && dgh)), MGX_ERR_NULL, "mgx_gru_step_bwd: NULL pointer");
    MGX_REQUIRE(!dgh_next || WhhT, MGX_ERR_NULL, "mgx_gru_step_bwd: dgh_next needs the transposed recurrent weight");
    MGX_REQUIRE(B > 0 && H > 0 && H % 64 == 0, MGX_ERR_SHAPE, "mgx_gru_step_bwd: need H %% 64 == 0 (got B=%d H=%d)", B, H);
    const dim3 grid(H / 32, (B + 31) / 32);
    if ((3 * H) % 128 == 0)      // 8 waves split the 3H gates (one round of loads per wave at H = 512)
        hipLaunchKernelGGL(gru_step_bwd_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, gi, gh, h_prev, dh_direct, dgh_next, WhhT,
                           dy, dgi, dgh, dh_out, B, H, final);
    else                         // 3H / 4 is a multiple of 16 for every H % 64 == 0
        hipLaunchKernelGGL(gru_step_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, gi, gh, h_prev, dh_direct, dgh_next, WhhT,
                           dy, dgi, dgh, dh_out, B, H, final);
    MGX_CHECK_LAUNCH("mgx_gru_step_bwd");
    return MGX_OK;
}

extern "C" int mgx_gru_step_x_fwd(const uint16_t* x, const uint16_t* Wih, const float* bih, int Kx, const uint16_t* h_prev_bf,
                                  const float* h_prev, const uint16_t* Whh, const float* bhh, float* h_next, uint16_t* y, int B,
                                  int H, void* stream) {
    MGX_REQUIRE(x && Wih && bih && h_prev_bf && h_prev && Whh && bhh && h_next && y, MGX_ERR_NULL, "mgx_gru_step_x_fwd: NULL pointer");
    MGX_REQUIRE(h_next != h_prev && y != h_prev_bf, MGX_ERR_SHAPE, "mgx_gru_step_x_fwd: the state is not updated in place "
                "(other workgroups still read h_prev): pass a second pair of buffers");
    MGX_REQUIRE(B > 0 && H > 0 && H % 64 == 0 && Kx > 0 && Kx % 64 == 0, MGX_ERR_SHAPE,
                "mgx_gru_step_x_fwd: need H %% 64 == 0 and Kx %% 64 == 0 (got B=%d H=%d Kx=%d)", B, H, Kx);
    hipLaunchKernelGGL(gru_step_x_fwd_kernel, dim3(H / 32, (B + 31) / 32), dim3(256), 0, (hipStream_t)stream, x, Wih, bih, Kx,
                       h_prev_bf, h_prev, Whh, bhh, h_next, y, B, H);
    MGX_CHECK_LAUNCH("mgx_gru_step_x_fwd");
    return MGX_OK;
}
