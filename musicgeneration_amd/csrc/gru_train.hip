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
