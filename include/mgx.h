/* mgx.h -- C ABI of libmgx.so: the MI355X (gfx950) kernels of the Music Transformer hot path.
 *
 * The reference (SJTMusicTeam/MusicGeneration) has no native/FFI layer: its hot path is a
 * sequence of ATen ops inside torch.nn.Module classes.  Each entry point below replaces one such
 * op sequence; the reference lines it replaces are cited per function (paths relative to
 * mg/model/MusicTransformer/).  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked "host".
 *  - the caller owns all buffers; the library allocates nothing and keeps no state.
 *  - all launches are asynchronous on `stream` (a hipStream_t passed as void*); capturable.
 *  - bf16 tensors are uint16_t storage (round-to-nearest-even), row-major, innermost dim last.
 *  - return value: MGX_OK (0) or a negative mgx_status; mgx_last_error() gives a message.
 *  - dh (head dim) is fixed at 64 and heads = d/64, as in the reference (layers.py:219).
 */
#ifndef MGX_H
#define MGX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MGX_OK = 0,
    MGX_ERR_SHAPE = -1,    /* unsupported shape (message says which constraint) */
    MGX_ERR_NULL = -2,     /* required pointer is NULL */
    MGX_ERR_LAUNCH = -3,   /* hipGetLastError() != hipSuccess after launch */
    MGX_ERR_NO_DEVICE = -4 /* no HIP device visible */
} mgx_status;

/* thread-local, NUL-terminated description of the last failure on this thread */
const char* mgx_last_error(void);
/* library/ABI version (bumped on any signature change) */
int mgx_abi_version(void);   /* 2: mgx_rel_attn_bwd takes a workspace; 3: mgx_linear_dx takes an addend; 4: mgx_linear_dw_grouped; 5: GRU training ops; 6: sampler grammar mask; 7: mgx_linear_ln_fwd; 8: mgx_rel_attn_fwd/_weights take a workspace; 9: mgx_rel_attn_decode takes a workspace (split-K); 10: mgx_linear_dw_grouped takes a workspace; 11: decode K/V caches are head-major [B,h,Lmax,64]; 12: mgx_decode_embed_linear, mgx_rel_attn_decode_splits; attention partials are 68 floats (acc[64], m, l, 2 pad); 13: mgx_rel_attn_bwd_parts: dK/dV stores the dS tiles, bits 1/3 read them, bit 5 = dQ by recomputation; 14: mgx_gru_step_fwd/bwd, mgx_gru_step_x_fwd, fragment-ordered weights (*_frag), mgx_rel_attn_fwd_nomask; 15: mgx_sample_topk_topp_rows; 16: mgx_smooth_ce_bwd takes a device-side scale, mgx_pad_bitmap a flag, mgx_set_deterministic; 17: mgx_rel_attn_bwd_parts bit 6; 18: mgx_pad_bitmap's flag records LEADING pads only, mgx_stream_create_cu_mask / mgx_stream_set_cus / mgx_stream_cus / mgx_stream_destroy, mgx_set_deterministic_stream, mgx_linear_kernel_id */
#define MGX_ABI_VERSION 18
/* number of visible HIP devices, or a negative mgx_status */
int mgx_device_count(void);

/* ---- deterministic-reduction mode (ABI 16)                               SURVEY 5.2 / 8c "DP N-GPU vs 1-GPU"
 * The sums that cross workgroups -- dE (mgx_rel_attn_bwd), the dW / db of mgx_linear_dw (the vocabulary projection), the bias
 * gradients of mgx_linear_dw_grouped, the embedding gradient (mgx_embed_bwd) and the four statistics of mgx_smooth_ce_fwd --
 * end in fp32 atomics, so their last bits depend on the order in which workgroups arrive.  After
 * mgx_set_deterministic(scratch, bytes) those kernels add 64-bit FIXED-POINT integers (value * 2^30, each partial sum rounded
 * once) into `scratch` and a fold pass converts the totals: integer addition is associative, so two runs on the same inputs
 * -- and a data-parallel run against a single-process run of the same global batch, up to the all-reduce's own order -- give
 * the same bits.  Partial sums are quantised to 2^-30 ~ 9.3e-10 (a FIXED quantum: an element whose partials are of magnitude m
 * keeps log2(m) + 30 bits -- tests/test_gpu_fullsize.py::test_deterministic_mode_at_the_bench_shape states the bound) and must
 * stay below 2^31 ~ 2.1e9 in magnitude: a partial that is NaN, infinite or larger POISONS its destination -- the fold writes NaN
 * there, as it does for a total outside +-2^31 -- so a diverging run stays recognisable from its loss and gradients.  The poison
 * is sticky (ABI 18): it lives in a word of its own, only ever OR-ed, never in the sum that later partials keep adding to.  The
 * cfg2 training step takes 1.1 % longer.  Everything else in the library is deterministic as it is.
 *   scratch: device memory, EXACTLY 32 MiB used, aligned to 32 MiB (ABI 18: sums in the lower 16 MiB -- 2 M destinations, the
 *            largest user needs max(N*K + N, V*d, 64*L) of them, cfg4: 1.8 M -- and each sum's poison word at the same offset in
 *            the upper 16 MiB, so that no kernel needs a second pointer), owned by the caller and alive until the mode is
 *            switched off with mgx_set_deterministic(NULL, 0).
 *   PROCESS-GLOBAL STATE -- the one exception to "every entry point is stateless and takes its stream" (SURVEY 8b): (a) the
 *   setting applies to all threads and all model instances of the process, (b) switching it while such a call is in flight is
 *   undefined, (c) the scratch registered here serves the calls of ONE stream at a time (two streams sharing it would zero / fold
 *   each other's partial sums): every FURTHER stream that issues such calls -- the CU-masked side stream of the backward's
 *   off-critical-path kernels, below -- registers a buffer of its own with mgx_set_deterministic_stream (ABI 18); a stream
 *   without one uses the buffer registered here.                                                                        */
int mgx_set_deterministic(void* scratch, size_t bytes);
int mgx_deterministic(void);          /* 1 while a scratch is registered */
/* ABI 18: a scratch of its own for the calls issued on `stream` (same requirements as above; NULL forgets it).  The mode must be
 * on (mgx_set_deterministic); switching the mode off forgets every per-stream buffer.                                    */
int mgx_set_deterministic_stream(void* stream, void* scratch, size_t bytes);

/* ---- CU-masked streams (ABI 18)                      the backward of layers.py:152-161 as train.py:265-277 drives it; SURVEY 8e
 * A training step is half MFMA-bound kernels (attention dK/dV and forward, the QKV / output-projection GEMMs) and half kernels
 * at the HBM rate, some of which feed only the optimiser (dE, the weight gradients): they can run BESIDE the critical path
 * instead of inside it -- if the two are kept on disjoint CUs, since workgroups of two unrestricted streams are dispatched
 * one kernel after the other.  The same mask keeps CUs free for RCCL's kernels under data parallelism.
 *   mgx_stream_create_cu_mask: *stream = a new HIP stream (hipExtStreamCreateWithCUMask) restricted to the CUs whose bits are
 *     set in mask[0 .. words-1].  Bit i is the driver's logical CU i: consecutive bits go round the XCDs first, then round the
 *     shader engines of an XCD, so "the low n bits" = n/8 CUs of every XCD (tools/cumask_probe.hip prints the mapping of a box).
 *   mgx_stream_set_cus / mgx_stream_cus: the number of CUs the library assumes for `stream` (set by the call above; settable for
 *     a stream created elsewhere; 0 forgets; an unknown stream = the whole device).  The persistent GEMM kernels launch one
 *     workgroup per CU of the stream they run on and plan their M-splits for that many.
 *   mgx_stream_destroy: forgets the stream (CU count, deterministic scratch) and destroys it.                            */
int mgx_stream_create_cu_mask(void** stream, const uint32_t* mask, int words);
int mgx_stream_set_cus(void* stream, int cus);
int mgx_stream_cus(void* stream);
int mgx_stream_destroy(void* stream);

/* ---- K1: token embedding * sqrt(d) + sinusoid PE (+dropout)      layers.py:226-229, 22-39 ----
 * tok int32 [rows] (rows = B*L), table f32 [V,d], pe f32 [L,d] (precomputed once, device resident),
 * out bf16 [rows,d].  dropout: keep-prob 1-p, inverted scaling; mask is a pure function of
 * (seed, element index) so the backward regenerates it.  p == 0 disables it.                  */
int mgx_embed_pe_fwd(const int32_t* tok, const float* table, const float* pe, uint16_t* out,
                     int B, int L, int d, int V, float p_drop, uint64_t seed, void* stream);
/* dtable f32 [V,d] += scatter-add of dout bf16 [rows,d] (per vocabulary row and token range one block-wide sum, added with
 * one fp32 atomic per element: the order of those few adds, hence the last bits, may differ between runs) */
int mgx_embed_bwd(const int32_t* tok, const uint16_t* dout, float* dtable,
                  int B, int L, int d, int V, float p_drop, uint64_t seed, void* stream);

/* ---- A3: key-padding bitmap from tokens                          utils.py:58-83 --------------
 * bits uint32 [B, L/32]: bit (j&31) of word j>>5 set iff tok[b,j] == pad.  L % 32 == 0.
 * flag (ABI 16; device uint32[1] or NULL): bit 0 is OR-ed in, and never cleared, when a row STARTS with a pad token and holds a
 *   real token later (ABI 18; until then: any pad followed by a real token).  Only leading padding creates queries whose
 *   visible keys are all padding; such a query has no defined result in the reference (softmax of -1e9 + x in fp32,
 *   layers.py:99-102) and is outside the parity contract, so the host side reads this flag at its next synchronisation point
 *   and refuses such input (no host sync on the hot path).  Trailing and interior pads are masked as keys exactly like the
 *   reference's look-ahead mask does (utils.py:58-83) and are accepted.                                                  */
int mgx_pad_bitmap(const int32_t* tok, uint32_t* bits, uint32_t* flag, int B, int L, int pad, void* stream);

/* ---- K3+K4: fused relative global attention (Shaw/Huang skewing)  layers.py:86-106,111-133 ---
 * qkv bf16 [B,L,3d]: columns [0,d)=Q, [d,2d)=K, [2d,3d)=V, head hd at columns hd*64..hd*64+63
 *     (exactly the output of one fused [d -> 3d] projection; no permute is needed).
 * E   bf16 [M,64] relative embedding of this layer (shared by heads and batch), M >= L.
 * padbits uint32 [B,L/32] from mgx_pad_bitmap, or NULL for "no key padding".
 * ctx bf16 [B,L,d] (heads merged: the input of the `fc` projection), lse f32 [B,h,L].
 *   logit[i,j] = (q_i.k_j + q_i.E[M-1-(i-j)])/8 ; j>i masked ; padded keys get -1e9 ;
 *   ctx_i = softmax_j(logit) v_j.  The L x L matrix is never materialised.
 * Constraints: d % 64 == 0, L % 32 == 0, M >= L.
 * Rows whose every key j<=i is padding (leading pads; never produced by the reference's data
 * path, data.py:96-107) attend uniformly over j<=i (the reference's result there is a rounding
 * artefact of -1e9+x in fp32 and is outside the parity contract).                             */
/* workspace: caller-provided scratch, 256-byte aligned, >= mgx_rel_attn_fwd_workspace(L) bytes (a copy of E in
 * the order the MFMA lanes consume it, rebuilt by a pre-pass on every call: 128*L bytes).                    */
size_t mgx_rel_attn_fwd_workspace(int L);
int mgx_rel_attn_fwd(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits,
                     uint16_t* ctx, float* lse, void* workspace, size_t ws_bytes,
                     int B, int L, int d, int M, void* stream);
/* The reference's SAMPLING call Decoder(x, mask=None) (network.py:60-62): every query attends to every key j < Lk (no look-ahead, no
 * padding mask), the relative term is what _qe_masking + _skewing leave of it: q_i.E[M-1-(i-j)] for j <= i, 0 for j > i.  Lk <= L =
 * real length of the window; rows / keys Lk..L-1 only pad it to a multiple of 32.  Inference only.                          */
int mgx_rel_attn_fwd_nomask(const uint16_t* qkv, const uint16_t* E, uint16_t* ctx, float* lse, void* workspace, size_t ws_bytes,
                            int B, int L, int Lk, int d, int M, void* stream);
/* eval/debug output of the reference (`attention_weights`, layers.py:102,109): weights f32 [B,h,L,L] =
 * softmax rows recomputed from qkv and the lse of mgx_rel_attn_fwd.  The caller zero-fills `weights`
 * first (tiles above the diagonal are not visited); masked entries inside visited tiles are written as 0. */
int mgx_rel_attn_weights(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits, const float* lse,
                         float* weights, void* workspace, size_t ws_bytes, int B, int L, int d, int M, void* stream);
/* backward of the above (autograd of layers.py:86-106).  dctx bf16 [B,L,d] -> dqkv bf16 [B,L,3d];
 * dE f32 [M,64] is ACCUMULATED into (caller zeroes it once per optimiser step).
 * workspace: caller-provided scratch, 256-byte aligned, >= mgx_rel_attn_bwd_workspace(B,L,d) bytes
 * (rowsum(dctx*ctx) [B,h,L] f32, two lane-ordered bf16 copies of E, and the causal half of dS by (query tile, key tile)
 * in 32x32 bf16 tiles that the dK/dV kernel leaves for the dQ and dE kernels -- B*h*(L/32)(L/32+1)/2 * 2 KB,
 * 0.55 GB at cfg2/B=16). */
size_t mgx_rel_attn_bwd_workspace(int B, int L, int d);
int mgx_rel_attn_bwd(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits,
                     const uint16_t* ctx, const uint16_t* dctx, const float* lse,
                     uint16_t* dqkv, float* dE, void* workspace, size_t ws_bytes,
                     int B, int L, int d, int M, void* stream);
/* same, running only the selected sub-kernels, in this order inside one call:
 *   bit0 (1)  pre-pass: delta = rowsum(dctx*ctx), E re-layout
 *   bit2 (4)  dK + dV; stores every dS tile in the workspace (L % 128 == 0: the 64-keys-per-wave kernel whose sweep is a generated,
 *             hand-scheduled gfx950 asm block, csrc/rel_attn_dkv64.hip; any other L: the 32-key HIP kernel)
 *   bit6 (64) dK + dV by the 32-key HIP kernel whatever the shape (instead of bit2: cross-check, the two give the same bits)
 *   bit1 (2)  dQ from the dS tiles a bit2 run (this call or an earlier one) left in the SAME workspace
 *   bit5 (32) dQ by full recomputation instead (independent of the stored tiles: cross-check; not together with bit1)
 *   bit3 (8)  dE from the stored dS tiles
 *   bit4 (16) dE by full recomputation (independent of the stored tiles: cross-check)
 * bench.py times each kernel on its own this way.  parts == 15 is mgx_rel_attn_bwd.                     */
int mgx_rel_attn_bwd_parts(const uint16_t* qkv, const uint16_t* E, const uint32_t* padbits,
                           const uint16_t* ctx, const uint16_t* dctx, const float* lse,
                           uint16_t* dqkv, float* dE, void* workspace, size_t ws_bytes,
                           int B, int L, int d, int M, int parts, void* stream);

/* ---- K6: out = LayerNorm(dropout(x) + res) * gamma + beta, eps    layers.py:154-155,159-160 --
 * x,res,out bf16 [rows,d]; gamma,beta f32 [d]; mean,rstd f32 [rows] saved for the backward.    */
int mgx_add_ln_fwd(const uint16_t* x, const uint16_t* res, const float* gamma, const float* beta,
                   uint16_t* out, float* mean, float* rstd, int rows, int d, float eps,
                   float p_drop, uint64_t seed, void* stream);
/* dout bf16 [rows,d] -> dx (grad of x, dropout applied), dres (grad of res) bf16 [rows,d];
 * dgamma,dbeta f32 [d] are ACCUMULATED into; dxsum f32 [d] (or NULL) += column sums of dx, i.e. the bias
 * gradient of the projection that produced x.  dx may alias dres when p_drop == 0.
 * workspace: caller scratch >= mgx_add_ln_bwd_workspace(rows,d) bytes (per-block column partials).  */
size_t mgx_add_ln_bwd_workspace(int rows, int d);
int mgx_add_ln_bwd(const uint16_t* dout, const uint16_t* x, const uint16_t* res, const float* gamma,
                   const float* mean, const float* rstd, uint16_t* dx, uint16_t* dres,
                   float* dgamma, float* dbeta, float* dxsum, void* workspace, size_t ws_bytes,
                   int rows, int d, float p_drop, uint64_t seed, void* stream);

/* ---- K9+K10: label-smoothed cross entropy + accuracy              criterion.py:43-67, metrics.py:22-60
 * logits bf16 [rows,ld] (row stride ld >= V elements; columns >= V are ignored), target int32 [rows].
 * stats f32 [4] is ACCUMULATED into: [0]=sum of per-row loss over target!=pad, [1]=#target!=pad,
 *   [2]=#(argmax==target) over ALL rows, [3]=rows.   argmax int32 [rows] (LogitsBucketting).
 * row_lse f32 [rows] is saved for the backward.                                               */
int mgx_smooth_ce_fwd(const uint16_t* logits, const int32_t* target, float* stats, int32_t* argmax,
                      float* row_lse, int rows, int V, int ld, float eps_ls, int pad, void* stream);
/* dlogits bf16 [rows,ld] (columns >= V written as 0) = g/stats[1] * (softmax - q') for target!=pad rows, else 0, where
 * g = gscale * (gscale_dev ? *gscale_dev : 1): the upstream gradient of the scalar loss -- a host constant (1/accum) times,
 * optionally, a device-side f32 scalar (autograd's grad_output, the data-parallel loss weight): both are read on the device,
 * as stats is (no host sync, no extra elementwise pass over dlogits) (ABI 16).                    */
int mgx_smooth_ce_bwd(const uint16_t* logits, const int32_t* target, const float* stats,
                      const float* row_lse, uint16_t* dlogits, int rows, int V, int ld,
                      float eps_ls, int pad, float gscale, const float* gscale_dev, void* stream);

/* ---- K11: Adam over one flat fp32 parameter buffer                train.py:143, criterion.py:81-87
 * p,g,m,v f32 [n]; shadow bf16 [n] (the bf16 copy the GEMMs/attention read) written in the same
 * pass.  step >= 1 (bias correction as torch.optim.Adam), gscale multiplies g (e.g. 1/world).   */
int mgx_adam_step(float* p, const float* g, float* m, float* v, uint16_t* shadow, size_t n,
                  float lr, float beta1, float beta2, float eps, int step, float gscale, void* stream);
/* shadow bf16 [n] = round(p f32 [n]) */
int mgx_cast_bf16(const float* p, uint16_t* shadow, size_t n, void* stream);

/* ---- K2/K5/K7/K8: C = act(A @ W^T + bias)                         layers.py:71-84,108,157-158; network.py:39
 * A bf16 [M,K] row-major, W bf16 [N,K] row-major (torch.nn.Linear layout), bias f32 [N] or NULL,
 * C bf16 [M,N].  act: 0 none, 1 ReLU.  K % 64 == 0, N % 4 == 0 (pad the weight rows, as the
 * vocabulary projection does); M arbitrary (edge tiles are masked).   */
int mgx_linear_fwd(const uint16_t* A, const uint16_t* W, const float* bias, uint16_t* C,
                   int M, int N, int K, int act, void* stream);

/* decode-size fusion (M <= 32, K <= 1024): Z bf16 [M,K] = LayerNorm(X + RES) (layers.py:154-155,159-160 in eval mode,
 * eps as given) and C bf16 [M,N] = act(Z W^T + bias) in one launch -- the LayerNorm of a decode step rides in the
 * projection that consumes it.                                                                         */
int mgx_linear_ln_fwd(const uint16_t* X, const uint16_t* RES, const float* gamma, const float* beta, float eps,
                      const uint16_t* W, const float* bias, uint16_t* C, uint16_t* Z, int M, int N, int K, int act,
                      void* stream);

/* Decode-step fusion (ABI 12; M <= 32 rows): H bf16 [M,K] = table[tok]*sqrt(K) + pe[t] (layers.py:226-229) and
 * C bf16 [M,N] = H W^T + bias in one launch (the embedding rides in the first QKV projection of a decode step).          */
int mgx_decode_embed_linear(const int32_t* tok, const float* table, const float* pe, const int32_t* pos_dev,
                            const uint16_t* W, const float* bias, uint16_t* C, uint16_t* H, int M, int N, int K, int V,
                            void* stream);
/* The three decode-size projections with the weight in MFMA FRAGMENT ORDER (see the fused GRU step below for the layout; rows
 * zero-padded to a multiple of 32): a wave load of weights is 1 KB contiguous instead of 32 bytes of 32 different rows.   */
int mgx_skinny_fwd_frag(const uint16_t* A, const uint16_t* Wf, const float* bias, uint16_t* C, int M, int N, int K, int act,
                        void* stream);                                    /* M <= 32 only: MGX_ERR_SHAPE otherwise (a fragment-ordered
                                                                           * weight must never reach mgx_linear_fwd) */
int mgx_linear_ln_fwd_frag(const uint16_t* X, const uint16_t* RES, const float* gamma, const float* beta, float eps,
                           const uint16_t* Wf, const float* bias, uint16_t* C, uint16_t* Z, int M, int N, int K, int act,
                           void* stream);
int mgx_decode_embed_linear_frag(const int32_t* tok, const float* table, const float* pe, const int32_t* pos_dev,
                                 const uint16_t* Wf, const float* bias, uint16_t* C, uint16_t* H, int M, int N, int K, int V,
                                 void* stream);

/* backward of the above (autograd of the same reference lines):
 * dX bf16 [M,K] = dY bf16 [M,N] @ W bf16 [N,K]; if relu_y (bf16 [M,K]) is given, dX is zeroed where
 * relu_y <= 0 (the backward of a ReLU fused into the producer of this layer's input); if addend
 * (bf16 [M,K]) is given it is added last -- the gradient arriving over the residual connection
 * (layers.py:155,160: out = LN(x + branch(x))) joins the branch's input gradient without a separate pass.
 * N % 8 == 0, K % 8 == 0.                                                                         */
int mgx_linear_dx(const uint16_t* dY, const uint16_t* W, const uint16_t* relu_y, const uint16_t* addend,
                  uint16_t* dX, int M, int N, int K, void* stream);
/* gW f32 [N,K] += dY^T @ X (dY bf16 [M,N], X bf16 [M,K]); gb f32 [N] += column sums of dY (or NULL).
 * Both ACCUMULATE (fp32 atomics), so gradient accumulation over micro-batches needs no extra pass. */
int mgx_linear_dw(const uint16_t* dY, const uint16_t* X, float* gW, float* gb,
                  int M, int N, int K, void* stream);
/* ABI 18: which kernel family mgx_linear_fwd (kind 0: C [M,N], reduction K) or mgx_linear_dx (kind 1 / 2 / 3: no epilogue
 * operand / ReLU mask / residual addend; dX [M,K], reduction N) launches for this shape on `stream` -- no launch, no device
 * access.  For tests and profiles: which of the binary's GEMM kernels a measured or checked call really ran.              */
#define MGX_GEMM_SKINNY 0    /* M <= 32: weight-streaming kernel */
#define MGX_GEMM_TILE128 1   /* 128 x 128 tiles, four waves */
#define MGX_GEMM_RING8 2     /* 256 x 256 tiles, persistent LDS-DMA ring, eight waves */
#define MGX_GEMM_RING4 3     /* 256 x 256 tiles, four waves, generated asm tile statement */
int mgx_linear_kernel_id(int kind, int M, int N, int K, void* stream);
/* the same for up to MGX_DW_MAX_GROUP weights whose dY / X share the row count M (the four projections of one
 * encoder block, layers.py:152-161), in one launch: fewer M-splits fill the chip, so less atomic traffic.  */
#define MGX_DW_MAX_GROUP 8
typedef struct mgx_dw_problem {
    const uint16_t* dY;   /* bf16 [M,N] */
    const uint16_t* X;    /* bf16 [M,K] */
    float* gW;            /* f32 [N,K], accumulated */
    float* gb;            /* f32 [N], accumulated, or NULL */
    int N, K;
} mgx_dw_problem;
/* workspace: caller scratch >= mgx_linear_dw_grouped_workspace(problems, count, M) bytes, 16-byte aligned (fp32 partial
 * tiles of the M-splits; 0 when the group runs on the tiled kernel, NULL is accepted then).  With M >= 4096, M % 32 == 0,
 * the weights whose N x K fills its 256 x 256 tiles to 60 % or more (N, K % 8 == 0; a ragged last tile row / column is
 * allowed: the vocabulary projection 448 x 512) take the split-then-fix-up path -- one fp32 partial tile per (tile, M-split)
 * in the workspace, added into gW in split order by a second pass, no fp32 atomics on gW; a group of ONE weight is the way
 * to send a single large weight gradient (mgx_linear_dw's shapes) down that path.                                         */
size_t mgx_linear_dw_grouped_workspace(const mgx_dw_problem* problems, int count, int M);
int mgx_linear_dw_grouped(const mgx_dw_problem* problems, int count, int M, void* workspace, size_t ws_bytes,
                          void* stream);

/* ---- K12: autoregressive decode with a KV cache (replaces the per-token full-window recompute of
 * network.py:52-77; causal semantics).  The current position t lives in device memory
 * (pos_dev[0]) so that one captured graph of a whole step can be replayed for every token.
 * mgx_decode_embed: out bf16 [B,d] = table[tok]*sqrt(d) + pe[t]            (layers.py:226-229)          */
int mgx_decode_embed(const int32_t* tok, const float* table, const float* pe, const int32_t* pos_dev,
                     uint16_t* out, int B, int d, int V, void* stream);
/* qkv_new bf16 [B,3d] (projection of the token at position t): k_t, v_t are appended to
 * kcache/vcache bf16 [B,h,Lmax,64] (head-major: the workgroup of (b,h) streams one contiguous run of 128-byte rows)
 * at row t, then ctx bf16 [B,d] = softmax_j((q.k_j + q.E[M-1-(t-j)])/8) v_j
 * over j = 0..t.  t < Lmax <= M.                                                                      */
/* Long caches are split over several workgroups per (b,h) whose partial results a second kernel merges: workspace =
 * caller scratch >= mgx_rel_attn_decode_workspace(B, Lmax, d) bytes (0 for short caches: NULL is accepted then).      */
size_t mgx_rel_attn_decode_workspace(int B, int Lmax, int d);
/* key splits per (b,h) for this cache length (1 = no workspace, no merge launch).                                    */
int mgx_rel_attn_decode_splits(int B, int Lmax, int d);
int mgx_rel_attn_decode(const uint16_t* qkv_new, uint16_t* kcache, uint16_t* vcache, const uint16_t* E,
                        const int32_t* pos_dev, uint16_t* ctx, void* workspace, size_t ws_bytes,
                        int B, int Lmax, int d, int M, void* stream);
/* logits bf16 [B,ld] -> next_tok int32 [B] drawn from softmax(logits/temperature) restricted to the top_k
 * most likely ids (0 = all) and then to the smallest set whose mass reaches top_p (1 = all).
 * out_tokens int32 [B,out_ld] (or NULL): column t+1 receives the token; probs_out f32 [B,V] (or NULL)
 * receives the unfiltered softmax.  The draw is a pure function of (seed, t, row).
 * advance != 0: pos_dev[0] += 1 after sampling.  V <= 1024.
 * allow_table (optional, NULL = none): uint32 [V, ceil(V/32)] grammar mask -- bit v of row t set iff token v may follow
 * token t (t = the token next_tok holds on entry); disallowed logits are -inf before temperature/top-k/top-p.     */
int mgx_sample_topk_topp(const uint16_t* logits, int V, int ld, float temperature, int top_k, float top_p,
                         uint64_t seed, int32_t* pos_dev, int32_t* next_tok, int32_t* out_tokens, int out_ld,
                         float* probs_out, int B, int advance, const uint32_t* allow_table, void* stream);
/* The same for rows [row0, row0+B) of a larger batch (every pointer addresses the sub-batch's first row): the draw is a
 * function of (seed, t, row0 + row), so a batch sampled in several independent sub-batches (KV-cache decode runs them on
 * separate streams, each with its own pos_dev) gets the tokens the whole batch would get.                              */
int mgx_sample_topk_topp_rows(const uint16_t* logits, int V, int ld, float temperature, int top_k, float top_p,
                              uint64_t seed, int32_t* pos_dev, int32_t* next_tok, int32_t* out_tokens, int out_ld,
                              float* probs_out, int B, int row0, int advance, const uint32_t* allow_table, void* stream);

/* ---- K13: Event_Melody_RNN step (Event_MelodyRNN/network.py:51-61): the GRU projections run on
 * mgx_linear_fwd; these two kernels are the rest of a step.
 * out bf16 [B,ld] = table bf16 [V,ld][tok] (ld = embedding width padded to a multiple of 64).          */
int mgx_gather_rows(const int32_t* tok, const uint16_t* table, uint16_t* out, int B, int ld, int V, void* stream);
/* torch.nn.GRU cell, gate order (r,z,n): gi = x W_ih^T + b_ih, gh = h W_hh^T + b_hh bf16 [B,3H];
 * h f32 [B,H] is updated in place, h_bf16 [B,H] receives its bf16 copy (input of the next projection). */
int mgx_gru_gates(const uint16_t* gi, const uint16_t* gh, float* h, uint16_t* h_bf16, int B, int H, void* stream);

/* ---- K13b: Event_Melody_RNN training (Event_MelodyRNN/network.py:63-84,109-116 SeqForward/Train; torch.nn.GRU cell).
 * The sequence's input projections are one mgx_linear_fwd per layer; per time step the recurrent projection
 * (mgx_linear_fwd, M = B) and this cell remain:  h_next f32, y bf16 [B,H] = cell(gi, gh bf16 [B,3H], h_prev f32).   */
int mgx_gru_cell_fwd(const uint16_t* gi, const uint16_t* gh, const float* h_prev, float* h_next, uint16_t* y,
                     int B, int H, void* stream);
/* backward of one step: dh = dh_direct (f32) + d_rec (bf16: step t+1's dgh @ W_hh) + dy (bf16: layer above), each
 * may be NULL; gates are recomputed from gi/gh.  Out: dgi, dgh bf16 [B,3H], dh_prev_direct f32 [B,H] = dh * z.   */
int mgx_gru_cell_bwd(const uint16_t* gi, const uint16_t* gh, const float* h_prev, const float* dh_direct,
                     const uint16_t* d_rec, const uint16_t* dy, uint16_t* dgi, uint16_t* dgh,
                     float* dh_prev_direct, int B, int H, void* stream);
/* Fused time step (ABI 14): the recurrent projection and the cell in ONE launch each way (H % 64 == 0).
 * The weights of these three entry points are bf16 in FRAGMENT ORDER: for a matrix W [N,K] (N % 32 == 0, K % 16 == 0) the
 * 16-byte unit ((nt * K/16 + ks) * 64 + lane) holds W[32 nt + lane % 32][16 ks + 8 (lane / 32) .. +7] -- what lane `lane` feeds
 * the MFMA for row tile nt and k-step ks, so a wave load is 1 KB contiguous (ops.pack_frag / melody_rnn._pack build it).
 * forward: gh = h_prev_bf16 W_hh^T + b_hh (W_hh [3H,H] in fragment order, rounded to bf16 and stored in gh_out [B,3H] for the backward),
 *          then h_next f32, y bf16 [B,H] = cell(gi, gh, h_prev) exactly as mgx_gru_cell_fwd.                               */
int mgx_gru_step_fwd(const uint16_t* gi, const uint16_t* h_prev_bf16, const float* h_prev, const uint16_t* Whh,
                     const float* bhh, float* h_next, uint16_t* y, uint16_t* gh_out, int B, int H, void* stream);
/* sampling step of one layer: gi = x W_ih^T + b_ih (x bf16 [B,Kx], W_ih [3H,Kx] and W_hh in fragment order, Kx % 64 == 0), gh as above, the cell -- one
 * launch instead of two projections + mgx_gru_gates.  NOT in place: h_next / y must differ from h_prev / h_prev_bf16.     */
int mgx_gru_step_x_fwd(const uint16_t* x, const uint16_t* Wih, const float* bih, int Kx, const uint16_t* h_prev_bf16,
                       const float* h_prev, const uint16_t* Whh, const float* bhh, float* h_next, uint16_t* y, int B, int H,
                       void* stream);
/* backward of step t: d_rec = dgh_next W_hh (dgh_next bf16 [B,3H] = step t+1's dgh, NULL at the last step; WhhT = W_hh^T
 * [H,3H] in fragment order), then mgx_gru_cell_bwd with it: dgi, dgh bf16 [B,3H], dh_out f32 [B,H] = dh * z.  final != 0: no cell,
 * dh_out = dh_direct + d_rec (the gradient of the layer's initial state; gi/gh/h_prev/dgi/dgh may be NULL).           */
int mgx_gru_step_bwd(const uint16_t* gi, const uint16_t* gh, const float* h_prev, const float* dh_direct,
                     const uint16_t* dgh_next, const uint16_t* WhhT, const uint16_t* dy, uint16_t* dgi, uint16_t* dgh,
                     float* dh_out, int B, int H, int final, void* stream);
/* inverted dropout on a bf16 buffer (nn.GRU's inter-layer dropout); stateless mask = f(seed, index): calling it on
 * the gradient with the same (p, seed) is its backward.  n % 8 == 0.                                             */
int mgx_dropout_bf16(const uint16_t* x, uint16_t* out, size_t n, float p_drop, uint64_t seed, void* stream);
/* dst f32 [V,cols] row idx[r] += src bf16 [n,ld] row r, columns 0..cols-1 (nn.Embedding backward).                */
int mgx_scatter_add_rows(const int32_t* idx, const uint16_t* src, float* dst, int n, int ld, int cols, int V,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
