/*
 * vqa_mi355x.h -- C ABI of libvqa_mi355x.so: the MI355X (gfx950) kernels behind the CoR2 / ODA
 * hot path of bupt-cist/vqa-playground-pytorch.
 *
 * The reference has no FFI: its boundary is the Python nn.Module surface of config/CoR2.py /
 * config/ODA.py.  Each entry point below replaces one *sequence of ATen ops* in the
 * reference, cited per function as file:line under /root/reference.  The Python host
 * (vqa_playground_pytorch_amd/ops.py) binds these through ctypes from torch.autograd.Function
 * forward/backward; INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer to fp32 unless stated;
 *     all tensors are row-major and dense unless a stride argument says otherwise;
 *   - the CALLER allocates every buffer, including workspaces (sizes from the *_workspace_bytes
 *     queries); the library never allocates, frees or synchronises, so every launcher can be
 *     captured into a hipGraph;
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*);
 *   - return value 0 = success, negative = error (VQA_E_*); the text of the last error of the
 *     calling thread is available from vqa_last_error(); no C++ exception crosses the ABI;
 *   - no global mutable state: re-entrant, safe to call from one host thread per device.
 *
 * Symbols: B samples, N regions per sample, D region feature width, G glimpses, L low dim,
 * H hidden dim, R rank, M = B*N rows.
 */
#ifndef VQA_MI355X_H
#define VQA_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQA_ABI_VERSION 5

#define VQA_OK 0
#define VQA_E_BADARG (-1)      /* null pointer, non-positive size, size over a documented limit */
#define VQA_E_UNSUPPORTED (-2) /* alignment / shape the kernels do not cover */
#define VQA_E_LAUNCH (-3)      /* HIP reported an error at launch */

typedef void* vqa_stream_t; /* hipStream_t */
typedef uint16_t vqa_bf16_t; /* storage of one bfloat16 (upper half of an IEEE fp32) */

int vqa_version(void);
const char* vqa_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * K1  pairwise relation build + alpha-weighted reduce.
 * Replaces config/CoR2.py:191-199 (decare_cat: two .repeat() + two putils.bmul + add, a
 * [B,N,N,D] tensor) and config/CoR2.py:216 ((alpha1[0].view(b,36,1,1) * v2_cat).sum(1)).
 *
 *   v2[b,j,:] = sum_i alpha[b,i] * ( v[b,i,:]*q1[b,:] + v[b,j,:]*q2[b,:] )
 *
 * v [B,N,D]; q1,q2 [B,D]; alpha: element (b,i) at alpha[(b*N+i)*alpha_stride] so glimpse 0 of a
 * [B,N,G] attention tensor is passed without a copy (alpha_stride = G); v2 [B,N,D].
 * mode 0 = pairwise: the N*N inner sum is evaluated term by term from the LDS-staged region tile
 *          (the reference's summation structure);
 * mode 1 = factored: q1*(sum_i alpha_i v_i) + (sum_i alpha_i)*q2*v_j -- the same value, one pass.
 * Limits: D % 4 == 0, 16-byte aligned v/q1/q2/v2; N <= 144 in mode 0, N <= 4096 in mode 1.
 * ------------------------------------------------------------------------------------------- */
int vqa_pairwise_relation_reduce_fwd(const float* v, const float* q1, const float* q2,
                                     const float* alpha, int alpha_stride, float* v2,
                                     int B, int N, int D, int mode, vqa_stream_t stream);

/* Backward of K1.  g_v2 = dL/dv2 [B,N,D]; g_v2_b = a second gradient tensor of the same shape that is
 * added to it on the fly, or NULL (CoR2's v2 has two consumers, config/CoR2.py:218-220; handing both
 * gradients over saves the B*N*D-element add in front of this kernel).  Outputs: d_alpha [B,N] dense,
 * d_q1 [B,D], d_q2 [B,D], d_v [B,N,D] or NULL (v is a leaf in CoR2).  d_alpha is accumulated across the
 * D-chunks of a sample with float atomics after being zeroed on `stream` by this call. */
int vqa_pairwise_relation_reduce_bwd(const float* v, const float* q1, const float* q2,
                                     const float* alpha, int alpha_stride, const float* g_v2,
                                     const float* g_v2_b, float* d_alpha, float* d_q1, float* d_q2,
                                     float* d_v, int B, int N, int D, vqa_stream_t stream);

/* K1, closed form with the pooled feature given.  s = sum_i alpha_i v_i is glimpse 0 of the first attention's
 * pooled output and a softmax alpha sums to 1, so the relation step (config/CoR2.py:191-199 + :216) is the
 * per-sample affine map v2[b,n,:] = t[b,:] + c2[b,:]*v[b,n,:] with t = q1*s, c2 = (sum_i alpha_i)*q2; its one
 * consumer that needs it materialised is the second compress layer, which applies dropout first
 * (config/CoR2.py:72-75 at :218):
 *
 *   out[b,n,:] = keep(b,n,:) * (t[b,:] + c2[b,:] * v[b,n,:])
 *
 * keep() = 1 when p_drop == 0, else 0 or 1/(1-p_drop) from the counter-based generator keyed by
 * (seed [+ *seed_ptr], (b*N+n)*D + d) -- vqa_linear_dropout_mask(B*N, D, ...) writes the same mask.
 * v, out [B,N,D]; t, c2 [B,D] fp32.  Backward: g = dL/dout -> d_t = sum_n keep*g, d_c2 = sum_n keep*g*v
 * ([B,D] fp32, overwritten) and d_v = c2*keep*g [B,N,D] or NULL.  Limits: D % 4 == 0, B*N*D < 2^32. */
int vqa_relation_apply_fwd(const float* v, const float* t, const float* c2, float* out, float p_drop,
                           uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D,
                           vqa_stream_t stream);
int vqa_relation_apply_bwd(const float* v, const float* c2, const float* g, float* d_t, float* d_c2,
                           float* d_v, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                           int B, int N, int D, vqa_stream_t stream);
int vqa_relation_apply_fwd_bf16(const vqa_bf16_t* v, const float* t, const float* c2, vqa_bf16_t* out,
                                float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D,
                                vqa_stream_t stream);
int vqa_relation_apply_bwd_bf16(const vqa_bf16_t* v, const float* c2, const vqa_bf16_t* g, float* d_t,
                                float* d_c2, vqa_bf16_t* d_v, float p_drop, uint64_t seed,
                                const uint64_t* seed_ptr, int B, int N, int D, vqa_stream_t stream);

/* K1 -> K5 fusion, backward.  When relation_apply's output x = keep * (t + c2 * v) feeds the second region projection
 * (compress_v2, config/CoR2.py:218) and nothing else, the projection's data gradient is only ever reduced to d_t and d_c2:
 *   dx[m,:] = sum_l gz[m,l] w[l,:];   d_t[b,:] = sum_n keep(m,:) dx[m,:];   d_c2[b,:] = sum_n keep(m,:) dx[m,:] v[m,:]
 * with gz [B*N, L] the gradient at the projection's pre-activation (relu gate applied), w [L, D] its weight
 * (nn.Conv1d(D, L, 1).weight), v [B,N,D], keep = the (p_drop, seed) mask of vqa_relation_apply_fwd.  ONE kernel: the
 * GEMM tile is four whole samples (144 rows) and is reduced in registers; dx (B*N*D*4 bytes) is never written.
 * vqa_relation_projection_dgrad_supported(): N == 36, D % 64 == 0, even L >= 32, tensors < 4 GiB. */
int vqa_relation_projection_dgrad_supported(int B, int N, int D, int L);
int vqa_relation_projection_dgrad(const float* gz, const float* w, const float* v, float* d_t, float* d_c2,
                                  float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D, int L,
                                  vqa_stream_t stream);

/* K1 with the region tensors (v, v2, g_v2, d_v) stored as bf16 -- the mixed-precision path of BASELINE
 * configs[4] (bf16 storage, fp32 arithmetic and accumulation; q1, q2, alpha and their gradients stay
 * fp32).  Same semantics and limits, with 8-byte instead of 16-byte alignment of the bf16 tensors. */
int vqa_pairwise_relation_reduce_fwd_bf16(const vqa_bf16_t* v, const float* q1, const float* q2,
                                          const float* alpha, int alpha_stride, vqa_bf16_t* v2,
                                          int B, int N, int D, int mode, vqa_stream_t stream);
int vqa_pairwise_relation_reduce_bwd_bf16(const vqa_bf16_t* v, const float* q1, const float* q2,
                                          const float* alpha, int alpha_stride, const vqa_bf16_t* g_v2,
                                          const vqa_bf16_t* g_v2_b, float* d_alpha, float* d_q1,
                                          float* d_q2, vqa_bf16_t* d_v, int B, int N, int D,
                                          vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3  softmax over regions + attention-weighted region pooling.
 * Replaces F.softmax(x, dim=1) inside MyConv1d (config/CoR2.py:83-87 as configured at :132) and
 * putils.bmatmul(x_att.transpose(1,2), inputs) (config/CoR2.py:142; putils/__init__.py:89-95).
 *
 *   alpha[b,n,g] = softmax_n(logits[b,n,g]);   pooled[b,g,:] = sum_n alpha[b,n,g] * v[b,n,:]
 *
 * logits, alpha [B,N,G]; v [B,N,D]; pooled [B,G,D].  Limits: 1 <= G <= 8, N <= 1024, D % 4 == 0.
 * ------------------------------------------------------------------------------------------- */
int vqa_softmax_attention_pool_fwd(const float* logits, const float* v, float* alpha, float* pooled,
                                   int B, int N, int D, int G, vqa_stream_t stream);

/* Backward of K3.  d_pooled [B,G,D]; d_alpha_ext [B,N,G] or NULL = gradient that reaches alpha
 * directly (CoR2 feeds alpha1[...,0] to K1); outputs d_logits [B,N,G], d_v [B,N,D] or NULL.
 * Limit: G*D*4 + 4*N*G*4 bytes of LDS <= 160 KiB. */
int vqa_softmax_attention_pool_bwd(const float* alpha, const float* v, const float* d_pooled,
                                   const float* d_alpha_ext, float* d_logits, float* d_v,
                                   int B, int N, int D, int G, vqa_stream_t stream);

/* K3 with what MyATT does to the pooled features next folded in (config/CoR2.py:143-147: every glimpse's MyLinear
 * starts with F.dropout(p=0.5) on its slice of `pooled`), plus the undropped glimpse 0 that CoR2's relation step reads
 * (config/CoR2.py:216 with alpha1[..., 0]):
 *   pooled[b,g,:] = keep(b,g,:) * sum_n alpha[b,n,g] v[b,n,:]        keep: the counter-hash dropout of K2 / K5 over the
 *                                                                    element index of the [B,G,D] tensor (seed, seed_ptr
 *                                                                    as there); p_drop = 0: plain K3
 *   first[b,:]    = sum_n alpha[b,n,0] v[b,n,:]                       [B,D] or NULL
 * Backward: d_pooled is the gradient of the DROPPED pooled (the mask is re-applied in the load), d_first [B,D] or NULL
 * the gradient of `first`; the rest as vqa_softmax_attention_pool_bwd.  Limit with dropout: B*G*D < 2^32. */
int vqa_softmax_attention_pool_drop_fwd(const float* logits, const float* v, float* alpha, float* pooled,
                                        float* first, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                        int B, int N, int D, int G, vqa_stream_t stream);
int vqa_softmax_attention_pool_drop_bwd(const float* alpha, const float* v, const float* d_pooled,
                                        const float* d_first, const float* d_alpha_ext, float* d_logits,
                                        float* d_v, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B,
                                        int N, int D, int G, vqa_stream_t stream);
int vqa_softmax_attention_pool_drop_fwd_bf16(const float* logits, const vqa_bf16_t* v, float* alpha, float* pooled,
                                             float* first, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                             int B, int N, int D, int G, vqa_stream_t stream);
int vqa_softmax_attention_pool_drop_bwd_bf16(const float* alpha, const vqa_bf16_t* v, const float* d_pooled,
                                             const float* d_first, const float* d_alpha_ext, float* d_logits,
                                             vqa_bf16_t* d_v, float p_drop, uint64_t seed,
                                             const uint64_t* seed_ptr, int B, int N, int D, int G,
                                             vqa_stream_t stream);

/* K3 with v / d_v stored as bf16 (logits, alpha, pooled and their gradients stay fp32). */
int vqa_softmax_attention_pool_fwd_bf16(const float* logits, const vqa_bf16_t* v, float* alpha,
                                        float* pooled, int B, int N, int D, int G, vqa_stream_t stream);
int vqa_softmax_attention_pool_bwd_bf16(const float* alpha, const vqa_bf16_t* v, const float* d_pooled,
                                        const float* d_alpha_ext, float* d_logits, vqa_bf16_t* d_v,
                                        int B, int N, int D, int G, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K3a  attention logits: the dropout + 1x1 conv in front of the softmax of MyATT.
 * Replaces MyConv1d(fuse_dim, glimpses, 1, 1, p=0.5).forward up to its activation (config/CoR2.py:72-82 as
 * configured at :132): F.dropout, two transposes, nn.Conv1d with G output channels.
 *
 *   logits[m,g] = bias[g] + sum_k w[g,k] * keep(m,k) * x[m,k]            (m = b*N + n)
 *
 * keep() as in K2 / K5: 1 when p_drop == 0, else 0 or 1/(1-p_drop) from the counter-based generator keyed by
 * (seed [+ *seed_ptr], m*K + k) -- vqa_linear_dropout_mask(M, K, ...) writes the same mask for a test.
 * x [M,K] with row stride ldx (fp32, or bf16 in the _bf16 forms, where ldx may be K padded to 64 with zeros);
 * w [G,K]; bias [G]; logits [M,G] fp32.  Limits: G <= 8, K even and <= 512.
 * Backward: d_logits [M,G] -> d_x [M,ldx] (or NULL), d_w [G,K], d_bias [G]; workspace:
 * vqa_attention_logits_bwd_workspace_bytes(M, K, G) bytes; workgroup partials are added in a fixed order.
 * ------------------------------------------------------------------------------------------- */
int vqa_attention_logits_fwd(const float* x, int ldx, const float* w, const float* bias, float* logits,
                             float p_drop, uint64_t seed, const uint64_t* seed_ptr, int M, int K, int G,
                             vqa_stream_t stream);
int vqa_attention_logits_fwd_bf16(const vqa_bf16_t* x, int ldx, const float* w, const float* bias,
                                  float* logits, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                  int M, int K, int G, vqa_stream_t stream);
size_t vqa_attention_logits_bwd_workspace_bytes(int M, int K, int G);
int vqa_attention_logits_bwd(const float* x, int ldx, const float* w, const float* d_logits, float* d_x,
                             float* d_w, float* d_bias, void* workspace, size_t workspace_bytes,
                             float p_drop, uint64_t seed, const uint64_t* seed_ptr, int M, int K, int G,
                             vqa_stream_t stream);
int vqa_attention_logits_bwd_bf16(const vqa_bf16_t* x, int ldx, const float* w, const float* d_logits,
                                  vqa_bf16_t* d_x, float* d_w, float* d_bias, void* workspace,
                                  size_t workspace_bytes, float p_drop, uint64_t seed,
                                  const uint64_t* seed_ptr, int M, int K, int G, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K4  low-rank bilinear (Mutan) fusion on the fp32 MFMA tile engine.
 * Replaces putils.MutanFusion.forward (putils/__init__.py:232-238): R x { Linear(in1->H) on the
 * region side, putils.bmul against the question-side factor, total += }.
 *
 *   h1[m,r,:] = x[m,:] W1_r^T + b1_r          (m = b*N + n)
 *   out[m,:]  = sum_r h1[m,r,:] * h2[b,r,:]   (h2 = the question-side Linear2_r(x2), [B,R,H])
 *
 * x [M,L] with row stride ldx; w1[r] -> [H,L] dense (R host-side entries of device pointers, the
 * reference keeps one nn.Linear per rank); b1[r] -> [H]; h2 [B,R,H]; out [M,H] dense;
 * h1 [M,R,H] or NULL (saved for backward).  N = rows per sample (1 for 2-D inputs).
 * Limits: R <= 8; L, H, ldx even; 8-byte aligned pointers.
 * ------------------------------------------------------------------------------------------- */
int vqa_lowrank_bilinear_fusion_fwd(const float* x, int ldx, const float* const* w1,
                                    const float* const* b1, const float* h2, float* out, float* h1,
                                    int B, int N, int L, int H, int R, vqa_stream_t stream);

size_t vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(int B, int N, int L, int H, int R);

/* Backward of K4.  g = dL/dout [M,H]; h1 as saved by the forward.  Outputs: d_x [M,L] dense or
 * NULL; d_w1[r] -> [H,L]; d_b1[r] -> [H]; d_h2 [B,R,H].  All outputs are overwritten (not
 * accumulated).  workspace: vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(...) bytes. */
int vqa_lowrank_bilinear_fusion_bwd(const float* x, int ldx, const float* const* w1, const float* h2,
                                    const float* h1, const float* g, float* d_x, float* const* d_w1,
                                    float* const* d_b1, float* d_h2, void* workspace,
                                    size_t workspace_bytes, int B, int N, int L, int H, int R,
                                    vqa_stream_t stream);

/* K4, rank-folded form (csrc/bilinear_folded.hip).  The question-side factor is the same for every region of
 * a sample, so the sum over ranks of putils.MutanFusion.forward (putils/__init__.py:232-238) commutes with the
 * contraction:  out[b,n,:] = Weff_b x[b,n,:] + c_b,  Weff_b[j,k] = sum_r h2[b,r,j] W1_r[j,k],
 * c_b[j] = sum_r h2[b,r,j] b1_r[j]  -- one contraction per sample against a weight built on the fly: 1/R of the
 * matrix work and no [M,R,H] intermediate.  Same arguments as vqa_lowrank_bilinear_fusion_fwd without h1.
 * Limits: N <= 112 regions per sample, R <= 4 (R <= 2 above 48 regions), L / H / ldx even; vqa_lowrank_bilinear_fusion_folded_supported()
 * returns 1 when a shape qualifies (callers use the unfolded entry points otherwise). */
int vqa_lowrank_bilinear_fusion_folded_supported(int B, int N, int L, int H, int R);
int vqa_lowrank_bilinear_fusion_folded_fwd(const float* x, int ldx, const float* const* w1,
                                           const float* const* b1, const float* h2, float* out, int B,
                                           int N, int L, int H, int R, vqa_stream_t stream);

/* Backward of the rank-folded form: nothing saved by the forward.  dx = Weff_b^T g on the folded kernel; the
 * per-sample product P_b = g_b^T x_b yields dW1_r = sum_b h2[b,r,:] (.) P_b, db1_r, and
 * dh2[b,r,h] = sum_l W1_r[h,l] P_b[h,l] + b1_r[h] sum_n g[b,n,h] in one pass (fixed-order slab reductions).
 * Arguments as vqa_lowrank_bilinear_fusion_bwd with b1 in place of h1. */
size_t vqa_lowrank_bilinear_fusion_folded_bwd_workspace_bytes(int B, int N, int L, int H, int R);
int vqa_lowrank_bilinear_fusion_folded_bwd(const float* x, int ldx, const float* const* w1,
                                           const float* const* b1, const float* h2, const float* g,
                                           float* d_x, float* const* d_w1, float* const* d_b1, float* d_h2,
                                           void* workspace, size_t workspace_bytes, int B, int N, int L,
                                           int H, int R, vqa_stream_t stream);
/* The same with gate_dx != 0: d_x[b,n,l] is zeroed where x[b,n,l] <= 0.  x is then the relu output of the layer in
 * front of the fusion (compress_v / compress_v2, config/CoR2.py:213-214,:218-219: F.relu at :86 feeds MutanFusion
 * directly), and relu's own backward -- grad * (y > 0) -- is applied here, in the store of d_x, instead of as a masked
 * operand in that layer's weight- and data-gradient kernels.  Needs ldx == L. */
int vqa_lowrank_bilinear_fusion_folded_bwd_gated(const float* x, int ldx, const float* const* w1,
                                                 const float* const* b1, const float* h2, const float* g,
                                                 float* d_x, float* const* d_w1, float* const* d_b1,
                                                 float* d_h2, void* workspace, size_t workspace_bytes, int B,
                                                 int N, int L, int H, int R, int gate_dx, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Mixed-precision (bf16) side -- BASELINE configs[4] "CoR2 bf16, 100x2048 dense regions": bf16 storage and
 * bf16 MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 accumulation, fp32 master weights.
 * Operand contract: feature dims are zero-padded to a multiple of 64 by the caller (H of K4 to a multiple
 * of 256), so the kernels carry no tail handling; row counts (M = B*N) are free.
 * ------------------------------------------------------------------------------------------- */

/* fp32 [batch, rows, cols] dense -> bf16 scattered to dst[b*dst_batch_stride + r*dst_row_stride +
 * c*dst_col_stride]; with zero_fill != 0 the destination (dst_elems elements) is zeroed first, which writes
 * the pads (pass 0 when several calls interleave into one destination that was zeroed by the first).  Used
 * for the padded / transposed bf16 shadow copies of the fp32 master weights (nn.Linear weights of
 * putils/__init__.py:16-33) and to narrow activations. */
int vqa_pack_bf16(const float* src, int batch, int rows, int cols, vqa_bf16_t* dst,
                  long dst_batch_stride, long dst_row_stride, long dst_col_stride, size_t dst_elems,
                  int zero_fill, vqa_stream_t stream);

/* c[M,N] (bf16, row stride ldc) = act(a[M,K] * b[N,K]^T + bias[N]); bias fp32 or NULL; act 0 none, 1 relu.
 * The nn.Linear / 1x1 nn.Conv1d contraction of MyLinear / MyConv1d (config/CoR2.py:56-122) in bf16.
 * Limits: K % 64 == 0; lda, ldb % 8 == 0; 16-byte aligned a, b. */
int vqa_gemm_bf16_nt(const vqa_bf16_t* a, int lda, const vqa_bf16_t* b, int ldb, const float* bias,
                     vqa_bf16_t* c, int ldc, int M, int N, int K, int act, vqa_stream_t stream);

/* c[N1,N2] (fp32, dense) = a[K,N1]^T * b[K,N2] -- the weight-gradient contraction over the batch rows, split
 * over K into fp32 slabs that are reduced in a fixed order (bitwise reproducible).
 * Limits: N1, N2, lda, ldb % 8 == 0; 16-byte aligned pointers. */
size_t vqa_gemm_bf16_tn_workspace_bytes(int K, int N1, int N2);
int vqa_gemm_bf16_tn(const vqa_bf16_t* a, int lda, const vqa_bf16_t* b, int ldb, float* c, void* workspace,
                     size_t workspace_bytes, int K, int N1, int N2, vqa_stream_t stream);

/* K4 in bf16 (putils/__init__.py:232-238).  x [M,L] bf16 dense; w1 [R,H,L] bf16 (the R region-side weights,
 * padded and stacked); b1 [R,H] fp32; h2 [B,R,H] fp32; out [M,H] bf16; h1 [M,R,H] bf16 or NULL.
 * Limits: L % 64 == 0, H % 256 == 0, R <= 8. */
int vqa_lowrank_bilinear_fusion_fwd_bf16(const vqa_bf16_t* x, const vqa_bf16_t* w1, const float* b1,
                                         const float* h2, vqa_bf16_t* out, vqa_bf16_t* h1,
                                         int B, int N, int L, int H, int R, vqa_stream_t stream);

size_t vqa_lowrank_bilinear_fusion_bwd_bf16_workspace_bytes(int B, int N, int L, int H, int R);

/* Backward of the bf16 K4.  w1t [L, R*H] bf16 = the weights transposed with the rank axis concatenated
 * (w1t[l, r*H+h] = w1[r,h,l]; only read when d_x != NULL); g = dL/dout [M,H] bf16.  Outputs: d_x [M,L] bf16
 * or NULL; d_w1 [R,H,L] fp32; d_b1 [R,H] fp32; d_h2 [B,R,H] fp32 (all overwritten). */
int vqa_lowrank_bilinear_fusion_bwd_bf16(const vqa_bf16_t* x, const vqa_bf16_t* w1t, const float* h2,
                                         const vqa_bf16_t* h1, const vqa_bf16_t* g, vqa_bf16_t* d_x,
                                         float* d_w1, float* d_b1, float* d_h2, void* workspace,
                                         size_t workspace_bytes, int B, int N, int L, int H, int R,
                                         vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K2  object-difference attention logits (ODA).
 * Replaces the 36x36 python loop config/ODA.py:216-222 (vq[b,i,j*L+d] = (vl[b,i,d]-vl[b,j,d])*ql[b,d],
 * a [B,N,N*L] tensor) and the dropout + 1x1 conv of MyConv1d on it (config/ODA.py:149, :89-105):
 *
 *   logits[b,i,g] = bias[g] + sum_{j,d} w[g,j*L+d] * keep(b,i,j*L+d) * (vl[b,i,d]-vl[b,j,d]) * ql[b,d]
 *
 * keep() = 1 when p_drop == 0, else 0 or 1/(1-p_drop) from a counter-based generator keyed by
 * (seed, element index).  seed_ptr (optional, DEVICE memory): when not NULL the effective seed is *seed_ptr + seed,
 * read at run time -- a captured hipGraph can then be replayed with a fresh mask per step by updating that word; vqa_object_difference_dropout_mask writes the same mask as fp32 [B,N,N*L]
 * so a test can hand it to the oracle.  vl [B,N,L]; ql [B,L]; w [G,N*L]; bias [G]; logits [B,N,G].
 * Limits: G <= 8, N <= 128, L <= 1024.
 * ------------------------------------------------------------------------------------------- */
int vqa_object_difference_attention_fwd(const float* vl, const float* ql, const float* w,
                                        const float* bias, float* logits, float p_drop,
                                        uint64_t seed, const uint64_t* seed_ptr, int B, int N, int L, int G,
                                        vqa_stream_t stream);

size_t vqa_object_difference_attention_bwd_workspace_bytes(int B, int N, int L, int G);

/* Backward of K2: d_logits [B,N,G] -> d_vl [B,N,L], d_ql [B,L], d_w [G,N*L], d_bias [G]
 * (all overwritten).  Same (p_drop, seed) as the forward regenerates the mask. */
int vqa_object_difference_attention_bwd(const float* vl, const float* ql, const float* w,
                                        const float* d_logits, float* d_vl, float* d_ql, float* d_w,
                                        float* d_bias, void* workspace, size_t workspace_bytes,
                                        float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N,
                                        int L, int G, vqa_stream_t stream);

int vqa_object_difference_dropout_mask(float* mask, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B,
                                       int N, int L, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * K5  fused dropout + linear + bias + activation on the fp32 MFMA tile engine.
 * Replaces MyConv1d.forward with kernel_size 1 (config/CoR2.py:72-88: F.dropout, transpose, nn.Conv1d,
 * transpose, F.relu) as used by compress_v / compress_v2 (config/CoR2.py:168-169,213,218) -- and, being a
 * plain  act(drop(x) W^T + b), MyLinear.forward (config/CoR2.py:106-121) as well.
 *
 *   y[m,:] = act( (x[m,:] * keep(m,:)) W^T + bias )        x [M,K] row stride ldx; w [N,K]; y [M,N] dense
 *
 * act: 0 = none, 1 = relu.  keep() as in K2: 1 when p_drop == 0, else 0 or 1/(1-p) from the counter hash of
 * (seed, m*K+k) -- seed / seed_ptr as in K2; vqa_linear_dropout_mask writes it as fp32 [M,K] for tests.  bias may be NULL.
 * Limits: K, N, ldx even; 8-byte aligned pointers.
 * ------------------------------------------------------------------------------------------- */
int vqa_linear_act_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int M, int K,
                       int N, int act, float p_drop, uint64_t seed, const uint64_t* seed_ptr, vqa_stream_t stream);

size_t vqa_linear_act_bwd_workspace_bytes(int M, int K, int N);

/* Backward of K5.  y = the forward output (its sign is the relu mask); gy = dL/dy [M,N].  Outputs (overwritten):
 * d_x [M,K] dense or NULL, d_w [N,K], d_b [N] or NULL.  Same (p_drop, seed) as the forward. */
int vqa_linear_act_bwd(const float* x, int ldx, const float* w, const float* y, const float* gy, float* d_x,
                       float* d_w, float* d_b, void* workspace, size_t workspace_bytes, int M, int K, int N,
                       int act, float p_drop, uint64_t seed, const uint64_t* seed_ptr, vqa_stream_t stream);

int vqa_linear_dropout_mask(float* mask, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int M, int K,
                            vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Column sums: out[n] = sum_m x[m,n] -- the bias gradient grad_output.sum(0) of every nn.Linear / 1x1 nn.Conv1d
 * on the path (putils/__init__.py:16-33; config/CoR2.py:56-122).  x [M,N] with row stride ld (fp32, or bf16 in
 * the _bf16 form); out [N] fp32; workspace: vqa_column_sum_workspace_bytes(M, N) bytes (0 for short matrices).
 * Row slabs are added in a fixed order (bitwise reproducible; no atomics, no memset -- safe to replay).
 * ------------------------------------------------------------------------------------------- */
size_t vqa_column_sum_workspace_bytes(int M, int N);
int vqa_column_sum(const float* x, int ld, float* out, void* workspace, size_t workspace_bytes,
                   int M, int N, vqa_stream_t stream);
int vqa_column_sum_bf16(const vqa_bf16_t* x, int ld, float* out, void* workspace, size_t workspace_bytes,
                        int M, int N, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Epilogue / prologue of the batched [B,.] layers (G same-shaped MyLinear / putils.Linear run as one batched
 * library GEMM: config/CoR2.py:94-122, putils/__init__.py:16-33).
 *   vqa_bias_act:       out = act(y[g,b,:] + bias[g,:]);  y [G,B,A]; bias row g at bias + g*bias_stride or NULL;
 *                       out [G,B,A] (group_first != 0) or [B,G,A].  act: 0 none, 1 relu, 2 sigmoid.
 *   vqa_act_bwd_colsum: gz[g,b,:] = gy * act'(out) written [G,B,A]; d_bias row g (at d_bias + g*d_bias_stride) =
 *                       sum_b gz[g,b,:] (fixed order), or NULL; gy and out in the layout vqa_bias_act wrote (group_first).
 * ------------------------------------------------------------------------------------------- */
int vqa_bias_act(const float* y, const float* bias, int bias_stride, float* out, int G, int B, int A,
                 int act, int group_first, vqa_stream_t stream);
int vqa_act_bwd_colsum(const float* gy, const float* out, float* gz, float* d_bias, int d_bias_stride, int G, int B,
                       int A, int act, int group_first, vqa_stream_t stream);

/* MyLinear's input dropout (config/CoR2.py:108-110, F.dropout(x, p, training)) on the [B,.]-sized tensors of the path, with
 * the counter-hash mask of K2 / K5 (seed / seed_ptr as there) instead of a Philox stream; G > 1 = the G independent draws
 * that G same-shaped MyLinear layers make over ONE shared input (CoR2's four question projections, config/CoR2.py:205,
 * :193-194,:230):
 *   out[g,m,k] = x[m,k] * keep((g M + m) K + k)       x [M,K] row stride ldx; out [G,M,K] dense
 *   d_x[m,k]   = sum_g keep(.) * gy[g,m,k]             gy [G,M,K], d_x [M,K] dense
 * vqa_linear_dropout_mask(mask, p, seed, seed_ptr, G*M, K) writes the same keep() as fp32.  Limit: G*M*K < 2^32. */
int vqa_dropout_groups_fwd(const float* x, int ldx, float* out, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                           int G, int M, int K, vqa_stream_t stream);
int vqa_dropout_groups_bwd(const float* gy, float* d_x, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int G,
                           int M, int K, vqa_stream_t stream);

/* Rank sum of the vector-vector Mutan fusion (putils.MutanFusion.forward, putils/__init__.py:232-238, with 2-D inputs --
 * fusion_final of config/CoR2.py:182 / config/ODA.py:197):  out[b,:] = sum_r h1[b,r,:] * h2[b,r,:]  (the reference's
 * bmul + `total +=` over the ranks);  backward d_h1 = g * h2, d_h2 = g * h1.  h1 / h2 [B,R,H] dense, H even. */
int vqa_rank_product_fwd(const float* h1, const float* h2, float* out, int B, int R, int H, vqa_stream_t stream);
int vqa_rank_product_bwd(const float* g, const float* h1, const float* h2, float* d_h1, float* d_h2, int B,
                         int R, int H, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Gate math of the BayesianGRU question encoder, one call per time step each way (putils/__init__.py:704-731
 * with the cell of :604-646 and the sequence-shared dropout of :503-539); the three recurrent GEMMs of a
 * step run as one batched library GEMM a[g] = hm[g] W_g^T in the host code (ops.GruSequence).
 *   forward : r = sigmoid(gi_r[t] + a_r); i = sigmoid(gi_i[t] + a_i); n = af(gi_n[t] + r*a_n);
 *             h_new = (1-i)*n + i*h_prev; hm_next[g] = h_new * masks[g] (the next step's GEMM inputs; NULL at
 *             the last step); r, i, n, a_n saved for backward.
 *   backward: dh = d_out_t + carry_in + sum_g dhm[g]*masks[g] (each may be NULL = 0); writes gz[g] = gradient
 *             at a[g], d_gi[:, :, t, :] and carry_out = dh * i.
 * gi, d_gi [3,B,T,H]; a, dhm, masks [3,B,H] (masks NULL = no dropout); h_prev, h_new, r, i, n, an, d_out_t,
 * carry [B,H]; hm_next / gz: three [B,H] slabs hist_group_stride elements apart (slot t of a [3,T,B,H]
 * history).  af: 1 relu, 3 tanh.  Limit: H % 4 == 0.
 * ------------------------------------------------------------------------------------------- */
int vqa_gru_gates_fwd(const float* gi, const float* a, const float* h_prev, const float* masks,
                      float* h_new, float* hm_next, size_t hist_group_stride, float* r_s, float* i_s,
                      float* n_s, float* an_s, int B, int T, int H, int t, int af, vqa_stream_t stream);
int vqa_gru_gates_bwd(const float* d_out_t, const float* carry_in, const float* dhm, const float* masks,
                      const float* r_s, const float* i_s, const float* n_s, const float* an_s,
                      const float* h_prev, float* gz, size_t hist_group_stride, float* d_gi,
                      float* carry_out, int B, int T, int H, int t, int af, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * KLD-sum loss on soft targets, with its gradient.
 * Replaces MyLoss (train.py:536-544): KLDivLoss(size_average=False)(F.log_softmax(logits), target).
 *
 *   loss[0]       = sum_{b,c} target[b,c] * (log target[b,c] - log_softmax(logits)[b,c])     (0 log 0 = 0)
 *   d_logits[b,c] = softmax(logits)[b,c] * sum_c target[b,c] - target[b,c]        (dloss/dlogits; may be NULL)
 *
 * logits, target, d_logits [B,C]; loss [1]; workspace: vqa_kld_sum_loss_workspace_bytes(B) bytes.
 * Row losses are added in a fixed order (bitwise reproducible).  Limit: C <= 4096.
 * ------------------------------------------------------------------------------------------- */
size_t vqa_kld_sum_loss_workspace_bytes(int B);
int vqa_kld_sum_loss(const float* logits, const float* target, float* loss, float* d_logits,
                     void* workspace, size_t workspace_bytes, int B, int C, vqa_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Train-step tail over flat fp32 buffers.
 * Replaces nn.utils.clip_grad_norm_(model.parameters(), 0.25) + optimizer.step() of torch.optim.Adam
 * (train.py:81-86, :286-292): two HBM-bound launches instead of ~70 per-tensor walks, no host round trip.
 *
 * vqa_grad_norm_clip_coef: norm_and_coef[0] = ||g||_2 (fp64 accumulation), norm_and_coef[1] =
 *   min(1, max_norm / (norm + 1e-6))  (max_norm <= 0: coef = 1).  workspace: vqa_grad_norm_workspace_bytes().
 * vqa_adam_step: Adam (betas, eps, no weight decay / amsgrad) with bias correction for `step` (1-based) on the
 *   gradients scaled by norm_and_coef[1] (NULL = unscaled); p, m, v updated in place.
 * ------------------------------------------------------------------------------------------- */
size_t vqa_grad_norm_workspace_bytes(void);
int vqa_grad_norm_clip_coef(const float* g, size_t n, float max_norm, float* norm_and_coef, void* workspace,
                            size_t workspace_bytes, vqa_stream_t stream);
int vqa_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* norm_and_coef, float lr,
                  float beta1, float beta2, float eps, int step, vqa_stream_t stream);
/* Same update with the two per-step scalars read from DEVICE memory: step_scalars[0] = lr / (1 - beta1^t),
 * step_scalars[1] = 1 / sqrt(1 - beta2^t) -- so a captured hipGraph of the step can be replayed as lr and t move. */
int vqa_adam_step_dyn(float* p, const float* g, float* m, float* v, size_t n, const float* norm_and_coef,
                      const float* step_scalars, float beta1, float beta2, float eps, vqa_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VQA_MI355X_H */
