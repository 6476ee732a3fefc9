// Fused gradient-norm clip + Adam over flat fp32 buffers (the reference's train step, train.py:81-86:
// nn.utils.clip_grad_norm_(model.parameters(), 0.25) followed by torch.optim.Adam(lr).step()).
//
// The reference walks ~70 parameter tensors twice (norm, then Adam); with every parameter, gradient and Adam moment
// living in ONE flat fp32 buffer each (the gradient buffer is also the all-reduce payload) the whole update is two
// HBM-bound launches: a two-level sum of squares, and one pass that reads p, g, m, v and writes p, m, v
// (28 B per parameter, 334 MB per step for CoR2's 11.9 M parameters).  The clip coefficient never visits the host.
#include "common.hpp"

namespace vqa {

constexpr int kNormBlocks = 1024;

// partial[blockIdx] = sum over this block's grid-stride slice of g^2 (fp64 accumulation across the block)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial) {
  __shared__ double red[4];
  const size_t n4 = n / 4;
  float acc = 0.f;
  double dacc = 0.0;
  int cnt = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = ld4(g + 4 * i);
    acc = fmaf(x.x, x.x, fmaf(x.y, x.y, fmaf(x.z, x.z, fmaf(x.w, x.w, acc))));
    if (++cnt == 16) {  // spill the short fp32 run into fp64 so 12M-element sums keep ~1e-7 relative accuracy
      dacc += (double)acc;
      acc = 0.f;
      cnt = 0;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n - n4 * 4)) {
    const float x = g[n4 * 4 + threadIdx.x];
    acc = fmaf(x, x, acc);
  }
  dacc += (double)acc;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dacc += __shfl_xor(dacc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dacc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm_out[0] = sqrt(sum partial) ; norm_out[1] = min(1, max_norm / (norm + 1e-6))   (clip_grad_norm_ semantics)
__global__ __launch_bounds__(256) void norm_finish_kernel(const double* __restrict__ partial, int nparts, float max_norm,
                                                          float* __restrict__ norm_out) {
  __shared__ double red[4];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += partial[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
    norm_out[0] = norm;
    norm_out[1] = max_norm > 0.f ? fminf(1.f, max_norm / (norm + 1e-6f)) : 1.f;
  }
}

// torch.optim.Adam (no amsgrad, no weight decay) on clipped gradients g*coef:
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// `dyn` (optional, device): {lr / (1 - b1^t), 1 / sqrt(1 - b2^t)} -- the per-step scalars live in device memory so a
// captured hipGraph of the step can be replayed while the learning rate and the bias corrections move on.
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, size_t n, const float* __restrict__ coef_ptr,
                                                   const float* __restrict__ dyn, float b1, float b2, float eps,
                                                   float step_size, float inv_sqrt_bc2) {
  const float coef = coef_ptr != nullptr ? coef_ptr[1] : 1.f;
  if (dyn != nullptr) {
    step_size = dyn[0];
    inv_sqrt_bc2 = dyn[1];
  }
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    const float4 gg = ld4(g + i);
    float4 pp = ld4(p + i), mm = ld4(m + i), vv = ld4(v + i);
    const float gs[4] = {gg.x * coef, gg.y * coef, gg.z * coef, gg.w * coef};
    float* pm[3] = {&pp.x, &mm.x, &vv.x};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float mk = b1 * pm[1][k] + (1.f - b1) * gs[k];
      const float vk = b2 * pm[2][k] + (1.f - b2) * gs[k] * gs[k];
      pm[1][k] = mk;
      pm[2][k] = vk;
      pm[0][k] -= step_size * mk / (sqrtf(vk) * inv_sqrt_bc2 + eps);
    }
    st4(p + i, pp);
    st4(m + i, mm);
    st4(v + i, vv);
  } else {
    for (size_t j = i; j < n; ++j) {
      const float gk = g[j] * coef;
      const float mk = b1 * m[j] + (1.f - b1) * gk;
      const float vk = b2 * v[j] + (1.f - b2) * gk * gk;
      m[j] = mk;
      v[j] = vk;
      p[j] -= step_size * mk / (sqrtf(vk) * inv_sqrt_bc2 + eps);
    }
  }
}

}  // namespace vqa

using namespace vqa;

extern "C" size_t vqa_grad_norm_workspace_bytes(void) { return kNormBlocks * sizeof(double); }

extern "C" int vqa_grad_norm_clip_coef(const float* g, size_t n, float max_norm, float* norm_and_coef, void* workspace,
                                       size_t workspace_bytes, vqa_stream_t stream) {
  VQA_REQUIRE(g && norm_and_coef && workspace && n > 0, VQA_E_BADARG, "grad_norm_clip_coef: null pointer or n == 0");
  VQA_REQUIRE(workspace_bytes >= vqa_grad_norm_workspace_bytes(), VQA_E_BADARG, "grad_norm_clip_coef: workspace too small");
  VQA_REQUIRE(aligned(g, 16) && aligned(workspace, 8), VQA_E_UNSUPPORTED, "grad_norm_clip_coef: g must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t need = (n / 4 + 255) / 256;
  const int blocks = (int)(need < (size_t)kNormBlocks ? (need ? need : 1) : (size_t)kNormBlocks);
  double* partial = static_cast<double*>(workspace);
  VQA_LAUNCH(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, s, g, n, partial);
  VQA_LAUNCH(norm_finish_kernel, dim3(1), dim3(256), 0, s, partial, blocks, max_norm, norm_and_coef);
  return check_launch("grad_norm_clip_coef");
}

extern "C" int vqa_adam_step(float* p, const float* g, float* m, float* v, size_t n, const float* norm_and_coef, float lr,
                             float beta1, float beta2, float eps, int step, vqa_stream_t stream) {
  VQA_REQUIRE(p && g && m && v && n > 0 && step >= 1, VQA_E_BADARG, "adam_step: null pointer, n == 0 or step < 1");
  VQA_REQUIRE(aligned(p, 16) && aligned(g, 16) && aligned(m, 16) && aligned(v, 16), VQA_E_UNSUPPORTED,
              "adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const float step_size = (float)(lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const size_t blocks = (n / 4 + 256) / 256;
  VQA_LAUNCH(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n,
                     norm_and_coef, static_cast<const float*>(nullptr), beta1, beta2, eps, step_size, inv_sqrt_bc2);
  return check_launch("adam_step");
}

extern "C" int vqa_adam_step_dyn(float* p, const float* g, float* m, float* v, size_t n, const float* norm_and_coef,
                                 const float* step_scalars, float beta1, float beta2, float eps, vqa_stream_t stream) {
  VQA_REQUIRE(p && g && m && v && step_scalars && n > 0, VQA_E_BADARG, "adam_step_dyn: null pointer or n == 0");
  VQA_REQUIRE(aligned(p, 16) && aligned(g, 16) && aligned(m, 16) && aligned(v, 16), VQA_E_UNSUPPORTED,
              "adam_step_dyn: buffers must be 16-byte aligned");
  const size_t blocks = (n / 4 + 256) / 256;
  VQA_LAUNCH(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), p, g, m, v, n,
                     norm_and_coef, step_scalars, beta1, beta2, eps, 0.f, 1.f);
  return check_launch("adam_step_dyn");
}
