// KLD-sum loss on soft targets, forward and gradient in one pass over the logits.
//
// Replaces MyLoss of the reference (train.py:536-544): KLDivLoss(size_average=False)(F.log_softmax(logits), target) =
//   loss = sum_{b,c} a[b,c] * (log a[b,c] - log_softmax(z)[b,c])          (0 * log 0 = 0)
//   dloss/dz[b,c] = softmax(z)[b,c] * sum_c a[b,c] - a[b,c]
// which torch runs as log_softmax + kl_div + a [B,C] -> scalar sum whose multi-block reduction zeroes its semaphores
// with a memset node -- the node that replays wrongly inside a hipGraph on ROCm 7.2 (see api.hip), so a graph-replayed
// step returned a garbage loss.  Here: one workgroup per sample row (C values in registers, wave64 DPP reductions),
// then one workgroup adds the B row losses in a fixed order (bitwise reproducible, no atomics, no memset).
// HBM-bound: reads z and a once, writes dz once = 3 * B * C * 4 bytes.
#include "common.hpp"

namespace vqa {

constexpr int kLossThreads = 256;
constexpr int kLossPerThread = 16;  // C <= 4096

__device__ __forceinline__ float block_sum(float x, float* red_s) {
  x = wave_sum(x);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red_s[wave] = x;
  __syncthreads();
  return red_s[0] + red_s[1] + red_s[2] + red_s[3];
}
__device__ __forceinline__ float block_max(float x, float* red_s) {
  x = wave_max(x);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red_s[wave] = x;
  __syncthreads();
  return fmaxf(fmaxf(red_s[0], red_s[1]), fmaxf(red_s[2], red_s[3]));
}

__global__ __launch_bounds__(kLossThreads) void kld_rows_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ target,
                                                                float* __restrict__ row_loss, float* __restrict__ d_logits,
                                                                int C) {
  __shared__ float red_s[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* z = logits + (size_t)b * C;
  const float* a = target + (size_t)b * C;
  float zv[kLossPerThread], av[kLossPerThread];
  float m = -INFINITY;
#pragma unroll
  for (int i = 0; i < kLossPerThread; ++i) {
    const int c = tid + i * kLossThreads;
    const int cc = min(c, C - 1);  // unconditional loads from a clamped column
    const float zt = z[cc], at = a[cc];
    zv[i] = c < C ? zt : -INFINITY;
    av[i] = c < C ? at : 0.f;
    m = fmaxf(m, zv[i]);
  }
  m = block_max(m, red_s);
  float se = 0.f, sa = 0.f, saz = 0.f, sal = 0.f;
#pragma unroll
  for (int i = 0; i < kLossPerThread; ++i) {
    const float e = expf(zv[i] - m);  // exp(-inf) = 0 for the padded columns
    se += e;
    sa += av[i];
    if (av[i] > 0.f) {
      saz = fmaf(av[i], zv[i] - m, saz);
      sal = fmaf(av[i], logf(av[i]), sal);
    }
    zv[i] = e;
  }
  se = block_sum(se, red_s);
  sa = block_sum(sa, red_s);
  saz = block_sum(saz, red_s);
  sal = block_sum(sal, red_s);
  // sum_c a (log a - (z - m - log se)) = sal - saz + sa * log se
  if (tid == 0) row_loss[b] = sal - saz + sa * logf(se);
  if (d_logits != nullptr) {
    const float k = sa / se;
#pragma unroll
    for (int i = 0; i < kLossPerThread; ++i) {
      const int c = tid + i * kLossThreads;
      if (c < C) d_logits[(size_t)b * C + c] = fmaf(zv[i], k, -av[i]);
    }
  }
}

__global__ __launch_bounds__(256) void kld_total_kernel(const float* __restrict__ row_loss, float* __restrict__ loss, int B) {
  __shared__ float red_s[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) s += row_loss[b];
  s = block_sum(s, red_s);
  if (threadIdx.x == 0) loss[0] = s;
}

}  // namespace vqa

using namespace vqa;

extern "C" size_t vqa_kld_sum_loss_workspace_bytes(int B) { return B > 0 ? (size_t)B * sizeof(float) : 0; }

extern "C" int vqa_kld_sum_loss(const float* logits, const float* target, float* loss, float* d_logits, void* workspace,
                                size_t workspace_bytes, int B, int C, vqa_stream_t stream) {
  VQA_REQUIRE(logits && target && loss && workspace, VQA_E_BADARG, "kld_sum_loss: null pointer");
  VQA_REQUIRE(B > 0 && C > 0, VQA_E_BADARG, "kld_sum_loss: bad sizes B=%d C=%d", B, C);
  VQA_REQUIRE(C <= kLossThreads * kLossPerThread, VQA_E_UNSUPPORTED, "kld_sum_loss: C=%d exceeds %d", C,
              kLossThreads * kLossPerThread);
  VQA_REQUIRE(workspace_bytes >= vqa_kld_sum_loss_workspace_bytes(B), VQA_E_BADARG,
              "kld_sum_loss: workspace of %zu B is too small", workspace_bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* row_loss = static_cast<float*>(workspace);
  VQA_LAUNCH(kld_rows_kernel, dim3(B), dim3(kLossThreads), 0, s, logits, target, row_loss, d_logits, C);
  VQA_LAUNCH(kld_total_kernel, dim3(1), dim3(256), 0, s, row_loss, loss, B);
  return check_launch("kld_sum_loss");
}
