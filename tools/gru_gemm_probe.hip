// Probe (round 4): the BayesianGRU's recurrent step product  a[g] = hm[g] W_g^T,  hm [512,2400], W_g [2400,2400],
// three gates (putils/__init__.py:704-731) -- on the register-tile fp32 MFMA engine (csrc/gemm_f32_rt.hpp) at the tile shapes
// that give 512 rows enough workgroups, against the 118 TF/s the library's batched GEMM reaches on the same product
// (profiles/).  Timing only (the engine's correctness is tests/test_gpu_kernels.py's); N = 7200 stands for the
// three gates (same FLOPs, one A operand).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gru_gemm_probe.hip -o tools/gru_gemm_probe && tools/gru_gemm_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "../vqa_playground_pytorch_amd/csrc/gemm_f32_rt.hpp"

namespace vqa {
char* error_buffer() {
  static thread_local char buf[512];
  return buf;
}
}  // namespace vqa
using namespace vqa;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Epi {
  float* y;
  int ldy;
  __device__ __forceinline__ void operator()(int row, int col, float v) const { y[(size_t)row * ldy + col] = v; }
};
__global__ void fill_kernel(float* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = ((float)(mask_word32((uint32_t)i, seed) >> 8) * (1.f / 8388608.f) - 1.f) * scale;
}
template <int RB, int CB, int WM, int WN, int WK>
static void run(const char* what, const float* A, const float* B, float* y, int M, int N, int K) {
  using S = rt::NtShape<RB, CB, WM, WN, WK>;
  const int tiles_m = (M + S::BM - 1) / S::BM, tiles_n = (N + S::BN - 1) / S::BN;
  rt::NtArgs p{A, B, K, K, M, N, K, tiles_n, nullptr};
  auto kern = rt::gemm_nt_kernel<RB, CB, WM, WN, WK, false, Epi, 0>;
  if (S::kLdsBytes > 65536) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLdsBytes));
  const DropCfg dc = make_drop(0.f, 0);
  auto go = [&] { hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(rt::kThreads), S::kLdsBytes, 0, p, dc, Epi{y, N}); };
  for (int i = 0; i < 10; ++i) go();
  CK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  CK(hipEventRecord(a));
  for (int i = 0; i < 20; ++i) go();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms / 20 * 1e3, flop = 2.0 * M * N * K;
  printf("  %-44s tile %3d x %3d, %4d workgroups: %7.1f us  %6.1f TF/s (%.2f of 157.3)\n", what, S::BM, S::BN, tiles_m * tiles_n, us,
         flop / us / 1e6, flop / us / 1e6 / 157.3);
}

int main() {
  const int M = 512, K = 2400, N = 7200;
  float *A, *B, *y;
  CK(hipMalloc(&A, (size_t)M * K * 4));
  CK(hipMalloc(&B, (size_t)N * K * 4));
  CK(hipMalloc(&y, (size_t)M * N * 4));
  fill_kernel<<<(unsigned)(((size_t)M * K + 255) / 256), 256>>>(A, (size_t)M * K, 11, 1.f);
  fill_kernel<<<(unsigned)(((size_t)N * K + 255) / 256), 256>>>(B, (size_t)N * K, 22, 1.f / 48.f);
  printf("GRU step product [512,2400] x [7200,2400]^T (three gates), 17.7 GFLOP; the library's batched GEMM: ~150 us, 118 TF/s\n");
  run<9, 5, 1, 2, 2>("K5's shape: 9x5 blocks, K split 2", A, B, y, M, N, K);
  run<8, 5, 1, 2, 2>("8x5 blocks, K split 2", A, B, y, M, N, K);
  run<4, 5, 2, 2, 1>("4x5 blocks, 2x2 waves", A, B, y, M, N, K);
  run<4, 5, 1, 2, 2>("4x5 blocks, K split 2", A, B, y, M, N, K);
  run<8, 3, 1, 2, 2>("8x3 blocks, K split 2", A, B, y, M, N, K);
  run<4, 7, 1, 2, 2>("4x7 blocks, K split 2", A, B, y, M, N, K);
  run<4, 5, 2, 1, 2>("4x5 blocks, 2x1 waves, K split 2", A, B, y, M, N, K);
  run<2, 5, 2, 1, 2>("2x5 blocks, 2x1 waves, K split 2", A, B, y, M, N, K);
  return 0;
}
