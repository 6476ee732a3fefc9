// Standalone probe of the register-tile fp32 MFMA engine (csrc/gemm_f32_rt.hpp): correctness against an fp64-accumulating
// reference kernel and timing at the region-projection shapes (M = 18432, K = 2048, N = 310).  No torch, no library GEMM.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/rt_probe.hip -o tools/rt_probe && tools/rt_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../vqa_playground_pytorch_amd/csrc/gemm_f32_rt.hpp"

namespace vqa {
char* error_buffer() {
  static thread_local char buf[512];
  return buf;
}
}  // namespace vqa

using namespace vqa;

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);   \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

struct EpiBiasRelu {
  float* y;
  const float* bias;
  int ldy, act;
  float scale;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    v = v * scale + (bias != nullptr ? bias[col] : 0.f);
    if (act == 1) v = fmaxf(v, 0.f);
    y[(size_t)row * ldy + col] = v;
  }
};

__global__ void fill_kernel(float* p, size_t n, uint32_t seed, float scale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t h = mask_word32((uint32_t)i, seed);
    p[i] = ((float)(h >> 8) * (1.f / 8388608.f) - 1.f) * scale;   // uniform [-1, 1) * scale
  }
}

// y[m][n] = act(sum_k drop(m,k) A[m][k] B[n][k] + bias[n]), fp64 accumulation
__global__ void ref_nt_kernel(const float* A, int lda, const float* B, int ldb, const float* bias, float* y, int M, int N,
                              int K, int act, DropCfg dc) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  double s = 0.0;
  for (int k = 0; k < K; ++k) {
    float a = A[(size_t)m * lda + k];
    if (dc.p8 > 0) a *= drop_one((uint32_t)m * (uint32_t)K + (uint32_t)k, dc);
    s += (double)a * (double)B[(size_t)n * ldb + k];
  }
  float v = (float)s + (bias ? bias[n] : 0.f);
  if (act == 1) v = fmaxf(v, 0.f);
  y[(size_t)m * N + n] = v;
}

// dw[n1][n2] = sum_m gate(P)[m][n1] * drop(Q)[m][n2], fp64 accumulation; db[n1] = sum_m gate(P)[m][n1]
__global__ void ref_tn_kernel(const float* P, const float* Y, const float* Q, float* dw, float* db, int M, int N1, int N2,
                              DropCfg dc) {
  const int n2 = blockIdx.x * blockDim.x + threadIdx.x, n1 = blockIdx.y;
  if (n2 >= N2) return;
  double s = 0.0, sb = 0.0;
  for (int m = 0; m < M; ++m) {
    float a = P[(size_t)m * N1 + n1];
    if (Y != nullptr && !(Y[(size_t)m * N1 + n1] > 0.f)) a = 0.f;
    float q = Q[(size_t)m * N2 + n2];
    if (dc.p8 > 0) q *= drop_one((uint32_t)m * (uint32_t)N2 + (uint32_t)n2, dc);
    s += (double)a * (double)q;
    sb += (double)a;
  }
  dw[(size_t)n1 * N2 + n2] = (float)s;
  if (n2 == 0) db[n1] = (float)sb;
}

__global__ void reduce_kernel(const float* slab, float* out, size_t n, int S) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float a = 0.f;
  for (int s = 0; s < S; ++s) a += slab[(size_t)s * n + i];
  out[i] = a;
}

static double compare(const char* what, const float* got, const float* ref, size_t n, int ld) {
  std::vector<float> hg(n), hr(n);
  CK(hipMemcpy(hg.data(), got, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hr.data(), ref, n * 4, hipMemcpyDeviceToHost));
  double maxref = 0, maxerr = 0;
  size_t worst = 0, bad = 0;
  for (size_t i = 0; i < n; ++i) {
    maxref = fmax(maxref, fabs((double)hr[i]));
    const double e = fabs((double)hg[i] - (double)hr[i]);
    if (!(e <= maxerr)) {
      maxerr = e;
      worst = i;
    }
  }
  for (size_t i = 0; i < n; ++i)
    if (!(fabs((double)hg[i] - (double)hr[i]) <= 2e-4 * maxref)) {
      if (bad < 8) printf("    bad [%zu,%zu]: got %g ref %g\n", i / ld, i % ld, hg[i], hr[i]);
      ++bad;
    }
  printf("  %-34s max|ref| %.4g  max err %.3e (rel %.2e) at [%zu,%zu]  bad %zu / %zu  %s\n", what, maxref, maxerr,
         maxerr / fmax(maxref, 1e-30), worst / ld, worst % ld, bad, n, bad == 0 ? "OK" : "FAIL");
  return maxerr / fmax(maxref, 1e-30);
}

template <class F>
static float time_ms(F&& f, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 10; ++i) f();   // warm-up: the first launches after an idle phase run at a lower clock
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

static void fill(float* p, size_t n, uint32_t seed, float scale) {
  fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, scale);
}

template <int RB, int CB, int WM, int WN, int WK, bool DROP, int TUNE = 0>
static void launch_nt(const float* A, int lda, const float* B, int ldb, const float* bias, float* y, int M, int N, int K,
                      int act, DropCfg dc, unsigned long long* stamps = nullptr) {
  using S = rt::NtShape<RB, CB, WM, WN, WK>;
  const int tiles_m = (M + S::BM - 1) / S::BM, tiles_n = (N + S::BN - 1) / S::BN;
  rt::NtArgs p{A, B, lda, ldb, M, N, K, tiles_n, stamps};
  auto kern = rt::gemm_nt_kernel<RB, CB, WM, WN, WK, DROP, EpiBiasRelu, TUNE>;
  static bool once = false;
  if (!once && S::kLdsBytes > 65536) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLdsBytes));
    once = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(rt::kThreads), S::kLdsBytes, 0, p, dc, EpiBiasRelu{y, bias, N, act, (DROP && dc.p8 > 0) ? dc.scale : 1.f});
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);

  // ---------------------------------------------------------------- NT: forward projection ----------------------
  {
    const int M = 18432, K = 2048, N = 310;
    float *A, *B, *bias, *y, *yref;
    CK(hipMalloc(&A, (size_t)M * K * 4));
    CK(hipMalloc(&B, (size_t)N * K * 4));
    CK(hipMalloc(&bias, N * 4));
    CK(hipMalloc(&y, (size_t)M * N * 4));
    CK(hipMalloc(&yref, (size_t)M * N * 4));
    fill(A, (size_t)M * K, 11, 1.f);
    fill(B, (size_t)N * K, 22, 1.f / 32.f);
    fill(bias, N, 33, 0.5f);
    const double flop = 2.0 * M * K * N;
    for (int drop = 0; drop < 2; ++drop) {
      const DropCfg dc = make_drop(drop ? 0.5f : 0.f, 0x1234567ull);
      ref_nt_kernel<<<dim3((N + 63) / 64, M), 64>>>(A, K, B, K, bias, yref, M, N, K, 1, dc);
      CK(hipMemset(y, 0xff, (size_t)M * N * 4));
      if (drop)
        launch_nt<9, 5, 1, 2, 2, true>(A, K, B, K, bias, y, M, N, K, 1, dc);
      else
        launch_nt<9, 5, 1, 2, 2, false>(A, K, B, K, bias, y, M, N, K, 1, dc);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare(drop ? "fwd 9x5 k-split, dropout 0.5" : "fwd 9x5 k-split", y, yref, (size_t)M * N, N);
      float ms = drop ? time_ms([&] { launch_nt<9, 5, 1, 2, 2, true>(A, K, B, K, bias, y, M, N, K, 1, dc); }, iters)
                      : time_ms([&] { launch_nt<9, 5, 1, 2, 2, false>(A, K, B, K, bias, y, M, N, K, 1, dc); }, iters);
      printf("  -> %.1f us  %.1f TF/s (%.1f%% of 157.3)\n", ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
    }
    // ---- experiments on the forward shape (timing only unless stated) ----
    {
      const DropCfg dc0 = make_drop(0.f, 0);
      unsigned long long* stamps;
      CK(hipMalloc(&stamps, 1024 * 2 * 8));
      auto clock_report = [&](const char* what) {
        std::vector<unsigned long long> h(2048);
        CK(hipMemcpy(h.data(), stamps, 2048 * 8, hipMemcpyDeviceToHost));
        std::vector<double> mhz, cyc;
        for (int i = 0; i < 1024; ++i)
          if (h[2 * i + 1] > 0) {
            mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
            cyc.push_back((double)h[2 * i]);
          }
        std::sort(mhz.begin(), mhz.end());
        std::sort(cyc.begin(), cyc.end());
        if (!mhz.empty())
          printf("     %-28s loop clock median %.0f MHz (min %.0f max %.0f), loop cycles median %.0f max %.0f (ideal %d)\n", what,
                 mhz[mhz.size() / 2], mhz.front(), mhz.back(), cyc[cyc.size() / 2], cyc.back(), 64 * 180 * 32);
      };
#define RT_TIME(label, ...)                                                                                   \
  {                                                                                                           \
    float ms = time_ms([&] { __VA_ARGS__; }, iters);                                                          \
    printf("  exp %-40s %.1f us  %.1f TF/s\n", label, ms * 1e3, flop / ms / 1e9);                              \
  }
      {
        const DropCfg dch = make_drop(0.5f, 777);
        RT_TIME("dropout, VALU unpinned (TUNE 0)", (launch_nt<9, 5, 1, 2, 2, true, 0>(A, K, B, K, bias, y, M, N, K, 1, dch)));
        RT_TIME("dropout, VALU 1 per MFMA (TUNE 32)", (launch_nt<9, 5, 1, 2, 2, true, 32>(A, K, B, K, bias, y, M, N, K, 1, dch)));
        RT_TIME("no dropout (TUNE 0)", (launch_nt<9, 5, 1, 2, 2, false, 0>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
        RT_TIME("dropout, VALU unpinned (TUNE 0) again", (launch_nt<9, 5, 1, 2, 2, true, 0>(A, K, B, K, bias, y, M, N, K, 1, dch)));
        // round 4: hash words shared across the four lane groups by ds_bpermute (TUNE 0) vs hashed per lane (64); the chunk's
        // masks in one burst in front of its MFMAs (0, 64: the default) vs staggered over its row blocks (128, 192: round 2)
        RT_TIME("dropout, per-lane hashes, one burst (TUNE 64)", (launch_nt<9, 5, 1, 2, 2, true, 64>(A, K, B, K, bias, y, M, N, K, 1, dch)));
        RT_TIME("dropout, shared hashes, staggered (TUNE 128)", (launch_nt<9, 5, 1, 2, 2, true, 128>(A, K, B, K, bias, y, M, N, K, 1, dch)));
        RT_TIME("dropout, per-lane hashes, staggered (192)", (launch_nt<9, 5, 1, 2, 2, true, 192>(A, K, B, K, bias, y, M, N, K, 1, dch)));
        RT_TIME("dropout, shared hashes, one burst (TUNE 0) again", (launch_nt<9, 5, 1, 2, 2, true, 0>(A, K, B, K, bias, y, M, N, K, 1, dch)));
        RT_TIME("no dropout (TUNE 0) again", (launch_nt<9, 5, 1, 2, 2, false, 0>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      }
      RT_TIME("stamped (TUNE 16)", (launch_nt<9, 5, 1, 2, 2, false, 16>(A, K, B, K, bias, y, M, N, K, 1, dc0, stamps)));
      clock_report("real kernel");
      RT_TIME("no loads in loop (TUNE 17)", (launch_nt<9, 5, 1, 2, 2, false, 17>(A, K, B, K, bias, y, M, N, K, 1, dc0, stamps)));
      clock_report("no loads");
      RT_TIME("loads every 2 MFMAs (TUNE 4)", (launch_nt<9, 5, 1, 2, 2, false, 4>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      RT_TIME("loads every 3 MFMAs (TUNE 6)", (launch_nt<9, 5, 1, 2, 2, false, 6>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      RT_TIME("loads every 4 MFMAs (TUNE 8)", (launch_nt<9, 5, 1, 2, 2, false, 8>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      RT_TIME("loads every 7 MFMAs (TUNE 14)", (launch_nt<9, 5, 1, 2, 2, false, 14>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      // padded row stride (2048 + 16 floats): would channel / bank aliasing of the 8 KB row stride matter?
      float* Apad;
      CK(hipMalloc(&Apad, (size_t)M * (K + 16) * 4));
      fill(Apad, (size_t)M * (K + 16), 12, 1.f);
      RT_TIME("lda = 2064 (default spacing)", (launch_nt<9, 5, 1, 2, 2, false, 0>(Apad, K + 16, B, K, bias, y, M, N, K, 1, dc0)));
      RT_TIME("lda = 2064, loads every 3", (launch_nt<9, 5, 1, 2, 2, false, 6>(Apad, K + 16, B, K, bias, y, M, N, K, 1, dc0)));
      // other decompositions of the same problem
      RT_TIME("8x5 k-split tile 128x160", (launch_nt<8, 5, 1, 2, 2, false, 0>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      RT_TIME("6x5 2x2 tile 192x160 (full K)", (launch_nt<6, 5, 2, 2, 1, false, 0>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      RT_TIME("4x5 2x2 tile 128x160 (full K)", (launch_nt<4, 5, 2, 2, 1, false, 0>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
      CK(hipFree(Apad));
      CK(hipFree(stamps));
    }
    // ragged shape: M, N, K all off the tile grid
    {
      const int M2 = 1000, K2 = 352, N2 = 170;
      const DropCfg dc = make_drop(0.5f, 99);
      ref_nt_kernel<<<dim3((N2 + 63) / 64, M2), 64>>>(A, K2, B, K2, bias, yref, M2, N2, K2, 0, dc);
      CK(hipMemset(y, 0xff, (size_t)M2 * N2 * 4));
      launch_nt<9, 5, 1, 2, 2, true>(A, K2, B, K2, bias, y, M2, N2, K2, 0, dc);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare("fwd ragged 1000x352x170 dropout", y, yref, (size_t)M2 * N2, N2);
    }
    // ------------------------------------------------------------ NT: data gradient  dx = gz [M,310] W^T[2048,312]
    {
      const int Kd = 310, Nd = 2048, ldb = 312;
      float *gz = y, *wt, *dx, *dxref;   // reuse y as gz [M,310]
      fill(gz, (size_t)M * Kd, 44, 1.f);
      CK(hipMalloc(&wt, (size_t)Nd * ldb * 4));
      CK(hipMalloc(&dx, (size_t)M * Nd * 4));
      CK(hipMalloc(&dxref, (size_t)M * Nd * 4));
      fill(wt, (size_t)Nd * ldb, 55, 1.f / 16.f);
      const DropCfg dc0 = make_drop(0.f, 0);
      ref_nt_kernel<<<dim3((Nd + 63) / 64, M), 64>>>(gz, Kd, wt, ldb, nullptr, dxref, M, Nd, Kd, 0, dc0);
      CK(hipMemset(dx, 0xff, (size_t)M * Nd * 4));
      launch_nt<6, 8, 2, 2, 1, false>(gz, Kd, wt, ldb, nullptr, dx, M, Nd, Kd, 0, dc0);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare("dgrad 6x8 (K=310, ld 310/312)", dx, dxref, (size_t)M * Nd, Nd);
      float ms = time_ms([&] { launch_nt<6, 8, 2, 2, 1, false>(gz, Kd, wt, ldb, nullptr, dx, M, Nd, Kd, 0, dc0); }, iters);
      printf("  -> %.1f us  %.1f TF/s (%.1f%% of 157.3)\n", ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
      {
        unsigned long long* stamps;
        CK(hipMalloc(&stamps, 1024 * 2 * 8));
        CK(hipMemset(stamps, 0, 1024 * 2 * 8));
        ms = time_ms([&] { launch_nt<6, 8, 2, 2, 1, false, 16>(gz, Kd, wt, ldb, nullptr, dx, M, Nd, Kd, 0, dc0, stamps); }, iters);
        std::vector<unsigned long long> h(2048);
        CK(hipMemcpy(h.data(), stamps, 2048 * 8, hipMemcpyDeviceToHost));
        std::vector<double> mhz, cyc;
        for (int i = 0; i < 1024; ++i)
          if (h[2 * i + 1] > 0) {
            mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
            cyc.push_back((double)h[2 * i]);
          }
        std::sort(mhz.begin(), mhz.end());
        std::sort(cyc.begin(), cyc.end());
        printf("  exp dgrad 6x8 stamped: %.1f us; pair loop (18 chunks) clock median %.0f MHz, cycles median %.0f max %.0f (ideal %d)\n",
               ms * 1e3, mhz.empty() ? 0.0 : mhz[mhz.size() / 2], cyc.empty() ? 0.0 : cyc[cyc.size() / 2],
               cyc.empty() ? 0.0 : cyc.back(), 18 * 192 * 32);
        CK(hipFree(stamps));
        // aligned operand strides (gz rows padded to 320 floats, K = 320 with zero columns): what would alignment buy?
        float* gz2;
        CK(hipMalloc(&gz2, (size_t)M * 320 * 4));
        fill(gz2, (size_t)M * 320, 45, 1.f);
        float* wt2;
        CK(hipMalloc(&wt2, (size_t)Nd * 320 * 4));
        fill(wt2, (size_t)Nd * 320, 56, 1.f / 16.f);
        ms = time_ms([&] { launch_nt<6, 8, 2, 2, 1, false>(gz2, 320, wt2, 320, nullptr, dx, M, Nd, 320, 0, dc0); }, iters);
        printf("  exp dgrad 6x8, K = 320, lda = ldb = 320 (aligned, no tail): %.1f us  %.1f TF/s\n", ms * 1e3, 2.0 * M * Nd * 320 / ms / 1e9);
        ms = time_ms([&] { launch_nt<6, 8, 2, 2, 1, false>(gz2, 320, wt2, 320, nullptr, dx, M, Nd, 304, 0, dc0); }, iters);
        printf("  exp dgrad 6x8, K = 304 (19 chunks, odd, no tail), aligned: %.1f us  %.1f TF/s\n", ms * 1e3, 2.0 * M * Nd * 304 / ms / 1e9);
        ms = time_ms([&] { launch_nt<6, 8, 2, 2, 1, false>(gz2, 320, wt2, 320, nullptr, dx, M, Nd, 288, 0, dc0); }, iters);
        printf("  exp dgrad 6x8, K = 288 (18 chunks, even, no tail), aligned: %.1f us  %.1f TF/s\n", ms * 1e3, 2.0 * M * Nd * 288 / ms / 1e9);
        ms = time_ms([&] { launch_nt<6, 8, 2, 2, 1, false>(gz, Kd, wt2, 320, nullptr, dx, M, Nd, 288, 0, dc0); }, iters);
        printf("  exp dgrad 6x8, K = 288, lda = 310 (A rows 8-byte aligned only): %.1f us  %.1f TF/s\n", ms * 1e3, 2.0 * M * Nd * 288 / ms / 1e9);
        ms = time_ms([&] { launch_nt<6, 8, 2, 2, 1, false>(gz2, 320, wt2, 320, nullptr, dx, 6144, Nd, 320, 0, dc0); }, iters);
        printf("  exp dgrad 6x8, M = 6144 (ONE tile per CU), K = 320 aligned: %.1f us  %.1f TF/s\n", ms * 1e3, 2.0 * 6144 * Nd * 320 / ms / 1e9);
        CK(hipFree(gz2));
        CK(hipFree(wt2));
      }
      ref_nt_kernel<<<dim3((Nd + 63) / 64, M), 64>>>(gz, Kd, wt, ldb, nullptr, dxref, M, Nd, Kd, 0, dc0);
      CK(hipMemset(dx, 0xff, (size_t)M * Nd * 4));
      launch_nt<9, 5, 1, 2, 2, false>(gz, Kd, wt, ldb, nullptr, dx, M, Nd, Kd, 0, dc0);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare("dgrad 9x5 k-split", dx, dxref, (size_t)M * Nd, Nd);
      ms = time_ms([&] { launch_nt<9, 5, 1, 2, 2, false>(gz, Kd, wt, ldb, nullptr, dx, M, Nd, Kd, 0, dc0); }, iters);
      printf("  -> %.1f us  %.1f TF/s (%.1f%% of 157.3)\n", ms * 1e3, flop / ms / 1e9, flop / ms / 1e9 / 1.573);
      CK(hipFree(wt));
      CK(hipFree(dx));
      CK(hipFree(dxref));
    }
    // ------------------------------------------------------------ TN: weight gradient dW = gz^T drop(x)
    {
      const int N1 = 310, N2 = 2048, S = 16;
      float *P = y, *Y, *slab, *dbslab, *dw, *db, *dwref, *dbref;
      CK(hipMalloc(&Y, (size_t)M * N1 * 4));
      CK(hipMalloc(&slab, (size_t)S * N1 * N2 * 4));
      CK(hipMalloc(&dbslab, (size_t)S * N1 * 4));
      CK(hipMalloc(&dw, (size_t)N1 * N2 * 4));
      CK(hipMalloc(&db, N1 * 4));
      CK(hipMalloc(&dwref, (size_t)N1 * N2 * 4));
      CK(hipMalloc(&dbref, N1 * 4));
      fill(P, (size_t)M * N1, 66, 1.f);
      fill(Y, (size_t)M * N1, 77, 1.f);
      for (int variant = 0; variant < 2; ++variant) {
        const bool gated = variant == 1;
        const DropCfg dc = make_drop(gated ? 0.5f : 0.f, 4242);
        ref_tn_kernel<<<dim3((N2 + 63) / 64, N1), 64>>>(P, gated ? Y : nullptr, A, dwref, dbref, M, N1, N2, dc);
        rt::TnArgs p{P, gated ? Y : nullptr, A, slab, dbslab, N1, N2, M, N1, N2, (N1 + 319) / 320, (N2 + 127) / 128,
                     ((M + S - 1) / S + 15) / 16 * 16};
        const int grid = p.tiles1 * p.tiles2 * S;
        auto run = [&] {
          if (gated)
            hipLaunchKernelGGL((rt::gemm_tn_kernel<5, 2, true, true>), dim3(grid), dim3(rt::kThreads), 0, 0, p, dc);
          else
            hipLaunchKernelGGL((rt::gemm_tn_kernel<5, 2, false, false>), dim3(grid), dim3(rt::kThreads), 0, 0, p, dc);
        };
        CK(hipMemset(slab, 0xff, (size_t)S * N1 * N2 * 4));
        run();
        reduce_kernel<<<(N1 * N2 + 255) / 256, 256>>>(slab, dw, (size_t)N1 * N2, S);
        reduce_kernel<<<(N1 + 255) / 256, 256>>>(dbslab, db, N1, S);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        compare(gated ? "dW 5x8, relu gate + dropout" : "dW 5x8", dw, dwref, (size_t)N1 * N2, N2);
        compare(gated ? "db (gated)" : "db", db, dbref, N1, N1);
        if (gated) {   // round 4: hash words shared by groups of eight lanes (SHARE), the chunk's VALU side in one burst (TUNE 1)
          auto t = [&](const char* what, auto kern) {
            float ms_ = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(rt::kThreads), 0, 0, p, dc); }, iters);
            printf("  exp dW %-44s %.1f us\n", what, ms_ * 1e3);
          };
          t("per-lane hashes, staggered", rt::gemm_tn_kernel<5, 2, true, true, false, 0>);
          t("shared hashes, staggered", rt::gemm_tn_kernel<5, 2, true, true, true, 0>);
          t("per-lane hashes, one burst", rt::gemm_tn_kernel<5, 2, true, true, false, 1>);
          t("shared hashes, one burst", rt::gemm_tn_kernel<5, 2, true, true, true, 1>);
          t("shared hashes, no gate, staggered", rt::gemm_tn_kernel<5, 2, false, true, true, 0>);
          t("shared hashes, no gate, one burst", rt::gemm_tn_kernel<5, 2, false, true, true, 1>);
          t("no dropout, no gate", rt::gemm_tn_kernel<5, 2, false, false, false, 0>);
        }
        float ms = time_ms(run, iters);
        printf("  -> %.1f us  %.1f TF/s (%.1f%% of 157.3)   [+ slab reduce %.1f us]\n", ms * 1e3, flop / ms / 1e9,
               flop / ms / 1e9 / 1.573,
               1e3 * time_ms([&] { reduce_kernel<<<(N1 * N2 + 255) / 256, 256>>>(slab, dw, (size_t)N1 * N2, S); }, iters));
      }
      // ragged: M not a multiple of the split, N1 / N2 off the tile grid
      {
        const int M2 = 1000, N1b = 70, N2b = 200, S2 = 3;
        const DropCfg dc = make_drop(0.5f, 7);
        ref_tn_kernel<<<dim3((N2b + 63) / 64, N1b), 64>>>(P, Y, A, dwref, dbref, M2, N1b, N2b, dc);
        rt::TnArgs p{P, Y, A, slab, dbslab, N1b, N2b, M2, N1b, N2b, (N1b + 319) / 320, (N2b + 127) / 128,
                     ((M2 + S2 - 1) / S2 + 15) / 16 * 16};
        hipLaunchKernelGGL((rt::gemm_tn_kernel<5, 2, true, true>), dim3(p.tiles1 * p.tiles2 * S2), dim3(rt::kThreads), 0, 0,
                           p, dc);
        reduce_kernel<<<(N1b * N2b + 255) / 256, 256>>>(slab, dw, (size_t)N1b * N2b, S2);
        reduce_kernel<<<1, 256>>>(dbslab, db, N1b, S2);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        compare("dW ragged 1000 x 70 x 200", dw, dwref, (size_t)N1b * N2b, N2b);
        compare("db ragged", db, dbref, N1b, N1b);
      }
    }
  }
  printf("done\n");
  return 0;
}
