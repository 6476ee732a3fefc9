// Standalone probe of the split-bf16 fp32 engine (csrc/gemm_f32_split.hpp) next to the fp32-MFMA register-tile engine
// (csrc/gemm_f32_rt.hpp): accuracy of both against an fp64-accumulating reference and timing at the region-projection
// shape (M = 18432, K = 2048, N = 310).  No torch, no library GEMM.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/split_probe.hip -o tools/split_probe && tools/split_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../vqa_playground_pytorch_amd/csrc/gemm_f32_split.hpp"

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

// uniform [-1, 1) * scale, with a wide spread of magnitudes when `wide` (exponent drawn from 2^-12 .. 2^0)
__global__ void fill_kernel(float* p, size_t n, uint32_t seed, float scale, int wide) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint32_t h = mask_word32((uint32_t)i, seed);
    float v = ((float)(h >> 8) * (1.f / 8388608.f) - 1.f) * scale;
    if (wide) v = ldexpf(v, -(int)(mask_word32((uint32_t)i, seed ^ 0x9e3779b9u) % 13u));
    p[i] = v;
  }
}

__global__ void ref_nt_kernel(const float* A, int lda, const float* B, int ldb, const float* bias, double* y, int M, int N,
                              int K, int act, DropCfg dc) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
  if (n >= N) return;
  double s = 0.0;
  for (int k = 0; k < K; ++k) {
    float a = A[(size_t)m * lda + k];
    if (dc.p8 > 0) a *= drop_one((uint32_t)m * (uint32_t)K + (uint32_t)k, dc);
    s += (double)a * (double)B[(size_t)n * ldb + k];
  }
  double v = s + (bias ? (double)bias[n] : 0.0);
  if (act == 1) v = v > 0 ? v : 0;
  y[(size_t)m * N + n] = v;
}

// dw[n1][n2] = sum_m gate(G)[m][n1] * drop(X)[m][n2] (fp64); db[n1] = sum_m gate(G)[m][n1]
__global__ void ref_tn_kernel(const float* G, const float* Y, const float* X, double* dw, double* db, int M, int N1, int N2,
                              DropCfg dc) {
  const int n2 = blockIdx.x * blockDim.x + threadIdx.x, n1 = blockIdx.y;
  if (n2 >= N2) return;
  double s = 0.0, sb = 0.0;
  for (int m = 0; m < M; ++m) {
    float a = G[(size_t)m * N1 + n1];
    if (Y != nullptr && !(Y[(size_t)m * N1 + n1] > 0.f)) a = 0.f;
    float q = X[(size_t)m * N2 + n2];
    if (dc.p8 > 0) q *= drop_one((uint32_t)m * (uint32_t)N2 + (uint32_t)n2, dc);
    s += (double)a * (double)q;
    sb += (double)a;
  }
  dw[(size_t)n1 * N2 + n2] = s;
  if (n2 == 0) db[n1] = sb;
}

// error of an fp32 result against the fp64 reference: max and rms, relative to the rms of the reference
static void compare(const char* what, const float* got, const double* ref, size_t n, int ld) {
  std::vector<float> hg(n);
  std::vector<double> hr(n);
  CK(hipMemcpy(hg.data(), got, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hr.data(), ref, n * 8, hipMemcpyDeviceToHost));
  double ss = 0, se = 0, maxerr = 0, maxref = 0;
  size_t worst = 0;
  for (size_t i = 0; i < n; ++i) {
    const double e = fabs((double)hg[i] - hr[i]);
    ss += hr[i] * hr[i];
    se += e * e;
    maxref = fmax(maxref, fabs(hr[i]));
    if (!(e <= maxerr)) {
      maxerr = e;
      worst = i;
    }
  }
  const double rms = sqrt(ss / n);
  printf("  %-44s rms|ref| %.4g max|ref| %.4g  err: max %.3e (%.2e of rms)  rms %.3e (%.2e of rms) worst [%zu,%zu]\n", what, rms,
         maxref, maxerr, maxerr / rms, sqrt(se / n), sqrt(se / n) / rms, worst / ld, worst % ld);
}

template <class F>
static float time_ms(F&& f, int iters) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int i = 0; i < 10; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / iters;
}

static void fill(float* p, size_t n, uint32_t seed, float scale, int wide = 0) {
  fill_kernel<<<(unsigned)((n + 255) / 256), 256>>>(p, n, seed, scale, wide);
}

template <int RB, int CB, int WM, int WN, int WK, bool DROP, int TUNE = 0>
static void launch_rt(const float* A, int lda, const float* B, int ldb, const float* bias, float* y, int M, int N, int K,
                      int act, DropCfg dc) {
  using S = rt::NtShape<RB, CB, WM, WN, WK>;
  const int tiles_m = (M + S::BM - 1) / S::BM, tiles_n = (N + S::BN - 1) / S::BN;
  rt::NtArgs p{A, B, lda, ldb, M, N, K, tiles_n, nullptr};
  auto kern = rt::gemm_nt_kernel<RB, CB, WM, WN, WK, DROP, EpiBiasRelu, TUNE>;
  static bool once = false;
  if (!once && S::kLdsBytes > 65536) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::kLdsBytes));
    once = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(rt::kThreads), S::kLdsBytes, 0, p, dc,
                     EpiBiasRelu{y, bias, N, act, (DROP && dc.p8 > 0) ? dc.scale : 1.f});
}

static void pack_dot(const float* B, int ldb, int N, int K, sp::u32x4* Bp) {
  const long threads = (long)((N + 15) / 16) * (K / sp::kChunk) * 64;
  sp::split_pack_kernel<true><<<(unsigned)((threads + 255) / 256), 256>>>(B, ldb, N, K, Bp);
}
static void pack(const float* B, int ldb, int N, int K, sp::u32x4* Bp) {
  const long threads = (long)((N + 15) / 16) * (K / sp::kChunk) * 64;
  sp::split_pack_kernel<false><<<(unsigned)((threads + 255) / 256), 256>>>(B, ldb, N, K, Bp);
}
// the dot form of the split against the plain one, bit for bit, on the planes of a whole matrix
static void check_dot_form(const float* X, int N, int K, const char* what) {
  const size_t bytes = sp::packed_bytes(N, K);
  sp::u32x4 *p0, *p1;
  CK(hipMalloc(&p0, bytes));
  CK(hipMalloc(&p1, bytes));
  pack_dot(X, K, N, K, p0);
  pack(X, K, N, K, p1);
  std::vector<uint32_t> h0(bytes / 4), h1(bytes / 4);
  CK(hipMemcpy(h0.data(), p0, bytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(h1.data(), p1, bytes, hipMemcpyDeviceToHost));
  size_t bad = 0;
  for (size_t i = 0; i < h0.size(); ++i) bad += h0[i] != h1[i];
  printf("  planes by v_dot2c_f32_bf16 vs shift/and/subtract, %s: %zu of %zu words differ  %s\n", what, bad, h0.size(), bad ? "FAIL" : "OK");
  CK(hipFree(p0));
  CK(hipFree(p1));
}

template <int RB, int CB, int WM, int WN, int WK, bool DROP, int TUNE = 0, int NR = RB, bool SHARE_A = false>
static void launch_sp(const float* A, int lda, const sp::u32x4* Bp, const float* bias, float* y, int M, int N, int K, int act,
                      DropCfg dc) {
  using S = rt::NtShape<RB, CB, WM, WN, WK>;
  const int tiles_m = (M + S::BM - 1) / S::BM, tiles_n = (N + S::BN - 1) / S::BN;
  sp::NtArgs p{A, Bp, lda, M, N, K, tiles_n};
  auto kern = sp::gemm_nt_kernel<RB, CB, WM, WN, WK, DROP, EpiBiasRelu, TUNE, NR, SHARE_A>;
  constexpr size_t lds = sp::nt_lds_bytes<RB, WN, WK, SHARE_A>(S::kLdsBytes);
  static bool once = false;
  if (!once && lds > 65536) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  hipLaunchKernelGGL(kern, dim3(tiles_m * tiles_n), dim3(sp::kThreads), lds, 0, p, dc,
                     EpiBiasRelu{y, bias, N, act, (DROP && dc.p8 > 0) ? dc.scale : 1.f});
}

template <bool MASK, bool DROP>
static void launch_rt_tn(const float* P, const float* Yg, const float* Q, float* slab, float* dbslab, float* dw, float* db, int M,
                         int N1, int N2, int S, DropCfg dc) {
  int rows = (M + S - 1) / S;
  rows = (rows + 15) / 16 * 16;
  rt::TnArgs a{P, Yg, Q, slab, dbslab, N1, N2, M, N1, N2, (N1 + 319) / 320, (N2 + 127) / 128, rows};
  hipLaunchKernelGGL((rt::gemm_tn_kernel<5, 2, MASK, DROP, DROP, DROP ? 1 : 0>), dim3(a.tiles1 * a.tiles2 * S), dim3(rt::kThreads), 0, 0,
                     a, dc);
  const int NK = N1 * N2;
  hipLaunchKernelGGL(sp::slab_sum_kernel, dim3(sp::slab_sum_blocks(NK, N1)), dim3(256), 0, 0, slab, dbslab, dw, db, NK, N1, S, S, 1.f);
}
template <bool DROP, int TUNE = 0>
static void launch_sp_tn(const float* G, const float* Yg, const float* X, void* ws, float* dw, float* db, int M, int N1, int N2,
                         int S, DropCfg dc, bool pack_only = false, bool gemm_only = false) {
  const sp::TnPlan pl = sp::tn_plan(M, S);
  const int nblocks = (N1 + 15) / 16;
  sp::u32x4* gp = (sp::u32x4*)ws;
  float* slab = (float*)((char*)ws + ((sp::packed_tn_bytes(pl.slabs, pl.cps, N1) + 255) & ~(size_t)255));
  float* dbslab = slab + (size_t)pl.slabs * N1 * N2;
  if (!gemm_only) {
    const size_t lds = sp::pack_tn_lds_bytes(nblocks);
    if (Yg)
      hipLaunchKernelGGL((sp::pack_tn_kernel<true, false>), dim3(pl.slabs * sp::kPackParts), dim3(256), lds, 0, G, Yg, N1, M, N1, nblocks, pl.cps, gp, dbslab, (float*)nullptr, nblocks);
    else
      hipLaunchKernelGGL((sp::pack_tn_kernel<false, false>), dim3(pl.slabs * sp::kPackParts), dim3(256), lds, 0, G, Yg, N1, M, N1, nblocks, pl.cps, gp, dbslab, (float*)nullptr, nblocks);
  }
  if (pack_only) return;
  const int tiles1 = (nblocks + 19) / 20, tiles2 = (N2 + 127) / 128;
  sp::TnArgs a{gp, X, slab, N2, M, N1, N2, nblocks, pl.cps, tiles1, tiles2};
  if constexpr ((TUNE & 256) != 0)
    hipLaunchKernelGGL((sp::gemm_tn_shared_kernel<5, DROP, (TUNE & 255)>), dim3(tiles1 * tiles2 * pl.slabs), dim3(sp::kThreads), sp::kTnSharedLds, 0, a, dc);
  else
    hipLaunchKernelGGL((sp::gemm_tn_kernel<5, 2, DROP, TUNE>), dim3(tiles1 * tiles2 * pl.slabs), dim3(sp::kThreads), 0, 0, a, dc);
  if (gemm_only) return;
  const int NK = N1 * N2;
  hipLaunchKernelGGL(sp::slab_sum_kernel, dim3(sp::slab_sum_blocks(NK, N1)), dim3(256), 0, 0, slab, dbslab, dw, db, NK, N1, pl.slabs,
                     pl.slabs * sp::kPackParts, DROP ? dc.scale : 1.f);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);

  const int M = 18432, K = 2048, N = 310;
  float *A, *B, *bias, *y;
  double* yref;
  sp::u32x4* Bp;
  CK(hipMalloc(&A, (size_t)M * K * 4));
  CK(hipMalloc(&B, (size_t)N * K * 4));
  CK(hipMalloc(&Bp, sp::packed_bytes(N, K)));
  CK(hipMalloc(&bias, N * 4));
  CK(hipMalloc(&y, (size_t)M * N * 4));
  CK(hipMalloc(&yref, (size_t)M * N * 8));
  const double flop = 2.0 * M * K * N;
  for (int wide = 0; wide < 2; ++wide) {
    fill(A, (size_t)M * K, 11, 1.f, wide);
    fill(B, (size_t)N * K, 22, 1.f / 32.f, wide);
    fill(bias, N, 33, 0.5f);
    pack(B, K, N, K, Bp);
    printf("operands: %s\n", wide ? "magnitudes spread over 2^-12 .. 1" : "uniform");
    check_dot_form(A, 4096, K, "A rows 0..4095");
    check_dot_form(B, N, K, "B");
    for (int drop = 0; drop < 2; ++drop) {
      const DropCfg dc = make_drop(drop ? 0.5f : 0.f, 0x1234567ull);
      ref_nt_kernel<<<dim3((N + 63) / 64, M), 64>>>(A, K, B, K, bias, yref, M, N, K, 0, dc);
      CK(hipMemset(y, 0xff, (size_t)M * N * 4));
      if (drop)
        launch_rt<9, 5, 1, 2, 2, true>(A, K, B, K, bias, y, M, N, K, 0, dc);
      else
        launch_rt<9, 5, 1, 2, 2, false>(A, K, B, K, bias, y, M, N, K, 0, dc);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare(drop ? "fp32 MFMA 9x5, dropout" : "fp32 MFMA 9x5", y, yref, (size_t)M * N, N);
      CK(hipMemset(y, 0xff, (size_t)M * N * 4));
      if (drop)
        launch_sp<9, 5, 1, 2, 2, true>(A, K, Bp, bias, y, M, N, K, 0, dc);
      else
        launch_sp<9, 5, 1, 2, 2, false>(A, K, Bp, bias, y, M, N, K, 0, dc);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare(drop ? "split bf16 (6 products) 9x5, dropout" : "split bf16 (6 products) 9x5", y, yref, (size_t)M * N, N);
      CK(hipMemset(y, 0xff, (size_t)M * N * 4));
      if (drop)
        launch_sp<9, 5, 1, 2, 2, true, 0, 6>(A, K, Bp, bias, y, M, N, K, 0, dc);
      else
        launch_sp<9, 5, 1, 2, 2, false, 0, 6>(A, K, Bp, bias, y, M, N, K, 0, dc);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare(drop ? "split bf16 9x5 ring 6, dropout" : "split bf16 9x5 ring 6", y, yref, (size_t)M * N, N);
      CK(hipMemset(y, 0xff, (size_t)M * N * 4));
      if (drop)
        launch_sp<9, 5, 1, 2, 2, true, 0, 3, true>(A, K, Bp, bias, y, M, N, K, 0, dc);
      else
        launch_sp<9, 5, 1, 2, 2, false, 0, 3, true>(A, K, Bp, bias, y, M, N, K, 0, dc);
      CK(hipDeviceSynchronize());
      CK(hipGetLastError());
      compare(drop ? "split bf16 9x5 SHARED A, dropout" : "split bf16 9x5 SHARED A", y, yref, (size_t)M * N, N);
    }
  }
  const DropCfg dc0 = make_drop(0.f, 0), dch = make_drop(0.5f, 777);
#define T(label, ...)                                                                                  \
  {                                                                                                    \
    float ms = time_ms([&] { __VA_ARGS__; }, iters);                                                   \
    printf("  %-52s %.1f us  %.1f TF/s\n", label, ms * 1e3, flop / ms / 1e9);                           \
  }
  T("pack B (310 x 2048)", pack(B, K, N, K, Bp));
  T("fp32 MFMA 9x5", (launch_rt<9, 5, 1, 2, 2, false>(A, K, B, K, bias, y, M, N, K, 1, dc0)));
  T("fp32 MFMA 9x5 dropout", (launch_rt<9, 5, 1, 2, 2, true>(A, K, B, K, bias, y, M, N, K, 1, dch)));
  T("split 9x5", (launch_sp<9, 5, 1, 2, 2, false>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 dropout", (launch_sp<9, 5, 1, 2, 2, true>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 6", (launch_sp<9, 5, 1, 2, 2, false, 0, 6>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 6 dropout", (launch_sp<9, 5, 1, 2, 2, true, 0, 6>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 3", (launch_sp<9, 5, 1, 2, 2, false, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 6, no split no loads (TUNE 35)", (launch_sp<9, 5, 1, 2, 2, false, 35, 6>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 3 dropout", (launch_sp<9, 5, 1, 2, 2, true, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 3, dot residuals (TUNE 64)", (launch_sp<9, 5, 1, 2, 2, false, 64, 3>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 3, dot residuals, dropout", (launch_sp<9, 5, 1, 2, 2, true, 64, 3>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 3 (plain residuals)", (launch_sp<9, 5, 1, 2, 2, false, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 3 dropout (plain residuals)", (launch_sp<9, 5, 1, 2, 2, true, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 3, 1 VALU per MFMA (TUNE 4)", (launch_sp<9, 5, 1, 2, 2, false, 4, 3>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 3, 1 VALU per MFMA, dropout", (launch_sp<9, 5, 1, 2, 2, true, 4, 3>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 3, 3 VALU per MFMA (TUNE 12)", (launch_sp<9, 5, 1, 2, 2, false, 12, 3>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 3 (2 VALU per MFMA)", (launch_sp<9, 5, 1, 2, 2, false, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 3 dropout (2 VALU per MFMA)", (launch_sp<9, 5, 1, 2, 2, true, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5, A split shared by the wave pair", (launch_sp<9, 5, 1, 2, 2, false, 0, 3, true>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5, shared, dropout", (launch_sp<9, 5, 1, 2, 2, true, 0, 3, true>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 3 once more", (launch_sp<9, 5, 1, 2, 2, false, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  T("split 9x5 ring 3 dropout once more", (launch_sp<9, 5, 1, 2, 2, true, 0, 3>(A, K, Bp, bias, y, M, N, K, 1, dch)));
  T("split 9x5 ring 6 again", (launch_sp<9, 5, 1, 2, 2, false, 0, 6>(A, K, Bp, bias, y, M, N, K, 1, dc0)));
  // ---------------------------------------------------------------- TN: weight gradient dW = gate(G)^T drop(X)
  {
    const int N1 = 310, N2 = 2048, S = 16;
    float *G, *Yg, *dw, *db, *slab_rt, *dbslab_rt;
    double *dwref, *dbref;
    void* ws;
    const sp::TnPlan pl = sp::tn_plan(M, S);
    const size_t ws_bytes = sp::packed_tn_bytes(pl.slabs, pl.cps, N1) + 256 + (size_t)pl.slabs * N1 * N2 * 4 + (size_t)pl.slabs * sp::kPackParts * N1 * 4 + 256;
    CK(hipMalloc(&G, (size_t)M * N1 * 4));
    CK(hipMalloc(&Yg, (size_t)M * N1 * 4));
    CK(hipMalloc(&dw, (size_t)N1 * N2 * 4));
    CK(hipMalloc(&db, N1 * 4));
    CK(hipMalloc(&dwref, (size_t)N1 * N2 * 8));
    CK(hipMalloc(&dbref, N1 * 8));
    CK(hipMalloc(&slab_rt, (size_t)S * N1 * N2 * 4));
    CK(hipMalloc(&dbslab_rt, (size_t)S * N1 * 4));
    CK(hipMalloc(&ws, ws_bytes));
    printf("TN: M %d, N1 %d, N2 %d, %d slabs of %d chunks\n", M, N1, N2, pl.slabs, pl.cps);
    for (int wide = 0; wide < 2; ++wide) {
      fill(A, (size_t)M * N2, 11, 1.f, wide);
      fill(G, (size_t)M * N1, 66, 1.f / 64.f, wide);
      fill(Yg, (size_t)M * N1, 77, 1.f);
      for (int mode = 0; mode < 2; ++mode) {   // 0: plain; 1: relu gate + dropout
        const DropCfg dc = make_drop(mode ? 0.5f : 0.f, 0x7654321ull);
        const float* yy = mode ? Yg : nullptr;
        ref_tn_kernel<<<dim3((N2 + 63) / 64, N1), 64>>>(G, yy, A, dwref, dbref, M, N1, N2, dc);
        CK(hipMemset(dw, 0xff, (size_t)N1 * N2 * 4));
        if (mode) {
          launch_rt_tn<true, true>(G, Yg, A, slab_rt, dbslab_rt, dw, db, M, N1, N2, S, dc);
          hipLaunchKernelGGL(sp::slab_sum_kernel, dim3(sp::slab_sum_blocks(N1 * N2, N1)), dim3(256), 0, 0, slab_rt, dbslab_rt, dw, db, N1 * N2, N1, S, S, 2.f);
        } else {
          launch_rt_tn<false, false>(G, nullptr, A, slab_rt, dbslab_rt, dw, db, M, N1, N2, S, dc);
        }
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        compare(mode ? "dW fp32 MFMA 5x2, gate + dropout" : "dW fp32 MFMA 5x2", dw, dwref, (size_t)N1 * N2, N2);
        compare(mode ? "db fp32 MFMA, gate" : "db fp32 MFMA", db, dbref, N1, N1);
        CK(hipMemset(dw, 0xff, (size_t)N1 * N2 * 4));
        CK(hipMemset(db, 0xff, N1 * 4));
        if (mode)
          launch_sp_tn<true>(G, Yg, A, ws, dw, db, M, N1, N2, S, dc);
        else
          launch_sp_tn<false>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dc);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        compare(mode ? "dW split, gate + dropout" : "dW split", dw, dwref, (size_t)N1 * N2, N2);
        compare(mode ? "db split, gate" : "db split", db, dbref, N1, N1);
        CK(hipMemset(dw, 0xff, (size_t)N1 * N2 * 4));
        if (mode)
          launch_sp_tn<true, 256>(G, Yg, A, ws, dw, db, M, N1, N2, S, dc);
        else
          launch_sp_tn<false, 256>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dc);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        compare(mode ? "dW split SHARED, gate + dropout" : "dW split SHARED", dw, dwref, (size_t)N1 * N2, N2);
      }
    }
    const double flop2 = 2.0 * M * N1 * N2;
    const DropCfg dcz = make_drop(0.f, 0), dch2 = make_drop(0.5f, 777);
#define T2(label, ...)                                                                                  \
  {                                                                                                    \
    float ms = time_ms([&] { __VA_ARGS__; }, iters);                                                   \
    printf("  %-52s %.1f us  %.1f TF/s\n", label, ms * 1e3, flop2 / ms / 1e9);                          \
  }
    T2("dW fp32 MFMA 5x2 + slab sum", (launch_rt_tn<false, false>(G, nullptr, A, slab_rt, dbslab_rt, dw, db, M, N1, N2, S, dcz)));
    T2("dW fp32 MFMA 5x2 gate + dropout + slab sum", (launch_rt_tn<true, true>(G, Yg, A, slab_rt, dbslab_rt, dw, db, M, N1, N2, S, dch2)));
    T2("dW split: pack + gemm + slab sum", (launch_sp_tn<false>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz)));
    T2("dW split gate + dropout: pack + gemm + slab sum", (launch_sp_tn<true>(G, Yg, A, ws, dw, db, M, N1, N2, S, dch2)));
    T2("  pack only", (launch_sp_tn<false>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, true)));
    T2("  pack only, gate", (launch_sp_tn<false>(G, Yg, A, ws, dw, db, M, N1, N2, S, dcz, true)));
    T2("  gemm only", (launch_sp_tn<false>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, dropout", (launch_sp_tn<true>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dch2, false, true)));
    auto slab_sum_only = [&] {
      hipLaunchKernelGGL(sp::slab_sum_kernel, dim3(sp::slab_sum_blocks(N1 * N2, N1)), dim3(256), 0, 0, slab_rt, dbslab_rt, dw, db, N1 * N2, N1, S, S, 1.f);
    };
    T2("  slab sum only", slab_sum_only());
    T2("  gemm only, X split shared through LDS", (launch_sp_tn<false, 256>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, shared, dropout", (launch_sp_tn<true, 256>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dch2, false, true)));
    T2("  gemm only, shared, unpinned (TUNE 64)", (launch_sp_tn<false, 256 + 64>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, shared, unpinned, dropout", (launch_sp_tn<true, 256 + 64>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dch2, false, true)));
    T2("  gemm only, shared, no X loads", (launch_sp_tn<false, 258>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, shared, no G loads", (launch_sp_tn<false, 288>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, 1 VALU per MFMA (TUNE 4)", (launch_sp_tn<false, 4>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, 1 VALU per MFMA, dropout", (launch_sp_tn<true, 4>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dch2, false, true)));
    T2("  gemm only, 3 VALU per MFMA (TUNE 12)", (launch_sp_tn<false, 12>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only again", (launch_sp_tn<false>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, dropout again", (launch_sp_tn<true>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dch2, false, true)));
    T2("  gemm only, no X split (TUNE 1)", (launch_sp_tn<false, 1>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, no X loads (TUNE 2)", (launch_sp_tn<false, 2>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, no G loads (TUNE 32)", (launch_sp_tn<false, 32>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
    T2("  gemm only, no loads no split (TUNE 35)", (launch_sp_tn<false, 35>(G, nullptr, A, ws, dw, db, M, N1, N2, S, dcz, false, true)));
  }
  return 0;
}
