// Micro-probe: issue rate of v_mfma_f32_16x16x4_f32 chains of the shape the K4 weight-gradient kernel uses (one wave
// per SIMD, NB independent accumulators, the A operand shared by the NB MFMAs of a step).  Prints shader cycles per MFMA.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NB, int MODE>
__global__ __launch_bounds__(256, 1) void probe(const float* in, float* out, unsigned long long* stamps, int iters) {
  const int lane = threadIdx.x & 63;
  float a[9], b[9][NB];
  for (int s = 0; s < 9; ++s) {
    a[s] = in[lane + 64 * s];
    for (int q = 0; q < NB; ++q) b[s][q] = in[lane * 7 + s * NB + q];
  }
  f32x4 P[NB];
  for (int q = 0; q < NB; ++q) P[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool one = in[lane] > 0.5f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      if (MODE & 1) {   // the two selects of the ones column
        b[s][NB - 1] = one ? 1.f : b[s][NB - 1];
        b[s][NB - 2] = one ? 0.f : b[s][NB - 2];
      }
#pragma unroll
      for (int q = 0; q < NB; ++q) P[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s][q], (MODE & 2) && s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : P[q], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE & 4) {     // perturb the operands like a reload would
#pragma unroll
      for (int s = 0; s < 9; ++s) a[s] += 1.f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 acc = P[0];
  for (int q = 1; q < NB; ++q) acc += P[q];
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (lane == 0) {
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0;
    stamps[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0;
  }
}

template <int NB, int MODE>
int run(const char* name, int grid, int iters) {
  float *in, *out;
  unsigned long long* st;
  CK(hipMalloc(&in, 1 << 20));
  CK(hipMemset(in, 0, 1 << 20));
  CK(hipMalloc(&out, grid * 256 * 4));
  CK(hipMalloc(&st, grid * 4 * 2 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<NB, MODE>), dim3(grid), dim3(256), 0, 0, in, out, st, iters);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<NB, MODE>), dim3(grid), dim3(256), 0, 0, in, out, st, iters);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * 8);
  CK(hipMemcpy(h.data(), st, grid * 8 * 8, hipMemcpyDeviceToHost));
  double cyc = 0, ticks = 0;
  for (int i = 0; i < grid * 4; ++i) { cyc += h[2 * i]; ticks += h[2 * i + 1]; }
  cyc /= grid * 4; ticks /= grid * 4;
  const double n = (double)iters * 9 * NB;
  printf("%-40s grid %4d: %.1f us, %.2f shader cyc / MFMA, %.2f ns / MFMA (100 MHz ticks), clock %.2f GHz\n", name, grid, ms * 1e3, cyc / n,
         ticks * 10.0 / n, cyc / (ticks * 10.0));
  CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(st));
  return 0;
}

int main() {
  for (int grid : {1, 256}) {
    run<10, 0>("10 acc, plain", grid, 320);
    run<10, 1>("10 acc, 2 selects / step", grid, 320);
    run<10, 3>("10 acc, selects, C=0 at step 0", grid, 320);
    run<10, 7>("10 acc, selects, C=0, operand updates", grid, 320);
    run<5, 0>("5 acc, plain", grid, 640);
    run<20, 0>("20 acc, plain", grid, 160);
  }
  return 0;
}
