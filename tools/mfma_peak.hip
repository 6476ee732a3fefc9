// Raw v_mfma_f32_32x32x2_f32 issue-rate probe: how many TFLOP/s does the chip sustain with operands in registers?
// hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NACC, int WITH_LDS>
__global__ __launch_bounds__(256) void probe(float* out, int iters, float seed) {
  __shared__ float lds[16 * 130 * 2];
  for (int i = threadIdx.x; i < 16 * 130 * 2; i += 256) lds[i] = seed + i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) acc[a][e] = 0.f;
  float x = seed + threadIdx.x, y = seed * 2 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    float av[8], bv[8];
    if (WITH_LDS) {
#pragma unroll
      for (int kp = 0; kp < 8; ++kp) {
        av[kp] = lds[(kp * 2 + (lane >> 5)) * 130 + (lane & 31) + (it & 1) * 16 * 130];
        bv[kp] = lds[(kp * 2 + (lane >> 5)) * 130 + 64 + (lane & 31) + (it & 1) * 16 * 130];
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int kp = 0; kp < 8; ++kp) { av[kp] = x + kp; bv[kp] = y - kp; }
    }
#pragma unroll
    for (int kp = 0; kp < 8; ++kp)
#pragma unroll
      for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kp], bv[kp] + a, acc[a], 0, 0, 0);
    if (WITH_LDS == 2) __syncthreads();
  }
  float s = 0.f;
  for (int a = 0; a < NACC; ++a)
    for (int e = 0; e < 16; ++e) s += acc[a][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int WITH_LDS>
void run(const char* name, int blocks) {
  float* out;
  hipMalloc(&out, blocks * 256 * sizeof(float));
  const int iters = 4000;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  probe<NACC, WITH_LDS><<<blocks, 256>>>(out, 10, 1.f);
  hipDeviceSynchronize();
  hipEventRecord(a);
  probe<NACC, WITH_LDS><<<blocks, 256>>>(out, iters, 1.f);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double flop = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
  printf("%-44s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flop / ms / 1e9);
  hipFree(out);
}

int main() {
  run<4, 0>("regs, 4 acc, 1 wave/SIMD", 256);
  run<1, 0>("regs, 1 acc, 1 wave/SIMD", 256);
  run<1, 0>("regs, 1 acc, 4 waves/SIMD", 1024);
  run<4, 0>("regs, 4 acc, 2 waves/SIMD", 512);
  run<4, 1>("lds frags, 4 acc, 1 wave/SIMD", 256);
  run<4, 2>("lds frags + barrier, 4 acc, 1 wave/SIMD", 256);
  run<1, 1>("lds frags, 1 acc, 4 waves/SIMD", 1024);
  run<1, 2>("lds frags + barrier, 1 acc, 4 waves/SIMD", 1024);
  return 0;
}
