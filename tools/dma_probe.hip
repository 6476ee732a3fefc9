// Micro-probe (round 4): do 16-byte global loads -- to registers and straight to LDS (buffer_load_dwordx4 ... lds) -- accept
// source addresses that are only 8- or 4-byte aligned, and what do they cost?  The [B,.] layers' operands have row pitches of
// 310 / 510 / 155 floats: rows start 8- (or 4-) byte aligned, and the grouped GEMM's staging wants 16-byte pieces.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/dma_probe.hip -o tools/dma_probe && tools/dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
using rsrc_t = __amdgpu_buffer_rsrc_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows of `pitch` floats; a workgroup stages ROWS rows x 32 floats (128 B) per step into LDS, then sums what it staged
template <int MODE>   // 0: DMA x4, 1: register x4 + ds_write_b128, 2: register x2 + ds_write_b64
__global__ __launch_bounds__(256) void stage(const float* __restrict__ src, float* __restrict__ out, int pitch, int rows, int steps, int shift) {
  __shared__ __attribute__((aligned(16))) float lds[2][64 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, (int)0x7FFFFFFF, 0x00020000);
  float acc = 0.f;
  for (int s = 0; s < steps; ++s) {
    float* buf = lds[s & 1];
    const int row0 = (blockIdx.x * 64 + 0) % rows;
    // 64 rows x 128 B: wave w stages rows 16 w .. 16 w + 15, lane l = (row l >> 3 ... ) 8 lanes per row
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int r = 16 * wave + 8 * h + (lane >> 3), piece = lane & 7;
      const uint32_t goff = ((uint32_t)((row0 + r) % rows) * (uint32_t)pitch + (uint32_t)shift + 32u * (uint32_t)s + 4u * piece) * 4u;
      float* dst = buf + (16 * wave + 8 * h) * 32;      // wave-uniform base; lane l lands at + 4 l floats
      if (MODE == 0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, (int)goff, 0, 0, 0);
      } else if (MODE == 1) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)goff, 0, 0));
        *reinterpret_cast<f32x4*>(dst + 4 * lane) = v;
      } else {
        const float2 a = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(src) + goff);
        const float2 b = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(src) + goff + 8);
        *reinterpret_cast<float2*>(dst + 4 * lane) = a;
        *reinterpret_cast<float2*>(dst + 4 * lane + 2) = b;
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 64 * 32; i += 256) acc += buf[i];
    __syncthreads();
  }
  out[blockIdx.x * 256 + tid] = acc;
}

template <int MODE>
int run(const char* name, int pitch, int shift) {
  const int rows = 4096, steps = 8, grid = 1024;
  std::vector<float> h((size_t)rows * pitch + 4096);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 17) - 8.f;
  float *src, *out;
  CK(hipMalloc(&src, h.size() * 4));
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, grid * 256 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((stage<MODE>), dim3(grid), dim3(256), 0, 0, src, out, pitch, rows, steps, shift);
  CK(hipEventRecord(e0));
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((stage<MODE>), dim3(grid), dim3(256), 0, 0, src, out, pitch, rows, steps, shift);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> o(grid * 256);
  CK(hipMemcpy(o.data(), out, o.size() * 4, hipMemcpyDeviceToHost));
  // reference: the sum a workgroup staged, spread over its threads -> compare the workgroup totals
  double worst = 0;
  for (int b = 0; b < grid; ++b) {
    double want = 0, got = 0;
    for (int s = 0; s < steps; ++s)
      for (int r = 0; r < 64; ++r)
        for (int k = 0; k < 32; ++k) want += h[(size_t)((b * 64 + r) % rows) * pitch + shift + 32 * s + k];
    for (int t = 0; t < 256; ++t) got += o[b * 256 + t];
    worst = fmax(worst, fabs(want - got));
  }
  const double bytes = (double)grid * steps * 64 * 128;
  printf("%-34s pitch %4d shift %d: %s (max |diff| %.1f), %.2f us / launch, %.2f TB/s staged\n", name, pitch, shift,
         worst < 0.5 ? "CORRECT" : "WRONG", worst, ms * 100.0, bytes / (ms * 1e-4) / 1e12);
  CK(hipFree(src)); CK(hipFree(out));
  return 0;
}

// does an out-of-range lane of a buffer_load ... lds write ZERO into its LDS slot (what the K tail of a staged tile needs)?
__global__ void oob_probe(const float* src, float* out, int valid_bytes) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  const int lane = threadIdx.x;
  lds[4 * lane + 0] = lds[4 * lane + 1] = lds[4 * lane + 2] = lds[4 * lane + 3] = -7.f;
  __syncthreads();
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), (short)0, valid_bytes, 0x00020000);
  const uint32_t off = (lane & 1) ? 0x7FFFFF00u : (uint32_t)lane * 16u;          // odd lanes: far out of range
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, (int)off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[4 * lane + i] = lds[4 * lane + i];
}

int oob() {
  std::vector<float> h(1024, 3.f);
  float *src, *out;
  CK(hipMalloc(&src, 4096));
  CK(hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 1024));
  for (int valid : {4096, 520}) {       // 520: lanes 0..31 in range (lane 32's 16 bytes straddle the end), the rest beyond
    hipLaunchKernelGGL(oob_probe, dim3(1), dim3(64), 0, 0, src, out, valid);
    std::vector<float> o(256);
    CK(hipMemcpy(o.data(), out, 1024, hipMemcpyDeviceToHost));
    printf("OOB probe, num_records %4d B:", valid);
    for (int l : {0, 1, 2, 3, 30, 32, 33, 34, 40, 62, 63}) printf("  lane %d: %g %g %g %g", l, o[4 * l], o[4 * l + 1], o[4 * l + 2], o[4 * l + 3]);
    printf("\n");
  }
  return 0;
}

int main() {
  if (oob()) return 1;
  for (int pitch : {320, 310, 155}) {
    for (int shift : {0, 2, 1}) {
      if (pitch == 310 && shift == 1) continue;
      if (run<0>("DMA dwordx4 -> LDS", pitch, shift)) return 1;
      if (run<1>("buffer_load b128 + ds_write_b128", pitch, shift)) return 1;
      if ((pitch * 4) % 8 == 0 && shift % 2 == 0 && run<2>("2 x float2 + 2 x ds_write_b64", pitch, shift)) return 1;
    }
  }
  return 0;
}
