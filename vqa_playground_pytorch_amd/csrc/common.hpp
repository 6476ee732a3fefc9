// Shared host/device helpers for libvqa_mi355x.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/vqa_mi355x.h"

namespace vqa {

// ---- per-thread error text (host) -----------------------------------------------------------
char* error_buffer();  // defined in api.hip, thread_local storage of 512 bytes

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(VQA_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return VQA_OK;
}

// api.hip: value of a tuning / diagnostic knob -- the environment variable of that name as it was at the FIRST query (or
// what vqa_set_option() installed since), nullptr when unset.  Launchers never read the environment themselves.
// The returned string is interned (never mutated or freed), so it stays valid across later vqa_set_option() calls.
const char* option(const char* name);
inline bool option_is(const char* name, char first) {   // knob set and starting with `first` (e.g. "0" = switched off)
  const char* v = option(name);
  return v != nullptr && v[0] == first;
}

// api.hip: per-thread log of the launches of the current C-ABI call (grid in work-items, as rocprofv3's Grid_Size reports
// it) -- bench.py keys the committed PMC tables by (kernel, grid) with it.  A few stores per launch, always on.
void note_launch(const char* kernel, dim3 grid, dim3 block);
#define VQA_LAUNCH(kernel, grid, block, lds, stream, ...)                  \
  do {                                                                      \
    ::vqa::note_launch(#kernel, (grid), (block));                                   \
    hipLaunchKernelGGL(kernel, (grid), (block), (lds), (stream), __VA_ARGS__); \
  } while (0)

int zero_async(void* ptr, size_t bytes, hipStream_t s);  // api.hip: zero-fill kernel (never hipMemsetAsync: see there)

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Raise a kernel's dynamic-LDS cap (default 64 KiB) once per device and size; one static per call site.
#define VQA_ENSURE_LDS(kernel, bytes)                                                                   \
  do {                                                                                                  \
    static std::atomic<size_t> cap_[16];                                                                \
    int dev_ = 0;                                                                                       \
    (void)hipGetDevice(&dev_);                                                                          \
    dev_ &= 15;                                                                                         \
    if ((size_t)(bytes) > 65536 && (size_t)(bytes) > cap_[dev_].load(std::memory_order_relaxed)) {      \
      hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                        \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));    \
      if (e_ != hipSuccess) return ::vqa::fail(VQA_E_LAUNCH, "hipFuncSetAttribute(%s, %zu): %s", #kernel, \
                                               (size_t)(bytes), hipGetErrorString(e_));                 \
      cap_[dev_].store((size_t)(bytes), std::memory_order_relaxed);                                     \
    }                                                                                                   \
  } while (0)

#define VQA_REQUIRE(cond, code, ...) \
  do {                               \
    if (!(cond)) return ::vqa::fail(code, __VA_ARGS__); \
  } while (0)

// ---- device helpers --------------------------------------------------------------------------
constexpr int kWave = 64;
using f32x16 = __attribute__((ext_vector_type(16))) float;  // one 32x32 MFMA accumulator tile per wave

// wave64 reductions at VALU speed: four DPP steps (quad_perm, quad_perm, row_half_mirror, row_mirror) leave every lane
// with the sum of its 16-lane row, four v_readlane combine the rows.  (__shfl_xor lowers to ds_bpermute: an LDS
// round trip per step, ~600 cycles of dependent latency per reduction -- it made the streaming backward kernels
// latency-bound.)  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float x) {
  x += dpp_mov<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_mov<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_mov<0x141>(x);  // row_half_mirror
  x += dpp_mov<0x140>(x);  // row_mirror
  const int xi = __float_as_int(x);
  return __int_as_float(__builtin_amdgcn_readlane(xi, 0)) + __int_as_float(__builtin_amdgcn_readlane(xi, 16)) +
         __int_as_float(__builtin_amdgcn_readlane(xi, 32)) + __int_as_float(__builtin_amdgcn_readlane(xi, 48));
}

// lane-wise sum over the four 16-lane rows of the wave: every lane l ends with x[l & 15] + x[(l & 15) + 16] + ... (gfx950's
// v_permlane16_swap / v_permlane32_swap: the odd rows of one copy change places with the even rows of the other)
__device__ __forceinline__ float rows_sum(float x) {
  const unsigned u = __float_as_uint(x);
  const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const float s = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const unsigned v = __float_as_uint(s);
  const auto b = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// Butterfly reduce-scatter of 16 per-lane values over a 16-lane DPP row: lane l ends with value (l & 15) summed over the 16
// lanes of its row.  15 exchanges (8 + 4 + 2 + 1: quad_perm, quad_perm, row_ror:4, row_ror:8) instead of 16 x 4.
__device__ __forceinline__ float row_reduce_scatter16(const float (&val)[16], int lane) {
  const bool b0 = (lane & 1) != 0, b1 = (lane & 2) != 0, b2 = (lane & 4) != 0, b3 = (lane & 8) != 0;
  float s8[8], s4[4], s2[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) s8[i] = (b0 ? val[2 * i + 1] : val[2 * i]) + dpp_mov<0xB1>(b0 ? val[2 * i] : val[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 4; ++i) s4[i] = (b1 ? s8[2 * i + 1] : s8[2 * i]) + dpp_mov<0x4E>(b1 ? s8[2 * i] : s8[2 * i + 1]);
#pragma unroll
  for (int i = 0; i < 2; ++i) s2[i] = (b2 ? s4[2 * i + 1] : s4[2 * i]) + dpp_mov<0x124>(b2 ? s4[2 * i] : s4[2 * i + 1]);
  return (b3 ? s2[1] : s2[0]) + dpp_mov<0x128>(b3 ? s2[0] : s2[1]);
}

__device__ __forceinline__ float wave_max(float x) {
  x = fmaxf(x, dpp_mov<0xB1>(x));
  x = fmaxf(x, dpp_mov<0x4E>(x));
  x = fmaxf(x, dpp_mov<0x141>(x));
  x = fmaxf(x, dpp_mov<0x140>(x));
  const int xi = __float_as_int(x);
  return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(xi, 0)), __int_as_float(__builtin_amdgcn_readlane(xi, 16))),
               fmaxf(__int_as_float(__builtin_amdgcn_readlane(xi, 32)), __int_as_float(__builtin_amdgcn_readlane(xi, 48))));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }

// bf16 storage (the mixed-precision path: bf16 in HBM, fp32 in registers).  Four elements = one 8-byte access;
// widening is a shift, narrowing is the hardware round-to-nearest-even convert (v_cvt_pk_bf16_f32, NaN stays NaN).
using bf16 = __bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const bf16x2 t = {(__bf16)lo, (__bf16)hi};
  return __builtin_bit_cast(uint32_t, t);
}
__device__ __forceinline__ float4 ld4(const bf16* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16_lo(w.x), bf16_hi(w.x), bf16_lo(w.y), bf16_hi(w.y));
}
__device__ __forceinline__ void st4(bf16* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16(v.x, v.y), pack_bf16(v.z, v.w));
}

__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 scale4(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }

// Bijective XCD-aware remap of a linear workgroup id: workgroups b and b+8 share an XCD (round-robin
// dispatch, speed only -- never correctness), so give every XCD a contiguous chunk of the grid.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, k = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

// ---- counter-based dropout (the fused linear kernels, K1 apply, K3a; K2 keeps its own counter layout) -----------
// The mask of element e of a row-major tensor is a pure function of (seed, e); forward and backward regenerate it.
//   p = 0.5 (p8 == 128, the only rate the models use): ONE BIT per element -- keep iff bit (e & 31) of the hash word of
//     counter e >> 5 is set, kept values scaled by 2.  32 elements per hash word: next to fp32 MFMAs VALU work is not
//     hidden (the matrix pipe and the vector ALU share the FMA datapath), so the hashes per element are what the mask costs.
//   any other rate: one BYTE per element (byte e & 3 of the word of counter e >> 2), keep iff byte >= p8, so the drop
//     probability is realised as round(p*256)/256 and kept values are scaled by 256/(256-p8).
// lowbias32 finaliser (two 32-bit multiplies -- v_mul_lo_u32 is quarter rate, so multiplies are what a hash costs
// here).  A bijection of the 32-bit counter xor-ed with a per-call key, so distinct elements never share a word.
__device__ __forceinline__ uint32_t mask_word32(uint32_t counter, uint32_t key) {
  uint32_t x = counter ^ key;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return x;
}
// 64-bit counters (K2 at very large batches): the high half perturbs the key
__device__ __forceinline__ uint32_t mask_word(uint64_t counter, uint64_t seed) {
  const uint32_t hi = (uint32_t)(counter >> 32);
  const uint32_t key = (uint32_t)seed ^ ((uint32_t)(seed >> 32) * 0x9E3779B9u) ^ (hi * 0x85EBCA6Bu);
  return mask_word32((uint32_t)counter, key);
}
struct DropCfg {
  uint32_t p8;   // drop when byte < p8
  float scale;   // 256 / (256 - p8)
  uint64_t seed;              // host seed, or the salt added to *seed_ptr
  const uint64_t* seed_ptr;   // optional DEVICE word holding the step's seed (graph replays: the host value is frozen at
                              // capture, the device word moves on); the effective seed is *seed_ptr + seed
  __device__ __forceinline__ uint64_t effective() const { return seed_ptr != nullptr ? seed_ptr[0] + seed : seed; }
};

inline DropCfg make_drop(float p, uint64_t seed, const uint64_t* seed_ptr = nullptr) {
  int p8 = (int)(p * 256.f + 0.5f);
  if (p8 < 0) p8 = 0;
  if (p8 > 255) p8 = 255;
  return DropCfg{(uint32_t)p8, 256.f / (256.f - (float)p8), seed, seed_ptr};
}


__device__ __forceinline__ uint32_t drop_key(const DropCfg& dc) {
  const uint64_t s = dc.effective();
  return (uint32_t)s ^ ((uint32_t)(s >> 32) * 0x9E3779B9u);
}
constexpr uint32_t kDropHalf = 128u;   // p8 of p = 0.5: the one-bit-per-element form
// factor (0 or 1/(1-p)) of element e of a row-major tensor with < 2^32 elements
__device__ __forceinline__ float drop_one(uint32_t e, const DropCfg& dc) {
  if (dc.p8 == kDropHalf) return ((mask_word32(e >> 5, drop_key(dc)) >> (e & 31u)) & 1u) != 0u ? 2.f : 0.f;
  const uint32_t w = mask_word32(e >> 2, drop_key(dc)) >> (8 * (e & 3));
  return (w & 255u) >= dc.p8 ? dc.scale : 0.f;
}
// the two consecutive elements e (even) and e + 1
__device__ __forceinline__ float2 drop_pair(uint32_t e, const DropCfg& dc) {
  if (dc.p8 == kDropHalf) {
    const uint32_t w = mask_word32(e >> 5, drop_key(dc)) >> (e & 31u);
    return make_float2((w & 1u) != 0u ? 2.f : 0.f, (w & 2u) != 0u ? 2.f : 0.f);
  }
  const uint32_t w = mask_word32(e >> 2, drop_key(dc)) >> (8 * (e & 3));
  return make_float2((w & 255u) >= dc.p8 ? dc.scale : 0.f, ((w >> 8) & 255u) >= dc.p8 ? dc.scale : 0.f);
}

// the four consecutive elements e .. e + 3 (e a multiple of 4): one hash word either way
__device__ __forceinline__ float4 drop_quad(uint32_t e, const DropCfg& dc) {
  if (dc.p8 == kDropHalf) {
    const uint32_t w = mask_word32(e >> 5, drop_key(dc)) >> (e & 31u);
    return make_float4((w & 1u) != 0u ? 2.f : 0.f, (w & 2u) != 0u ? 2.f : 0.f, (w & 4u) != 0u ? 2.f : 0.f, (w & 8u) != 0u ? 2.f : 0.f);
  }
  const uint32_t w = mask_word32(e >> 2, drop_key(dc));
  return make_float4((w & 255u) >= dc.p8 ? dc.scale : 0.f, ((w >> 8) & 255u) >= dc.p8 ? dc.scale : 0.f,
                     ((w >> 16) & 255u) >= dc.p8 ? dc.scale : 0.f, (w >> 24) >= dc.p8 ? dc.scale : 0.f);
}

}  // namespace vqa
