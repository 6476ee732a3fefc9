// K3a -- attention logits: the dropout + 1x1 conv in front of the softmax of MyATT.
//
// Replaces MyConv1d(fuse_dim, glimpses, 1, 1, p=0.5).forward up to the softmax (config/CoR2.py:72-82 as configured at
// :132): F.dropout on the [B,N,H] fusion output, two transposes, conv1d with G = 4 output channels.  As library calls
// that is a dropout kernel + a GEMM with a 4-wide output (65-95 us at M = 18432: the tile is padded to 16-32 columns)
// and, backward, two more such GEMMs, a mask multiply and a bias reduction.  Here:
//
//   forward : logits[m,g] = bias[g] + sum_k w[g,k] * keep(m,k) * x[m,k]
//   backward: d_x[m,k] = keep(m,k) * sum_g d_logits[m,g] * w[g,k]
//             d_w[g,k] = sum_m d_logits[m,g] * keep(m,k) * x[m,k] ;  d_bias[g] = sum_m d_logits[m,g]
//
// One wave per row at a time, lane = a fixed slice of the K columns (so the G x K weights sit in registers for the
// whole kernel: G * K/64 values per lane), rows taken RB at a time for memory parallelism; keep() is the counter-hash
// dropout of common.hpp (the mask of vqa_linear_dropout_mask for the same seed; never stored).  HBM-bound: x is read
// once forward, once backward (+ d_x written): M*K*4 bytes each.  The weight gradient is accumulated per lane over the
// wave's rows, the 4 waves of a workgroup meet in LDS, workgroup partials are added in a fixed order by a second kernel
// (bitwise reproducible, no atomics).  T = float or bf16 storage of x / d_x.
#include "common.hpp"

namespace vqa {

constexpr int kAlMaxG = 8;
constexpr int kAlThreads = 256;
constexpr int kAlWaves = kAlThreads / 64;

// A lane owns VEC adjacent columns in each of P passes: k = (pass * 64 + lane) * VEC.  K <= 64 * VEC * P.
template <typename T>
struct AlVec;
template <>
struct AlVec<float> {
  static constexpr int VEC = 2, P = 4;
  static __device__ __forceinline__ void ld(const float* p, float (&v)[2]) {
    const float2 t = ld2(p);
    v[0] = t.x;
    v[1] = t.y;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[2]) { st2(p, make_float2(v[0], v[1])); }
};
template <>
struct AlVec<bf16> {
  static constexpr int VEC = 4, P = 2;
  static __device__ __forceinline__ void ld(const bf16* p, float (&v)[4]) {
    const float4 t = ld4(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
  static __device__ __forceinline__ void st(bf16* p, const float (&v)[4]) { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
};

// keep(m, k .. k+VEC-1): pairs of the hash stream at even element indices (m*K + k is even: K and k are)
template <int VEC>
__device__ __forceinline__ void al_keep(uint32_t e, const DropCfg& dc, float (&s)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    const float2 t = dc.p8 > 0 ? drop_pair(e + j, dc) : make_float2(1.f, 1.f);
    s[j] = t.x;
    s[j + 1] = t.y;
  }
}

template <typename T, int G>
__global__ __launch_bounds__(kAlThreads) void attention_logits_fwd_kernel(const T* __restrict__ x, int ldx,
                                                                          const float* __restrict__ w,
                                                                          const float* __restrict__ bias,
                                                                          float* __restrict__ logits, int M, int K,
                                                                          DropCfg dc) {
  using V = AlVec<T>;
  constexpr int VEC = V::VEC, P = V::P, RB = 4;
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * kAlWaves + (threadIdx.x >> 6), nwaves = gridDim.x * kAlWaves;
  float wr[G][P][VEC];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int k = (p * 64 + lane) * VEC + j;
        wr[g][p][j] = k < K ? w[(size_t)g * K + k] : 0.f;
      }
  for (int m0 = wave * RB; m0 < M; m0 += nwaves * RB) {
    float xv[RB][P][VEC];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int k = (p * 64 + lane) * VEC;
        V::ld(x + (size_t)min(m0 + r, M - 1) * ldx + min(k, ldx - VEC), xv[r][p]);  // clamped, unconditional
      }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int m = m0 + r;
      float acc[G];
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = 0.f;
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int k = (p * 64 + lane) * VEC;
        float s[VEC];
        al_keep<VEC>((uint32_t)m * (uint32_t)K + (uint32_t)k, dc, s);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const float xs = xv[r][p][j] * s[j];  // columns >= K meet zero weights
#pragma unroll
          for (int g = 0; g < G; ++g) acc[g] = fmaf(wr[g][p][j], xs, acc[g]);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float t = wave_sum(acc[g]);
        if (lane == 0 && m < M) logits[(size_t)m * G + g] = t + bias[g];
      }
    }
  }
}

// Partial sums of one workgroup: part[blockIdx.x][g][k] (k < Kpad = 64 * VEC * P) and partb[blockIdx.x][g].
template <typename T, int G>
__global__ __launch_bounds__(kAlThreads) void attention_logits_bwd_kernel(const T* __restrict__ x, int ldx,
                                                                          const float* __restrict__ w,
                                                                          const float* __restrict__ d_logits,
                                                                          T* __restrict__ d_x, float* __restrict__ part,
                                                                          float* __restrict__ partb, int M, int K,
                                                                          DropCfg dc) {
  using V = AlVec<T>;
  constexpr int VEC = V::VEC, P = V::P, RB = 2, KPAD = 64 * VEC * P;
  __shared__ float red_s[kAlWaves - 1][G][KPAD];
  __shared__ float redb_s[kAlWaves][G];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * kAlWaves + wv, nwaves = gridDim.x * kAlWaves;
  float wr[G][P][VEC], dw[G][P][VEC], db[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    db[g] = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int k = (p * 64 + lane) * VEC + j;
        wr[g][p][j] = k < K ? w[(size_t)g * K + k] : 0.f;
        dw[g][p][j] = 0.f;
      }
  }
  for (int m0 = wave * RB; m0 < M; m0 += nwaves * RB) {
    float xv[RB][P][VEC];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int k = (p * 64 + lane) * VEC;
        V::ld(x + (size_t)min(m0 + r, M - 1) * ldx + min(k, ldx - VEC), xv[r][p]);
      }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int m = m0 + r;
      if (m < M) {  // wave-uniform
        float dl[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
          dl[g] = d_logits[(size_t)m * G + g];  // wave-uniform address: scalar load
          db[g] += dl[g];
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int k = (p * 64 + lane) * VEC;
          float s[VEC], o[VEC];
          al_keep<VEC>((uint32_t)m * (uint32_t)K + (uint32_t)k, dc, s);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float xs = xv[r][p][j] * s[j];
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
              dw[g][p][j] = fmaf(dl[g], xs, dw[g][p][j]);
              t = fmaf(dl[g], wr[g][p][j], t);
            }
            o[j] = t * s[j];
          }
          if (d_x != nullptr && k < ldx) V::st(d_x + (size_t)m * ldx + k, o);  // pad columns (k >= K): w = 0 -> 0
        }
      }
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int j = 0; j < VEC; ++j) red_s[wv - 1][g][(p * 64 + lane) * VEC + j] = dw[g][p][j];
  }
  if (lane == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g) redb_s[wv][g] = db[g];
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int p = 0; p < P; ++p)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int k = (p * 64 + lane) * VEC + j;
          float t = dw[g][p][j];
#pragma unroll
          for (int q = 0; q < kAlWaves - 1; ++q) t += red_s[q][g][k];
          part[((size_t)blockIdx.x * G + g) * KPAD + k] = t;
        }
    if (lane < G) partb[(size_t)blockIdx.x * G + lane] = redb_s[0][lane] + redb_s[1][lane] + redb_s[2][lane] + redb_s[3][lane];
  }
}

// d_w[g][k] = sum_blocks part[block][g][k] ; d_bias[g] = sum_blocks partb[block][g]   (fixed order).
// 256 lanes = 64 outputs x 4 block slices, 4 partials in flight per lane (a serial loop over 256 partials per output
// is a chain of dependent L2 round trips: 77 us); output G*K + g is d_bias[g].
__global__ __launch_bounds__(256) void attention_logits_finish_kernel(const float* __restrict__ part,
                                                                      const float* __restrict__ partb,
                                                                      float* __restrict__ d_w, float* __restrict__ d_bias,
                                                                      int K, int G, int KPAD, int nblocks) {
  __shared__ float red_s[3][64];
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + c;
  const bool is_w = e < G * K, is_b = !is_w && e < G * K + G;
  const float* src = partb;
  size_t stride = (size_t)G, off = 0;
  if (is_w) {
    src = part;
    stride = (size_t)G * KPAD;
    off = (size_t)(e / K) * KPAD + (e % K);
  } else if (is_b) {
    off = (size_t)(e - G * K);
  }
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (is_w || is_b) {
    int b = slice;
    for (; b + 12 < nblocks; b += 16) {
      a0 += src[(size_t)b * stride + off];
      a1 += src[(size_t)(b + 4) * stride + off];
      a2 += src[(size_t)(b + 8) * stride + off];
      a3 += src[(size_t)(b + 12) * stride + off];
    }
    for (; b < nblocks; b += 4) a0 += src[(size_t)b * stride + off];
  }
  const float a = (a0 + a1) + (a2 + a3);
  if (slice > 0) red_s[slice - 1][c] = a;
  __syncthreads();
  if (slice == 0) {
    const float t = a + red_s[0][c] + red_s[1][c] + red_s[2][c];
    if (is_w) d_w[e] = t;
    if (is_b) d_bias[e - G * K] = t;
  }
}

static int al_blocks(int M) {
  int b = (M + 4 * kAlWaves * 4 - 1) / (4 * kAlWaves * 4);  // >= 4 row batches of 4 rows per wave
  if (b > 256) b = 256;                                      // one workgroup per CU: 256 weight-gradient partials
  if (b < 1) b = 1;
  return b;
}

template <typename T>
static int al_check(const char* who, const T* x, int ldx, int M, int K, int G, float p_drop) {
  using V = AlVec<T>;
  VQA_REQUIRE(M > 0 && K > 0 && G > 0, VQA_E_BADARG, "%s: bad sizes M=%d K=%d G=%d", who, M, K, G);
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "%s: p_drop=%f outside [0,1)", who, (double)p_drop);
  VQA_REQUIRE(G <= kAlMaxG && K <= 64 * V::VEC * V::P && K % 2 == 0 && ldx >= K && ldx % V::VEC == 0 &&
                  aligned(x, V::VEC * sizeof(T)) && (long)M * K < (1L << 32),
              VQA_E_UNSUPPORTED, "%s: needs G <= 8, even K <= %d, ldx %% %d == 0, aligned x, M*K < 2^32 (K=%d ldx=%d G=%d)", who,
              64 * V::VEC * V::P, V::VEC, K, ldx, G);
  return VQA_OK;
}

#define VQA_AL_SWITCH_G(G, CALL) \
  switch (G) {                   \
    case 1: CALL(1); break;      \
    case 2: CALL(2); break;      \
    case 3: CALL(3); break;      \
    case 4: CALL(4); break;      \
    case 5: CALL(5); break;      \
    case 6: CALL(6); break;      \
    case 7: CALL(7); break;      \
    default: CALL(8); break;     \
  }

template <typename T>
static int al_fwd_impl(const char* who, const T* x, int ldx, const float* w, const float* bias, float* logits, float p_drop,
                       uint64_t seed, const uint64_t* seed_ptr, int M, int K, int G, vqa_stream_t stream) {
  VQA_REQUIRE(x && w && bias && logits, VQA_E_BADARG, "%s: null pointer", who);
  int rc = al_check(who, x, ldx, M, K, G, p_drop);
  if (rc != VQA_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  int blocks = (M + 4 * kAlWaves - 1) / (4 * kAlWaves);
  if (blocks > 1024) blocks = 1024;
#define CALL(G_) \
  hipLaunchKernelGGL((attention_logits_fwd_kernel<T, G_>), dim3(blocks), dim3(kAlThreads), 0, s, x, ldx, w, bias, logits, M, K, dc)
  VQA_AL_SWITCH_G(G, CALL);
#undef CALL
  return check_launch(who);
}

template <typename T>
static size_t al_ws_bytes(int M, int G) {
  using V = AlVec<T>;
  return (size_t)al_blocks(M) * G * (64 * V::VEC * V::P + 1) * sizeof(float);
}

template <typename T>
static int al_bwd_impl(const char* who, const T* x, int ldx, const float* w, const float* d_logits, T* d_x, float* d_w,
                       float* d_bias, void* workspace, size_t workspace_bytes, float p_drop, uint64_t seed,
                       const uint64_t* seed_ptr, int M, int K, int G, vqa_stream_t stream) {
  using V = AlVec<T>;
  VQA_REQUIRE(x && w && d_logits && d_w && d_bias && workspace, VQA_E_BADARG, "%s: null pointer", who);
  int rc = al_check(who, x, ldx, M, K, G, p_drop);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(d_x == nullptr || aligned(d_x, V::VEC * sizeof(T)), VQA_E_UNSUPPORTED, "%s: d_x is not aligned", who);
  VQA_REQUIRE(workspace_bytes >= al_ws_bytes<T>(M, G), VQA_E_BADARG, "%s: workspace of %zu B is too small", who, workspace_bytes);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  const int blocks = al_blocks(M);
  constexpr int KPAD = 64 * V::VEC * V::P;
  float* part = static_cast<float*>(workspace);
  float* partb = part + (size_t)blocks * G * KPAD;
#define CALL(G_)                                                                                                          \
  hipLaunchKernelGGL((attention_logits_bwd_kernel<T, G_>), dim3(blocks), dim3(kAlThreads), 0, s, x, ldx, w, d_logits, d_x, \
                     part, partb, M, K, dc)
  VQA_AL_SWITCH_G(G, CALL);
#undef CALL
  hipLaunchKernelGGL(attention_logits_finish_kernel, dim3((G * K + G + 63) / 64), dim3(256), 0, s, part, partb, d_w, d_bias, K,
                     G, KPAD, blocks);
  return check_launch(who);
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_attention_logits_fwd(const float* x, int ldx, const float* w, const float* bias, float* logits,
                                        float p_drop, uint64_t seed, const uint64_t* seed_ptr, int M, int K, int G,
                                        vqa_stream_t stream) {
  return al_fwd_impl<float>("attention_logits_fwd", x, ldx, w, bias, logits, p_drop, seed, seed_ptr, M, K, G, stream);
}
extern "C" int vqa_attention_logits_fwd_bf16(const vqa_bf16_t* x, int ldx, const float* w, const float* bias, float* logits,
                                             float p_drop, uint64_t seed, const uint64_t* seed_ptr, int M, int K, int G,
                                             vqa_stream_t stream) {
  return al_fwd_impl<bf16>("attention_logits_fwd_bf16", reinterpret_cast<const bf16*>(x), ldx, w, bias, logits, p_drop, seed,
                           seed_ptr, M, K, G, stream);
}
extern "C" size_t vqa_attention_logits_bwd_workspace_bytes(int M, int K, int G) {
  if (M <= 0 || K <= 0 || G <= 0 || G > kAlMaxG) return 0;
  return al_ws_bytes<float>(M, G) > al_ws_bytes<bf16>(M, G) ? al_ws_bytes<float>(M, G) : al_ws_bytes<bf16>(M, G);
}
extern "C" int vqa_attention_logits_bwd(const float* x, int ldx, const float* w, const float* d_logits, float* d_x,
                                        float* d_w, float* d_bias, void* workspace, size_t workspace_bytes, float p_drop,
                                        uint64_t seed, const uint64_t* seed_ptr, int M, int K, int G, vqa_stream_t stream) {
  return al_bwd_impl<float>("attention_logits_bwd", x, ldx, w, d_logits, d_x, d_w, d_bias, workspace, workspace_bytes, p_drop,
                            seed, seed_ptr, M, K, G, stream);
}
extern "C" int vqa_attention_logits_bwd_bf16(const vqa_bf16_t* x, int ldx, const float* w, const float* d_logits,
                                             vqa_bf16_t* d_x, float* d_w, float* d_bias, void* workspace,
                                             size_t workspace_bytes, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                             int M, int K, int G, vqa_stream_t stream) {
  return al_bwd_impl<bf16>("attention_logits_bwd_bf16", reinterpret_cast<const bf16*>(x), ldx, w, d_logits,
                           reinterpret_cast<bf16*>(d_x), d_w, d_bias, workspace, workspace_bytes, p_drop, seed, seed_ptr, M, K,
                           G, stream);
}
