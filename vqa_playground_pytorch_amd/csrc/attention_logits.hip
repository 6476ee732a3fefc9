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
#include <cstdlib>
#include <type_traits>

#include "gemm_f32_rt.hpp"

namespace vqa {

constexpr int kAlMaxG = 8;
constexpr int kAlThreads = 256;
constexpr int kAlWaves = kAlThreads / 64;

// A lane owns VEC adjacent columns in each of P passes: k = (pass * 64 + lane) * VEC.  K <= 64 * VEC * P.
template <typename T>
struct AlVec;
// (ldb / stb: the same 8-byte access as a buffer load / store -- a wave-uniform descriptor + a 32-bit per-lane column
// offset + a scalar row offset.  With 64-bit per-lane addresses a batch of 6 - 9 rows x 4 passes kept 70+ address
// registers alive and the kernels ran at one or two waves per SIMD.)
template <>
struct AlVec<float> {
  static constexpr int VEC = 2, P = 4;
  static __device__ __forceinline__ void ldb(rt::rsrc_t r, uint32_t voff, uint32_t soff, float (&v)[2]) {
    const auto t = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
    v[0] = t.x;
    v[1] = t.y;
  }
  static __device__ __forceinline__ void stb(rt::rsrc_t r, uint32_t voff, uint32_t soff, const float (&v)[2]) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{__float_as_uint(v[0]), __float_as_uint(v[1])}, r, (int)voff, (int)soff, 0);
  }
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
  static __device__ __forceinline__ void ldb(rt::rsrc_t r, uint32_t voff, uint32_t soff, float (&v)[4]) {
    const auto w = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0));
    v[0] = bf16_lo(w.x);
    v[1] = bf16_hi(w.x);
    v[2] = bf16_lo(w.y);
    v[3] = bf16_hi(w.y);
  }
  static __device__ __forceinline__ void stb(rt::rsrc_t r, uint32_t voff, uint32_t soff, const float (&v)[4]) {
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(u32x2{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3])}, r, (int)voff, (int)soff, 0);
  }
  static __device__ __forceinline__ void ld(const bf16* p, float (&v)[4]) {
    const float4 t = ld4(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
  static __device__ __forceinline__ void st(bf16* p, const float (&v)[4]) { st4(p, make_float4(v[0], v[1], v[2], v[3])); }
};

// keep(m, k .. k+VEC-1): pairs of the hash stream at even element indices (m*K + k is even: K and k are).  MODE (0 none,
// 1 the one-bit-per-element p = 0.5 form, 2 a byte per element) and the key are fixed ONCE per kernel: read through the
// DropCfg at every call, the mask cost a scalar load of the seed word + a wait + two branches per pair of elements.
struct AlMask {
  uint32_t key, p8;
  float scale;
};
__device__ __forceinline__ AlMask al_mask(const DropCfg& dc) { return AlMask{dc.p8 > 0 ? drop_key(dc) : 0u, dc.p8, dc.scale}; }
template <int MODE, int VEC>
__device__ __forceinline__ void al_keep(uint32_t e, const AlMask& mk, float (&s)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 2) {
    if constexpr (MODE == 0) {
      s[j] = 1.f, s[j + 1] = 1.f;
    } else if constexpr (MODE == 1) {
      const uint32_t w = mask_word32((e + j) >> 5, mk.key) >> ((e + j) & 31u);
      s[j] = (w & 1u) != 0u ? 2.f : 0.f;
      s[j + 1] = (w & 2u) != 0u ? 2.f : 0.f;
    } else {
      const uint32_t w = mask_word32((e + j) >> 2, mk.key) >> (8 * ((e + j) & 3));
      s[j] = (w & 255u) >= mk.p8 ? mk.scale : 0.f;
      s[j + 1] = ((w >> 8) & 255u) >= mk.p8 ? mk.scale : 0.f;
    }
  }
}
// the lane's slice of the G x K weights, branch-free (columns >= K read a clamped address and become 0)
template <int G, int P, int VEC>
__device__ __forceinline__ void al_load_weights(const float* __restrict__ w, int K, int lane, float (&wr)[G][P][VEC]) {
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int k = (p * 64 + lane) * VEC + j;
        const float t = w[(size_t)g * K + min(k, K - 1)];
        wr[g][p][j] = k < K ? t : 0.f;
      }
}

// Rows per wave and batch: with kAlRows * G a multiple of 16 the G sums of the rows are reduced by butterflies of 16 values
// over the 16-lane DPP rows + one lane-wise sum over the four rows (common.hpp) instead of kAlRows * G full wave reductions,
// and the grid is sized so that every wave takes exactly ONE batch (a second, mostly empty batch per wave was a second
// full memory latency for the whole kernel).
constexpr int kAlRows = 8;

template <typename T, int G, int MODE>
__global__ __launch_bounds__(kAlThreads, 2) void attention_logits_fwd_kernel(const T* __restrict__ x, int ldx,
                                                                          const float* __restrict__ w,
                                                                          const float* __restrict__ bias,
                                                                          float* __restrict__ logits, int M, int K,
                                                                          DropCfg dc) {
  using V = AlVec<T>;
  constexpr int VEC = V::VEC, P = V::P, RB = kAlRows;
  constexpr bool BUTTERFLY = (RB * G) % 16 == 0 && 16 % G == 0;
  const int lane = threadIdx.x & 63;
  // (readfirstlane: the wave index is uniform, but hipcc only knows that when told -- without it every row load's scalar offset
  //  was "divergent" and each of the 64 buffer loads ran inside a waterfall loop; tools/waterfall_check.py)
  const int wave = blockIdx.x * kAlWaves + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * kAlWaves;
  // the row loads go out first: the weights (a few KB, L2-resident after the first workgroups) land under them
  float xv[RB][P][VEC];
  const int m_first = wave * RB;
  const rt::rsrc_t Xb = rt::make_rsrc(x, (size_t)M * ldx * sizeof(T));
  uint32_t voff[P];
#pragma unroll
  for (int p = 0; p < P; ++p) voff[p] = (uint32_t)min((p * 64 + lane) * VEC, ldx - VEC) * (uint32_t)sizeof(T);   // clamped columns
  auto load_rows = [&](int m0) {            // rows clamped too: the loads are unconditional
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const uint32_t soff = (uint32_t)min(m0 + r, M - 1) * (uint32_t)ldx * (uint32_t)sizeof(T);
#pragma unroll
      for (int p = 0; p < P; ++p) V::ldb(Xb, voff[p], soff, xv[r][p]);
    }
  };
  if (m_first < M) load_rows(m_first);
  float wr[G][P][VEC];
  al_load_weights<G, P, VEC>(w, K, lane, wr);
  const float bias_l = bias[(lane & 15) % G];   // (asked for here: loaded where it is added, the store waited a round trip for it)
  const AlMask mk = al_mask(dc);
  for (int m0 = m_first; m0 < M; m0 += nwaves * RB) {
    if (m0 != m_first) load_rows(m0);
    float acc[RB][G];
    {
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int m = min(m0 + r, M - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[r][g] = 0.f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int k = (p * 64 + lane) * VEC;
          float s[VEC];
          al_keep<MODE, VEC>((uint32_t)m * (uint32_t)K + (uint32_t)k, mk, s);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float xs = xv[r][p][j] * s[j];  // columns >= K meet zero weights
#pragma unroll
            for (int g = 0; g < G; ++g) acc[r][g] = fmaf(wr[g][p][j], xs, acc[r][g]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);     // (see the backward kernel: keeps the register count at 2-3 waves per SIMD)
      }
    }
    if constexpr (BUTTERFLY) {
      constexpr int RPB = 16 / G;               // rows per butterfly
#pragma unroll
      for (int q = 0; q < RB / RPB; ++q) {
        float val[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) val[i] = acc[q * RPB + i / G][i % G];
        const float t = rows_sum(row_reduce_scatter16(val, lane));   // lane l: value l & 15 = (row (l & 15) / G, glimpse % G)
        const int m = m0 + q * RPB + (lane & 15) / G;
        if (lane < 16 && m < M) logits[(size_t)m * G + (lane & 15) % G] = t + bias_l;
      }
    } else {
#pragma unroll
      for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float t = wave_sum(acc[r][g]);
          if (lane == 0 && m0 + r < M) logits[(size_t)(m0 + r) * G + g] = t + bias[g];
        }
    }
  }
}

// Rows per wave and batch of the backward kernel: 6 rows x 32 B per lane in flight (2 rows per batch on 256 workgroups left
// 16 KB in flight per CU and nine dependent memory latencies: 31.7 us; 9 rows need more registers than two waves per SIMD have).
constexpr int kAlBwdRows = 6;

// Partial sums of one workgroup: part[blockIdx.x][g][k] (k < Kpad = 64 * VEC * P) and partb[blockIdx.x][g].
template <typename T, int G, int MODE>
__global__ __launch_bounds__(kAlThreads, 2) void attention_logits_bwd_kernel(const T* __restrict__ x, int ldx,
                                                                          const float* __restrict__ w,
                                                                          const float* __restrict__ d_logits,
                                                                          T* __restrict__ d_x, float* __restrict__ part,
                                                                          float* __restrict__ partb, int M, int K,
                                                                          DropCfg dc) {
  using V = AlVec<T>;
  constexpr int VEC = V::VEC, P = V::P, RB = kAlBwdRows, KPAD = 64 * VEC * P;
  __shared__ float red_s[kAlWaves - 1][G][KPAD];
  __shared__ float redb_s[kAlWaves][G];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: see the forward kernel)
  const int wave = blockIdx.x * kAlWaves + wv, nwaves = gridDim.x * kAlWaves;
  float wr[G][P][VEC], dw[G][P][VEC], db[G];
  al_load_weights<G, P, VEC>(w, K, lane, wr);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    db[g] = 0.f;
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
      for (int j = 0; j < VEC; ++j) dw[g][p][j] = 0.f;
  }
  const AlMask mk = al_mask(dc);
  const rt::rsrc_t Xb = rt::make_rsrc(x, (size_t)M * ldx * sizeof(T));
  const rt::rsrc_t Db = rt::make_rsrc(d_x != nullptr ? d_x : const_cast<T*>(x), (size_t)M * ldx * sizeof(T));
  uint32_t voff[P];
#pragma unroll
  for (int p = 0; p < P; ++p) voff[p] = (uint32_t)min((p * 64 + lane) * VEC, ldx - VEC) * (uint32_t)sizeof(T);   // clamped columns
  for (int m0 = wave * RB; m0 < M; m0 += nwaves * RB) {
    float xv[RB][P][VEC];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const uint32_t soff = (uint32_t)min(m0 + r, M - 1) * (uint32_t)ldx * (uint32_t)sizeof(T);
#pragma unroll
      for (int p = 0; p < P; ++p) V::ldb(Xb, voff[p], soff, xv[r][p]);
    }
    // the rows' d_logits (wave-uniform addresses: scalar loads), all requested before the first row is worked on; rows
    // beyond M read a clamped address and count as zero
    float dl[RB][G];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float t = d_logits[(size_t)min(m0 + r, M - 1) * G + g];
        dl[r][g] = m0 + r < M ? t : 0.f;
      }
    {
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int m = m0 + r;
#pragma unroll
        for (int g = 0; g < G; ++g) db[g] += dl[r][g];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const int k = (p * 64 + lane) * VEC;
          float s[VEC], o[VEC];
          al_keep<MODE, VEC>((uint32_t)min(m, M - 1) * (uint32_t)K + (uint32_t)k, mk, s);
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float xs = xv[r][p][j] * s[j];
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
              dw[g][p][j] = fmaf(dl[r][g], xs, dw[g][p][j]);
              t = fmaf(dl[r][g], wr[g][p][j], t);
            }
            o[j] = t * s[j];
          }
          if (d_x != nullptr && m < M && k < ldx)                                  // pad columns (k >= K): w = 0 -> 0
            V::stb(Db, (uint32_t)k * (uint32_t)sizeof(T), (uint32_t)m * (uint32_t)ldx * (uint32_t)sizeof(T), o);
        }
        // rows stay apart in the schedule: hoisting the hash words of all rows above the first row's FMAs (to cover the load
        // latency that the other resident waves cover anyway) cost 400 registers, i.e. one wave per SIMD
        __builtin_amdgcn_sched_barrier(0);
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
// 256 lanes = 16 outputs x 16 block slices, 16 partials in flight per lane (a serial loop over the partials of an output
// is a chain of dependent L2 round trips: 77 us; 4 slices x 4 in flight on 33 workgroups: 7 us); output G*K + g is d_bias[g].
__global__ __launch_bounds__(256) void attention_logits_finish_kernel(const float* __restrict__ part,
                                                                      const float* __restrict__ partb,
                                                                      float* __restrict__ d_w, float* __restrict__ d_bias,
                                                                      int K, int G, int KPAD, int nblocks) {
  __shared__ float red_s[16][17];
  const int c = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + c;
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
  float a = 0.f;
  if (is_w || is_b) {
    for (int b0 = slice; b0 < nblocks; b0 += 256) {       // 16 slices x 16 partials per round
      float t[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int b = b0 + 16 * i;
        t[i] = b < nblocks ? src[(size_t)b * stride + off] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) a += t[i];
    }
  }
  red_s[slice][c] = a;
  __syncthreads();
  if (slice == 0) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red_s[q][c];
    if (is_w) d_w[e] = t;
    if (is_b) d_bias[e - G * K] = t;
  }
}

static int al_blocks(int M) {
  int b = (M + kAlBwdRows * kAlWaves - 1) / (kAlBwdRows * kAlWaves);  // one batch of rows per wave ...
  if (b > 512) b = 512;   // ... up to 2 workgroups per CU (the kernel needs ~240 registers): 512 weight-gradient partials.
                          // (measured at M = 18432: 256 / 384 / 512 / 768 workgroups = 23.5 / 20.4 / 19.6 / 21.1 us)
  if (b < 1) b = 1;
  return b;
}

template <typename T>
static int al_check(const char* who, const T* x, int ldx, int M, int K, int G, float p_drop) {
  using V = AlVec<T>;
  VQA_REQUIRE(M > 0 && K > 0 && G > 0, VQA_E_BADARG, "%s: bad sizes M=%d K=%d G=%d", who, M, K, G);
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "%s: p_drop=%f outside [0,1)", who, (double)p_drop);
  VQA_REQUIRE(G <= kAlMaxG && K <= 64 * V::VEC * V::P && K % 2 == 0 && ldx >= K && ldx % V::VEC == 0 &&
                  aligned(x, V::VEC * sizeof(T)) && (size_t)M * ldx * sizeof(T) < (1ull << 32),
              VQA_E_UNSUPPORTED, "%s: needs G <= 8, even K <= %d, ldx %% %d == 0, aligned x, M*ldx*sizeof < 4 GiB (K=%d ldx=%d G=%d)", who,
              64 * V::VEC * V::P, V::VEC, K, ldx, G);
  return VQA_OK;
}

#define VQA_AL_SWITCH_G(G, CALL)      \
  switch (G) {                        \
    case 1: { CALL(1); } break;       \
    case 2: { CALL(2); } break;       \
    case 3: { CALL(3); } break;       \
    case 4: { CALL(4); } break;       \
    case 5: { CALL(5); } break;       \
    case 6: { CALL(6); } break;       \
    case 7: { CALL(7); } break;       \
    default: { CALL(8); } break;      \
  }

template <typename T>
static int al_fwd_impl(const char* who, const T* x, int ldx, const float* w, const float* bias, float* logits, float p_drop,
                       uint64_t seed, const uint64_t* seed_ptr, int M, int K, int G, vqa_stream_t stream) {
  VQA_REQUIRE(x && w && bias && logits, VQA_E_BADARG, "%s: null pointer", who);
  int rc = al_check(who, x, ldx, M, K, G, p_drop);
  if (rc != VQA_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  int blocks = (M + kAlRows * kAlWaves - 1) / (kAlRows * kAlWaves);   // one batch of rows per wave
  if (blocks > 2048) blocks = 2048;
  const int mode = dc.p8 == 0 ? 0 : (dc.p8 == kDropHalf ? 1 : 2);
#define CALL_M(G_, MODE_) \
  VQA_LAUNCH((attention_logits_fwd_kernel<T, G_, MODE_>), dim3(blocks), dim3(kAlThreads), 0, s, x, ldx, w, bias, logits, M, K, dc)
#define CALL(G_)                       \
  if (mode == 0) CALL_M(G_, 0);        \
  else if (mode == 1) CALL_M(G_, 1);   \
  else CALL_M(G_, 2)
  VQA_AL_SWITCH_G(G, CALL);
#undef CALL
#undef CALL_M
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
  const int mode = dc.p8 == 0 ? 0 : (dc.p8 == kDropHalf ? 1 : 2);
#define CALL_M(G_, MODE_)                                                                                                        \
  VQA_LAUNCH((attention_logits_bwd_kernel<T, G_, MODE_>), dim3(blocks), dim3(kAlThreads), 0, s, x, ldx, w, d_logits, d_x, \
                     part, partb, M, K, dc)
#define CALL(G_)                       \
  if (mode == 0) CALL_M(G_, 0);        \
  else if (mode == 1) CALL_M(G_, 1);   \
  else CALL_M(G_, 2)
  VQA_AL_SWITCH_G(G, CALL);
#undef CALL
#undef CALL_M
  VQA_LAUNCH(attention_logits_finish_kernel, dim3((G * K + G + 15) / 16), dim3(256), 0, s, part, partb, d_w, d_bias, K,
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
