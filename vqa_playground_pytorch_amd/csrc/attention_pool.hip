// K3 -- softmax over regions + attention-weighted region pooling.
//
// Replaces F.softmax(dim=1) of the attention conv (config/CoR2.py:83-87, :132) and
// putils.bmatmul(alpha^T, v) (config/CoR2.py:142; putils/__init__.py:89-95: a python loop of B
// [G,N]x[N,D] matmuls + torch.stack).  One pass over v:
//
//   forward : a workgroup owns (sample, 4*NT-wide feature chunk); every lane keeps G float4
//             accumulators and streams the N region rows with coalesced 16-byte loads; the N*G
//             softmax is recomputed per workgroup from the N*G logits (wave64 shuffles).
//   backward: same (sample, feature chunk) ownership as forward; every lane keeps its float4 column of the G rows
//             of d_pooled in registers, streams the N region rows once, writes d_v_n in the same sweep and
//             reduces the G dot products <d_pooled_g, v_n> with wave64 shuffles; a second tiny kernel closes the
//             softmax backward on [B,N,G].
//
// HBM-bound.  Algorithmic bytes per sample (fp32): forward (N*D + 2*N*G + G*D)*4 = 328 832 B at
// N=36, D=2048, G=4;  backward (N*D + G*D + 3*N*G)*4 (+ N*D*4 when d_v is written).
//
// The stream kernels are templates on the storage type T of the region tensors v / d_v (float, or bf16 for the
// mixed-precision path: half the bytes of the dominant stream); logits, alpha, pooled and their gradients are fp32.
#include <cstdlib>

#include "common.hpp"

namespace vqa {

constexpr int kMaxG = 8;

// softmax over n for every glimpse; logits_b -> alpha_s (LDS, [N][G]).  Called by all NT threads.
template <int NT>
__device__ __forceinline__ void block_softmax_regions(const float* __restrict__ logits_b, float* alpha_s, float* stat_s,
                                                      int N, int G) {
  const int tid = threadIdx.x;
  const int NG = N * G;
  for (int t = tid; t < NG; t += NT) alpha_s[t] = logits_b[t];
  __syncthreads();
  // one wave per glimpse (round-robin): max and sum over n with wave64 reductions
  const int wave = tid >> 6, lane = tid & 63;
  for (int g = wave; g < G; g += NT / 64) {
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, alpha_s[n * G + g]);
    m = wave_max(m);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += expf(alpha_s[n * G + g] - m);
    s = wave_sum(s);
    if (lane == 0) {
      stat_s[g] = m;
      stat_s[kMaxG + g] = 1.f / s;
    }
  }
  __syncthreads();
  for (int t = tid; t < NG; t += NT) {
    const int g = t % G;
    alpha_s[t] = expf(alpha_s[t] - stat_s[g]) * stat_s[kMaxG + g];
  }
  __syncthreads();
}

template <typename T, int NT, int G>
__global__ __launch_bounds__(NT) void attention_pool_fwd_kernel(const float* __restrict__ logits,
                                                                const T* __restrict__ v, float* __restrict__ alpha,
                                                                float* __restrict__ pooled, float* __restrict__ first,
                                                                int N, int D, DropCfg dc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);  // [N][G]
  float* stat_s = alpha_s + N * G;                  // [2*kMaxG]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + tid) * 4;
  const bool active = d < D;
  const T* vb = v + (size_t)b * N * D + d;

  // start the first rows of the stream before the softmax prologue so HBM latency overlaps it
  constexpr int PF = 4;
  float4 pf[PF];
#pragma unroll
  for (int i = 0; i < PF; ++i) pf[i] = (active && i < N) ? ld4(vb + (size_t)i * D) : make_float4(0.f, 0.f, 0.f, 0.f);

  block_softmax_regions<NT>(logits + (size_t)b * N * G, alpha_s, stat_s, N, G);
  if (blockIdx.x == 0)
    for (int t = tid; t < N * G; t += NT) alpha[(size_t)b * N * G + t] = alpha_s[t];
  if (!active) return;

  float4 acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    if (i < N) {
#pragma unroll
      for (int g = 0; g < G; ++g) acc[g] = fma4(alpha_s[i * G + g], pf[i], acc[g]);
    }
  }
#pragma unroll 8
  for (int n = PF; n < N; ++n) {
    const float4 x = ld4(vb + (size_t)n * D);
#pragma unroll
    for (int g = 0; g < G; ++g) acc[g] = fma4(alpha_s[n * G + g], x, acc[g]);
  }
  // `first`: glimpse 0 as pooled (CoR2's relation step reads it undropped); `pooled` carries the glimpse projections'
  // input dropout when dc asks for one (element index (b G + g) D + d of the [B,G,D] tensor)
  if (first != nullptr) st4(first + (size_t)b * D + d, acc[0]);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const size_t e = ((size_t)b * G + g) * D + d;
    float4 o = acc[g];
    if (dc.p8 > 0) {
      const float4 k = drop_quad((uint32_t)e, dc);
      o = make_float4(o.x * k.x, o.y * k.y, o.z * k.z, o.w * k.w);
    }
    st4(pooled + e, o);
  }
}

// Backward, streaming form.  Kernel A: grid (D/1024, B), lane = one float4 column with the G rows of d_pooled in
// registers; it streams the N region rows once, writes d_v in the same sweep and reduces the G dot products
// <d_pooled_g, v_n> per row with wave64 shuffles -> LDS -> one float atomic per (workgroup, n, g) into a zeroed
// accumulator.  Kernel B closes the softmax backward on the tiny [B,N,G] tensors.
// RS (row split): the NT/64 waves of a workgroup either sit side by side over NT*4 columns (RS = 1) or share NT*4/RS
// columns and take every RS-th region row each (RS = NT/64).  A small batch (B*D/4 lanes, one 8..16-byte load per row
// each) cannot keep enough bytes in flight to cover HBM latency; RS = 4 quadruples the lanes.
template <typename T, int NT, int G, int RS>
__global__ __launch_bounds__(NT) void attention_pool_bwd_stream_kernel(const float* __restrict__ alpha,
                                                                       const T* __restrict__ v,
                                                                       const float* __restrict__ d_pooled,
                                                                       const float* __restrict__ d_first,
                                                                       float* __restrict__ dal_acc, T* __restrict__ d_v,
                                                                       int N, int D, DropCfg dc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);  // [N][G]
  float* red_s = alpha_s + N * G;                   // [N][G]
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.y;
  constexpr int COLS = NT / RS;  // float4 columns per workgroup
  const int rs = tid / COLS;     // which rows: n = rs, rs + RS, ...
  const int d = (blockIdx.x * COLS + tid % COLS) * 4;
  const bool active = d < D;
  const int dcol = active ? d : 0;
  const int NG = N * G;
  for (int t = tid; t < NG; t += NT) {
    alpha_s[t] = alpha[(size_t)b * NG + t];
    red_s[t] = 0.f;
  }
  float4 p[G];
#pragma unroll
  for (int gI = 0; gI < G; ++gI) {
    const size_t e = ((size_t)b * G + gI) * D + dcol;
    float4 t = ld4(d_pooled + e);
    if (dc.p8 > 0) {        // the forward's dropout of `pooled`: its gradient passes the same mask
      const float4 k = drop_quad((uint32_t)e, dc);
      t = make_float4(t.x * k.x, t.y * k.y, t.z * k.z, t.w * k.w);
    }
    if (gI == 0 && d_first != nullptr) {   // the gradient that arrived on the undropped glimpse 0
      const float4 f = ld4(d_first + (size_t)b * D + dcol);
      t = make_float4(t.x + f.x, t.y + f.y, t.z + f.z, t.w + f.w);
    }
    p[gI] = active ? t : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __syncthreads();
  const T* vb = v + (size_t)b * N * D + dcol;
  T* dvb = d_v ? d_v + (size_t)b * N * D + dcol : nullptr;
  if constexpr (G == 1 || G == 2 || G == 4 || G == 8) {
    // RB rows per batch with RB G = 16 dot products per lane: RB independent 16-byte loads in flight, then ONE butterfly
    // reduce-scatter of the 16 values over the 16 lanes of a DPP row (each stage halves the values a lane carries: 8 + 4 +
    // 2 + 1 exchanges; lane l & 15 ends with value l & 15 summed over its row) and one LDS atomic per lane -- instead of
    // 16 full wave reductions and 16 single-lane atomics (the kernel was VALU-bound on them: 41 us against 32 forward).
    constexpr int RB = 16 / G;
    const int my_k = (lane & 15) / G, my_g = (lane & 15) % G;
    for (int n0 = rs; n0 < N; n0 += RB * RS) {
      float4 x[RB];
#pragma unroll
      for (int k = 0; k < RB; ++k) x[k] = ld4(vb + (size_t)min(n0 + k * RS, N - 1) * D);
      float val[16];
#pragma unroll
      for (int k = 0; k < RB; ++k) {
        const int n = n0 + k * RS;
        if (dvb != nullptr && active && n < N) {
          float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int gI = 0; gI < G; ++gI) o = fma4(alpha_s[n * G + gI], p[gI], o);
          st4(dvb + (size_t)n * D, o);
        }
#pragma unroll
        for (int gI = 0; gI < G; ++gI) val[k * G + gI] = dot4(x[k], p[gI]);     // (p = 0 in inactive lanes)
      }
      const float s1 = row_reduce_scatter16(val, lane);
      const int n = n0 + my_k * RS;
      if (n < N) atomicAdd(&red_s[n * G + my_g], s1);      // the four DPP rows of the wave meet here
    }
  } else {
  constexpr int RB = 4;  // rows per batch: RB independent 16-byte loads in flight per lane, then RB*G wave reductions
  for (int n0 = rs; n0 < N; n0 += RB * RS) {
    float4 x[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) x[k] = ld4(vb + (size_t)min(n0 + k * RS, N - 1) * D);
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int n = n0 + k * RS;
      if (n < N) {
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        float dots[G];
#pragma unroll
        for (int gI = 0; gI < G; ++gI) {
          dots[gI] = dot4(x[k], p[gI]);
          o = fma4(alpha_s[n * G + gI], p[gI], o);
        }
        if (dvb && active) st4(dvb + (size_t)n * D, o);
#pragma unroll
        for (int gI = 0; gI < G; ++gI) {
          const float r = wave_sum(dots[gI]);
          if (lane == 0) atomicAdd(&red_s[n * G + gI], r);
        }
      }
    }
  }
  }
  __syncthreads();
  for (int t = tid; t < NG; t += NT) atomicAdd(&dal_acc[(size_t)b * NG + t], red_s[t]);
}

// Backward in ONE kernel when a workgroup can own a whole sample (D <= 4 NT CG columns: D = 2048 at NT = 256, CG = 2):
// every lane keeps CG float4 columns of the G rows of d_pooled, the sample's N x G dot products are complete inside the
// workgroup (LDS), and the softmax backward d_logits = alpha (dal - sum_n alpha dal) closes in the same launch -- no
// zeroed accumulator, no atomics to global memory, no second kernel (three launches -> one: the two small ones were
// launch-latency floors of ~5 us each).
template <typename T, int NT, int G, int CG>
__global__ __launch_bounds__(NT) void attention_pool_bwd_fused_kernel(const float* __restrict__ alpha, const T* __restrict__ v,
                                                                      const float* __restrict__ d_pooled,
                                                                      const float* __restrict__ d_first,
                                                                      const float* __restrict__ d_alpha_ext,
                                                                      float* __restrict__ d_logits, T* __restrict__ d_v, int N,
                                                                      int D, DropCfg dc) {
  static_assert(G == 1 || G == 2 || G == 4 || G == 8, "the butterfly takes 16 / G rows per batch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);  // [N][G]
  float* red_s = alpha_s + N * G;                   // [N][G]
  float* inner_s = red_s + N * G;                   // [kMaxG]
  float* part_s = inner_s + kMaxG;                  // [NT / 64][N][G]: every wave's partial dot products, summed in fixed order
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const int NG = N * G;
  for (int t = tid; t < NG; t += NT) alpha_s[t] = alpha[(size_t)b * NG + t];
  float4 p[CG][G];
  int dcol[CG];
  bool active[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    const int d = (c * NT + tid) * 4;
    active[c] = d < D;
    dcol[c] = active[c] ? d : 0;
#pragma unroll
    for (int gI = 0; gI < G; ++gI) {
      const size_t e = ((size_t)b * G + gI) * D + dcol[c];
      float4 t = ld4(d_pooled + e);
      if (dc.p8 > 0) {
        const float4 k = drop_quad((uint32_t)e, dc);
        t = make_float4(t.x * k.x, t.y * k.y, t.z * k.z, t.w * k.w);
      }
      if (gI == 0 && d_first != nullptr) {
        const float4 f = ld4(d_first + (size_t)b * D + dcol[c]);
        t = make_float4(t.x + f.x, t.y + f.y, t.z + f.z, t.w + f.w);
      }
      p[c][gI] = active[c] ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  const T* vb = v + (size_t)b * N * D;
  T* dvb = d_v ? d_v + (size_t)b * N * D : nullptr;
  constexpr int RB = 16 / G;
  const int my_k = (lane & 15) / G, my_g = (lane & 15) % G;
  for (int n0 = 0; n0 < N; n0 += RB) {
    float4 x[CG][RB];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int k = 0; k < RB; ++k) x[c][k] = ld4(vb + (size_t)min(n0 + k, N - 1) * D + dcol[c]);
    float val[16];
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int n = n0 + k;
#pragma unroll
      for (int gI = 0; gI < G; ++gI) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < CG; ++c) t += dot4(x[c][k], p[c][gI]);     // (p = 0 in inactive columns)
        val[k * G + gI] = t;
      }
      if (dvb != nullptr && n < N) {
#pragma unroll
        for (int c = 0; c < CG; ++c)
          if (active[c]) {
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int gI = 0; gI < G; ++gI) o = fma4(alpha_s[n * G + gI], p[c][gI], o);
            st4(dvb + (size_t)n * D + dcol[c], o);
          }
      }
    }
    // (round 6: the wave's value goes to its OWN slot -- rounds 1-5 added the 16 partial sums of a (region, glimpse) pair with LDS
    //  float atomics, in whatever order the waves arrived: d_logits, and every gradient upstream of it, differed in the last bit
    //  from run to run; tools/determinism_probe.py)
    const float s1 = rows_sum(row_reduce_scatter16(val, lane));
    const int n = n0 + my_k;
    if (lane < 16 && n < N) part_s[wave * NG + n * G + my_g] = s1;
  }
  __syncthreads();
  // softmax backward on the sample's [N][G]: dal = <d_pooled_g, v_n> (+ what arrived on alpha directly)
  for (int t = tid; t < NG; t += NT) {
    float sum = part_s[t];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) sum += part_s[w * NG + t];
    red_s[t] = sum + (d_alpha_ext != nullptr ? d_alpha_ext[(size_t)b * NG + t] : 0.f);
  }
  __syncthreads();
  for (int gI = wave; gI < G; gI += NT / 64) {
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) sum += alpha_s[n * G + gI] * red_s[n * G + gI];
    sum = wave_sum(sum);
    if (lane == 0) inner_s[gI] = sum;
  }
  __syncthreads();
  for (int t = tid; t < NG; t += NT) d_logits[(size_t)b * NG + t] = alpha_s[t] * (red_s[t] - inner_s[t % G]);
}

// d_logits = alpha * (dal - sum_n alpha*dal),  dal = accumulated <d_pooled, v> (+ the gradient arriving on alpha)
__global__ __launch_bounds__(256) void attention_softmax_bwd_kernel(const float* __restrict__ alpha,
                                                                    const float* __restrict__ dal_acc,
                                                                    const float* __restrict__ d_alpha_ext,
                                                                    float* __restrict__ d_logits, int N, int G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);
  float* dal_s = alpha_s + N * G;
  float* inner_s = dal_s + N * G;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, b = blockIdx.x, NG = N * G;
  for (int t = tid; t < NG; t += 256) {
    alpha_s[t] = alpha[(size_t)b * NG + t];
    dal_s[t] = dal_acc[(size_t)b * NG + t] + (d_alpha_ext ? d_alpha_ext[(size_t)b * NG + t] : 0.f);
  }
  __syncthreads();
  for (int gI = wave; gI < G; gI += 4) {
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += alpha_s[n * G + gI] * dal_s[n * G + gI];
    s = wave_sum(s);
    if (lane == 0) inner_s[gI] = s;
  }
  __syncthreads();
  for (int t = tid; t < NG; t += 256) d_logits[(size_t)b * NG + t] = alpha_s[t] * (dal_s[t] - inner_s[t % G]);
}

template <typename T, int G>
static int launch_fwd(const float* logits, const T* v, float* alpha, float* pooled, float* first, const DropCfg& dc, int B,
                      int N, int D, hipStream_t s) {
  constexpr int NT = 256;
  const size_t lds = ((size_t)N * G + 2 * kMaxG) * sizeof(float);
  dim3 grid((D / 4 + NT - 1) / NT, B);
  VQA_LAUNCH((attention_pool_fwd_kernel<T, NT, G>), grid, dim3(NT), lds, s, logits, v, alpha, pooled, first, N, D, dc);
  return check_launch("softmax_attention_pool_fwd");
}

template <typename T, int G>
static int launch_bwd(const float* alpha, const T* v, const float* d_pooled, const float* d_first, const float* d_alpha_ext,
                      float* d_logits, T* d_v, const DropCfg& dc, int B, int N, int D, hipStream_t s) {
  constexpr int NT = 256;
  if constexpr (G == 1 || G == 2 || G == 4 || G == 8) {
    // one workgroup per sample, everything in one launch -- when the batch alone fills the chip (>= 2 workgroups per CU)
    // (VQA_K3_FUSED_MIN_B: the smallest batch that takes this form; tests set 1, a huge value keeps the three-launch form)
    const char* env = vqa::option("VQA_K3_FUSED_MIN_B");
    // default: fp32 from 64 samples on (measured at B = 128, N = 36: 1.109 ms per step against 1.111 with the three-launch form --
    // and this form sums in a fixed order, the other one with float atomics); bf16 regions from 512 (N = 100, B = 128: +2 %)
    const int min_b = env != nullptr ? std::atoi(env) : (sizeof(T) == 4 ? 64 : 512);
    const size_t lds_f = ((size_t)(2 + NT / 64) * N * G + kMaxG) * sizeof(float);
    if (B >= min_b && D % 4 == 0 && D <= 4 * NT * 2 && D > 4 * NT) {
      VQA_LAUNCH((attention_pool_bwd_fused_kernel<T, NT, G, 2>), dim3(B), dim3(NT), lds_f, s, alpha, v, d_pooled, d_first,
                         d_alpha_ext, d_logits, d_v, N, D, dc);
      return check_launch("softmax_attention_pool_bwd");
    }
  }
  // d_logits doubles as the zeroed accumulator of the first kernel (same [B,N,G] shape; kernel B reads each of its
  // elements before overwriting it)
  int rc = zero_async(d_logits, (size_t)B * N * G * sizeof(float), s);
  if (rc != VQA_OK) return rc;
  const size_t lds = 2 * (size_t)N * G * sizeof(float);
  if ((long)B * D / 4 < 4 * 65536) {  // fewer than 4 waves per CU worth of lanes: split the rows over the waves
    constexpr int RS = NT / 64;
    VQA_LAUNCH((attention_pool_bwd_stream_kernel<T, NT, G, RS>), dim3((D / 4 + 63) / 64, B), dim3(NT), lds, s, alpha,
                       v, d_pooled, d_first, d_logits, d_v, N, D, dc);
  } else {
    VQA_LAUNCH((attention_pool_bwd_stream_kernel<T, NT, G, 1>), dim3((D / 4 + NT - 1) / NT, B), dim3(NT), lds, s, alpha,
                       v, d_pooled, d_first, d_logits, d_v, N, D, dc);
  }
  VQA_LAUNCH(attention_softmax_bwd_kernel, dim3(B), dim3(256), lds + kMaxG * sizeof(float), s, alpha, d_logits,
                     d_alpha_ext, d_logits, N, G);
  return check_launch("softmax_attention_pool_bwd");
}

template <typename T>
static int pool_fwd_impl(const char* who, const float* logits, const T* v, float* alpha, float* pooled, float* first,
                         float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D, int G, vqa_stream_t stream) {
  constexpr size_t kAlign = 4 * sizeof(T);
  VQA_REQUIRE(logits && v && alpha && pooled, VQA_E_BADARG, "%s: null pointer", who);
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && G > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d D=%d G=%d", who, B, N, D, G);
  VQA_REQUIRE(G <= kMaxG && N <= 1024, VQA_E_UNSUPPORTED, "%s: needs G <= 8 and N <= 1024 (G=%d N=%d)", who, G, N);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, kAlign) && aligned(pooled, 16), VQA_E_UNSUPPORTED,
              "%s: needs D %% 4 == 0, %zu-byte aligned v and 16-byte aligned pooled (D=%d)", who, kAlign, D);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "%s: B=%d exceeds 65535", who, B);
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "%s: p_drop=%f outside [0,1)", who, (double)p_drop);
  VQA_REQUIRE(p_drop == 0.f || (size_t)B * G * D < (1ull << 32), VQA_E_UNSUPPORTED, "%s: dropout needs B*G*D < 2^32", who);
  VQA_REQUIRE(first == nullptr || aligned(first, 16), VQA_E_UNSUPPORTED, "%s: first is not 16-byte aligned", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
#define CALL_FWD(G_) launch_fwd<T, G_>(logits, v, alpha, pooled, first, dc, B, N, D, s)
  switch (G) {
    case 1: return CALL_FWD(1);
    case 2: return CALL_FWD(2);
    case 3: return CALL_FWD(3);
    case 4: return CALL_FWD(4);
    case 5: return CALL_FWD(5);
    case 6: return CALL_FWD(6);
    case 7: return CALL_FWD(7);
    default: return CALL_FWD(8);
  }
#undef CALL_FWD
}

template <typename T>
static int pool_bwd_impl(const char* who, const float* alpha, const T* v, const float* d_pooled, const float* d_first,
                         const float* d_alpha_ext, float* d_logits, T* d_v, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                         int B, int N, int D, int G, vqa_stream_t stream) {
  constexpr size_t kAlign = 4 * sizeof(T);
  VQA_REQUIRE(alpha && v && d_pooled && d_logits, VQA_E_BADARG, "%s: null pointer", who);
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && G > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d D=%d G=%d", who, B, N, D, G);
  VQA_REQUIRE(G <= kMaxG && N <= 1024, VQA_E_UNSUPPORTED, "%s: needs G <= 8 and N <= 1024 (G=%d N=%d)", who, G, N);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, kAlign) && aligned(d_pooled, 16) && (d_v == nullptr || aligned(d_v, kAlign)),
              VQA_E_UNSUPPORTED, "%s: needs D %% 4 == 0, 16-byte aligned d_pooled and %zu-byte aligned v/d_v (D=%d)", who,
              kAlign, D);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "%s: B=%d exceeds 65535", who, B);
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "%s: p_drop=%f outside [0,1)", who, (double)p_drop);
  VQA_REQUIRE(p_drop == 0.f || (size_t)B * G * D < (1ull << 32), VQA_E_UNSUPPORTED, "%s: dropout needs B*G*D < 2^32", who);
  VQA_REQUIRE(d_first == nullptr || aligned(d_first, 16), VQA_E_UNSUPPORTED, "%s: d_first is not 16-byte aligned", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
#define CALL_BWD(G_) launch_bwd<T, G_>(alpha, v, d_pooled, d_first, d_alpha_ext, d_logits, d_v, dc, B, N, D, s)
  switch (G) {
    case 1: return CALL_BWD(1);
    case 2: return CALL_BWD(2);
    case 3: return CALL_BWD(3);
    case 4: return CALL_BWD(4);
    case 5: return CALL_BWD(5);
    case 6: return CALL_BWD(6);
    case 7: return CALL_BWD(7);
    default: return CALL_BWD(8);
  }
#undef CALL_BWD
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_softmax_attention_pool_fwd(const float* logits, const float* v, float* alpha, float* pooled, int B,
                                              int N, int D, int G, vqa_stream_t stream) {
  return pool_fwd_impl<float>("softmax_attention_pool_fwd", logits, v, alpha, pooled, nullptr, 0.f, 0, nullptr, B, N, D, G, stream);
}

extern "C" int vqa_softmax_attention_pool_fwd_bf16(const float* logits, const vqa_bf16_t* v, float* alpha, float* pooled,
                                                   int B, int N, int D, int G, vqa_stream_t stream) {
  return pool_fwd_impl<bf16>("softmax_attention_pool_fwd_bf16", logits, reinterpret_cast<const bf16*>(v), alpha, pooled, nullptr,
                             0.f, 0, nullptr, B, N, D, G, stream);
}

extern "C" int vqa_softmax_attention_pool_bwd(const float* alpha, const float* v, const float* d_pooled,
                                              const float* d_alpha_ext, float* d_logits, float* d_v, int B, int N,
                                              int D, int G, vqa_stream_t stream) {
  return pool_bwd_impl<float>("softmax_attention_pool_bwd", alpha, v, d_pooled, nullptr, d_alpha_ext, d_logits, d_v, 0.f, 0,
                              nullptr, B, N, D, G, stream);
}

extern "C" int vqa_softmax_attention_pool_bwd_bf16(const float* alpha, const vqa_bf16_t* v, const float* d_pooled,
                                                   const float* d_alpha_ext, float* d_logits, vqa_bf16_t* d_v, int B,
                                                   int N, int D, int G, vqa_stream_t stream) {
  return pool_bwd_impl<bf16>("softmax_attention_pool_bwd_bf16", alpha, reinterpret_cast<const bf16*>(v), d_pooled, nullptr,
                             d_alpha_ext, d_logits, reinterpret_cast<bf16*>(d_v), 0.f, 0, nullptr, B, N, D, G, stream);
}

// MyATT with its glimpse projections' input dropout folded in (config/CoR2.py:142-147): see include/vqa_mi355x.h
extern "C" int vqa_softmax_attention_pool_drop_fwd(const float* logits, const float* v, float* alpha, float* pooled,
                                                   float* first, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                                   int B, int N, int D, int G, vqa_stream_t stream) {
  return pool_fwd_impl<float>("softmax_attention_pool_drop_fwd", logits, v, alpha, pooled, first, p_drop, seed, seed_ptr, B, N,
                              D, G, stream);
}

extern "C" int vqa_softmax_attention_pool_drop_fwd_bf16(const float* logits, const vqa_bf16_t* v, float* alpha, float* pooled,
                                                        float* first, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                                        int B, int N, int D, int G, vqa_stream_t stream) {
  return pool_fwd_impl<bf16>("softmax_attention_pool_drop_fwd_bf16", logits, reinterpret_cast<const bf16*>(v), alpha, pooled,
                             first, p_drop, seed, seed_ptr, B, N, D, G, stream);
}

extern "C" int vqa_softmax_attention_pool_drop_bwd(const float* alpha, const float* v, const float* d_pooled,
                                                   const float* d_first, const float* d_alpha_ext, float* d_logits,
                                                   float* d_v, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B,
                                                   int N, int D, int G, vqa_stream_t stream) {
  return pool_bwd_impl<float>("softmax_attention_pool_drop_bwd", alpha, v, d_pooled, d_first, d_alpha_ext, d_logits, d_v, p_drop,
                              seed, seed_ptr, B, N, D, G, stream);
}

extern "C" int vqa_softmax_attention_pool_drop_bwd_bf16(const float* alpha, const vqa_bf16_t* v, const float* d_pooled,
                                                        const float* d_first, const float* d_alpha_ext, float* d_logits,
                                                        vqa_bf16_t* d_v, float p_drop, uint64_t seed,
                                                        const uint64_t* seed_ptr, int B, int N, int D, int G,
                                                        vqa_stream_t stream) {
  return pool_bwd_impl<bf16>("softmax_attention_pool_drop_bwd_bf16", alpha, reinterpret_cast<const bf16*>(v), d_pooled, d_first,
                             d_alpha_ext, d_logits, reinterpret_cast<bf16*>(d_v), p_drop, seed, seed_ptr, B, N, D, G, stream);
}
