// K3 -- softmax over regions + attention-weighted region pooling.
//
// Replaces F.softmax(dim=1) of the attention conv (config/CoR2.py:83-87, :132) and
// putils.bmatmul(alpha^T, v) (config/CoR2.py:142; putils/__init__.py:89-95: a python loop of B
// [G,N]x[N,D] matmuls + torch.stack).  One pass over v:
//
//   forward : a workgroup owns (sample, 4*NT-wide feature chunk); every lane keeps G float4
//             accumulators and streams the N region rows with coalesced 16-byte loads; the N*G
//             softmax is recomputed per workgroup from the N*G logits (wave64 shuffles).
//   backward: a workgroup owns a sample; d_pooled [G,D] sits in LDS; each wave takes region rows
//             round-robin, forms the G dot products <d_pooled_g, v_n> with a wave64 reduction and
//             (optionally) writes d_v_n in the same sweep; softmax backward closes in LDS.
//
// HBM-bound.  Algorithmic bytes per sample (fp32): forward (N*D + 2*N*G + G*D)*4 = 328 832 B at
// N=36, D=2048, G=4;  backward (N*D + G*D + 3*N*G)*4 (+ N*D*4 when d_v is written).
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

template <int NT, int G>
__global__ __launch_bounds__(NT) void attention_pool_fwd_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ v, float* __restrict__ alpha,
                                                                float* __restrict__ pooled, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);  // [N][G]
  float* stat_s = alpha_s + N * G;                  // [2*kMaxG]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + tid) * 4;
  const bool active = d < D;
  const float* vb = v + (size_t)b * N * D + d;

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
#pragma unroll
  for (int g = 0; g < G; ++g) st4(pooled + ((size_t)b * G + g) * D + d, acc[g]);
}

// Backward: one workgroup per sample.
template <int NT, int G>
__global__ __launch_bounds__(NT) void attention_pool_bwd_kernel(const float* __restrict__ alpha,
                                                                const float* __restrict__ v,
                                                                const float* __restrict__ d_pooled,
                                                                const float* __restrict__ d_alpha_ext,
                                                                float* __restrict__ d_logits, float* __restrict__ d_v,
                                                                int N, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dp_s = reinterpret_cast<float*>(smem);  // [G][D]
  float* alpha_s = dp_s + (size_t)G * D;         // [N][G]
  float* dal_s = alpha_s + N * G;                // [N][G]  dL/dalpha
  float* inner_s = dal_s + N * G;                // [kMaxG]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int b = blockIdx.x;
  const int NG = N * G;
  const float* dpb = d_pooled + (size_t)b * G * D;
  for (int t = tid * 4; t < G * D; t += NT * 4) *reinterpret_cast<float4*>(dp_s + t) = ld4(dpb + t);
  for (int t = tid; t < NG; t += NT) alpha_s[t] = alpha[(size_t)b * NG + t];
  __syncthreads();

  const float* vb = v + (size_t)b * N * D;
  float* dvb = d_v ? d_v + (size_t)b * N * D : nullptr;
  for (int n = wave; n < N; n += NT / 64) {
    float dot[G];
    float a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      dot[g] = 0.f;
      a[g] = alpha_s[n * G + g];
    }
#pragma unroll 4
    for (int d = lane * 4; d < D; d += 256) {
      const float4 x = ld4(vb + (size_t)n * D + d);
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 p = *reinterpret_cast<const float4*>(dp_s + (size_t)g * D + d);
        dot[g] += dot4(x, p);
        o = fma4(a[g], p, o);
      }
      if (dvb) st4(dvb + (size_t)n * D + d, o);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float r = wave_sum(dot[g]);
      if (lane == 0) dal_s[n * G + g] = r;
    }
  }
  __syncthreads();
  if (d_alpha_ext != nullptr)
    for (int t = tid; t < NG; t += NT) dal_s[t] += d_alpha_ext[(size_t)b * NG + t];
  __syncthreads();
  for (int g = wave; g < G; g += NT / 64) {
    float s = 0.f;
    for (int n = lane; n < N; n += 64) s += alpha_s[n * G + g] * dal_s[n * G + g];
    s = wave_sum(s);
    if (lane == 0) inner_s[g] = s;
  }
  __syncthreads();
  for (int t = tid; t < NG; t += NT) d_logits[(size_t)b * NG + t] = alpha_s[t] * (dal_s[t] - inner_s[t % G]);
}

template <int G>
static int launch_fwd(const float* logits, const float* v, float* alpha, float* pooled, int B, int N, int D,
                      hipStream_t s) {
  constexpr int NT = 256;
  const size_t lds = ((size_t)N * G + 2 * kMaxG) * sizeof(float);
  dim3 grid((D / 4 + NT - 1) / NT, B);
  hipLaunchKernelGGL((attention_pool_fwd_kernel<NT, G>), grid, dim3(NT), lds, s, logits, v, alpha, pooled, N, D);
  return check_launch("softmax_attention_pool_fwd");
}

template <int G>
static int launch_bwd(const float* alpha, const float* v, const float* d_pooled, const float* d_alpha_ext,
                      float* d_logits, float* d_v, int B, int N, int D, hipStream_t s) {
  constexpr int NT = 512;
  const size_t lds = ((size_t)G * D + 2 * (size_t)N * G + kMaxG) * sizeof(float);
  VQA_REQUIRE(lds <= 160 * 1024, VQA_E_UNSUPPORTED, "softmax_attention_pool_bwd: G*D=%d needs %zu B of LDS (> 160 KiB)",
              G * D, lds);
  VQA_ENSURE_LDS((attention_pool_bwd_kernel<NT, G>), lds);
  hipLaunchKernelGGL((attention_pool_bwd_kernel<NT, G>), dim3(B), dim3(NT), lds, s, alpha, v, d_pooled, d_alpha_ext,
                     d_logits, d_v, N, D);
  return check_launch("softmax_attention_pool_bwd");
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_softmax_attention_pool_fwd(const float* logits, const float* v, float* alpha, float* pooled, int B,
                                              int N, int D, int G, vqa_stream_t stream) {
  VQA_REQUIRE(logits && v && alpha && pooled, VQA_E_BADARG, "softmax_attention_pool_fwd: null pointer");
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && G > 0, VQA_E_BADARG, "softmax_attention_pool_fwd: bad sizes B=%d N=%d D=%d G=%d",
              B, N, D, G);
  VQA_REQUIRE(G <= kMaxG && N <= 1024, VQA_E_UNSUPPORTED, "softmax_attention_pool_fwd: needs G <= 8 and N <= 1024 (G=%d N=%d)",
              G, N);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, 16) && aligned(pooled, 16), VQA_E_UNSUPPORTED,
              "softmax_attention_pool_fwd: needs D %% 4 == 0 and 16-byte aligned v/pooled (D=%d)", D);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "softmax_attention_pool_fwd: B=%d exceeds 65535", B);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL_FWD(G_) launch_fwd<G_>(logits, v, alpha, pooled, B, N, D, s)
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

extern "C" int vqa_softmax_attention_pool_bwd(const float* alpha, const float* v, const float* d_pooled,
                                              const float* d_alpha_ext, float* d_logits, float* d_v, int B, int N,
                                              int D, int G, vqa_stream_t stream) {
  VQA_REQUIRE(alpha && v && d_pooled && d_logits, VQA_E_BADARG, "softmax_attention_pool_bwd: null pointer");
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && G > 0, VQA_E_BADARG, "softmax_attention_pool_bwd: bad sizes B=%d N=%d D=%d G=%d",
              B, N, D, G);
  VQA_REQUIRE(G <= kMaxG && N <= 1024, VQA_E_UNSUPPORTED, "softmax_attention_pool_bwd: needs G <= 8 and N <= 1024 (G=%d N=%d)",
              G, N);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, 16) && aligned(d_pooled, 16) && (d_v == nullptr || aligned(d_v, 16)),
              VQA_E_UNSUPPORTED, "softmax_attention_pool_bwd: needs D %% 4 == 0 and 16-byte aligned v/d_pooled/d_v (D=%d)", D);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL_BWD(G_) launch_bwd<G_>(alpha, v, d_pooled, d_alpha_ext, d_logits, d_v, B, N, D, s)
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
