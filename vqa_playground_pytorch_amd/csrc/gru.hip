// Gate math of the BayesianGRU question encoder (SkipThoughts' uni-skip GRU), one kernel per time step each way.
//
// Replaces, per step t of putils.BayesianGRU.forward (putils/__init__.py:704-731 with the cell of :604-646):
//     hr, hi, hn = h*m_r, h*m_i, h*m_n                     (sequence-shared dropout masks, SequentialDropout :503-539)
//     r = sigmoid(gi_r[t] + W_hr hr);  i = sigmoid(gi_i[t] + W_hi hi);  n = af(gi_n[t] + r * (W_hn hn))
//     h' = (1 - i) * n + i * h
// -- about fifteen elementwise launches around three small GEMMs, and twice that in autograd's backward.  Here the
// three recurrent GEMMs of a step are ONE batched library GEMM a[g] = hm[g] W_g^T (ops.GruSequence) and everything
// else is this file:
//   forward : (gi[:, :, t, :], a, h) -> h', the three masked copies of h' that feed the next step's GEMM (written
//             straight into the [3,T,B,H] history the weight-gradient GEMM reads at the end), and r, i, n, a_n saved
//   backward: dh_t = d_out[t] + carry + sum_g dhm[g]*m_g;  gate derivatives -> gz[g] (gradient at the GEMM outputs,
//             written into its [3,T,B,H] history), d_gi[:, :, t, :], carry' = dh_t * i
// HBM-light (B*H elements, ~12 floats each).  af: 1 relu, 3 tanh.  Lane = 4 adjacent hidden units.
#include "common.hpp"

namespace vqa {

__device__ __forceinline__ float sigmoidf(float z) { return 1.f / (1.f + expf(-z)); }
__device__ __forceinline__ float gru_af(float z, int af) { return af == 1 ? fmaxf(z, 0.f) : tanhf(z); }
__device__ __forceinline__ float gru_af_grad(float n, int af) { return af == 1 ? (n > 0.f ? 1.f : 0.f) : 1.f - n * n; }

#define VQA_GRU_FOR4(EXPR)                 \
  {                                        \
    { constexpr int c = 0; EXPR; }         \
    { constexpr int c = 1; EXPR; }         \
    { constexpr int c = 2; EXPR; }         \
    { constexpr int c = 3; EXPR; }         \
  }
__device__ __forceinline__ float& f4(float4& v, int c) { return reinterpret_cast<float*>(&v)[c]; }
__device__ __forceinline__ float f4(const float4& v, int c) { return reinterpret_cast<const float*>(&v)[c]; }

// gi [3,B,T,H]; a [3,B,H]; h_prev [B,H]; masks [3,B,H] or null; h_new [B,H] (slot t of the [T,B,H] output);
// hm_next: base of slot t+1 of the [3,T,B,H] history (group stride hist_gs) or null at the last step;
// saved r, i, n, an: slot t of [T,B,H] each.
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ a,
                                                            const float* __restrict__ h_prev,
                                                            const float* __restrict__ masks, float* __restrict__ h_new,
                                                            float* __restrict__ hm_next, size_t hist_gs,
                                                            float* __restrict__ r_s, float* __restrict__ i_s,
                                                            float* __restrict__ n_s, float* __restrict__ an_s, int B, int T,
                                                            int H, int t, int af) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t BH = (size_t)B * H;
  if (e >= BH) return;
  const size_t b = e / H, h = e % H;
  const size_t gio = (b * T + t) * H + h, gig = (size_t)B * T * H;
  const float4 gr = ld4(gi + gio), gz = ld4(gi + gig + gio), gn = ld4(gi + 2 * gig + gio);
  const float4 ar = ld4(a + e), ai = ld4(a + BH + e), an = ld4(a + 2 * BH + e);
  const float4 hp = ld4(h_prev + e);
  float4 r, i, n, hn;
  VQA_GRU_FOR4(f4(r, c) = sigmoidf(f4(gr, c) + f4(ar, c)); f4(i, c) = sigmoidf(f4(gz, c) + f4(ai, c));
               f4(n, c) = gru_af(f4(gn, c) + f4(r, c) * f4(an, c), af);
               f4(hn, c) = (1.f - f4(i, c)) * f4(n, c) + f4(i, c) * f4(hp, c));
  st4(h_new + e, hn);
  st4(r_s + e, r);
  st4(i_s + e, i);
  st4(n_s + e, n);
  st4(an_s + e, an);
  if (hm_next != nullptr) {
#pragma unroll
    for (int g = 0; g < 3; ++g) st4(hm_next + g * hist_gs + e, masks != nullptr ? mul4(hn, ld4(masks + g * BH + e)) : hn);
  }
}

// d_out_t [B,H] or null; carry_in [B,H] or null; dhm [3,B,H] or null (both null at the last step); masks [3,B,H] or null;
// saved r, i, n, an (slot t), h_prev [B,H]; outputs gz: slot t of the [3,T,B,H] history (group stride hist_gs),
// d_gi [3,B,T,H] (slot t), carry_out [B,H].
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ d_out_t,
                                                            const float* __restrict__ carry_in,
                                                            const float* __restrict__ dhm, const float* __restrict__ masks,
                                                            const float* __restrict__ r_s, const float* __restrict__ i_s,
                                                            const float* __restrict__ n_s, const float* __restrict__ an_s,
                                                            const float* __restrict__ h_prev, float* __restrict__ gz,
                                                            size_t hist_gs, float* __restrict__ d_gi,
                                                            float* __restrict__ carry_out, int B, int T, int H, int t, int af) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const size_t BH = (size_t)B * H;
  if (e >= BH) return;
  const size_t b = e / H, h = e % H;
  float4 dh = d_out_t != nullptr ? ld4(d_out_t + e) : make_float4(0.f, 0.f, 0.f, 0.f);
  if (carry_in != nullptr) dh = add4(dh, ld4(carry_in + e));
  if (dhm != nullptr) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float4 d = ld4(dhm + g * BH + e);
      dh = add4(dh, masks != nullptr ? mul4(d, ld4(masks + g * BH + e)) : d);
    }
  }
  const float4 r = ld4(r_s + e), i = ld4(i_s + e), n = ld4(n_s + e), an = ld4(an_s + e), hp = ld4(h_prev + e);
  float4 dzr, dzi, dzn, dan, co;
  VQA_GRU_FOR4(const float d = f4(dh, c); const float dn = d * (1.f - f4(i, c)) * gru_af_grad(f4(n, c), af);
               f4(dzn, c) = dn; f4(dan, c) = dn * f4(r, c);
               f4(dzr, c) = dn * f4(an, c) * f4(r, c) * (1.f - f4(r, c));
               f4(dzi, c) = d * (f4(hp, c) - f4(n, c)) * f4(i, c) * (1.f - f4(i, c)); f4(co, c) = d * f4(i, c));
  st4(gz + e, dzr);
  st4(gz + hist_gs + e, dzi);
  st4(gz + 2 * hist_gs + e, dan);
  const size_t gio = (b * T + t) * H + h, gig = (size_t)B * T * H;
  st4(d_gi + gio, dzr);
  st4(d_gi + gig + gio, dzi);
  st4(d_gi + 2 * gig + gio, dzn);
  st4(carry_out + e, co);
}

static int gru_check(const char* who, int B, int T, int H, int t, int af) {
  VQA_REQUIRE(B > 0 && T > 0 && H > 0 && t >= 0 && t < T, VQA_E_BADARG, "%s: bad sizes B=%d T=%d H=%d t=%d", who, B, T, H, t);
  VQA_REQUIRE(H % 4 == 0, VQA_E_UNSUPPORTED, "%s: needs H %% 4 == 0 (H=%d)", who, H);
  VQA_REQUIRE(af == 1 || af == 3, VQA_E_BADARG, "%s: af must be 1 (relu) or 3 (tanh), got %d", who, af);
  return VQA_OK;
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_gru_gates_fwd(const float* gi, const float* a, const float* h_prev, const float* masks, float* h_new,
                                 float* hm_next, size_t hist_group_stride, float* r_s, float* i_s, float* n_s, float* an_s,
                                 int B, int T, int H, int t, int af, vqa_stream_t stream) {
  VQA_REQUIRE(gi && a && h_prev && h_new && r_s && i_s && n_s && an_s, VQA_E_BADARG, "gru_gates_fwd: null pointer");
  int rc = gru_check("gru_gates_fwd", B, T, H, t, af);
  if (rc != VQA_OK) return rc;
  const size_t n4 = (size_t)B * H / 4;
  VQA_LAUNCH(gru_gates_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), gi,
                     a, h_prev, masks, h_new, hm_next, hist_group_stride, r_s, i_s, n_s, an_s, B, T, H, t, af);
  return check_launch("gru_gates_fwd");
}

extern "C" int vqa_gru_gates_bwd(const float* d_out_t, const float* carry_in, const float* dhm, const float* masks,
                                 const float* r_s, const float* i_s, const float* n_s, const float* an_s,
                                 const float* h_prev, float* gz, size_t hist_group_stride, float* d_gi, float* carry_out,
                                 int B, int T, int H, int t, int af, vqa_stream_t stream) {
  VQA_REQUIRE(r_s && i_s && n_s && an_s && h_prev && gz && d_gi && carry_out, VQA_E_BADARG, "gru_gates_bwd: null pointer");
  int rc = gru_check("gru_gates_bwd", B, T, H, t, af);
  if (rc != VQA_OK) return rc;
  const size_t n4 = (size_t)B * H / 4;
  VQA_LAUNCH(gru_gates_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     d_out_t, carry_in, dhm, masks, r_s, i_s, n_s, an_s, h_prev, gz, hist_group_stride, d_gi, carry_out, B, T, H,
                     t, af);
  return check_launch("gru_gates_bwd");
}
