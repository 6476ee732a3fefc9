// Epilogue / prologue of the batched [B, .] layers (layers.my_linears: the question projections, the sigmoid gates,
// Mutan's question-side ranks, the per-glimpse projections -- MyLinear of config/CoR2.py:94-122, putils.Linear of
// putils/__init__.py:16-33): what surrounds their one batched library GEMM, as one kernel each way.
//
//   forward : out = act(y[g,b,:] + bias[g,:]), written [G,B,A] or [B,G,A]          (bias add + activation + layout)
//   backward: gz[g,b,:] = gy * act'(out), always [G,B,A] (the layout the two backward GEMMs read);
//             d_bias[g,:] = sum_b gz[g,b,:]   (16 columns x 16 row slices per workgroup, fixed order, no atomics)
//
// act: 0 none, 1 relu, 2 sigmoid.  Tiny tensors (B*G*A floats): these kernels exist to cut launches, not bytes.
#include "common.hpp"

namespace vqa {

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == 1) return fmaxf(z, 0.f);
  if (act == 2) return 1.f / (1.f + expf(-z));
  return z;
}
__device__ __forceinline__ float act_bwd(float g, float out, int act) {
  if (act == 1) return out > 0.f ? g : 0.f;
  if (act == 2) return g * out * (1.f - out);
  return g;
}

__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ y, const float* __restrict__ bias,
                                                       int bias_stride, float* __restrict__ out, int G, int B, int A, int act,
                                                       int group_first) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= (size_t)G * B * A) return;
  const int a = (int)(e % A);
  const size_t t = e / A;
  const int b = (int)(t % B), g = (int)(t / B);
  const float z = y[e] + (bias != nullptr ? bias[(size_t)g * bias_stride + a] : 0.f);
  out[group_first ? e : ((size_t)b * G + g) * A + a] = act_fwd(z, act);
}

// grid (ceil(A/16), G); gy / out are [G,B,A] (group_first) or [B,G,A]
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const float* __restrict__ gy, const float* __restrict__ out,
                                                             float* __restrict__ gz, float* __restrict__ d_bias,
                                                             int d_bias_stride, int G, int B, int A, int act,
                                                             int group_first) {
  __shared__ float part[16][17];
  const int c = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int g = blockIdx.y;
  const int a = blockIdx.x * 16 + c;
  const bool active = a < A;
  const int ac = active ? a : A - 1;
  const size_t row_stride = group_first ? (size_t)A : (size_t)G * A;
  const size_t base = group_first ? (size_t)g * B * A + ac : (size_t)g * A + ac;
  float acc = 0.f;
  for (int b0 = slice; b0 < B; b0 += 128) {  // 16 slices x 8 rows in flight (the loop is a chain of L2 round trips)
    float gv[8], ov[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const size_t off = base + (size_t)min(b0 + 16 * k, B - 1) * row_stride;
      gv[k] = gy[off];
      ov[k] = out[off];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int b = b0 + 16 * k;
      if (b < B) {
        const float z = act_bwd(gv[k], ov[k], act);
        if (active) gz[((size_t)g * B + b) * A + a] = z;
        acc += z;
      }
    }
  }
  part[slice][c] = acc;
  __syncthreads();
  if (slice == 0 && active && d_bias != nullptr) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[q][c];
    d_bias[(size_t)g * d_bias_stride + a] = t;
  }
}

// Rank sum of the vector-vector Mutan fusion (putils/__init__.py:232-238 with 2-D inputs: fusion_final):
//   out[b,:] = sum_r h1[b,r,:] * h2[b,r,:];   backward d_h1 = g * h2, d_h2 = g * h1 (g broadcast over r)
// one kernel each way instead of multiply + reduce and two broadcast multiplies.  Lane = one float2 column of a sample.
__global__ __launch_bounds__(256) void rank_product_fwd_kernel(const float* __restrict__ h1, const float* __restrict__ h2,
                                                               float* __restrict__ out, int B, int R, int H) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= (size_t)B * H) return;
  const size_t b = e / H, h = e % H;
  float2 acc = make_float2(0.f, 0.f);
  for (int r = 0; r < R; ++r) {
    const size_t o = (b * R + r) * H + h;
    const float2 x = ld2(h1 + o), y = ld2(h2 + o);
    acc.x = fmaf(x.x, y.x, acc.x);
    acc.y = fmaf(x.y, y.y, acc.y);
  }
  st2(out + e, acc);
}
__global__ __launch_bounds__(256) void rank_product_bwd_kernel(const float* __restrict__ g, const float* __restrict__ h1,
                                                               const float* __restrict__ h2, float* __restrict__ d_h1,
                                                               float* __restrict__ d_h2, int B, int R, int H) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= (size_t)B * H) return;
  const size_t b = e / H, h = e % H;
  const float2 gv = ld2(g + e);
  for (int r = 0; r < R; ++r) {
    const size_t o = (b * R + r) * H + h;
    const float2 x = ld2(h1 + o), y = ld2(h2 + o);
    st2(d_h1 + o, make_float2(gv.x * y.x, gv.y * y.y));
    st2(d_h2 + o, make_float2(gv.x * x.x, gv.y * x.y));
  }
}

// F.dropout of the [B,.]-sized tensors of the path (config/CoR2.py:108-110: MyLinear's input dropout) with the counter-hash
// mask of K2 / K5 instead of torch's Philox stream, and -- G > 1 -- the G independent draws over ONE input that G
// same-shaped MyLinear layers reading the same tensor make (the four question projections of CoR2 read q_feature):
//   out[g][m][k] = x[m][k] * keep((g M + m) K + k);     backward  d_x[m][k] = sum_g keep(.) * gy[g][m][k]
// VEC elements per lane (K % VEC == 0).
template <int VEC>
__device__ __forceinline__ void drop_vec(uint32_t e, const DropCfg& dc, float (&k)[VEC]) {
  if constexpr (VEC == 4) {
    const float4 t = drop_quad(e, dc);
    k[0] = t.x, k[1] = t.y, k[2] = t.z, k[3] = t.w;
  } else if constexpr (VEC == 2) {
    const float2 t = drop_pair(e, dc);
    k[0] = t.x, k[1] = t.y;
  } else {
    k[0] = drop_one(e, dc);
  }
}
template <int VEC>
__global__ __launch_bounds__(256) void dropout_groups_fwd_kernel(const float* __restrict__ x, int ldx, float* __restrict__ out,
                                                                 int G, int M, int K, DropCfg dc) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;     // element of out [G,M,K]
  if (e >= (size_t)G * M * K) return;
  const int k = (int)(e % K);
  const int m = (int)((e / K) % M);
  float kp[VEC];
  drop_vec<VEC>((uint32_t)e, dc, kp);
#pragma unroll
  for (int j = 0; j < VEC; ++j) out[e + j] = x[(size_t)m * ldx + k + j] * kp[j];
}
template <int VEC>
__global__ __launch_bounds__(256) void dropout_groups_bwd_kernel(const float* __restrict__ gy, float* __restrict__ d_x, int G,
                                                                 int M, int K, DropCfg dc) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;     // element of d_x [M,K]
  const size_t n = (size_t)M * K;
  if (e >= n) return;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  for (int g = 0; g < G; ++g) {
    float kp[VEC];
    drop_vec<VEC>((uint32_t)(g * n + e), dc, kp);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = fmaf(gy[g * n + e + j], kp[j], acc[j]);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) d_x[e + j] = acc[j];
}

}  // namespace vqa

using namespace vqa;

namespace vqa {
// t = a * b with b a strided-row view (rows of `cols` floats, ldb apart): the relation step's t = q1 * pooled[:, 0] read straight
// from the [B,G,D] pooled tensor (no contiguous copy of the slice, no torch multiply)
__global__ __launch_bounds__(256) void gate_product_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long ldb,
                                                               float* __restrict__ t, int cols4, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const size_t row = i / (size_t)cols4, c4 = i - row * (size_t)cols4;
  st4(t + 4 * i, mul4(ld4(a + 4 * i), ld4(b + row * ldb + 4 * c4)));
}
// backward of (t, c) = (a * b, c) handed to TWO consumers each: d_a = (g1 + g2) b, d_b = (g1 + g2) a, d_c = h1 + h2
// (g2 / h2 may be NULL).  One launch for what autograd would do in two accumulations and two products.  b: rows ldb apart.
__global__ __launch_bounds__(256) void gate_product_bwd_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                               const float* __restrict__ h1, const float* __restrict__ h2,
                                                               const float* __restrict__ a, const float* __restrict__ b, long ldb,
                                                               float* __restrict__ d_a, float* __restrict__ d_b,
                                                               float* __restrict__ d_c, int cols4, size_t n4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const size_t row = i / (size_t)cols4, c4 = i - row * (size_t)cols4;
  float4 g = ld4(g1 + 4 * i);
  if (g2 != nullptr) g = add4(g, ld4(g2 + 4 * i));
  float4 h = ld4(h1 + 4 * i);
  if (h2 != nullptr) h = add4(h, ld4(h2 + 4 * i));
  st4(d_a + 4 * i, mul4(g, ld4(b + row * ldb + 4 * c4)));
  st4(d_b + 4 * i, mul4(g, ld4(a + 4 * i)));
  st4(d_c + 4 * i, h);
}
}  // namespace vqa

extern "C" int vqa_gate_product_fwd(const float* a, const float* b, long ldb, float* t, int rows, int cols, vqa_stream_t stream) {
  VQA_REQUIRE(a && b && t, VQA_E_BADARG, "gate_product_fwd: null pointer");
  VQA_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ldb >= cols && ldb % 4 == 0, VQA_E_UNSUPPORTED,
              "gate_product_fwd: cols = %d and ldb = %ld must be positive multiples of 4, ldb >= cols", cols, ldb);
  VQA_REQUIRE(vqa::aligned(a, 16) && vqa::aligned(b, 16) && vqa::aligned(t, 16), VQA_E_UNSUPPORTED,
              "gate_product_fwd: tensors must be 16-byte aligned");
  const size_t n4 = (size_t)rows * cols / 4;
  VQA_LAUNCH(vqa::gate_product_fwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a, b,
             ldb, t, cols / 4, n4);
  return vqa::check_launch("gate_product_fwd");
}

extern "C" int vqa_gate_product_bwd(const float* g1, const float* g2, const float* h1, const float* h2, const float* a,
                                    const float* b, long ldb, float* d_a, float* d_b, float* d_c, int rows, int cols,
                                    vqa_stream_t stream) {
  VQA_REQUIRE(g1 && h1 && a && b && d_a && d_b && d_c, VQA_E_BADARG, "gate_product_bwd: null pointer");
  VQA_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ldb >= cols && ldb % 4 == 0, VQA_E_UNSUPPORTED,
              "gate_product_bwd: cols = %d and ldb = %ld must be positive multiples of 4, ldb >= cols", cols, ldb);
  VQA_REQUIRE(vqa::aligned(g1, 16) && vqa::aligned(h1, 16) && vqa::aligned(a, 16) && vqa::aligned(b, 16) && vqa::aligned(d_a, 16) &&
                  vqa::aligned(d_b, 16) && vqa::aligned(d_c, 16) && (g2 == nullptr || vqa::aligned(g2, 16)) &&
                  (h2 == nullptr || vqa::aligned(h2, 16)),
              VQA_E_UNSUPPORTED, "gate_product_bwd: tensors must be 16-byte aligned");
  const size_t n4 = (size_t)rows * cols / 4;
  VQA_LAUNCH(vqa::gate_product_bwd_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g1, g2, h1, h2, a, b, ldb, d_a, d_b, d_c, cols / 4, n4);
  return vqa::check_launch("gate_product_bwd");
}

// bf16 -> fp32, eight elements per lane: the feed's bf16 TRANSPORT format of the region features (half the PCIe bytes of the
// step's dominant stream) widened on the device for the fp32 path -- exact (every bf16 is an fp32), so a step fed this way on
// bf16-representable features is bit-identical to the fp32-fed one
__global__ __launch_bounds__(256) void widen_bf16_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, size_t n) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
  if (i + 8 <= n) {
    const uint4 p = *reinterpret_cast<const uint4*>(src + i);
    const uint32_t w[4] = {p.x, p.y, p.z, p.w};
    float4 a, b;
    a.x = __uint_as_float(w[0] << 16);
    a.y = __uint_as_float(w[0] & 0xFFFF0000u);
    a.z = __uint_as_float(w[1] << 16);
    a.w = __uint_as_float(w[1] & 0xFFFF0000u);
    b.x = __uint_as_float(w[2] << 16);
    b.y = __uint_as_float(w[2] & 0xFFFF0000u);
    b.z = __uint_as_float(w[3] << 16);
    b.w = __uint_as_float(w[3] & 0xFFFF0000u);
    *reinterpret_cast<float4*>(dst + i) = a;
    *reinterpret_cast<float4*>(dst + i + 4) = b;
  } else {
    for (size_t k = i; k < n; ++k) dst[k] = __uint_as_float((uint32_t)src[k] << 16);
  }
}

extern "C" int vqa_widen_bf16(const vqa_bf16_t* src, float* dst, size_t n, vqa_stream_t stream) {
  VQA_REQUIRE(src && dst, VQA_E_BADARG, "widen_bf16: null pointer");
  VQA_REQUIRE(aligned(src, 16) && aligned(dst, 16), VQA_E_UNSUPPORTED, "widen_bf16: src and dst must be 16-byte aligned");
  if (n == 0) return VQA_OK;
  VQA_LAUNCH(widen_bf16_kernel, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, static_cast<hipStream_t>(stream), src, dst, n);
  return check_launch("widen_bf16");
}

extern "C" int vqa_bias_act(const float* y, const float* bias, int bias_stride, float* out, int G, int B, int A, int act,
                            int group_first, vqa_stream_t stream) {
  VQA_REQUIRE(y && out, VQA_E_BADARG, "bias_act: null pointer");
  VQA_REQUIRE(G > 0 && B > 0 && A > 0 && (bias == nullptr || bias_stride >= A), VQA_E_BADARG,
              "bias_act: bad sizes G=%d B=%d A=%d bias_stride=%d", G, B, A, bias_stride);
  VQA_REQUIRE(act >= 0 && act <= 2, VQA_E_BADARG, "bias_act: act must be 0 (none), 1 (relu) or 2 (sigmoid), got %d", act);
  const size_t n = (size_t)G * B * A;
  VQA_LAUNCH(bias_act_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), y, bias,
                     bias_stride, out, G, B, A, act, group_first);
  return check_launch("bias_act");
}

extern "C" int vqa_act_bwd_colsum(const float* gy, const float* out, float* gz, float* d_bias, int d_bias_stride, int G, int B,
                                  int A, int act, int group_first, vqa_stream_t stream) {
  VQA_REQUIRE(gy && out && gz, VQA_E_BADARG, "act_bwd_colsum: null pointer");
  VQA_REQUIRE(d_bias == nullptr || d_bias_stride >= A, VQA_E_BADARG, "act_bwd_colsum: d_bias_stride %d < A = %d", d_bias_stride, A);
  VQA_REQUIRE(G > 0 && B > 0 && A > 0 && G <= 65535, VQA_E_BADARG, "act_bwd_colsum: bad sizes G=%d B=%d A=%d", G, B, A);
  VQA_REQUIRE(act >= 0 && act <= 2, VQA_E_BADARG, "act_bwd_colsum: act must be 0, 1 or 2, got %d", act);
  VQA_LAUNCH(act_bwd_colsum_kernel, dim3((A + 15) / 16, G), dim3(256), 0, static_cast<hipStream_t>(stream), gy, out, gz,
                     d_bias, d_bias_stride, G, B, A, act, group_first);
  return check_launch("act_bwd_colsum");
}

extern "C" int vqa_rank_product_fwd(const float* h1, const float* h2, float* out, int B, int R, int H, vqa_stream_t stream) {
  VQA_REQUIRE(h1 && h2 && out, VQA_E_BADARG, "rank_product_fwd: null pointer");
  VQA_REQUIRE(B > 0 && R > 0 && H > 0 && H % 2 == 0, VQA_E_BADARG, "rank_product_fwd: bad sizes B=%d R=%d H=%d (H even)", B, R, H);
  VQA_REQUIRE(aligned(h1, 8) && aligned(h2, 8) && aligned(out, 8), VQA_E_UNSUPPORTED, "rank_product_fwd: tensors must be 8-byte aligned");
  const size_t n = (size_t)B * H / 2;
  VQA_LAUNCH(rank_product_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     h1, h2, out, B, R, H);
  return check_launch("rank_product_fwd");
}

extern "C" int vqa_rank_product_bwd(const float* g, const float* h1, const float* h2, float* d_h1, float* d_h2, int B, int R,
                                    int H, vqa_stream_t stream) {
  VQA_REQUIRE(g && h1 && h2 && d_h1 && d_h2, VQA_E_BADARG, "rank_product_bwd: null pointer");
  VQA_REQUIRE(B > 0 && R > 0 && H > 0 && H % 2 == 0, VQA_E_BADARG, "rank_product_bwd: bad sizes B=%d R=%d H=%d (H even)", B, R, H);
  VQA_REQUIRE(aligned(g, 8) && aligned(h1, 8) && aligned(h2, 8) && aligned(d_h1, 8) && aligned(d_h2, 8), VQA_E_UNSUPPORTED,
              "rank_product_bwd: tensors must be 8-byte aligned");
  const size_t n = (size_t)B * H / 2;
  VQA_LAUNCH(rank_product_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     g, h1, h2, d_h1, d_h2, B, R, H);
  return check_launch("rank_product_bwd");
}

extern "C" int vqa_dropout_groups_fwd(const float* x, int ldx, float* out, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                      int G, int M, int K, vqa_stream_t stream) {
  VQA_REQUIRE(x && out, VQA_E_BADARG, "dropout_groups_fwd: null pointer");
  VQA_REQUIRE(G > 0 && M > 0 && K > 0 && ldx >= K, VQA_E_BADARG, "dropout_groups_fwd: bad sizes G=%d M=%d K=%d ldx=%d", G, M, K, ldx);
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "dropout_groups_fwd: p_drop=%f outside [0,1)", (double)p_drop);
  VQA_REQUIRE((size_t)G * M * K < (1ull << 32), VQA_E_UNSUPPORTED, "dropout_groups_fwd: needs G*M*K < 2^32");
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  const size_t n = (size_t)G * M * K;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (K % 4 == 0)
    VQA_LAUNCH(dropout_groups_fwd_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, x, ldx, out, G, M, K, dc);
  else if (K % 2 == 0)
    VQA_LAUNCH(dropout_groups_fwd_kernel<2>, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, x, ldx, out, G, M, K, dc);
  else
    VQA_LAUNCH(dropout_groups_fwd_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, ldx, out, G, M, K, dc);
  return check_launch("dropout_groups_fwd");
}

extern "C" int vqa_dropout_groups_bwd(const float* gy, float* d_x, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int G,
                                      int M, int K, vqa_stream_t stream) {
  VQA_REQUIRE(gy && d_x, VQA_E_BADARG, "dropout_groups_bwd: null pointer");
  VQA_REQUIRE(G > 0 && M > 0 && K > 0, VQA_E_BADARG, "dropout_groups_bwd: bad sizes G=%d M=%d K=%d", G, M, K);
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "dropout_groups_bwd: p_drop=%f outside [0,1)", (double)p_drop);
  VQA_REQUIRE((size_t)G * M * K < (1ull << 32), VQA_E_UNSUPPORTED, "dropout_groups_bwd: needs G*M*K < 2^32");
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  const size_t n = (size_t)M * K;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n % 4 == 0)
    VQA_LAUNCH(dropout_groups_bwd_kernel<4>, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, gy, d_x, G, M, K, dc);
  else if (n % 2 == 0)
    VQA_LAUNCH(dropout_groups_bwd_kernel<2>, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, gy, d_x, G, M, K, dc);
  else
    VQA_LAUNCH(dropout_groups_bwd_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, gy, d_x, G, M, K, dc);
  return check_launch("dropout_groups_bwd");
}
