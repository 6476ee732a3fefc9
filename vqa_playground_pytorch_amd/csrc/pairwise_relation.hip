// K1 -- pairwise relation build + alpha-weighted reduce (CoR2 step 2).
//
// Replaces config/CoR2.py:191-199 (decare_cat) + :216 of the reference, which materialise five
// [B,N,N,D] fp32 tensors (5.4 GB each at B=512).  Here the [N,N,D] relation tensor of a sample never
// exists: a workgroup stages the [N x 4*NT] region tile of one sample in LDS with coalesced 16-byte
// loads (each lane owns one float4 column and its N LDS slots, so the tile needs no barrier and its
// ds_read_b128 / ds_write_b128 are conflict-free: consecutive lanes, consecutive 16-byte slots) and
// produces the N output rows straight from it.
//
// HBM-bound.  Algorithmic bytes per sample (fp32): forward (2*N*D + 2*D + N)*4 = 606 352 B at
// N=36, D=2048; backward (v, g read once, dq1/dq2/dalpha written) ~ 622 880 B.
#include "common.hpp"

namespace vqa {

template <int NT>
__global__ __launch_bounds__(NT) void pairwise_fwd_kernel(const float* __restrict__ v, const float* __restrict__ q1,
                                                          const float* __restrict__ q2, const float* __restrict__ alpha,
                                                          int astride, float* __restrict__ v2, int N, int D, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tile = reinterpret_cast<float4*>(smem);                  // [N][NT]
  float* alpha_s = reinterpret_cast<float*>(tile + (size_t)N * NT);  // [N]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + tid) * 4;
  const bool active = d < D;
  for (int i = tid; i < N; i += NT) alpha_s[i] = alpha[((size_t)b * N + i) * astride];
  const size_t base = (size_t)b * N * D + d;
  float4 q1v = make_float4(0.f, 0.f, 0.f, 0.f), q2v = q1v;
  if (active) {
    q1v = ld4(q1 + (size_t)b * D + d);
    q2v = ld4(q2 + (size_t)b * D + d);
#pragma unroll 12
    for (int i = 0; i < N; ++i) tile[i * NT + tid] = ld4(v + base + (size_t)i * D);
  }
  __syncthreads();  // alpha_s; the tile column is private to this lane
  if (!active) return;
  if (mode == 1) {
    // factored: v2_j = q1 * (sum_i a_i v_i) + (sum_i a_i) * q2 * v_j
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float asum = 0.f;
#pragma unroll 4
    for (int i = 0; i < N; ++i) {
      const float a = alpha_s[i];
      s = fma4(a, tile[i * NT + tid], s);
      asum += a;
    }
    const float4 s1 = mul4(q1v, s);
    const float4 c2 = scale4(asum, q2v);
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
      const float4 t = tile[j * NT + tid];
      st4(v2 + base + (size_t)j * D,
          make_float4(fmaf(c2.x, t.x, s1.x), fmaf(c2.y, t.y, s1.y), fmaf(c2.z, t.z, s1.z), fmaf(c2.w, t.w, s1.w)));
    }
  } else {
    // pairwise: every (i, j) term of the relation tensor is formed and weighted, as the reference sums it
    for (int j = 0; j < N; ++j) {
      const float4 t = mul4(tile[j * NT + tid], q2v);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int i = 0; i < N; ++i) {
        const float4 vi = tile[i * NT + tid];
        const float a = alpha_s[i];
        acc.x = fmaf(a, fmaf(vi.x, q1v.x, t.x), acc.x);
        acc.y = fmaf(a, fmaf(vi.y, q1v.y, t.y), acc.y);
        acc.z = fmaf(a, fmaf(vi.z, q1v.z, t.z), acc.z);
        acc.w = fmaf(a, fmaf(vi.w, q1v.w, t.w), acc.w);
      }
      st4(v2 + base + (size_t)j * D, acc);
    }
  }
}

// Backward.  g = dL/dv2.
//   gsum = sum_j g_j ; gv = sum_j g_j*v_j ; pooled = sum_i a_i v_i ; asum = sum_i a_i
//   dq1 = pooled*gsum ; dq2 = asum*gv ; dalpha_i = <v_i, q1*gsum> + <gv, q2> ; dv_i = a_i*q1*gsum + asum*q2*g_i
template <int NT>
__global__ __launch_bounds__(NT) void pairwise_bwd_kernel(const float* __restrict__ v, const float* __restrict__ q1,
                                                          const float* __restrict__ q2, const float* __restrict__ alpha,
                                                          int astride, const float* __restrict__ g,
                                                          float* __restrict__ d_alpha, float* __restrict__ d_q1,
                                                          float* __restrict__ d_q2, float* __restrict__ d_v, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tile = reinterpret_cast<float4*>(smem);                  // [N][NT]  v tile
  float* alpha_s = reinterpret_cast<float*>(tile + (size_t)N * NT);  // [N]
  float* red_s = alpha_s + N;                                      // [N] cross-wave partial sums
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + tid) * 4;
  const bool active = d < D;
  for (int i = tid; i < N; i += NT) {
    alpha_s[i] = alpha[((size_t)b * N + i) * astride];
    red_s[i] = 0.f;
  }
  __syncthreads();
  const size_t base = (size_t)b * N * D + d;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 q1v = z, q2v = z, gsum = z, gv = z, pooled = z;
  float asum = 0.f;
  if (active) {
    q1v = ld4(q1 + (size_t)b * D + d);
    q2v = ld4(q2 + (size_t)b * D + d);
#pragma unroll 6
    for (int j = 0; j < N; ++j) {
      const float4 vj = ld4(v + base + (size_t)j * D);
      const float4 gj = ld4(g + base + (size_t)j * D);
      tile[j * NT + tid] = vj;
      gsum = add4(gsum, gj);
      gv = add4(gv, mul4(gj, vj));
      pooled = fma4(alpha_s[j], vj, pooled);
    }
  }
  for (int i = 0; i < N; ++i) asum += alpha_s[i];
  const float4 u = mul4(q1v, gsum);
  if (active) {
    st4(d_q1 + (size_t)b * D + d, mul4(pooled, gsum));
    st4(d_q2 + (size_t)b * D + d, scale4(asum, gv));
  }
  const float cpart = active ? dot4(gv, q2v) : 0.f;
  const int lane = tid & 63;
  for (int i = 0; i < N; ++i) {
    float p = active ? dot4(tile[i * NT + tid], u) + cpart : 0.f;
    p = wave_sum(p);
    if (lane == 0) atomicAdd(&red_s[i], p);  // LDS atomic, NT/64 adders
  }
  if (d_v != nullptr && active) {
    const float4 c2 = scale4(asum, q2v);
#pragma unroll 6
    for (int i = 0; i < N; ++i) {
      const float4 gi = ld4(g + base + (size_t)i * D);  // second touch of the tile just streamed: L2
      st4(d_v + base + (size_t)i * D, add4(scale4(alpha_s[i], u), mul4(c2, gi)));
    }
  }
  __syncthreads();
  for (int i = tid; i < N; i += NT) atomicAdd(&d_alpha[(size_t)b * N + i], red_s[i]);
}

// ---- register-tile variants (N <= kRegN, the reference's 36 regions) ------------------------------------------
// One wave per workgroup, lane = one float4 column, the N region rows of that column held in VGPRs (144 for N=36):
// no LDS, so occupancy is set by registers (3 waves per SIMD, 12 per CU, against 4 per CU for the 73 KB LDS tile)
// and every lane has N independent 16-byte loads in flight.  Loads are unconditional from clamped rows (a load
// under a branch would serialise on vmcnt(0), see gemm_f32_mfma.hpp); rows >= N are zeroed by a select.
constexpr int kRegN = 36;

__global__ __launch_bounds__(64) void pairwise_fwd_reg_kernel(const float* __restrict__ v, const float* __restrict__ q1,
                                                              const float* __restrict__ q2, const float* __restrict__ alpha,
                                                              int astride, float* __restrict__ v2, int N, int D) {
  const int b = blockIdx.y;
  const int d = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  const size_t base = (size_t)b * N * D + d;
  float4 r[kRegN];
#pragma unroll
  for (int i = 0; i < kRegN; ++i) r[i] = ld4(v + base + (size_t)min(i, N - 1) * D);
  const float4 q1v = ld4(q1 + (size_t)b * D + d), q2v = ld4(q2 + (size_t)b * D + d);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float asum = 0.f;
#pragma unroll
  for (int i = 0; i < kRegN; ++i) {
    const float a = i < N ? alpha[((size_t)b * N + i) * astride] : 0.f;  // wave-uniform address: scalar load
    s = fma4(a, r[i], s);
    asum += a;
  }
  const float4 s1 = mul4(q1v, s);
  const float4 c2 = scale4(asum, q2v);
#pragma unroll
  for (int j = 0; j < kRegN; ++j) {
    if (j < N)
      st4(v2 + base + (size_t)j * D, make_float4(fmaf(c2.x, r[j].x, s1.x), fmaf(c2.y, r[j].y, s1.y),
                                                 fmaf(c2.z, r[j].z, s1.z), fmaf(c2.w, r[j].w, s1.w)));
  }
}

// Backward, two-pass streaming form (any N): pass 1 streams (v_j, g_j) once and keeps only the three column
// accumulators (sum g, sum g*v, sum alpha*v), so the kernel runs at full occupancy; pass 2 re-reads the v rows this
// workgroup has just streamed (36 KB per wave, served by L2 / Infinity Cache, not HBM) for dalpha_i = <v_i, q1*sum g>.
// dalpha partials: wave64 shuffles -> LDS -> one float atomic per (workgroup, region).
template <int NT>
__global__ __launch_bounds__(NT) void pairwise_bwd_stream_kernel(const float* __restrict__ v, const float* __restrict__ q1,
                                                                 const float* __restrict__ q2,
                                                                 const float* __restrict__ alpha, int astride,
                                                                 const float* __restrict__ g, float* __restrict__ d_alpha,
                                                                 float* __restrict__ d_q1, float* __restrict__ d_q2,
                                                                 float* __restrict__ d_v, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);  // [N]
  float* red_s = alpha_s + N;                       // [N]
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + tid) * 4;
  const bool active = d < D;
  const int dc = active ? d : 0;
  for (int i = tid; i < N; i += NT) {
    alpha_s[i] = alpha[((size_t)b * N + i) * astride];
    red_s[i] = 0.f;
  }
  __syncthreads();
  const size_t base = (size_t)b * N * D + dc;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gsum = z, gv = z, pooled = z;
  float asum = 0.f;
#pragma unroll 6
  for (int j = 0; j < N; ++j) {
    const float4 vj = ld4(v + base + (size_t)j * D);
    const float4 gj = ld4(g + base + (size_t)j * D);
    const float a = alpha_s[j];
    gsum = add4(gsum, gj);
    gv = add4(gv, mul4(gj, vj));
    pooled = fma4(a, vj, pooled);
    asum += a;
  }
  const float4 q1v = ld4(q1 + (size_t)b * D + dc), q2v = ld4(q2 + (size_t)b * D + dc);
  const float4 u = active ? mul4(q1v, gsum) : z;
  if (active) {
    st4(d_q1 + (size_t)b * D + d, mul4(pooled, gsum));
    st4(d_q2 + (size_t)b * D + d, scale4(asum, gv));
  }
  const float cpart = active ? dot4(gv, q2v) : 0.f;
  const float4 c2 = scale4(asum, q2v);
  constexpr int RB = 6;  // rows per batch: RB independent 16-byte loads in flight per lane, then RB wave reductions
  for (int i0 = 0; i0 < N; i0 += RB) {
    float4 vi[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) vi[k] = ld4(v + base + (size_t)min(i0 + k, N - 1) * D);
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int i = i0 + k;
      if (i < N) {
        if (d_v != nullptr && active) {
          const float4 gi = ld4(g + base + (size_t)i * D);
          st4(d_v + base + (size_t)i * D, add4(scale4(alpha_s[i], u), mul4(c2, gi)));
        }
        const float p = wave_sum(dot4(vi[k], u) + cpart);
        if (lane == 0) atomicAdd(&red_s[i], p);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < N; i += NT) atomicAdd(&d_alpha[(size_t)b * N + i], red_s[i]);
}

static int pick_threads(int N) { return N <= 36 ? 128 : 64; }

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_pairwise_relation_reduce_fwd(const float* v, const float* q1, const float* q2, const float* alpha,
                                                int alpha_stride, float* v2, int B, int N, int D, int mode,
                                                vqa_stream_t stream) {
  VQA_REQUIRE(v && q1 && q2 && alpha && v2, VQA_E_BADARG, "pairwise_relation_reduce_fwd: null pointer");
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && alpha_stride > 0, VQA_E_BADARG,
              "pairwise_relation_reduce_fwd: bad sizes B=%d N=%d D=%d alpha_stride=%d", B, N, D, alpha_stride);
  VQA_REQUIRE(mode == 0 || mode == 1, VQA_E_BADARG, "pairwise_relation_reduce_fwd: mode must be 0 or 1, got %d", mode);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, 16) && aligned(q1, 16) && aligned(q2, 16) && aligned(v2, 16), VQA_E_UNSUPPORTED,
              "pairwise_relation_reduce_fwd: needs D %% 4 == 0 and 16-byte aligned tensors (D=%d)", D);
  VQA_REQUIRE(N <= 144, VQA_E_UNSUPPORTED, "pairwise_relation_reduce_fwd: N=%d exceeds the LDS tile limit 144", N);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "pairwise_relation_reduce_fwd: B=%d exceeds 65535", B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == 1 && N <= kRegN) {
    hipLaunchKernelGGL(pairwise_fwd_reg_kernel, dim3((D / 4 + 63) / 64, B), dim3(64), 0, s, v, q1, q2, alpha, alpha_stride,
                       v2, N, D);
    return check_launch("pairwise_relation_reduce_fwd");
  }
  const int nt = pick_threads(N);
  const size_t lds = (size_t)N * nt * 16 + (size_t)N * 4;
  dim3 grid((D / 4 + nt - 1) / nt, B);
  if (nt == 128) {
    VQA_ENSURE_LDS(pairwise_fwd_kernel<128>, lds);
    hipLaunchKernelGGL(pairwise_fwd_kernel<128>, grid, dim3(128), lds, s, v, q1, q2, alpha, alpha_stride, v2, N, D, mode);
  } else {
    VQA_ENSURE_LDS(pairwise_fwd_kernel<64>, lds);
    hipLaunchKernelGGL(pairwise_fwd_kernel<64>, grid, dim3(64), lds, s, v, q1, q2, alpha, alpha_stride, v2, N, D, mode);
  }
  return check_launch("pairwise_relation_reduce_fwd");
}

extern "C" int vqa_pairwise_relation_reduce_bwd(const float* v, const float* q1, const float* q2, const float* alpha,
                                                int alpha_stride, const float* g_v2, float* d_alpha, float* d_q1,
                                                float* d_q2, float* d_v, int B, int N, int D, vqa_stream_t stream) {
  VQA_REQUIRE(v && q1 && q2 && alpha && g_v2 && d_alpha && d_q1 && d_q2, VQA_E_BADARG,
              "pairwise_relation_reduce_bwd: null pointer");
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && alpha_stride > 0, VQA_E_BADARG,
              "pairwise_relation_reduce_bwd: bad sizes B=%d N=%d D=%d alpha_stride=%d", B, N, D, alpha_stride);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, 16) && aligned(q1, 16) && aligned(q2, 16) && aligned(g_v2, 16) &&
                  aligned(d_q1, 16) && aligned(d_q2, 16) && (d_v == nullptr || aligned(d_v, 16)),
              VQA_E_UNSUPPORTED, "pairwise_relation_reduce_bwd: needs D %% 4 == 0 and 16-byte aligned tensors (D=%d)", D);
  VQA_REQUIRE(N <= 4096, VQA_E_UNSUPPORTED, "pairwise_relation_reduce_bwd: N=%d exceeds 4096", N);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "pairwise_relation_reduce_bwd: B=%d exceeds 65535", B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  hipError_t e = hipMemsetAsync(d_alpha, 0, (size_t)B * N * sizeof(float), s);
  if (e != hipSuccess) return fail(VQA_E_LAUNCH, "pairwise_relation_reduce_bwd: memset: %s", hipGetErrorString(e));
  {
    constexpr int NT = 256;
    hipLaunchKernelGGL(pairwise_bwd_stream_kernel<NT>, dim3((D / 4 + NT - 1) / NT, B), dim3(NT), (size_t)N * 8, s, v, q1, q2,
                       alpha, alpha_stride, g_v2, d_alpha, d_q1, d_q2, d_v, N, D);
    if (true) return check_launch("pairwise_relation_reduce_bwd");
  }
  const int nt = pick_threads(N);
  const size_t lds = (size_t)N * nt * 16 + (size_t)N * 8;
  dim3 grid((D / 4 + nt - 1) / nt, B);
  if (nt == 128) {
    VQA_ENSURE_LDS(pairwise_bwd_kernel<128>, lds);
    hipLaunchKernelGGL(pairwise_bwd_kernel<128>, grid, dim3(128), lds, s, v, q1, q2, alpha, alpha_stride, g_v2, d_alpha,
                       d_q1, d_q2, d_v, N, D);
  } else {
    VQA_ENSURE_LDS(pairwise_bwd_kernel<64>, lds);
    hipLaunchKernelGGL(pairwise_bwd_kernel<64>, grid, dim3(64), lds, s, v, q1, q2, alpha, alpha_stride, g_v2, d_alpha, d_q1,
                       d_q2, d_v, N, D);
  }
  return check_launch("pairwise_relation_reduce_bwd");
}
