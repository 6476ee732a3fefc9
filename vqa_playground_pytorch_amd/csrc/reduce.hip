// Column sums of a row-major matrix: out[n] = sum_m x[m, n] -- the bias gradient of every nn.Linear / 1x1 nn.Conv1d on
// the path (grad_output.sum(0) in torch.autograd's linear backward; putils.Linear putils/__init__.py:16-33, MyLinear /
// MyConv1d config/CoR2.py:56-122).
//
// Why not torch's sum: for tall matrices (M = B*N = 18432 rows) torch's reduce_kernel splits the rows over workgroups
// that meet through semaphores zeroed by a memset -- and a memset node replays wrongly inside a hipGraph on ROCm 7.2
// (see api.hip), so graph-replayed steps got corrupt bias gradients.  Here: grid (column blocks, row slabs); a
// workgroup = 4 waves = 4 interleaved row slices of one slab, every lane VEC adjacent columns, 4 independent rows in
// flight per lane; slices meet in LDS; slabs are written to a workspace and added in a fixed order by a second tiny
// kernel (bitwise reproducible, no atomics, no memset).  HBM-bound: reads x once.
#include "common.hpp"

namespace vqa {

template <typename T, int VEC>
struct ColVec;
template <>
struct ColVec<float, 1> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[1]) { v[0] = p[0]; }
};
template <>
struct ColVec<float, 2> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[2]) {
    const float2 t = ld2(p);
    v[0] = t.x;
    v[1] = t.y;
  }
};
template <>
struct ColVec<float, 4> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const float4 t = ld4(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
};
template <>
struct ColVec<bf16, 1> {
  static __device__ __forceinline__ void ld(const bf16* p, float (&v)[1]) { v[0] = (float)p[0]; }
};
template <>
struct ColVec<bf16, 2> {
  static __device__ __forceinline__ void ld(const bf16* p, float (&v)[2]) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(p);
    v[0] = bf16_lo(w);
    v[1] = bf16_hi(w);
  }
};
template <>
struct ColVec<bf16, 4> {
  static __device__ __forceinline__ void ld(const bf16* p, float (&v)[4]) {
    const float4 t = ld4(p);
    v[0] = t.x;
    v[1] = t.y;
    v[2] = t.z;
    v[3] = t.w;
  }
};

constexpr int kColThreads = 256;

template <typename T, int VEC>
__global__ __launch_bounds__(kColThreads) void column_sum_kernel(const T* __restrict__ x, int ld, float* __restrict__ dst,
                                                                 int M, int N, int rows_per_slab) {
  __shared__ float part[3][64 * VEC];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int n = (blockIdx.x * 64 + lane) * VEC;
  const bool active = n < N;
  const int nc = active ? n : 0;
  const int m_lo = blockIdx.y * rows_per_slab, m_hi = min(M, m_lo + rows_per_slab);
  float acc[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
  const T* col = x + nc;
  for (int m0 = m_lo + slice; m0 < m_hi; m0 += 16) {  // 4 slices x 4 rows in flight
    float v[4][VEC];
#pragma unroll
    for (int k = 0; k < 4; ++k) ColVec<T, VEC>::ld(col + (size_t)min(m0 + 4 * k, m_hi - 1) * ld, v[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (m0 + 4 * k < m_hi) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] += v[k][e];
      }
  }
  if (slice > 0) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) part[slice - 1][lane * VEC + e] = acc[e];
  }
  __syncthreads();
  if (slice == 0 && active) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = acc[e];
#pragma unroll
      for (int s = 0; s < 3; ++s) t += part[s][lane * VEC + e];
      dst[(size_t)blockIdx.y * N + n + e] = t;
    }
  }
}

// Short matrices (M <= kColShortM: the [B, .] question-side layers): ONE kernel, no workspace.  A workgroup = 16
// adjacent columns x 16 row slices, 8 rows in flight per lane (the matrix was just written: L2 hits), slices meet in
// LDS in a fixed order.
constexpr int kColShortM = 1024;
template <typename T>
__global__ __launch_bounds__(256) void column_sum_short_kernel(const T* __restrict__ x, int ld, float* __restrict__ out, int M,
                                                               int N) {
  __shared__ float part[16][17];
  const int c = threadIdx.x & 15, slice = threadIdx.x >> 4;
  const int n = blockIdx.x * 16 + c;
  const int nc = min(n, N - 1);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int m0 = slice; m0 < M; m0 += 128) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (float)x[(size_t)min(m0 + 16 * k, M - 1) * ld + nc];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (m0 + 16 * k < M) acc[k] += v[k];
  }
  part[slice][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (slice == 0 && n < N) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += part[q][c];
    out[n] = t;
  }
}

// out[n] = sum_s slabs[s][n], fixed order; 256 lanes = 64 columns x 4 slab slices, 4 slabs in flight per lane
__global__ __launch_bounds__(256) void column_sum_finish_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                int N, int S) {
  __shared__ float red_s[3][64];
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + c;
  const int nc = min(n, N - 1);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int q = slice;
  for (; q + 12 < S; q += 16) {
    a0 += slabs[(size_t)q * N + nc];
    a1 += slabs[(size_t)(q + 4) * N + nc];
    a2 += slabs[(size_t)(q + 8) * N + nc];
    a3 += slabs[(size_t)(q + 12) * N + nc];
  }
  for (; q < S; q += 4) a0 += slabs[(size_t)q * N + nc];
  const float a = (a0 + a1) + (a2 + a3);
  if (slice > 0) red_s[slice - 1][c] = a;
  __syncthreads();
  if (slice == 0 && n < N) out[n] = a + red_s[0][c] + red_s[1][c] + red_s[2][c];
}

static int column_slabs(int M, int N, int vec) {
  const int col_blocks = (N + 64 * vec - 1) / (64 * vec);
  int s = (512 + col_blocks - 1) / col_blocks;  // aim at >= 512 workgroups
  const int max_by_rows = (M + 63) / 64;        // keep >= 64 rows per slab
  if (s > max_by_rows) s = max_by_rows;
  if (s < 1) s = 1;
  return s;
}
template <typename T>
static int column_vec(const T* x, int ld, int N) {
  if (N % 4 == 0 && ld % 4 == 0 && aligned(x, 4 * sizeof(T))) return 4;
  if (N % 2 == 0 && ld % 2 == 0 && aligned(x, 2 * sizeof(T))) return 2;
  return 1;
}

template <typename T>
static int column_sum_impl(const char* who, const T* x, int ld, float* out, void* workspace, size_t workspace_bytes, int M,
                           int N, vqa_stream_t stream) {
  VQA_REQUIRE(x && out, VQA_E_BADARG, "%s: null pointer", who);
  VQA_REQUIRE(M > 0 && N > 0 && ld >= N, VQA_E_BADARG, "%s: bad sizes M=%d N=%d ld=%d", who, M, N, ld);
  if (M <= kColShortM) {
    VQA_LAUNCH(column_sum_short_kernel<T>, dim3((N + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld,
                       out, M, N);
    return check_launch(who);
  }
  const int vec = column_vec(x, ld, N);
  const int S = column_slabs(M, N, vec);
  const size_t need = S > 1 ? (size_t)S * N * sizeof(float) : 0;
  VQA_REQUIRE(workspace_bytes >= need && (need == 0 || workspace != nullptr), VQA_E_BADARG,
              "%s: workspace of %zu B is too small (needs %zu)", who, workspace_bytes, need);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rows_per_slab = (M + S - 1) / S;
  float* dst = S > 1 ? static_cast<float*>(workspace) : out;
  const dim3 grid((N + 64 * vec - 1) / (64 * vec), S);
  if (vec == 4) {
    VQA_LAUNCH((column_sum_kernel<T, 4>), grid, dim3(kColThreads), 0, s, x, ld, dst, M, N, rows_per_slab);
  } else if (vec == 2) {
    VQA_LAUNCH((column_sum_kernel<T, 2>), grid, dim3(kColThreads), 0, s, x, ld, dst, M, N, rows_per_slab);
  } else {
    VQA_LAUNCH((column_sum_kernel<T, 1>), grid, dim3(kColThreads), 0, s, x, ld, dst, M, N, rows_per_slab);
  }
  if (S > 1)
    VQA_LAUNCH(column_sum_finish_kernel, dim3((N + 63) / 64), dim3(256), 0, s, static_cast<const float*>(workspace),
                       out, N, S);
  return check_launch(who);
}

}  // namespace vqa

using namespace vqa;

extern "C" size_t vqa_column_sum_workspace_bytes(int M, int N) {
  if (M <= 0 || N <= 0 || M <= kColShortM) return 0;
  // the slab count depends on the vector width chosen at launch; size for the largest (vec = 1 gives the fewest column
  // blocks per row of workgroups, hence the most slabs -- bounded by the rows)
  int worst = 1;
  for (int vec = 1; vec <= 4; vec *= 2) worst = column_slabs(M, N, vec) > worst ? column_slabs(M, N, vec) : worst;
  return worst > 1 ? (size_t)worst * N * sizeof(float) : 0;
}

extern "C" int vqa_column_sum(const float* x, int ld, float* out, void* workspace, size_t workspace_bytes, int M, int N,
                              vqa_stream_t stream) {
  return column_sum_impl<float>("column_sum", x, ld, out, workspace, workspace_bytes, M, N, stream);
}

extern "C" int vqa_column_sum_bf16(const vqa_bf16_t* x, int ld, float* out, void* workspace, size_t workspace_bytes, int M,
                                   int N, vqa_stream_t stream) {
  return column_sum_impl<bf16>("column_sum_bf16", reinterpret_cast<const bf16*>(x), ld, out, workspace, workspace_bytes, M, N,
                               stream);
}
