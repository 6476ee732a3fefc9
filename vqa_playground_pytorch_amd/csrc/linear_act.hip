// K5 -- fused dropout + linear + bias + activation on the fp32 MFMA tile engine.
//
// Replaces MyConv1d.forward with kernel_size 1 (config/CoR2.py:72-88: F.dropout, transpose, nn.Conv1d, transpose,
// F.relu) for `compress_v` / `compress_v2` (config/CoR2.py:168-169,213,218): the largest contraction of the model
// ([B*36, 2048] x [2048, 310], 23.4 GFLOP at B=512, 91.4 of the 171.9 MFLOP per sample).  The reference (and a
// library GEMM) make three extra passes over the [B,36,2048] activations: the Bernoulli mask + multiply before the
// GEMM and bias/relu after it.  Here the mask is drawn from a counter hash while the A tile is staged (never
// stored; backward regenerates it), bias + relu are the epilogue, and the relu mask of backward is read off the
// saved output.
//
//   forward : y  = act( drop(x) W^T + b )                      2*M*K*N FLOP
//   backward: gz = gy * act'(y);  dW = gz^T drop(x)  (split over M, slabs reduced in fixed order; db = colsum gz)
//             dx = (gz W) * dropmask                           2*M*K*N FLOP each
#include "gemm_f32_mfma.hpp"
#include "gemm_f32_rt.hpp"

namespace vqa {

// A[m][k] = x[m][k] * dropmask(m,k)   K-contiguous
template <bool DROP>
struct SrcDropKC {
  using Raw = float2;
  const float* x;
  int ld, M, K;
  DropCfg dc;
  __device__ __forceinline__ Raw fetch(int m, int k) const { return ld2(x + (size_t)min(m, M - 1) * ld + min(k, K - 2)); }
  __device__ __forceinline__ float2 finish(Raw v, int m, int k) const {
    if (DROP) {
      const float2 s = drop_pair((uint32_t)m * (uint32_t)K + (uint32_t)k, dc);
      v = make_float2(v.x * s.x, v.y * s.y);
    }
    return keep_if(m < M && k < K, v);
  }
};
// B[k = m][mn = kk] = x[m][kk] * dropmask(m,kk)   MN-contiguous view of the same matrix, rows m < m_hi
template <bool DROP>
struct SrcDropMC {
  using Raw = float2;
  const float* x;
  int ld, K, m_hi;
  DropCfg dc;
  __device__ __forceinline__ Raw fetch(int kk, int m) const { return ld2(x + (size_t)min(m, m_hi - 1) * ld + min(kk, K - 2)); }
  __device__ __forceinline__ float2 finish(Raw v, int kk, int m) const {
    if (DROP) {
      const float2 s = drop_pair((uint32_t)m * (uint32_t)K + (uint32_t)kk, dc);
      v = make_float2(v.x * s.x, v.y * s.y);
    }
    return keep_if(m < m_hi && kk < K, v);
  }
};
struct GzRaw {
  float2 g, y;
};
__device__ __forceinline__ float2 gz_of(GzRaw v, int act) {
  if (act == 1) return make_float2(v.y.x > 0.f ? v.g.x : 0.f, v.y.y > 0.f ? v.g.y : 0.f);
  return v.g;
}
// A[m][k = n] = gz[m][n]   K-contiguous (dx GEMM)
struct SrcGzKC {
  using Raw = GzRaw;
  const float* gy;
  const float* y;
  int M, N, act;
  __device__ __forceinline__ Raw fetch(int m, int n) const {
    const size_t o = (size_t)min(m, M - 1) * N + min(n, N - 2);
    return Raw{ld2(gy + o), ld2(y + o)};
  }
  __device__ __forceinline__ float2 finish(Raw v, int m, int n) const { return keep_if(m < M && n < N, gz_of(v, act)); }
};
// A[k = m][mn = n] = gz[m][n]   MN-contiguous (dW GEMM), rows m < m_hi
struct SrcGzMC {
  using Raw = GzRaw;
  const float* gy;
  const float* y;
  int m_hi, N, act;
  __device__ __forceinline__ Raw fetch(int n, int m) const {
    const size_t o = (size_t)min(m, m_hi - 1) * N + min(n, N - 2);
    return Raw{ld2(gy + o), ld2(y + o)};
  }
  __device__ __forceinline__ float2 finish(Raw v, int n, int m) const { return keep_if(m < m_hi && n < N, gz_of(v, act)); }
};

template <int BM, int BN, int PF, bool DROP>
__global__ __launch_bounds__(kGemmThreads) void linear_fwd_kernel(const float* __restrict__ x, int ldx,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y,
                                                                  int M, int K, int N, int act, DropCfg dc, int tiles_n) {
  using T = GemmTile<BM, BN, 16, true, true>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  f32x16 acc[T::TM][T::TN];
  zero_acc(acc);
  const SrcDropKC<DROP> sa{x, ldx, M, K, dc};
  const SrcKC sb{w, K, N, K};
  gemm_tile<BM, BN, 16, PF, true, true>(sa, sb, m0, n0, 0, K, smem, acc);
  const AccCoord<BM, BN> cc(m0, n0);
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    const int col = cc.col(tn);
    const float bv = bias != nullptr ? bias[min(col, N - 1)] : 0.f;
    if (col < N) {
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          float v = acc[tm][tn][i] + bv;
          if (act == 1) v = fmaxf(v, 0.f);
          if (row < M) y[(size_t)row * N + col] = v;
        }
    }
  }
}

// dx[m][kk] = dropmask(m,kk) * sum_n gz[m][n] W[n][kk]
template <int BM, int BN, int PF, bool DROP>
__global__ __launch_bounds__(kGemmThreads) void linear_dx_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                 const float* __restrict__ w, float* __restrict__ dx,
                                                                 int M, int K, int N, int act, DropCfg dc, int tiles_n) {
  using T = GemmTile<BM, BN, 16, true, false>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  f32x16 acc[T::TM][T::TN];
  zero_acc(acc);
  const SrcGzKC sa{gy, y, M, N, act};
  const SrcMC sb{w, K, K, N};
  gemm_tile<BM, BN, 16, PF, true, false>(sa, sb, m0, n0, 0, N, smem, acc);
  const AccCoord<BM, BN> cc(m0, n0);
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    const int col = cc.col(tn);
    if (col < K) {
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          if (row < M) {
            float v = acc[tm][tn][i];
            if (DROP) {
              v *= drop_one((uint32_t)row * (uint32_t)K + (uint32_t)col, dc);
            }
            dx[(size_t)row * K + col] = v;
          }
        }
    }
  }
}

// slab[s][n][kk] = sum_{m in split s} gz[m][n] * drop(x)[m][kk];  dbslab[s][n] = sum_m gz[m][n]
template <int BM, int BN, int PF, bool DROP>
__global__ __launch_bounds__(kGemmThreads) void linear_dw_kernel(const float* __restrict__ gy, const float* __restrict__ y,
                                                                 const float* __restrict__ x, int ldx,
                                                                 float* __restrict__ slab, float* __restrict__ dbslab,
                                                                 int M, int K, int N, int act, DropCfg dc, int tiles_m,
                                                                 int tiles_n, int rows_per_split) {
  using T = GemmTile<BM, BN, 16, false, false>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  // 1-D grid over (split, k tile, n tile), n fastest, remapped so that every XCD gets a contiguous run: the workgroups
  // that share a row slab -- and, next to each other, the ones that share an x tile -- run on one XCD's L2 at the same
  // time (PMC: 405 MB fetched for 174 MB of operands with the (tile, split) grid whose neighbours shared nothing big)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = tiles_m * tiles_n;
  const int s = bid / tiles, tile = bid % tiles;
  const int n0 = (tile % tiles_m) * BM, k0 = (tile / tiles_m) * BN;
  const int m_lo = s * rows_per_split, m_hi = min(M, m_lo + rows_per_split);
  f32x16 acc[T::TM][T::TN];
  zero_acc(acc);
  float colsum[T::TM];
#pragma unroll
  for (int i = 0; i < T::TM; ++i) colsum[i] = 0.f;
  const SrcGzMC sa{gy, y, max(m_hi, 1), N, act};
  const SrcDropMC<DROP> sb{x, ldx, K, max(m_hi, 1), dc};
  gemm_tile<BM, BN, 16, PF, false, false>(sa, sb, n0, k0, m_lo, m_hi, smem, acc, colsum);
  if (k0 == 0 && (threadIdx.x >> 6 & 1) == 0) {
    const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      const float t = colsum[i] + __shfl_xor(colsum[i], 32, 64);
      const int n = n0 + wm * (T::TM * 32) + i * 32 + (lane & 31);
      if (lane < 32 && n < N) dbslab[(size_t)s * N + n] = t;
    }
  }
  float* __restrict__ dst = slab + (size_t)s * N * K;
  const AccCoord<BM, BN> cc(n0, k0);
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    const int col = cc.col(tn);
    if (col < K) {
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          if (row < N) dst[(size_t)row * K + col] = acc[tm][tn][i];
        }
    }
  }
}

// d_w[e] = sum_s slab[s][e] (e over N*K, float2 lanes);  d_b[n] = sum_s dbslab[s][n]   (fixed order)
__global__ __launch_bounds__(256) void linear_dw_reduce_kernel(const float* __restrict__ slab,
                                                               const float* __restrict__ dbslab, float* __restrict__ d_w,
                                                               float* __restrict__ d_b, int NK, int N, int S) {
  const int e = (blockIdx.x * 256 + threadIdx.x) * 2;
  if (d_b != nullptr && e < N) {
    float2 a = make_float2(0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float2 t = ld2(dbslab + (size_t)s * N + e);
      a.x += t.x;
      a.y += t.y;
    }
    st2(d_b + e, a);
  }
  if (e >= NK) return;
  float2 a = make_float2(0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const float2 t = ld2(slab + (size_t)s * NK + e);
    a.x += t.x;
    a.y += t.y;
  }
  st2(d_w + e, a);
}

static int splits_for_linear_dw(int M, int K, int N, TileChoice t) {
  const long tiles = (long)((N + t.bm - 1) / t.bm) * ((K + t.bn - 1) / t.bn);
  long s = (2560 + tiles - 1) / tiles;  // 10 workgroups per CU (sweep 4..24 at N x K = 310 x 2048, M = 18432: 16 splits, 256 us
                                        // with the slab reduction against 267 at 8; the curve is flat from 12 to 24)
  if (const char* e = vqa::option("VQA_LINEAR_DW_SPLITS")) s = std::atol(e);  // experiment knob
  const long max_by_rows = (M + 255) / 256;
  if (s > max_by_rows) s = max_by_rows;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return (int)s;
}
static TileChoice linear_dw_tile() {
  TileChoice t = tile_override_or({64, 64, 2});
  if (const char* e = vqa::option("VQA_LINEAR_DW_TILE")) {  // experiment knob
    int bm = 0, bn = 0, pf = 2;
    const int got = std::sscanf(e, "%dx%dx%d", &bm, &bn, &pf);
    if (got >= 2 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128) && pf >= 1 && pf <= 3) t = {bm, bn, pf};
  }
  return t;
}

// ---- register-tile engine (gemm_f32_rt.hpp): the default for the tall region projections ----
struct EpiBiasAct {
  float* y;
  const float* bias;
  int ldy, act;
  float scale;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    v = fmaf(v, scale, bias != nullptr ? bias[col] : 0.f);      // scale: 2 when the operand was masked unscaled (p = 0.5)
    if (act == 1) v = fmaxf(v, 0.f);
    y[(size_t)row * ldy + col] = v;
  }
};

// "0" = never, "1" (default) = where it applies (tall matrices with 4-element-aligned rows; everything else stays on
// the 64 x 64 LDS-tile engine above)
static bool rt_enabled() {
  const char* e = vqa::option("VQA_RT_ENGINE");
  return e == nullptr || e[0] != '0';
}
// p8: 0 (no dropout) or 128 (p = 0.5, the one-bit mask; needs rows of whole 32-element hash words); other rates stay on
// the LDS-tile engine
static bool rt_fwd_ok(int M, int K, int N, int ldx, uint32_t p8) {
  return rt_enabled() && M >= 1152 && K >= 64 && K % 4 == 0 && ldx % 4 == 0 && (size_t)M * ldx * 4 < (1ull << 32) &&
         (size_t)N * K * 4 < (1ull << 32) && (p8 == 0 || (p8 == kDropHalf && K % 32 == 0));
}
static int rt_dw_splits(int M) {
  int s = 16;   // 4 n1 tiles x 16 n2 tiles x 16 row splits = 1024 waves for 310 x 2048 (one per SIMD)
  if (const char* e = vqa::option("VQA_RT_DW_SPLITS")) s = std::atoi(e);
  const int max_by_rows = M / 64 > 0 ? M / 64 : 1;
  if (s > max_by_rows) s = max_by_rows;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return s;
}
static bool rt_dw_ok(int M, int K, int N, int ldx, uint32_t p8) {
  return rt_enabled() && M >= 4096 && K % 4 == 0 && ldx % 4 == 0 && (size_t)M * ldx * 4 < (1ull << 32) &&
         (size_t)M * N * 4 < (1ull << 32) && (size_t)M * K < (1ull << 32) && (p8 == 0 || p8 == kDropHalf);
}

}  // namespace vqa

using namespace vqa;

static int linear_check(const char* who, const void* x, int ldx, int M, int K, int N, int act, float p) {
  VQA_REQUIRE(M > 0 && K > 0 && N > 0, VQA_E_BADARG, "%s: bad sizes M=%d K=%d N=%d", who, M, K, N);
  VQA_REQUIRE(K % 2 == 0 && N % 2 == 0 && ldx % 2 == 0 && ldx >= K, VQA_E_UNSUPPORTED,
              "%s: needs even K, N, ldx and ldx >= K (K=%d N=%d ldx=%d)", who, K, N, ldx);
  VQA_REQUIRE(act == 0 || act == 1, VQA_E_BADARG, "%s: act must be 0 (none) or 1 (relu), got %d", who, act);
  VQA_REQUIRE(p >= 0.f && p < 1.f, VQA_E_BADARG, "%s: p_drop=%f outside [0,1)", who, (double)p);
  VQA_REQUIRE(aligned(x, 8), VQA_E_UNSUPPORTED, "%s: x must be 8-byte aligned", who);
  VQA_REQUIRE((long)M * (ldx > N ? ldx : N) < (1L << 32), VQA_E_UNSUPPORTED, "%s: M*K exceeds 2^32 elements", who);
  return VQA_OK;
}

extern "C" int vqa_linear_act_fwd(const float* x, int ldx, const float* w, const float* bias, float* y, int M, int K,
                                  int N, int act, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                  vqa_stream_t stream) {
  VQA_REQUIRE(x && w && y, VQA_E_BADARG, "linear_act_fwd: null pointer");
  int rc = linear_check("linear_act_fwd", x, ldx, M, K, N, act, p_drop);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(aligned(w, 8) && aligned(y, 8), VQA_E_UNSUPPORTED, "linear_act_fwd: w/y must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  if (rt_fwd_ok(M, K, N, ldx, dc.p8)) {
    // 144 x 160 workgroup tiles, each wave 9 x 5 accumulator blocks over half of K (M = 18432, N = 310: 256 workgroups)
    using S = rt::NtShape<9, 5, 1, 2, 2>;
    const int tiles_m = (M + S::BM - 1) / S::BM, tiles_n = (N + S::BN - 1) / S::BN;
    const rt::NtArgs a{x, w, ldx, K, M, N, K, tiles_n, nullptr};
    const EpiBiasAct epi{y, bias, N, act, dc.p8 > 0 ? dc.scale : 1.f};
    if (dc.p8 > 0) {
      VQA_ENSURE_LDS((rt::gemm_nt_kernel<9, 5, 1, 2, 2, true, EpiBiasAct>), S::kLdsBytes);
      VQA_LAUNCH((rt::gemm_nt_kernel<9, 5, 1, 2, 2, true, EpiBiasAct>), dim3(tiles_m * tiles_n), dim3(rt::kThreads),
                         S::kLdsBytes, s, a, dc, epi);
    } else {
      VQA_ENSURE_LDS((rt::gemm_nt_kernel<9, 5, 1, 2, 2, false, EpiBiasAct>), S::kLdsBytes);
      VQA_LAUNCH((rt::gemm_nt_kernel<9, 5, 1, 2, 2, false, EpiBiasAct>), dim3(tiles_m * tiles_n), dim3(rt::kThreads),
                         S::kLdsBytes, s, a, dc, epi);
    }
    return check_launch("linear_act_fwd");
  }
  const TileChoice t = tile_override_or(choose_tile(M, N, 1));
  const int tiles_m = (M + t.bm - 1) / t.bm, tiles_n = (N + t.bn - 1) / t.bn;
#define LAUNCH(BM_, BN_, PF_)                                                                                           \
  {                                                                                                                     \
    const size_t lds = GemmTile<BM_, BN_, 16, true, true>::kSmemBytes;                                                  \
    if (dc.p8 > 0) {                                                                                                    \
      VQA_ENSURE_LDS((linear_fwd_kernel<BM_, BN_, PF_, true>), lds);                                                    \
      VQA_LAUNCH((linear_fwd_kernel<BM_, BN_, PF_, true>), dim3(tiles_m * tiles_n), dim3(kGemmThreads), lds, s,  \
                         x, ldx, w, bias, y, M, K, N, act, dc, tiles_n);                                                \
    } else {                                                                                                            \
      VQA_ENSURE_LDS((linear_fwd_kernel<BM_, BN_, PF_, false>), lds);                                                   \
      VQA_LAUNCH((linear_fwd_kernel<BM_, BN_, PF_, false>), dim3(tiles_m * tiles_n), dim3(kGemmThreads), lds, s, \
                         x, ldx, w, bias, y, M, K, N, act, dc, tiles_n);                                                \
    }                                                                                                                   \
  }
  VQA_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
  return check_launch("linear_act_fwd");
}

extern "C" size_t vqa_linear_act_bwd_workspace_bytes(int M, int K, int N) {
  if (M <= 0 || K <= 0 || N <= 0) return 0;
  int S = splits_for_linear_dw(M, K, N, linear_dw_tile());
  if (rt_enabled() && M >= 4096 && rt_dw_splits(M) > S) S = rt_dw_splits(M);   // either engine fits
  return ((size_t)S * N * K + (size_t)S * N) * sizeof(float);
}

extern "C" int vqa_linear_act_bwd(const float* x, int ldx, const float* w, const float* y, const float* gy, float* d_x,
                                  float* d_w, float* d_b, void* workspace, size_t workspace_bytes, int M, int K, int N,
                                  int act, float p_drop, uint64_t seed, const uint64_t* seed_ptr, vqa_stream_t stream) {
  VQA_REQUIRE(x && w && y && gy && d_w && workspace, VQA_E_BADARG, "linear_act_bwd: null pointer");
  int rc = linear_check("linear_act_bwd", x, ldx, M, K, N, act, p_drop);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(workspace_bytes >= vqa_linear_act_bwd_workspace_bytes(M, K, N), VQA_E_BADARG,
              "linear_act_bwd: workspace of %zu B is too small", workspace_bytes);
  VQA_REQUIRE(aligned(w, 8) && aligned(y, 8) && aligned(gy, 8) && aligned(d_w, 8) && aligned(workspace, 16) &&
                  (d_x == nullptr || aligned(d_x, 8)) && (d_b == nullptr || aligned(d_b, 8)),
              VQA_E_UNSUPPORTED, "linear_act_bwd: tensors must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  if (d_x != nullptr) {
    const TileChoice t = tile_override_or(choose_tile(M, K, 1));
    const int tiles_m = (M + t.bm - 1) / t.bm, tiles_n = (K + t.bn - 1) / t.bn;
#define LAUNCH(BM_, BN_, PF_)                                                                                          \
  {                                                                                                                    \
    const size_t lds = GemmTile<BM_, BN_, 16, true, false>::kSmemBytes;                                                \
    if (dc.p8 > 0) {                                                                                                   \
      VQA_ENSURE_LDS((linear_dx_kernel<BM_, BN_, PF_, true>), lds);                                                    \
      VQA_LAUNCH((linear_dx_kernel<BM_, BN_, PF_, true>), dim3(tiles_m * tiles_n), dim3(kGemmThreads), lds, s,  \
                         gy, y, w, d_x, M, K, N, act, dc, tiles_n);                                                    \
    } else {                                                                                                           \
      VQA_ENSURE_LDS((linear_dx_kernel<BM_, BN_, PF_, false>), lds);                                                   \
      VQA_LAUNCH((linear_dx_kernel<BM_, BN_, PF_, false>), dim3(tiles_m * tiles_n), dim3(kGemmThreads), lds, s, \
                         gy, y, w, d_x, M, K, N, act, dc, tiles_n);                                                    \
    }                                                                                                                  \
  }
    VQA_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
  }
  if (rt_dw_ok(M, K, N, ldx, dc.p8)) {
    // register-tile TN form: workgroup = 4 waves x (80 rows of d_w) x 128 columns x one row split; the relu gate (y > 0)
    // and the dropout mask of x are applied to the operands in registers
    const int S = rt_dw_splits(M);
    float* slab = static_cast<float*>(workspace);
    float* dbslab = slab + (size_t)S * N * K;
    int rows_per_split = (M + S - 1) / S;
    rows_per_split = (rows_per_split + 15) / 16 * 16;
    const rt::TnArgs a{gy, act == 1 ? y : nullptr, x, slab, dbslab, N, ldx, M, N, K, (N + 319) / 320, (K + 127) / 128,
                       rows_per_split};
    const dim3 grid(a.tiles1 * a.tiles2 * S);
    const bool gate = act == 1, drop = dc.p8 > 0;
    // (the dropout mask's hash words shared within groups of eight lanes when the rows are whole 64-column spans)
    const bool share = drop && K % 64 == 0 && !vqa::option_is("VQA_RT_SHARE_HASH", '0');
    if (gate && drop && share)      // (with the gate the VALU side of a chunk goes in one burst: 195.0 -> 192.9 us, tools/rt_probe.hip)
      VQA_LAUNCH((rt::gemm_tn_kernel<5, 2, true, true, true, 1>), grid, dim3(rt::kThreads), 0, s, a, dc);
    else if (drop && share)
      VQA_LAUNCH((rt::gemm_tn_kernel<5, 2, false, true, true>), grid, dim3(rt::kThreads), 0, s, a, dc);
    else if (gate && drop)
      VQA_LAUNCH((rt::gemm_tn_kernel<5, 2, true, true>), grid, dim3(rt::kThreads), 0, s, a, dc);
    else if (gate)
      VQA_LAUNCH((rt::gemm_tn_kernel<5, 2, true, false>), grid, dim3(rt::kThreads), 0, s, a, dc);
    else if (drop)
      VQA_LAUNCH((rt::gemm_tn_kernel<5, 2, false, true>), grid, dim3(rt::kThreads), 0, s, a, dc);
    else
      VQA_LAUNCH((rt::gemm_tn_kernel<5, 2, false, false>), grid, dim3(rt::kThreads), 0, s, a, dc);
    const int NK = N * K;
    VQA_LAUNCH(linear_dw_reduce_kernel, dim3((NK / 2 + 255) / 256), dim3(256), 0, s, slab, dbslab, d_w, d_b, NK, N, S);
  } else {
    const TileChoice t = linear_dw_tile();
    const int S = splits_for_linear_dw(M, K, N, t);
    float* slab = static_cast<float*>(workspace);
    float* dbslab = slab + (size_t)S * N * K;
    const int tiles_m = (N + t.bm - 1) / t.bm, tiles_n = (K + t.bn - 1) / t.bn;
    int rows_per_split = (M + S - 1) / S;
    rows_per_split = (rows_per_split + 15) / 16 * 16;
#define LAUNCH(BM_, BN_, PF_)                                                                                            \
  {                                                                                                                      \
    const size_t lds = GemmTile<BM_, BN_, 16, false, false>::kSmemBytes;                                                 \
    if (dc.p8 > 0) {                                                                                                     \
      VQA_ENSURE_LDS((linear_dw_kernel<BM_, BN_, PF_, true>), lds);                                                      \
      VQA_LAUNCH((linear_dw_kernel<BM_, BN_, PF_, true>), dim3(tiles_m * tiles_n * S), dim3(kGemmThreads), lds, s, \
                         gy, y, x, ldx, slab, dbslab, M, K, N, act, dc, tiles_m, tiles_n, rows_per_split);                \
    } else {                                                                                                             \
      VQA_ENSURE_LDS((linear_dw_kernel<BM_, BN_, PF_, false>), lds);                                                     \
      VQA_LAUNCH((linear_dw_kernel<BM_, BN_, PF_, false>), dim3(tiles_m * tiles_n * S), dim3(kGemmThreads), lds, s, \
                         gy, y, x, ldx, slab, dbslab, M, K, N, act, dc, tiles_m, tiles_n, rows_per_split);                \
    }                                                                                                                    \
  }
    VQA_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
    const int NK = N * K;
    VQA_LAUNCH(linear_dw_reduce_kernel, dim3((NK / 2 + 255) / 256), dim3(256), 0, s, slab, dbslab, d_w, d_b, NK, N, S);
  }
  return check_launch("linear_act_bwd");
}

// fp32 [M,K] keep/(1-p) mask exactly as the fused kernels draw it (test / debugging aid)
__global__ __launch_bounds__(256) void linear_mask_kernel(float* __restrict__ mask, size_t n, DropCfg dc) {
  const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
  if (e >= n) return;
  const float2 m = dc.p8 > 0 ? drop_pair((uint32_t)e, dc) : make_float2(1.f, 1.f);
  mask[e] = m.x;
  if (e + 1 < n) mask[e + 1] = m.y;
}

extern "C" int vqa_linear_dropout_mask(float* mask, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int M, int K,
                                       vqa_stream_t stream) {
  VQA_REQUIRE(mask && M > 0 && K > 0 && K % 2 == 0, VQA_E_BADARG, "linear_dropout_mask: bad arguments");
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "linear_dropout_mask: p_drop=%f outside [0,1)", (double)p_drop);
  const size_t n = (size_t)M * K;
  VQA_LAUNCH(linear_mask_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     mask, n, make_drop(p_drop, seed, seed_ptr));
  return check_launch("linear_dropout_mask");
}
