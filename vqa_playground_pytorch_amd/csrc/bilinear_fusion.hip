// K4 -- low-rank bilinear (Mutan) fusion on the fp32 MFMA tile engine.
//
// Replaces putils.MutanFusion.forward (putils/__init__.py:232-238): per rank r a Linear(in1->H) on the
// region side, a Linear(in2->H) on the question side, putils.bmul (python loop over B + stack) and
// `total +=`.  The region-side contraction is dense (M = B*N rows, K = L, N = R*H) and is the one
// MFMA-bound kernel of the path; the question-side multiply and the sum over ranks are its epilogue,
// so the [M,R,H] intermediate is written at most once (only when backward needs it) and never re-read
// in forward.
//
//   forward : out[m,:] = sum_r (x[m,:] W1_r^T + b1_r) * h2[b(m),r,:]              2*M*L*H*R FLOP
//   backward: dx  = sum_r (g * h2_r) W1_r                                          2*M*H*L*R FLOP
//             dW1_r = (g * h2_r)^T x     (split over M, slabs reduced in fixed order) 2*M*H*L*R FLOP
//             dh2[b,r,:] = sum_n g[b,n,:] * h1[b,n,r,:] ;  db1_r = sum_b h2[b,r,:] * sum_n g[b,n,:]
// (g * h2_r) is never materialised: it is formed while the A tile is staged.
#include <cstdlib>

#include "bilinear_folded.hpp"
#include "gemm_f32_mfma.hpp"

namespace vqa {

constexpr int kMaxR = 8;

struct RankPtrs {
  const float* w[kMaxR];
  const float* b[kMaxR];
};
struct RankOutPtrs {
  float* w[kMaxR];
  float* b[kMaxR];
};

// ------------------------------------------------------------------------------------------ forward
template <int BM, int BN, int PF>
__global__ __launch_bounds__(kGemmThreads) void bilinear_fwd_kernel(const float* __restrict__ x, int ldx, RankPtrs rp,
                                                                    const float* __restrict__ h2,
                                                                    float* __restrict__ out, float* __restrict__ h1,
                                                                    int M, int N, int L, int H, int R, int tiles_n) {
  using T = GemmTile<BM, BN, 16, true, true>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  int* rowb_s = reinterpret_cast<int*>(smem + 2 * T::kStageFloats);  // [BM] sample index of each tile row
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  for (int t = threadIdx.x; t < BM; t += kGemmThreads) rowb_s[t] = min(m0 + t, M - 1) / N;
  // (visible after the first barrier inside gemm_tile)
  const AccCoord<BM, BN> cc(m0, n0);
  f32x16 total[T::TM][T::TN];
  zero_acc(total);
  for (int r = 0; r < R; ++r) {
    f32x16 acc[T::TM][T::TN];
    zero_acc(acc);
    const SrcKC sa{x, ldx, M, L};
    const SrcKC sb{rp.w[r], L, H, L};
    gemm_tile<BM, BN, 16, PF, true, true>(sa, sb, m0, n0, 0, L, smem, acc);
    const float* __restrict__ bias = rp.b[r];
    // epilogue of rank r: total += (acc + b1_r) * h2[b(row), r, :].  The h2 / bias loads are unconditional
    // (clamped) and issued as one batch per 32x32 tile, so they overlap instead of serialising behind branches.
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
      const int col = cc.col(tn);
      const int colc = min(col, H - 1);
      const float bv = bias[colc];
      const float* __restrict__ h2c = h2 + (size_t)r * H + colc;
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm) {
        float qv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) qv[i] = h2c[(size_t)rowb_s[min(cc.row(tm, i), M - 1) - m0] * (R * H)];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          const float hv = acc[tm][tn][i] + bv;
          if (h1 != nullptr && row < M && col < H) h1[((size_t)row * R + r) * H + col] = hv;
          total[tm][tn][i] = fmaf(hv, qv[i], total[tm][tn][i]);
        }
      }
    }
  }
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    const int col = cc.col(tn);
    if (col < H) {
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          if (row < M) out[(size_t)row * H + col] = total[tm][tn][i];
        }
    }
  }
}

// --------------------------------------------------------------------------------- backward sources
struct ScaledRaw {
  float2 a, s;
};
// sample index of row m: m / N by multiply-high with inv = floor(2^32 / N) + 1 (exact for m < 2^32 / N, checked on
// the host; an integer division costs ~25 VALU instructions, and the dW source needs one per staged slot per stage)
struct RowToSample {
  uint32_t inv;  // 0 when N == 1
  __device__ __forceinline__ int operator()(int m) const { return inv ? (int)__umulhi((uint32_t)m, inv) : m; }
};
static RowToSample make_row_to_sample(int N) { return RowToSample{N == 1 ? 0u : (uint32_t)((1ull << 32) / (uint32_t)N) + 1u}; }
// A[m][k=h] = g[m][h] * h2[b(m)][r][h]   (K-contiguous; the scale is applied while staging)
struct SrcScaledKC {
  using Raw = ScaledRaw;
  const float* g;
  const float* h2r;  // h2 + r*H
  int M, H, RH;
  RowToSample samp;
  __device__ __forceinline__ Raw fetch(int m, int h) const {
    const int mc = min(m, M - 1), hc = min(h, H - 2);
    return Raw{ld2(g + (size_t)mc * H + hc), ld2(h2r + (size_t)samp(mc) * RH + hc)};
  }
  __device__ __forceinline__ float2 finish(Raw v, int m, int h) const {
    return keep_if(m < M && h < H, make_float2(v.a.x * v.s.x, v.a.y * v.s.y));
  }
};
// A[k=m][mn=h] = g[m][h] * h2[b(m)][r][h]   (MN-contiguous view of the same matrix, rows m < m_hi)
struct SrcScaledMC {
  using Raw = ScaledRaw;
  const float* g;
  const float* h2r;
  int m_hi, H, RH;
  RowToSample samp;
  __device__ __forceinline__ Raw fetch(int h, int m) const {
    const int mc = min(m, m_hi - 1), hc = min(h, H - 2);
    return Raw{ld2(g + (size_t)mc * H + hc), ld2(h2r + (size_t)samp(mc) * RH + hc)};
  }
  __device__ __forceinline__ float2 finish(Raw v, int h, int m) const {
    return keep_if(m < m_hi && h < H, make_float2(v.a.x * v.s.x, v.a.y * v.s.y));
  }
};

// dx[m][l] = sum_r sum_h (g*h2_r)[m][h] * W1_r[h][l]
template <int BM, int BN, int PF>
__global__ __launch_bounds__(kGemmThreads) void bilinear_dx_kernel(const float* __restrict__ g, RankPtrs rp,
                                                                   const float* __restrict__ h2, float* __restrict__ dx,
                                                                   int M, int N, int L, int H, int R, int tiles_n,
                                                                   RowToSample samp) {
  using T = GemmTile<BM, BN, 16, true, false>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  f32x16 acc[T::TM][T::TN];
  zero_acc(acc);
  for (int r = 0; r < R; ++r) {
    const SrcScaledKC sa{g, h2 + (size_t)r * H, M, H, R * H, samp};
    const SrcMC sb{rp.w[r], L, L, H};
    gemm_tile<BM, BN, 16, PF, true, false>(sa, sb, m0, n0, 0, H, smem, acc);
  }
  const AccCoord<BM, BN> cc(m0, n0);
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    const int col = cc.col(tn);
    if (col < L) {
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          if (row < M) dx[(size_t)row * L + col] = acc[tm][tn][i];
        }
    }
  }
}

// slab[s][r][h][l] = sum_{m in split s} (g*h2_r)[m][h] * x[m][l]
template <int BM, int BN, int PF>
__global__ __launch_bounds__(kGemmThreads) void bilinear_dw_kernel(const float* __restrict__ g,
                                                                   const float* __restrict__ h2,
                                                                   const float* __restrict__ x, int ldx,
                                                                   float* __restrict__ slab, float* __restrict__ dbslab,
                                                                   int M, int N, int L, int H, int R, int tiles_m,
                                                                   int tiles_n, int rows_per_split, RowToSample samp) {
  using T = GemmTile<BM, BN, 16, false, false>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  // 1-D grid over (split, rank, tile), remapped so that every XCD gets a contiguous run: the R * tiles workgroups of a
  // split read the same rows of g and x (~3 MB) and now do so through one XCD's L2 (PMC: 190 MB fetched for 60 MB of
  // operands with the (tile, rank, split) grid, whose neighbours were dealt round-robin over the 8 XCDs)
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = tiles_m * tiles_n * R;
  const int s = bid / per_split, r = (bid % per_split) / (tiles_m * tiles_n), tile = bid % (tiles_m * tiles_n);
  const int h0 = (tile / tiles_n) * BM, l0 = (tile % tiles_n) * BN;
  const int m_lo = s * rows_per_split, m_hi = min(M, m_lo + rows_per_split);
  f32x16 acc[T::TM][T::TN];
  zero_acc(acc);
  const SrcScaledMC sa{g, h2 + (size_t)r * H, m_hi, H, R * H, samp};
  const SrcMC sb{x, ldx, L, m_hi};
  float colsum[T::TM];
#pragma unroll
  for (int i = 0; i < T::TM; ++i) colsum[i] = 0.f;
  gemm_tile<BM, BN, 16, PF, false, false>(sa, sb, h0, l0, m_lo, m_hi, smem, acc, colsum);
  // db1 partial of this split: column sums of the (g*h2_r) tile; written once per tile row (first tile column,
  // waves of the first wave column)
  if (l0 == 0 && (threadIdx.x >> 6 & 1) == 0) {
    const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
#pragma unroll
    for (int i = 0; i < T::TM; ++i) {
      const float t = colsum[i] + __shfl_xor(colsum[i], 32, 64);
      const int h = h0 + wm * (T::TM * 32) + i * 32 + (lane & 31);
      if (lane < 32 && h < H) dbslab[((size_t)s * R + r) * H + h] = t;
    }
  }
  float* __restrict__ dst = slab + ((size_t)s * R + r) * H * L;
  const AccCoord<BM, BN> cc(h0, l0);
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    const int col = cc.col(tn);
    if (col < L) {
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          if (row < H) dst[(size_t)row * L + col] = acc[tm][tn][i];
        }
    }
  }
}

// ---- weight gradient, per-sample form ------------------------------------------------------------------------
// dW1_r[h,l] = sum_b h2[b,r,h] * P_b[h,l],  P_b = G_b^T X_b over the N rows of sample b: the rank scale is constant inside
// a sample, so ONE MFMA accumulation P serves all R ranks (the (g*h2_r)-scaled form above repeats the whole contraction
// per rank) and both operands are staged as they lie in memory (no scale, no second load per slot).  The K loop walks
// the samples of a row slab inside one software pipeline: every sample is padded to whole 16-row stages (36 -> 48
// virtual rows, the pad zero-filled at LDS-store time) and after its last stage a hook folds P into the R accumulators
// (acc_r += h2[b,r,row] * P) and clears it.  MFMA work: (padded/N)/R of the scaled form = 0.67x at N = 36, R = 2.
struct SampleRows {
  int N, NP, b_lo, nb;
  uint32_t inv;  // floor(2^32 / NP) + 1: exact quotient for the few thousand virtual rows of a slab
  __device__ __forceinline__ void split(int kp, int& b, int& i) const {
    b = (int)__umulhi((uint32_t)kp, inv);
    i = kp - b * NP;
  }
};
struct SrcSampleMC {  // X[virtual row][mn], MN-contiguous rows of stride ld
  using Raw = float2;
  const float* p;
  int ld, MN;
  SampleRows sr;
  __device__ __forceinline__ Raw fetch(int mn, int kp) const {
    int b, i;
    sr.split(kp, b, i);
    const int row = (sr.b_lo + min(b, sr.nb - 1)) * sr.N + min(i, sr.N - 1);
    return ld2(p + (size_t)row * ld + min(mn, MN - 2));
  }
  __device__ __forceinline__ float2 finish(Raw v, int mn, int kp) const {
    int b, i;
    sr.split(kp, b, i);
    return keep_if(mn < MN && i < sr.N && b < sr.nb, v);
  }
};

template <int BM, int BN, int PF, int R>
__global__ __launch_bounds__(kGemmThreads) void bilinear_dw_sample_kernel(
    const float* __restrict__ g, const float* __restrict__ h2, const float* __restrict__ x, int ldx, float* __restrict__ slab,
    float* __restrict__ dbslab, int B, int N, int L, int H, int tiles_m, int tiles_n, int samples_per_split, int SP,
    uint32_t inv) {
  using T = GemmTile<BM, BN, 16, false, false>;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  float* h2_s = smem + 2 * T::kStageFloats;  // [samples of the slab][R][BM]
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = tiles_m * tiles_n;
  const int s = bid / tiles, tile = bid % tiles;
  const int h0 = (tile / tiles_n) * BM, l0 = (tile % tiles_n) * BN;
  const int b_lo = s * samples_per_split;
  const int nb = max(0, min(B - b_lo, samples_per_split));
  for (int t = threadIdx.x; t < nb * R * BM; t += kGemmThreads) {
    const int hh = t % BM, r = (t / BM) % R, bb = t / (BM * R);
    h2_s[t] = h0 + hh < H ? h2[((size_t)(b_lo + bb) * R + r) * H + h0 + hh] : 0.f;
  }
  // (visible after the first barrier inside gemm_tile)
  const int lane = threadIdx.x & 63, wm = threadIdx.x >> 7;
  f32x16 P[T::TM][T::TN], acc[R][T::TM][T::TN];
  zero_acc(P);
  float colsum[T::TM], dbacc[R][T::TM];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    zero_acc(acc[r]);
#pragma unroll
    for (int i = 0; i < T::TM; ++i) dbacc[r][i] = 0.f;
  }
#pragma unroll
  for (int i = 0; i < T::TM; ++i) colsum[i] = 0.f;
  const SampleRows sr{N, SP * 16, b_lo, max(nb, 1), inv};
  const SrcSampleMC sa{g, H, H, sr};
  const SrcSampleMC sb{x, ldx, L, sr};
  const int row_base = wm * (BM / 2) + 4 * (lane >> 5);
  auto fold = [&](int st) {
    if ((st + 1) % SP != 0) return;
    const float* __restrict__ hb = h2_s + (size_t)(st / SP) * R * BM;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float sc = hb[r * BM + row_base + tm * 32 + (i & 3) + 8 * (i >> 2)];
#pragma unroll
          for (int tn = 0; tn < T::TN; ++tn) acc[r][tm][tn][i] = fmaf(sc, P[tm][tn][i], acc[r][tm][tn][i]);
        }
        dbacc[r][tm] = fmaf(hb[r * BM + wm * (BM / 2) + tm * 32 + (lane & 31)], colsum[tm], dbacc[r][tm]);
      }
    zero_acc(P);
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm) colsum[tm] = 0.f;
  };
  if (nb > 0) gemm_tile<BM, BN, 16, PF, false, false>(sa, sb, h0, l0, 0, nb * SP * 16, smem, P, colsum, fold);
  const AccCoord<BM, BN> cc(h0, l0);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (l0 == 0 && (threadIdx.x >> 6 & 1) == 0) {
#pragma unroll
      for (int i = 0; i < T::TM; ++i) {
        const float t = dbacc[r][i] + __shfl_xor(dbacc[r][i], 32, 64);
        const int h = h0 + wm * (T::TM * 32) + i * 32 + (lane & 31);
        if (lane < 32 && h < H) dbslab[((size_t)s * R + r) * H + h] = t;
      }
    }
    float* __restrict__ dst = slab + ((size_t)s * R + r) * H * L;
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
      const int col = cc.col(tn);
      if (col < L) {
#pragma unroll
        for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = cc.row(tm, i);
            if (row < H) dst[(size_t)row * L + col] = acc[r][tm][tn][i];
          }
      }
    }
  }
}

// Rank-folded backward, weight side.  The per-sample product P_b = g_b^T x_b that the kernel above accumulates into
// the weight gradients also carries the question-side gradient:
//     dh2[b,r,h] = sum_n g[b,n,h] (W1_r x[b,n,:] + b1_r)[h] = sum_l W1_r[h,l] P_b[h,l] + b1_r[h] sum_n g[b,n,h]
// so no [M,R,H] intermediate has to be saved by the forward.  The MFMA operands are swapped (gemm_tile SWAP_AB): a lane
// then holds ONE feature h and 16 region-side columns l of the 32x32 block, the rank scaling reads one h2 value per
// lane, and the sum over l is an in-lane dot product with the lane's W1_r registers -- the two lane halves meet in one
// cross-lane add and the 2 * tiles_n partial sums per (b, r, h) go to a slab that a fixed-order kernel reduces.
// BK: rows of a pipeline stage.  16 in general (a sample = SP stages of 16 rows); 40 when a whole sample fits one stage
// (32 < N <= 40, the reference's 36 regions): one barrier and one fold per sample instead of three, 40 instead of 48
// padded rows.
template <int BM, int BN, int PF, int R, int BK>
__global__ __launch_bounds__(kGemmThreads) void bilinear_dw_dh2_kernel(
    const float* __restrict__ g, const float* __restrict__ h2, const float* __restrict__ x, int ldx, RankPtrs rp,
    float* __restrict__ slab, float* __restrict__ dbslab, float* __restrict__ dh2part, int B, int N, int L, int H,
    int tiles_m, int tiles_n, int samples_per_split, int SP, uint32_t inv) {
  using T = GemmTile<BM, BN, BK, false, false>;
  static_assert(T::TM == 1 && T::TN == 1, "one 32x32 block per wave");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  float* h2_s = smem + 2 * T::kStageFloats;  // [samples of the slab][R][BM]
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = tiles_m * tiles_n;
  const int s = bid / tiles, tile = bid % tiles;
  const int tile_n = tile % tiles_n;
  const int h0 = (tile / tiles_n) * BM, l0 = tile_n * BN;
  const int b_lo = s * samples_per_split;
  const int nb = max(0, min(B - b_lo, samples_per_split));
  for (int t = threadIdx.x; t < nb * R * BM; t += kGemmThreads) {
    const int hh = t % BM, r = (t / BM) % R, bb = t / (BM * R);
    h2_s[t] = h0 + hh < H ? h2[((size_t)(b_lo + bb) * R + r) * H + h0 + hh] : 0.f;
  }
  // (visible after the first barrier inside gemm_tile)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
  const int hloc = wm * 32 + (lane & 31), h = h0 + hloc;        // this lane's feature
  const int lbase = l0 + wn * 32 + 4 * (lane >> 5);            // + (i & 3) + 8 * (i >> 2): its 16 region-side columns
  float wreg[R][16], b1v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    b1v[r] = (l0 == 0 && wn == 0 && h < H) ? rp.b[r][h] : 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int l = lbase + (i & 3) + 8 * (i >> 2);
      wreg[r][i] = (h < H && l < L) ? rp.w[r][(size_t)h * L + l] : 0.f;
    }
  }
  f32x16 P[1][1], acc[R];
  zero_acc(P);
  float colsum[1] = {0.f}, dbacc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    dbacc[r] = 0.f;
  }
  const SampleRows sr{N, SP * BK, b_lo, max(nb, 1), inv};
  const SrcSampleMC sa{g, H, H, sr};
  const SrcSampleMC sb{x, ldx, L, sr};
  float* __restrict__ part = dh2part + (size_t)(tile_n * 2 + wn) * B * R * H;
  auto fold = [&](int st) {
    if ((st + 1) % SP != 0) return;
    const int bb = st / SP;
    const float* __restrict__ hb = h2_s + (size_t)bb * R * BM;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float sc = hb[r * BM + hloc];
      float d = b1v[r] * colsum[0];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[r][i] = fmaf(sc, P[0][0][i], acc[r][i]);
        d = fmaf(wreg[r][i], P[0][0][i], d);
      }
      dbacc[r] = fmaf(sc, colsum[0], dbacc[r]);
      d += __shfl_xor(d, 32, 64);
      if (lane < 32 && h < H) part[((size_t)(b_lo + bb) * R + r) * H + h] = d;
    }
    zero_acc(P);
    colsum[0] = 0.f;
  };
  if (nb > 0) gemm_tile<BM, BN, BK, PF, false, false, true>(sa, sb, h0, l0, 0, nb * SP * BK, smem, P, colsum, fold);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (l0 == 0 && wn == 0) {
      const float t = dbacc[r] + __shfl_xor(dbacc[r], 32, 64);
      if (lane < 32 && h < H) dbslab[((size_t)s * R + r) * H + h] = t;
    }
    if (h < H) {
      float* __restrict__ dst = slab + ((size_t)s * R + r) * H * L + (size_t)h * L;
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // registers 4q .. 4q+3 are four consecutive columns
        const int l = lbase + 8 * q;
        if (l < L) st2(dst + l, make_float2(acc[r][4 * q], acc[r][4 * q + 1]));
        if (l + 2 < L) st2(dst + l + 2, make_float2(acc[r][4 * q + 2], acc[r][4 * q + 3]));
      }
    }
  }
}

// dh2[e] = sum_p part[p][e]   (e over B*R*H, float2 lanes; fixed order)
__device__ __forceinline__ void dh2_reduce_block(const float* __restrict__ part, float* __restrict__ dh2, size_t n, int parts,
                                                 unsigned block) {
  const size_t e = ((size_t)block * 256 + threadIdx.x) * 2;
  if (e >= n) return;
  float2 a = make_float2(0.f, 0.f);
  for (int p = 0; p < parts; ++p) {
    const float2 t = ld2(part + (size_t)p * n + e);
    a.x += t.x;
    a.y += t.y;
  }
  st2(dh2 + e, a);
}

// d_w1[r][h][l] = sum_s slab[s][r][h][l]   (fixed order: bitwise reproducible)
__device__ __forceinline__ void dw_reduce_block(const float* __restrict__ slab, const float* __restrict__ dbslab,
                                                const RankOutPtrs& out, int HL, int H, int R, int S, int r, int block) {
  const int e = (block * 256 + threadIdx.x) * 2;
  if (e < H) {  // bias gradient: the first H/2 lanes also fold the db partials (same fixed order)
    float2 a = make_float2(0.f, 0.f);
    for (int s = 0; s < S; ++s) {
      const float2 t = ld2(dbslab + ((size_t)s * R + r) * H + e);
      a.x += t.x;
      a.y += t.y;
    }
    st2(out.b[r] + e, a);
  }
  if (e >= HL) return;
  float2 a = make_float2(0.f, 0.f);
  for (int s = 0; s < S; ++s) {
    const float2 t = ld2(slab + ((size_t)s * R + r) * HL + e);
    a.x += t.x;
    a.y += t.y;
  }
  st2(out.w[r] + e, a);
}

__global__ __launch_bounds__(256) void bilinear_dh2_reduce_kernel(const float* __restrict__ part, float* __restrict__ dh2,
                                                                  size_t n, int parts) {
  dh2_reduce_block(part, dh2, n, parts, blockIdx.x);
}
__global__ __launch_bounds__(256) void bilinear_dw_reduce_kernel(const float* __restrict__ slab,
                                                                 const float* __restrict__ dbslab, RankOutPtrs out,
                                                                 int HL, int H, int R, int S) {
  dw_reduce_block(slab, dbslab, out, HL, H, R, S, blockIdx.y, blockIdx.x);
}
// both fixed-order reductions of the folded backward in ONE launch (they are independent and ~5 us each, i.e. launch-sized):
// blocks [0, R * nb_dw) take the weight slabs, the rest the dh2 partial sums
__global__ __launch_bounds__(256) void bilinear_dw_dh2_reduce_kernel(const float* __restrict__ slab,
                                                                     const float* __restrict__ dbslab, RankOutPtrs out, int HL,
                                                                     int H, int R, int S, int nb_dw,
                                                                     const float* __restrict__ part, float* __restrict__ dh2,
                                                                     size_t n, int parts) {
  const int id = blockIdx.x;
  if (id < R * nb_dw)
    dw_reduce_block(slab, dbslab, out, HL, H, R, S, id / nb_dw, id % nb_dw);
  else
    dh2_reduce_block(part, dh2, n, parts, (unsigned)(id - R * nb_dw));
}

// dh2[b][r][h] = sum_n g[b,n,h] * h1[b,n,r,h].  grid (ceil(H/128), B); 256 lanes = 64 feature pairs x 4 region
// slices (more loads in flight than one lane per feature pair); the slices meet in LDS.
__global__ __launch_bounds__(256) void bilinear_dh2_kernel(const float* __restrict__ g, const float* __restrict__ h1,
                                                           float* __restrict__ dh2, int N, int H, int R) {
  __shared__ float2 part[3][kMaxR][64];
  const int b = blockIdx.y;
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int h = (blockIdx.x * 64 + c) * 2;
  const bool active = h < H;
  const int hc = active ? h : 0;
  float2 acc[kMaxR];
#pragma unroll
  for (int r = 0; r < kMaxR; ++r) acc[r] = make_float2(0.f, 0.f);
  const float* gb = g + (size_t)b * N * H + hc;
  const float* hb = h1 + (size_t)b * N * R * H + hc;
#pragma unroll 3
  for (int n = slice; n < N; n += 4) {
    const float2 gv = ld2(gb + (size_t)n * H);
#pragma unroll
    for (int r = 0; r < kMaxR; ++r) {
      if (r < R) {
        const float2 hv = ld2(hb + ((size_t)n * R + r) * H);
        acc[r].x = fmaf(gv.x, hv.x, acc[r].x);
        acc[r].y = fmaf(gv.y, hv.y, acc[r].y);
      }
    }
  }
  if (slice > 0) {
#pragma unroll
    for (int r = 0; r < kMaxR; ++r)
      if (r < R) part[slice - 1][r][c] = acc[r];
  }
  __syncthreads();
  if (slice == 0 && active) {
#pragma unroll
    for (int r = 0; r < kMaxR; ++r) {
      if (r < R) {
        float2 t = acc[r];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          t.x += part[q][r][c].x;
          t.y += part[q][r][c].y;
        }
        st2(dh2 + ((size_t)b * R + r) * H + h, t);
      }
    }
  }
}

static int splits_for_dw(int M, int H, int L, int R, TileChoice t) {
  const long tiles = (long)((H + t.bm - 1) / t.bm) * ((L + t.bn - 1) / t.bn) * R;
  long s = (1280 + tiles - 1) / tiles;  // 5 workgroups per CU
  if (const char* e = vqa::option("VQA_K4_DW_SPLITS")) s = std::atol(e);  // experiment knob
  const long max_by_rows = (M + 255) / 256;  // keep >= 256 rows (16 stages) per split
  if (s > max_by_rows) s = max_by_rows;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return (int)s;
}

// 64x64: measured fastest for both forms once the grid is dealt per XCD (step 3.59 ms against 3.66 with 128x64)
static TileChoice dw_tile() { return tile_override_or({64, 64, 2}); }

// per-sample form: worth it when a sample fills most of its padded stages and the ranks fit in registers
static bool dw_per_sample(int N, int R) {
  const int padded = (N + 15) / 16 * 16;
  if (const char* e = vqa::option("VQA_K4_DW_FORM")) return std::atoi(e) != 0 && R <= 4;  // experiment knob
  return R >= 2 && R <= 4 && N * 4 >= padded * 3;
}
static long dw_max_samples_per_slab(int R) { return 65536 / ((long)R * 64 * (long)sizeof(float)); }
static int dw_sample_splits(int B, int H, int L, TileChoice t, int R) {
  const long tiles = (long)((H + t.bm - 1) / t.bm) * ((L + t.bn - 1) / t.bn);
  long s = (1280 + tiles - 1) / tiles;  // 5 workgroups per CU: 32 slabs of 16 samples at B = 512 (sweep: 20..64)
  if (const char* e = vqa::option("VQA_K4_DW_SPLITS")) s = std::atol(e);  // experiment knob
  if (s > 64) s = 64;
  // the question-side factors of a slab's samples wait in LDS ([samples][R][BM] floats): at most 64 KB of them
  const long per = dw_max_samples_per_slab(R) * 64 / t.bm;
  const long by_lds = (B + per - 1) / per;
  if (s < by_lds) s = by_lds;
  if (s > B) s = B;
  if (s < 1) s = 1;
  return (int)s;
}
static int dw_slabs(int B, int N, int L, int H, int R) {
  return dw_per_sample(N, R) ? dw_sample_splits(B, H, L, dw_tile(), R) : splits_for_dw(B * N, H, L, R, dw_tile());
}

}  // namespace vqa

using namespace vqa;

static int check_common(const char* who, const void* x, int ldx, int B, int N, int L, int H, int R) {
  VQA_REQUIRE(B > 0 && N > 0 && L > 0 && H > 0 && R > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d L=%d H=%d R=%d", who, B, N,
              L, H, R);
  VQA_REQUIRE(R <= kMaxR, VQA_E_UNSUPPORTED, "%s: R=%d exceeds %d", who, R, kMaxR);
  VQA_REQUIRE(L % 2 == 0 && H % 2 == 0 && ldx % 2 == 0 && ldx >= L, VQA_E_UNSUPPORTED,
              "%s: needs even L, H, ldx and ldx >= L (L=%d H=%d ldx=%d)", who, L, H, ldx);
  VQA_REQUIRE(aligned(x, 8), VQA_E_UNSUPPORTED, "%s: x must be 8-byte aligned", who);
  VQA_REQUIRE((long)B * N < (1L << 30) && (long)B * N * N < (1L << 32), VQA_E_UNSUPPORTED, "%s: B*N too large", who);
  return VQA_OK;
}

extern "C" int vqa_lowrank_bilinear_fusion_fwd(const float* x, int ldx, const float* const* w1, const float* const* b1,
                                               const float* h2, float* out, float* h1, int B, int N, int L, int H, int R,
                                               vqa_stream_t stream) {
  VQA_REQUIRE(x && w1 && b1 && h2 && out, VQA_E_BADARG, "lowrank_bilinear_fusion_fwd: null pointer");
  int rc = check_common("lowrank_bilinear_fusion_fwd", x, ldx, B, N, L, H, R);
  if (rc != VQA_OK) return rc;
  RankPtrs rp{};
  for (int r = 0; r < R; ++r) {
    VQA_REQUIRE(w1[r] && b1[r] && aligned(w1[r], 8), VQA_E_BADARG, "lowrank_bilinear_fusion_fwd: w1[%d]/b1[%d] null or unaligned", r, r);
    rp.w[r] = w1[r];
    rp.b[r] = b1[r];
  }
  VQA_REQUIRE(aligned(h2, 8) && aligned(out, 8), VQA_E_UNSUPPORTED, "lowrank_bilinear_fusion_fwd: h2/out must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int M = B * N;
  const TileChoice t = tile_override_or(choose_tile(M, H, 1));
  const int tiles_m = (M + t.bm - 1) / t.bm, tiles_n = (H + t.bn - 1) / t.bn;
#define LAUNCH(BM_, BN_, BK_)                                                                                              \
  {                                                                                                                   \
    const size_t lds = GemmTile<BM_, BN_, 16, true, true>::kSmemBytes + BM_ * sizeof(int);                                \
    VQA_ENSURE_LDS((bilinear_fwd_kernel<BM_, BN_, BK_>), lds);                                                        \
    VQA_LAUNCH((bilinear_fwd_kernel<BM_, BN_, BK_>), dim3(tiles_m * tiles_n), dim3(kGemmThreads), lds, s, x, ldx,   \
                       rp, h2, out, h1, M, N, L, H, R, tiles_n);                                                      \
  }
  VQA_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
  return check_launch("lowrank_bilinear_fusion_fwd");
}

extern "C" size_t vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(int B, int N, int L, int H, int R) {
  if (B <= 0 || N <= 0 || L <= 0 || H <= 0 || R <= 0 || R > kMaxR) return 0;
  const int S = dw_slabs(B, N, L, H, R);
  return ((size_t)S * R * H * L + (size_t)S * R * H) * sizeof(float);
}

extern "C" int vqa_lowrank_bilinear_fusion_bwd(const float* x, int ldx, const float* const* w1, const float* h2,
                                               const float* h1, const float* g, float* d_x, float* const* d_w1,
                                               float* const* d_b1, float* d_h2, void* workspace, size_t workspace_bytes,
                                               int B, int N, int L, int H, int R, vqa_stream_t stream) {
  VQA_REQUIRE(x && w1 && h2 && h1 && g && d_w1 && d_b1 && d_h2 && workspace, VQA_E_BADARG,
              "lowrank_bilinear_fusion_bwd: null pointer");
  int rc = check_common("lowrank_bilinear_fusion_bwd", x, ldx, B, N, L, H, R);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(workspace_bytes >= vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(B, N, L, H, R), VQA_E_BADARG,
              "lowrank_bilinear_fusion_bwd: workspace of %zu B is too small", workspace_bytes);
  RankPtrs rp{};
  RankOutPtrs ro{};
  for (int r = 0; r < R; ++r) {
    VQA_REQUIRE(w1[r] && d_w1[r] && d_b1[r] && aligned(w1[r], 8) && aligned(d_w1[r], 8) && aligned(d_b1[r], 8), VQA_E_BADARG,
                "lowrank_bilinear_fusion_bwd: rank %d pointer null or unaligned", r);
    rp.w[r] = w1[r];
    ro.w[r] = d_w1[r];
    ro.b[r] = d_b1[r];
  }
  VQA_REQUIRE(aligned(h2, 8) && aligned(h1, 8) && aligned(g, 8) && aligned(d_h2, 8) && aligned(workspace, 16) &&
                  (d_x == nullptr || aligned(d_x, 8)),
              VQA_E_UNSUPPORTED, "lowrank_bilinear_fusion_bwd: tensors must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int M = B * N;
  const TileChoice tw = dw_tile();
  const int S = dw_slabs(B, N, L, H, R);
  float* slab = static_cast<float*>(workspace);
  float* dbslab = slab + (size_t)S * R * H * L;

  // (1) dh2
  VQA_LAUNCH(bilinear_dh2_kernel, dim3((H / 2 + 63) / 64, B), dim3(256), 0, s, g, h1, d_h2, N, H, R);
  // (2) dx
  if (d_x != nullptr) {
    const TileChoice t = tile_override_or(choose_tile(M, L, 1));
    const int tiles_m = (M + t.bm - 1) / t.bm, tiles_n = (L + t.bn - 1) / t.bn;
#define LAUNCH(BM_, BN_, BK_)                                                                                             \
  {                                                                                                                  \
    const size_t lds = GemmTile<BM_, BN_, 16, true, false>::kSmemBytes;                                                  \
    VQA_ENSURE_LDS((bilinear_dx_kernel<BM_, BN_, BK_>), lds);                                                        \
    VQA_LAUNCH((bilinear_dx_kernel<BM_, BN_, BK_>), dim3(tiles_m * tiles_n), dim3(kGemmThreads), lds, s, g, rp, h2, \
                       d_x, M, N, L, H, R, tiles_n, make_row_to_sample(N));                                                                 \
  }
    VQA_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
  }
  // (3) dW1 (+ db1 partials): split over rows into slabs, (4) reduce the slabs in fixed order
  {
    const int tiles_m = (H + tw.bm - 1) / tw.bm, tiles_n = (L + tw.bn - 1) / tw.bn;
    if (dw_per_sample(N, R)) {
      const int spl = (B + S - 1) / S, SP = (N + 15) / 16;
      const uint32_t inv = (uint32_t)((1ull << 32) / (uint32_t)(SP * 16)) + 1u;
#define LAUNCH_R(BM_, BN_, PF_, R_)                                                                                        \
  {                                                                                                                        \
    const size_t lds = GemmTile<BM_, BN_, 16, false, false>::kSmemBytes + (size_t)spl * R_ * BM_ * sizeof(float);         \
    VQA_ENSURE_LDS((bilinear_dw_sample_kernel<BM_, BN_, PF_, R_>), lds);                                                   \
    VQA_LAUNCH((bilinear_dw_sample_kernel<BM_, BN_, PF_, R_>), dim3(tiles_m * tiles_n * S), dim3(kGemmThreads), lds, \
                       s, g, h2, x, ldx, slab, dbslab, B, N, L, H, tiles_m, tiles_n, spl, SP, inv);                        \
  }
#define LAUNCH(BM_, BN_, PF_)                 \
  {                                           \
    if (R == 2) {                             \
      LAUNCH_R(BM_, BN_, PF_, 2)              \
    } else if (R == 3) {                      \
      LAUNCH_R(BM_, BN_, PF_, 3)              \
    } else {                                  \
      LAUNCH_R(BM_, BN_, PF_, 4)              \
    }                                         \
  }
      VQA_TILE_SWITCH(tw, LAUNCH);
#undef LAUNCH
#undef LAUNCH_R
    } else {
    int rows_per_split = (M + S - 1) / S;
    rows_per_split = (rows_per_split + 31) / 32 * 32;
#define LAUNCH(BM_, BN_, BK_)                                                                                            \
  {                                                                                                                 \
    const size_t lds = GemmTile<BM_, BN_, 16, false, false>::kSmemBytes;                                                \
    VQA_ENSURE_LDS((bilinear_dw_kernel<BM_, BN_, BK_>), lds);                                                       \
    VQA_LAUNCH((bilinear_dw_kernel<BM_, BN_, BK_>), dim3(tiles_m * tiles_n * R * S), dim3(kGemmThreads), lds, s, g,  \
                       h2, x, ldx, slab, dbslab, M, N, L, H, R, tiles_m, tiles_n, rows_per_split, make_row_to_sample(N));                          \
  }
    VQA_TILE_SWITCH(tw, LAUNCH);
#undef LAUNCH
    }
    const int HL = H * L;
    VQA_LAUNCH(bilinear_dw_reduce_kernel, dim3((HL / 2 + 255) / 256, R), dim3(256), 0, s, slab, dbslab, ro, HL, H, R, S);
  }
  return check_launch("lowrank_bilinear_fusion_bwd");
}

// ---- rank-folded backward (forward: bilinear_folded.hip) -----------------------------------------------------------
// sample slabs of the folded weight-gradient kernel: ONE round of resident workgroups (register budget: 3 per CU at
// ranks 1-2 with 16-row stages, 2 otherwise), as many slabs as fit -- at B = 512, N = 36: 12 slabs of the one-stage-per-
// sample form, 133 us + 8 us reduce (sweeps: 19 slabs of the 16-row form 137 + 12; 32 slabs 141 + 19).  The kernel is
// bound by the L2 -> LDS operand traffic of its 64x64 tiles (16 FLOP per byte), not by barriers or padding.
static int dw_fold_bk(int N, int R) {
  static const int knob = [] {   // experiment knob: 16 forces the three-stage form at 32 < N <= 40
    const char* e = vqa::option("VQA_K4_FOLD_DW_BK");
    return e != nullptr ? std::atoi(e) : 0;
  }();
  return (N > 32 && N <= 40 && R <= 2 && knob != 16) ? 40 : 16;
}
static int dw_fold_splits(int B, int N, int H, int L, int R) {
  const long tiles = (long)((H + 63) / 64) * ((L + 63) / 64);
  const int resident = dw_fold_bk(N, R) == 40 ? 2 : (R <= 2 ? 3 : 2);   // workgroups per CU (register budget)
  long s = (256L * resident) / tiles;
  if (const char* e = vqa::option("VQA_K4_DW_SPLITS")) s = std::atol(e);  // experiment knob
  if (s > 64) s = 64;
  // the question-side factors of a slab's samples wait in LDS ([samples][R][64] floats): at most 64 KB of them
  const long by_lds = (B + dw_max_samples_per_slab(R) - 1) / dw_max_samples_per_slab(R);
  if (s < by_lds) s = by_lds;
  if (s > B) s = B;
  if (s < 1) s = 1;
  return (int)s;
}
static size_t folded_bwd_floats(int B, int N, int L, int H, int R, size_t* wt_off, size_t* slab_off, size_t* db_off,
                                size_t* part_off) {
  int S = dw_fold_splits(B, N, H, L, R);
  if (dw_split_supported(B, N, L, H, R, L) && S < kDwSplitSlabs) S = kDwSplitSlabs;      // (the split form's slab count)
  const int tiles_n = (L + 63) / 64;
  size_t off = 0;
  *wt_off = off;
  off += (size_t)R * L * H;
  *slab_off = off;
  off += (size_t)S * R * H * L;
  *db_off = off;
  off += (size_t)S * R * H;
  off = (off + 3) / 4 * 4;
  *part_off = off;
  off += (size_t)2 * tiles_n * B * R * H;
  return off;
}

extern "C" size_t vqa_lowrank_bilinear_fusion_folded_bwd_workspace_bytes(int B, int N, int L, int H, int R) {
  if (!folded_supported(B, N, L, H, R)) return 0;
  size_t a, b, c, d;
  return folded_bwd_floats(B, N, L, H, R, &a, &b, &c, &d) * sizeof(float);
}

extern "C" int vqa_lowrank_bilinear_fusion_folded_bwd(const float* x, int ldx, const float* const* w1,
                                                      const float* const* b1, const float* h2, const float* g, float* d_x,
                                                      float* const* d_w1, float* const* d_b1, float* d_h2, void* workspace,
                                                      size_t workspace_bytes, int B, int N, int L, int H, int R,
                                                      vqa_stream_t stream) {
  return vqa_lowrank_bilinear_fusion_folded_bwd_gated(x, ldx, w1, b1, h2, g, d_x, d_w1, d_b1, d_h2, workspace, workspace_bytes, B,
                                                      N, L, H, R, 0, stream);
}

extern "C" int vqa_lowrank_bilinear_fusion_folded_bwd_gated(const float* x, int ldx, const float* const* w1,
                                                            const float* const* b1, const float* h2, const float* g,
                                                            float* d_x, float* const* d_w1, float* const* d_b1, float* d_h2,
                                                            void* workspace, size_t workspace_bytes, int B, int N, int L,
                                                            int H, int R, int gate_dx, vqa_stream_t stream) {
  VQA_REQUIRE(x && w1 && b1 && h2 && g && d_w1 && d_b1 && d_h2 && workspace, VQA_E_BADARG,
              "lowrank_bilinear_fusion_folded_bwd: null pointer");
  VQA_REQUIRE(folded_supported(B, N, L, H, R), VQA_E_UNSUPPORTED,
              "lowrank_bilinear_fusion_folded_bwd: shape outside the folded form (B=%d N=%d L=%d H=%d R=%d)", B, N, L, H, R);
  int rc = check_common("lowrank_bilinear_fusion_folded_bwd", x, ldx, B, N, L, H, R);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(workspace_bytes >= vqa_lowrank_bilinear_fusion_folded_bwd_workspace_bytes(B, N, L, H, R), VQA_E_BADARG,
              "lowrank_bilinear_fusion_folded_bwd: workspace of %zu B is too small", workspace_bytes);
  RankPtrs rp{};
  RankOutPtrs ro{};
  for (int r = 0; r < R; ++r) {
    VQA_REQUIRE(w1[r] && b1[r] && d_w1[r] && d_b1[r] && aligned(w1[r], 8) && aligned(d_w1[r], 8) && aligned(d_b1[r], 8),
                VQA_E_BADARG, "lowrank_bilinear_fusion_folded_bwd: rank %d pointer null or unaligned", r);
    rp.w[r] = w1[r];
    rp.b[r] = b1[r];
    ro.w[r] = d_w1[r];
    ro.b[r] = d_b1[r];
  }
  VQA_REQUIRE(aligned(h2, 8) && aligned(g, 8) && aligned(d_h2, 8) && aligned(workspace, 16) && (d_x == nullptr || aligned(d_x, 8)),
              VQA_E_UNSUPPORTED, "lowrank_bilinear_fusion_folded_bwd: tensors must be 8-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t wt_off, slab_off, db_off, part_off;
  folded_bwd_floats(B, N, L, H, R, &wt_off, &slab_off, &db_off, &part_off);
  float* ws = static_cast<float*>(workspace);
  float *wt = ws + wt_off, *slab = ws + slab_off, *dbslab = ws + db_off, *part = ws + part_off;

  // (1) dx = Weff^T g on the folded kernel (needs W1_r^T, contraction-contiguous)
  if (d_x != nullptr) {
    rc = folded_transpose_weights(w1, wt, L, H, R, s);
    if (rc != VQA_OK) return rc;
    const float* wtp[kFoldMaxR];
    for (int r = 0; r < R; ++r) wtp[r] = wt + (size_t)r * L * H;
    VQA_REQUIRE(!gate_dx || ldx == L, VQA_E_UNSUPPORTED, "lowrank_bilinear_fusion_folded_bwd: gate_dx needs a dense x (ldx == L)");
    rc = folded_data_gradient(g, wtp, h2, d_x, B, N, L, H, R, s, gate_dx ? x : nullptr);
    if (rc != VQA_OK) return rc;
  }
  // (2) P_b = g_b^T x_b once per sample: dW1_r, db1_r slabs and dh2
  if (dw_split_supported(B, N, L, H, R, ldx)) {
    // split engine (bilinear_dw_split.hip): 16 slabs, dh2 in four partial sums (the workspace holds 2 * tiles_n >= 4 of them)
    rc = dw_split_launch(g, x, h2, w1, b1, slab, dbslab, part, B, N, L, H, R, s);
    if (rc != VQA_OK) return rc;
    const int HL2 = H * L;
    const size_t n2 = (size_t)B * R * H;
    const int nb_dw = (HL2 / 2 + 255) / 256, nb_dh2 = (int)((n2 / 2 + 255) / 256);
    VQA_LAUNCH(bilinear_dw_dh2_reduce_kernel, dim3(R * nb_dw + nb_dh2), dim3(256), 0, s, slab, dbslab, ro, HL2, H, R,
                       kDwSplitSlabs, nb_dw, part, d_h2, n2, 4);
    return check_launch("lowrank_bilinear_fusion_folded_bwd");
  }
  if (dw_rt_supported(B, N, L, H, R, ldx) && dw_fold_splits(B, N, H, L, R) >= kDwRtGroups) {
    // register-tile form (bilinear_dw_rt.hip): 8 slabs, dh2 in two partial sums (the workspace holds >= 2 of them)
    rc = dw_rt_launch(g, x, h2, w1, b1, slab, dbslab, part, B, N, L, H, R, s);
    if (rc != VQA_OK) return rc;
    const int HL2 = H * L;
    const size_t n2 = (size_t)B * R * H;
    const int nb_dw = (HL2 / 2 + 255) / 256, nb_dh2 = (int)((n2 / 2 + 255) / 256);
    VQA_LAUNCH(bilinear_dw_dh2_reduce_kernel, dim3(R * nb_dw + nb_dh2), dim3(256), 0, s, slab, dbslab, ro, HL2, H, R,
                       kDwRtGroups, nb_dw, part, d_h2, n2, 2);
    return check_launch("lowrank_bilinear_fusion_folded_bwd");
  }
  const int S = dw_fold_splits(B, N, H, L, R);
  const int tiles_m = (H + 63) / 64, tiles_n = (L + 63) / 64;
  static const int pf = [] {   // register sets in flight (experiment knob; 1 keeps three waves per SIMD at R = 2)
    const char* e = vqa::option("VQA_K4_FOLD_DW_PF");
    return e != nullptr && std::atoi(e) == 2 ? 2 : 1;
  }();
  const int bk = dw_fold_bk(N, R);
  const int spl = (B + S - 1) / S, SP = (N + bk - 1) / bk;
  const uint32_t inv = (uint32_t)((1ull << 32) / (uint32_t)(SP * bk)) + 1u;
#define LAUNCH_K(R_, PF_, BK_)                                                                                                  \
  {                                                                                                                             \
    const size_t lds = GemmTile<64, 64, BK_, false, false>::kSmemBytes + (size_t)spl * R_ * 64 * sizeof(float);                 \
    VQA_ENSURE_LDS((bilinear_dw_dh2_kernel<64, 64, PF_, R_, BK_>), lds);                                                        \
    VQA_LAUNCH((bilinear_dw_dh2_kernel<64, 64, PF_, R_, BK_>), dim3(tiles_m * tiles_n * S), dim3(kGemmThreads), lds, s, \
                       g, h2, x, ldx, rp, slab, dbslab, part, B, N, L, H, tiles_m, tiles_n, spl, SP, inv);                      \
  }
#define LAUNCH_PF(R_, PF_) \
  if (bk == 40) LAUNCH_K(R_, PF_, 40) else LAUNCH_K(R_, PF_, 16)
#define LAUNCH_R(R_) \
  if (pf == 2) LAUNCH_PF(R_, 2) else LAUNCH_PF(R_, 1)
  switch (R) {
    case 1: LAUNCH_R(1) break;
    case 2: LAUNCH_R(2) break;
    case 3: LAUNCH_R(3) break;
    default: LAUNCH_R(4) break;
  }
#undef LAUNCH_R
#undef LAUNCH_PF
#undef LAUNCH_K
  // (3) fixed-order reductions
  const int HL = H * L;
  const size_t n = (size_t)B * R * H;
  const int nb_dw = (HL / 2 + 255) / 256, nb_dh2 = (int)((n / 2 + 255) / 256);
  VQA_LAUNCH(bilinear_dw_dh2_reduce_kernel, dim3(R * nb_dw + nb_dh2), dim3(256), 0, s, slab, dbslab, ro, HL, H, R, S, nb_dw,
                     part, d_h2, n, 2 * tiles_n);
  return check_launch("lowrank_bilinear_fusion_folded_bwd");
}
