// fp32 MFMA tile engine for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD, 157 TF chip peak).
//
// One workgroup = 256 threads = 4 waves arranged 2 x 2 over a BM x BN output tile; every wave owns
// TM x TN = (BM/64) x (BN/64) accumulators of 32 x 32 (16 VGPRs each).  The K loop advances 16 at a
// time through a two-stage LDS ring: global -> registers (8-byte loads, issued one stage ahead, so HBM/L2
// latency hides under the previous stage's MFMAs) -> LDS -> MFMA operands.
//
// LDS image (both operands): k-major, Xs[k][mn], so the 32 lanes that feed one MFMA operand row read
// 32 consecutive dwords (ds_read_b32, conflict-free for any stride) -- lane l supplies A[i = l&31][k = l>>5]
// and B[k = l>>5][j = l&31].  A K-contiguous source (x[m][k], W[n][k]) is transposed on the LDS write;
// its row stride is odd (BMN+1) which keeps that transposing ds_write_b32 at <= 2-way (free).  An
// MN-contiguous source (W[k][n], g[m][h] read as [k=m][mn=h]) is written with ds_write_b64.
//
// Sources are functors:  float2 src(int mn, int k)  returning the two elements the calling lane stages --
//   K-contiguous : (mn, k) and (mn, k+1)       MN-contiguous: (k, mn) and (k, mn+1)
// zero outside the matrix, with any prologue (e.g. the question-side scale of the bilinear backward)
// applied on the fly.  k and mn passed to a source are always even, so even extents never straddle.
#pragma once
#include "common.hpp"

namespace vqa {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kBK = 16;
constexpr int kGemmThreads = 256;

template <int BM, int BN, bool A_KC, bool B_KC>
struct GemmTile {
  static_assert(BM == 64 || BM == 128, "BM must be 64 or 128");
  static_assert(BN == 64 || BN == 128, "BN must be 64 or 128");
  static constexpr int TM = BM / 64, TN = BN / 64;
  static constexpr int SA = BM + (A_KC ? 1 : 0);
  static constexpr int SB = BN + (B_KC ? 1 : 0);
  static constexpr int kStageFloats = kBK * (SA + SB);
  static constexpr int kSmemBytes = 2 * kStageFloats * (int)sizeof(float);
  static constexpr int RA = BM / 32, RB = BN / 32;  // float2 registers per thread per stage
};

template <int BMN, bool KC>
struct Stager {
  static constexpr int NREG = BMN / 32;
  static constexpr int VPR = BMN / 2;                  // float2 per k-row of an MN-contiguous tile
  static constexpr int RPP = kGemmThreads / VPR;       // k-rows covered by one pass
  template <class Src>
  __device__ __forceinline__ static void load(float2 (&reg)[NREG], const Src& src, int mn0, int k0, int tid) {
    if constexpr (KC) {
      const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int p = 0; p < NREG; ++p) reg[p] = src(mn0 + p * 32 + rr, k0 + 2 * kq);
    } else {
      const int c = tid % VPR, kr = tid / VPR;
#pragma unroll
      for (int p = 0; p < NREG; ++p) reg[p] = src(mn0 + 2 * c, k0 + p * RPP + kr);
    }
  }
  __device__ __forceinline__ static void store(const float2 (&reg)[NREG], float* Xs, int S, int tid) {
    if constexpr (KC) {
      const int kq = tid & 7, rr = tid >> 3;
#pragma unroll
      for (int p = 0; p < NREG; ++p) {
        Xs[(2 * kq) * S + p * 32 + rr] = reg[p].x;
        Xs[(2 * kq + 1) * S + p * 32 + rr] = reg[p].y;
      }
    } else {
      const int c = tid % VPR, kr = tid / VPR;
#pragma unroll
      for (int p = 0; p < NREG; ++p) st2(&Xs[(p * RPP + kr) * S + 2 * c], reg[p]);
    }
  }
};

// acc += A[m0 : m0+BM, k_begin : k_end) * B[k_begin : k_end), n0 : n0+BN].  All 256 threads call it; it
// ends on a barrier, so the LDS ring may be reused immediately by the next call.
template <int BM, int BN, bool A_KC, bool B_KC, class SrcA, class SrcB>
__device__ __forceinline__ void gemm_tile(const SrcA& srcA, const SrcB& srcB, int m0, int n0, int k_begin, int k_end,
                                          float* smem, f32x16 (&acc)[BM / 64][BN / 64]) {
  using T = GemmTile<BM, BN, A_KC, B_KC>;
  using StA = Stager<BM, A_KC>;
  using StB = Stager<BN, B_KC>;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int a_off = (lane >> 5) * T::SA + wm * (T::TM * 32) + (lane & 31);
  const int b_off = (lane >> 5) * T::SB + wn * (T::TN * 32) + (lane & 31);
  const int nsteps = (k_end - k_begin + kBK - 1) / kBK;
  float2 ra[T::RA], rb[T::RB];
  if (nsteps > 0) {
    StA::load(ra, srcA, m0, k_begin, tid);
    StB::load(rb, srcB, n0, k_begin, tid);
    StA::store(ra, smem, T::SA, tid);
    StB::store(rb, smem + kBK * T::SA, T::SB, tid);
  }
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    const float* As = smem + (s & 1) * T::kStageFloats;
    const float* Bs = As + kBK * T::SA;
    const bool more = s + 1 < nsteps;
    if (more) {
      StA::load(ra, srcA, m0, k_begin + (s + 1) * kBK, tid);
      StB::load(rb, srcB, n0, k_begin + (s + 1) * kBK, tid);
    }
#pragma unroll
    for (int kp = 0; kp < kBK / 2; ++kp) {
      float a[T::TM], b[T::TN];
#pragma unroll
      for (int i = 0; i < T::TM; ++i) a[i] = As[kp * 2 * T::SA + a_off + i * 32];
#pragma unroll
      for (int j = 0; j < T::TN; ++j) b[j] = Bs[kp * 2 * T::SB + b_off + j * 32];
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      float* An = smem + ((s + 1) & 1) * T::kStageFloats;
      StA::store(ra, An, T::SA, tid);
      StB::store(rb, An + kBK * T::SA, T::SB, tid);
    }
    __syncthreads();
  }
}

// Row / column of accumulator register `i` of tile (tm, tn) for the calling lane (C/D layout of
// v_mfma_f32_32x32x2_f32: col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)).
template <int BM, int BN>
struct AccCoord {
  int row0, col0;  // of tile (0,0), register 0
  __device__ __forceinline__ AccCoord(int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    row0 = m0 + (wave >> 1) * (BM / 2) + 4 * (lane >> 5);
    col0 = n0 + (wave & 1) * (BN / 2) + (lane & 31);
  }
  __device__ __forceinline__ int row(int tm, int i) const { return row0 + tm * 32 + (i & 3) + 8 * (i >> 2); }
  __device__ __forceinline__ int col(int tn) const { return col0 + tn * 32; }
};

template <int TM, int TN>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

// ---- plain sources ---------------------------------------------------------------------------
struct SrcKC {  // X[mn][k], K-contiguous rows of stride ld
  const float* p;
  int ld, MN, K;
  __device__ __forceinline__ float2 operator()(int mn, int k) const {
    if (mn < MN && k < K) return ld2(p + (size_t)mn * ld + k);
    return make_float2(0.f, 0.f);
  }
};
struct SrcMC {  // X[k][mn], MN-contiguous rows of stride ld
  const float* p;
  int ld, MN, K;
  __device__ __forceinline__ float2 operator()(int mn, int k) const {
    if (mn < MN && k < K) return ld2(p + (size_t)k * ld + mn);
    return make_float2(0.f, 0.f);
  }
};

// Host-side tile choice: fewest CU-rounds of (padded) work, mild preference for the larger tile.
struct TileChoice {
  int bm, bn;
};
inline TileChoice choose_tile(long M, long N, long splits) {
  const int cand[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
  const double pref[4] = {1.00, 1.04, 1.04, 1.10};
  double best = 1e300;
  TileChoice out{128, 128};
  for (int c = 0; c < 4; ++c) {
    const long tm = (M + cand[c][0] - 1) / cand[c][0], tn = (N + cand[c][1] - 1) / cand[c][1];
    const long tiles = tm * tn * splits;
    const long rounds = (tiles + 255) / 256;
    const double cost = (double)rounds * cand[c][0] * cand[c][1] * pref[c];
    if (cost < best) {
      best = cost;
      out = {cand[c][0], cand[c][1]};
    }
  }
  return out;
}

}  // namespace vqa
