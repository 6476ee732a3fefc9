// fp32 MFMA tile engine for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, 64 FLOP/clk/SIMD, 157 TF chip peak).
//
// One workgroup = 256 threads = 4 waves arranged 2 x 2 over a BM x BN output tile; every wave owns
// TM x TN = (BM/64) x (BN/64) accumulators of 32 x 32 (16 VGPRs each).  The K loop advances 16 at a
// time through a two-stage LDS ring: global -> registers (8-byte loads, issued TWO stages ahead, so HBM/L2
// latency hides under a full stage of MFMAs) -> LDS -> MFMA operands; the staging instructions are interleaved
// into the gaps between MFMAs with sched_group_barrier.
//
// LDS image (both operands): k-major, Xs[k][mn], so the 32 lanes that feed one MFMA operand row read
// 32 consecutive dwords (ds_read_b32, conflict-free for any stride) -- lane l supplies A[i = l&31][k = l>>5]
// and B[k = l>>5][j = l&31].  A K-contiguous source (x[m][k], W[n][k]) is transposed on the LDS write;
// its row stride is odd (BMN+1) which keeps that transposing ds_write_b32 at <= 2-way (free).  An
// MN-contiguous source (W[k][n], g[m][h] read as [k=m][mn=h]) is written with ds_write_b64.
//
// Sources are functors (see Stager) that stage two elements per slot --
//   K-contiguous : (mn, k) and (mn, k+1)       MN-contiguous: (k, mn) and (k, mn+1)
// zero outside the matrix, with any prologue (e.g. the question-side scale of the bilinear backward)
// applied on the fly.  k and mn passed to a source are always even, so even extents never straddle.
#pragma once
#include <cstdlib>

#include "common.hpp"

namespace vqa {

constexpr int kGemmThreads = 256;

template <int BM, int BN, int BK, bool A_KC, bool B_KC>
struct GemmTile {
  static_assert(BK % 8 == 0 && BK >= 16 && BK <= 48, "BK must be a multiple of 8 in 16..48");
  static_assert(BM == 64 || BM == 128, "BM must be 64 or 128");
  static_assert(BN == 64 || BN == 128, "BN must be 64 or 128");
  static constexpr int TM = BM / 64, TN = BN / 64;
  static constexpr int SA = BM + (A_KC ? 1 : 0);
  static constexpr int SB = BN + (B_KC ? 1 : 0);
  static constexpr int kStageFloats = BK * (SA + SB);
  static constexpr int kSmemBytes = 2 * kStageFloats * (int)sizeof(float);
  static constexpr int RA = BM * BK / 512, RB = BN * BK / 512;  // float2 slots per thread per stage
};

// A source functor provides
//   using Raw = ...;                          what one staging slot keeps in registers while in flight
//   Raw   fetch(int mn, int k) const;         UNCONDITIONAL loads from clamped (always valid) addresses
//   float2 finish(Raw, int mn, int k) const;  zero outside the matrix + any prologue, applied at LDS-store time
// Splitting fetch from finish matters: a load inside `if (in_range)` makes hipcc branch around every load and wait
// vmcnt(0) behind each one, and a select right after the load drags the wait in front of the MFMAs -- either way
// the stage serialises on memory latency (measured 1.3-4x on the K loop).
template <int BMN, int BK, bool KC>
struct Stager {
  static constexpr int NREG = BMN * BK / 512;          // float2 slots per thread per stage (256 threads)
  static constexpr int KPR = BK / 2;                   // float2 per mn-row of a K-contiguous tile
  static constexpr int MPP = kGemmThreads / KPR;       // mn-rows covered by one pass (K-contiguous)
  static constexpr int VPR = BMN / 2;                  // float2 per k-row of an MN-contiguous tile
  static constexpr int RPP = kGemmThreads / VPR;       // k-rows covered by one pass (MN-contiguous)
  __device__ __forceinline__ static int mn_of(int p, int tid) { return KC ? p * MPP + tid / KPR : 2 * (tid % VPR); }
  __device__ __forceinline__ static int k_of(int p, int tid) { return KC ? 2 * (tid % KPR) : p * RPP + tid / VPR; }
  template <class Src>
  __device__ __forceinline__ static void load(typename Src::Raw (&reg)[NREG], const Src& src, int mn0, int k0, int tid) {
#pragma unroll
    for (int p = 0; p < NREG; ++p) reg[p] = src.fetch(mn0 + mn_of(p, tid), k0 + k_of(p, tid));
  }
  // ROWK (K-contiguous sources only): the LDS image keeps the source's orientation, [mn][k] rows of pitch S = BK + 4 floats,
  // so a slot is ONE 8-byte store instead of two 4-byte stores a row apart (see gemm_tile's ROWK)
  // CHECK = false (a tile that lies inside both matrices over its whole contraction range): the slots go to LDS as loaded
  // (Src::plain), without the per-slot bounds arithmetic of Src::finish
  template <bool ROWK = false, bool CHECK = true, class Src>
  __device__ __forceinline__ static void store(const typename Src::Raw (&reg)[NREG], const Src& src, int mn0, int k0,
                                               float* Xs, int S, int tid) {
#pragma unroll
    for (int p = 0; p < NREG; ++p) {
      const int mn = mn_of(p, tid), k = k_of(p, tid);
      float2 v;
      if constexpr (CHECK) v = src.finish(reg[p], mn0 + mn, k0 + k);
      else v = src.plain(reg[p]);
      if constexpr (KC && ROWK) {
        st2(&Xs[mn * S + k], v);
      } else if constexpr (KC) {
        Xs[k * S + mn] = v.x;
        Xs[(k + 1) * S + mn] = v.y;
      } else {
        st2(&Xs[k * S + mn], v);
      }
    }
  }
};

// LDS floats of one stage of an operand tile (BMN x BK): [k][mn] rows (pitch BMN, + 1 when the source is K-contiguous and
// stored transposed), or with ROWK and a K-contiguous source [mn][k] rows of pitch BK + 4
template <int BMN, int BK, bool KC, bool ROWK>
struct OperandImage {
  static constexpr bool kRows = KC && ROWK;
  static constexpr int kPitch = kRows ? BK + 4 : BMN + (KC ? 1 : 0);
  static constexpr int kFloats = kRows ? BMN * kPitch : BK * kPitch;
};
template <int BM, int BN, int BK, bool A_KC, bool B_KC, bool ROWK>
constexpr int gemm_stage_floats() {
  return OperandImage<BM, BK, A_KC, ROWK>::kFloats + OperandImage<BN, BK, B_KC, ROWK>::kFloats;
}

// acc += A[m0 : m0+BM, k_begin : k_end) * B[k_begin : k_end), n0 : n0+BN].  All 256 threads call it; it
// ends on a barrier, so the LDS ring may be reused immediately by the next call.
template <int N>
struct IntC {
  static constexpr int value = N;
};

// a_colsum (optional): per-lane partial sums over k of the A fragments this lane fed to the MFMAs, i.e. for
// fragment row i: sum over the k's with (k & 1) == lane >> 5 of A[k][wave_row0 + i*32 + (lane & 31)].  Adding the
// two lane halves (shfl_xor 32) gives the column sums of the A tile -- the bias gradient of a weight-gradient GEMM
// for one extra VALU add per fragment.
struct NoStageHook {
  __device__ __forceinline__ void operator()(int) const {}
};

// hook (optional): called by every thread after stage s is complete (stage s covers k_begin + BK s .. + BK; the call sits
// behind the stage's barrier, outside the MFMA / staging interleave, so it may branch) -- lets a caller fold the
// accumulators away at segment boundaries inside ONE software pipeline (the per-sample rank scaling of the bilinear
// weight gradient).
// SWAP_AB: feed the B fragment as the instruction's A operand and vice versa -- the accumulator block then holds the
// TRANSPOSED 32x32 tile (lane: column = its A-tile row index, 16 rows = B-tile column indices), which turns a reduction
// over the B-side index into an in-lane sum (dh2 of the rank-folded bilinear backward).
// ROWK / FAST (the grouped head's kernel).  Ablation builds of a 4 x [512,310,2400] phase (56 us with its epilogue launch, 22
// of them fixed cost): K loop 35 us; without the staging pass's LDS stores (and the bounds arithmetic that feeds them) 21 us --
// the MFMAs alone are 19.4; without the global loads instead 29 us; without the barriers: no change.  So the loop pays ~8 us
// for the store side of the staging pass and ~6 us waiting for loads, not for fragment reads or barriers.
//   ROWK: a K-contiguous source stored transposed costs two 4-byte stores a row apart per slot.  ROWK keeps such an operand in
//     its own orientation ([mn][k], pitch BK + 4: one 8-byte store per slot) and lets lane half h of the 32x32x2 MFMAs take the
//     contraction steps k = (BK/2) h + kp instead of 2 kp + h, so that a lane's BK/2 fragment values are CONSECUTIVE floats of
//     its row: two ds_read_b128 (conflict-free at that pitch) instead of eight ds_read_b32.  An MN-contiguous operand keeps
//     the [k][mn] image and follows the same k order through its addresses.
//   FAST: an interior tile skips the per-slot bounds selects (see the end of the function).
// Measured on the CoR2 step (grouped GEMM launches per step): 0.350 ms -> 0.337 with both (either one alone: 0.345-0.350).
template <int BM, int BN, int BK, int PF, bool A_KC, bool B_KC, bool SWAP_AB = false, bool ROWK = false, bool FAST = false,
          class SrcA, class SrcB, class Hook = NoStageHook>
__device__ __forceinline__ void gemm_tile(const SrcA& srcA, const SrcB& srcB, int m0, int n0, int k_begin, int k_end,
                                          float* smem, f32x16 (&acc)[BM / 64][BN / 64],
                                          float* a_colsum = nullptr, Hook hook = Hook()) {
  using T = GemmTile<BM, BN, BK, A_KC, B_KC>;
  using StA = Stager<BM, BK, A_KC>;
  using StB = Stager<BN, BK, B_KC>;
  using ImA = OperandImage<BM, BK, A_KC, ROWK>;
  using ImB = OperandImage<BN, BK, B_KC, ROWK>;
  constexpr int kStage = ImA::kFloats + ImB::kFloats;      // (== T::kStageFloats without ROWK)
  static_assert(!ROWK || BK == 16, "ROWK: a lane half's BK/2 = 8 steps are two 16-byte reads");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // offset of (the lane's row / column, its first contraction step) in an operand image; ROWK: steps (BK/2) h .. + BK/2 - 1
  const int a_off = ImA::kRows ? (wm * (T::TM * 32) + (lane & 31)) * ImA::kPitch + (BK / 2) * (lane >> 5)
                               : (ROWK ? (BK / 2) : 1) * (lane >> 5) * ImA::kPitch + wm * (T::TM * 32) + (lane & 31);
  const int b_off = ImB::kRows ? (wn * (T::TN * 32) + (lane & 31)) * ImB::kPitch + (BK / 2) * (lane >> 5)
                               : (ROWK ? (BK / 2) : 1) * (lane >> 5) * ImB::kPitch + wn * (T::TN * 32) + (lane & 31);
  const int nsteps = (k_end - k_begin + BK - 1) / BK;
  // Pipeline (PF register sets in flight, two LDS stages, one barrier per stage):
  //   stage s:  read all fragments of LDS[s&1]  ->  MFMA chain of stage s, with, in the shadow of the MFMAs,
  //             { finish + ds_write of the register set holding stage s+1 into LDS[(s+1)&1] ; global loads of
  //               stage s+1+PF into that same set }  ->  barrier.
  // Every global load therefore has PF full stages of MFMAs to land (measured load-to-use latency under load is
  // ~1.5 us, i.e. more than one 64x64 stage), and the staging instructions issue in the 64-cycle gaps between
  // v_mfma_f32_32x32x2_f32 instead of in a serial phase of their own.  Set indices are compile-time (the stage
  // loop is unrolled by PF) so the sets stay in registers.
  typename SrcA::Raw ra[PF][T::RA];
  typename SrcB::Raw rb[PF][T::RB];
  if (nsteps > 0) {
    StA::load(ra[0], srcA, m0, k_begin, tid);
    StB::load(rb[0], srcB, n0, k_begin, tid);
    StA::template store<ROWK>(ra[0], srcA, m0, k_begin, smem, ImA::kPitch, tid);
    StB::template store<ROWK>(rb[0], srcB, n0, k_begin, smem + ImA::kFloats, ImB::kPitch, tid);
#pragma unroll
    for (int t = 1; t <= PF; ++t) {  // stage t -> set t % PF  (clamped loads: harmless past the end)
      StA::load(ra[t % PF], srcA, m0, k_begin + t * BK, tid);
      StB::load(rb[t % PF], srcB, n0, k_begin + t * BK, tid);
    }
  }
  __syncthreads();
  auto stage = [&](int s, auto set_c, auto check_c) {
    constexpr int q = decltype(set_c)::value;  // set holding stage s+1
    constexpr bool CHECK = decltype(check_c)::value != 0;
    const float* As = smem + (s & 1) * kStage;
    const float* Bs = As + ImA::kFloats;
    float a[BK / 2][T::TM], b[BK / 2][T::TN];
    if constexpr (ImA::kRows) {
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
          const float4 t = ld4(As + a_off + i * 32 * ImA::kPitch + 4 * q);
          a[4 * q][i] = t.x, a[4 * q + 1][i] = t.y, a[4 * q + 2][i] = t.z, a[4 * q + 3][i] = t.w;
        }
    } else {
#pragma unroll
      for (int kp = 0; kp < BK / 2; ++kp)
#pragma unroll
        for (int i = 0; i < T::TM; ++i) a[kp][i] = As[kp * (ROWK ? 1 : 2) * ImA::kPitch + a_off + i * 32];
    }
    if constexpr (ImB::kRows) {
#pragma unroll
      for (int j = 0; j < T::TN; ++j)
#pragma unroll
        for (int q = 0; q < BK / 8; ++q) {
          const float4 t = ld4(Bs + b_off + j * 32 * ImB::kPitch + 4 * q);
          b[4 * q][j] = t.x, b[4 * q + 1][j] = t.y, b[4 * q + 2][j] = t.z, b[4 * q + 3][j] = t.w;
        }
    } else {
#pragma unroll
      for (int kp = 0; kp < BK / 2; ++kp)
#pragma unroll
        for (int j = 0; j < T::TN; ++j) b[kp][j] = Bs[kp * (ROWK ? 1 : 2) * ImB::kPitch + b_off + j * 32];
    }
    __builtin_amdgcn_sched_barrier(0);  // fragment reads stay above; everything below is interleaved by the groups
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp)
#pragma unroll
      for (int i = 0; i < T::TM; ++i)
#pragma unroll
        for (int j = 0; j < T::TN; ++j)
          acc[i][j] = SWAP_AB ? __builtin_amdgcn_mfma_f32_32x32x2f32(b[kp][j], a[kp][i], acc[i][j], 0, 0, 0)
                              : __builtin_amdgcn_mfma_f32_32x32x2f32(a[kp][i], b[kp][j], acc[i][j], 0, 0, 0);
    if (a_colsum != nullptr) {
#pragma unroll
      for (int kp = 0; kp < BK / 2; ++kp)
#pragma unroll
        for (int i = 0; i < T::TM; ++i) a_colsum[i] += a[kp][i];
    }
    {
      // Unconditional on purpose: a branch here would split the basic block and nothing could be interleaved with
      // the MFMAs.  Past the last stage the loads hit clamped (valid) addresses, `finish` zero-fills, and the
      // LDS stage written is not read again before the next call's prologue overwrites it.
      float* An = smem + ((s + 1) & 1) * kStage;
      const int k_next = k_begin + (s + 1) * BK;
      StA::template store<ROWK, CHECK>(ra[q], srcA, m0, k_next, An, ImA::kPitch, tid);
      StB::template store<ROWK, CHECK>(rb[q], srcB, n0, k_next, An + ImA::kFloats, ImB::kPitch, tid);
      StA::load(ra[q], srcA, m0, k_next + PF * BK, tid);
      StB::load(rb[q], srcB, n0, k_next + PF * BK, tid);
    }
    // one MFMA, then a few of the staging instructions, repeated: DS writes, VALU (selects, addresses), VMEM reads
#pragma unroll
    for (int g = 0; g < (BK / 2) * T::TM * T::TN; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);  // DS write
      __builtin_amdgcn_sched_group_barrier(0x002, 12, 0); // VALU
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);  // VMEM read
    }
    __syncthreads();
  };
  // PF stages per pass of a STRAIGHT-LINE body, the last nsteps % PF stages behind the loop.  (With `if (s + 1 < nsteps)`
  // inside the body two paths met at the back edge, the compiler lost count of the loads in flight and waited for
  // vmcnt(0) in front of a stage's LDS writes -- for the register set refilled one stage earlier as well: the loads ran one
  // stage ahead, not PF.  tools/sunk_loads_check.py, round 6.)
  auto loop = [&](auto check_c) {
    int s = 0;
    for (; s + PF <= nsteps; s += PF) {
      stage(s, IntC<1 % PF>{}, check_c);
      hook(s);
      if constexpr (PF > 1) {
        stage(s + 1, IntC<2 % PF>{}, check_c);
        hook(s + 1);
      }
      if constexpr (PF > 2) {
        stage(s + 2, IntC<3 % PF>{}, check_c);
        hook(s + 2);
      }
    }
    if (s < nsteps) {
      stage(s, IntC<1 % PF>{}, check_c);
      hook(s);
      if constexpr (PF > 2) {
        if (s + 1 < nsteps) {
          stage(s + 1, IntC<2 % PF>{}, check_c);
          hook(s + 1);
        }
      }
    }
  };
  if constexpr (FAST) {
    // FAST (sources with covers() / plain()): an interior tile -- whole rows, whole columns, a contraction range of whole
    // stages inside both operands -- runs the loop without the staging pass's bounds arithmetic: next to fp32 MFMAs those
    // VALU instructions are not free (workgroup-uniform choice; the stage written past the last one is never read)
    if (srcA.covers(m0, BM, k_begin, k_end) && srcB.covers(n0, BN, k_begin, k_end) && (k_end - k_begin) % BK == 0)
      loop(IntC<0>{});
    else
      loop(IntC<1>{});
  } else {
    loop(IntC<1>{});
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
__device__ __forceinline__ float2 keep_if(bool ok, float2 v) { return make_float2(ok ? v.x : 0.f, ok ? v.y : 0.f); }

struct SrcKC {  // X[mn][k], K-contiguous rows of stride ld.  Needs MN >= 1, K >= 2 (even).
  using Raw = float2;
  const float* p;
  int ld, MN, K;
  __device__ __forceinline__ Raw fetch(int mn, int k) const { return ld2(p + (size_t)min(mn, MN - 1) * ld + min(k, K - 2)); }
  __device__ __forceinline__ float2 finish(Raw v, int mn, int k) const { return keep_if(mn < MN && k < K, v); }
  __device__ __forceinline__ float2 plain(Raw v) const { return v; }
  __device__ __forceinline__ bool covers(int mn0, int n, int k0, int k1) const { return mn0 + n <= MN && k1 <= K; }
};
struct SrcMC {  // X[k][mn], MN-contiguous rows of stride ld.  Needs K >= 1, MN >= 2 (even).
  using Raw = float2;
  const float* p;
  int ld, MN, K;
  __device__ __forceinline__ Raw fetch(int mn, int k) const { return ld2(p + (size_t)min(k, K - 1) * ld + min(mn, MN - 2)); }
  __device__ __forceinline__ float2 finish(Raw v, int mn, int k) const { return keep_if(mn < MN && k < K, v); }
  __device__ __forceinline__ float2 plain(Raw v) const { return v; }
  __device__ __forceinline__ bool covers(int mn0, int n, int k0, int k1) const { return mn0 + n <= MN && k1 <= K; }
};

// Tried and rejected (round 1, measured with tools/kbench.py): BK = 32 (no gain at K = 310, fewer workgroups per CU);
// PF = 3 register sets (occupancy loss outweighs the extra latency cover); a 3-stage LDS ring with double-buffered
// operand fragments read one stage ahead (+63 VGPRs -> 2 waves per SIMD instead of 4: 83 vs 90 TF/s on the K4 forward);
// a persistent workgroup streaming ONE pipeline across all its (tile, rank) segments so that the next tile's loads
// overlap the current epilogue (per-stage descriptor/address recomputation cost more than the hidden pipeline fill:
// 60 vs 90 TF/s) -- to be retried with incrementally updated per-slot addresses; BK = 32 for the split-row weight
// gradient of the region projections (2304 rows per split: step 3.66 ms against 3.61 with BK = 16); 128x64 / 64x128 tiles
// for the same kernel (3.84 / 3.70 ms against 3.68).

// Host-side tile choice: fewest CU-rounds of (padded) work, mild preference for the larger tile.
struct TileChoice {
  int bm, bn, pf;  // tile rows, tile cols, prefetch distance in stages (register sets in flight)
};
inline TileChoice choose_tile(long M, long N, long splits) {
  const int cand[4][2] = {{128, 128}, {64, 128}, {128, 64}, {64, 64}};
  const double pref[4] = {1.35, 1.15, 1.25, 1.00};  // measured: the 64x64 tile (4 workgroups per CU) wins at these K
  double best = 1e300;
  TileChoice out{128, 128, 2};
  for (int c = 0; c < 4; ++c) {
    const long tm = (M + cand[c][0] - 1) / cand[c][0], tn = (N + cand[c][1] - 1) / cand[c][1];
    const long tiles = tm * tn * splits;
    const long rounds = (tiles + 255) / 256;
    const double cost = (double)rounds * cand[c][0] * cand[c][1] * pref[c];
    if (cost < best) {
      best = cost;
      out = {cand[c][0], cand[c][1], 2};
    }
  }
  return out;
}

inline TileChoice tile_override_or(TileChoice c) {
  const char* e = vqa::option("VQA_GEMM_TILE");  // experiment knob, e.g. "128x64" or "64x64x3" (BM x BN [x PF])
  if (e != nullptr) {
    int bm = 0, bn = 0, pf = 0;
    const int n = std::sscanf(e, "%dx%dx%d", &bm, &bn, &pf);
    if (n >= 2 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128)) {
      c.bm = bm;
      c.bn = bn;
      c.pf = 2;
      if (n == 3 && pf >= 1 && pf <= 3) c.pf = pf;
    }
  }
  return c;
}


}  // namespace vqa

#define VQA_TILE_SWITCH_BK(t, LAUNCH, BK_)          \
  do {                                             \
    if ((t).bm == 128 && (t).bn == 128) {          \
      LAUNCH(128, 128, BK_)                        \
    } else if ((t).bm == 64 && (t).bn == 128) {    \
      LAUNCH(64, 128, BK_)                         \
    } else if ((t).bm == 128 && (t).bn == 64) {    \
      LAUNCH(128, 64, BK_)                         \
    } else {                                       \
      LAUNCH(64, 64, BK_)                          \
    }                                              \
  } while (0)
#define VQA_TILE_SWITCH(t, LAUNCH)         \
  do {                                     \
    if ((t).pf == 1) {                     \
      VQA_TILE_SWITCH_BK(t, LAUNCH, 1);    \
    } else if ((t).pf == 2) {              \
      VQA_TILE_SWITCH_BK(t, LAUNCH, 2);    \
    } else {                               \
      VQA_TILE_SWITCH_BK(t, LAUNCH, 3);    \
    }                                      \
  } while (0)

