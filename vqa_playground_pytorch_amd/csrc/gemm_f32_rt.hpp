// fp32 MFMA "register-tile" engine for gfx950: ONE wave per SIMD, a large accumulator grid per wave, operand fragments
// straight from global memory (L2) into registers -- no LDS, no barrier in the main loop.
//
// Why a second engine next to gemm_f32_mfma.hpp: the region projections of CoR2 / ODA (config/CoR2.py:72-88 at :168-169,
// :213,:218 -- M = B*36 = 18432 rows, K = 2048, N = 310) are 23.4 GFLOP each and MFMA-bound.  fp32 MFMA runs at the
// vector rate (64 FLOP/clk/SIMD), so a wave has to keep its SIMD's matrix pipe busy on every cycle; with 64x64 LDS tiles
// the operand traffic (16 FLOP per L2->LDS byte), the ds_write/ds_read passes and a barrier per 16-deep stage cost 25-40 %
// (profiles/r01_f: 92 TF).  Here a wave owns RB x CB accumulator blocks of 16x16 (v_mfma_f32_16x16x4_f32: 32-cycle issue,
// 4 accumulator registers per block) -- 9 x 5 = 45 blocks = 180 registers of the 512 a wave has when it is alone on its
// SIMD -- and every 16-byte fragment load feeds 4 MFMA steps of a whole block row / column: 14 loads per 180 MFMAs.  At
// that ratio the loads need neither LDS staging nor sharing between waves: they come from L2 / L1 directly, the compiler
// counts them (vmcnt), and the only synchronisation in the kernel is the K-split reduction at the very end.
//
// Operand maps of v_mfma_f32_16x16x4_f32 (cdna_hip_programming.md, section 3): lane l = (r = l & 15, g = l >> 4) supplies
// A[i = r][k = g] and B[k = g][j = r]; D[row = 4 g + t][col = r] is register t.  The contraction index is free to be
// permuted as long as A and B agree: in step kb (0..3) of a 16-deep chunk lane group g supplies k = 4 g + kb, i.e. the
// four components of ONE 16-byte load of a K-contiguous row (NT form).  In the TN form (weight gradient, contraction over
// the rows of both operands) it is the OUTPUT column index that is permuted: lane r holds the 4 consecutive columns
// 4 r .. 4 r + 3 of a 64-column span as one 16-byte load, component c feeds accumulator block c, and the four blocks of a
// span come back together as 16-byte stores.
#pragma once
#include <type_traits>

#include "common.hpp"

namespace vqa {
namespace rt {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kThreads = 256;   // 4 waves, one per SIMD (the kernels need > 256 registers per lane: one workgroup per CU)

// Fragment loads are buffer loads: a wave-uniform 128-bit descriptor in SGPRs + a 32-bit per-lane byte offset + a scalar
// byte offset that carries the chunk / row advance.  No per-load address arithmetic on the VALU (next to fp32 MFMAs VALU
// instructions are NOT free: measured, each one costs its full issue time), and no 64-bit address registers.
using rsrc_t = __amdgpu_buffer_rsrc_t;
using u32x4 = __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, size_t bytes) {
  const unsigned int n = bytes > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned int)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, (int)n, 0x00020000);
}
__device__ __forceinline__ f32x4 ldg16(rsrc_t r, uint32_t off, uint32_t soff) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, (int)soff, 0));
}
__device__ __forceinline__ float ldg4(rsrc_t r, uint32_t off, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, (int)soff, 0));
}

// p = 0.5 dropout in registers (common.hpp: ONE BIT per element, bit e & 31 of the hash word of counter e >> 5).  `bits`
// holds the four mask bits of the float4 in its low nibble; the kept values are NOT scaled here -- the factor 2 is applied
// once per output element in the epilogue.  Two VALU instructions per element (v_bfe_i32 + v_and_b32).
__device__ __forceinline__ void keep4_bits(f32x4& x, uint32_t bits) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const uint32_t m = 0u - ((bits >> e) & 1u);      // all ones / all zeros (v_bfe_i32)
    const float xe = x[e];                           // (a bit_cast of the vector-element lvalue itself reads element 0)
    x[e] = __uint_as_float(__float_as_uint(xe) & m);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// NT form:  C[m][n] = sum_k A[m][k] * B[n][k]     A [M,K] row stride lda, B [N,K] row stride ldb (both K-contiguous)
//
// Workgroup tile (16 RB WM) x (16 CB WN); WK waves split the contraction range and are summed through LDS at the end
// (WM WN WK = 4).  `epi(row, col, value)` is called once per valid output element by the lane that holds it: lanes r = 0..15
// hold 16 consecutive columns of one row, register t the rows 4 g + t.
// DROP: A is masked by the p = 0.5 dropout of element (m, k) while it is in registers (needs K % 32 == 0: a row block's
// hash word then serves two consecutive chunks); the kept values' factor 2 is the epilogue's (`epi` receives acc * 2... the
// caller's functor applies dc.scale).
// ------------------------------------------------------------------------------------------------------------------
struct NtArgs {
  const float* A;
  const float* B;
  int lda, ldb;
  int M, N, K;
  int tiles_n;
  unsigned long long* stamps;   // diagnostic builds only (TUNE & 16): {shader cycles, 100 MHz ticks} of the main loop per wave
};

template <int RB, int CB, int WM, int WN, int WK>
struct NtShape {
  static_assert(WM * WN * WK == 4, "four waves per workgroup");
  static_assert(WK == 1 || WK == 2, "the contraction is split over at most two waves");
  static constexpr int BM = 16 * RB * WM, BN = 16 * CB * WN;
  static constexpr int NB = RB * CB;           // accumulator blocks per wave
  static constexpr int HALF = (NB + 1) / 2;    // blocks a wave hands to its partner in the K-split reduction
  static constexpr size_t kLdsBytes = WK == 2 ? (size_t)WM * WN * 2 * HALF * 64 * sizeof(f32x4) : 0;
};

// TUNE (experiments, tools/rt_probe.hip; 0 in the library): bit 0 = no loads inside the loop (MFMA-stream ceiling),
// bits 1-3 = MFMAs between two loads (0 = default spacing), bit 4 = stamp the main loop's clocks into p.stamps,
// bit 6 = the dropout hash words per lane instead of shared by ds_bpermute, bit 7 = the masks staggered over the row blocks.
template <int RB, int CB, int WM, int WN, int WK, bool DROP, class Epi, int TUNE = 0>
__global__ __launch_bounds__(kThreads, 1) void gemm_nt_kernel(NtArgs p, DropCfg dc, Epi epi) {
  using S = NtShape<RB, CB, WM, WN, WK>;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / p.tiles_n) * S::BM + wm * (16 * RB);
  const int n0 = (tile % p.tiles_n) * S::BN + wn * (16 * CB);

  // per-lane byte offsets of the fragment rows (clamped: loads are unconditional, out-of-range rows are never stored)
  uint32_t offA[RB], offB[CB];
  uint32_t wordA[RB];   // DROP: index of the hash word of (row, k = 0), to which c / 2 is added per pair of chunks
  uint32_t hw[RB];      // DROP: the hash words of the current pair of chunks
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int row = min(m0 + 16 * i + r, p.M - 1);
    offA[i] = ((uint32_t)row * (uint32_t)p.lda + 4u * g) * 4u;
    wordA[i] = ((uint32_t)row * (uint32_t)p.K) >> 5;
    hw[i] = 0u;
  }
  // DROP, shared hashes (round 4).  The four lanes (r, g = 0..3) of a row need the SAME hash word per pair of chunks and used
  // to compute it four times.  Now lane (r, g) hashes for the row blocks g, g + 4, g + 8 only and the words travel across the
  // lane groups by ds_bpermute (the LDS crossbar: no VALU slot, no LDS memory): 3 hashes + RB permutes per pair instead of RB
  // hashes.  (TUNE & 64 keeps the per-lane form for the A/B probe.)
  constexpr bool SHARE = DROP && (TUNE & 64) == 0;
  constexpr int NSEL = (RB + 3) / 4;
  uint32_t wordSel[NSEL];
  int permAddr[4];
#pragma unroll
  for (int k = 0; k < NSEL; ++k) wordSel[k] = ((uint32_t)min(m0 + 16 * min(g + 4 * k, RB - 1) + r, p.M - 1) * (uint32_t)p.K) >> 5;
#pragma unroll
  for (int q = 0; q < 4; ++q) permAddr[q] = 4 * (r + 16 * q);
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int col = min(n0 + 16 * j + r, p.N - 1);
    offB[j] = ((uint32_t)col * (uint32_t)p.ldb + 4u * g) * 4u;
  }
  const rsrc_t Ab = make_rsrc(p.A, ((size_t)(p.M - 1) * p.lda + p.K) * 4);
  const rsrc_t Bb = make_rsrc(p.B, ((size_t)(p.N - 1) * p.ldb + p.K) * 4);
  const uint32_t key = DROP ? drop_key(dc) : 0u;

  f32x4 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nfull = p.K >> 4;                       // whole 16-deep chunks
  const int per = ((nfull + WK - 1) / WK + 1) & ~1; // per wave: even, so every wave starts on an even chunk (DROP pairs them)
  const int c_lo = min(nfull, wk * per), c_hi = min(nfull, c_lo + per);

  auto load = [&](f32x4(&a)[RB], f32x4(&b)[CB], int c) {
    const uint32_t so = (uint32_t)c * 64u;
#pragma unroll
    for (int j = 0; j < CB; ++j) b[j] = ldg16(Bb, offB[j], so);
#pragma unroll
    for (int i = 0; i < RB; ++i) a[i] = ldg16(Ab, offA[i], so);
  };
  // Row-block-major MFMA order: a[i] is needed from MFMA 4 CB i on (its wait and, with DROP, its mask are staggered
  // over the chunk); an accumulator is revisited every CB MFMAs (>= 2 hides the 40-cycle dependent latency).
  auto mfma_row = [&](const f32x4& ai, f32x4(&b)[CB], int i) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int j = 0; j < CB; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(ai[kb], b[j][kb], acc[i][j], 0, 0, 0);
  };
  // DROP: chunk c covers the elements k = 16 c + 4 g .. + 3 of the lane's row: bits 16 (c & 1) + 4 g .. + 3 of the hash word
  // of counter (row K + 16 c) >> 5.  FRESH (c even): the word is computed; c odd: the one of c - 1 is reused.
  auto hash_pair = [&](int pair) {      // SHARE: the words of all RB row blocks for the pair of chunks 2 pair, 2 pair + 1
    uint32_t h[NSEL];
#pragma unroll
    for (int k = 0; k < NSEL; ++k) h[k] = mask_word32(wordSel[k] + (uint32_t)pair, key);
#pragma unroll
    for (int i = 0; i < RB; ++i) hw[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(permAddr[i & 3], (int)h[i >> 2]);
  };
  auto mask_row = [&](f32x4& ai, int i, int c, bool fresh) {
    if constexpr (SHARE) {
      // row block 0 of a pair is masked first (under the previous chunk's last MFMAs): that is where the pair's words are made
      if (fresh && i == 0) hash_pair(c >> 1);
    } else {
      if (fresh) hw[i] = mask_word32(wordA[i] + (uint32_t)(c >> 1), key);
    }
    keep4_bits(ai, hw[i] >> (4u * g + 16u * (uint32_t)(c & 1)));
  };
  auto compute = [&](f32x4(&a)[RB], f32x4(&b)[CB], int c) {   // one chunk on its own (c even)
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if constexpr (DROP) mask_row(a[i], i, c, true);
      mfma_row(a[i], b, i);
    }
  };
  // One pipeline step: the loads of chunk cn go out in the shadow of the first MFMAs of chunk c (one load per PER MFMAs),
  // so they have the rest of this chunk -- thousands of cycles -- to land.  Left to itself hipcc clusters the loads right
  // in front of their first use (it prices a global load at tens of cycles), which exposes the whole L2 / HBM latency.
  // DROP: the mask of row block i + 1 is applied in the shadow of row block i's 4 CB MFMAs, and the one of the NEXT
  // chunk's block 0 under this chunk's last block: on entry a[0] is masked already.  EVEN = parity of c (compile time):
  // the hash words are computed at their first use, which for block 0 of a pair is the odd step in front of it.
  // The VALU instructions are NOT pinned between the MFMAs: next to fp32 MFMAs they are not free (the matrix pipe and the
  // vector ALU share the FMA datapath), and what they cost is mostly per interrupted MFMA-to-MFMA hand-over (~10 cycles,
  // one VALU instruction or two) -- measured: spread one per MFMA, 118 instructions per chunk cost as much as 240 did;
  // the compiler's own placement, a burst of ~13 in front of each row block, interrupts the MFMA stream 9 times per chunk.
  auto step = [&](f32x4(&an)[RB], f32x4(&bn)[CB], int cn, f32x4(&a)[RB], f32x4(&b)[CB], int c, auto even) {
    constexpr bool EVEN = decltype(even)::value;
    if constexpr ((TUNE & 1) == 0) load(an, bn, cn);
    // Round 4: the masks of the chunk's row blocks 1 .. RB - 1 in ONE burst in front of its MFMAs (block 0 was masked under the
    // previous chunk's last block).  Staggered one row block ahead of its MFMAs -- the round-2 form, TUNE & 128 -- the same
    // instructions interrupt the MFMA stream RB times per chunk instead of once: 205.1 us against 194.6 at M = 18432, K = 2048,
    // N = 310 (no dropout: 177.5; tools/rt_probe.hip).
    constexpr bool BURST = DROP && (TUNE & 128) == 0;
    if constexpr (BURST) {
#pragma unroll
      for (int i = 1; i < RB; ++i) mask_row(a[i], i, c, EVEN);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      if constexpr (DROP) {
        if (i + 1 < RB) {
          if constexpr (!BURST) mask_row(a[i + 1], i + 1, c, EVEN);
        } else {
          mask_row(an[0], 0, cn, !EVEN);
        }
      }
      mfma_row(a[i], b, i);
    }
    constexpr int NL = (TUNE & 1) == 0 ? RB + CB : 0, NM = 4 * RB * CB;
    constexpr int PER = ((TUNE >> 1) & 7) != 0 ? ((TUNE >> 1) & 7) : (NM / (2 * (RB + CB)) > 0 ? NM / (2 * (RB + CB)) : 1);
    if constexpr (BURST) __builtin_amdgcn_sched_group_barrier(0x002, 8 * (RB - 1), 0);
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                          // MFMA
      if constexpr (DROP && (TUNE & 32) != 0) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);      // VALU (experiment)
      if (m % PER == PER - 1 && m / PER < NL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  {
    // Two named register sets.  The loop body is branch-free over PAIRS of chunks (even, odd): with a conditional second
    // half hipcc sinks that half's loads into the conditional block.  A last odd-numbered chunk follows on its own.
    f32x4 a0[RB], b0[CB], a1[RB], b1[CB];
    const int c_pairs = c_lo + ((c_hi - c_lo) & ~1);
    int c = c_lo;
    if (c < c_pairs) {
      load(a0, b0, c);
      if constexpr ((TUNE & 1) != 0) load(a1, b1, c + 1);
      if constexpr (DROP) mask_row(a0[0], 0, c, true);
      __builtin_amdgcn_sched_barrier(0);
      unsigned long long t0 = 0, r0 = 0;
      if constexpr ((TUNE & 16) != 0) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
      }
      for (; c < c_pairs; c += 2) {
        step(a1, b1, c + 1, a0, b0, c, std::true_type{});
        step(a0, b0, min(c + 2, c_pairs - 1), a1, b1, c + 1, std::false_type{});   // (last pair: a harmless reload)
      }
      if constexpr ((TUNE & 16) != 0) {
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0 && p.stamps != nullptr) {
          p.stamps[(size_t)(blockIdx.x * 4 + wave) * 2] = t1 - t0;
          p.stamps[(size_t)(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
        }
      }
    }
    if (c_pairs < c_hi) {          // (c_pairs is even: the lone chunk computes its own hash words)
      load(a0, b0, c_pairs);
      compute(a0, b0, c_pairs);
    }
  }
  if ((p.K & 15) != 0 && wk == WK - 1) {
    // K tail (< 16): per-component guarded loads, zero beyond K
    const int kbase = nfull * 16 + 4 * g;
    f32x4 a[RB], b[CB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kbase + e;
        const float v = ldg4(Ab, offA[i] - 16u * g + 4u * (uint32_t)min(k, p.K - 1), 0u);
        a[i][e] = k < p.K ? v : 0.f;
      }
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int k = kbase + e;
        const float v = ldg4(Bb, offB[j] - 16u * g + 4u * (uint32_t)min(k, p.K - 1), 0u);
        b[j][e] = k < p.K ? v : 0.f;
      }
    // (DROP requires K % 32 == 0: no tail)
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < CB; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][kb], b[j][kb], acc[i][j], 0, 0, 0);
  }

  auto finish = [&](int blk, f32x4 v) {   // blk = i * CB + j (compile-time after unrolling)
    const int i = blk / CB, j = blk % CB;
    const int col = n0 + 16 * j + r;
    if (col < p.N) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = m0 + 16 * i + 4 * g + t;
        if (row < p.M) epi(row, col, v[t]);
      }
    }
  };

  if constexpr (WK == 2) {
    // the two K halves of a tile: each wave hands the blocks it does not finish to its partner through LDS
    extern __shared__ __attribute__((aligned(16))) char rt_smem[];
    f32x4* red = reinterpret_cast<f32x4*>(rt_smem);
    f32x4* out_box = red + (size_t)((wmn * 2 + wk) * S::HALF) * 64 + lane;
    const f32x4* in_box = red + (size_t)((wmn * 2 + (wk ^ 1)) * S::HALF) * 64 + lane;
    if (wk == 0) {
#pragma unroll
      for (int blk = S::HALF; blk < S::NB; ++blk) out_box[(blk - S::HALF) * 64] = acc[blk / CB][blk % CB];
    } else {
#pragma unroll
      for (int blk = 0; blk < S::HALF; ++blk) out_box[blk * 64] = acc[blk / CB][blk % CB];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int blk = 0; blk < S::HALF; ++blk) finish(blk, acc[blk / CB][blk % CB] + in_box[blk * 64]);
    } else {
#pragma unroll
      for (int blk = S::HALF; blk < S::NB; ++blk) finish(blk, acc[blk / CB][blk % CB] + in_box[(blk - S::HALF) * 64]);
    }
  } else {
#pragma unroll
    for (int blk = 0; blk < S::NB; ++blk) finish(blk, acc[blk / CB][blk % CB]);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// TN form (weight gradient):  slab[s][n1][n2] = sum_{m in split s} P[m][n1] * Q[m][n2]
//   P [M,N1] row stride ldp (the output gradient; with MASK multiplied by (Y[m][n1] > 0), the relu gate read off the
//   saved forward output), Q [M,N2] row stride ldq (the layer input; with DROP multiplied by its dropout mask).
// Workgroup = 4 waves = 4 consecutive 16 RB-row groups of n1 (all of N1 = 310 for RB = 5) x one span group of 64 SP
// columns of n2 x one row split; the waves read the same Q rows (L1 serves three of the four).  dbslab[s][n1] = column sums
// of the gated P over the split (the bias gradient), written by the workgroups of the first n2 tile.
// ------------------------------------------------------------------------------------------------------------------
struct TnArgs {
  const float* P;
  const float* Y;
  const float* Q;
  float* slab;
  float* dbslab;
  int ldp, ldq;
  int M, N1, N2;
  int tiles1, tiles2;       // workgroup tiles along n1 (64 RB each) and n2 (64 SP each)
  int rows_per_split;       // multiple of 16
};

// SHARE (with DROP; needs N2 % 64 == 0 and SP <= 2): a hash word covers 32 columns of a row, i.e. the 8 lanes r & ~7 .. + 7 of a
// lane group need the same word for each of the 4 SP (contraction step, span) pairs of a chunk -- each of the 8 lanes hashes
// ONE of them and the words travel by ds_bpermute: 1 hash + 4 SP permutes per chunk instead of 4 SP hashes.
// TUNE (experiments): 1 = the VALU side of a chunk's steps 1..3 in ONE burst in front of its MFMAs instead of one per step.
template <int RB, int SP, bool MASK, bool DROP, bool SHARE = false, int TUNE = 0>
__global__ __launch_bounds__(kThreads, 1) void gemm_tn_kernel(TnArgs p, DropCfg dc) {
  static_assert(!SHARE || (DROP && 4 * SP <= 8), "SHARE: dropout on, at most two 64-column spans");
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.tiles1 * p.tiles2;
  const int s = bid / tiles, tile = bid % tiles;
  const int t2 = tile % p.tiles2, t1 = tile / p.tiles2;
  const int n1_0 = (t1 * 4 + wave) * (16 * RB);
  const int n2_0 = t2 * (64 * SP);
  const int m_lo = s * p.rows_per_split, m_hi = min(p.M, m_lo + p.rows_per_split);

  uint32_t offP[RB], offQ[SP], wordQ[SP];
#pragma unroll
  for (int i = 0; i < RB; ++i) offP[i] = ((uint32_t)(4 * g) * (uint32_t)p.ldp + (uint32_t)min(n1_0 + 16 * i + r, p.N1 - 1)) * 4u;
#pragma unroll
  for (int q = 0; q < SP; ++q) {
    const uint32_t col = (uint32_t)min(n2_0 + 64 * q + 4 * r, p.N2 - 4);
    offQ[q] = ((uint32_t)(4 * g) * (uint32_t)p.ldq + col) * 4u;
    wordQ[q] = (uint32_t)(4 * g) * (uint32_t)p.N2 + col;      // element index of (row 4 g, col) -- a multiple of 4
  }
  const rsrc_t Pb = make_rsrc(p.P, ((size_t)(p.M - 1) * p.ldp + p.N1) * 4);
  const rsrc_t Yb = make_rsrc(MASK ? p.Y : p.P, ((size_t)(p.M - 1) * p.ldp + p.N1) * 4);
  const rsrc_t Qb = make_rsrc(p.Q, ((size_t)(p.M - 1) * p.ldq + p.N2) * 4);
  const uint32_t key = DROP ? drop_key(dc) : 0u;

  f32x4 acc[RB][SP][4];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int q = 0; q < SP; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) acc[i][q][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbp[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) dbp[i] = 0.f;

  struct Set {
    float a[4][RB];
    float y[4][MASK ? RB : 1];
    f32x4 q[4][SP];
    uint32_t hw[SHARE ? 4 : 1][SHARE ? SP : 1];   // SHARE: the hash words of the chunk's (step, span) pairs
  };
  // SHARE: lane t = r & 7 of its group of eight hashes for step t / SP, span t % SP; its 4 columns sit at bit 4 (r & 7)
  const int t8 = r & 7;
  const uint32_t share_col = (uint32_t)n2_0 + 64u * (uint32_t)(t8 % SP) + 32u * (uint32_t)(r >> 3);
  const uint32_t share_row = (uint32_t)(4 * g + t8 / SP);
  const uint32_t share_shift = 4u * (uint32_t)t8;
  const int share_addr = 4 * (lane & ~7);
  auto share_words = [&](Set& st, int m_c) {
    if constexpr (SHARE) {
      const uint32_t e = ((uint32_t)m_c + share_row) * (uint32_t)p.N2 + share_col;
      const int h = (int)mask_word32(e >> 5, key);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int q = 0; q < SP; ++q) st.hw[kb][q] = (uint32_t)__builtin_amdgcn_ds_bpermute(share_addr + 4 * (kb * SP + q), h);
    }
  };
  // rows m_c + 4 g + kb of the chunk that starts at row m_c (all < m_hi: whole chunks only)
  auto load = [&](Set& st, int m_c) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const uint32_t sp = (uint32_t)(m_c + kb) * (uint32_t)p.ldp * 4u;
      const uint32_t sq = (uint32_t)(m_c + kb) * (uint32_t)p.ldq * 4u;
#pragma unroll
      for (int i = 0; i < RB; ++i) st.a[kb][i] = ldg4(Pb, offP[i], sp);
      if constexpr (MASK) {
#pragma unroll
        for (int i = 0; i < RB; ++i) st.y[kb][i] = ldg4(Yb, offP[i], sp);
      }
#pragma unroll
      for (int q = 0; q < SP; ++q) st.q[kb][q] = ldg16(Qb, offQ[q], sq);
    }
  };
  // VALU side of contraction step kb of a chunk: relu gate, dropout mask
  auto prep = [&](Set& st, int kb, int m_c) {
    if constexpr (MASK) {
#pragma unroll
      for (int i = 0; i < RB; ++i) st.a[kb][i] = st.y[kb][i] > 0.f ? st.a[kb][i] : 0.f;
    }
    if constexpr (SHARE) {
#pragma unroll
      for (int q = 0; q < SP; ++q) keep4_bits(st.q[kb][q], st.hw[kb][q] >> share_shift);
    } else if constexpr (DROP) {     // p = 0.5 bit mask of the four columns (unscaled: the factor 2 is the epilogue's)
      const uint32_t erow = (uint32_t)(m_c + kb) * (uint32_t)p.N2;
#pragma unroll
      for (int q = 0; q < SP; ++q) {
        const uint32_t e = erow + wordQ[q];
        keep4_bits(st.q[kb][q], mask_word32(e >> 5, key) >> (e & 31u));
      }
    }
  };
  auto mfmas = [&](Set& st, int kb) {   // (the bias-gradient sums live here: a prepared-ahead step may belong to a dummy reload)
#pragma unroll
    for (int i = 0; i < RB; ++i) dbp[i] += st.a[kb][i];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int q = 0; q < SP; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[i][q][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.a[kb][i], st.q[kb][q][e], acc[i][q][e], 0, 0, 0);
  };
  auto compute = [&](Set& st, int m_c) {   // one chunk on its own
    share_words(st, m_c);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      prep(st, kb, m_c);
      mfmas(st, kb);
    }
  };
  // One pipeline step: chunk m_n is loaded into `sn` under the MFMAs of chunk m_c (see gemm_nt_kernel).  The VALU side
  // of step kb + 1 runs in the shadow of step kb's MFMAs, that of the next chunk's step 0 under this chunk's step 3: on
  // entry step 0 of `st` is prepared already.
  auto step = [&](Set& sn, int m_n, Set& st, int m_c) {
    load(sn, m_n);
    constexpr bool BURST = (TUNE & 1) != 0 && (MASK || DROP);
    if constexpr (BURST) {
#pragma unroll
      for (int kb = 1; kb < 4; ++kb) prep(st, kb, m_c);
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      if (kb + 1 < 4) {
        if constexpr (!BURST) prep(st, kb + 1, m_c);
      } else {
        share_words(sn, m_n);
        prep(sn, 0, m_n);
      }
      mfmas(st, kb);
    }
    constexpr int NL = 4 * (RB * (MASK ? 2 : 1) + SP), NM = 16 * RB * SP;
    constexpr int PER = (3 * NM) / (4 * NL) > 0 ? (3 * NM) / (4 * NL) : 1;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
      if (m % PER == PER - 1 && m / PER < NL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nfull = (m_hi - m_lo) >> 4;
  {
    Set s0, s1;   // branch-free over pairs of chunks (see gemm_nt_kernel)
    int c = 0;
    if ((nfull & 1) != 0) {
      load(s0, m_lo);
      compute(s0, m_lo);
      ++c;
    }
    if (c < nfull) {
      load(s0, m_lo + 16 * c);
      share_words(s0, m_lo + 16 * c);
      prep(s0, 0, m_lo + 16 * c);
      __builtin_amdgcn_sched_barrier(0);
      for (; c < nfull; c += 2) {
        step(s1, m_lo + 16 * (c + 1), s0, m_lo + 16 * c);
        step(s0, m_lo + 16 * min(c + 2, nfull - 1), s1, m_lo + 16 * (c + 1));
      }
    }
  }
  if (((m_hi - m_lo) & 15) != 0) {
    // row tail of the split: clamped loads, rows >= m_hi contribute zero through the P side
    const int m_c = m_lo + 16 * nfull;
    Set st;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int row = m_c + 4 * g + kb;
      const bool ok = row < m_hi;
      const uint32_t rc = (uint32_t)min(row, m_hi - 1);
      const uint32_t sp = rc * (uint32_t)p.ldp * 4u, sq = rc * (uint32_t)p.ldq * 4u;
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const uint32_t o = offP[i] - (uint32_t)(4 * g) * (uint32_t)p.ldp * 4u;
        float v = ldg4(Pb, o + sp, 0u);
        if constexpr (MASK) v = ldg4(Yb, o + sp, 0u) > 0.f ? v : 0.f;
        st.a[kb][i] = ok ? v : 0.f;
        if constexpr (MASK) st.y[kb][i] = 1.f;
      }
#pragma unroll
      for (int q = 0; q < SP; ++q) {
        const uint32_t o = offQ[q] - (uint32_t)(4 * g) * (uint32_t)p.ldq * 4u;
        f32x4 v = ldg16(Qb, o + sq, 0u);
        if constexpr (DROP) {
          const uint32_t e = rc * (uint32_t)p.N2 + (o >> 2);
          keep4_bits(v, mask_word32(e >> 5, key) >> (e & 31u));
        }
        st.q[kb][q] = v;
      }
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int i = 0; i < RB; ++i) dbp[i] += st.a[kb][i];
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int q = 0; q < SP; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[i][q][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.a[kb][i], st.q[kb][q][e], acc[i][q][e], 0, 0, 0);
    }
  }

  const float qscale = DROP ? dc.scale : 1.f;      // the kept values of Q were not scaled in the loop
  float* __restrict__ dst = p.slab + (size_t)s * p.N1 * p.N2;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int q = 0; q < SP; ++q) {
      const int col = n2_0 + 64 * q + 4 * r;
      if (col < p.N2) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int row = n1_0 + 16 * i + 4 * g + t;
          if (row < p.N1)
            *reinterpret_cast<f32x4*>(dst + (size_t)row * p.N2 + col) =
                f32x4{acc[i][q][0][t], acc[i][q][1][t], acc[i][q][2][t], acc[i][q][3][t]} * qscale;
        }
      }
    }
  if (p.dbslab != nullptr && t2 == 0) {
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      float v = dbp[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n1 = n1_0 + 16 * i + r;
      if (g == 0 && n1 < p.N1) p.dbslab[(size_t)s * p.N1 + n1] = v;
    }
  }
}

}  // namespace rt
}  // namespace vqa
