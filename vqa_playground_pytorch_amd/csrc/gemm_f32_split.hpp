// fp32 GEMM on the bf16 matrix pipe: every fp32 operand is split EXACTLY into three bf16 terms x = x0 + x1 + x2 (8 + 8 + 8
// significant bits, round-to-nearest at each level, so the residuals are zero-mean) and a product a b is formed from the six
// partial products whose weight is >= 2^-16 of it:  a0 b0 + (a0 b1 + a1 b0) + (a1 b1 + a0 b2 + a2 b0), each exact in the
// MFMA's fp32 accumulator arithmetic (a product of two 8-bit significands has 16 bits).  What is dropped -- a1 b2 + a2 b1 +
// a2 b2 -- is bounded by 2^-24 |a b| (|x1| <= 2^-8 |x|, |x2| <= 2^-17 |x|): below half an ulp of the fp32 product itself.
// The accumulation is the MFMA's fp32 accumulation, as in gemm_f32_rt.hpp.  So the result carries the rounding of an fp32
// GEMM (tests/test_gpu_split.py holds both engines against float64 at M = 18432, K = 2048: rms error 4-6e-7 of the result's
// rms on either, largest error equal), but the multiplies run on v_mfma_f32_16x16x32_bf16 (1024 FLOP/clk/SIMD) instead of
// v_mfma_f32_16x16x4_f32 (64): six bf16 MFMAs replace eight fp32 ones of twice the issue time per 16x16x32 block step,
// 2.67 x fewer matrix-pipe cycles.  (The chip clocks bf16 MFMA loops lower -- 1.5-1.7 GHz against 1.9-2.1 -- so what arrives is
// about 1.7 x: tools/split_probe.hip.)
//
// Engine shape = gemm_f32_rt.hpp's: ONE wave per SIMD, RB x CB accumulator blocks of 16x16 per wave, fragments straight from
// L2 by buffer loads, no LDS in the main loop, a K split over two waves reduced through LDS at the end.
//   A [M,K] fp32 (the activations: split in registers, 9 VALU instructions per pair of elements, issued in the shadow of the
//     previous row block's MFMAs -- a v_mfma_f32_16x16x32_bf16 holds the vector issue port for 8 of its 16 cycles);
//   B [N,K] pre-split ONCE per step into a packed plane image (`split_pack_kernel`): the weights are small and every
//     workgroup re-reads them, so splitting them in every wave would double the VALU work for nothing.
//
// Contraction order inside a 32-deep chunk: lane group g = lane >> 4 supplies the 8 values k = 4 g .. 4 g + 3 and
// 16 + 4 g .. 16 + 4 g + 3 (two 16-byte loads of a K-contiguous fp32 row; each load instruction covers a contiguous 64 bytes
// per row).  The packed B image stores exactly those 8 values of a plane as one 16-byte group, in the order the lanes of a wave
// read them, so that a fragment load is ONE contiguous KiB (8 whole cache lines instead of 16-32 partly used ones):
//   Bp[n / 16][chunk][plane][lane = (n % 16) + 16 g][8 bf16]      (rows padded to a multiple of 16 with zeros)
#pragma once
#include <type_traits>

#include "gemm_f32_rt.hpp"

namespace vqa {
namespace sp {

using rt::f32x4;
using rt::ldg16;
using rt::make_rsrc;
using rt::rsrc_t;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int kChunk = 32;            // contraction depth of one v_mfma_f32_16x16x32_bf16
constexpr int kPackedChunkBytes = 3 * 1024;   // one 16-row block's chunk in the packed image: 3 planes x 64 lanes x 16 bytes

// x0, x1 -> the three bf16 planes of both, packed (x0 in the low half): 9 VALU instructions.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t opaque(uint32_t v) {   // (keeps hipcc from re-deriving the halves of a pack by a second convert)
  asm("" : "+v"(v));
  return v;
}
__device__ __forceinline__ uint32_t pack2(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
// x - (the bf16 pair p widened).  Default: shift / and / packed subtract, three instructions per pair.  DOT: v_dot2c_f32_bf16 with
// the constant pairs (-1, 0) and (0, -1), one instruction per element (x + p.lo * -1 + p.hi * 0); the difference is exact (it is
// representable: p is x rounded to 8 significant bits) and tools/split_probe.hip checks the planes come out bit for bit the
// same -- but the kernel runs 8 % slower with it (the dot instruction is not a full-rate one), so it stays a probe option.
template <bool DOT>
__device__ __forceinline__ f32x2 residual(f32x2 x, uint32_t p) {
  if constexpr (DOT) {
    // (the constant pairs come through registers: written as literals hipcc folds (-1, 0) into the INLINE constant -1.0, which
    //  the instruction reads as the f32 pattern 0xBF800000 = the pair (0, -1))
    const bf16x2 pv = __builtin_bit_cast(bf16x2, p);
    const bf16x2 m0 = __builtin_bit_cast(bf16x2, opaque(0x0000BF80u)), m1 = __builtin_bit_cast(bf16x2, opaque(0xBF800000u));
    return f32x2{__builtin_amdgcn_fdot2_f32_bf16(pv, m0, x[0], false), __builtin_amdgcn_fdot2_f32_bf16(pv, m1, x[1], false)};
  } else {
    const f32x2 w = {bf16_lo(p), bf16_hi(p)};
    return x - w;
  }
}
// a pair of fp32 -> its three bf16 planes, packed (element 0 in the low half): 7 VALU instructions (9 without the dot form)
template <bool DOT>
__device__ __forceinline__ void split_pair(f32x2 x, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
  p0 = opaque(pack2(x));
  const f32x2 r = residual<DOT>(x, p0);   // |r| <= 2^-9 |x|
  p1 = opaque(pack2(r));
  const f32x2 t = residual<DOT>(r, p1);   // |t| <= 2^-18 |x|: at most 8 significant bits remain, the last pack is exact
  p2 = pack2(t);
}
struct Planes {
  u32x4 p[3];
};
template <bool DOT = true>
__device__ __forceinline__ void split8(const f32x4& lo, const f32x4& hi, Planes& o) {
  uint32_t w[3][4];
  split_pair<DOT>(__builtin_shufflevector(lo, lo, 0, 1), w[0][0], w[1][0], w[2][0]);
  split_pair<DOT>(__builtin_shufflevector(lo, lo, 2, 3), w[0][1], w[1][1], w[2][1]);
  split_pair<DOT>(__builtin_shufflevector(hi, hi, 0, 1), w[0][2], w[1][2], w[2][2]);
  split_pair<DOT>(__builtin_shufflevector(hi, hi, 2, 3), w[0][3], w[1][3], w[2][3]);
#pragma unroll
  for (int q = 0; q < 3; ++q) o.p[q] = u32x4{w[q][0], w[q][1], w[q][2], w[q][3]};
}
__device__ __forceinline__ f32x4 mfma_bf16(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// ---- the engine's domain is ALL of fp32: the repair path ------------------------------------------------------------------
// The three-way split is exact for every finite x whose leading plane is finite.  Outside of that -- x = +-Inf or NaN, or
// |x| within half a bf16 ulp of FLT_MAX (plane 0 rounds to Inf) -- plane 1 is Inf - Inf = NaN, and a zero plane of the other
// operand meeting an Inf plane is NaN as well (this includes the zero padding of a contraction: K = 320 over rows of 310).  In
// every such case the affected ACCUMULATORS come out non-finite (NaN and Inf survive any further accumulation), so the
// epilogues test their outputs (7 VALU instructions per four) and recompute a non-finite one as a plain fp32 dot product of
// the ORIGINAL operands: an fp32 GEMM's answer -- finite where the fp32 MFMA engine's is finite (inputs next to FLT_MAX),
// +-Inf / NaN where the data says so.  The path is never taken on finite well-scaled data (one execz branch per four outputs);
// on a tensor full of NaNs it makes the kernel slow, not wrong.  tests/test_gpu_split.py::test_split_engine_edge_values.
__device__ __forceinline__ uint32_t abs_bits_max(const f32x4& v) {
  const u32x4 b = __builtin_bit_cast(u32x4, v) & 0x7FFFFFFFu;
  return max(max(b[0], b[1]), max(b[2], b[3]));
}
__device__ __forceinline__ bool any_nonfinite(const f32x4& v) { return abs_bits_max(v) >= 0x7F800000u; }
__device__ __forceinline__ bool nonfinite(float v) { return (__float_as_uint(v) & 0x7FFFFFFFu) >= 0x7F800000u; }

// ---- the packed plane image of a [N, K] fp32 matrix (K % 32 == 0) -------------------------------------------------
// one thread per (row block, chunk, lane): 8 fp32 in, 3 x 16 bytes out
template <bool DOT = false>
__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, int ldw, int N, int K,
                                                         u32x4* __restrict__ out) {
  const int chunks = K / kChunk;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int nblocks = (N + 15) / 16;
  if (t >= (long)nblocks * chunks * 64) return;
  const int lane = (int)(t & 63), r = lane & 15, g = lane >> 4;
  const long bc = t >> 6;
  const int c = (int)(bc % chunks), n = (int)(bc / chunks) * 16 + r;
  f32x4 lo = f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
  if (n < N) {
    const float* src = w + (size_t)n * ldw + c * kChunk + 4 * g;
    lo = *reinterpret_cast<const f32x4*>(src);
    hi = *reinterpret_cast<const f32x4*>(src + 16);
  }
  Planes pl;
  split8<DOT>(lo, hi, pl);
  u32x4* dst = out + (size_t)bc * 192 + lane;
  dst[0] = pl.p[0];
  dst[64] = pl.p[1];
  dst[128] = pl.p[2];
}
__host__ __device__ inline size_t packed_bytes(int N, int K) { return (size_t)((N + 15) / 16) * (K / kChunk) * kPackedChunkBytes; }

// ------------------------------------------------------------------------------------------------------------------
// NT form:  C[m][n] = sum_k A[m][k] * B[n][k]     A [M,K] fp32 row stride lda; B as the packed plane image of [N,K]
// Workgroup tile (16 RB WM) x (16 CB WN), WK waves split the contraction range (WM WN WK = 4).  K % 64 == 0.
// DROP: A masked by the p = 0.5 dropout of element (m, k) before it is split (one hash word per row and chunk: 32 elements).
// ------------------------------------------------------------------------------------------------------------------
struct NtArgs {
  const float* A;
  const u32x4* Bp;
  int lda;
  int M, N, K;
  int tiles_n;
  const float* Bf = nullptr;   // the fp32 original of the packed image, [N, K] row stride ldb: operand of the repair path
  int ldb = 0;                 // (nullptr: the kernel's caller repairs in its own epilogue, or not at all)
  // REL (K1 -> K5 fused, config/CoR2.py:215-218): the operand is not A but the relation step's output of it,
  //   x[m][k] = T[s][k] + C2[s][k] * A[m][k],   s = m / rps   (T, C2 [M / rps, K] fp32, dense),
  // formed in registers from the staged rows of T / C2 before the mask and the split: v2 is never written to HBM.
  const float* T = nullptr;
  const float* C2 = nullptr;
  int rps = 0;                 // rows per sample
};
// REL: samples a BM-row tile can touch, and the LDS bytes of their staged (T, C2) rows: [sample][T | C2][K] floats
__host__ __device__ inline int rel_samples(int BM, int rps) { return (BM - 1) / rps + 2; }
__host__ __device__ inline size_t rel_lds_bytes(int BM, int rps, int K) { return (size_t)rel_samples(BM, rps) * 2 * K * 4; }
// the repair path of the NT form: C[row][col] as an fp32 dot product of the original operands (DROP: the same mask bits)
template <bool DROP>
__device__ __forceinline__ float nt_repair(const float* A, int lda, const float* Bf, int ldb, int K, int row, int col, uint32_t key,
                                           const float* T = nullptr, const float* C2 = nullptr, int rps = 1) {
  const float* a = A + (size_t)row * lda;
  const float* b = Bf + (size_t)col * ldb;
  float s = 0.f;
  for (int k = 0; k < K; ++k) {
    float x = a[k];
    if (T != nullptr) x = fmaf(C2[(size_t)(row / rps) * K + k], x, T[(size_t)(row / rps) * K + k]);
    if constexpr (DROP) {
      const uint32_t e = (uint32_t)row * (uint32_t)K + (uint32_t)k;
      x = ((mask_word32(e >> 5, key) >> (e & 31u)) & 1u) != 0u ? x : 0.f;
    }
    s = fmaf(x, b[k], s);
  }
  return s;
}

// NR = raw A row blocks in flight per wave (a ring: the load of row block n + NR goes out when n has been split; 2 RB % NR == 0).
// TUNE (tools/split_probe.hip): bit 0 = no A split (the planes are the raw registers: MFMA + load ceiling, wrong results),
// bit 1 = no A loads in the loop, bits 2-3 = VALU instructions pinned per MFMA (0 = the default 2), bit 4 = column-major tile
// order (an XCD then works on one column tile, but A is fetched once per column tile), bit 5 = no B loads in the loop,
// bit 6 = residuals by v_dot2c_f32_bf16 (bit-identical planes, 7 instead of 9 VALU instructions per pair -- and 8 % SLOWER:
// 129.7 against 119.5 us, tools/split_probe.hip), bit 7 = nt policy on the A loads (162 against 117 us)
// The main loop of the NT form as a device function: acc[i][j] += the contraction over the chunks [c_lo, c_hi) (an even count) of
// rows m0 + 16 i .. of A against the column blocks (n0 >> 4) + j of the packed image.  a_bytes = the extent of A the loads may
// touch (beyond it they read zeros: a contraction padded past the row end -- K = 320 over rows of 310 -- leans on that and on
// the image's zero planes there).
// REL: `rel_lds` = the staged (T, C2) rows of the samples from `rel_s0` on (rel_stage), read back one row block ahead of the
// block's split (two register sets: an LDS read issued in the region that needs it would stall the wave's whole issue stream).
template <int RB, int CB, bool DROP, int TUNE, int NR, bool REL = false>
__device__ __forceinline__ void nt_accumulate(const NtArgs& p, const DropCfg& dc, size_t a_bytes, int m0, int n0, int c_lo, int c_hi,
                                              f32x4 (&acc)[RB][CB], const char* rel_lds = nullptr, int rel_s0 = 0) {
  static_assert((2 * RB) % NR == 0 && NR >= 2 && NR <= 2 * RB, "the ring must divide a pair of chunks");
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int chunks = p.K / kChunk;
  uint32_t offA[RB], offB[CB], wordA[RB];
  uint32_t offT[REL ? RB : 1];     // REL: byte offset of (this lane's sample in row block i, k = 4 g) in the staged image
#pragma unroll
  for (int i = 0; i < RB; ++i) {
    const int row = min(m0 + 16 * i + r, p.M - 1);
    offA[i] = ((uint32_t)row * (uint32_t)p.lda + 4u * g) * 4u;
    wordA[i] = ((uint32_t)row * (uint32_t)p.K) >> 5;
    if constexpr (REL) offT[i] = ((uint32_t)(row / p.rps - rel_s0) * 2u * (uint32_t)p.K + 4u * g) * 4u;
  }
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int blk = min((n0 >> 4) + j, (p.N + 15) / 16 - 1);     // (n0 is a multiple of 16; rows past N are zeros in the image)
    offB[j] = (uint32_t)blk * (uint32_t)chunks * (uint32_t)kPackedChunkBytes + 16u * lane;
  }
  const rsrc_t Ab = make_rsrc(p.A, a_bytes);
  const rsrc_t Bb = make_rsrc(p.Bp, packed_bytes(p.N, p.K));
  const uint32_t key = DROP ? drop_key(dc) : 0u;

  struct ARaw {
    f32x4 lo, hi;
  };
  auto loadA = [&](ARaw& a, int i, int c) {
    const uint32_t so = (uint32_t)c * 128u;
    constexpr int AUX = (TUNE & 128) ? 2 : 0;   // (experiment: nt = streaming policy for the activations)
    a.lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(Ab, (int)offA[i], (int)so, AUX));
    a.hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(Ab, (int)(offA[i] + 64u), (int)so, AUX));
  };
  auto loadB = [&](Planes(&b)[CB], int c) {
    const uint32_t so = (uint32_t)c * (uint32_t)kPackedChunkBytes;
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q)
        b[j].p[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(Bb, (int)(offB[j] + 1024u * q), (int)so, 0));
  };
  // the planes of row block i of chunk c from its raw registers (DROP: masked first -- lane (r, g) holds the elements
  // k = 32 c + 4 g .. + 3 and 32 c + 16 + 4 g .. + 3 of its row: bits 4 g .. and 16 + 4 g .. of the row's hash word of the chunk)
  struct TC {
    f32x4 tlo, thi, clo, chi;
  };
  TC tc[REL ? 2 : 1];
  auto loadTC = [&](TC& o, int i, int c) {     // the (T, C2) values of row block i's lanes for chunk c, from LDS
    if constexpr (REL) {
      const char* base = rel_lds + offT[i] + (uint32_t)c * 128u;
      o.tlo = *reinterpret_cast<const f32x4*>(base);
      o.thi = *reinterpret_cast<const f32x4*>(base + 64);
      o.clo = *reinterpret_cast<const f32x4*>(base + (size_t)p.K * 4);
      o.chi = *reinterpret_cast<const f32x4*>(base + (size_t)p.K * 4 + 64);
    }
  };
  auto prepare = [&](const ARaw& a, Planes& pl, int i, int c, int n = 0) {
    if constexpr ((TUNE & 1) != 0) {
      pl.p[0] = __builtin_bit_cast(u32x4, a.lo);
      pl.p[1] = __builtin_bit_cast(u32x4, a.hi);
      pl.p[2] = __builtin_bit_cast(u32x4, a.lo);
    } else {
      ARaw m = a;
      if constexpr (REL) {      // x = T + C2 * a (two v_pk_fma_f32 per four), before the mask and the split
        const TC& q = tc[n & 1];
        m.lo = __builtin_elementwise_fma(q.clo, m.lo, q.tlo);
        m.hi = __builtin_elementwise_fma(q.chi, m.hi, q.thi);
      }
      if constexpr (DROP) {
        const uint32_t w = mask_word32(wordA[i] + (uint32_t)c, key);
        rt::keep4_bits(m.lo, w >> (4u * g));
        rt::keep4_bits(m.hi, w >> (4u * g + 16u));
      }
      split8<(TUNE & 64) != 0>(m.lo, m.hi, pl);
    }
  };
  // 6 CB MFMAs of a row block; an accumulator is revisited every CB MFMAs
  auto mfma_row = [&](const Planes& a, const Planes(&b)[CB], int i) {
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = mfma_bf16(a.p[PA[q]], b[j].p[PB[q]], acc[i][j]);
  };

  // Row blocks are numbered n = P RB + i over a PAIR of chunks (P = parity of the chunk inside the pair); raw row block n lives in
  // ring slot n % NR and its planes in plane set n & 1.  Step n: split row block n + 1 (under the MFMAs of n), send the load of
  // n + 1 + NR into the slot just freed, issue the 6 CB MFMAs of n.  B: the planes of the next chunk load at the start of a chunk.
  ARaw a[NR];
  Planes b0[CB], b1[CB], pl[2];
  auto fetch = [&](int n, int c_pair) {   // raw row block n (may run past the pair: n >= 2 RB wraps into the next pairs)
    const int c = min(c_pair + n / RB, c_hi - 1);
    loadA(a[n % NR], n % RB, c);
  };
  auto loadB1 = [&](Planes(&b)[CB], int idx, int c) {   // plane idx % 3 of column block idx / 3
    b[idx / 3].p[idx % 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
        Bb, (int)(offB[idx / 3] + 1024u * (idx % 3)), (int)((uint32_t)c * (uint32_t)kPackedChunkBytes), 0));
  };
  // Every row block is its own scheduling region: its loads are consumed in LATER regions, so hipcc cannot sink them down to
  // their first use (where it puts a load it believes costs tens of cycles) -- they go out where the group barriers place them.
  auto chunk = [&](Planes(&b)[CB], Planes(&bn)[CB], int c_pair, auto parity) {
    constexpr int P = decltype(parity)::value;
    constexpr int LB = (3 * CB + RB - 1) / RB;   // B plane loads per row block
    const int cb = min(c_pair + P + 1, c_hi - 1);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = P * RB + i;
      prepare(a[(n + 1) % NR], pl[(n + 1) & 1], (n + 1) % RB, min(c_pair + (n + 1) / RB, c_hi - 1), n + 1);
      if constexpr (REL) loadTC(tc[n & 1], (n + 2) % RB, min(c_pair + (n + 2) / RB, c_hi - 1));   // (row block n's set is free)
      if constexpr ((TUNE & 2) == 0) fetch(n + 1 + NR, c_pair);
      int nb = 0;
      if constexpr ((TUNE & 32) == 0) {
#pragma unroll
        for (int idx = i * LB; idx < (i + 1) * LB && idx < 3 * CB; ++idx, ++nb) loadB1(bn, idx, cb);
      }
      mfma_row(pl[n & 1], b, i);
      constexpr int NM = 6 * CB;
      constexpr int VPM = ((TUNE >> 2) & 3) != 0 ? ((TUNE >> 2) & 3) : 2;
      const int nl = nb + ((TUNE & 2) == 0 ? 2 : 0);
      const int per = nl > 0 ? NM / nl : NM;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
        if (REL && m < 4) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read: (T, C2) of the row block after next
        __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);   // VALU
        if (nl > 0 && m % per == per - 1 && m / per < nl) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (c_lo < c_hi) {
    loadB(b0, c_lo);
#pragma unroll
    for (int n = 0; n < NR; ++n) fetch(n, c_lo);
    if constexpr (REL) {
      loadTC(tc[0], 0, c_lo);
      loadTC(tc[1], 1 % RB, min(c_lo + 1 / RB, c_hi - 1));
    }
    prepare(a[0], pl[0], 0, c_lo, 0);
    fetch(NR, c_lo);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = c_lo; c < c_hi; c += 2) {
      chunk(b0, b1, c, std::integral_constant<int, 0>{});
      chunk(b1, b0, c, std::integral_constant<int, 1>{});
    }
  }

}

// nt_accumulate for a workgroup whose NW waves work on the SAME rows of A (different column blocks, the whole contraction each):
// instead of every wave loading and splitting all RB row blocks, wave w loads and splits the row blocks w, w + NW, ... of the
// NEXT chunk under this chunk's MFMAs and writes their planes into LDS (2 buffers x RB x 3 KiB); all waves read every row
// block's planes back, one row block ahead of its MFMAs.  One workgroup barrier per chunk, in front of the last row block.
// `lds`: 2 * RB * 3 * 1024 bytes.  Chunks [c_lo, c_hi), an even count.
// DROP: A masked by the p = 0.5 dropout of element (m, k) before it is split, as in nt_accumulate.  All waves that meet at the
// barrier must run the same number of chunks.
template <int RB, int CB, int NW, bool DROP, int TUNE>
__device__ __forceinline__ void nt_accumulate_shared(const NtArgs& p, const DropCfg& dc, size_t a_bytes, int m0, int n0, int c_lo,
                                                     int c_hi, int wave, char* lds_bytes, f32x4 (&acc)[RB][CB]) {
  constexpr int MINE = (RB + NW - 1) / NW;     // row blocks a wave produces at most
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int chunks = p.K / kChunk;
  u32x4* const lds = reinterpret_cast<u32x4*>(lds_bytes);   // [buffer][row block][plane][lane]
  uint32_t offA[MINE], offB[CB], wordA[MINE];
#pragma unroll
  for (int k = 0; k < MINE; ++k) {
    const int i = min(wave + NW * k, RB - 1);
    const int row = min(m0 + 16 * i + r, p.M - 1);
    offA[k] = ((uint32_t)row * (uint32_t)p.lda + 4u * g) * 4u;
    wordA[k] = ((uint32_t)row * (uint32_t)p.K) >> 5;
  }
  const uint32_t key = DROP ? drop_key(dc) : 0u;
#pragma unroll
  for (int j = 0; j < CB; ++j) {
    const int blk = min((n0 >> 4) + j, (p.N + 15) / 16 - 1);
    offB[j] = (uint32_t)blk * (uint32_t)chunks * (uint32_t)kPackedChunkBytes + 16u * lane;
  }
  const rsrc_t Ab = make_rsrc(p.A, a_bytes);
  const rsrc_t Bb = make_rsrc(p.Bp, packed_bytes(p.N, p.K));
  struct ARaw {
    f32x4 lo, hi;
  };
  ARaw raw0[MINE], raw1[MINE];
  Planes b0[CB], b1[CB], xp[2];
  auto loadA = [&](ARaw(&raw)[MINE], int c) {
    const uint32_t so = (uint32_t)min(c, c_hi - 1) * 128u;
#pragma unroll
    for (int k = 0; k < MINE; ++k) {
      raw[k].lo = ldg16(Ab, offA[k], so);
      raw[k].hi = ldg16(Ab, offA[k] + 64u, so);
    }
  };
  auto loadB1 = [&](Planes(&b)[CB], int idx, int c) {
    b[idx / 3].p[idx % 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
        Bb, (int)(offB[idx / 3] + 1024u * (idx % 3)), (int)((uint32_t)min(c, c_hi - 1) * (uint32_t)kPackedChunkBytes), 0));
  };
  auto produce = [&](const ARaw& a, int k, int buf, int c) {     // row block wave + NW k of chunk c (uniform guard)
    if (wave + NW * k < RB) {
      Planes pl;
      ARaw m = a;
      if constexpr (DROP) {
        const uint32_t w = mask_word32(wordA[k] + (uint32_t)min(c, c_hi - 1), key);
        rt::keep4_bits(m.lo, w >> (4u * g));
        rt::keep4_bits(m.hi, w >> (4u * g + 16u));
      }
      split8<false>(m.lo, m.hi, pl);
      u32x4* dst = lds + ((size_t)(buf * RB + wave + NW * k) * 3) * 64 + lane;
#pragma unroll
      for (int q = 0; q < 3; ++q) dst[64 * q] = pl.p[q];
    }
  };
  auto fetch = [&](Planes& o, int i, int buf) {
    const u32x4* src = lds + ((size_t)(buf * RB + i) * 3) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 3; ++q) o.p[q] = src[64 * q];
  };
  auto mfma_row = [&](const Planes& a, const Planes(&b)[CB], int i) {
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int j = 0; j < CB; ++j) acc[i][j] = mfma_bf16(a.p[PA[q]], b[j].p[PB[q]], acc[i][j]);
  };
  // parity P: the planes of chunk c are in LDS buffer P, row block 0's in xp[RB P & 1]...: row blocks are numbered n = P RB + i
  // over a pair of chunks so that the plane sets alternate across the chunk boundary (RB odd)
  auto chunk = [&](Planes(&b)[CB], Planes(&bn)[CB], ARaw(&rn)[MINE], ARaw(&rf)[MINE], int c, auto parity) {
    constexpr int P = decltype(parity)::value;
    constexpr int LB = (3 * CB + RB - 2) / (RB - 1);   // B plane loads per row block (none in the last)
    int bidx = 0, pk = 0;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int n = P * RB + i;
      if (i == RB - 1) __syncthreads();     // chunk c + 1's planes are all in buffer P ^ 1
      if (i + 1 < RB) fetch(xp[(n + 1) & 1], i + 1, P);
      else fetch(xp[(n + 1) & 1], 0, P ^ 1);
      if (i == 0) loadA(rf, c + 2);
      // this wave's row blocks of chunk c + 1: one per region, from region 1 on
      if (i >= 1 && pk < MINE && i < RB - 1) {
        produce(rn[pk], pk, P ^ 1, c + 1);
        ++pk;
      }
      int nl = 0;
      if (i < RB - 1) {
#pragma unroll
        for (int k = 0; k < LB; ++k)
          if (bidx < 3 * CB) {
            loadB1(bn, bidx, c + 1);
            ++bidx;
            ++nl;
          }
      }
      mfma_row(xp[n & 1], b, i);
      constexpr int NM = 6 * CB;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
        if (m < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        // DS read
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                   // VALU
        if (m >= 4 && m < 4 + 2 * (2 * MINE + LB) && (m & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        if (m >= NM - 6 && (m & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  static_assert(MINE <= RB - 2, "a wave's row blocks are produced in the regions 1 .. RB - 2");
  if (c_lo < c_hi) {
#pragma unroll
    for (int idx = 0; idx < 3 * CB; ++idx) loadB1(b0, idx, c_lo);
    loadA(raw0, c_lo);
    loadA(raw1, c_lo + 1);
#pragma unroll
    for (int k = 0; k < MINE; ++k) produce(raw0[k], k, 0, c_lo);
    __syncthreads();
    fetch(xp[0], 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = c_lo; c < c_hi; c += 2) {
      chunk(b0, b1, raw1, raw0, c, std::integral_constant<int, 0>{});
      chunk(b1, b0, raw0, raw1, c + 1, std::integral_constant<int, 1>{});
    }
  }
}

// SHARE_A (WM == 1): the WN waves of a K range work on the same rows of A and share its split through LDS (nt_accumulate_shared);
// needs the K ranges of the WK groups to be equally long (the host checks K % (64 WK) == 0).  Dynamic LDS: nt_lds_bytes().
template <int RB, int WN, int WK, bool SHARE_A>
constexpr size_t nt_lds_bytes(size_t reduction_bytes) {
  return SHARE_A ? (reduction_bytes > (size_t)WK * 2 * RB * 3 * 1024 ? reduction_bytes : (size_t)WK * 2 * RB * 3 * 1024) : reduction_bytes;
}
// REL: the tile's (T, C2) rows are staged into LDS first (rel_lds_bytes(); the K-split reduction reuses the space afterwards).
template <int RB, int CB, int WM, int WN, int WK, bool DROP, class Epi, int TUNE = 0, int NR = RB, bool SHARE_A = false, bool REL = false>
__global__ __launch_bounds__(kThreads, 1) void gemm_nt_kernel(NtArgs p, DropCfg dc, Epi epi) {
  using S = rt::NtShape<RB, CB, WM, WN, WK>;
  static_assert(!SHARE_A || WM == 1, "SHARE_A: the waves of a K range share their rows");
  static_assert(!(REL && SHARE_A), "REL is built on the per-wave split");
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wk = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  // Row-major tile order: xcd_remap gives an XCD a contiguous range of tiles, so the tiles_n column tiles of a row tile run on ONE
  // XCD and its A rows come from HBM once (column-major -- TUNE & 16 -- runs as fast but fetches A once per column tile: 311 MB
  // counted per launch against 177 algorithmic at N = 310); the packed B image (3.8 MB) is then wanted whole in every L2.
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles_m = gridDim.x / p.tiles_n;
  const int tm = (TUNE & 16) ? tile % tiles_m : tile / p.tiles_n, tn = (TUNE & 16) ? tile / tiles_m : tile % p.tiles_n;
  const int m0 = tm * S::BM + wm * (16 * RB);
  const int n0 = tn * S::BN + wn * (16 * CB);
  const int chunks = p.K / kChunk;

  f32x4 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int per = ((chunks + WK - 1) / WK + 1) & ~1;   // per wave: an even number of chunks
  const int c_lo = min(chunks, wk * per), c_hi = min(chunks, c_lo + per);

  if constexpr (SHARE_A) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    nt_accumulate_shared<RB, CB, WN, DROP, TUNE>(p, dc, ((size_t)(p.M - 1) * p.lda + p.K) * 4, m0, n0, c_lo, c_hi, wn,
                                                 sp_smem + (size_t)wk * 2 * RB * 3 * 1024, acc);
    __syncthreads();   // the K-split reduction below reuses the plane buffers
  } else if constexpr (REL) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    // stage T and C2 of the samples this workgroup's BM rows belong to: [sample][T | C2][K], 16-byte pieces, coalesced
    const int s0 = (tm * S::BM) / p.rps;
    const int ns = (min(p.M, (tm + 1) * S::BM) - 1) / p.rps - s0 + 1;     // (<= rel_samples(BM, rps): what the host sized the LDS for)
    const int k4 = p.K / 4;
    // (a loop with a UNIFORM trip count and a guarded body: after a loop whose exit is divergent hipcc made the buffer descriptors
    //  of nt_accumulate PHIs of that region, kept them in VGPRs and wrapped every one of the main loop's 89 buffer loads in a
    //  waterfall loop -- v_readfirstlane x 4, compare, branch -- : 157 us instead of 13x.  Check: count v_readfirstlane in the ISA)
    const int total = ns * 2 * k4, iters = (total + kThreads - 1) / kThreads;
    for (int it = 0; it < iters; ++it) {
      const int idx = min(it * kThreads + (int)threadIdx.x, total - 1);     // (the last lanes repeat the last piece: no branch at all)
      const int s = idx / (2 * k4), rem = idx - s * 2 * k4, which = rem / k4, k = rem - which * k4;
      const float* src = (which == 0 ? p.T : p.C2) + (size_t)(s0 + s) * p.K + 4 * k;
      reinterpret_cast<f32x4*>(sp_smem)[idx] = *reinterpret_cast<const f32x4*>(src);
    }
    __syncthreads();
    nt_accumulate<RB, CB, DROP, TUNE, NR, true>(p, dc, ((size_t)(p.M - 1) * p.lda + p.K) * 4, m0, n0, c_lo, c_hi, acc, sp_smem, s0);
    __syncthreads();   // the K-split reduction below reuses the staged rows' space
  } else {
    nt_accumulate<RB, CB, DROP, TUNE, NR>(p, dc, ((size_t)(p.M - 1) * p.lda + p.K) * 4, m0, n0, c_lo, c_hi, acc);
  }

  uint64_t bad = 0;      // blocks of this lane that hold a non-finite output: recomputed below (any_nonfinite)
  auto finish = [&](int blk, f32x4 v) {
    const int i = blk / CB, j = blk % CB;
    const int col = n0 + 16 * j + r;
    if (any_nonfinite(v)) bad |= 1ull << blk;
    if (col < p.N) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int row = m0 + 16 * i + 4 * g + t;
        if (row < p.M) epi(row, col, v[t]);
      }
    }
  };
  if constexpr (WK == 2) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    f32x4* red = reinterpret_cast<f32x4*>(sp_smem);
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
  static_assert(S::NB <= 64, "one bit per accumulator block");
  if (p.Bf != nullptr && bad != 0) {     // the repair path: the flagged blocks' outputs again, as fp32 dot products
    const uint32_t key = DROP ? drop_key(dc) : 0u;
    for (int blk = 0; blk < S::NB; ++blk) {
      if (((bad >> blk) & 1ull) == 0) continue;
      const int col = n0 + 16 * (blk % CB) + r;
      if (col >= p.N) continue;
      for (int t = 0; t < 4; ++t) {
        const int row = m0 + 16 * (blk / CB) + 4 * g + t;
        if (row < p.M) epi(row, col, nt_repair<DROP>(p.A, p.lda, p.Bf, p.ldb, p.K, row, col, key, REL ? p.T : nullptr, p.C2, REL ? p.rps : 1));
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// TN form (weight gradient):  slab[s][n1][n2] = sum_{m in slab s} G[m][n1] * X[m][n2]
//   G [M,N1] (the gated output gradient) comes as a packed plane image written by `pack_tn_kernel` (gate, split, column sums
//   for the bias gradient -- one small pass over 23 MB, so that only ONE operand is split inside the GEMM: with both split in
//   the loop a wave would issue 2.25 VALU instructions per MFMA, more than the 8 free issue cycles of a 16-cycle MFMA take);
//   X [M,N2] fp32 row stride ldx (the layer input; DROP: times its p = 0.5 mask), split in registers.
// Contraction = rows.  Chunk c = rows 32 c .. 32 c + 31; lane (r, g) supplies the rows 32 c + 8 g + j, j = 0..7.
//   G image: Gp[chunk][n1 / 16][plane][lane = (n1 % 16) + 16 g][8 bf16 = rows j]
//   X: the OUTPUT column index is permuted as in gemm_f32_rt.hpp's TN form -- lane r loads the 4 consecutive columns
//   4 r .. 4 r + 3 of a 64-column span as one 16-byte load per row, component cc feeds accumulator block cc (columns
//   {4 r + cc}), and the four blocks of a span come back together as 16-byte stores.
// Workgroup = 4 waves = 4 x NA consecutive 16-row blocks of n1 (all of N1 = 310 for NA = 5) x one group of SPN spans of n2 x
// one row slab; loop order per chunk: the 4 SPN column blocks outermost (a block's planes are made under the MFMAs of the
// one before), the NA row blocks x 6 products inside.
// ------------------------------------------------------------------------------------------------------------------
struct TnArgs {
  const u32x4* Gp;
  const float* X;
  float* slab;
  int ldx;
  int M, N1, N2;
  int nblocks;          // (N1 + 15) / 16
  int cps;              // chunks per slab (even)
  int tiles1, tiles2;   // workgroup tiles along n1 (4 NA blocks each) and n2 (64 SPN columns each)
  const float* Gf = nullptr;   // the fp32 original of the packed image ([M, N1], row stride ldg) and, when it was gated, the
  const float* Yf = nullptr;   // forward output whose sign gates it: operands of the repair path (any_nonfinite); Gf == nullptr:
  int ldg = 0;                 // no repair
  // REL (gemm_tn_shared_kernel): the layer input is x[m][n2] = T[s][n2] + C2[s][n2] * X[m][n2], s = m / rps (see NtArgs), recomputed
  // from X = v while it is staged -- the weight gradient of compress_v2 without v2 in HBM.  rps_magic = 2^32 / rps + 1.
  const float* T = nullptr;
  const float* C2 = nullptr;
  int rps = 0;
  uint32_t rps_magic = 0;
};
inline uint32_t rps_magic_of(int rps) { return (uint32_t)((1ull << 32) / (uint64_t)rps + 1ull); }   // m / rps = umulhi(m, magic), m rps < 2^32
// The epilogue of both TN kernels: D[row = n1 % 16 = 4 g + t][col r] of block (i, b = 4 q + cc) is column n2_0 + 64 q + 4 r + cc,
// so the four cc of a span leave as one 16-byte store.  A group of four that holds a non-finite value is stored again from the
// repair path: the slab's rows contracted as fp32 dot products of the original operands (same gate, same mask bits).
template <int NA, int SPN, bool DROP>
__device__ __forceinline__ void tn_store(const TnArgs& p, const DropCfg& dc, const f32x4 (&acc)[NA][4 * SPN], int slab, int b1, int n2_0,
                                         int r, int g) {
  float* out = p.slab + (size_t)slab * p.N1 * p.N2;
  uint64_t bad = 0;
  static_assert(NA * SPN * 4 <= 64, "one bit per 16-byte store");
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int q = 0; q < SPN; ++q) {
      const int col = n2_0 + 64 * q + 4 * r;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int n1 = 16 * (b1 + i) + 4 * g + t;
        const f32x4 v = f32x4{acc[i][4 * q][t], acc[i][4 * q + 1][t], acc[i][4 * q + 2][t], acc[i][4 * q + 3][t]};
        if (b1 + i < p.nblocks && n1 < p.N1 && col + 3 < p.N2) {
          *reinterpret_cast<f32x4*>(out + (size_t)n1 * p.N2 + col) = v;
          if (any_nonfinite(v)) bad |= 1ull << ((i * SPN + q) * 4 + t);
        }
      }
    }
  if (p.Gf != nullptr && bad != 0) {
    const uint32_t key = DROP ? drop_key(dc) : 0u;
    const int m_lo = slab * p.cps * kChunk, m_hi = min(p.M, m_lo + p.cps * kChunk);
    for (int k = 0; k < NA * SPN * 4; ++k) {
      if (((bad >> k) & 1ull) == 0) continue;
      const int t = k & 3, q = (k >> 2) % SPN, i = (k >> 2) / SPN;
      const int n1 = 16 * (b1 + i) + 4 * g + t, col = n2_0 + 64 * q + 4 * r;
      f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int m = m_lo; m < m_hi; ++m) {
        float gv = p.Gf[(size_t)m * p.ldg + n1];
        if (p.Yf != nullptr) gv = p.Yf[(size_t)m * p.ldg + n1] > 0.f ? gv : 0.f;
        f32x4 x = *reinterpret_cast<const f32x4*>(p.X + (size_t)m * p.ldx + col);
        if (p.T != nullptr) {
          const size_t at = (size_t)(m / p.rps) * p.N2 + col;
          x = __builtin_elementwise_fma(*reinterpret_cast<const f32x4*>(p.C2 + at), x, *reinterpret_cast<const f32x4*>(p.T + at));
        }
        if constexpr (DROP) {
          const uint32_t e = (uint32_t)m * (uint32_t)p.N2 + (uint32_t)col;     // a multiple of 4
          const uint32_t w = mask_word32(e >> 5, key) >> (e & 31u);
#pragma unroll
          for (int c = 0; c < 4; ++c) x[c] = ((w >> c) & 1u) != 0u ? x[c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) sum[c] = fmaf(gv, x[c], sum[c]);
      }
      *reinterpret_cast<f32x4*>(out + (size_t)n1 * p.N2 + col) = sum;
    }
  }
}
struct TnPlan {
  int slabs, cps;
};
inline TnPlan tn_plan(int M, int want_slabs) {
  const int chunks = (M + kChunk - 1) / kChunk;
  int S = want_slabs < 1 ? 1 : want_slabs;
  if (S > (chunks + 1) / 2) S = (chunks + 1) / 2;
  int cps = ((chunks + S - 1) / S + 1) & ~1;
  S = (chunks + cps - 1) / cps;
  return {S, cps};
}
__host__ __device__ inline size_t packed_tn_bytes(int slabs, int cps, int N1) {
  return (size_t)slabs * cps * ((N1 + 15) / 16) * kPackedChunkBytes;
}

// gate + split + pack of G, and the column sums of the gated G (the bias gradient's partial sums, fixed order); optionally the
// gated G itself in fp32 (what torch's threshold_backward would have written for the data gradient's kernel).
// One workgroup per (slab, quarter of the slab's chunks): whole rows come in by coalesced 16-byte loads (a row is N1 floats,
// 8-byte aligned at least), are gated, and cross an LDS tile of 32 rows x (N1 padded) floats to reach the lanes that pack
// them: lane (r, g) of wave w packs column 16 blk + r, rows 8 g .. 8 g + 7, for the blocks blk = w, w + 4, ...
constexpr int kPackParts = 36;   // 16 slabs x 36 = 576 workgroups of one chunk at M = 18432 (18 parts of two chunks: +5 us per step)
// Wide G (N1 > 16 kPackMaxBlocks columns: the question encoder's 2400-wide gradients) is packed in column groups of `nblocks`
// blocks: blockIdx.y = group, its first block blk0 = blockIdx.y * nblocks of the image's nblk_total (gridDim.y == 1, nblk_total ==
// nblocks: the whole matrix at once, as K5 calls it).
template <bool GATE, bool WRITE_GZ>
__global__ __launch_bounds__(256) void pack_tn_kernel(const float* __restrict__ gy, const float* __restrict__ y, int ld, int M,
                                                      int N1_all, int nblocks, int cps, u32x4* __restrict__ out,
                                                      float* __restrict__ dbslab, float* __restrict__ gz, int nblk_total) {
  extern __shared__ __attribute__((aligned(16))) float pk_tile[];   // [32][pitch], pitch = 16 nblocks + 1 (odd: the 4 lane groups'
  const int pitch = 16 * nblocks + 1;                               //  rows 8 g land on different banks)
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const int slab = blockIdx.x / kPackParts, part = blockIdx.x % kPackParts;
  const int per = (cps + kPackParts - 1) / kPackParts;
  const int lo = part * per, hi = min(cps, lo + per);
  const int blk0 = blockIdx.y * nblocks, c0 = 16 * blk0;      // this group's first block / column
  const int N1 = min(N1_all - c0, 16 * nblocks);              // the group's width
  gy += c0;
  if (GATE) y += c0;
  if (WRITE_GZ) gz += c0;
  const int pairs = N1 / 2;                   // (N1 is even: rows are float2-aligned)
  float colsum[8];                            // per block this wave owns (blk = wave + 4 k), column r, rows of lane group g
#pragma unroll
  for (int k = 0; k < 8; ++k) colsum[k] = 0.f;
  for (int cc = lo; cc < hi; ++cc) {
    const int c = slab * cps + cc;
    const int m0 = c * kChunk;
    // 32 rows x N1 floats: wave w takes the rows w, w + 4, ..., its lanes the float2 pairs of a row (no index division; a row
    // is one contiguous run of N1 / 2 eight-byte loads)
#pragma unroll 2
    for (int row = wave; row < 32; row += 4) {
      const int m = m0 + row;
      for (int cp = lane; cp < pairs; cp += 64) {
        float2 v = make_float2(0.f, 0.f);
        if (m < M) {
          v = *reinterpret_cast<const float2*>(gy + (size_t)m * ld + 2 * cp);
          if constexpr (GATE) {
            const float2 yy = *reinterpret_cast<const float2*>(y + (size_t)m * ld + 2 * cp);
            v.x = yy.x > 0.f ? v.x : 0.f;
            v.y = yy.y > 0.f ? v.y : 0.f;
          }
          if constexpr (WRITE_GZ) *reinterpret_cast<float2*>(gz + (size_t)m * ld + 2 * cp) = v;
        }
        pk_tile[row * pitch + 2 * cp] = v.x;
        pk_tile[row * pitch + 2 * cp + 1] = v.y;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int blk = wave + 4 * k;
      if (blk < nblocks && blk0 + blk < nblk_total) {
        const int col = 16 * blk + r;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = col < N1 ? pk_tile[(8 * g + j) * pitch + col] : 0.f;
        uint32_t w[3][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          split_pair<false>(f32x2{v[2 * jj], v[2 * jj + 1]}, w[0][jj], w[1][jj], w[2][jj]);
          colsum[k] += v[2 * jj] + v[2 * jj + 1];
        }
        u32x4* dst = out + ((size_t)c * nblk_total + blk0 + blk) * 192 + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q) dst[64 * q] = u32x4{w[q][0], w[q][1], w[q][2], w[q][3]};
      }
    }
    __syncthreads();
  }
  if (dbslab != nullptr) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float sum = colsum[k];
      sum += __shfl_xor(sum, 16);
      sum += __shfl_xor(sum, 32);
      const int col = 16 * (wave + 4 * k) + r;
      if (g == 0 && wave + 4 * k < nblocks && col < N1) dbslab[(size_t)blockIdx.x * N1_all + c0 + col] = sum;
    }
  }
}
inline size_t pack_tn_lds_bytes(int nblocks) { return (size_t)32 * (16 * nblocks + 1) * sizeof(float); }
constexpr int kPackMaxBlocks = 30;   // (8 blocks per wave; the tile stays under the 64 KiB default LDS cap)

// TUNE: bit 0 = no X split (wrong results: ceiling), bit 1 = no X loads in the loop, bit 5 = no G loads in the loop
template <int NA, int SPN, bool DROP, int TUNE = 0>
__global__ __launch_bounds__(kThreads, 1) void gemm_tn_kernel(TnArgs p, DropCfg dc) {
  constexpr int NBX = 4 * SPN;   // X column blocks per chunk
  static_assert(NBX % 2 == 0, "plane sets alternate per block");
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.tiles1 * p.tiles2;
  const int slab = bid / tiles, tile = bid % tiles;
  const int t2 = tile % p.tiles2, t1 = tile / p.tiles2;
  const int b1 = (t1 * 4 + wave) * NA;       // first n1 block of this wave
  const int n2_0 = t2 * (64 * SPN);
  const int c_lo = slab * p.cps, c_hi = c_lo + p.cps;

  uint32_t offX[8], offG[NA];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    offX[j] = ((uint32_t)(8 * g + j) * (uint32_t)p.ldx + (uint32_t)min(n2_0 + 4 * r, p.N2 - 4)) * 4u;
#pragma unroll
  for (int i = 0; i < NA; ++i) offG[i] = (uint32_t)min(b1 + i, p.nblocks - 1) * (uint32_t)kPackedChunkBytes + 16u * lane;
  // rows past M read as zeros (the buffer's range check); columns past N2 are clamped and never stored
  const rsrc_t Xb = make_rsrc(p.X, ((size_t)(p.M - 1) * p.ldx + p.N2) * 4);
  const rsrc_t Gb = make_rsrc(p.Gp, (size_t)gridDim.x / tiles * p.cps * p.nblocks * kPackedChunkBytes);
  const uint32_t key = DROP ? drop_key(dc) : 0u;
  const uint32_t chunk_stride_x = 32u * (uint32_t)p.ldx * 4u, chunk_stride_g = (uint32_t)p.nblocks * (uint32_t)kPackedChunkBytes;

  f32x4 acc[NA][NBX];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int b = 0; b < NBX; ++b) acc[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 xr[SPN][8];
  Planes g0[NA], g1[NA], pl[2];
  auto loadX = [&](int q, int c) {
    const uint32_t so = (uint32_t)c * chunk_stride_x;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      xr[q][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(Xb, (int)(offX[j] + 256u * q), (int)so, 0));
  };
  auto loadG1 = [&](Planes(&gp)[NA], int idx, int c) {
    gp[idx / 3].p[idx % 3] = __builtin_bit_cast(
        u32x4, __builtin_amdgcn_raw_buffer_load_b128(Gb, (int)(offG[idx / 3] + 1024u * (idx % 3)), (int)((uint32_t)c * chunk_stride_g), 0));
  };
  // DROP: element (row m, column n2) keeps iff bit (m N2 + n2) & 31 of the hash word of counter (m N2 + n2) >> 5 is set.  A lane's
  // four columns 4 r .. 4 r + 3 of span q sit at bits 4 (r & 7) .. + 3 of the word of (row, 32-column group 2 q + (r >> 3)).  The 8 lanes
  // r & ~7 .. + 7 of a lane group need the same 8 SPN words per chunk (8 rows x SPN spans): lane t = r & 7 hashes the words of
  // row j = t (every span) and they travel by ds_bpermute -- SPN hashes + 8 SPN permutes per chunk instead of 8 SPN hashes.
  uint32_t hw[SPN][8];
  const int t8 = r & 7;
  auto hash_chunk = [&](int q, int c) {
    const uint32_t row = (uint32_t)c * 32u + 8u * (uint32_t)g + (uint32_t)t8;
    const uint32_t e = row * (uint32_t)p.N2 + (uint32_t)n2_0 + 64u * (uint32_t)q + 32u * (uint32_t)(r >> 3);
    const int h = (int)mask_word32(e >> 5, key);
#pragma unroll
    for (int j = 0; j < 8; ++j) hw[q][j] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((lane & ~7) + j), h);
  };
  // planes of X block (q, cc) of the chunk whose raw rows are in xr[q]
  auto prepare = [&](Planes& o, int q, int cc) {
    if constexpr ((TUNE & 1) != 0) {
      o.p[0] = __builtin_bit_cast(u32x4, xr[q][cc]);
      o.p[1] = __builtin_bit_cast(u32x4, xr[q][cc + 4]);
      o.p[2] = __builtin_bit_cast(u32x4, xr[q][cc]);
    } else {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = xr[q][j][cc];
        if constexpr (DROP) {
          const uint32_t m = 0u - ((hw[q][j] >> (4u * (uint32_t)t8 + (uint32_t)cc)) & 1u);
          v[j] = __uint_as_float(__float_as_uint(x) & m);
        } else {
          v[j] = x;
        }
      }
      uint32_t w[3][4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) split_pair<false>(f32x2{v[2 * jj], v[2 * jj + 1]}, w[0][jj], w[1][jj], w[2][jj]);
#pragma unroll
      for (int k = 0; k < 3; ++k) o.p[k] = u32x4{w[k][0], w[k][1], w[k][2], w[k][3]};
    }
  };
  auto mfma_block = [&](const Planes& x, const Planes(&gp)[NA], int b) {
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i][b] = mfma_bf16(gp[i].p[PA[k]], x.p[PB[k]], acc[i][b]);
  };
  // One chunk.  On entry pl[0] holds the planes of block 0 of chunk c, xr the raw rows of chunk c (span 0's of chunk c + 1 once
  // its last block has been split).  Region of block b: planes of block b + 1, loads, 6 NA MFMAs.
  auto chunk = [&](Planes(&gp)[NA], Planes(&gn)[NA], int c) {
    const int cn = min(c + 1, c_hi - 1);
    constexpr int LG = (3 * NA + NBX - SPN - 1) / (NBX - SPN);   // G plane loads per region that carries no X loads
    int gidx = 0;
#pragma unroll
    for (int b = 0; b < NBX; ++b) {
      const int q = b / 4, cc = b % 4;
      int nl = 0;
      if (b + 1 < NBX) {
        if constexpr (DROP) {
          if ((b + 1) % 4 == 0) hash_chunk((b + 1) / 4, c);
        }
        prepare(pl[(b + 1) & 1], (b + 1) / 4, (b + 1) % 4);
      } else {
        if constexpr (DROP) hash_chunk(0, cn);
        prepare(pl[0], 0, 0);      // next chunk's block 0: its span was reloaded in region 3
      }
      if (cc == 3) {
        // span q's last block was split in the previous region: its registers take the next chunk's rows
        if constexpr ((TUNE & 2) == 0) {
          loadX(q, cn);
          nl = 8;
        }
      } else if constexpr ((TUNE & 32) == 0) {
#pragma unroll
        for (int k = 0; k < LG; ++k)
          if (gidx < 3 * NA) {
            loadG1(gn, gidx, cn);
            ++gidx;
            ++nl;
          }
      }
      mfma_block(pl[b & 1], gp, b);
      constexpr int NM = 6 * NA;
      const int per = nl > 0 ? NM / nl : NM;
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, ((TUNE >> 2) & 3) != 0 ? ((TUNE >> 2) & 3) : 2, 0);   // VALU
        if (nl > 0 && m % per == per - 1 && m / per < nl) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
#pragma unroll
    for (int idx = 0; idx < 3 * NA; ++idx) loadG1(g0, idx, c_lo);
#pragma unroll
    for (int q = 0; q < SPN; ++q) loadX(q, c_lo);
    if constexpr (DROP) hash_chunk(0, c_lo);
    prepare(pl[0], 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = c_lo; c < c_hi; c += 2) {
      chunk(g0, g1, c);
      chunk(g1, g0, c + 1);
    }
  }

  tn_store<NA, SPN, DROP>(p, dc, acc, slab, b1, n2_0, r, g);
}

// The TN form with the split of X shared by the workgroup.  The four waves of a workgroup work on the SAME rows and columns of X
// (they differ in n1), so in gemm_tn_kernel each of them loads and splits all of it: 4 x the loads, 4 x the VALU work (a quarter
// of that kernel's time: 104 us against 80 with the split switched off).  Here wave w loads only the two columns of each
// 4-column group that make X blocks 2 w and 2 w + 1 (8-byte loads), splits those two blocks of the NEXT chunk under this chunk's
// MFMAs and writes their planes into LDS (2 x 24 KiB, double-buffered); all four waves read every block's planes back (three
// ds_read_b128 per block, one block ahead).  One workgroup barrier per chunk, placed in front of the last block so that the next
// chunk's block 0 is prefetched under it.  SPN = 2 (8 blocks: two per wave).
// REL: see TnArgs.  A lane's 8 rows of a chunk belong to at most two samples (rps >= 8): the (T, C2) pairs of both come in by
// 8-byte loads two chunks ahead, like the raw rows, and each row picks its sample's pair.
constexpr int kTnSharedLds = 2 * 8 * 3 * 1024;
template <int NA, bool DROP, int TUNE = 0, bool REL = false>
__global__ __launch_bounds__(kThreads, 1) void gemm_tn_shared_kernel(TnArgs p, DropCfg dc) {
  constexpr int SPN = 2, NBX = 8;
  extern __shared__ __attribute__((aligned(16))) char tn_smem[];
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tiles = p.tiles1 * p.tiles2;
  const int slab = bid / tiles, tile = bid % tiles;
  const int t2 = tile % p.tiles2, t1 = tile / p.tiles2;
  const int b1 = (t1 * 4 + wave) * NA;       // first n1 block of this wave
  const int n2_0 = t2 * (64 * SPN);
  const int c_lo = slab * p.cps, c_hi = c_lo + p.cps;
  const int pq = wave >> 1, pc = 2 * (wave & 1);   // this wave produces the X blocks 4 pq + pc, 4 pq + pc + 1

  uint32_t offX[8], offG[NA];
#pragma unroll
  for (int j = 0; j < 8; ++j)
    offX[j] = ((uint32_t)(8 * g + j) * (uint32_t)p.ldx + (uint32_t)min(n2_0 + 64 * pq + 4 * r + pc, p.N2 - 2)) * 4u;
#pragma unroll
  for (int i = 0; i < NA; ++i) offG[i] = (uint32_t)min(b1 + i, p.nblocks - 1) * (uint32_t)kPackedChunkBytes + 16u * lane;
  const rsrc_t Xb = make_rsrc(p.X, ((size_t)(p.M - 1) * p.ldx + p.N2) * 4);
  const rsrc_t Gb = make_rsrc(p.Gp, (size_t)gridDim.x / tiles * p.cps * p.nblocks * kPackedChunkBytes);
  const uint32_t key = DROP ? drop_key(dc) : 0u;
  const uint32_t chunk_stride_x = 32u * (uint32_t)p.ldx * 4u, chunk_stride_g = (uint32_t)p.nblocks * (uint32_t)kPackedChunkBytes;
  u32x4* const lds = reinterpret_cast<u32x4*>(tn_smem);     // [buffer][block][plane][lane]

  f32x4 acc[NA][NBX];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int b = 0; b < NBX; ++b) acc[i][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x2 raw0[8], raw1[8];
  Planes g0[NA], g1[NA], xp[2];
  uint32_t hw[8];
  const int t8 = r & 7;
  struct TCPair {
    f32x2 ta, ca, tb, cb;   // (T, C2) of this lane's two columns for the sample of its first row (a) and for the next one (b)
    int na;                 // how many of the lane's 8 rows belong to sample a
  };
  TCPair tc0, tc1;
  const int nsamples = (p.M + (REL ? p.rps : 1) - 1) / (REL ? p.rps : 1);
  const uint32_t colT = (uint32_t)min(n2_0 + 64 * pq + 4 * r + pc, p.N2 - 2) * 4u;
  const rsrc_t Tb = make_rsrc(REL ? p.T : p.X, (size_t)nsamples * p.N2 * 4), Cb = make_rsrc(REL ? p.C2 : p.X, (size_t)nsamples * p.N2 * 4);
  auto loadTC = [&](TCPair& o, int c) {
    if constexpr (REL) {
      const uint32_t row = (uint32_t)min(c, c_hi - 1) * 32u + 8u * (uint32_t)g;
      const uint32_t sa = min(__umulhi(row, p.rps_magic), (uint32_t)(nsamples - 1)), sb = min(sa + 1u, (uint32_t)(nsamples - 1));
      o.na = (int)((sa + 1u) * (uint32_t)p.rps - row);      // rows until the sample ends (>= 8: all of them)
      const uint32_t oa = sa * (uint32_t)p.N2 * 4u + colT, ob = sb * (uint32_t)p.N2 * 4u + colT;
      o.ta = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(Tb, (int)oa, 0, 0));
      o.ca = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(Cb, (int)oa, 0, 0));
      o.tb = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(Tb, (int)ob, 0, 0));
      o.cb = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(Cb, (int)ob, 0, 0));
    }
  };
  auto loadX = [&](f32x2(&raw)[8], int c) {
    const uint32_t so = (uint32_t)min(c, c_hi - 1) * chunk_stride_x;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      raw[j] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(Xb, (int)offX[j], (int)so, 0));
    }
  };
  auto loadG1 = [&](Planes(&gp)[NA], int idx, int c) {
    gp[idx / 3].p[idx % 3] = __builtin_bit_cast(
        u32x4, __builtin_amdgcn_raw_buffer_load_b128(Gb, (int)(offG[idx / 3] + 1024u * (idx % 3)), (int)((uint32_t)min(c, c_hi - 1) * chunk_stride_g), 0));
  };
  // the hash words of this wave's span for the 8 rows of chunk c (gemm_tn_kernel's sharing: lane t = r & 7 hashes row t)
  auto hash_chunk = [&](int c) {
    const uint32_t row = (uint32_t)min(c, c_hi - 1) * 32u + 8u * (uint32_t)g + (uint32_t)t8;
    const uint32_t e = row * (uint32_t)p.N2 + (uint32_t)n2_0 + 64u * (uint32_t)pq + 32u * (uint32_t)(r >> 3);
    const int h = (int)mask_word32(e >> 5, key);
#pragma unroll
    for (int j = 0; j < 8; ++j) hw[j] = (uint32_t)__builtin_amdgcn_ds_bpermute(4 * ((lane & ~7) + j), h);
  };
  // split column k (0 / 1) of the raw rows into the planes of X block 4 pq + pc + k and write them to LDS buffer `buf`
  auto produce = [&](const f32x2(&raw)[8], int k, int buf, const TCPair& tc) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float x = raw[j][k];
      if constexpr (REL) x = j < tc.na ? fmaf(tc.ca[k], x, tc.ta[k]) : fmaf(tc.cb[k], x, tc.tb[k]);
      if constexpr (DROP) {
        const uint32_t m = 0u - ((hw[j] >> (4u * (uint32_t)t8 + (uint32_t)(pc + k))) & 1u);
        v[j] = __uint_as_float(__float_as_uint(x) & m);
      } else {
        v[j] = x;
      }
    }
    uint32_t w[3][4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) split_pair<false>(f32x2{v[2 * jj], v[2 * jj + 1]}, w[0][jj], w[1][jj], w[2][jj]);
    u32x4* dst = lds + ((size_t)(buf * NBX + 4 * pq + pc + k) * 3) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 3; ++q) dst[64 * q] = u32x4{w[q][0], w[q][1], w[q][2], w[q][3]};
  };
  auto fetch = [&](Planes& o, int b, int buf) {
    const u32x4* src = lds + ((size_t)(buf * NBX + b) * 3) * 64 + lane;
#pragma unroll
    for (int q = 0; q < 3; ++q) o.p[q] = src[64 * q];
  };
  auto mfma_block = [&](const Planes& x, const Planes(&gp)[NA], int b) {
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i][b] = mfma_bf16(gp[i].p[PA[k]], x.p[PB[k]], acc[i][b]);
  };
  // One chunk (parity P inside a pair).  On entry: the planes of chunk c are in LDS buffer P (published by the last barrier) and
  // block 0's are in xp[0]; `rn` holds the raw rows of chunk c + 1, `rf` is free for chunk c + 2's.
  auto chunk = [&](Planes(&gp)[NA], Planes(&gn)[NA], f32x2(&rn)[8], f32x2(&rf)[8], TCPair& tn, TCPair& tf, int c, auto parity) {
    constexpr int P = decltype(parity)::value;
    int gidx = 0;
#pragma unroll
    for (int b = 0; b < NBX; ++b) {
      if (b == 7) __syncthreads();            // every wave's planes of chunk c + 1 are in buffer P ^ 1; buffer P's last block was read in region 6
      if (b + 1 < NBX) fetch(xp[(b + 1) & 1], b + 1, P);
      else fetch(xp[0], 0, P ^ 1);
      if (b == 0) {
        if constexpr ((TUNE & 2) == 0) loadX(rf, c + 2);
        if constexpr (DROP) hash_chunk(c + 1);
      }
      if (b == 1) produce(rn, 0, P ^ 1, tn);
      if (b == 4) produce(rn, 1, P ^ 1, tn);
      if (REL && b == 1) loadTC(tf, c + 2);      // (tf was the previous chunk's tn: free since its region 4; the caller swaps the two)
      if constexpr ((TUNE & 32) == 0) {
        if (b == 2 || b == 3 || b == 5 || b == 6) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (gidx < 3 * NA) {
              loadG1(gn, gidx, c + 1);
              ++gidx;
            }
        }
      }
      mfma_block(xp[b & 1], gp, b);
      if constexpr ((TUNE & 64) == 0) {
        // the block's 6 NA MFMAs with the region's other work pinned between them: the three LDS reads of the next block first,
        // then the split's VALU instructions (two per MFMA), the loads spread over the middle, the LDS writes at the end
        constexpr int NM = 6 * NA;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // MFMA
          if (m < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                        // DS read
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                   // VALU
          if (m >= 6 && m < 22 && (m & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (<= 8 per region)
          if (m >= NM - 6 && (m & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write (<= 3 per region)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
#pragma unroll
    for (int idx = 0; idx < 3 * NA; ++idx) loadG1(g0, idx, c_lo);
    loadX(raw0, c_lo);
    loadX(raw1, c_lo + 1);
    loadTC(tc0, c_lo);
    loadTC(tc1, c_lo + 1);
    if constexpr (DROP) hash_chunk(c_lo);
    produce(raw0, 0, 0, tc0);
    produce(raw0, 1, 0, tc0);
    __syncthreads();
    fetch(xp[0], 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    for (int c = c_lo; c < c_hi; c += 2) {
      chunk(g0, g1, raw1, raw0, tc1, tc0, c, std::integral_constant<int, 0>{});
      chunk(g1, g0, raw0, raw1, tc0, tc1, c + 1, std::integral_constant<int, 1>{});
    }
  }

  tn_store<NA, SPN, DROP>(p, dc, acc, slab, b1, n2_0, r, g);
}

// d_w[e] = scale * (fixed-order sum of the S slabs), float4 lanes, four slabs' loads in flight;  d_b[n] = sum of the Sb rows
// of partial column sums (the blocks past the d_w range)
static __global__ __launch_bounds__(256) void slab_sum_kernel(const float* __restrict__ slab, const float* __restrict__ dbslab,
                                                       float* __restrict__ d_w, float* __restrict__ d_b, int NK, int N, int S,
                                                       int Sb, float scale) {
  const int wblocks = (NK / 4 + 255) / 256;
  if ((int)blockIdx.x >= wblocks) {
    // 4 columns per block: thread (column tid & 3, row group tid >> 2) adds the rows s = group, group + 64, ... in order, then
    // the 64 groups' partial sums are added in order (round 6: 64 groups instead of 16 -- pack_tn hands over 576 rows of partial
    // sums now, and a thread's loads are a chain of dependent latencies)
    __shared__ float part[64][5];
    const int n = ((int)blockIdx.x - wblocks) * 4 + (threadIdx.x & 3), grp = threadIdx.x >> 2;
    float a = 0.f;
    if (d_b != nullptr && n < N)
      for (int s = grp; s < Sb; s += 64) a += dbslab[(size_t)s * N + n];
    part[grp][threadIdx.x & 3] = a;
    __syncthreads();
    if (d_b != nullptr && threadIdx.x < 4 && n < N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 64; ++k) t += part[k][threadIdx.x];
      d_b[n] = t;
    }
    return;
  }
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= NK) return;
  const float* src = slab + e;
  f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 4 <= S; s += 4) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + (size_t)s * NK), v1 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 1) * NK),
                v2 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 2) * NK), v3 = *reinterpret_cast<const f32x4*>(src + (size_t)(s + 3) * NK);
    a = (((a + v0) + v1) + v2) + v3;
  }
  for (; s < S; ++s) a += *reinterpret_cast<const f32x4*>(src + (size_t)s * NK);
  *reinterpret_cast<f32x4*>(d_w + e) = a * scale;
}
inline int slab_sum_blocks(int NK, int N) { return (NK / 4 + 255) / 256 + (N + 3) / 4; }

}  // namespace sp
}  // namespace vqa
