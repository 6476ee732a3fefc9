// bf16 MFMA tile engine for gfx950 (v_mfma_f32_32x32x16_bf16: bf16 operands, fp32 accumulate, 8 passes).
//
// The mixed-precision path (BASELINE configs[4]) keeps every region-side matrix in bf16 with feature dims padded
// to a multiple of 64 and zero-filled, so the engine has no K tail and no column masks; rows past a matrix end are
// read from a clamped (valid) row and only ever reach accumulator rows / columns that are not stored.
//
// One workgroup = 256 threads = 4 waves arranged 2 x 2 over a BM x BN output tile; every wave owns
// (BM/64) x (BN/64) accumulators of 32 x 32.  The K loop advances 64 at a time through a two-stage LDS ring.
// LDS image of both operands: [mn][64 k] bf16 rows with a 144-byte pitch -- lane l of a wave feeds the MFMA with the
// 8 consecutive k of row (l & 31) starting at 8 * (l >> 5), i.e. one ds_read_b128, and 9 * row mod 16 is a
// bijection, so the 16 lanes of a read group land on 16 distinct 16-byte slots of the 256-byte bank row.
//
//   NT  C[M,N]  = A[M,K] * B[N,K]^T   both K-contiguous: 16-byte global loads, ds_write_b128 as loaded
//   TN  C[N1,N2] = A[K,N1]^T * B[K,N2]  both K-strided (weight gradients: K = rows of the batch): each thread loads an
//       8(k) x 8(mn) block as eight 16-byte rows, transposes it in registers (32 v_perm_b32) and writes eight
//       ds_write_b128 -- the same LDS image, so the MFMA side is shared.
//
// Global loads run two stages ahead in two register sets (write-after-barrier pipeline, as in gemm_f32_mfma.hpp).
#pragma once
#include "common.hpp"

namespace vqa {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));  // a 16-byte register quad (native vector: stays in VGPRs)

constexpr int kBfThreads = 256;
constexpr int kBfBK = 64;                  // k per LDS stage
constexpr int kBfPitch = kBfBK * 2 + 16;   // bytes per LDS row

template <int BM, int BN>
struct BfTile {
  static_assert(BM == 64 || BM == 128, "BM must be 64 or 128");
  static_assert(BN == 64 || BN == 128, "BN must be 64 or 128");
  static constexpr int TM = BM / 64, TN = BN / 64;
  static constexpr int CA = BM * 8 / kBfThreads, CB = BN * 8 / kBfThreads;  // 16-byte chunks per thread per stage (NT)
  static constexpr int kStageBytes = (BM + BN) * kBfPitch;
  static constexpr int kSmemBytes = 2 * kStageBytes;
};

// Operand transforms applied while a chunk of 8 consecutive bf16 of one source row travels from its registers into LDS.
// `row` is the (clamped) source row, `col` the first of the 8 columns.
struct BfNoTransform {
  static constexpr bool kActive = false;
  __device__ __forceinline__ u32x4 operator()(u32x4 v, int, int) const { return v; }
  __device__ __forceinline__ uint32_t word(int, int) const { return 0u; }
  __device__ __forceinline__ static u32x4 apply(u32x4 v, uint32_t) { return v; }
};
// p = 0.5 dropout of a row-major [rows, ld] tensor in the library's one-bit counter-hash form (common.hpp: keep element e
// iff bit e & 31 of mask_word32(e >> 5, key) is set).  Only ZEROES the dropped elements: the factor 1/(1-p) = 2 is uniform
// and is applied to the accumulator by the kernel's epilogue.  ld % 32 == 0, so the 32 columns [32 q, 32 q + 32) of a row
// sit in one hash word and a chunk's 8 bits in one byte of it.
struct BfDropHalf {
  static constexpr bool kActive = true;
  uint32_t key;
  uint32_t ld;
  // the hash word that holds element (row, col)
  __device__ __forceinline__ uint32_t word(int row, int col) const {
    return mask_word32(((uint32_t)row * ld + (uint32_t)col) >> 5, key);
  }
  // bits: bit i <-> element i of the chunk
  __device__ __forceinline__ static u32x4 apply(u32x4 v, uint32_t bits) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t lo = (bits >> (2 * i)) & 1u, hi = (bits >> (2 * i + 1)) & 1u;
      o[i] = v[i] & (((0u - lo) & 0x0000FFFFu) | ((0u - hi) & 0xFFFF0000u));
    }
    return o;
  }
  __device__ __forceinline__ u32x4 operator()(u32x4 v, int row, int col) const {
    return apply(v, word(row, col) >> ((uint32_t)col & 31u));
  }
};
// The hash (two quarter-rate 32-bit multiplies) is what the mask costs.  Both staging layouts give the four lanes of a quad
// the four 8-column chunks of ONE 32-column group, and a lane's C chunks (C <= 4) are that group in C different rows: lane j of
// the quad hashes the word of chunk j % C only and the quad exchanges them (DPP quad_perm broadcast) -- one hash per lane and
// stage instead of C.
template <int P>
__device__ __forceinline__ uint32_t bf_quad_bcast(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, P * 0x55, 0xF, 0xF, true);   // quad_perm [P, P, P, P]
}
template <int C>
__device__ __forceinline__ void bf_quad_words(uint32_t own, uint32_t (&w)[C]) {
  static_assert(C >= 1 && C <= 4, "a quad exchanges at most four words");
  w[0] = bf_quad_bcast<0>(own);
  if constexpr (C > 1) w[1] = bf_quad_bcast<1>(own);
  if constexpr (C > 2) w[2] = bf_quad_bcast<2>(own);
  if constexpr (C > 3) w[3] = bf_quad_bcast<3>(own);
}

// MFMA side of one stage: all fragments of LDS stage `base`, then the TM x TN x 4 MFMAs.
template <int BM, int BN>
__device__ __forceinline__ void bf_stage_mfma(const char* base, f32x16 (&acc)[BM / 64][BN / 64]) {
  using T = BfTile<BM, BN>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const char* ap = base + (wm * (BM / 2) + (lane & 31)) * kBfPitch + (lane >> 5) * 16;
  const char* bp = base + BM * kBfPitch + (wn * (BN / 2) + (lane & 31)) * kBfPitch + (lane >> 5) * 16;
  bf16x8 a[4][T::TM], b[4][T::TN];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i) a[kk][i] = *reinterpret_cast<const bf16x8*>(ap + i * 32 * kBfPitch + kk * 32);
#pragma unroll
    for (int j = 0; j < T::TN; ++j) b[kk][j] = *reinterpret_cast<const bf16x8*>(bp + j * 32 * kBfPitch + kk * 32);
  }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
}

// Per-thread staging state of the NT form: chunk p of a thread is 16 bytes (8 k) of tile row tid/8 + 32 p.
template <int BM, int BN>
struct NtStager {
  using T = BfTile<BM, BN>;
  const bf16* a;
  const bf16* b;
  int oa[T::CA], ob[T::CB];  // element offsets of the (clamped) source rows
  int lds_a, lds_b;          // byte offsets of chunk 0 in an LDS stage; chunk p is 32 rows further
  int nsteps;
  int arow0, acol0;          // first A row of this thread (chunk p: + 32 p, clamped to a_rows - 1) / its column inside a stage
  int a_last;
  // b_interleave > 0 (BN = 128 only): the B tile is two stacked matrices' rows [n0, n0 + 64), the second one b_interleave rows
  // below the first, laid out in blocks of 32 as {first[0:32), second[0:32), first[32:64), second[32:64)} -- every wave's two
  // 32-column accumulators are then the SAME 32 columns of the two matrices (K4 with two ranks: one staged x tile serves both)
  __device__ __forceinline__ NtStager(const bf16* A, int lda, int a_rows, const bf16* B, int ldb, int b_rows, int m0, int n0,
                                      int K, int b_interleave = 0)
      : a(A), b(B), nsteps(K / kBfBK), a_last(a_rows - 1) {
    const int row = threadIdx.x >> 3, kc = threadIdx.x & 7;
    arow0 = m0 + row;
    acol0 = kc * 8;
#pragma unroll
    for (int p = 0; p < T::CA; ++p) oa[p] = min(m0 + row + 32 * p, a_rows - 1) * lda + kc * 8;
#pragma unroll
    for (int p = 0; p < T::CB; ++p) {
      const int src = b_interleave > 0 ? (p & 1) * b_interleave + n0 + 32 * (p >> 1) + row : n0 + row + 32 * p;
      ob[p] = min(src, b_rows - 1) * ldb + kc * 8;
    }
    lds_a = row * kBfPitch + kc * 16;
    lds_b = (BM + row) * kBfPitch + kc * 16;
  }
  __device__ __forceinline__ void load(u32x4 (&ra)[T::CA], u32x4 (&rb)[T::CB], int step) const {
    const int k0 = min(step, nsteps - 1) * kBfBK;  // past the end: a valid (re-read) stage that is never consumed
#pragma unroll
    for (int p = 0; p < T::CA; ++p) ra[p] = *reinterpret_cast<const u32x4*>(a + oa[p] + k0);
#pragma unroll
    for (int p = 0; p < T::CB; ++p) rb[p] = *reinterpret_cast<const u32x4*>(b + ob[p] + k0);
  }
  // `step`: the K stage these registers hold (the A transform sees the chunk's source row and first column)
  template <class XA>
  __device__ __forceinline__ void store(const u32x4 (&ra)[T::CA], const u32x4 (&rb)[T::CB], char* base, int step,
                                        const XA& xa) const {
    const int col = min(step, nsteps - 1) * kBfBK + acol0;
    uint32_t words[T::CA];
    if constexpr (XA::kActive) {   // lanes 8 i .. 8 i + 3 hold columns [32 q, 32 q + 32) of rows arow0 + 32 p
      const int pj = (threadIdx.x & 3) % T::CA;
      bf_quad_words<T::CA>(xa.word(min(arow0 + 32 * pj, a_last), col), words);
    }
#pragma unroll
    for (int p = 0; p < T::CA; ++p)
      *reinterpret_cast<u32x4*>(base + lds_a + p * 32 * kBfPitch) =
          XA::kActive ? XA::apply(ra[p], words[p] >> ((uint32_t)col & 31u)) : ra[p];
#pragma unroll
    for (int p = 0; p < T::CB; ++p) *reinterpret_cast<u32x4*>(base + lds_b + p * 32 * kBfPitch) = rb[p];
  }
};

template <int MFMAS, int VALU>
__device__ __forceinline__ void bf_interleave() {
#pragma unroll
  for (int g = 0; g < MFMAS; ++g) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
    __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);  // VALU
  }
}

// acc += A[m0 : m0+BM, 0 : K) * B[n0 : n0+BN, 0 : K)^T.  K % 64 == 0; element offsets must fit 31 bits.  Two register
// sets run the global loads two stages ahead of the MFMAs.  Ends on a barrier (LDS reusable at once).
template <int BM, int BN, class XA = BfNoTransform>
__device__ __forceinline__ void gemm_bf16_nt_tile(const bf16* __restrict__ A, int lda, int a_rows,
                                                  const bf16* __restrict__ B, int ldb, int b_rows, int m0, int n0, int K,
                                                  char* smem, f32x16 (&acc)[BM / 64][BN / 64], const XA& xa = XA(),
                                                  int b_interleave = 0) {
  using T = BfTile<BM, BN>;
  const NtStager<BM, BN> st(A, lda, a_rows, B, ldb, b_rows, m0, n0, K, b_interleave);
  u32x4 ra0[T::CA], rb0[T::CB], ra1[T::CA], rb1[T::CB];
  st.load(ra0, rb0, 0);
  st.store(ra0, rb0, smem, 0, xa);
  st.load(ra1, rb1, 1);  // stage 1 -> set 1, stage 2 -> set 0, ...
  st.load(ra0, rb0, 2);
  __syncthreads();
  // (pairs of stages in a straight-line body, an odd last stage behind the loop: with the second half under `if (s + 1 <
  //  nsteps)` two paths met at the back edge, the compiler could not count the loads in flight any more and waited for
  //  vmcnt(0) at the top of the body -- for the register set it had just refilled as well: the loads ran ONE stage ahead,
  //  not two.  tools/sunk_loads_check.py, round 6.)
  int s = 0;
  for (; s + 1 < st.nsteps; s += 2) {
    bf_stage_mfma<BM, BN>(smem, acc);
    st.store(ra1, rb1, smem + T::kStageBytes, s + 1, xa);
    st.load(ra1, rb1, s + 3);
    bf_interleave<4 * T::TM * T::TN, 4>();
    __syncthreads();
    bf_stage_mfma<BM, BN>(smem + T::kStageBytes, acc);
    st.store(ra0, rb0, smem, s + 2, xa);
    st.load(ra0, rb0, s + 4);
    bf_interleave<4 * T::TM * T::TN, 4>();
    __syncthreads();
  }
  if (s < st.nsteps) {
    bf_stage_mfma<BM, BN>(smem, acc);
    __syncthreads();
  }
}

// 8 x 8 transpose of 16-bit elements held as eight 16-byte rows: out[mn] = the 8 k-values of column mn.
__device__ __forceinline__ void transpose8x8_b16(const uint4 (&r)[8], uint4 (&t)[8]) {
  const uint32_t* in = reinterpret_cast<const uint32_t*>(r);  // in[4*k + i]: row k, columns 2i, 2i+1
  uint32_t* out = reinterpret_cast<uint32_t*>(t);             // out[4*mn + q]: column mn, rows 2q, 2q+1
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t hi = in[4 * (2 * q + 1) + i], lo = in[4 * (2 * q) + i];
      out[4 * (2 * i) + q] = __builtin_amdgcn_perm(hi, lo, 0x05040100u);      // low halves  -> column 2i
      out[4 * (2 * i + 1) + q] = __builtin_amdgcn_perm(hi, lo, 0x07060302u);  // high halves -> column 2i+1
    }
}

// Per-thread staging state of the TN form: one 8(k) x 8(mn) block per stage.
template <int BM, int BN>
struct TnStager {
  const bf16* src;  // first element of this thread's column block, row 0
  int ld, kb, lds_off, k_lo, k_hi;
  int col0;         // first source column of this thread's block
  bool is_b;        // the block belongs to the B operand (transforms apply to B only)
  __device__ __forceinline__ TnStager(const bf16* A, int lda, int a_cols, const bf16* B, int ldb, int b_cols, int m0, int n0,
                                      int k_lo_, int k_hi_)
      : k_lo(k_lo_), k_hi(k_hi_) {
    const int tid = threadIdx.x;
    // BM + BN blocks per stage (8 k-blocks x mn/8 column blocks per operand); spare threads redo the last ones
    const int blk = tid < BM + BN ? tid : tid - (kBfThreads - (BM + BN));
    const bool is_a = blk < BM;
    const int ob = is_a ? blk : blk - BM;
    kb = ob & 7;  // 16 consecutive lanes = 8 k-blocks x 2 column blocks: conflict-free ds_write_b128
    const int mb = ob >> 3;
    ld = is_a ? lda : ldb;
    const int col = min((is_a ? m0 : n0) + mb * 8, (is_a ? a_cols : b_cols) - 8);
    col0 = col;
    is_b = !is_a;
    src = (is_a ? A : B) + col;
    lds_off = ((is_a ? 0 : BM) + mb * 8) * kBfPitch + kb * 16;
  }
  __device__ __forceinline__ void load(uint4 (&r)[8], int step) const {
    const int k0 = k_lo + step * kBfBK + kb * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const uint4*>(src + (size_t)min(k0 + j, k_hi - 1) * ld);
  }
  template <class XB>
  __device__ __forceinline__ void store(const uint4 (&r)[8], int step, char* base, const XB& xb) const {
    const int k0 = k_lo + step * kBfBK + kb * 8;
    uint4 m[8], t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool ok = k0 + j < k_hi;
      m[j] = make_uint4(ok ? r[j].x : 0u, ok ? r[j].y : 0u, ok ? r[j].z : 0u, ok ? r[j].w : 0u);
      if constexpr (XB::kActive) {
        if (is_b) {   // (wave-uniform: a wave's 64 threads stage blocks of one operand)
          const u32x4 v = {m[j].x, m[j].y, m[j].z, m[j].w};
          const u32x4 o = xb(v, min(k0 + j, k_hi - 1), col0);
          m[j] = make_uint4(o[0], o[1], o[2], o[3]);
        }
      }
    }
    transpose8x8_b16(m, t);
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4*>(base + lds_off + j * kBfPitch) = t[j];
  }
};

// acc += A[k_lo : k_hi, m0 : m0+BM)^T * B[k_lo : k_hi, n0 : n0+BN).  a_cols / b_cols: widths of A / B (multiples of 8,
// >= 8).  Rows >= k_hi contribute zero.  Ends on a barrier.
template <int BM, int BN, class XB = BfNoTransform>
__device__ __forceinline__ void gemm_bf16_tn_tile(const bf16* __restrict__ A, int lda, int a_cols,
                                                  const bf16* __restrict__ B, int ldb, int b_cols, int m0, int n0,
                                                  int k_lo, int k_hi, char* smem, f32x16 (&acc)[BM / 64][BN / 64],
                                                  const XB& xb = XB()) {
  using T = BfTile<BM, BN>;
  const TnStager<BM, BN> st(A, lda, a_cols, B, ldb, b_cols, m0, n0, k_lo, k_hi);
  const int nsteps = (k_hi - k_lo + kBfBK - 1) / kBfBK;
  uint4 r0[8], r1[8];
  st.load(r0, 0);
  st.store(r0, 0, smem, xb);
  st.load(r1, 1);
  st.load(r0, 2);
  __syncthreads();
  int s = 0;                                   // (pairs of stages, straight-line: see gemm_bf16_nt_tile)
  for (; s + 1 < nsteps; s += 2) {
    bf_stage_mfma<BM, BN>(smem, acc);
    st.store(r1, s + 1, smem + T::kStageBytes, xb);
    st.load(r1, s + 3);
    bf_interleave<4 * T::TM * T::TN, 8>();
    __syncthreads();
    bf_stage_mfma<BM, BN>(smem + T::kStageBytes, acc);
    st.store(r0, s + 2, smem, xb);
    st.load(r0, s + 4);
    bf_interleave<4 * T::TM * T::TN, 8>();
    __syncthreads();
  }
  if (s < nsteps) {
    bf_stage_mfma<BM, BN>(smem, acc);
    __syncthreads();
  }
}

// ---- TN form on transposed LDS reads (ds_read_b64_tr_b16) ---------------------------------------------------------------
// The weight-gradient operands are K-strided (k = batch row): the form above transposes every 8x8 block in registers
// (32 v_perm_b32 + 8 ds_write_b128 per block) to build the [mn][k] image the NT fragments read.  gfx950 can transpose on
// the READ side instead: the tile is stored as it arrives, [k][mn] rows (16-byte global loads along mn, ds_write_b128 of
// the same 16 bytes), and a lane gets the 4 consecutive k of its column with one ds_read_b64_tr_b16 -- two of them are the
// 8-element MFMA operand.  Row pitch = 2 BMN + 64 bytes: the four k-rows a 32-lane half reads (4 x 64 contiguous bytes)
// start 64 bytes apart modulo the 256-byte bank row, so the reads are conflict-free; the writes are 128 contiguous bytes
// per 8 lanes.
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN>
struct BfTileTr {
  static constexpr int TM = BM / 64, TN = BN / 64;
  static constexpr int PA = BM * 2 + 64, PB = BN * 2 + 64;             // row pitches in bytes
  static constexpr int kStageBytes = kBfBK * (PA + PB);
  static constexpr int kSmemBytes = 2 * kStageBytes;
  static constexpr int CA = BM * kBfBK / 8 / kBfThreads, CB = BN * kBfBK / 8 / kBfThreads;   // 16-byte chunks per thread
};

__device__ __forceinline__ bf16x8 bf_tr_frag(const char* p, int pitch) {
  typedef s16x4 __attribute__((address_space(3))) * lds_ptr;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * pitch));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

template <int BM, int BN>
__device__ __forceinline__ void bf_stage_mfma_tr(const char* base, f32x16 (&acc)[BM / 64][BN / 64]) {
  using T = BfTileTr<BM, BN>;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // lane -> (k row inside a 16-deep step, first column) of its transposed 4 x 16 block: lane 4q+p of a 16-lane group
  // addresses row q, columns 4p .. 4p+3; groups 0/1 are the low k half of the step, 2/3 the high one; odd groups take
  // the second 16 columns of the operand's 32
  const int krow = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const int col = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
  const char* ap = base + krow * T::PA + (wm * (BM / 2) + col) * 2;
  const char* bp = base + kBfBK * T::PA + krow * T::PB + (wn * (BN / 2) + col) * 2;
  bf16x8 a[4][T::TM], b[4][T::TN];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
    for (int i = 0; i < T::TM; ++i) a[kk][i] = bf_tr_frag(ap + kk * 16 * T::PA + i * 64, T::PA);
#pragma unroll
    for (int j = 0; j < T::TN; ++j) b[kk][j] = bf_tr_frag(bp + kk * 16 * T::PB + j * 64, T::PB);
  }
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int i = 0; i < T::TM; ++i)
#pragma unroll
      for (int j = 0; j < T::TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[kk][i], b[kk][j], acc[i][j], 0, 0, 0);
}

// Per-thread staging state: chunk c = tid + 256 p of an operand is 16 bytes (8 columns) of tile row c / (BMN/8).
template <int BM, int BN>
struct TrStager {
  using T = BfTileTr<BM, BN>;
  const bf16* a;
  const bf16* b;
  int lda, ldb, k_lo, k_hi;
  int ca[T::CA], cb[T::CB];   // source column of each chunk (clamped to a whole in-range chunk)
  __device__ __forceinline__ TrStager(const bf16* A, int lda_, int a_cols, const bf16* B, int ldb_, int b_cols, int m0, int n0,
                                      int k_lo_, int k_hi_)
      : a(A), b(B), lda(lda_), ldb(ldb_), k_lo(k_lo_), k_hi(k_hi_) {
#pragma unroll
    for (int p = 0; p < T::CA; ++p) ca[p] = min(m0 + ((threadIdx.x + kBfThreads * p) % (BM / 8)) * 8, a_cols - 8);
#pragma unroll
    for (int p = 0; p < T::CB; ++p) cb[p] = min(n0 + ((threadIdx.x + kBfThreads * p) % (BN / 8)) * 8, b_cols - 8);
  }
  __device__ __forceinline__ static int row_a(int p) { return (threadIdx.x + kBfThreads * p) / (BM / 8); }
  __device__ __forceinline__ static int row_b(int p) { return (threadIdx.x + kBfThreads * p) / (BN / 8); }
  __device__ __forceinline__ void load(u32x4 (&ra)[T::CA], u32x4 (&rb)[T::CB], int step) const {
    const int k0 = k_lo + step * kBfBK;
#pragma unroll
    for (int p = 0; p < T::CA; ++p) ra[p] = *reinterpret_cast<const u32x4*>(a + (size_t)min(k0 + row_a(p), k_hi - 1) * lda + ca[p]);
#pragma unroll
    for (int p = 0; p < T::CB; ++p) rb[p] = *reinterpret_cast<const u32x4*>(b + (size_t)min(k0 + row_b(p), k_hi - 1) * ldb + cb[p]);
  }
  template <class XB>
  __device__ __forceinline__ void store(const u32x4 (&ra)[T::CA], const u32x4 (&rb)[T::CB], int step, char* base, int m0, int n0,
                                        const XB& xb) const {
    const int k0 = k_lo + step * kBfBK;
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < T::CA; ++p) {
      const int r = row_a(p), c = ((threadIdx.x + kBfThreads * p) % (BM / 8)) * 8;
      *reinterpret_cast<u32x4*>(base + r * T::PA + c * 2) = k0 + r < k_hi ? ra[p] : z;
    }
    uint32_t words[T::CB];
    if constexpr (XB::kActive) {
      // a lane's CB chunks are ONE column chunk (kBfThreads % (BN/8) == 0) in CB rows, and the four lanes of a quad hold the
      // four chunks of one 32-column group (n0 % 32 == 0, ldb % 32 == 0; a chunk clamped at the matrix edge only feeds output
      // columns that are not stored)
      const int pj = (threadIdx.x & 3) % T::CB;
      bf_quad_words<T::CB>(xb.word(min(k0 + row_b(pj), k_hi - 1), cb[0]), words);
    }
#pragma unroll
    for (int p = 0; p < T::CB; ++p) {
      const int r = row_b(p), c = ((threadIdx.x + kBfThreads * p) % (BN / 8)) * 8;
      u32x4 v = k0 + r < k_hi ? rb[p] : z;
      if constexpr (XB::kActive) v = XB::apply(v, words[p] >> ((uint32_t)cb[0] & 31u));
      *reinterpret_cast<u32x4*>(base + kBfBK * T::PA + r * T::PB + c * 2) = v;
    }
  }
};

// acc += A[k_lo : k_hi, m0 : m0+BM)^T * B[k_lo : k_hi, n0 : n0+BN) -- same contract as gemm_bf16_tn_tile.
template <int BM, int BN, class XB = BfNoTransform>
__device__ __forceinline__ void gemm_bf16_tn_tile_tr(const bf16* __restrict__ A, int lda, int a_cols,
                                                     const bf16* __restrict__ B, int ldb, int b_cols, int m0, int n0,
                                                     int k_lo, int k_hi, char* smem, f32x16 (&acc)[BM / 64][BN / 64],
                                                     const XB& xb = XB()) {
  using T = BfTileTr<BM, BN>;
  const TrStager<BM, BN> st(A, lda, a_cols, B, ldb, b_cols, m0, n0, k_lo, k_hi);
  const int nsteps = (k_hi - k_lo + kBfBK - 1) / kBfBK;
  u32x4 ra0[T::CA], rb0[T::CB], ra1[T::CA], rb1[T::CB];
  st.load(ra0, rb0, 0);
  st.store(ra0, rb0, 0, smem, m0, n0, xb);
  st.load(ra1, rb1, 1);
  st.load(ra0, rb0, 2);
  __syncthreads();
  int s = 0;                                   // (pairs of stages, straight-line: see gemm_bf16_nt_tile)
  for (; s + 1 < nsteps; s += 2) {
    bf_stage_mfma_tr<BM, BN>(smem, acc);
    st.store(ra1, rb1, s + 1, smem + T::kStageBytes, m0, n0, xb);
    st.load(ra1, rb1, s + 3);
    bf_interleave<4 * T::TM * T::TN, 4>();
    __syncthreads();
    bf_stage_mfma_tr<BM, BN>(smem + T::kStageBytes, acc);
    st.store(ra0, rb0, s + 2, smem, m0, n0, xb);
    st.load(ra0, rb0, s + 4);
    bf_interleave<4 * T::TM * T::TN, 4>();
    __syncthreads();
  }
  if (s < nsteps) {
    bf_stage_mfma_tr<BM, BN>(smem, acc);
    __syncthreads();
  }
}

// Row / column of accumulator register i of tile (tm, tn) for the calling lane (C/D layout of the 32x32 MFMAs:
// col = lane & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)).  Split into a wave-uniform part (SGPRs: the
// epilogue addresses become scalar base + one 32-bit lane offset instead of a 64-bit VGPR pair per element) and the
// lane's own part.
template <int BM, int BN>
struct BfAccCoord {
  int urow0, ucol0;  // wave-uniform: first row / column of the wave's sub-tile
  int lrow, lcol;    // lane part
  __device__ __forceinline__ BfAccCoord(int m0, int n0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    urow0 = m0 + (wave >> 1) * (BM / 2);
    ucol0 = n0 + (wave & 1) * (BN / 2);
    lrow = 4 * (lane >> 5);
    lcol = lane & 31;
  }
  __device__ __forceinline__ int urow(int tm, int i) const { return urow0 + tm * 32 + (i & 3) + 8 * (i >> 2); }
  __device__ __forceinline__ int ucol(int tn) const { return ucol0 + tn * 32; }
  __device__ __forceinline__ int row(int tm, int i) const { return urow(tm, i) + lrow; }
  __device__ __forceinline__ int col(int tn) const { return ucol(tn) + lcol; }
  // element (row, col) of a row-major matrix with leading dimension ld, as uniform offset + lane offset
  __device__ __forceinline__ size_t uoff(int tm, int tn, int i, int ld) const { return (size_t)urow(tm, i) * ld + ucol(tn); }
  __device__ __forceinline__ unsigned loff(int ld) const { return (unsigned)(lrow * ld + lcol); }
};

// Coalesced bf16 store of a BM x BN tile of fp32 values held in the accumulator layout: the waves write their values
// to an LDS image [BM][BN] bf16 (pitch BN*2 + 16 bytes: the two lane halves of a ds_write_b16 land 4 rows = 16 banks
// apart), then every thread stores 16-byte chunks of whole rows -- BN*2 contiguous bytes per row instead of 2-byte
// scattered stores, which are what bound the wide-output kernels.  `tile_s` needs BM * (BN*2 + 16) bytes and must not
// alias anything live; both barriers are inside.  rows_valid = number of tile rows inside the matrix.
template <int BM, int BN>
struct BfTileStore {
  static constexpr int kPitch = BN * 2 + 16;
  static constexpr int kBytes = BM * kPitch;
  // gate (optional, same tile origin as dst, row stride ldg, 16-byte aligned rows): elements whose gate value is not > 0
  // are stored as zero (a relu gradient applied in the store)
  template <class F>
  __device__ __forceinline__ static void run(char* tile_s, bf16* __restrict__ dst, size_t ld, int rows_valid, F value,
                                             const bf16* __restrict__ gate = nullptr, size_t ldg = 0) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r0 = (wave >> 1) * (BM / 2) + 4 * (lane >> 5), c0 = (wave & 1) * (BN / 2) + (lane & 31);
#pragma unroll
    for (int tm = 0; tm < BM / 64; ++tm)
#pragma unroll
      for (int tn = 0; tn < BN / 64; ++tn)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = r0 + tm * 32 + (i & 3) + 8 * (i >> 2), col = c0 + tn * 32;
          *reinterpret_cast<bf16*>(tile_s + row * kPitch + col * 2) = (bf16)value(tm, tn, i);
        }
    __syncthreads();
    constexpr int CPR = BN / 8;                 // 16-byte chunks per row
    constexpr int RPP = kBfThreads / CPR;       // rows per pass
    const int cr = threadIdx.x / CPR, cc = threadIdx.x % CPR;
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
      const int row = p * RPP + cr;
      if (row < rows_valid) {
        u32x4 val = *reinterpret_cast<const u32x4*>(tile_s + row * kPitch + cc * 16);
        if (gate != nullptr) {
          const u32x4 gv = *reinterpret_cast<const u32x4*>(gate + (size_t)row * ldg + cc * 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) {   // bf16 > 0  <=>  sign clear and magnitude bits non-zero
            const uint32_t lo = gv[i] & 0xFFFFu, hi = gv[i] >> 16;
            const uint32_t keep_lo = (lo - 1u) < 0x7FFFu ? 0x0000FFFFu : 0u, keep_hi = (hi - 1u) < 0x7FFFu ? 0xFFFF0000u : 0u;
            val[i] &= keep_lo | keep_hi;
          }
        }
        *reinterpret_cast<u32x4*>(dst + (size_t)row * ld + cc * 8) = val;
      }
    }
    __syncthreads();
  }
};

template <int TM, int TN>
__device__ __forceinline__ void bf_zero_acc(f32x16 (&acc)[TM][TN]) {
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
}

}  // namespace vqa
