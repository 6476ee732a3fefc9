// Mixed-precision (bf16 storage / bf16 MFMA operands / fp32 accumulate) side of the library -- BASELINE configs[4]:
// "CoR2 bf16, 100x2048 dense regions".  Holds the generic bf16 GEMMs on the tile engine of gemm_bf16_mfma.hpp and
// K4 (low-rank bilinear fusion, putils/__init__.py:232-238) built on them.
//
// Operand contract: feature dims are padded to a multiple of 64 with zeros by the caller (the Python host keeps bf16
// shadow copies of the fp32 master weights in this layout), so the kernels carry no tail handling; row counts
// (M = B*N) are free.
//
//   K4 forward : out[m,:] = sum_r (x[m,:] W1_r^T + b1_r) * h2[b(m),r,:]       NT GEMM per rank, fused epilogue
//   K4 backward: prep  gs[m,r,:] = g[m,:] * h2[b(m),r,:]  (bf16),  dh2[b,r,:] = sum_n g[b,n,:] * h1[b,n,r,:],
//                      gsum[b,:] = sum_n g[b,n,:]          -- one streaming pass over g and h1
//                db1[r,:] = sum_b h2[b,r,:] * gsum[b,:]
//                dx  = gs[M, R*H] * W1t[L, R*H]^T           one NT GEMM over the concatenated rank axis
//                dW1 = gs[M, R*H]^T * x[M, L]               one TN GEMM, split over M into fp32 slabs, fixed-order reduce
#include <cstdlib>

#include "gemm_bf16_mfma.hpp"

namespace vqa {

constexpr int kBfMaxR = 8;

struct BfTileChoice {
  int bm, bn;
};
static bool bf_tile_override(BfTileChoice* t) {
  if (const char* e = vqa::option("VQA_BF16_TILE")) {  // experiment knob, e.g. "128x64"
    int bm = 0, bn = 0;
    if (std::sscanf(e, "%dx%d", &bm, &bn) == 2 && (bm == 64 || bm == 128) && (bn == 64 || bn == 128)) {
      *t = {bm, bn};
      return true;
    }
  }
  return false;
}
// Tile shapes from the sweeps of tools/kbench.py --only bf16 (M = 12800 rows):
//   NT (forward / data-gradient form): 128x64 -- a 64-wide tile keeps N = 320 at 5 exact column tiles and 3 workgroups
//      per CU; with a short K (<= 512: five stages) and a wide output (the data gradient [M,320] x [2048,320]^T) the
//      128x128 tile wins since the output leaves through the LDS image in 16-byte row stores (34.0 us against 39.7 for
//      64x64, which won while the stores were 2-byte scattered; hipBLASLt: 41.6).
//   TN (weight-gradient form): 128x128 -- every staged 8x8 block costs 32 v_perm + 8 ds_write_b128, so the tile with
//      the most MFMAs per staged element wins by 1.5-2x even where it pads N2 = 320 to 384.
static BfTileChoice choose_bf_tile(long M, long N, long K = 1 << 20) {
  BfTileChoice t{128, 64};
  if (bf_tile_override(&t)) return t;
  if (M <= 64) t.bm = 64;
  if (K <= 512 && N >= 1024) t.bn = 128;
  return t;
}
static BfTileChoice choose_bf_tile_tn(long N1, long N2) {
  BfTileChoice t{N1 > 64 ? 128 : 64, N2 > 64 ? 128 : 64};
  bf_tile_override(&t);
  return t;
}

#define VQA_BF_TILE_SWITCH(t, LAUNCH)            \
  do {                                           \
    if ((t).bm == 128 && (t).bn == 128) {        \
      LAUNCH(128, 128)                           \
    } else if ((t).bm == 128 && (t).bn == 64) {  \
      LAUNCH(128, 64)                            \
    } else if ((t).bm == 64 && (t).bn == 128) {  \
      LAUNCH(64, 128)                            \
    } else {                                     \
      LAUNCH(64, 64)                             \
    }                                            \
  } while (0)

// ------------------------------------------------------------------------------------- generic NT
// C[M,N] (bf16, row stride ldc) = act(scale * XA(A)[M,K] * B[N,K]^T + bias[N]);  act: 0 none, 1 relu.
// XA: operand transform on A while it is staged (BfDropHalf: the p = 0.5 input dropout of MyConv1d, config/CoR2.py:72-75,
// whose factor 2 is `scale`).  gate (optional, [M, ldg] bf16): the stored value is zeroed where gate[m,n] <= 0 -- the relu
// gradient of the layer that produced `gate`, applied where its data gradient is written.
template <int BM, int BN, class XA>
__global__ __launch_bounds__(kBfThreads) void gemm_bf16_nt_kernel(const bf16* __restrict__ A, int lda,
                                                                  const bf16* __restrict__ B, int ldb,
                                                                  const float* __restrict__ bias, bf16* __restrict__ C,
                                                                  int ldc, int M, int N, int K, int act, int tiles_n,
                                                                  float scale, const bf16* __restrict__ gate, int ldg,
                                                                  DropCfg dc, uint32_t mask_ld) {
  using T = BfTile<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  f32x16 acc[T::TM][T::TN];
  bf_zero_acc(acc);
  if constexpr (XA::kActive) {
    gemm_bf16_nt_tile<BM, BN, XA>(A, lda, M, B, ldb, N, m0, n0, K, smem, acc, XA{drop_key(dc), mask_ld});
  } else {
    gemm_bf16_nt_tile<BM, BN>(A, lda, M, B, ldb, N, m0, n0, K, smem, acc);
  }
  const BfAccCoord<BM, BN> cc(m0, n0);
  if (n0 + BN <= N && (ldc & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
      (gate == nullptr || ((ldg & 7) == 0 && (reinterpret_cast<uintptr_t>(gate) & 15) == 0))) {
    // whole column tile inside the matrix: coalesced 16-byte row stores through an LDS image (the staging ring is idle)
    float bv[T::TN];
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) bv[tn] = bias != nullptr ? bias[cc.col(tn)] : 0.f;
    BfTileStore<BM, BN>::run(
        smem, C + (size_t)m0 * ldc + n0, ldc, M - m0,
        [&](int tm, int tn, int i) {
          const float y = fmaf(acc[tm][tn][i], scale, bv[tn]);
          return act == 1 ? fmaxf(y, 0.f) : y;
        },
        gate != nullptr ? gate + (size_t)m0 * ldg + n0 : nullptr, (size_t)ldg);
    return;
  }
  const unsigned lo = cc.loff(ldc);
  const unsigned lg = cc.loff(ldg);
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    const int col = cc.col(tn);
    if (col < N) {
      const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float y = fmaf(acc[tm][tn][i], scale, bv);
          if (act == 1) y = fmaxf(y, 0.f);
          if (cc.row(tm, i) < M) {
            if (gate != nullptr && !((float)(gate + cc.uoff(tm, tn, i, ldg))[lg] > 0.f)) y = 0.f;
            (C + cc.uoff(tm, tn, i, ldc))[lo] = (bf16)y;
          }
        }
    }
  }
}

// ------------------------------------------------------------------------------------- generic TN
// slab[s][N1][N2] (fp32) = A[rows of split s, 0:N1)^T * B[rows of split s, 0:N2)
// XB: operand transform on B while it is staged (BfDropHalf: B is the dropped-out input of the layer whose weight gradient
// this is; the factor 2 is applied by the slab reduction).
// TR: stage the operands untransposed and transpose on the LDS read (ds_read_b64_tr_b16; default) instead of in registers
template <int BM, int BN, class XB, bool TR>
__global__ __launch_bounds__(kBfThreads) void gemm_bf16_tn_kernel(const bf16* __restrict__ A, int lda,
                                                                  const bf16* __restrict__ B, int ldb,
                                                                  float* __restrict__ slab, int Kdim, int N1, int N2,
                                                                  int rows_per_split, int tiles_n, DropCfg dc,
                                                                  uint32_t mask_ld, int tiles_m_fast) {
  using T = BfTile<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  // neighbouring workgroups (one XCD, one L2) walk the axis with FEWER tiles first: they share the tile of the operand that
  // is split into more tiles -- the larger matrix ([K, 2048] regions against [K, 320] gradients), fetched once per XCD
  const int tm_ = tiles_m_fast > 0 ? bid % tiles_m_fast : bid / tiles_n;
  const int tn_ = tiles_m_fast > 0 ? bid / tiles_m_fast : bid % tiles_n;
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const int s = blockIdx.z;
  const int k_lo = s * rows_per_split, k_hi = min(Kdim, k_lo + rows_per_split);
  f32x16 acc[T::TM][T::TN];
  bf_zero_acc(acc);
  if constexpr (TR) {
    if constexpr (XB::kActive) {
      gemm_bf16_tn_tile_tr<BM, BN, XB>(A, lda, N1, B, ldb, N2, m0, n0, k_lo, k_hi, smem, acc, XB{drop_key(dc), mask_ld});
    } else {
      gemm_bf16_tn_tile_tr<BM, BN>(A, lda, N1, B, ldb, N2, m0, n0, k_lo, k_hi, smem, acc);
    }
  } else if constexpr (XB::kActive) {
    gemm_bf16_tn_tile<BM, BN, XB>(A, lda, N1, B, ldb, N2, m0, n0, k_lo, k_hi, smem, acc, XB{drop_key(dc), mask_ld});
  } else {
    gemm_bf16_tn_tile<BM, BN>(A, lda, N1, B, ldb, N2, m0, n0, k_lo, k_hi, smem, acc);
  }
  float* __restrict__ dst = slab + (size_t)s * N1 * N2;
  const BfAccCoord<BM, BN> cc(m0, n0);
  const unsigned lo = cc.loff(N2);
#pragma unroll
  for (int tn = 0; tn < T::TN; ++tn) {
    if (cc.col(tn) < N2) {
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (cc.row(tm, i) < N1) (dst + cc.uoff(tm, tn, i, N2))[lo] = acc[tm][tn][i];
    }
  }
}

// Fixed-order (bitwise reproducible) reduction of the S slabs [N1, N2], written CROPPED and per group: row n1 = g * gp + r
// (r < out_rows) goes to outs[g][r * out_ld + c] for c < out_cols, times `scale`.  One group with gp = N1 is the plain
// case; K4 hands R groups (one nn.Linear weight gradient each), a padded layer a single cropped one -- the gradients land
// in the master-shaped tensors the optimizer reads, with no slice / copy kernels behind the GEMM.
struct BfOuts {
  float* p[kBfMaxR];
};
// An optional second job of the slab-reduction launch (K4's weight gradient): the bias gradients of the R rank layers,
// db1[r][h] = sum_b h2[b,r,h] * gsum[b,h] -- 256 lanes = 64 columns x 4 sample slices (a serial loop over B per column is a
// chain of B dependent L2 round trips: 36 us at B = 128); the slices meet in LDS, fixed order.  Until round 4 a launch of
// its own (4.8 us at the launch floor, twice per step); now the blocks behind the reduction's own rows.  (The plain layers'
// bias gradient -- column sums of a [12800, 320] bf16 tensor -- was tried as such a job too: its 40 blocks walk 50 rows per
// lane and take 24 us, longer than the two launches they replace; it keeps its column_sum kernels.)
struct BfDbJob {
  const float* h2 = nullptr;     // [B, R, Hout] (unpadded); nullptr: no job
  const float* gsum = nullptr;   // [B, H]
  BfOuts db1{};                  // R vectors of Hout values: the nn.Linear bias gradients themselves
  int B = 0, H = 0, R = 0, Hout = 0;
  int blocks = 0;                // R * H / 64
};
__device__ __forceinline__ void bilinear_db_block(const BfDbJob& j, int block) {
  __shared__ float part[3][64];
  const float* __restrict__ h2 = j.h2;
  const float* __restrict__ gsum = j.gsum;
  const int B = j.B, H = j.H, R = j.R, Hout = j.Hout;
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int e = block * 64 + c;  // < R*H: H % 256 == 0
  const int h = e % H;
  const int eo = (e / H) * Hout + min(h, Hout - 1);   // (pad columns: a clamped read, not stored)
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int b = slice;
  for (; b + 12 < B; b += 16) {
    a0 = fmaf(h2[(size_t)b * R * Hout + eo], gsum[(size_t)b * H + h], a0);
    a1 = fmaf(h2[(size_t)(b + 4) * R * Hout + eo], gsum[(size_t)(b + 4) * H + h], a1);
    a2 = fmaf(h2[(size_t)(b + 8) * R * Hout + eo], gsum[(size_t)(b + 8) * H + h], a2);
    a3 = fmaf(h2[(size_t)(b + 12) * R * Hout + eo], gsum[(size_t)(b + 12) * H + h], a3);
  }
  for (; b < B; b += 4) a0 = fmaf(h2[(size_t)b * R * Hout + eo], gsum[(size_t)b * H + h], a0);
  const float a = (a0 + a1) + (a2 + a3);
  if (slice > 0) part[slice - 1][c] = a;
  __syncthreads();
  if (slice == 0 && h < Hout) j.db1.p[e / H][h] = a + part[0][c] + part[1][c] + part[2][c];
}
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slab, BfOuts outs, int S, int N1, int N2,
                                                          int gp, int out_rows, int out_cols, int out_ld, float scale,
                                                          BfDbJob dbjob) {
  if ((int)blockIdx.y >= N1) {            // (block-uniform) the rows behind the reduction's own: the bias-gradient job
    const int block = ((int)blockIdx.y - N1) * (int)gridDim.x + (int)blockIdx.x;
    if (block < dbjob.blocks) bilinear_db_block(dbjob, block);
    return;
  }
  const int c = (blockIdx.x * 256 + threadIdx.x) * 2;
  const int n1 = blockIdx.y;
  const int g = n1 / gp, r = n1 - g * gp;
  if (c >= out_cols || r >= out_rows) return;
  const size_t e = (size_t)n1 * N2 + c, count = (size_t)N1 * N2;
  float2 a = ld2(slab + e);
  for (int s = 1; s < S; ++s) {
    const float2 t = ld2(slab + (size_t)s * count + e);
    a.x += t.x;
    a.y += t.y;
  }
  float* o = outs.p[g] + (size_t)r * out_ld + c;
  o[0] = a.x * scale;
  if (c + 1 < out_cols) o[1] = a.y * scale;
}

static int tn_splits(int Kdim, int N1, int N2, BfTileChoice t) {
  const long tiles = (long)((N1 + t.bm - 1) / t.bm) * ((N2 + t.bn - 1) / t.bn);
  // Workgroups a launch aims for.  The 128x128 transposed-read tile holds 80 KB of LDS: ONE workgroup per CU, so the split
  // count is the largest that keeps the grid within one round of the 256 CUs (measured at K = 12800, B = 128 x 100 regions:
  // [320 x 2048] 48.1 -> 41.4 us, [1024 x 320] 28.1 -> 21.8 us against the former "at least 512 workgroups", and the slab
  // reduction behind it shrinks with the split count); the smaller tiles fit two or more per CU.
  const bool one_per_cu = t.bm == 128 && t.bn == 128;
  const long want = one_per_cu ? 256 : 512;
  long s = one_per_cu ? want / tiles : (want + tiles - 1) / tiles;
  const long max_by_rows = (Kdim + 511) / 512;  // keep >= 512 rows (8 stages) per split
  if (s > max_by_rows) s = max_by_rows;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  return (int)s;
}
static BfTileChoice tn_tile(int N1, int N2) { return choose_bf_tile_tn(N1, N2); }

struct NtExtra {          // optional parts of an NT launch
  float scale = 1.f;      // factor on the accumulator (1/(1-p) of a dropout applied to A)
  const bf16* gate = nullptr;
  int ldg = 0;
  bool drop = false;      // A is dropped out in its p = 0.5 one-bit form while staged
  DropCfg dc{};
  uint32_t mask_ld = 0;
};

static int launch_nt(const char* who, const bf16* A, int lda, const bf16* B, int ldb, const float* bias, bf16* C, int ldc,
                     int M, int N, int K, int act, hipStream_t s, const NtExtra& ex = NtExtra()) {
  const BfTileChoice t = choose_bf_tile(M, N, K);
  const int tiles_m = (M + t.bm - 1) / t.bm, tiles_n = (N + t.bn - 1) / t.bn;
#define LAUNCH_X(BM_, BN_, XA_)                                                                                            \
  {                                                                                                                        \
    const size_t lds = BfTile<BM_, BN_>::kSmemBytes;                                                                       \
    VQA_ENSURE_LDS((gemm_bf16_nt_kernel<BM_, BN_, XA_>), lds);                                                             \
    VQA_LAUNCH((gemm_bf16_nt_kernel<BM_, BN_, XA_>), dim3(tiles_m * tiles_n), dim3(kBfThreads), lds, s, A, lda, B, \
                       ldb, bias, C, ldc, M, N, K, act, tiles_n, ex.scale, ex.gate, ex.ldg, ex.dc, ex.mask_ld);            \
  }
#define LAUNCH(BM_, BN_)                   \
  if (ex.drop) {                           \
    LAUNCH_X(BM_, BN_, BfDropHalf)         \
  } else {                                 \
    LAUNCH_X(BM_, BN_, BfNoTransform)      \
  }
  VQA_BF_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
#undef LAUNCH_X
  return check_launch(who);
}

static size_t tn_workspace_bytes(int Kdim, int N1, int N2) {   // always at least one slab: the reduction also crops
  const int S = tn_splits(Kdim, N1, N2, tn_tile(N1, N2));
  return (size_t)S * N1 * N2 * sizeof(float);
}

struct TnExtra {
  float scale = 1.f;
  bool drop = false;      // B is dropped out (p = 0.5 one-bit form) while staged
  DropCfg dc{};
  uint32_t mask_ld = 0;
};

// outs: `groups` output matrices [out_rows, out_cols] (row stride out_ld); row n1 of the product belongs to group n1 / gp.
static int launch_tn(const char* who, const bf16* A, int lda, const bf16* B, int ldb, const BfOuts& outs, int groups, int gp,
                     int out_rows, int out_cols, int out_ld, float* workspace, int Kdim, int N1, int N2, hipStream_t s,
                     const TnExtra& ex = TnExtra(), const BfDbJob& dbjob = BfDbJob()) {
  const BfTileChoice t = tn_tile(N1, N2);
  const int S = tn_splits(Kdim, N1, N2, t);
  const int tiles_m = (N1 + t.bm - 1) / t.bm, tiles_n = (N2 + t.bn - 1) / t.bn;
  int rows_per_split = (Kdim + S - 1) / S;
  rows_per_split = (rows_per_split + kBfBK - 1) / kBfBK * kBfBK;
  const int m_fast = tiles_m < tiles_n ? tiles_m : 0;    // (measured: [320 x 2048], K = 12800: 41 -> 38.5 us)
  const char* form = vqa::option("VQA_BF16_TN");           // "perm": the register-transpose staging (comparison knob)
  const bool tr = !(form != nullptr && form[0] == 'p');
#define LAUNCH_T(BM_, BN_, XB_, TR_)                                                                                        \
  {                                                                                                                         \
    const size_t lds = TR_ ? (size_t)BfTileTr<BM_, BN_>::kSmemBytes : (size_t)BfTile<BM_, BN_>::kSmemBytes;                 \
    VQA_ENSURE_LDS((gemm_bf16_tn_kernel<BM_, BN_, XB_, TR_>), lds);                                                         \
    VQA_LAUNCH((gemm_bf16_tn_kernel<BM_, BN_, XB_, TR_>), dim3(tiles_m * tiles_n, 1, S), dim3(kBfThreads), lds, s,  \
                       A, lda, B, ldb, workspace, Kdim, N1, N2, rows_per_split, tiles_n, ex.dc, ex.mask_ld, m_fast);        \
  }
#define LAUNCH_X(BM_, BN_, XB_)    \
  if (tr) {                        \
    LAUNCH_T(BM_, BN_, XB_, true)  \
  } else {                         \
    LAUNCH_T(BM_, BN_, XB_, false) \
  }
#define LAUNCH(BM_, BN_)                   \
  if (ex.drop) {                           \
    LAUNCH_X(BM_, BN_, BfDropHalf)         \
  } else {                                 \
    LAUNCH_X(BM_, BN_, BfNoTransform)      \
  }
  VQA_BF_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
#undef LAUNCH_X
#undef LAUNCH_T
  (void)groups;
  const unsigned gx = (unsigned)((out_cols + 511) / 512);
  const unsigned extra = dbjob.h2 != nullptr ? ((unsigned)dbjob.blocks + gx - 1) / gx : 0u;
  VQA_LAUNCH(slab_reduce_kernel, dim3(gx, (unsigned)N1 + extra), dim3(256), 0, s, workspace, outs, S, N1, N2, gp, out_rows,
             out_cols, out_ld, ex.scale, dbjob);
  return check_launch(who);
}

// ------------------------------------------------------------------------------------------ K4 forward
// The question-side factor of the epilogue is staged through LDS: a BM-row tile touches only the samples
// b0 .. b0 + ceil(BM/N), so their h2[b, r, n0 : n0+BN) rows are loaded once per (tile, rank), coalesced, into the
// (idle) staging buffers, and every accumulator register finds its multiplier with one ds_read_b32.
template <int BM, int BN>
__global__ __launch_bounds__(kBfThreads) void bilinear_fwd_bf16_kernel(const bf16* __restrict__ x,
                                                                       const bf16* __restrict__ w1,
                                                                       const float* __restrict__ b1,
                                                                       const float* __restrict__ h2, bf16* __restrict__ out,
                                                                       bf16* __restrict__ h1, int M, int N, int L, int H,
                                                                       int R, int tiles_n, int Hin) {
  using T = BfTile<BM, BN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* h2_s = reinterpret_cast<float*>(smem);                  // [samples in tile][BN], overlays the staging ring
  int* rowoff_s = reinterpret_cast<int*>(smem + T::kSmemBytes);  // [BM] (sample of the row - b0) * BN
  char* tile_s = smem + T::kSmemBytes + BM * sizeof(int);        // bf16 output image for the coalesced row stores
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int b0 = m0 / N;
  const int ns = min(m0 + BM - 1, M - 1) / N - b0 + 1;  // samples touched by this tile (<= BM)
  for (int t = threadIdx.x; t < BM; t += kBfThreads) rowoff_s[t] = (min(m0 + t, M - 1) / N - b0) * BN;
  // (visible after the first barrier inside the tile loop)
  const BfAccCoord<BM, BN> cc(m0, n0);
  const unsigned lo_h1 = cc.loff(R * H);
  f32x16 total[T::TM][T::TN];
  bf_zero_acc(total);
  for (int r = 0; r < R; ++r) {
    f32x16 acc[T::TM][T::TN];
    bf_zero_acc(acc);
    gemm_bf16_nt_tile<BM, BN>(x, L, M, w1 + (size_t)r * H * L, L, H, m0, n0, L, smem, acc);
    for (int t = threadIdx.x; t < ns * BN; t += kBfThreads) {   // h2 is the unpadded [B,R,Hin]: the pad columns multiply by 0
      const int col = n0 + (t % BN);
      h2_s[t] = col < Hin ? h2[((size_t)(b0 + t / BN) * R + r) * Hin + col] : 0.f;
    }
    __syncthreads();
    // epilogue of rank r: h1_r = acc + b1_r (kept in acc), total += h1_r * h2[b(row), r, :]
#pragma unroll
    for (int tn = 0; tn < T::TN; ++tn) {
      const int col = cc.col(tn);  // < H: H % 256 == 0 and the grid covers exactly H columns
      const float bv = b1[(size_t)r * H + col];
      const float* h2c = h2_s + (col - n0);
#pragma unroll
      for (int tm = 0; tm < T::TM; ++tm) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          const float hv = acc[tm][tn][i] + bv;
          if (h1 != nullptr && row < M) (h1 + (cc.uoff(tm, tn, i, R * H) + (size_t)r * H))[lo_h1] = (bf16)hv;
          total[tm][tn][i] = fmaf(hv, h2c[rowoff_s[row - m0]], total[tm][tn][i]);
        }
      }
    }
    __syncthreads();  // h2_s is the next rank's staging buffer
  }
  BfTileStore<BM, BN>::run(tile_s, out + (size_t)m0 * H + n0, (size_t)H, M - m0,
                           [&](int tm, int tn, int i) { return total[tm][tn][i]; });
}

// Two ranks (every region fusion of the models): both ranks' weight rows of a 64-column block are ONE 128-row B tile,
// interleaved in blocks of 32 so that a wave's two 32-column accumulators are rank 0 and rank 1 of the same columns
// (NtStager, b_interleave): x is staged once instead of once per rank and a stage carries 16 MFMAs per wave instead of 8.
// LDS: the 128x128 ring only -- h2 (both ranks) and the output image overlay it after the K loop (N >= 2 regions per sample,
// so a 128-row tile touches <= 65 samples: 33 KB + 18 KB of the ring's 72).
__global__ __launch_bounds__(kBfThreads) void bilinear_fwd2_bf16_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w1,
                                                                        const float* __restrict__ b1,
                                                                        const float* __restrict__ h2, bf16* __restrict__ out,
                                                                        bf16* __restrict__ h1, int M, int N, int L, int H,
                                                                        int tiles_n, int Hin) {
  constexpr int BM = 128, BC = 64, R = 2;
  using T = BfTile<BM, 128>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* rowoff_s = reinterpret_cast<int*>(smem + T::kSmemBytes);  // [BM] (sample of the row - b0) * BC
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BC;
  const int b0 = m0 / N;
  const int ns = min(m0 + BM - 1, M - 1) / N - b0 + 1;  // samples touched by this tile
  float* h2_s = reinterpret_cast<float*>(smem);                                   // [R][ns][BC]
  char* tile_s = smem + (((size_t)R * ns * BC * sizeof(float) + 15) / 16) * 16;   // bf16 output image
  for (int t = threadIdx.x; t < BM; t += kBfThreads) rowoff_s[t] = (min(m0 + t, M - 1) / N - b0) * BC;
  f32x16 acc[2][2];
  bf_zero_acc(acc);
  gemm_bf16_nt_tile<BM, 128>(x, L, M, w1, L, R * H, m0, n0, L, smem, acc, BfNoTransform(), H);
  for (int t = threadIdx.x; t < R * ns * BC; t += kBfThreads) {   // h2 is the unpadded [B,R,Hin]: the pad columns multiply by 0
    const int r = t / (ns * BC), u = t - r * ns * BC;
    const int col = n0 + (u % BC);
    h2_s[t] = col < Hin ? h2[((size_t)(b0 + u / BC) * R + r) * Hin + col] : 0.f;
  }
  __syncthreads();
  const BfAccCoord<BM, BC> cc(m0, n0);
  const unsigned lo_h1 = cc.loff(R * H);
  const int col = cc.col(0);  // < H
  const float bv0 = b1[col], bv1 = b1[(size_t)H + col];
  const float* h2c0 = h2_s + (col - n0);
  const float* h2c1 = h2c0 + ns * BC;
#pragma unroll
  for (int tm = 0; tm < 2; ++tm) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = cc.row(tm, i);
      const float hv0 = acc[tm][0][i] + bv0, hv1 = acc[tm][1][i] + bv1;
      if (h1 != nullptr && row < M) {
        bf16* hp = h1 + cc.uoff(tm, 0, i, R * H);
        hp[lo_h1] = (bf16)hv0;
        (hp + H)[lo_h1] = (bf16)hv1;
      }
      const int ro = rowoff_s[row - m0];
      acc[tm][0][i] = fmaf(hv0, h2c0[ro], hv1 * h2c1[ro]);
    }
  }
  BfTileStore<BM, BC>::run(tile_s, out + (size_t)m0 * H + n0, (size_t)H, M - m0,
                           [&](int tm, int, int i) { return acc[tm][0][i]; });
}

// ------------------------------------------------------------------------------------------ K4 backward prep
// 256 lanes = (256/SL) columns-of-4 x SL region slices; the slices meet in LDS.  SL = 4 at large batches (grid
// (H/256, B)); SL = 16 (grid (H/64, B)) when that grid would leave the chip short of loads in flight.
template <int SL>
__global__ __launch_bounds__(256) void bilinear_bwd_prep_bf16_kernel(const bf16* __restrict__ g, const bf16* __restrict__ h1,
                                                                     const float* __restrict__ h2, bf16* __restrict__ gs,
                                                                     float* __restrict__ dh2, float* __restrict__ gsum,
                                                                     int N, int H, int R, int Hin) {
  constexpr int COLS = 256 / SL;
  __shared__ float4 part[SL - 1][kBfMaxR + 1][COLS];
  const int b = blockIdx.y;
  const int c = threadIdx.x % COLS, slice = threadIdx.x / COLS;
  const int h = (blockIdx.x * COLS + c) * 4;  // < H: H % 256 == 0
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 acc[kBfMaxR], q[kBfMaxR], gt = z;
#pragma unroll
  for (int r = 0; r < kBfMaxR; ++r) {
    acc[r] = z;
    if (r < R) {   // h2 / dh2 are the unpadded [B,R,Hin]
      const float* hr = h2 + ((size_t)b * R + r) * Hin;
      q[r] = make_float4(h < Hin ? hr[h] : 0.f, h + 1 < Hin ? hr[h + 1] : 0.f, h + 2 < Hin ? hr[h + 2] : 0.f, h + 3 < Hin ? hr[h + 3] : 0.f);
    } else {
      q[r] = z;
    }
  }
#pragma unroll 4
  for (int n = slice; n < N; n += SL) {
    const size_t m = (size_t)b * N + n;
    const float4 gv = ld4(g + m * H + h);
    gt = add4(gt, gv);
#pragma unroll
    for (int r = 0; r < kBfMaxR; ++r) {
      if (r < R) {
        const float4 hv = ld4(h1 + (m * R + r) * H + h);
        acc[r] = add4(acc[r], mul4(gv, hv));
        st4(gs + (m * R + r) * H + h, mul4(gv, q[r]));
      }
    }
  }
  if (slice > 0) {
#pragma unroll
    for (int r = 0; r < kBfMaxR; ++r)
      if (r < R) part[slice - 1][r][c] = acc[r];
    part[slice - 1][kBfMaxR][c] = gt;
  }
  __syncthreads();
  if (slice == 0) {
#pragma unroll
    for (int r = 0; r < kBfMaxR; ++r) {
      if (r < R) {
        float4 t = acc[r];
        for (int s = 0; s < SL - 1; ++s) t = add4(t, part[s][r][c]);
        float* dr = dh2 + ((size_t)b * R + r) * Hin;
        if (h < Hin) dr[h] = t.x;
        if (h + 1 < Hin) dr[h + 1] = t.y;
        if (h + 2 < Hin) dr[h + 2] = t.z;
        if (h + 3 < Hin) dr[h + 3] = t.w;
      }
    }
    for (int s = 0; s < SL - 1; ++s) gt = add4(gt, part[s][kBfMaxR][c]);
    st4(gsum + (size_t)b * H + h, gt);
  }
}

// The same for a compile-time rank count (1..5: every fusion of the models) with 16-byte accesses: 256 lanes = 16 column
// groups of 8 (128 columns) x 16 region slices, grid (H/128, B); a lane's loads of one region are independent of the next
// region's, so the unrolled loop keeps ~3 (1 + RT) x 16 B x 4 per lane in flight.  (The 8-byte form above moved 65 MB in
// 30 us at B = 128, N = 100 -- 2.2 TB/s.)
template <int RT>
__global__ __launch_bounds__(256) void bilinear_bwd_prep8_bf16_kernel(const bf16* __restrict__ g, const bf16* __restrict__ h1,
                                                                      const float* __restrict__ h2, bf16* __restrict__ gs,
                                                                      float* __restrict__ dh2, float* __restrict__ gsum,
                                                                      int N, int H, int Hin) {
  constexpr int SL = 16, COLS = 16;
  __shared__ float part[SL - 1][RT + 1][COLS][8];
  const int b = blockIdx.y;
  const int c = threadIdx.x % COLS, slice = threadIdx.x / COLS;
  const int h = (blockIdx.x * COLS + c) * 8;  // < H: H % 256 == 0
  float q[RT][8], acc[RT][8], gt[8];
#pragma unroll
  for (int r = 0; r < RT; ++r) {
    const float* hr = h2 + ((size_t)b * RT + r) * Hin;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      q[r][j] = h + j < Hin ? hr[h + j] : 0.f;
      acc[r][j] = 0.f;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) gt[j] = 0.f;
  const auto unpack = [](u32x4 w, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = bf16_lo(w[i]);
      v[2 * i + 1] = bf16_hi(w[i]);
    }
  };
#pragma unroll 4
  for (int n = slice; n < N; n += SL) {
    const size_t m = (size_t)b * N + n;
    float gv[8];
    unpack(*reinterpret_cast<const u32x4*>(g + m * H + h), gv);
#pragma unroll
    for (int j = 0; j < 8; ++j) gt[j] += gv[j];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      float hv[8];
      unpack(*reinterpret_cast<const u32x4*>(h1 + (m * RT + r) * H + h), hv);
      u32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[r][2 * i] = fmaf(gv[2 * i], hv[2 * i], acc[r][2 * i]);
        acc[r][2 * i + 1] = fmaf(gv[2 * i + 1], hv[2 * i + 1], acc[r][2 * i + 1]);
        o[i] = pack_bf16(gv[2 * i] * q[r][2 * i], gv[2 * i + 1] * q[r][2 * i + 1]);
      }
      *reinterpret_cast<u32x4*>(gs + (m * RT + r) * H + h) = o;
    }
  }
  if (slice > 0) {
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
      for (int j = 0; j < 8; ++j) part[slice - 1][r][c][j] = acc[r][j];
#pragma unroll
    for (int j = 0; j < 8; ++j) part[slice - 1][RT][c][j] = gt[j];
  }
  __syncthreads();
  if (slice == 0) {
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      float* dr = dh2 + ((size_t)b * RT + r) * Hin;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = acc[r][j];
        for (int s = 0; s < SL - 1; ++s) t += part[s][r][c][j];
        if (h + j < Hin) dr[h + j] = t;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = gt[j];
      for (int s = 0; s < SL - 1; ++s) t += part[s][RT][c][j];
      gsum[(size_t)b * H + h + j] = t;
    }
  }
}

// fp32 [batch, rows, cols] -> bf16 at dst[b*sb + r*sr + c*sc]  (dst zero-filled beforehand: the pads)
__global__ __launch_bounds__(256) void pack_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int rows,
                                                        int cols, long sb, long sr, long sc, size_t count) {
  const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= count) return;
  const int c = (int)(e % cols);
  const size_t t = e / cols;
  const int r = (int)(t % rows);
  const size_t b = t / rows;
  dst[b * sb + (size_t)r * sr + (size_t)c * sc] = (bf16)src[e];
}

// Every bf16 / padded-fp32 shadow of a step's master weights in ONE launch.  table[j] = {src (fp32 [rows, cols] dense),
// dst, rows, cols, dst row stride, dst column stride (elements), dst kind (0 bf16, 1 fp32), first flat element of the
// job}.  The jobs' elements are numbered consecutively, every job starting on a multiple of kPackBlock = 1024 (the numbers
// between a job's end and the next multiple belong to nobody): a workgroup of 256 threads x 4 elements lies inside ONE job,
// finds it with a scalar scan of the (short) table, and splits an element number into (row, column) with one 32-bit division
// per thread instead of a 64-bit one per element.  A transposed job (dst row stride 1) counts ceil(rows/32) x ceil(cols/32)
// tiles of 1024 numbers instead of its elements.  (The one-element-per-thread form, transposed jobs read straight down the
// source columns: 20.8 us for the 3.2 M elements of the mixed-precision CoR2.)
constexpr int kPackBlock = 1024;
struct PackJob {
  const float* src;
  void* dst;
  long long rows, cols, row_stride, col_stride, kind, first;
};
__global__ __launch_bounds__(256) void pack_many_kernel(const PackJob* __restrict__ table, int jobs, size_t total) {
  __shared__ float tile[32][33];
  const size_t e0 = (size_t)blockIdx.x * kPackBlock;
  int j = 0;
  while (j + 1 < jobs && (size_t)table[j + 1].first <= e0) ++j;
  j = __builtin_amdgcn_readfirstlane(j);
  const PackJob& job = table[j];
  const uint32_t rows = (uint32_t)job.rows, cols = (uint32_t)job.cols;
  const float* __restrict__ src = job.src;
  const bool to_bf16 = job.kind == 0;
  const uint32_t blk = (uint32_t)((e0 - (size_t)job.first) / kPackBlock);
  if (job.row_stride == 1 && job.col_stride != 1) {
    // a transposed shadow (dst[c][r]): the job is numbered in 32 x 32 tiles (1024 numbers each, row-major over
    // ceil(rows/32) x ceil(cols/32) tiles).  The tile is read along the source rows and written along the destination rows
    // through LDS: both sides move whole lines (read straight down a source column, a wave touched 256 lines for 1 KB)
    const uint32_t tiles_c = (cols + 31) / 32;
    const uint32_t r0 = (blk / tiles_c) * 32, c0 = (blk % tiles_c) * 32;
    if (r0 >= rows) return;
    const uint32_t a = threadIdx.x >> 3, b4 = (threadIdx.x & 7) * 4;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t r = r0 + a, c = c0 + b4 + t;
      tile[a][b4 + t] = (r < rows && c < cols) ? src[(size_t)r * cols + c] : 0.f;
    }
    __syncthreads();
    const long long cs = job.col_stride;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint32_t c = c0 + a, r = r0 + b4 + t;
      if (r < rows && c < cols) {
        const float v = tile[b4 + t][a];
        const size_t o = (size_t)((long long)c * cs + r);
        if (to_bf16) {
          static_cast<bf16*>(job.dst)[o] = (bf16)v;
        } else {
          static_cast<float*>(job.dst)[o] = v;
        }
      }
    }
    return;
  }
  const uint32_t count = rows * cols;
  const uint32_t local = blk * kPackBlock + 4u * threadIdx.x;
  if (local >= count) return;
  uint32_t r = local / cols, c = local - r * cols;
  const long long rs = job.row_stride, cs = job.col_stride;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    if (local + t < count) {
      const float v = src[local + t];
      const size_t o = (size_t)((long long)r * rs + (long long)c * cs);
      if (to_bf16) {
        static_cast<bf16*>(job.dst)[o] = (bf16)v;
      } else {
        static_cast<float*>(job.dst)[o] = v;
      }
    }
    if (++c == cols) {
      c = 0;
      ++r;
    }
  }
}

static int check_k4(const char* who, int B, int N, int L, int H, int R) {
  VQA_REQUIRE(B > 0 && N > 0 && L > 0 && H > 0 && R > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d L=%d H=%d R=%d", who, B, N, L,
              H, R);
  VQA_REQUIRE(R <= kBfMaxR, VQA_E_UNSUPPORTED, "%s: R=%d exceeds %d", who, R, kBfMaxR);
  VQA_REQUIRE(L % 64 == 0 && H % 256 == 0, VQA_E_UNSUPPORTED,
              "%s: the bf16 path needs zero-padded dims L %% 64 == 0 and H %% 256 == 0 (L=%d H=%d)", who, L, H);
  VQA_REQUIRE((long)B * N < (1L << 30) && B <= 65535, VQA_E_UNSUPPORTED, "%s: B*N too large", who);
  return VQA_OK;
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_pack_bf16(const float* src, int batch, int rows, int cols, vqa_bf16_t* dst, long dst_batch_stride,
                             long dst_row_stride, long dst_col_stride, size_t dst_elems, int zero_fill,
                             vqa_stream_t stream) {
  VQA_REQUIRE(src && dst, VQA_E_BADARG, "pack_bf16: null pointer");
  VQA_REQUIRE(batch > 0 && rows > 0 && cols > 0 && dst_batch_stride >= 0 && dst_row_stride > 0 && dst_col_stride > 0,
              VQA_E_BADARG, "pack_bf16: bad sizes batch=%d rows=%d cols=%d", batch, rows, cols);
  const size_t last = (size_t)(batch - 1) * dst_batch_stride + (size_t)(rows - 1) * dst_row_stride +
                      (size_t)(cols - 1) * dst_col_stride;
  VQA_REQUIRE(last < dst_elems, VQA_E_BADARG, "pack_bf16: destination of %zu elements is too small (needs %zu)", dst_elems,
              last + 1);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (zero_fill) {
    int rc = zero_async(dst, dst_elems * sizeof(vqa_bf16_t), s);
    if (rc != VQA_OK) return rc;
  }
  const size_t count = (size_t)batch * rows * cols;
  VQA_LAUNCH(pack_bf16_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, src,
                     reinterpret_cast<bf16*>(dst), rows, cols, dst_batch_stride, dst_row_stride, dst_col_stride, count);
  return check_launch("pack_bf16");
}

extern "C" int vqa_pack_many(const void* table, int jobs, size_t total, vqa_stream_t stream) {
  VQA_REQUIRE(table != nullptr && jobs > 0 && total > 0, VQA_E_BADARG, "pack_many: empty job table");
  VQA_REQUIRE(aligned(table, 8), VQA_E_UNSUPPORTED, "pack_many: the table must be 8-byte aligned");
  VQA_LAUNCH(pack_many_kernel, dim3((unsigned)((total + kPackBlock - 1) / kPackBlock)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const PackJob*>(table), jobs, total);
  return check_launch("pack_many");
}

static int check_nt(const char* who, const void* a, int lda, const void* b, int ldb, const void* c, int ldc, int M, int N, int K,
                    int act) {
  VQA_REQUIRE(a && b && c, VQA_E_BADARG, "%s: null pointer", who);
  VQA_REQUIRE(M > 0 && N > 0 && K > 0, VQA_E_BADARG, "%s: bad sizes M=%d N=%d K=%d", who, M, N, K);
  VQA_REQUIRE(act == 0 || act == 1, VQA_E_BADARG, "%s: act must be 0 (none) or 1 (relu), got %d", who, act);
  VQA_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K && ldc >= N && aligned(a, 16) &&
                  aligned(b, 16),
              VQA_E_UNSUPPORTED, "%s: needs K %% 64 == 0, lda/ldb %% 8 == 0 and 16-byte aligned a/b (K=%d lda=%d ldb=%d)", who,
              K, lda, ldb);
  return VQA_OK;
}

extern "C" int vqa_gemm_bf16_nt(const vqa_bf16_t* a, int lda, const vqa_bf16_t* b, int ldb, const float* bias,
                                vqa_bf16_t* c, int ldc, int M, int N, int K, int act, vqa_stream_t stream) {
  const int rc = check_nt("gemm_bf16_nt", a, lda, b, ldb, c, ldc, M, N, K, act);
  if (rc != VQA_OK) return rc;
  return launch_nt("gemm_bf16_nt", reinterpret_cast<const bf16*>(a), lda, reinterpret_cast<const bf16*>(b), ldb, bias,
                   reinterpret_cast<bf16*>(c), ldc, M, N, K, act, static_cast<hipStream_t>(stream));
}

extern "C" int vqa_gemm_bf16_nt_ex(const vqa_bf16_t* a, int lda, const vqa_bf16_t* b, int ldb, const float* bias,
                                   vqa_bf16_t* c, int ldc, int M, int N, int K, int act, const vqa_bf16_t* gate, int ldg,
                                   float p_drop, uint64_t seed, const uint64_t* seed_ptr, vqa_stream_t stream) {
  const int rc = check_nt("gemm_bf16_nt_ex", a, lda, b, ldb, c, ldc, M, N, K, act);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(gate == nullptr || ldg >= N, VQA_E_BADARG, "gemm_bf16_nt_ex: gate row stride %d < N=%d", ldg, N);
  VQA_REQUIRE(p_drop == 0.f || p_drop == 0.5f, VQA_E_UNSUPPORTED,
              "gemm_bf16_nt_ex: the in-kernel input dropout exists in its one-bit form only (p_drop 0 or 0.5, got %f)", (double)p_drop);
  VQA_REQUIRE(p_drop == 0.f || ((long)M * lda < (1L << 32) && lda == K), VQA_E_UNSUPPORTED,
              "gemm_bf16_nt_ex: dropout needs a dense a (lda == K) of fewer than 2^32 elements");
  NtExtra ex;
  ex.gate = reinterpret_cast<const bf16*>(gate);
  ex.ldg = ldg;
  if (p_drop > 0.f) {
    ex.drop = true;
    ex.dc = make_drop(p_drop, seed, seed_ptr);
    ex.scale = ex.dc.scale;
    ex.mask_ld = (uint32_t)lda;
  }
  return launch_nt("gemm_bf16_nt_ex", reinterpret_cast<const bf16*>(a), lda, reinterpret_cast<const bf16*>(b), ldb, bias,
                   reinterpret_cast<bf16*>(c), ldc, M, N, K, act, static_cast<hipStream_t>(stream), ex);
}

extern "C" size_t vqa_gemm_bf16_tn_workspace_bytes(int K, int N1, int N2) {
  if (K <= 0 || N1 <= 0 || N2 <= 0) return 0;
  return tn_workspace_bytes(K, N1, N2);
}

static int check_tn(const char* who, const void* a, int lda, const void* b, int ldb, const void* workspace,
                    size_t workspace_bytes, int K, int N1, int N2) {
  VQA_REQUIRE(a && b, VQA_E_BADARG, "%s: null pointer", who);
  VQA_REQUIRE(K > 0 && N1 > 0 && N2 > 0, VQA_E_BADARG, "%s: bad sizes K=%d N1=%d N2=%d", who, K, N1, N2);
  VQA_REQUIRE(N1 % 8 == 0 && N2 % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && lda >= N1 && ldb >= N2 && aligned(a, 16) &&
                  aligned(b, 16),
              VQA_E_UNSUPPORTED, "%s: needs N1, N2, lda, ldb %% 8 == 0 and 16-byte aligned pointers", who);
  const size_t need = tn_workspace_bytes(K, N1, N2);
  VQA_REQUIRE(workspace_bytes >= need && workspace && aligned(workspace, 16), VQA_E_BADARG,
              "%s: workspace of %zu B is too small (needs %zu)", who, workspace_bytes, need);
  return VQA_OK;
}

extern "C" int vqa_gemm_bf16_tn(const vqa_bf16_t* a, int lda, const vqa_bf16_t* b, int ldb, float* c, void* workspace,
                                size_t workspace_bytes, int K, int N1, int N2, vqa_stream_t stream) {
  VQA_REQUIRE(c != nullptr, VQA_E_BADARG, "gemm_bf16_tn: null pointer");
  const int rc = check_tn("gemm_bf16_tn", a, lda, b, ldb, workspace, workspace_bytes, K, N1, N2);
  if (rc != VQA_OK) return rc;
  BfOuts outs{};
  outs.p[0] = c;
  return launch_tn("gemm_bf16_tn", reinterpret_cast<const bf16*>(a), lda, reinterpret_cast<const bf16*>(b), ldb, outs, 1, N1,
                   N1, N2, N2, static_cast<float*>(workspace), K, N1, N2, static_cast<hipStream_t>(stream));
}

extern "C" int vqa_gemm_bf16_tn_ex(const vqa_bf16_t* a, int lda, const vqa_bf16_t* b, int ldb, float* const* outs, int groups,
                                   int rows_per_group, int out_rows, int out_cols, int out_ld, void* workspace,
                                   size_t workspace_bytes, int K, int N1, int N2, float p_drop, uint64_t seed,
                                   const uint64_t* seed_ptr, vqa_stream_t stream) {
  VQA_REQUIRE(outs != nullptr && groups >= 1 && groups <= kBfMaxR, VQA_E_BADARG, "gemm_bf16_tn_ex: 1..%d output groups", kBfMaxR);
  const int rc = check_tn("gemm_bf16_tn_ex", a, lda, b, ldb, workspace, workspace_bytes, K, N1, N2);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(rows_per_group > 0 && groups * rows_per_group <= N1 && out_rows > 0 && out_rows <= rows_per_group &&
                  out_cols > 0 && out_cols <= N2 && out_ld >= out_cols,
              VQA_E_BADARG, "gemm_bf16_tn_ex: output window %d x %d (stride %d) of %d groups x %d rows does not fit [%d,%d]",
              out_rows, out_cols, out_ld, groups, rows_per_group, N1, N2);
  VQA_REQUIRE(p_drop == 0.f || p_drop == 0.5f, VQA_E_UNSUPPORTED,
              "gemm_bf16_tn_ex: the in-kernel dropout of b exists in its one-bit form only (p_drop 0 or 0.5, got %f)", (double)p_drop);
  VQA_REQUIRE(p_drop == 0.f || ((long)K * ldb < (1L << 32) && ldb % 32 == 0), VQA_E_UNSUPPORTED,
              "gemm_bf16_tn_ex: dropout needs ldb %% 32 == 0 and fewer than 2^32 elements");
  BfOuts o{};
  for (int g = 0; g < groups; ++g) {
    VQA_REQUIRE(outs[g] != nullptr, VQA_E_BADARG, "gemm_bf16_tn_ex: null output %d", g);
    o.p[g] = outs[g];
  }
  TnExtra ex;
  if (p_drop > 0.f) {
    ex.drop = true;
    ex.dc = make_drop(p_drop, seed, seed_ptr);
    ex.scale = ex.dc.scale;
    ex.mask_ld = (uint32_t)ldb;
  }
  // rows past groups * rows_per_group (if any) belong to no group: the reduction skips them (r >= out_rows for g >= groups
  // cannot happen because N1 / gp < kBfMaxR is not guaranteed -- so clamp by requiring an exact cover)
  VQA_REQUIRE(groups * rows_per_group == N1, VQA_E_BADARG, "gemm_bf16_tn_ex: groups * rows_per_group must equal N1");
  return launch_tn("gemm_bf16_tn_ex", reinterpret_cast<const bf16*>(a), lda, reinterpret_cast<const bf16*>(b), ldb, o, groups,
                   rows_per_group, out_rows, out_cols, out_ld, static_cast<float*>(workspace), K, N1, N2,
                   static_cast<hipStream_t>(stream), ex);
}

extern "C" int vqa_lowrank_bilinear_fusion_fwd_bf16(const vqa_bf16_t* x, const vqa_bf16_t* w1, const float* b1,
                                                    const float* h2, vqa_bf16_t* out, vqa_bf16_t* h1, int B, int N, int L,
                                                    int H, int R, int H_in, vqa_stream_t stream) {
  VQA_REQUIRE(x && w1 && b1 && h2 && out, VQA_E_BADARG, "lowrank_bilinear_fusion_fwd_bf16: null pointer");
  int rc = check_k4("lowrank_bilinear_fusion_fwd_bf16", B, N, L, H, R);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(H_in > 0 && H_in <= H, VQA_E_BADARG, "lowrank_bilinear_fusion_fwd_bf16: h2 width %d exceeds the padded H = %d", H_in, H);
  VQA_REQUIRE(aligned(x, 16) && aligned(w1, 16), VQA_E_UNSUPPORTED, "lowrank_bilinear_fusion_fwd_bf16: x/w1 must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int M = B * N;
  if (R == 2 && N >= 2 && M > 64) {
    const size_t lds = BfTile<128, 128>::kSmemBytes + 128 * sizeof(int);
    VQA_ENSURE_LDS(bilinear_fwd2_bf16_kernel, lds);
    const int tm_ = (M + 127) / 128, tn_ = H / 64;
    VQA_LAUNCH(bilinear_fwd2_bf16_kernel, dim3(tm_ * tn_), dim3(kBfThreads), lds, s, reinterpret_cast<const bf16*>(x),
                       reinterpret_cast<const bf16*>(w1), b1, h2, reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(h1), M, N,
                       L, H, tn_, H_in);
    return check_launch("lowrank_bilinear_fusion_fwd_bf16");
  }
  BfTileChoice t = choose_bf_tile(M, H, L);
  if (t.bm == 128 && t.bn == 128) t.bn = 64;  // total + per-rank accumulators: 128x128 would drop to one wave per SIMD
  const int tiles_m = (M + t.bm - 1) / t.bm, tiles_n = H / t.bn;
#define LAUNCH(BM_, BN_)                                                                                                  \
  {                                                                                                                       \
    const size_t lds = BfTile<BM_, BN_>::kSmemBytes + BM_ * sizeof(int) + BfTileStore<BM_, BN_>::kBytes;                  \
    VQA_ENSURE_LDS((bilinear_fwd_bf16_kernel<BM_, BN_>), lds);                                                            \
    VQA_LAUNCH((bilinear_fwd_bf16_kernel<BM_, BN_>), dim3(tiles_m * tiles_n), dim3(kBfThreads), lds, s,           \
                       reinterpret_cast<const bf16*>(x), reinterpret_cast<const bf16*>(w1), b1, h2,                       \
                       reinterpret_cast<bf16*>(out), reinterpret_cast<bf16*>(h1), M, N, L, H, R, tiles_n, H_in);         \
  }
  VQA_BF_TILE_SWITCH(t, LAUNCH);
#undef LAUNCH
  return check_launch("lowrank_bilinear_fusion_fwd_bf16");
}

// workspace layout: gs bf16 [M, R*H] | gsum f32 [B, H] | TN slabs
static size_t k4_gs_bytes(int B, int N, int H, int R) { return ((size_t)B * N * R * H * sizeof(vqa_bf16_t) + 255) / 256 * 256; }
static size_t k4_gsum_bytes(int B, int H) { return ((size_t)B * H * sizeof(float) + 255) / 256 * 256; }

extern "C" size_t vqa_lowrank_bilinear_fusion_bwd_bf16_workspace_bytes(int B, int N, int L, int H, int R) {
  if (B <= 0 || N <= 0 || L <= 0 || H <= 0 || R <= 0 || R > kBfMaxR) return 0;
  return k4_gs_bytes(B, N, H, R) + k4_gsum_bytes(B, H) + tn_workspace_bytes(B * N, R * H, L);
}

extern "C" int vqa_lowrank_bilinear_fusion_bwd_bf16(const vqa_bf16_t* x, const vqa_bf16_t* w1t, const float* h2,
                                                    const vqa_bf16_t* h1, const vqa_bf16_t* g, vqa_bf16_t* d_x,
                                                    float* const* d_w1, float* const* d_b1, float* d_h2, void* workspace,
                                                    size_t workspace_bytes, int B, int N, int L, int H, int R, int H_out,
                                                    int L_out, int gate_dx, int phases, vqa_stream_t stream) {
  VQA_REQUIRE(x && h2 && h1 && g && d_w1 && d_b1 && d_h2 && workspace, VQA_E_BADARG,
              "lowrank_bilinear_fusion_bwd_bf16: null pointer");
  VQA_REQUIRE(phases >= 1 && phases <= 3, VQA_E_BADARG, "lowrank_bilinear_fusion_bwd_bf16: phases must be 1, 2 or 3, got %d", phases);
  VQA_REQUIRE(d_x == nullptr || w1t != nullptr, VQA_E_BADARG, "lowrank_bilinear_fusion_bwd_bf16: d_x needs w1t");
  int rc = check_k4("lowrank_bilinear_fusion_bwd_bf16", B, N, L, H, R);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(H_out > 0 && H_out <= H && L_out > 0 && L_out <= L, VQA_E_BADARG,
              "lowrank_bilinear_fusion_bwd_bf16: master shape [%d,%d] exceeds the padded one [%d,%d]", H_out, L_out, H, L);
  VQA_REQUIRE(workspace_bytes >= vqa_lowrank_bilinear_fusion_bwd_bf16_workspace_bytes(B, N, L, H, R), VQA_E_BADARG,
              "lowrank_bilinear_fusion_bwd_bf16: workspace of %zu B is too small", workspace_bytes);
  VQA_REQUIRE(aligned(x, 16) && aligned(g, 16) && aligned(h1, 16) && aligned(workspace, 256) && (d_x == nullptr || (aligned(d_x, 16) && aligned(w1t, 16))),
              VQA_E_UNSUPPORTED, "lowrank_bilinear_fusion_bwd_bf16: tensors must be 16-byte aligned (workspace 256)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int M = B * N, RH = R * H;
  BfOuts dw{}, db{};
  for (int r = 0; r < R; ++r) {
    VQA_REQUIRE(d_w1[r] != nullptr && d_b1[r] != nullptr, VQA_E_BADARG, "lowrank_bilinear_fusion_bwd_bf16: null gradient %d", r);
    dw.p[r] = d_w1[r];
    db.p[r] = d_b1[r];
  }
  char* ws = static_cast<char*>(workspace);
  bf16* gs = reinterpret_cast<bf16*>(ws);
  float* gsum = reinterpret_cast<float*>(ws + k4_gs_bytes(B, N, H, R));
  float* slabs = reinterpret_cast<float*>(ws + k4_gs_bytes(B, N, H, R) + k4_gsum_bytes(B, H));
  const bf16* gb = reinterpret_cast<const bf16*>(g);
  const bf16* h1b = reinterpret_cast<const bf16*>(h1);
#define PREP8(RT_)                                                                                                        \
  VQA_LAUNCH(bilinear_bwd_prep8_bf16_kernel<RT_>, dim3(H / 128, B), dim3(256), 0, s, gb, h1b, h2, gs, d_h2, gsum, \
                     N, H, H_out)
  if (phases & 1) switch (R) {
    case 1: PREP8(1); break;
    case 2: PREP8(2); break;
    case 3: PREP8(3); break;
    case 4: PREP8(4); break;
    case 5: PREP8(5); break;
    default:
      if ((long)B * H < 4 * 65536) {
        VQA_LAUNCH(bilinear_bwd_prep_bf16_kernel<16>, dim3(H / 64, B), dim3(256), 0, s, gb, h1b, h2, gs, d_h2, gsum, N, H,
                           R, H_out);
      } else {
        VQA_LAUNCH(bilinear_bwd_prep_bf16_kernel<4>, dim3(H / 256, B), dim3(256), 0, s, gb, h1b, h2, gs, d_h2, gsum, N, H,
                           R, H_out);
      }
  }
#undef PREP8
  if ((phases & 1) && d_x != nullptr) {
    NtExtra ex;
    if (gate_dx) {   // x is the relu output of the layer in front: its gradient gate rides in this store
      ex.gate = reinterpret_cast<const bf16*>(x);
      ex.ldg = L;
    }
    rc = launch_nt("lowrank_bilinear_fusion_bwd_bf16(dx)", gs, RH, reinterpret_cast<const bf16*>(w1t), RH, nullptr,
                   reinterpret_cast<bf16*>(d_x), L, M, L, RH, 0, s, ex);
    if (rc != VQA_OK) return rc;
  }
  if (!(phases & 2)) return check_launch("lowrank_bilinear_fusion_bwd_bf16");
  BfDbJob dbjob;        // the bias gradients ride in the weight gradient's slab-reduction launch
  dbjob.h2 = h2;
  dbjob.gsum = gsum;
  dbjob.db1 = db;
  dbjob.B = B;
  dbjob.H = H;
  dbjob.R = R;
  dbjob.Hout = H_out;
  dbjob.blocks = RH / 64;
  return launch_tn("lowrank_bilinear_fusion_bwd_bf16(dw)", gs, RH, reinterpret_cast<const bf16*>(x), L, dw, R, H, H_out, L_out,
                   L_out, slabs, M, RH, L, s, TnExtra(), dbjob);
}
