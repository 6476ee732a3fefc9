// The question encoder's products on the split engine (gemm_f32_split.hpp): SkipThoughts' BayesianGRU
// (putils/__init__.py:604-746) multiplies, per time step t of 26, the masked hidden state by the three recurrent weights
//     a_g = (h_{t-1} * m_g) W_hg^T,   g in {r, i, n}      ([B,2400] x [2400,2400]^T, three times)
// forward and gz_g W_hg backward, and projects all T steps' inputs once (x_t * m'_g) W_ig^T + b_ig ([B*T,620] x [2400,620]^T).
// Rounds 1-5 ran these as batched library GEMMs (torch.bmm -> hipBLASLt, 118 TFLOP/s at the step's shape; VERDICT r05 missing
// #1); here they are fp32 products on the bf16 matrix pipe like K5's: G same-shaped problems per launch (blockIdx.y), the
// activations split in registers (nt_accumulate), the weights split ONCE per training step into packed plane images
// (gru_pack_kernel: as they are for the forward, transposed for the data gradients) and reused by all 26 time steps.
//   C_g[m][n] = sum_{k < Kr} A_g[m][k] * B_g[n][k] (+ bias_g[n])        A_g [M,Kr] fp32 rows of stride lda, C_g [M,N] rows of ldc
// The contraction is padded to a multiple of 64 (2400 -> 2432, 620 -> 640): the image holds zero planes there, the A loads run
// into the next row (finite garbage times zero) or past the operand's extent (the buffer range check returns zeros); a non-finite
// neighbour makes the accumulator non-finite and the output is recomputed over the real Kr by the repair path, as everywhere
// on this engine (gemm_f32_split.hpp, any_nonfinite).
// Tile: 16 RB x 160 per workgroup, the contraction split over two pairs of waves.  RB = 7 for the recurrent step (M = 512 rows:
// 5 x 15 x 3 = 225 workgroups in one wave of the 256 CUs; RB = 8 gives 180 longer ones, RB = 6 270: two waves), RB = 9 for the
// tall input projections.
#include "common.hpp"
#include "gemm_f32_split.hpp"

namespace vqa {

// B[n][k] = w[n * ldw + k] (TRANS: w[k * ldw + n]) for n < N, k < Kr, else 0 -> Bp[n / 16][chunk][plane][lane][8 bf16], Kp / 32 chunks
template <bool TRANS>
__global__ __launch_bounds__(256) void gru_pack_kernel(const float* __restrict__ w, long w_gs, int ldw, int N, int Kr, int Kp,
                                                       sp::u32x4* __restrict__ out, size_t out_gs) {
  const int chunks = Kp / sp::kChunk, nblocks = (N + 15) / 16;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)nblocks * chunks * 64) return;
  const int lane = (int)(t & 63), r = lane & 15, g = lane >> 4;
  const long bc = t >> 6;
  const int c = (int)(bc % chunks), n = (int)(bc / chunks) * 16 + r;
  const float* src = w + (size_t)blockIdx.y * w_gs;
  sp::f32x4 lo = sp::f32x4{0.f, 0.f, 0.f, 0.f}, hi = lo;
  if (n < N) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = c * sp::kChunk + 4 * g + j, k1 = k0 + 16;
      if (k0 < Kr) lo[j] = TRANS ? src[(size_t)k0 * ldw + n] : src[(size_t)n * ldw + k0];
      if (k1 < Kr) hi[j] = TRANS ? src[(size_t)k1 * ldw + n] : src[(size_t)n * ldw + k1];
    }
  }
  sp::Planes pl;
  sp::split8<false>(lo, hi, pl);
  sp::u32x4* dst = out + (size_t)blockIdx.y * out_gs + (size_t)bc * 192 + lane;
  dst[0] = pl.p[0];
  dst[64] = pl.p[1];
  dst[128] = pl.p[2];
}

struct BatchNtArgs {
  const float* A;
  long a_gs;            // elements between the A operands of consecutive problems (0: one shared A)
  int lda;
  const sp::u32x4* Bp;
  size_t bp_gs;         // 16-byte units between the packed images
  float* C;
  long c_gs;
  int ldc;
  const float* bias;    // [G][N] rows of stride bias_gs, or null
  int bias_gs;
  int M, N, Kp, Kr;
  int tiles_n, tiles_m;
  int col_major;        // tile order inside a problem: 1 = the row tiles of one column tile are neighbours (short M: they share the
                        // column tile's slice of the weight image in their XCD's L2; the activations are small), 0 = row-major
  const float* Bf;      // the fp32 weights (repair path): B_g[n][k] = Bf[g * bf_gs + n * bf_sn + k * bf_sk]
  long bf_gs;
  int bf_sn, bf_sk;
};

template <int RB, int CB, int NR>
__global__ __launch_bounds__(sp::kThreads, 1) void gemm_nt_batched_kernel(BatchNtArgs q) {
  using S = rt::NtShape<RB, CB, 1, 2, 2>;
  using sp::f32x4;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wk = wave / 2, wn = wave % 2;
  // one linear workgroup id over all problems (the hardware deals ids round-robin to the 8 XCDs; xcd_remap hands every XCD a
  // contiguous range of tiles)
  const int tiles = q.tiles_m * q.tiles_n;
  const int lin = xcd_remap(blockIdx.x, gridDim.x);
  // (readfirstlane: the division runs on the vector ALU, and without it hipcc kept everything derived from `prob` -- the operands'
  //  buffer descriptors -- in VGPRs and wrapped each of the main loop's buffer loads in a waterfall loop: 8-20 v_readfirstlane per
  //  row block.  Check: count v_readfirstlane between the MFMAs of the ISA)
  const int prob = __builtin_amdgcn_readfirstlane(lin / tiles), tile = lin - prob * tiles;
  const float* A = q.A + (size_t)prob * q.a_gs;
  float* C = q.C + (size_t)prob * q.c_gs;
  const sp::NtArgs p{A, q.Bp + (size_t)prob * q.bp_gs, q.lda, q.M, q.N, q.Kp, q.tiles_n};
  const int tm = __builtin_amdgcn_readfirstlane(q.col_major ? tile % q.tiles_m : tile / q.tiles_n);
  const int tn = __builtin_amdgcn_readfirstlane(q.col_major ? tile / q.tiles_m : tile % q.tiles_n);
  const int m0 = tm * S::BM, n0 = tn * S::BN + wn * (16 * CB);
  const int chunks = q.Kp / sp::kChunk;

  f32x4 acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int per = ((chunks + 1) / 2 + 1) & ~1;   // per pair of waves: an even number of chunks
  const int c_lo = min(chunks, wk * per), c_hi = min(chunks, c_lo + per);
  const DropCfg dc{};
  // (the extent the A loads may touch ends with the last REAL element: the padded contraction's reads past it return zeros)
  sp::nt_accumulate<RB, CB, false, 0, NR>(p, dc, ((size_t)(q.M - 1) * q.lda + q.Kr) * 4, m0, n0, c_lo, c_hi, acc);

  const float* bias = q.bias != nullptr ? q.bias + (size_t)prob * q.bias_gs : nullptr;
  const float* Bf = q.Bf != nullptr ? q.Bf + (size_t)prob * q.bf_gs : nullptr;
  auto finish = [&](int blk, f32x4 v) {
    const int i = blk / CB, j = blk % CB;
    const int col = n0 + 16 * j + r;
    if (col >= q.N) return;
    const bool bad = Bf != nullptr && sp::any_nonfinite(v);
    const float bv = bias != nullptr ? bias[col] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = m0 + 16 * i + 4 * g + t;
      if (row >= q.M) continue;
      float out = v[t];
      if (bad) {       // the repair path: an fp32 dot product of the original operands over the real contraction
        const float* a = A + (size_t)row * q.lda;
        const float* b = Bf + (size_t)col * q.bf_sn;
        float s = 0.f;
        for (int k = 0; k < q.Kr; ++k) s = fmaf(a[k], b[(size_t)k * q.bf_sk], s);
        out = s;
      }
      C[(size_t)row * q.ldc + col] = out + bv;
    }
  };
  extern __shared__ __attribute__((aligned(16))) char gg_smem[];
  f32x4* red = reinterpret_cast<f32x4*>(gg_smem);
  f32x4* out_box = red + (size_t)((wn * 2 + wk) * S::HALF) * 64 + lane;
  const f32x4* in_box = red + (size_t)((wn * 2 + (wk ^ 1)) * S::HALF) * 64 + lane;
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
}

// ---- TN form: d_w[n1][n2] = sum_m g[m][n1] * x[m][n2] over tall operands (the encoder's weight gradients: M = T * B rows) ----------
// gemm_f32_split.hpp's weight-gradient pair: g is split and packed once (pack_tn_kernel, in column groups of <= 30 blocks), x is
// split inside the GEMM and shared by the workgroup's four waves (gemm_tn_shared_kernel), row slabs are summed in fixed order.
struct TnSplitPlan {
  int slabs, cps, nblocks, grp, groups, tiles1, tiles2;
};
static TnSplitPlan tn_split_plan(int M, int N1, int N2) {
  TnSplitPlan pl{};
  pl.nblocks = (N1 + 15) / 16;
  pl.grp = pl.nblocks < sp::kPackMaxBlocks ? pl.nblocks : sp::kPackMaxBlocks;
  pl.groups = (pl.nblocks + pl.grp - 1) / pl.grp;
  pl.tiles1 = (pl.nblocks + 19) / 20;
  pl.tiles2 = (N2 + 127) / 128;
  // row slabs: the count whose workgroups fill whole rounds of the 256 CUs best (at least 8 chunks of 32 rows each)
  const int chunks = (M + sp::kChunk - 1) / sp::kChunk, tiles = pl.tiles1 * pl.tiles2;
  int want = 1;
  double best = -1.0;
  // the fewest slabs within 3 % of the best fill of the 256 CUs, a slab no shorter than 32 chunks (a workgroup's prologue, its
  // 40-block store and the slab sum are paid per slab: 19 slabs of 22 chunks ran [2400 x 620] at 115 TFLOP/s)
  for (int s = 1; s <= 64 && (s == 1 || s * 32 <= chunks); ++s) {
    const long wgs = (long)tiles * s;
    const double eff = (double)wgs / (double)(((wgs + 255) / 256) * 256);
    if (eff > best + 0.03) best = eff, want = s;
  }
  const sp::TnPlan tp = sp::tn_plan(M, want);
  pl.slabs = tp.slabs;
  pl.cps = tp.cps;
  return pl;
}
static size_t r256(size_t b) { return (b + 255) & ~(size_t)255; }

static int padded_k(int K) { return (K + 63) / 64 * 64; }
static size_t image_bytes(int N, int K) { return ((sp::packed_bytes(N, padded_k(K)) + 255) & ~(size_t)255); }

template <int RB, int NR>
static int launch_batched(const BatchNtArgs& a, int G, hipStream_t s) {
  using S = rt::NtShape<RB, 5, 1, 2, 2>;
  BatchNtArgs q = a;
  q.tiles_m = (a.M + S::BM - 1) / S::BM;
  q.tiles_n = (a.N + S::BN - 1) / S::BN;
  q.col_major = vqa::option("VQA_GRU_GEMM_ORDER") ? (vqa::option_is("VQA_GRU_GEMM_ORDER", 'c') ? 1 : 0) : (q.tiles_m <= 8 ? 1 : 0);
  VQA_ENSURE_LDS((gemm_nt_batched_kernel<RB, 5, NR>), S::kLdsBytes);
  VQA_LAUNCH((gemm_nt_batched_kernel<RB, 5, NR>), dim3(q.tiles_m * q.tiles_n * G), dim3(sp::kThreads), S::kLdsBytes, s, q);
  return check_launch("gemm_nt_split_batched");
}

}  // namespace vqa

using namespace vqa;

extern "C" size_t vqa_split_weights_bytes(int G, int N, int K) { return (size_t)G * image_bytes(N, K); }

extern "C" int vqa_split_weights_pack(const float* w, long w_gs, int ldw, int transposed, void* image, size_t image_size, int G, int N,
                                      int K, vqa_stream_t stream) {
  VQA_REQUIRE(w && image, VQA_E_BADARG, "split_weights_pack: null pointer");
  VQA_REQUIRE(G >= 1 && N >= 1 && K >= 1 && ldw >= 1, VQA_E_BADARG, "split_weights_pack: sizes must be positive");
  VQA_REQUIRE(aligned(image, 16) && image_size >= vqa_split_weights_bytes(G, N, K), VQA_E_BADARG,
              "split_weights_pack: image must be 16-byte aligned and hold vqa_split_weights_bytes(G, N, K) bytes");
  const int Kp = padded_k(K);
  const long threads = (long)((N + 15) / 16) * (Kp / sp::kChunk) * 64;
  const dim3 grid((unsigned)((threads + 255) / 256), G);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (transposed)
    VQA_LAUNCH((gru_pack_kernel<true>), grid, dim3(256), 0, s, w, w_gs, ldw, N, K, Kp, static_cast<sp::u32x4*>(image), image_bytes(N, K) / 16);
  else
    VQA_LAUNCH((gru_pack_kernel<false>), grid, dim3(256), 0, s, w, w_gs, ldw, N, K, Kp, static_cast<sp::u32x4*>(image), image_bytes(N, K) / 16);
  return check_launch("split_weights_pack");
}

extern "C" int vqa_gemm_nt_split_batched_supported(int M, int N, int K, int lda, int ldc) {
  return (M >= 64 && N >= 16 && K >= 64 && lda % 4 == 0 && lda >= K && ldc >= N && (size_t)M * lda * 4 < (1ull << 32) &&
          sp::packed_bytes(N, padded_k(K)) < (1ull << 32)) ? 1 : 0;
}

extern "C" int vqa_gemm_nt_split_batched(const float* a, long a_gs, int lda, const void* image, float* c, long c_gs, int ldc,
                                         const float* bias, int bias_gs, const float* w, long w_gs, int w_sn, int w_sk, int G, int M,
                                         int N, int K, vqa_stream_t stream) {
  VQA_REQUIRE(a && image && c, VQA_E_BADARG, "gemm_nt_split_batched: null pointer");
  VQA_REQUIRE(G >= 1 && G <= 4096, VQA_E_BADARG, "gemm_nt_split_batched: G out of range");
  VQA_REQUIRE(vqa_gemm_nt_split_batched_supported(M, N, K, lda, ldc) == 1, VQA_E_UNSUPPORTED,
              "gemm_nt_split_batched: shape outside the engine (M=%d N=%d K=%d lda=%d ldc=%d): M >= 64, lda %% 4 == 0", M, N, K, lda, ldc);
  VQA_REQUIRE(aligned(a, 16) && a_gs % 4 == 0 && aligned(image, 16), VQA_E_UNSUPPORTED,
              "gemm_nt_split_batched: a (and its batch stride) and the image must be 16-byte aligned");
  BatchNtArgs q{a, a_gs, lda, static_cast<const sp::u32x4*>(image), image_bytes(N, K) / 16, c, c_gs, ldc, bias, bias_gs, M, N, padded_k(K), K,
                0, 0, 0, w, w_gs, w_sn, w_sk};
  hipStream_t s = static_cast<hipStream_t>(stream);
  // row blocks per workgroup: what leaves the fewest idle CUs in the last wave of workgroups (ties: the taller tile)
  int rb = 9;
  if (const char* e = vqa::option("VQA_GRU_GEMM_RB")) rb = std::atoi(e);
  else if (M <= 1024) {
    double best = 1e30;
    for (int cand : {9, 8, 7}) {
      const long wgs = (long)((M + 16 * cand - 1) / (16 * cand)) * ((N + 159) / 160) * G;
      const double cost = (double)((wgs + 255) / 256) * cand;      // waves of workgroups x time per workgroup
      if (cost < best - 1e-9) best = cost, rb = cand;
    }
  }
  if (rb == 7) return launch_batched<7, 7>(q, G, s);
  if (rb == 8) return launch_batched<8, 4>(q, G, s);
  return launch_batched<9, 3>(q, G, s);
}

extern "C" int vqa_gemm_tn_split_supported(int M, int N1, int N2, int ldg, int ldx) {
  return (M >= 1152 && N1 >= 16 && N1 % 2 == 0 && N2 >= 64 && N2 % 4 == 0 && ldg % 2 == 0 && ldg >= N1 && ldx % 4 == 0 && ldx >= N2 &&
          (size_t)M * ldx * 4 < (1ull << 32) && (size_t)M * ldg * 4 < (1ull << 32)) ? 1 : 0;
}

extern "C" size_t vqa_gemm_tn_split_workspace_bytes(int M, int N1, int N2) {
  const TnSplitPlan pl = tn_split_plan(M, N1, N2);
  return r256(sp::packed_tn_bytes(pl.slabs, pl.cps, N1)) + r256((size_t)pl.slabs * N1 * N2 * 4);
}

extern "C" int vqa_gemm_tn_split(const float* g, int ldg, const float* x, int ldx, float* d_w, void* workspace, size_t workspace_bytes,
                                 int M, int N1, int N2, vqa_stream_t stream) {
  VQA_REQUIRE(g && x && d_w && workspace, VQA_E_BADARG, "gemm_tn_split: null pointer");
  VQA_REQUIRE(vqa_gemm_tn_split_supported(M, N1, N2, ldg, ldx) == 1, VQA_E_UNSUPPORTED,
              "gemm_tn_split: shape outside the engine (M=%d N1=%d N2=%d ldg=%d ldx=%d)", M, N1, N2, ldg, ldx);
  VQA_REQUIRE(aligned(g, 8) && aligned(x, 16) && aligned(d_w, 16) && aligned(workspace, 16), VQA_E_UNSUPPORTED,
              "gemm_tn_split: x, d_w, workspace must be 16-byte aligned, g 8-byte");
  VQA_REQUIRE(workspace_bytes >= vqa_gemm_tn_split_workspace_bytes(M, N1, N2), VQA_E_BADARG, "gemm_tn_split: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const TnSplitPlan pl = tn_split_plan(M, N1, N2);
  char* base = static_cast<char*>(workspace);
  sp::u32x4* gp = reinterpret_cast<sp::u32x4*>(base);
  float* slab = reinterpret_cast<float*>(base + r256(sp::packed_tn_bytes(pl.slabs, pl.cps, N1)));
  VQA_LAUNCH((sp::pack_tn_kernel<false, false>), dim3(pl.slabs * sp::kPackParts, pl.groups), dim3(256), sp::pack_tn_lds_bytes(pl.grp), s, g,
             (const float*)nullptr, ldg, M, N1, pl.grp, pl.cps, gp, (float*)nullptr, (float*)nullptr, pl.nblocks);
  const sp::TnArgs a{gp, x, slab, ldx, M, N1, N2, pl.nblocks, pl.cps, pl.tiles1, pl.tiles2, g, nullptr, ldg};
  const DropCfg dc{};
  VQA_LAUNCH((sp::gemm_tn_shared_kernel<5, false>), dim3(pl.tiles1 * pl.tiles2 * pl.slabs), dim3(sp::kThreads), sp::kTnSharedLds, s, a, dc);
  const int NK = N1 * N2;
  VQA_LAUNCH((sp::slab_sum_kernel), dim3(sp::slab_sum_blocks(NK, N1)), dim3(256), 0, s, slab, (const float*)nullptr, d_w, (float*)nullptr, NK,
             N1, pl.slabs, 0, 1.f);
  return check_launch("gemm_tn_split");
}
