// K6: the [B, .]-sized layers of the heads -- the four question projections, the two sigmoid gates, Mutan's question-side
// rank factors, the per-glimpse projections, fusion_final and the classifier (MyLinear / putils.Linear: config/CoR2.py:
// 94-122,133-134,170,180-189; putils/__init__.py:16-33,232-238) -- forward and backward, as GROUPED launches:
//
//   vqa_grouped_gemm      ONE launch runs every GEMM of a phase of the step (up to kGMaxProbs problems of any of the three
//                         forms NT / NN / TN -- forward, data gradient, weight gradient -- mixed freely) on the fp32 MFMA tile
//                         engine (gemm_f32_mfma.hpp: 64x64 tiles, v_mfma_f32_32x32x2_f32, LDS-staged operands), each
//                         problem split over its contraction so that the union of all tiles fills the chip.  A tile writes
//                         its partial product to slab `slab_base + split` of the problem's output; weight-gradient problems
//                         also emit the column sums of their A operand (the bias gradient) from the fragments they stage.
//   vqa_grouped_epilogue  ONE launch reduces the slabs of every output of the phase in a fixed order (bitwise
//                         reproducible, no atomics) and applies what surrounds the GEMM in the layer: bias, activation, the
//                         NEXT layer's input dropout (stored dropped: relu and mask gate the backward together), the rank
//                         product of the vector-vector Mutan fusion, activation / dropout gradients, layouts.
//
// M = 512 rows, N = 155..2048, K = 310..2400: 0.3-3 GFLOP per product.  One product cannot fill 256 CUs without a deep
// K split; a phase's products together can (a few hundred 64x64 tiles x 2-8 splits), and the 2-6 launch-floor kernels
// that used to surround each library GEMM (dropout, bias + activation, activation gradient + column sums, slices, adds)
// become arithmetic in the one epilogue launch.
#include "gemm_f32_mfma.hpp"

namespace vqa {

constexpr int kGMaxProbs = VQA_GROUPED_MAX;
constexpr int kGBM = 64, kGBN = 64, kGBK = 16, kGPF = 2;

struct GProbs {
  VqaGemmProblem p[kGMaxProbs];
  int first[kGMaxProbs + 1];  // first work item of each problem
  int n;
};

__device__ __forceinline__ DropCfg make_drop_dev(float p, uint64_t seed, const uint64_t* seed_ptr) {
  int p8 = (int)(p * 256.f + 0.5f);   // (as make_drop on the host)
  p8 = p8 < 0 ? 0 : (p8 > 255 ? 255 : p8);
  return DropCfg{(uint32_t)p8, 256.f / (256.f - (float)p8), seed, seed_ptr};
}

__device__ __forceinline__ float act_fwd_g(float z, int act) {
  if (act == 1) return fmaxf(z, 0.f);
  if (act == 2) return 1.f / (1.f + expf(-z));
  return z;
}

// Operands whose base or row stride is only 4-byte aligned (a 155-wide glimpse block inside a [B,620] tensor): the same
// sources with two 4-byte loads per slot instead of one 8-byte load.
struct SrcKC1 {  // X[mn][k], K-contiguous rows of stride ld; any alignment, any K >= 1
  struct Raw {
    float a, b;
  };
  const float* p;
  int ld, MN, K;
  __device__ __forceinline__ Raw fetch(int mn, int k) const {
    const float* row = p + (size_t)min(mn, MN - 1) * ld;
    return Raw{row[min(k, K - 1)], row[min(k + 1, K - 1)]};
  }
  __device__ __forceinline__ float2 finish(Raw v, int mn, int k) const {
    return make_float2(mn < MN && k < K ? v.a : 0.f, mn < MN && k + 1 < K ? v.b : 0.f);
  }
};
struct SrcMC1 {  // X[k][mn], MN-contiguous rows of stride ld; any alignment, any MN >= 1
  struct Raw {
    float a, b;
  };
  const float* p;
  int ld, MN, K;
  __device__ __forceinline__ Raw fetch(int mn, int k) const {
    const float* row = p + (size_t)min(k, K - 1) * ld;
    return Raw{row[min(mn, MN - 1)], row[min(mn + 1, MN - 1)]};
  }
  __device__ __forceinline__ float2 finish(Raw v, int mn, int k) const {
    return make_float2(k < K && mn < MN ? v.a : 0.f, k < K && mn + 1 < MN ? v.b : 0.f);
  }
};

template <bool A_KC, bool B_KC, class SrcA, class SrcB>
__device__ __forceinline__ void grouped_tile(const VqaGemmProblem& pr, const SrcA& sa, const SrcB& sb, int m0, int n0,
                                             int split, float* smem) {
  f32x16 acc[1][1];
  zero_acc(acc);
  const int k_begin = split * pr.ksplit, k_end = min(pr.K, k_begin + pr.ksplit);
  float colsum[1] = {0.f};
  const bool want_colsum = pr.colsum != nullptr && n0 == 0;   // (wave-uniform; form TN only)
  gemm_tile<kGBM, kGBN, kGBK, kGPF, A_KC, B_KC>(sa, sb, m0, n0, k_begin, k_end, smem, acc, want_colsum ? colsum : nullptr);
  const AccCoord<kGBM, kGBN> cc(m0, n0);
  float* __restrict__ dst = pr.slab + (size_t)(pr.slab_base + split) * pr.slab_stride;
  const int col = cc.col(0);
  if (col < pr.N) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = cc.row(0, i);
      if (row < pr.M) dst[(size_t)row * pr.N + col] = acc[0][0][i];
    }
  }
  if (want_colsum) {
    // lane l holds the sum over the staged k's with (k & 1) == l >> 5 of A[k][m0 + wave_row0 + (l & 31)]: add the halves
    const unsigned u = __float_as_uint(colsum[0]);
    const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const float total = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = m0 + (wave >> 1) * 32 + (lane & 31);
    if ((wave & 1) == 0 && lane < 32 && m < pr.M) pr.colsum[(size_t)(pr.slab_base + split) * pr.M + m] = total;
  }
}

__global__ __launch_bounds__(kGemmThreads, 4) void grouped_gemm_kernel(GProbs g, int items) {
  extern __shared__ __attribute__((aligned(16))) float smem_g[];
  const int bid = xcd_remap(blockIdx.x, items);
  int p = 0;
  while (p + 1 < g.n && g.first[p + 1] <= bid) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const VqaGemmProblem& pr = g.p[p];
  const int local = bid - g.first[p];
  const int tiles_n = (pr.N + kGBN - 1) / kGBN, tiles_m = (pr.M + kGBM - 1) / kGBM;
  const int split = local / (tiles_m * tiles_n), t = local % (tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * kGBM, n0 = (t % tiles_n) * kGBN;
  if (pr.form == 0) {          // NT: A [M,K] rows K-contiguous, B [N,K] rows K-contiguous
    grouped_tile<true, true>(pr, SrcKC{pr.A, pr.lda, pr.M, pr.Ka}, SrcKC{pr.B, pr.ldb, pr.N, pr.Kb}, m0, n0, split, smem_g);
  } else if (pr.form == 1) {   // NN: A [M,K] rows K-contiguous, B [K,N] rows N-contiguous
    grouped_tile<true, false>(pr, SrcKC{pr.A, pr.lda, pr.M, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  } else if (pr.form == 2) {   // TN: A [K,M] rows M-contiguous, B [K,N] rows N-contiguous
    grouped_tile<false, false>(pr, SrcMC{pr.A, pr.lda, pr.Ma, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  } else if (pr.form == 3) {   // NN with a 4-byte aligned A
    grouped_tile<true, false>(pr, SrcKC1{pr.A, pr.lda, pr.M, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  } else {                     // TN with a 4-byte aligned A
    grouped_tile<false, false>(pr, SrcMC1{pr.A, pr.lda, pr.Ma, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  }
}

// ------------------------------------------------------------------------------------------------ epilogue
struct EJobs {
  VqaEpilogueJob j[kGMaxProbs];
  int first[kGMaxProbs + 1];  // first thread of each job
  int n;
};

__device__ __forceinline__ float slab_sum(const VqaEpilogueJob& j, size_t e) {
  float a = j.slab[e];
  for (int s = 1; s < j.S; ++s) a += j.slab[(size_t)s * j.slab_stride + e];
  return a;
}

// keep / (1 - p) factor of element `e` of the mask the job names (0: no dropout)
__device__ __forceinline__ float job_keep(const VqaEpilogueJob& j, uint32_t e) {
  if (j.p_drop <= 0.f) return 1.f;
  return drop_one(e, make_drop_dev(j.p_drop, j.seed, j.seed_ptr));
}

__global__ __launch_bounds__(256) void grouped_epilogue_kernel(EJobs g, int threads) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (tid >= threads) return;
  int q = 0;
  while (q + 1 < g.n && g.first[q + 1] <= tid) ++q;
  const VqaEpilogueJob& j = g.j[q];
  const int e = tid - g.first[q];
  switch (j.kind) {
    case VQA_EPI_LINEAR: {   // out[m, n] = drop(act(sum + bias[n]))
      const int m = e / j.N, n = e - m * j.N;
      float z = slab_sum(j, (size_t)e) + (j.bias != nullptr ? j.bias[n] : 0.f);
      z = act_fwd_g(z, j.act);
      z *= job_keep(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)n);
      j.out[(size_t)m * j.ldo + n] = z;
      break;
    }
    case VQA_EPI_RANK_PRODUCT: {   // N = R * H; thread = (m, h): h1 = sum + bias stored; out2[m,h] = drop(sum_r h1 * aux)
      const int H = j.N / j.R, m = e / H, h = e - m * H;
      float x = 0.f;
      for (int r = 0; r < j.R; ++r) {
        const size_t o = (size_t)m * j.N + (size_t)r * H + h;
        const float h1 = slab_sum(j, o) + (j.bias != nullptr ? j.bias[r * H + h] : 0.f);
        j.out[o] = h1;
        x = fmaf(h1, j.aux[(size_t)m * j.ld_aux + (size_t)r * H + h], x);
      }
      j.out2[(size_t)m * j.ldo + h] = x * job_keep(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)h);
      break;
    }
    case VQA_EPI_GRAD: {   // out[m, seg(n)] = sum * gate(y[m, n])   (the data gradient of a layer, gated for the layer in front)
      const int m = e / j.N, n = e - m * j.N;
      float z = slab_sum(j, (size_t)e);
      if (j.gate == 1) {          // relu, y possibly stored dropped: y > 0 <=> kept and active; gate_scale = 1 or 1/(1-p)
        z = j.aux[(size_t)m * j.ld_aux + n] > 0.f ? z * j.gate_scale : 0.f;
      } else if (j.gate == 2) {   // sigmoid
        const float y = j.aux[(size_t)m * j.ld_aux + n];
        z *= y * (1.f - y);
      }
      z *= job_keep(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)n);
      const int nn = j.seg > 0 ? (n / j.seg) * j.seg_ld + (n % j.seg) : n;
      j.out[(size_t)m * j.ldo + nn] = z;
      break;
    }
    case VQA_EPI_RANK_PRODUCT_BWD: {   // N = H; thread = (m, h): g = drop(sum); out[m, r*H+h] = g * aux[m, r*H+h]; out2[..] = g * aux2[..]
      const int m = e / j.N, h = e - m * j.N;
      const float gx = slab_sum(j, (size_t)e) * job_keep(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)h);
      for (int r = 0; r < j.R; ++r) {
        const size_t o = (size_t)m * j.R * j.N + (size_t)r * j.N + h;
        j.out[o] = gx * j.aux[(size_t)m * j.ld_aux + (size_t)r * j.N + h];
        j.out2[o] = gx * j.aux2[(size_t)m * j.ld_aux + (size_t)r * j.N + h];
      }
      break;
    }
    default: {   // VQA_EPI_SUM: out[m, n] = sum   (weight gradients into their slots; bias gradients with M = 1)
      const int m = e / j.N, n = e - m * j.N;
      j.out[(size_t)m * j.ldo + n] = slab_sum(j, (size_t)e);
      break;
    }
  }
}

}  // namespace vqa

using namespace vqa;

static int split_of(const VqaGemmProblem& p) { return (p.K + p.ksplit - 1) / p.ksplit; }

extern "C" int vqa_grouped_gemm(const VqaGemmProblem* problems, int n, vqa_stream_t stream) {
  VQA_REQUIRE(problems != nullptr && n >= 1 && n <= kGMaxProbs, VQA_E_BADARG, "grouped_gemm: 1..%d problems (got %d)", kGMaxProbs, n);
  GProbs g{};
  g.n = n;
  int items = 0;
  for (int i = 0; i < n; ++i) {
    VqaGemmProblem p = problems[i];
    VQA_REQUIRE(p.A && p.B && p.slab, VQA_E_BADARG, "grouped_gemm[%d]: null pointer", i);
    VQA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.form >= 0 && p.form <= 4, VQA_E_BADARG,
                "grouped_gemm[%d]: bad sizes M=%d N=%d K=%d form=%d", i, p.M, p.N, p.K, p.form);
    VQA_REQUIRE(p.ksplit > 0 && p.ksplit % kGBK == 0 && p.slab_base >= 0 && p.slab_stride >= (long long)p.M * p.N, VQA_E_BADARG,
                "grouped_gemm[%d]: ksplit %d must be a positive multiple of %d, slab_stride >= M*N", i, p.ksplit, kGBK);
    // source extents default to the problem's own; a caller overrides them when an operand is zero-padded past the
    // contraction / output extent (Ka, Kb: valid contraction length of A / B; Ma, Nb: valid width of an MN-contiguous A / B)
    if (p.Ka <= 0) p.Ka = p.K;
    if (p.Kb <= 0) p.Kb = p.K;
    if (p.Ma <= 0) p.Ma = p.M;
    if (p.Nb <= 0) p.Nb = p.N;
    // 8-byte operand loads: even leading dimensions, 8-byte aligned bases, even extents along the contiguous axis
    // (forms 3 / 4 read A with 4-byte loads: no requirement on A)
    const bool a_kc = p.form != 2 && p.form != 4, b_kc = p.form == 0, a_free = p.form >= 3;
    VQA_REQUIRE(aligned(p.A, 4) && aligned(p.B, 8) && p.ldb % 2 == 0 && (b_kc ? p.Kb : p.Nb) % 2 == 0 && (b_kc ? p.Kb : p.Nb) >= 2 &&
                    (a_free || (p.lda % 2 == 0 && aligned(p.A, 8) && (a_kc ? p.Ka : p.Ma) % 2 == 0 && (a_kc ? p.Ka : p.Ma) >= 2)),
                VQA_E_UNSUPPORTED,
                "grouped_gemm[%d]: operands need even leading dimensions / contiguous extents and 8-byte aligned bases "
                "(form %d lda=%d ldb=%d)", i, p.form, p.lda, p.ldb);
    VQA_REQUIRE(p.colsum == nullptr || p.form == 2 || p.form == 4, VQA_E_BADARG,
                "grouped_gemm[%d]: column sums exist for the TN forms only", i);
    g.p[i] = p;
    g.first[i] = items;
    const long tiles = (long)((p.M + kGBM - 1) / kGBM) * ((p.N + kGBN - 1) / kGBN) * split_of(p);
    VQA_REQUIRE(items + tiles < (1L << 24), VQA_E_UNSUPPORTED, "grouped_gemm: too many tiles");
    items += (int)tiles;
  }
  g.first[n] = items;
  const size_t lds = GemmTile<kGBM, kGBN, kGBK, true, true>::kSmemBytes;   // the largest of the three forms
  hipLaunchKernelGGL(grouped_gemm_kernel, dim3(items), dim3(kGemmThreads), lds, static_cast<hipStream_t>(stream), g, items);
  return check_launch("grouped_gemm");
}

extern "C" int vqa_grouped_epilogue(const VqaEpilogueJob* jobs, int n, vqa_stream_t stream) {
  VQA_REQUIRE(jobs != nullptr && n >= 1 && n <= kGMaxProbs, VQA_E_BADARG, "grouped_epilogue: 1..%d jobs (got %d)", kGMaxProbs, n);
  EJobs g{};
  g.n = n;
  long threads = 0;
  for (int i = 0; i < n; ++i) {
    const VqaEpilogueJob& j = jobs[i];
    VQA_REQUIRE(j.slab && j.out && j.S >= 1 && j.M > 0 && j.N > 0, VQA_E_BADARG, "grouped_epilogue[%d]: bad job", i);
    VQA_REQUIRE(j.kind >= VQA_EPI_SUM && j.kind <= VQA_EPI_RANK_PRODUCT_BWD, VQA_E_BADARG, "grouped_epilogue[%d]: kind %d", i, j.kind);
    VQA_REQUIRE(j.p_drop >= 0.f && j.p_drop < 1.f, VQA_E_BADARG, "grouped_epilogue[%d]: p_drop=%f", i, (double)j.p_drop);
    long count = (long)j.M * j.N;
    if (j.kind == VQA_EPI_RANK_PRODUCT) {
      VQA_REQUIRE(j.R >= 1 && j.N % j.R == 0 && j.aux && j.out2, VQA_E_BADARG, "grouped_epilogue[%d]: rank product needs R | N, aux, out2", i);
      count /= j.R;
    }
    if (j.kind == VQA_EPI_RANK_PRODUCT_BWD)
      VQA_REQUIRE(j.R >= 1 && j.aux && j.aux2 && j.out2, VQA_E_BADARG, "grouped_epilogue[%d]: rank product backward needs aux, aux2, out2", i);
    if (j.kind == VQA_EPI_GRAD && j.gate != 0) VQA_REQUIRE(j.aux != nullptr, VQA_E_BADARG, "grouped_epilogue[%d]: gate needs aux", i);
    g.j[i] = j;
    g.first[i] = (int)threads;
    threads += count;
    VQA_REQUIRE(threads < (1L << 30), VQA_E_UNSUPPORTED, "grouped_epilogue: too many elements");
  }
  g.first[n] = (int)threads;
  hipLaunchKernelGGL(grouped_epilogue_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, (int)threads);
  return check_launch("grouped_epilogue");
}
