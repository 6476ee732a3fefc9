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
// Tried and removed (round 3, measured on the MI355X): the same grouped launch on the register-tile engine -- one WAVE per
// 64 x 80 (NT) / 64 x 64 (NN, TN) item, operand fragments straight from L2 -- was correct and no faster than this form
// (q-projection phase 49 us on either; whole step 2.62 ms against 2.46): at 8-9 sixteen-byte fragment loads per 64-80
// MFMAs the kernel runs at half its MFMA stream whatever the prefetch depth (2 / 3 register sets), the line usage (16- /
// 32-deep steps), the waves per SIMD or the workgroup placement -- the L1 / address path of 16-row fragment loads, as
// DESIGN.md 5c had found for small tiles; M = 512 rows leave no room for the 9 x 5-block tiles that make that engine pay.
//
// M = 512 rows, N = 155..2048, K = 310..2400: 0.3-3 GFLOP per product.  One product cannot fill 256 CUs without a deep
// K split; a phase's products together can (a few hundred 64x64 tiles x 2-8 splits), and the 2-6 launch-floor kernels
// that used to surround each library GEMM (dropout, bias + activation, activation gradient + column sums, slices, adds)
// become arithmetic in the one epilogue launch.
#include "gemm_f32_mfma.hpp"

namespace vqa {

constexpr int kGMaxProbs = VQA_GROUPED_MAX;
constexpr int kGMaxGemms = VQA_GROUPED_GEMM_MAX;
constexpr int kGBN = 64, kGBK = 16, kGPF = 2;   // 64-column tiles, 16-deep K steps, two register sets in flight; tile rows
                                                // BM: 64 (default) or 128 (VQA_GROUPED_BM), one value per launch

struct GProbs {
  VqaGemmProblem p[kGMaxGemms];
  int first[kGMaxGemms + 1];  // first work item of each problem
  int n;
};

// The problem / job tables are kernel arguments BY VALUE (no device allocation, no copy node in a captured graph) and are
// indexed with a run-time (wave-uniform) index.  Indexing the parameter object itself makes the compiler copy the whole
// table to scratch in every lane (2.7 KB per lane); reading it through the kernarg segment pointer keeps it in constant
// memory: scalar loads, no scratch.  (The table is the kernel's FIRST argument: offset 0 of the segment.)
template <class T>
__device__ __forceinline__ const T& kernarg_table() {
  return *(const T*)__builtin_amdgcn_kernarg_segment_ptr();   // (C-style: an address-space cast)
}

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
  __device__ __forceinline__ float2 plain(Raw v) const { return make_float2(v.a, v.b); }
  __device__ __forceinline__ bool covers(int mn0, int n, int k0, int k1) const { return mn0 + n <= MN && k1 <= K; }
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
  __device__ __forceinline__ float2 plain(Raw v) const { return make_float2(v.a, v.b); }
  __device__ __forceinline__ bool covers(int mn0, int n, int k0, int k1) const { return mn0 + n <= MN && k1 <= K; }
};

template <int BM, int BK, int PF, bool A_KC, bool B_KC, class SrcA, class SrcB>
__device__ __forceinline__ void grouped_tile(const VqaGemmProblem& pr, const SrcA& sa, const SrcB& sb, int m0, int n0,
                                             int split, float* smem) {
  constexpr int TM = BM / 64;
  f32x16 acc[TM][1];
  zero_acc(acc);
  const int k_begin = split * pr.ksplit, k_end = min(pr.K, k_begin + pr.ksplit);
  float colsum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) colsum[i] = 0.f;
  const bool want_colsum = (pr.colsum != nullptr || pr.colsum_out != nullptr) && n0 == 0;   // (wave-uniform; TN forms only)
  // (colsum is passed ALWAYS: `want_colsum ? colsum : nullptr` made the array's address a run-time value, the array went to
  //  scratch memory and every K step of every tile did a scratch load + add + store chain -- 49 scratch instructions in the ISA,
  //  8 bytes of private segment per lane.  Eight VALU adds per step are free next to the staging pass; the sums are only USED
  //  when the problem wants them.)
  gemm_tile<BM, kGBN, BK, PF, A_KC, B_KC, false, BK == 16, true>(sa, sb, m0, n0, k_begin, k_end, smem, acc, colsum);
  const AccCoord<BM, kGBN> cc(m0, n0);
  const int col = cc.col(0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (pr.out != nullptr) {
    // direct output: the layer's epilogue on the accumulators (bias, activation, gate of the layer in front, dropout)
    if (col < pr.N) {
      const float bv = pr.bias != nullptr ? pr.bias[col] : 0.f;
      const bool drop = pr.p_drop > 0.f;
      DropCfg dc{};
      if (drop) dc = make_drop_dev(pr.p_drop, pr.seed, pr.seed_ptr);
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          if (row < pr.M) {
            float z = act_fwd_g(acc[tm][0][i] + bv, pr.act);
            if (pr.gate != 0) {
              const float y = pr.gate_y[(size_t)row * pr.ld_gate + col];
              z = pr.gate == 1 ? (y > 0.f ? z * pr.gate_scale : 0.f) : z * y * (1.f - y);
            }
            if (drop) z *= drop_one(pr.drop_base + (uint32_t)row * pr.drop_ld + (uint32_t)col, dc);
            pr.out[(size_t)row * pr.ldo + col] = z;
          }
        }
    }
  } else {
    float* __restrict__ dst = pr.slab + (size_t)(pr.slab_base + split) * pr.slab_stride;
    if (col < pr.N) {
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = cc.row(tm, i);
          if (row < pr.M) dst[(size_t)row * pr.N + col] = acc[tm][0][i];
        }
    }
  }
  if (want_colsum) {
    // lane l holds the sum over the staged k's with (k & 1) == l >> 5 of A[k][m0 + wave_row0 + i*32 + (l & 31)]: add the halves
    float* cs = pr.colsum_out != nullptr ? pr.colsum_out : pr.colsum + (size_t)(pr.slab_base + split) * pr.M;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const unsigned u = __float_as_uint(colsum[i]);
      const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
      const float total = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      const int m = m0 + (wave >> 1) * (TM * 32) + i * 32 + (lane & 31);
      if ((wave & 1) == 0 && lane < 32 && m < pr.M) cs[m] = total;
    }
  }
}

template <int BM, int BK, int PF>
__global__ __launch_bounds__(kGemmThreads, BM == 64 ? 4 : 2) void grouped_gemm_kernel(GProbs g_arg, int items) {
  const GProbs& g = kernarg_table<GProbs>();
  extern __shared__ __attribute__((aligned(16))) float smem_g[];
  const int bid = xcd_remap(blockIdx.x, items);
  int p = 0;
  while (p + 1 < g.n && g.first[p + 1] <= bid) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const VqaGemmProblem& pr = g.p[p];
  const int local = bid - g.first[p];
  const int tiles_n = (pr.N + kGBN - 1) / kGBN, tiles_m = (pr.M + BM - 1) / BM;
  const int split = local / (tiles_m * tiles_n), t = local % (tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * kGBN;
  if (pr.form == 0) {          // NT: A [M,K] rows K-contiguous, B [N,K] rows K-contiguous
    grouped_tile<BM, BK, PF, true, true>(pr, SrcKC{pr.A, pr.lda, pr.M, pr.Ka}, SrcKC{pr.B, pr.ldb, pr.N, pr.Kb}, m0, n0, split, smem_g);
  } else if (pr.form == 1) {   // NN: A [M,K] rows K-contiguous, B [K,N] rows N-contiguous
    grouped_tile<BM, BK, PF, true, false>(pr, SrcKC{pr.A, pr.lda, pr.M, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  } else if (pr.form == 2) {   // TN: A [K,M] rows M-contiguous, B [K,N] rows N-contiguous
    grouped_tile<BM, BK, PF, false, false>(pr, SrcMC{pr.A, pr.lda, pr.Ma, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  } else if (pr.form == 3) {   // NN with a 4-byte aligned A
    grouped_tile<BM, BK, PF, true, false>(pr, SrcKC1{pr.A, pr.lda, pr.M, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  } else {                     // TN with a 4-byte aligned A
    grouped_tile<BM, BK, PF, false, false>(pr, SrcMC1{pr.A, pr.lda, pr.Ma, pr.Ka}, SrcMC{pr.B, pr.ldb, pr.Nb, pr.Kb}, m0, n0, split, smem_g);
  }
}

// ------------------------------------------------------------------------------------------------ epilogue
// A thread owns V consecutive elements of a row (V = 2 when the job's widths and pointers allow 8-byte accesses, which is
// every job of the models except the 155-wide glimpse blocks; else 1).
struct EJobs {
  VqaEpilogueJob j[kGMaxProbs];
  int first[kGMaxProbs + 1];  // first thread of each job
  float inv_w[kGMaxProbs];    // 1 / (threads per row)
  unsigned char vec[kGMaxProbs];
  int n;
};

template <int V>
struct Vals {
  float v[V];
};
template <int V>
__device__ __forceinline__ Vals<V> ldv(const float* p) {
  Vals<V> r;
  if constexpr (V == 2) {
    const float2 t = ld2(p);
    r.v[0] = t.x;
    r.v[1] = t.y;
  } else {
    r.v[0] = p[0];
  }
  return r;
}
template <int V>
__device__ __forceinline__ void stv(float* p, const Vals<V>& x) {
  if constexpr (V == 2) {
    st2(p, make_float2(x.v[0], x.v[1]));
  } else {
    p[0] = x.v[0];
  }
}
template <int V>
__device__ __forceinline__ Vals<V> slab_sum(const VqaEpilogueJob& j, size_t e) {
  Vals<V> a = ldv<V>(j.slab + e);
  for (int s = 1; s < j.S; ++s) {
    const Vals<V> t = ldv<V>(j.slab + (size_t)s * j.slab_stride + e);
#pragma unroll
    for (int i = 0; i < V; ++i) a.v[i] += t.v[i];
  }
  return a;
}
// keep / (1 - p) factors of mask elements e .. e + V - 1 (e even when V == 2); all ones without dropout
template <int V>
__device__ __forceinline__ Vals<V> job_keep(const VqaEpilogueJob& j, uint32_t e) {
  Vals<V> k;
#pragma unroll
  for (int i = 0; i < V; ++i) k.v[i] = 1.f;
  if (j.p_drop > 0.f) {
    const DropCfg dc = make_drop_dev(j.p_drop, j.seed, j.seed_ptr);
    if constexpr (V == 2) {
      const float2 t = drop_pair(e, dc);
      k.v[0] = t.x;
      k.v[1] = t.y;
    } else {
      k.v[0] = drop_one(e, dc);
    }
  }
  return k;
}

template <int V>
__device__ __forceinline__ void epilogue_item(const VqaEpilogueJob& j, int m, int c) {   // row m, first column c
  switch (j.kind) {
    case VQA_EPI_LINEAR: {   // out[m, n] = drop(act(sum + bias[n]))
      Vals<V> z = slab_sum<V>(j, (size_t)m * j.N + c);
      const Vals<V> k = job_keep<V>(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)c);
#pragma unroll
      for (int i = 0; i < V; ++i) z.v[i] = act_fwd_g(z.v[i] + (j.bias != nullptr ? j.bias[c + i] : 0.f), j.act) * k.v[i];
      stv<V>(j.out + (size_t)m * j.ldo + c, z);
      break;
    }
    case VQA_EPI_RANK_PRODUCT: {   // N = R * H; (m, h = c): h1 = sum + bias stored; out2[m,h] = drop(sum_r h1 * aux)
      const int H = j.N / j.R;
      Vals<V> x;
#pragma unroll
      for (int i = 0; i < V; ++i) x.v[i] = 0.f;
      for (int r = 0; r < j.R; ++r) {
        const size_t o = (size_t)m * j.N + (size_t)r * H + c;
        Vals<V> h1 = slab_sum<V>(j, o);
        const Vals<V> a = ldv<V>(j.aux + (size_t)m * j.ld_aux + (size_t)r * H + c);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          h1.v[i] += j.bias != nullptr ? j.bias[r * H + c + i] : 0.f;
          x.v[i] = fmaf(h1.v[i], a.v[i], x.v[i]);
        }
        stv<V>(j.out + o, h1);
      }
      const Vals<V> k = job_keep<V>(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)c);
#pragma unroll
      for (int i = 0; i < V; ++i) x.v[i] *= k.v[i];
      stv<V>(j.out2 + (size_t)m * j.ldo + c, x);
      break;
    }
    case VQA_EPI_GRAD: {   // out[m, n] = sum * gate(y[m, n]) * keep   (a data gradient, gated for the layer in front)
      Vals<V> z = slab_sum<V>(j, (size_t)m * j.N + c);
      if (j.gate != 0) {
        const Vals<V> y = ldv<V>(j.aux + (size_t)m * j.ld_aux + c);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          // 1: relu, y possibly stored dropped (y > 0 <=> kept and active; gate_scale = 1 or 1/(1-p));  2: sigmoid
          if (j.gate == 1) z.v[i] = y.v[i] > 0.f ? z.v[i] * j.gate_scale : 0.f;
          else z.v[i] *= y.v[i] * (1.f - y.v[i]);
        }
      }
      const Vals<V> k = job_keep<V>(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)c);
#pragma unroll
      for (int i = 0; i < V; ++i) z.v[i] *= k.v[i];
      stv<V>(j.out + (size_t)m * j.ldo + c, z);
      break;
    }
    case VQA_EPI_RANK_PRODUCT_BWD: {   // N = H; (m, h = c): g = drop(sum); out[m, r*H+h] = g * aux[..]; out2[..] = g * aux2[..]
      Vals<V> gx = slab_sum<V>(j, (size_t)m * j.N + c);
      const Vals<V> k = job_keep<V>(j, j.drop_base + (uint32_t)m * j.drop_ld + (uint32_t)c);
#pragma unroll
      for (int i = 0; i < V; ++i) gx.v[i] *= k.v[i];
      for (int r = 0; r < j.R; ++r) {
        const size_t o = (size_t)m * j.R * j.N + (size_t)r * j.N + c, oa = (size_t)m * j.ld_aux + (size_t)r * j.N + c;
        const Vals<V> a = ldv<V>(j.aux + oa), a2 = ldv<V>(j.aux2 + oa);
        Vals<V> o1, o2;
#pragma unroll
        for (int i = 0; i < V; ++i) {
          o1.v[i] = gx.v[i] * a.v[i];
          o2.v[i] = gx.v[i] * a2.v[i];
        }
        stv<V>(j.out + o, o1);
        stv<V>(j.out2 + o, o2);
      }
      break;
    }
    default: {   // VQA_EPI_SUM: out[m, n] = sum   (weight gradients into their slots; bias gradients with M = 1)
      stv<V>(j.out + (size_t)m * j.ldo + c, slab_sum<V>(j, (size_t)m * j.N + c));
      break;
    }
  }
}

__global__ __launch_bounds__(256) void grouped_epilogue_kernel(EJobs g_arg, int threads) {
  const EJobs& g = kernarg_table<EJobs>();
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (tid >= threads) return;
  int q = 0;
  while (q + 1 < g.n && g.first[q + 1] <= tid) ++q;
  const VqaEpilogueJob& j = g.j[q];
  const int e = tid - g.first[q];
  const int V = g.vec[q];
  int width = j.N;                                     // columns a row's threads cover
  if (j.kind == VQA_EPI_RANK_PRODUCT) width = j.N / j.R;
  const int tpr = width / V;                           // threads per row
  int m = (int)(((float)e + 0.5f) * g.inv_w[q]);       // e / tpr, corrected below (no integer division in the kernel)
  int t = e - m * tpr;
  if (t < 0) {
    --m;
    t += tpr;
  } else if (t >= tpr) {
    ++m;
    t -= tpr;
  }
  if (V == 2) epilogue_item<2>(j, m, 2 * t);
  else epilogue_item<1>(j, m, t);
}

}  // namespace vqa

using namespace vqa;

static int split_of(const VqaGemmProblem& p) { return (p.K + p.ksplit - 1) / p.ksplit; }

extern "C" int vqa_grouped_gemm(const VqaGemmProblem* problems, int n, vqa_stream_t stream) {
  VQA_REQUIRE(problems != nullptr && n >= 1 && n <= kGMaxGemms, VQA_E_BADARG, "grouped_gemm: 1..%d problems (got %d)", kGMaxGemms, n);
  GProbs g{};
  g.n = n;
  int items = 0;
  const char* tile = vqa::option("VQA_GROUPED_BM");     // tile rows: 64 (default) or 128
  const int bm = (tile != nullptr && std::atoi(tile) == 128) ? 128 : 64;
  constexpr int bk = kGBK;
  for (int i = 0; i < n; ++i) {
    VqaGemmProblem p = problems[i];
    VQA_REQUIRE(p.A && p.B && (p.slab || p.out), VQA_E_BADARG, "grouped_gemm[%d]: null pointer", i);
    VQA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.form >= 0 && p.form <= 4, VQA_E_BADARG,
                "grouped_gemm[%d]: bad sizes M=%d N=%d K=%d form=%d", i, p.M, p.N, p.K, p.form);
    VQA_REQUIRE(p.ksplit > 0 && p.ksplit % bk == 0 && p.slab_base >= 0 && (p.out != nullptr || p.slab_stride >= (long long)p.M * p.N),
                VQA_E_BADARG, "grouped_gemm[%d]: ksplit %d must be a positive multiple of %d, slab_stride >= M*N", i, p.ksplit, bk);
    if (p.out != nullptr) {
      VQA_REQUIRE(p.ksplit >= p.K, VQA_E_BADARG, "grouped_gemm[%d]: a direct output needs the contraction in one part", i);
      VQA_REQUIRE(p.ldo >= p.N && p.act >= 0 && p.act <= 2 && p.gate >= 0 && p.gate <= 2 && (p.gate == 0 || p.gate_y != nullptr) &&
                      p.p_drop >= 0.f && p.p_drop < 1.f,
                  VQA_E_BADARG, "grouped_gemm[%d]: bad direct-output epilogue", i);
    } else {
      VQA_REQUIRE(p.colsum_out == nullptr, VQA_E_BADARG, "grouped_gemm[%d]: colsum_out needs a direct output", i);
    }
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
    VQA_REQUIRE((p.colsum == nullptr && p.colsum_out == nullptr) || p.form == 2 || p.form == 4, VQA_E_BADARG,
                "grouped_gemm[%d]: column sums exist for the TN forms only", i);
    g.p[i] = p;
    g.first[i] = items;
    const long tiles = (long)((p.M + bm - 1) / bm) * ((p.N + kGBN - 1) / kGBN) * split_of(p);
    VQA_REQUIRE(items + tiles < (1L << 24), VQA_E_UNSUPPORTED, "grouped_gemm: too many tiles");
    items += (int)tiles;
  }
  g.first[n] = items;
  const auto launch = [&](auto kernel, size_t lds) {
    VQA_LAUNCH(kernel, dim3(items), dim3(kGemmThreads), lds, static_cast<hipStream_t>(stream), g, items);
  };
  // LDS: the largest of the three operand forms (both operands K-contiguous, in their own orientation)
  if (bm == 128)
    launch(grouped_gemm_kernel<128, kGBK, kGPF>, 2 * gemm_stage_floats<128, kGBN, kGBK, true, true, true>() * sizeof(float));
  else
    launch(grouped_gemm_kernel<64, kGBK, kGPF>, 2 * gemm_stage_floats<64, kGBN, kGBK, true, true, true>() * sizeof(float));
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
    int width = j.N;
    if (j.kind == VQA_EPI_RANK_PRODUCT) {
      VQA_REQUIRE(j.R >= 1 && j.N % j.R == 0 && j.aux && j.out2, VQA_E_BADARG, "grouped_epilogue[%d]: rank product needs R | N, aux, out2", i);
      width = j.N / j.R;
    }
    if (j.kind == VQA_EPI_RANK_PRODUCT_BWD)
      VQA_REQUIRE(j.R >= 1 && j.aux && j.aux2 && j.out2, VQA_E_BADARG, "grouped_epilogue[%d]: rank product backward needs aux, aux2, out2", i);
    if (j.kind == VQA_EPI_GRAD && j.gate != 0) VQA_REQUIRE(j.aux != nullptr, VQA_E_BADARG, "grouped_epilogue[%d]: gate needs aux", i);
    // 8-byte accesses when every width, stride and pointer the job touches is even / 8-byte aligned
    const bool v2 = width % 2 == 0 && j.N % 2 == 0 && j.ldo % 2 == 0 && j.ld_aux % 2 == 0 && j.slab_stride % 2 == 0 &&
                    j.drop_ld % 2 == 0 && j.drop_base % 2 == 0 && aligned(j.slab, 8) && aligned(j.out, 8) &&
                    (j.out2 == nullptr || aligned(j.out2, 8)) && (j.aux == nullptr || aligned(j.aux, 8)) &&
                    (j.aux2 == nullptr || aligned(j.aux2, 8)) && (j.kind != VQA_EPI_RANK_PRODUCT_BWD || (j.R * j.N) % 2 == 0);
    const int V = v2 ? 2 : 1;
    g.j[i] = j;
    g.vec[i] = (unsigned char)V;
    g.inv_w[i] = 1.0f / (float)(width / V);
    g.first[i] = (int)threads;
    threads += (long)j.M * (width / V);
    VQA_REQUIRE((long)j.M * (width / V) < (1L << 24), VQA_E_UNSUPPORTED, "grouped_epilogue[%d]: job too large", i);
    VQA_REQUIRE(threads < (1L << 30), VQA_E_UNSUPPORTED, "grouped_epilogue: too many elements");
  }
  g.first[n] = (int)threads;
  VQA_LAUNCH(grouped_epilogue_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, (int)threads);
  return check_launch("grouped_epilogue");
}
