// K5 on the split engine (gemm_f32_split.hpp): the tall region projections of CoR2 / ODA (config/CoR2.py:72-88 at :168-169,
// :213,:218; M = B*36 rows, K = 2048, N = 310) with every fp32 product formed from exact three-way bf16 splits of its
// operands on the bf16 matrix pipe.  Same contract as vqa_linear_act_fwd / the weight-gradient half of vqa_linear_act_bwd
// (same dropout mask, same epilogues, fp32 in and out); selected by the host when the option VQA_F32_PRODUCTS is "split".
#include "common.hpp"
#include "gemm_f32_split.hpp"

namespace vqa {

struct SplitEpiBiasAct {
  float* y;
  const float* bias;
  int ldy, act;
  float scale;
  __device__ __forceinline__ void operator()(int row, int col, float v) const {
    v = fmaf(v, scale, bias != nullptr ? bias[col] : 0.f);      // scale: 2 when the operand was masked unscaled (p = 0.5)
    if (act == 1) v = fmaxf(v, 0.f);
    y[(size_t)row * ldy + col] = v;
  }
};

static size_t round256(size_t b) { return (b + 255) & ~(size_t)255; }

static bool split_shape_ok(int M, int K, int N, int ldx, float p_drop) {
  const DropCfg dc = make_drop(p_drop, 0);
  return M >= 1152 && K >= 128 && K % 64 == 0 && N >= 16 && N % 2 == 0 && (N + 15) / 16 <= sp::kPackMaxBlocks && ldx % 4 == 0 && ldx >= K &&
         (size_t)M * ldx * 4 < (1ull << 32) && (size_t)M * N * 4 < (1ull << 32) && sp::packed_bytes(N, K) < (1ull << 32) &&
         (dc.p8 == 0 || dc.p8 == kDropHalf);
}
static int tn_slabs() {
  int s = 16;   // 16 n2 tiles x 16 row slabs = 256 workgroups at 310 x 2048
  if (const char* e = vqa::option("VQA_SPLIT_DW_SLABS")) s = std::atoi(e);
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_linear_split_supported(int M, int K, int N, int ldx, float p_drop) {
  return split_shape_ok(M, K, N, ldx, p_drop) ? 1 : 0;
}

extern "C" size_t vqa_linear_act_fwd_split_workspace_bytes(int K, int N) { return round256(sp::packed_bytes(N, K)); }

extern "C" int vqa_linear_act_fwd_split(const float* x, int ldx, const float* w, const float* bias, float* y, void* workspace,
                                        size_t workspace_bytes, int M, int K, int N, int act, float p_drop, uint64_t seed,
                                        const uint64_t* seed_ptr, vqa_stream_t stream) {
  VQA_REQUIRE(x && w && y && workspace, VQA_E_BADARG, "linear_act_fwd_split: null pointer");
  VQA_REQUIRE(act == 0 || act == 1, VQA_E_BADARG, "linear_act_fwd_split: act must be 0 (none) or 1 (relu), got %d", act);
  VQA_REQUIRE(split_shape_ok(M, K, N, ldx, p_drop), VQA_E_UNSUPPORTED,
              "linear_act_fwd_split: shape outside the split engine (M=%d K=%d N=%d ldx=%d p=%f); see vqa_linear_split_supported",
              M, K, N, ldx, (double)p_drop);
  VQA_REQUIRE(aligned(x, 16) && aligned(w, 16) && aligned(y, 8) && aligned(workspace, 16), VQA_E_UNSUPPORTED,
              "linear_act_fwd_split: x, w, workspace must be 16-byte aligned");
  VQA_REQUIRE(workspace_bytes >= vqa_linear_act_fwd_split_workspace_bytes(K, N), VQA_E_BADARG,
              "linear_act_fwd_split: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  sp::u32x4* wp = static_cast<sp::u32x4*>(workspace);
  {
    const long threads = (long)((N + 15) / 16) * (K / sp::kChunk) * 64;
    VQA_LAUNCH((sp::split_pack_kernel<false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w, K, N, K, wp);
  }
  // 144 x 160 workgroup tiles, each wave 9 x 5 accumulator blocks over half of K (M = 18432, N = 310: 256 workgroups)
  using S = rt::NtShape<9, 5, 1, 2, 2>;
  const int tiles_m = (M + S::BM - 1) / S::BM, tiles_n = (N + S::BN - 1) / S::BN;
  const sp::NtArgs a{x, wp, ldx, M, N, K, tiles_n, w, K};      // (w, K: the repair path's operand)
  const SplitEpiBiasAct epi{y, bias, N, act, dc.p8 > 0 ? dc.scale : 1.f};
  if (dc.p8 > 0) {
    VQA_ENSURE_LDS((sp::gemm_nt_kernel<9, 5, 1, 2, 2, true, SplitEpiBiasAct, 0, 3>), S::kLdsBytes);
    VQA_LAUNCH((sp::gemm_nt_kernel<9, 5, 1, 2, 2, true, SplitEpiBiasAct, 0, 3>), dim3(tiles_m * tiles_n), dim3(sp::kThreads),
               S::kLdsBytes, s, a, dc, epi);
  } else {
    VQA_ENSURE_LDS((sp::gemm_nt_kernel<9, 5, 1, 2, 2, false, SplitEpiBiasAct, 0, 3>), S::kLdsBytes);
    VQA_LAUNCH((sp::gemm_nt_kernel<9, 5, 1, 2, 2, false, SplitEpiBiasAct, 0, 3>), dim3(tiles_m * tiles_n), dim3(sp::kThreads),
               S::kLdsBytes, s, a, dc, epi);
  }
  return check_launch("linear_act_fwd_split");
}

// ---- K1 -> K5 fused (config/CoR2.py:215-218): y = act(drop(t + c2 * v) W^T + b) without v2 in HBM ---------------------------------
static bool relation_linear_ok(int B, int Nr, int D, int L, float p_drop) {
  using S = rt::NtShape<9, 5, 1, 2, 2>;
  return B >= 1 && Nr >= 8 && Nr <= 1024 && split_shape_ok(B * Nr, D, L, D, p_drop) && D % 128 == 0 &&
         sp::rel_lds_bytes(S::BM, Nr, D) <= 96 * 1024 && (size_t)B * Nr * Nr < (1ull << 32);
}
extern "C" int vqa_relation_linear_split_supported(int B, int N, int D, int L, float p_drop) {
  return relation_linear_ok(B, N, D, L, p_drop) ? 1 : 0;
}

extern "C" int vqa_relation_linear_fwd_split(const float* v, const float* t, const float* c2, const float* w, const float* bias,
                                             float* y, void* workspace, size_t workspace_bytes, int B, int N, int D, int L,
                                             int act, float p_drop, uint64_t seed, const uint64_t* seed_ptr, vqa_stream_t stream) {
  VQA_REQUIRE(v && t && c2 && w && y && workspace, VQA_E_BADARG, "relation_linear_fwd_split: null pointer");
  VQA_REQUIRE(act == 0 || act == 1, VQA_E_BADARG, "relation_linear_fwd_split: act must be 0 (none) or 1 (relu), got %d", act);
  VQA_REQUIRE(relation_linear_ok(B, N, D, L, p_drop), VQA_E_UNSUPPORTED,
              "relation_linear_fwd_split: shape outside the fused form (B=%d N=%d D=%d L=%d p=%f); see vqa_relation_linear_split_supported",
              B, N, D, L, (double)p_drop);
  VQA_REQUIRE(aligned(v, 16) && aligned(t, 16) && aligned(c2, 16) && aligned(w, 16) && aligned(y, 8) && aligned(workspace, 16),
              VQA_E_UNSUPPORTED, "relation_linear_fwd_split: v, t, c2, w, workspace must be 16-byte aligned");
  VQA_REQUIRE(workspace_bytes >= vqa_linear_act_fwd_split_workspace_bytes(D, L), VQA_E_BADARG,
              "relation_linear_fwd_split: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  const int M = B * N, K = D;
  sp::u32x4* wp = static_cast<sp::u32x4*>(workspace);
  {
    const long threads = (long)((L + 15) / 16) * (K / sp::kChunk) * 64;
    VQA_LAUNCH((sp::split_pack_kernel<false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w, K, L, K, wp);
  }
  using S = rt::NtShape<9, 5, 1, 2, 2>;
  const int tiles_m = (M + S::BM - 1) / S::BM, tiles_n = (L + S::BN - 1) / S::BN;
  sp::NtArgs a{v, wp, K, M, L, K, tiles_n, w, K};
  a.T = t;
  a.C2 = c2;
  a.rps = N;
  const SplitEpiBiasAct epi{y, bias, L, act, dc.p8 > 0 ? dc.scale : 1.f};
  const size_t rel = sp::rel_lds_bytes(S::BM, N, K), lds = rel > S::kLdsBytes ? rel : S::kLdsBytes;
  if (dc.p8 > 0) {
    VQA_ENSURE_LDS((sp::gemm_nt_kernel<9, 5, 1, 2, 2, true, SplitEpiBiasAct, 0, 3, false, true>), lds);
    VQA_LAUNCH((sp::gemm_nt_kernel<9, 5, 1, 2, 2, true, SplitEpiBiasAct, 0, 3, false, true>), dim3(tiles_m * tiles_n),
               dim3(sp::kThreads), lds, s, a, dc, epi);
  } else {
    VQA_ENSURE_LDS((sp::gemm_nt_kernel<9, 5, 1, 2, 2, false, SplitEpiBiasAct, 0, 3, false, true>), lds);
    VQA_LAUNCH((sp::gemm_nt_kernel<9, 5, 1, 2, 2, false, SplitEpiBiasAct, 0, 3, false, true>), dim3(tiles_m * tiles_n),
               dim3(sp::kThreads), lds, s, a, dc, epi);
  }
  return check_launch("relation_linear_fwd_split");
}

extern "C" size_t vqa_linear_act_dw_split_workspace_bytes(int M, int K, int N) {
  const sp::TnPlan pl = sp::tn_plan(M, tn_slabs());
  return round256(sp::packed_tn_bytes(pl.slabs, pl.cps, N)) + round256((size_t)pl.slabs * N * K * 4) +
         round256((size_t)pl.slabs * sp::kPackParts * N * 4);
}

static int linear_dw_split_impl(const float* x, int ldx, const float* y, const float* gy, float* d_w, float* d_b,
                                float* gz_out, void* workspace, size_t workspace_bytes, int M, int K, int N, int act, float p_drop,
                                uint64_t seed, const uint64_t* seed_ptr, vqa_stream_t stream, const float* rel_t, const float* rel_c2,
                                int rps) {
  VQA_REQUIRE(x && gy && d_w && workspace, VQA_E_BADARG, "linear_act_dw_split: null pointer");
  VQA_REQUIRE(act == 0 || (act == 1 && y != nullptr), VQA_E_BADARG, "linear_act_dw_split: act = 1 needs the forward output y");
  VQA_REQUIRE(gz_out == nullptr || (act == 1 && aligned(gz_out, 8)), VQA_E_BADARG,
              "linear_act_dw_split: gz_out (the gated gradient) only with act = 1, 8-byte aligned");
  VQA_REQUIRE(split_shape_ok(M, K, N, ldx, p_drop) && K % 128 == 0, VQA_E_UNSUPPORTED,
              "linear_act_dw_split: shape outside the split engine (M=%d K=%d N=%d ldx=%d p=%f)", M, K, N, ldx, (double)p_drop);
  VQA_REQUIRE(aligned(x, 16) && aligned(d_w, 16) && aligned(workspace, 16) && aligned(gy, 8) && (y == nullptr || aligned(y, 8)),
              VQA_E_UNSUPPORTED, "linear_act_dw_split: x, d_w, workspace must be 16-byte aligned, gy and y 8-byte");
  VQA_REQUIRE(workspace_bytes >= vqa_linear_act_dw_split_workspace_bytes(M, K, N), VQA_E_BADARG,
              "linear_act_dw_split: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  const sp::TnPlan pl = sp::tn_plan(M, tn_slabs());
  const int nblocks = (N + 15) / 16;
  char* base = static_cast<char*>(workspace);
  sp::u32x4* gp = reinterpret_cast<sp::u32x4*>(base);
  float* slab = reinterpret_cast<float*>(base + round256(sp::packed_tn_bytes(pl.slabs, pl.cps, N)));
  float* dbslab = slab + round256((size_t)pl.slabs * N * K * 4) / 4;
  {
    const dim3 pgrid(pl.slabs * sp::kPackParts);
    const size_t lds = sp::pack_tn_lds_bytes(nblocks);
    if (act == 1 && gz_out != nullptr)
      VQA_LAUNCH((sp::pack_tn_kernel<true, true>), pgrid, dim3(256), lds, s, gy, y, N, M, N, nblocks, pl.cps, gp, dbslab, gz_out, nblocks);
    else if (act == 1)
      VQA_LAUNCH((sp::pack_tn_kernel<true, false>), pgrid, dim3(256), lds, s, gy, y, N, M, N, nblocks, pl.cps, gp, dbslab, (float*)nullptr, nblocks);
    else
      VQA_LAUNCH((sp::pack_tn_kernel<false, false>), pgrid, dim3(256), lds, s, gy, y, N, M, N, nblocks, pl.cps, gp, dbslab, (float*)nullptr, nblocks);
  }
  constexpr int NA = 5, SPN = 2;
  const int tiles1 = (nblocks + 4 * NA - 1) / (4 * NA), tiles2 = (K + 64 * SPN - 1) / (64 * SPN);
  sp::TnArgs a{gp, x, slab, ldx, M, N, K, nblocks, pl.cps, tiles1, tiles2, gy, act == 1 ? y : nullptr, N};
  const dim3 grid(tiles1 * tiles2 * pl.slabs);
  if (rel_t != nullptr) {      // the layer input is t + c2 * x, recomputed while x is staged (always the shared-split kernel)
    a.T = rel_t;
    a.C2 = rel_c2;
    a.rps = rps;
    a.rps_magic = sp::rps_magic_of(rps);
    if (dc.p8 > 0)
      VQA_LAUNCH((sp::gemm_tn_shared_kernel<NA, true, 0, true>), grid, dim3(sp::kThreads), sp::kTnSharedLds, s, a, dc);
    else
      VQA_LAUNCH((sp::gemm_tn_shared_kernel<NA, false, 0, true>), grid, dim3(sp::kThreads), sp::kTnSharedLds, s, a, dc);
  } else
  // the split of x shared by the workgroup's four waves through LDS (91.9 / 94.0 us against 99 / 107 with every wave splitting
  // all of it, tools/split_probe.hip); VQA_SPLIT_TN_SHARED=0 keeps the per-wave form
  if (vqa::option_is("VQA_SPLIT_TN_SHARED", '0')) {
    if (dc.p8 > 0)
      VQA_LAUNCH((sp::gemm_tn_kernel<NA, SPN, true>), grid, dim3(sp::kThreads), 0, s, a, dc);
    else
      VQA_LAUNCH((sp::gemm_tn_kernel<NA, SPN, false>), grid, dim3(sp::kThreads), 0, s, a, dc);
  } else {
    if (dc.p8 > 0)
      VQA_LAUNCH((sp::gemm_tn_shared_kernel<NA, true>), grid, dim3(sp::kThreads), sp::kTnSharedLds, s, a, dc);
    else
      VQA_LAUNCH((sp::gemm_tn_shared_kernel<NA, false>), grid, dim3(sp::kThreads), sp::kTnSharedLds, s, a, dc);
  }
  const int NK = N * K;
  VQA_LAUNCH((sp::slab_sum_kernel), dim3(sp::slab_sum_blocks(NK, N)), dim3(256), 0, s, slab, dbslab, d_w, d_b, NK, N, pl.slabs,
             pl.slabs * sp::kPackParts, dc.p8 > 0 ? dc.scale : 1.f);
  return check_launch("linear_act_dw_split");
}


extern "C" int vqa_linear_act_dw_split(const float* x, int ldx, const float* y, const float* gy, float* d_w, float* d_b,
                                       float* gz_out, void* workspace, size_t workspace_bytes, int M, int K, int N, int act, float p_drop,
                                       uint64_t seed, const uint64_t* seed_ptr, vqa_stream_t stream) {
  return linear_dw_split_impl(x, ldx, y, gy, d_w, d_b, gz_out, workspace, workspace_bytes, M, K, N, act, p_drop, seed, seed_ptr, stream,
                              nullptr, nullptr, 0);
}

extern "C" int vqa_relation_linear_dw_split(const float* v, const float* t, const float* c2, const float* y, const float* gy,
                                            float* d_w, float* d_b, float* gz_out, void* workspace, size_t workspace_bytes, int B,
                                            int N, int D, int L, int act, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                            vqa_stream_t stream) {
  VQA_REQUIRE(v && t && c2, VQA_E_BADARG, "relation_linear_dw_split: null pointer");
  VQA_REQUIRE(relation_linear_ok(B, N, D, L, p_drop), VQA_E_UNSUPPORTED,
              "relation_linear_dw_split: shape outside the fused form (B=%d N=%d D=%d L=%d p=%f)", B, N, D, L, (double)p_drop);
  VQA_REQUIRE(aligned(t, 16) && aligned(c2, 16), VQA_E_UNSUPPORTED, "relation_linear_dw_split: t, c2 must be 16-byte aligned");
  return linear_dw_split_impl(v, D, y, gy, d_w, d_b, gz_out, workspace, workspace_bytes, B * N, D, L, act, p_drop, seed, seed_ptr,
                              stream, t, c2, N);
}
