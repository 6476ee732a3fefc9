// Host-side interface between bilinear_folded.hip (rank-folded forward / data gradient on v_mfma_f32_16x16x4_f32) and
// bilinear_fusion.hip (the per-sample weight-gradient kernel on the 32x32 tile engine).
#pragma once
#include "common.hpp"

namespace vqa {

constexpr int kFoldMaxR = 4;
constexpr int kFoldMaxN = 112;

bool folded_supported(int B, int N, int L, int H, int R);

// dx[b,n,:] = Weff_b^T g[b,n,:]   (w1t: R dense [L,H] transposes of W1_r in device memory); gate (optional, [B*N,L] like
// d_x): d_x is zeroed where gate <= 0
int folded_data_gradient(const float* g, const float* const* w1t, const float* h2, float* d_x, int B, int N, int L, int H,
                         int R, hipStream_t s, const float* gate = nullptr);

// wt[r] = w1[r]^T   ([H,L] -> [L,H]), all ranks in one launch
int folded_transpose_weights(const float* const* w1, float* wt, int L, int H, int R, hipStream_t s);

// Register-tile weight-gradient kernel of the folded backward (bilinear_dw_rt.hip): dW1_r / db1_r slabs over kDwRtGroups
// sample groups (layout [group][R][H*L] / [group][R][H], summed by bilinear_dw_reduce_kernel) and two partial sums of dh2
// (layout [2][B*R*H], summed by bilinear_dh2_reduce_kernel).
constexpr int kDwRtGroups = 8;
bool dw_rt_supported(int B, int N, int L, int H, int R, int ldx);
int dw_rt_launch(const float* g, const float* x, const float* h2, const float* const* w1, const float* const* b1, float* slab,
                 float* dbslab, float* part, int B, int N, int L, int H, int R, hipStream_t s);

// The same product on the split engine (bilinear_dw_split.hip: six bf16 partial products per fp32 product, both operands split
// while they are staged): same outputs, kDwSplitSlabs sample slabs.  Opt-in (VQA_K4_DW_SPLIT=1): measured slower than the fp32
// register-tile form (150 against 86 us at B = 512).
constexpr int kDwSplitSlabs = 16;
bool dw_split_supported(int B, int N, int L, int H, int R, int ldx);
int dw_split_launch(const float* g, const float* x, const float* h2, const float* const* w1, const float* const* b1, float* slab,
                    float* dbslab, float* part, int B, int N, int L, int H, int R, hipStream_t s);

// Register-tile forms of the folded forward and data gradient (bilinear_rt.hip); K = contraction length, NO = output rows
bool fold_rt_supported(int B, int N, int K, int NO, int R, int ldx, int ldw, int ldo);
int fold_rt_forward(const float* x, int ldx, const float* const* w1, const float* const* b1, const float* h2, float* out, int B,
                    int N, int L, int H, int R, hipStream_t s);
int fold_rt_data_gradient(const float* g, const float* const* w1t, const float* h2, float* d_x, int B, int N, int L, int H, int R,
                          hipStream_t s);

}  // namespace vqa
