// K1 -> K5 fusion, backward, on the split engine (gemm_f32_split.hpp): relation_dgrad.hip's contract -- the data gradient of
// compress_v2 (config/CoR2.py:218 on the relation tensor of :191-199,:216) reduced in the GEMM tile to d_t and d_c2, never a
// tensor -- with the fp32 products formed from exact three-way bf16 splits on the bf16 matrix pipe.
//
//     dx[m,:] = sum_l gz[m,l] W[l,:]      d_t[b,:] = sum_n keep(m,:) dx[m,:]      d_c2[b,:] = sum_n keep(m,:) dx[m,:] v[m,:]
//
// Tile = 144 rows (FOUR WHOLE SAMPLES of 36 regions, 9 row blocks) x 64 columns per wave, 256 columns per workgroup, as in
// relation_dgrad.hip; the main loop is gemm_f32_split.hpp's NT loop (`nt_accumulate`): A = gz, split in registers (rows of L
// floats; the contraction is padded to a whole, even number of 32-deep chunks -- what the loads pick up past a row's end meets
// zero planes), B = W^T as a packed plane image written once per launch (`pack_wt_kernel`) in the column order the epilogue
// wants: block e of a 64-column group holds the columns {4 r + e}, so a lane ends up with four CONSECUTIVE columns of four rows
// per row block -- one 16-byte load of v per row, one 16-byte store of d_t / d_c2 per sample.  One workgroup per CU (the
// accumulators, two plane sets of W^T and the ring of gz rows need ~300 registers); v is read in the epilogue.
#include <cstdlib>

#include "gemm_f32_split.hpp"

namespace vqa {
namespace {

constexpr int kRegions = 36, kRB = 9, kSamples = 4, kBM = 16 * kRB;

struct RelSplitArgs {
  const float* gz;      // [M, L]
  const sp::u32x4* wp;  // packed W^T: [D / 16][Kp / 32][3][64] x 16 bytes
  const float* v;       // [M, D]
  float* d_t;           // [B, D]
  float* d_c2;          // [B, D]
  int B, M, L, D;
  int Kp;               // L rounded up to an even number of 32-deep chunks
  int tiles_n;          // workgroup tiles of 256 columns
  int v_nt = 0;         // v loads with the nt (streaming) cache policy
  int col_major = 0;    // tile order (VQA_SPLIT_DGRAD_ORDER=col): an XCD owns ONE column tile (an eighth of the W^T image stays in
                        // its L2, gz streams through every XCD) instead of 16 row tiles x all column tiles
  const float* wf;      // W [L, D] itself: operand of the repair path (gemm_f32_split.hpp, any_nonfinite)
};

// W [L, D] -> the plane image of W^T.  Image row block cb = 4 (d / 64) + e, lane (r, g)  <->  column d = 64 (cb / 4) + 4 r + e,
// contraction indices l = 32 c + 4 g + {0..3} and 32 c + 16 + 4 g + {0..3} (zero past L).  One thread per (cb, chunk, lane).
__global__ __launch_bounds__(256) void pack_wt_kernel(const float* __restrict__ w, int L, int D, int chunks,
                                                      sp::u32x4* __restrict__ out) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)(D / 16) * chunks * 64) return;
  const int lane = (int)(t & 63), r = lane & 15, g = lane >> 4;
  const long bc = t >> 6;
  const int c = (int)(bc % chunks), cb = (int)(bc / chunks);
  const int d = 64 * (cb / 4) + 4 * r + (cb % 4);
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int l = 32 * c + 4 * g + (j & 3) + 16 * (j >> 2);
    v[j] = l < L ? w[(size_t)l * D + d] : 0.f;
  }
  uint32_t pw[3][4];
#pragma unroll
  for (int jj = 0; jj < 4; ++jj) sp::split_pair<false>(sp::f32x2{v[2 * jj], v[2 * jj + 1]}, pw[0][jj], pw[1][jj], pw[2][jj]);
  sp::u32x4* dst = out + (size_t)bc * 192 + lane;
#pragma unroll
  for (int q = 0; q < 3; ++q) dst[64 * q] = sp::u32x4{pw[q][0], pw[q][1], pw[q][2], pw[q][3]};
}

constexpr int kSharedLds = 2 * kRB * 3 * 1024;
// TUNE (ablations, VQA_SPLIT_DGRAD_TUNE, tools/dgrad_split_ablate.py; wrong results): 1 = no main loop, 2 = v read as ones (no v
// loads), 4 = no mask / no multiply / no per-sample sums (the accumulators are added up and stored)
template <bool SHARED, int TUNE = 0>
__global__ __launch_bounds__(sp::kThreads, 1) void relation_dgrad_split_kernel(RelSplitArgs p, DropCfg dc) {
  using rt::f32x4;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = xcd_remap(blockIdx.x, gridDim.x);   // the column tiles of a row tile are neighbours: gz rows come from L2
  const int tiles_m_ = gridDim.x / p.tiles_n;
  const int m0 = (p.col_major ? tile % tiles_m_ : tile / p.tiles_n) * kBM;
  const int n0 = (p.col_major ? tile / tiles_m_ : tile % p.tiles_n) * 256 + 64 * wave;
  if (!SHARED && n0 >= p.D) return;          // (SHARED: D % 256 == 0 is required, every wave takes part in the barriers)
  const int M = p.M, D = p.D;

  // (the descriptor of v is made HERE, ahead of every divergent region: made behind the repair branch below it became a PHI of
  //  a divergent region, lived in VGPRs, and each of the epilogue's 36 loads of v ran inside a waterfall loop: +50 us)
  const rt::rsrc_t Vb = rt::make_rsrc(p.v, (size_t)M * D * 4);
  f32x4 acc[kRB][4];   // [row block][column class e]: rows 16 i + 4 g + t (register t), column n0 + 4 r + e
#pragma unroll
  for (int i = 0; i < kRB; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const sp::NtArgs a{p.gz, p.wp, p.L, M, D, p.Kp, p.tiles_n};
    if constexpr ((TUNE & 1) != 0) {
      acc[0][0] = f32x4{(float)m0, (float)n0, 1.f, 2.f};
    } else if constexpr (SHARED) {
      // the four waves work on the same rows of gz: its split is shared through LDS (every wave splits a quarter of the row blocks)
      extern __shared__ __attribute__((aligned(16))) char rd_smem[];
      sp::nt_accumulate_shared<kRB, 4, 4, false, 0>(a, dc, (size_t)M * p.L * 4, m0, n0, 0, p.Kp / sp::kChunk, wave, rd_smem, acc);
    } else {
      sp::nt_accumulate<kRB, 4, false, 0, 3>(a, dc, (size_t)M * p.L * 4, m0, n0, 0, p.Kp / sp::kChunk, acc);
    }
  }

  // ---- outside the split's domain (a non-finite accumulator: gemm_f32_split.hpp, any_nonfinite): the lane's 144 values of dx
  // again as fp32 dot products of the original operands (this also undoes what a non-finite value in the NEXT row's first
  // Kp - L elements did through the contraction's padding)
  if constexpr (TUNE == 0) {
    uint32_t top = 0;     // the largest |bits| among the lane's accumulators (straight-line code: no short-circuit branches)
#pragma unroll
    for (int i = 0; i < kRB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) top = max(top, sp::abs_bits_max(acc[i][e]));
    if (top >= 0x7F800000u) {
#pragma unroll 1
      for (int it = 0; it < 4 * kRB; ++it) {       // (one rolled loop; the results go home through a select chain: cold code, kept small)
        const float* grow = p.gz + (size_t)min(m0 + 16 * (it >> 2) + 4 * g + (it & 3), M - 1) * p.L;
        const float* wcol = p.wf + n0 + 4 * r;
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < p.L; ++l) {
          const float gv = grow[l];
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wcol + (size_t)l * D);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] = fmaf(gv, wv[e], s[e]);
        }
#pragma unroll
        for (int i = 0; i < kRB; ++i)
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][e][t] = it == 4 * i + t ? s[e] : acc[i][e][t];
      }
    }
  }

  // ---- epilogue (relation_dgrad.hip's): mask, multiply by v, add the rows of each sample up ----
  if constexpr ((TUNE & 4) != 0) {
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kRB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += acc[i][e];
    if (g == 0) *reinterpret_cast<f32x4*>(p.d_t + (size_t)(m0 / kRegions) * D + n0 + 4 * r) = sum;
    return;
  }
  auto vt = [&](int i, int t) -> f32x4 {
    if constexpr ((TUNE & 2) != 0) return f32x4{1.f, 1.f, 1.f, 1.f};
    const int row = min(m0 + 16 * i + 4 * g + t, M - 1);
    const uint32_t off = (uint32_t)(n0 + 4 * r) * 4u + (uint32_t)row * (uint32_t)D * 4u;
    // v is read exactly once, at the end of a workgroup's life: with the streaming policy its 4.7 MB per round of workgroups do not
    // push the W^T image (3.9 MB, wanted by every workgroup) out of the XCD's 4 MB L2 (VQA_SPLIT_DGRAD_VNT, round 6)
    if (p.v_nt) return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(Vb, (int)off, 0, 2));
    return rt::ldg16(Vb, off, 0u);
  };
  const uint32_t key = dc.p8 > 0 ? drop_key(dc) : 0u;
  if (m0 + kBM > M) {                        // last, partial tile: the rows beyond M (clamped duplicates) count as zero
#pragma unroll
    for (int i = 0; i < kRB; ++i)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (m0 + 16 * i + 4 * g + t >= M) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][e][t] = 0.f;
        }
  }
  f32x4 st[kSamples], sc[kSamples];   // per sample: sum keep dx, sum keep dx v  (components = the 4 columns of the lane)
#pragma unroll
  for (int s = 0; s < kSamples; ++s) st[s] = sc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto reduce = [&](auto mode) {      // MODE: 0 no dropout, 1 the one-bit p = 0.5 mask, 2 a byte per element
    constexpr int MODE = decltype(mode)::value;
#pragma unroll
    for (int i = 0; i < kRB; ++i) {
      f32x4 xt = f32x4{0.f, 0.f, 0.f, 0.f}, xc = f32x4{0.f, 0.f, 0.f, 0.f};   // the 4 rows of group 4 i + g, one sample
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x4 x = f32x4{acc[i][0][t], acc[i][1][t], acc[i][2][t], acc[i][3][t]};
        if constexpr (MODE != 0) {
          const int m = min(m0 + 16 * i + 4 * g + t, M - 1);
          const uint32_t e = (uint32_t)m * (uint32_t)D + (uint32_t)(n0 + 4 * r);   // a multiple of 4
          if constexpr (MODE == 1) {         // kept values are scaled by 2 at the end
            const uint32_t w = mask_word32(e >> 5, key) >> (e & 31u);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float xe = x[c];
              x[c] = __uint_as_float(__float_as_uint(xe) & (0u - ((w >> c) & 1u)));
            }
          } else {
            const uint32_t w = mask_word32(e >> 2, key);
            x *= f32x4{(w & 255u) >= dc.p8 ? dc.scale : 0.f, ((w >> 8) & 255u) >= dc.p8 ? dc.scale : 0.f,
                       ((w >> 16) & 255u) >= dc.p8 ? dc.scale : 0.f, (w >> 24) >= dc.p8 ? dc.scale : 0.f};
          }
        }
        xt += x;
        xc += x * vt(i, t);
      }
      // group q = 4 i + g belongs to sample q / 9; for a fixed row block that is one of at most two samples
      constexpr int kGroups = kRegions / 4;
      const int s_lo = (4 * i) / kGroups, s_hi = (4 * i + 3) / kGroups;
      if (s_lo == s_hi) {
        st[s_lo] += xt;
        sc[s_lo] += xc;
      } else {
        const bool hi = (4 * i + g) / kGroups == s_hi;
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
        st[s_lo] += hi ? z : xt;
        sc[s_lo] += hi ? z : xc;
        st[s_hi] += hi ? xt : z;
        sc[s_hi] += hi ? xc : z;
      }
    }
  };
  if (dc.p8 == 0) reduce(std::integral_constant<int, 0>{});
  else if (dc.p8 == kDropHalf) reduce(std::integral_constant<int, 1>{});
  else reduce(std::integral_constant<int, 2>{});
  const float post = dc.p8 == kDropHalf ? 2.f : 1.f;
  const int b0 = m0 / kRegions;
#pragma unroll
  for (int s = 0; s < kSamples; ++s) {
    f32x4 a, b;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a[e] = rows_sum(st[s][e]) * post;
      b[e] = rows_sum(sc[s][e]) * post;
    }
    if (g == 0 && b0 + s < p.B) {
      *reinterpret_cast<f32x4*>(p.d_t + (size_t)(b0 + s) * D + n0 + 4 * r) = a;
      *reinterpret_cast<f32x4*>(p.d_c2 + (size_t)(b0 + s) * D + n0 + 4 * r) = b;
    }
  }
}

int padded_k(int L) { return (((L + sp::kChunk - 1) / sp::kChunk + 1) & ~1) * sp::kChunk; }

}  // namespace
}  // namespace vqa

using namespace vqa;

// include/vqa_mi355x.h
extern "C" int vqa_relation_projection_dgrad_split_supported(int B, int N, int D, int L) {
  return N == kRegions && B >= 1 && D % 64 == 0 && D >= 64 && L >= 32 && L % 2 == 0 && (size_t)B * N * D * 4 < (1ull << 32) &&
         sp::packed_bytes(D, padded_k(L)) < (1ull << 32) && (size_t)B * N * L * 4 < (1ull << 32);
}

extern "C" size_t vqa_relation_projection_dgrad_split_workspace_bytes(int D, int L) {
  return (sp::packed_bytes(D, padded_k(L)) + 255) & ~(size_t)255;
}

extern "C" int vqa_relation_projection_dgrad_split(const float* gz, const float* w, const float* v, float* d_t, float* d_c2,
                                                   void* workspace, size_t workspace_bytes, float p_drop, uint64_t seed,
                                                   const uint64_t* seed_ptr, int B, int N, int D, int L, vqa_stream_t stream) {
  VQA_REQUIRE(gz && w && v && d_t && d_c2 && workspace, VQA_E_BADARG, "relation_projection_dgrad_split: null pointer");
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "relation_projection_dgrad_split: p_drop=%f outside [0,1)", (double)p_drop);
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && L > 0, VQA_E_BADARG, "relation_projection_dgrad_split: bad sizes B=%d N=%d D=%d L=%d", B, N, D, L);
  VQA_REQUIRE(vqa_relation_projection_dgrad_split_supported(B, N, D, L), VQA_E_UNSUPPORTED,
              "relation_projection_dgrad_split: needs N = 36 regions, D %% 64 == 0, even L >= 32 (N=%d D=%d L=%d)", N, D, L);
  VQA_REQUIRE(aligned(w, 16) && aligned(v, 16) && aligned(d_t, 16) && aligned(d_c2, 16) && aligned(gz, 8) && aligned(workspace, 16),
              VQA_E_UNSUPPORTED, "relation_projection_dgrad_split: w, v, d_t, d_c2, workspace must be 16-byte aligned, gz 8-byte");
  VQA_REQUIRE(workspace_bytes >= vqa_relation_projection_dgrad_split_workspace_bytes(D, L), VQA_E_BADARG,
              "relation_projection_dgrad_split: workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int Kp = padded_k(L), chunks = Kp / sp::kChunk;
  sp::u32x4* wp = static_cast<sp::u32x4*>(workspace);
  {
    const long threads = (long)(D / 16) * chunks * 64;
    VQA_LAUNCH(pack_wt_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, w, L, D, chunks, wp);
  }
  RelSplitArgs a{};
  a.gz = gz;
  a.wp = wp;
  a.v = v;
  a.d_t = d_t;
  a.d_c2 = d_c2;
  a.B = B;
  a.M = B * N;
  a.L = L;
  a.D = D;
  a.Kp = Kp;
  a.tiles_n = (D + 255) / 256;
  a.wf = w;
  const int tiles_m = (a.M + kBM - 1) / kBM;
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  // D % 256 == 0: gz's split shared by the workgroup's four waves through LDS (VQA_SPLIT_DGRAD_SHARED=0: every wave splits all of it)
  a.col_major = vqa::option_is("VQA_SPLIT_DGRAD_ORDER", 'c') ? 1 : 0;
  a.v_nt = vqa::option_is("VQA_SPLIT_DGRAD_VNT", '1') ? 1 : 0;
  const int tune = vqa::option("VQA_SPLIT_DGRAD_TUNE") ? std::atoi(vqa::option("VQA_SPLIT_DGRAD_TUNE")) : 0;
  const dim3 grid_((unsigned)(tiles_m * a.tiles_n));
  if (tune == 1 && D % 256 == 0) VQA_LAUNCH((relation_dgrad_split_kernel<true, 1>), grid_, dim3(sp::kThreads), kSharedLds, s, a, dc);
  else if (tune == 2 && D % 256 == 0) VQA_LAUNCH((relation_dgrad_split_kernel<true, 2>), grid_, dim3(sp::kThreads), kSharedLds, s, a, dc);
  else if (tune == 4 && D % 256 == 0) VQA_LAUNCH((relation_dgrad_split_kernel<true, 4>), grid_, dim3(sp::kThreads), kSharedLds, s, a, dc);
  else if (tune == 5 && D % 256 == 0) VQA_LAUNCH((relation_dgrad_split_kernel<true, 5>), grid_, dim3(sp::kThreads), kSharedLds, s, a, dc);
  else if (D % 256 == 0 && !vqa::option_is("VQA_SPLIT_DGRAD_SHARED", '0'))
    VQA_LAUNCH(relation_dgrad_split_kernel<true>, dim3((unsigned)(tiles_m * a.tiles_n)), dim3(sp::kThreads), kSharedLds, s, a, dc);
  else
    VQA_LAUNCH(relation_dgrad_split_kernel<false>, dim3((unsigned)(tiles_m * a.tiles_n)), dim3(sp::kThreads), 0, s, a, dc);
  return check_launch("relation_projection_dgrad_split");
}
