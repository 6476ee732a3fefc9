// K4, rank-folded form -- the low-rank bilinear (Mutan) fusion with the sum over ranks moved INTO the weights.
//
// putils.MutanFusion.forward (putils/__init__.py:232-238) computes, per sample b and region n,
//     out[b,n,:] = sum_r (W1_r x[b,n,:] + b1_r) * h2[b,r,:]
// which the tile-engine kernels of bilinear_fusion.hip evaluate as R dense GEMMs over all B*N rows.  Because the
// question-side factor h2[b,r,:] is the same for every region of a sample, the sum over ranks commutes with the
// contraction:
//     out[b,n,:] = Weff_b x[b,n,:] + c_b,     Weff_b[j,k] = sum_r h2[b,r,j] W1_r[j,k],   c_b[j] = sum_r h2[b,r,j] b1_r[j]
// ONE contraction per sample against a weight that is built on the fly (R fused multiply-adds per weight element
// while the A fragment is on its way from LDS to the matrix core) -- 1/R of the matrix work, and the [B*N,R,H]
// intermediate h1 is never formed.  The data gradient folds the same way,
//     dx[b,n,:] = Weff_b^T g[b,n,:],
// and so do dh2 / dW1 / db1 through P_b = g_b^T x_b (bilinear_fusion.hip, per-sample weight-gradient kernel).
//
// Kernel shape: a workgroup = 4 waves = 4 consecutive samples x 64 output features; every wave owns one sample
// (all its regions: NB blocks of 16) and 4 blocks of 16 features, on v_mfma_f32_16x16x4_f32 (a sample has 36..100
// regions: 16-wide blocks waste 25 % / 11 % of the matrix core where 32-wide ones would waste 44 % / 22 %).
// The W1 tile [R][64][CK] is shared by the four waves, each stages its own x rows; both sit in LDS with a pitch of
// CK+4 = 20 floats, so the 16-byte fragment reads (16 rows x 4 quads per wave) are bank-conflict free.  A lane's four
// consecutive contraction indices feed four successive MFMAs (the k order inside a dot product is free).
//
// Where the time goes at B = 512, N = 36, R = 2 (forward, 88 us alone): ablation builds put ~60 us in the MFMA-paced
// chunk loop (within 10 % of the matrix core's rate for the padded work), ~12 us in staging (global -> LDS, barrier) and
// ~16 us in launch, prologue and the output stores of a grid that runs as ONE round of resident workgroups.  Tried and
// dropped: 32-column chunks (half the barriers, one workgroup less per CU: 104 us), A fragments fetched a feature block
// ahead in the source (30 more registers, one wave less per SIMD: 119 us), two register sets in flight (the x / g rows
// do not wait on HBM: no gain, one wave less per SIMD), a float2 select on the way into LDS (hipcc lowers it through
// scratch memory: +14 us on the data gradient -- the selects are component-wise for that reason).
#include <cstdlib>

#include "bilinear_folded.hpp"

namespace vqa {

#ifndef VQA_FOLD_CK
#define VQA_FOLD_CK 16
#endif
constexpr int kFoldCK = VQA_FOLD_CK;   // contraction chunk per LDS stage (16 or 32: pitch 20 / 36 floats, both conflict free)
using f32x4 = __attribute__((ext_vector_type(4))) float;

struct FoldPtrs {
  const float* w[kFoldMaxR];
  const float* b[kFoldMaxR];
};

// in  [B*N][ld_in]   rows of the contraction side (forward: x, C = L; data gradient: g, C = H)
// w_r [O][ldw]       C-contiguous weights (forward: W1_r [H][L]; data gradient: W1_r^T [L][H])
// h2  [B][R][h2_dim] indexed by the output feature (H2_ROWS, forward) or by the contraction index (data gradient)
// out [B*N][ld_out]
// (second launch bound: at least two waves per SIMD, i.e. at most 256 registers per lane)
template <int NB, int R, bool H2_ROWS, int WAVES, int OB>
__global__ __launch_bounds__(64 * WAVES, 2) void bilinear_folded_kernel(const float* __restrict__ in, int ld_in, FoldPtrs wp,
                                                                        int ldw, const float* __restrict__ h2, int h2_dim,
                                                                        float* __restrict__ out, int ld_out, int B, int N,
                                                                        int C, int O, int tiles_o,
                                                                        const float* __restrict__ gate) {
  constexpr int CK = kFoldCK, P = CK + 4;   // contraction chunk and LDS pitch in floats
  constexpr int CK2 = CK / 2;          // float2 per staged row
  constexpr int T = 64 * WAVES;        // threads = one wave per sample of the group
  constexpr int OBR = OB * 16;         // output features per workgroup (OB blocks of 16 per wave)
  constexpr int WPT = (OBR * CK2 + T - 1) / T;   // float2 of one rank's W tile per thread (upper bound)
  constexpr int IPT = (WAVES * NB * 16 * CK2 + T - 1) / T;  // upper bound of the `in` float2 per thread
  static_assert(WPT <= 8 && IPT <= 24, "slot validity bits live in one 32-bit mask");
  static_assert(T % CK2 == 0 && (CK == 16 || CK == 32), "chunk of 16 or 32 columns");
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* smem = reinterpret_cast<float*>(smem_raw);
  const int w_floats = R * OBR * P, in_floats = WAVES * N * P, h2_floats = H2_ROWS ? 0 : WAVES * R * CK;
  const int stage_floats = w_floats + in_floats + h2_floats;

  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int o0 = (bid % tiles_o) * OBR, b0 = (bid / tiles_o) * WAVES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, quad = lane >> 4;
  const int rows_in = min(WAVES, B - b0) * N;  // valid staged rows of `in`
  // forward form: the bias rides along as contraction column C (W column = b1_r, x column = 1), free whenever C is not
  // a multiple of the chunk and one extra chunk otherwise
  const int chunks = ((H2_ROWS ? C + 1 : C) + CK - 1) / CK;
  const int b = min(b0 + wave, B - 1);

  // per-lane question-side factors of the forward form: h2[b][r][o] for the lane's A rows
  float h2v[R][OB];
  if (H2_ROWS) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) h2v[r][ob] = h2[((size_t)b * R + r) * h2_dim + min(o0 + ob * 16 + l16, O - 1)];
  }

  // staging slots of this thread: 32-bit offsets from wave-uniform bases (saddr + voffset loads).  Loads are
  // unconditional on clamped addresses (a branch per guarded load serialises them); out-of-range slots are zeroed
  // on their way into LDS.
  const int c2 = 2 * (tid % CK2);      // the same column pair for every slot (T % CK2 == 0)
  int woff[WPT], ioff[IPT];
  unsigned okmask = 0;                 // bit i: W slot i in range, bit 8+i: `in` slot i in range
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int row = (tid + i * T) / CK2;
    woff[i] = min(row, O - 1 - o0) * ldw;
    okmask |= (o0 + row < O && row < OBR ? 1u : 0u) << i;
  }
#pragma unroll
  for (int i = 0; i < IPT; ++i) {
    const int row = (tid + i * T) / CK2;
    ioff[i] = min(row, rows_in - 1) * ld_in;
    okmask |= (row < rows_in ? 1u : 0u) << (8 + i);
  }
  const float* __restrict__ in_base = in + (size_t)b0 * N * ld_in;
  const int lds_slot = (tid / CK2) * P + c2;   // slot i sits (T / CK2) * i rows further down
  const int hs_cc = tid % CK, hs_r = (tid / CK) % R, hs_s = min(tid / (CK * R), WAVES - 1);
  const float* __restrict__ h2_src = h2 + ((size_t)min(b0 + hs_s, B - 1) * R + hs_r) * h2_dim;

  float2 wreg[R][WPT], ireg[IPT];
  float hreg = 0.f;
  float breg[R][WPT];   // b1_r of this thread's W rows (forward form)
  if (H2_ROWS) {
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < WPT; ++i) breg[r][i] = wp.b[r][min(o0 + (tid + i * T) / CK2, O - 1)];
  }
  auto load_chunk = [&](int ch) {
    const int cc = min(ch * CK + c2, C - 2);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float* __restrict__ wb = wp.w[r] + (size_t)o0 * ldw;
#pragma unroll
      for (int i = 0; i < WPT; ++i) wreg[r][i] = ld2(wb + (unsigned)(woff[i] + cc));
    }
#pragma unroll
    for (int i = 0; i < IPT; ++i) ireg[i] = ld2(in_base + (unsigned)(ioff[i] + cc));
    if (!H2_ROWS && tid < WAVES * R * CK) hreg = h2_src[min(ch * CK + hs_cc, C - 1)];
  };
  auto store_chunk = [&](int stage, int ch) {
    float* __restrict__ ws = smem + stage * stage_floats;
    float* __restrict__ is = ws + w_floats;
    const bool cok = ch * CK + c2 < C;
    const bool bias_col = H2_ROWS && ch * CK + c2 == C;   // (C and c2 are even: the bias column is always an .x)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < WPT; ++i) {
        const bool ok = cok && (okmask >> i & 1);   // (component-wise: a float2 select is lowered through scratch)
        float2 t = make_float2(ok ? wreg[r][i].x : 0.f, ok ? wreg[r][i].y : 0.f);
        if (bias_col) t.x = breg[r][i];
        if (OBR * CK2 % T == 0 || (tid + i * T) / CK2 < OBR) st2(ws + r * OBR * P + lds_slot + i * (T / CK2) * P, t);
      }
#pragma unroll
    for (int i = 0; i < IPT; ++i)
      if ((tid + i * T) / CK2 < WAVES * N) {
        const bool ok = cok && (okmask >> (8 + i) & 1);
        float2 t = make_float2(ok ? ireg[i].x : 0.f, ok ? ireg[i].y : 0.f);
        if (bias_col) t.x = 1.f;
        st2(is + lds_slot + i * (T / CK2) * P, t);
      }
    if (!H2_ROWS && tid < WAVES * R * CK)
      is[in_floats + tid] = (b0 + hs_s < B && ch * CK + hs_cc < C) ? hreg : 0.f;   // [sample][r][CK]
  };

  f32x4 acc[OB][NB];
#pragma unroll
  for (int ob = 0; ob < OB; ++ob)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[ob][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // region row of this lane per block, clamped (columns past N are computed on a duplicate row and never stored)
  int brow[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) brow[nb] = (wave * N + min(nb * 16 + l16, N - 1)) * P + 4 * quad;
  const int arow = l16 * P + 4 * quad;

  load_chunk(0);
  store_chunk(0, 0);
  __syncthreads();
  for (int ch = 0; ch < chunks; ++ch) {
    const int cur = ch & 1;
    if (ch + 1 < chunks) load_chunk(ch + 1);
    const float* __restrict__ ws = smem + cur * stage_floats;
    const float* __restrict__ is = ws + w_floats;
#pragma unroll
    for (int ks = 0; ks < CK; ks += 16) {
      f32x4 bf[NB];
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) bf[nb] = *reinterpret_cast<const f32x4*>(is + brow[nb] + ks);
      f32x4 hc[R];
      if (!H2_ROWS) {
#pragma unroll
        for (int r = 0; r < R; ++r)
          hc[r] = *reinterpret_cast<const f32x4*>(is + in_floats + (wave * R + r) * CK + ks + 4 * quad);
      }
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) {
        const float* __restrict__ wrow = ws + ob * 16 * P + arow + ks;
        f32x4 a = (H2_ROWS ? f32x4{h2v[0][ob], h2v[0][ob], h2v[0][ob], h2v[0][ob]} : hc[0]) *
                  *reinterpret_cast<const f32x4*>(wrow);
#pragma unroll
        for (int r = 1; r < R; ++r) {
          const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + r * OBR * P);
          a += (H2_ROWS ? f32x4{h2v[r][ob], h2v[r][ob], h2v[r][ob], h2v[r][ob]} : hc[r]) * wv;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            acc[ob][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bf[nb][i], acc[ob][nb], 0, 0, 0);
      }
    }
    if (ch + 1 < chunks) store_chunk(cur ^ 1, ch + 1);
    __syncthreads();
  }

  // epilogue: lane holds out[b][n = nb*16 + l16][o = o0 + ob*16 + 4*quad + 0..3]
  if (b0 + wave >= B) return;
#pragma unroll
  for (int ob = 0; ob < OB; ++ob) {
    const int o = o0 + ob * 16 + 4 * quad;
    if (o >= O) continue;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = nb * 16 + l16;
      if (n < N) {
        const size_t at = ((size_t)(b0 + wave) * N + n) * ld_out + o;
        float* __restrict__ dst = out + at;
        f32x4 t = acc[ob][nb];
        if (gate != nullptr) {
          // data gradient towards a relu output: zero where the forward value (laid out like `out`) is not positive -- the
          // gate of the layer that produced it, applied here so that layer's backward kernels need no masked operand
          const float2 g0 = ld2(gate + at), g1 = o + 2 < O ? ld2(gate + at + 2) : make_float2(0.f, 0.f);
          t[0] = g0.x > 0.f ? t[0] : 0.f;
          t[1] = g0.y > 0.f ? t[1] : 0.f;
          t[2] = g1.x > 0.f ? t[2] : 0.f;
          t[3] = g1.y > 0.f ? t[3] : 0.f;
        }
        st2(dst, make_float2(t[0], t[1]));
        if (o + 2 < O) st2(dst + 2, make_float2(t[2], t[3]));
      }
    }
  }
}

static size_t folded_lds_bytes(int N, int R, bool h2_rows, int waves, int ob = 4) {
  return 2 * sizeof(float) *
         ((size_t)R * ob * 16 * (kFoldCK + 4) + (size_t)waves * N * (kFoldCK + 4) + (h2_rows ? 0 : waves * R * kFoldCK));
}

// Workgroup shape: `waves` samples (one wave each) x `ob` blocks of 16 output features.  The kernel is MFMA-paced, so a
// CU's time is the work of the workgroups dealt to it: pick the shape that minimises ceil(workgroups / 256) x work
// per workgroup (8 waves share a W tile between more waves, 4 waves / 5 blocks change how O and B divide), with a
// penalty when fewer than 8 waves would be resident per CU.
struct FoldCfg {
  int waves, ob;
};
static FoldCfg folded_config(int B, int N, int O, int R, bool h2_rows) {
  double best = 1e30;
  FoldCfg pick{4, 4};
  for (int ob : {4, 5}) {
    if (ob == 5 && (h2_rows || N > 48)) continue;   // 5 blocks: instantiated for the data gradient, <= 3 region blocks
    for (int waves : {8, 4}) {
      const size_t lds = folded_lds_bytes(N, R, h2_rows, waves, ob);
      if (lds > 160 * 1024) continue;
      long per_cu = (160 * 1024) / lds;
      const long by_regs = 16 / waves;
      if (per_cu > by_regs) per_cu = by_regs;
      const long wgs = (long)((O + ob * 16 - 1) / (ob * 16)) * ((B + waves - 1) / waves);
      const long per = (wgs + 255) / 256;
      double cost = (double)per * ob * waves;
      if ((per < per_cu ? per : per_cu) * waves < 8) cost *= 1.3;
      if (waves == 4) cost *= 1.05;   // measured: 4-wave groups re-read W twice as often ...
      if (waves == 8 && (per < per_cu ? per : per_cu) == 1) cost *= 1.1;   // ... a lone 8-wave group has nobody to overlap its barriers with
      if (cost < best) {
        best = cost;
        pick = FoldCfg{waves, ob};
      }
    }
  }
  const char* ew = vqa::option("VQA_K4_FOLD_WAVES");   // experiment knobs
  const char* eo = vqa::option("VQA_K4_FOLD_OB");
  FoldCfg forced = pick;
  if (ew != nullptr && (std::atoi(ew) == 4 || std::atoi(ew) == 8)) forced.waves = std::atoi(ew);
  if (eo != nullptr && (std::atoi(eo) == 4 || (std::atoi(eo) == 5 && !h2_rows && N <= 48))) forced.ob = std::atoi(eo);
  if (folded_lds_bytes(N, R, h2_rows, forced.waves, forced.ob) <= 160 * 1024) pick = forced;
  return pick;
}

template <bool H2_ROWS>
static int launch_folded(const char* who, const float* in, int ld_in, const FoldPtrs& wp, int ldw, const float* h2, int h2_dim,
                         float* out, int ld_out, int B, int N, int C, int O, int R, hipStream_t s, const float* gate = nullptr) {
  const FoldCfg cfg = folded_config(B, N, O, R, H2_ROWS);
  const int waves = cfg.waves, obr = cfg.ob * 16;
  const int tiles_o = (O + obr - 1) / obr, groups = (B + waves - 1) / waves;
  const size_t lds = folded_lds_bytes(N, R, H2_ROWS, waves, cfg.ob);
#define LAUNCH_C(NB_, R_, W_, OB_)                                                                                           \
  {                                                                                                                          \
    VQA_ENSURE_LDS((bilinear_folded_kernel<NB_, R_, H2_ROWS, W_, OB_>), lds);                                                \
    VQA_LAUNCH((bilinear_folded_kernel<NB_, R_, H2_ROWS, W_, OB_>), dim3(tiles_o * groups), dim3(64 * W_), lds, s,   \
                       in, ld_in, wp, ldw, h2, h2_dim, out, ld_out, B, N, C, O, tiles_o, gate);                              \
  }
#define LAUNCH_W(NB_, R_, OB_) \
  if (waves == 8) LAUNCH_C(NB_, R_, 8, OB_) else LAUNCH_C(NB_, R_, 4, OB_)
#define LAUNCH(NB_, R_)                                   \
  if constexpr (!H2_ROWS && NB_ <= 3) {                   \
    if (cfg.ob == 5) LAUNCH_W(NB_, R_, 5) else LAUNCH_W(NB_, R_, 4) \
  } else {                                                \
    LAUNCH_W(NB_, R_, 4)                                  \
  }
  // region blocks per sample: the smallest instantiated count that covers N (blocks past N run on a duplicate row).
  // The 5- and 7-block variants keep 80 / 112 accumulator registers per lane: ranks 3 and 4 would spill there.
#define LAUNCH_R4(NB_)                \
  switch (R) {                        \
    case 1: LAUNCH(NB_, 1) break;     \
    case 2: LAUNCH(NB_, 2) break;     \
    case 3: LAUNCH(NB_, 3) break;     \
    default: LAUNCH(NB_, 4) break;    \
  }
#define LAUNCH_R2(NB_) \
  if (R == 1) LAUNCH(NB_, 1) else LAUNCH(NB_, 2)
  if (N <= 16) {
    LAUNCH_R4(1)
  } else if (N <= 32) {
    LAUNCH_R4(2)
  } else if (N <= 48) {
    LAUNCH_R4(3)
  } else if (N <= 80) {
    LAUNCH_R2(5)
  } else {
    LAUNCH_R2(7)
  }
#undef LAUNCH_R4
#undef LAUNCH_R2
#undef LAUNCH_W
#undef LAUNCH_C
#undef LAUNCH
  return check_launch(who);
}

// wt[r][l][h] = w[r][h][l]: 32x32 tiles through LDS, grid (ceil(L/32), ceil(H/32), R)
__global__ __launch_bounds__(256) void fold_transpose_kernel(FoldPtrs wp, float* __restrict__ wt, int L, int H) {
  __shared__ float tile[32][33];
  const float* __restrict__ w = wp.w[blockIdx.z];
  float* __restrict__ dst = wt + (size_t)blockIdx.z * L * H;
  const int l0 = blockIdx.x * 32, h0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int h = h0 + ty + 8 * k, l = l0 + tx;
    tile[ty + 8 * k][tx] = (h < H && l < L) ? w[(size_t)h * L + l] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int l = l0 + ty + 8 * k, h = h0 + tx;
    if (l < L && h < H) dst[(size_t)l * H + h] = tile[tx][ty + 8 * k];
  }
}

bool folded_supported(int B, int N, int L, int H, int R) {
  return B > 0 && N > 0 && N <= kFoldMaxN && L > 0 && H > 0 && R > 0 && R <= (N <= 48 ? kFoldMaxR : 2) && L % 2 == 0 &&
         H % 2 == 0;
}

int folded_transpose_weights(const float* const* w1, float* wt, int L, int H, int R, hipStream_t s) {
  FoldPtrs wp{};
  for (int r = 0; r < R; ++r) wp.w[r] = w1[r];
  VQA_LAUNCH(fold_transpose_kernel, dim3((L + 31) / 32, (H + 31) / 32, R), dim3(256), 0, s, wp, wt, L, H);
  return check_launch("lowrank_bilinear_fusion_folded_bwd (weight transpose)");
}

int folded_data_gradient(const float* g, const float* const* w1t, const float* h2, float* d_x, int B, int N, int L, int H,
                         int R, hipStream_t s, const float* gate) {
  if (gate == nullptr && fold_rt_supported(B, N, H, L, R, H, H, L)) return fold_rt_data_gradient(g, w1t, h2, d_x, B, N, L, H, R, s);
  FoldPtrs wp{};
  for (int r = 0; r < R; ++r) wp.w[r] = w1t[r];
  return launch_folded<false>("lowrank_bilinear_fusion_folded_bwd (dx)", g, H, wp, H, h2, H, d_x, L, B, N, H, L, R, s, gate);
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_lowrank_bilinear_fusion_folded_supported(int B, int N, int L, int H, int R) {
  return folded_supported(B, N, L, H, R) ? 1 : 0;
}

extern "C" int vqa_lowrank_bilinear_fusion_folded_fwd(const float* x, int ldx, const float* const* w1,
                                                      const float* const* b1, const float* h2, float* out, int B, int N,
                                                      int L, int H, int R, vqa_stream_t stream) {
  VQA_REQUIRE(x && w1 && b1 && h2 && out, VQA_E_BADARG, "lowrank_bilinear_fusion_folded_fwd: null pointer");
  VQA_REQUIRE(vqa_lowrank_bilinear_fusion_folded_supported(B, N, L, H, R) && ldx % 2 == 0 && ldx >= L, VQA_E_UNSUPPORTED,
              "lowrank_bilinear_fusion_folded_fwd: needs N <= %d, R <= %d (2 above 48 regions), even L, H, ldx >= L (B=%d N=%d L=%d H=%d R=%d ldx=%d)",
              kFoldMaxN, kFoldMaxR, B, N, L, H, R, ldx);
  VQA_REQUIRE((long)B * N < (1L << 30), VQA_E_UNSUPPORTED, "lowrank_bilinear_fusion_folded_fwd: B*N too large");
  FoldPtrs wp{};
  for (int r = 0; r < R; ++r) {
    VQA_REQUIRE(w1[r] && b1[r] && aligned(w1[r], 8), VQA_E_BADARG,
                "lowrank_bilinear_fusion_folded_fwd: w1[%d]/b1[%d] null or unaligned", r, r);
    wp.w[r] = w1[r];
    wp.b[r] = b1[r];
  }
  VQA_REQUIRE(aligned(x, 8) && aligned(h2, 8) && aligned(out, 8), VQA_E_UNSUPPORTED,
              "lowrank_bilinear_fusion_folded_fwd: x/h2/out must be 8-byte aligned");
  if (fold_rt_supported(B, N, L, H, R, ldx, L, H))
    return fold_rt_forward(x, ldx, w1, b1, h2, out, B, N, L, H, R, static_cast<hipStream_t>(stream));
  return launch_folded<true>("lowrank_bilinear_fusion_folded_fwd", x, ldx, wp, L, h2, H, out, H, B, N, L, H, R,
                             static_cast<hipStream_t>(stream));
}
