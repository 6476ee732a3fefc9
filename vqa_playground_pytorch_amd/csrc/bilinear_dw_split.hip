// K4's weight gradient on the SPLIT engine: the per-sample product of the rank-folded backward,
//
//     P_b = g_b^T [x_b | 1]   [H, L + 1]     dW1_r += diag(h2_r[b]) P_b[:, :L]     db1_r += h2_r[b] * P_b[:, L]
//                                             dh2_r[b,h] = sum_l P_b[h,l] W1_r[h,l] + P_b[h,L] b1_r[h]
//
// (putils/__init__.py:232-238 backward; same outputs and workspace layout as bilinear_dw_rt.hip, whose fp32-MFMA form ran at
// 50 % matrix-pipe occupancy and 1.43 x its algorithmic traffic for two rounds) with every fp32 product formed from exact
// three-way bf16 splits of BOTH operands -- six partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulation: an fp32
// product, csrc/gemm_f32_split.hpp.  The contraction runs over the 36 regions of ONE sample (two 32-deep steps, rows past 36
// zero), which is the ROW index of g and x in memory: both tiles are split while they are staged -- as they lie in memory, one
// bf16 image per plane -- and come back as MFMA fragments through gfx950's transposing LDS read (ds_read_b64_tr_b16: four
// consecutive rows of one column per lane; two of them are a lane's eight contraction indices).  No operand is split twice inside a workgroup; across
// workgroups x_b is split once per 64-row tile of H (8 x) and g_b once per column half (2 x): 8.7 us of VALU per launch at
// B = 512 against 29 us of matrix pipe.
//
// Workgroup = 64 features h x half of the (padded) columns x one slab of samples; 4 waves = 2 (h halves: 2 blocks of 16) x 2
// (column quarters: LBW blocks of 16).  Per sample: 2 x 6 x 2 x LBW MFMAs out of one LDS buffer while the next sample is split into
// the other (one barrier per sample), fold P into the slab's dW accumulators (fp32 registers), contract it against W1_r (registers,
// in the accumulator layout) for dh2.  Column L of the x tile is the constant 1: the bias gradient and dh2's bias term ride along.
#include "bilinear_folded.hpp"
#include "gemm_f32_split.hpp"

namespace vqa {
namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kR = 2, kRows = 48;

struct DwSplitArgs {
  const float* g;        // [B N, H]
  const float* x;        // [B N, L]
  const float* h2;       // [B, R, H]
  const float* w[kR];    // [H, L]
  const float* b[kR];    // [H]
  float* slab;           // [kDwSplitSlabs][R][H L]
  float* dbslab;         // [kDwSplitSlabs][R][H]
  float* part;           // [4][B R H]: dh2's partial sums over the four column quarters (2 workgroup halves x 2 waves)
  int B, N, L, H, sps;
};

typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4v mfma32(const u32x4v& a, const u32x4v& b, const f32x4v& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int LBW>
__global__ __launch_bounds__(256, 1) void bilinear_dw_split_kernel(DwSplitArgs p) {
  extern __shared__ __attribute__((aligned(16))) char dws_smem[];
  constexpr int LH = 32 * LBW;                     // columns of this workgroup's half (2 waves x LBW blocks x 16)
  constexpr int PG = 64 * 2 + 16, PX = LH * 2 + 16;   // row pitches of the plane images (bytes)
  constexpr int XPAIRS_ROW = LH / 2;
  constexpr int IMG = 3 * kRows * (PG + PX);       // one buffer: [3 planes][kRows][PG] then [3 planes][kRows][PX]
  float* h2s = reinterpret_cast<float*>(dws_smem + 2 * IMG);     // [sps][R][64]: the slab's question-side factors of this h tile
  const int L = p.L, H = p.H, N = p.N;
  const int lane = threadIdx.x & 63, r16 = lane & 15, gq = lane >> 4, wave = threadIdx.x >> 6;
  const int wh = wave >> 1, wl = wave & 1;
  const int ht = blockIdx.x % ((H + 63) / 64), lh = (blockIdx.x / ((H + 63) / 64)) & 1, slab = blockIdx.x / (2 * ((H + 63) / 64));
  const int h0 = ht * 64, lbase = lh * LH;
  const int b_lo = slab * p.sps, b_hi = min(p.B, b_lo + p.sps);
  // zero both image buffers once: the rows N .. 47 stay zero, everything else is rewritten per sample
  for (int t = threadIdx.x; t < 2 * IMG / 16; t += 256) reinterpret_cast<uint4*>(dws_smem)[t] = make_uint4(0u, 0u, 0u, 0u);
  for (int t = threadIdx.x; t < (b_hi - b_lo) * kR * 64; t += 256) {
    const int bb = t / (kR * 64), u = t - bb * (kR * 64), r = u >> 6, h = h0 + (u & 63);
    h2s[t] = h < H ? p.h2[((size_t)(b_lo + bb) * kR + r) * H + h] : 0.f;
  }
  // W1_r in the accumulator layout: block (j, lb): rows h0 + 16 (2 wh + j) + 4 gq + i, column lbase + 16 (LBW wl + lb) + r16;
  // column L holds b1_r (it meets P's column of region sums), columns past it and rows past H are zero
  float w1f[kR][2][LBW][4];
#pragma unroll
  for (int r = 0; r < kR; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int h = h0 + 16 * (2 * wh + j) + 4 * gq + i, l = lbase + 16 * (LBW * wl + lb) + r16;
          float v = 0.f;
          if (h < H) v = l < L ? p.w[r][(size_t)h * L + l] : (l == L ? p.b[r][h] : 0.f);
          w1f[r][j][lb][i] = v;
        }
  f32x4v dw[kR][2][LBW];
#pragma unroll
  for (int r = 0; r < kR; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb) dw[r][j][lb] = f32x4v{0.f, 0.f, 0.f, 0.f};

  // ---- staging: float2 pieces (rows of g / x are 8-byte aligned: H and L are even), split into the three plane images.
  // Everything about a piece but the sample is fixed per thread and hoisted: g tile -- thread (row t >> 5, pair t & 31), pieces 8
  // rows apart; x tile -- threads 0..239 as (row t / XPAIRS_ROW', pair), pieces 3 rows apart (N = 36 = 12 x 3); a sample only
  // moves the base pointer.  (Recomputed per piece and sample the addressing was 600 of the kernel's 970 VALU instructions per
  // sample and wave: rocprofv3 SQ_INSTS_VALU, tools/k4_dw_counters.sh.)
  constexpr int GP = 5, XP = 12;
  static_assert(XPAIRS_ROW * 3 <= 256, "three rows of x pairs per pass");
  const int gc = threadIdx.x & 31, gr = threadIdx.x >> 5;
  const bool g_ok = h0 + 2 * gc < H;                         // (only the last h tile has pairs past H)
  const uint32_t goff = (uint32_t)(gr * H + min(h0 + 2 * gc, H - 2));
  const uint32_t glds = (uint32_t)(gr * PG + 4 * gc);
  const sp::f32x2 gmask = g_ok ? sp::f32x2{1.f, 1.f} : sp::f32x2{0.f, 0.f};
  const bool x_thread = threadIdx.x < 3 * XPAIRS_ROW;
  const int xr = threadIdx.x / XPAIRS_ROW, xc = threadIdx.x - xr * XPAIRS_ROW, xl = lbase + 2 * xc;
  const uint32_t xoff = (uint32_t)(xr * L + min(xl, L - 2));
  const uint32_t xlds = (uint32_t)(xr * PX + 4 * xc);
  const sp::f32x2 xmask = xl < L ? sp::f32x2{1.f, 1.f} : sp::f32x2{0.f, 0.f};
  const sp::f32x2 xone = xl == L ? sp::f32x2{1.f, 0.f} : sp::f32x2{0.f, 0.f};     // column L: the constant 1
  sp::f32x2 sg[GP], sx[XP];
  auto fetch = [&](int b) {
    const float* gb = p.g + (size_t)b * N * H + goff;
    const float* xb = p.x + (size_t)b * N * L + xoff;
#pragma unroll
    for (int k = 0; k < GP; ++k)
      if (k < 4 || wave < 2) sg[k] = *reinterpret_cast<const sp::f32x2*>(gb + (size_t)k * 8 * H);     // rows 32..35: waves 0, 1
    if (x_thread) {
#pragma unroll
      for (int k = 0; k < XP; ++k) sx[k] = *reinterpret_cast<const sp::f32x2*>(xb + (size_t)k * 3 * L);
    }
  };
  auto stage = [&](char* buf) {
    char* gdst = buf + glds;
    char* xdst = buf + 3 * kRows * PG + xlds;
#pragma unroll
    for (int k = 0; k < GP; ++k)
      if (k < 4 || wave < 2) {
        uint32_t q0, q1, q2;
        sp::split_pair<false>(sg[k] * gmask, q0, q1, q2);
        *reinterpret_cast<uint32_t*>(gdst + k * 8 * PG) = q0;
        *reinterpret_cast<uint32_t*>(gdst + k * 8 * PG + kRows * PG) = q1;
        *reinterpret_cast<uint32_t*>(gdst + k * 8 * PG + 2 * kRows * PG) = q2;
      }
    if (x_thread) {
#pragma unroll
      for (int k = 0; k < XP; ++k) {
        uint32_t q0, q1, q2;
        sp::split_pair<false>(sx[k] * xmask + xone, q0, q1, q2);
        *reinterpret_cast<uint32_t*>(xdst + k * 3 * PX) = q0;
        *reinterpret_cast<uint32_t*>(xdst + k * 3 * PX + kRows * PX) = q1;
        *reinterpret_cast<uint32_t*>(xdst + k * 3 * PX + 2 * kRows * PX) = q2;
      }
    }
  };
  // Fragments of v_mfma_f32_16x16x32_bf16 by the transposing read: lane 4 q + p of a 16-lane group addresses row q, columns
  // 4 p .. 4 p + 3 of a 4 x 16 block and receives column (lane & 15), rows 0..3; a lane's eight contraction indices 8 gq .. 8 gq + 7
  // are two such reads.  The 36 regions are two 32-deep steps: step 1 holds rows 32..35, the lane groups past them read zero rows.
  const int tq = r16 >> 2, tp = r16 & 3;
  const int row_s[2] = {8 * gq + tq, 32 + 8 * min(gq, 1) + tq};
  auto frag = [&](const char* image, int pitch, int plane, int step, int colblock) -> u32x4v {
    const char* src = image + ((size_t)plane * kRows + row_s[step]) * pitch + (16 * colblock + 4 * tp) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(src));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(src + 4 * pitch));
    const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
    return u32x4v{a.x, a.y, b.x, b.y};
  };
  // One barrier per sample: while sample b is multiplied out of buffer `cur`, sample b + 1 is split into the other buffer (its
  // loads went out one sample earlier) and sample b + 2's loads are requested.
  if (b_lo < b_hi) fetch(b_lo);
  __syncthreads();                 // the buffers are zeroed, h2s is in place
  if (b_lo < b_hi) stage(dws_smem);
  if (b_lo + 1 < b_hi) fetch(b_lo + 1);
  __syncthreads();
  int cur = 0;
  for (int b = b_lo; b < b_hi; ++b, cur ^= 1) {
    const char* gim = dws_smem + cur * IMG;
    const char* xim = gim + 3 * kRows * PG;
    if (b + 1 < b_hi) {
      stage(dws_smem + (cur ^ 1) * IMG);
      if (b + 2 < b_hi) fetch(b + 2);
    }
    f32x4v P[2][LBW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb) P[j][lb] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4v a[3][2], bx[3][LBW];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
        for (int j = 0; j < 2; ++j) a[pl][j] = frag(gim, PG, pl, s, 2 * wh + j);
#pragma unroll
        for (int lb = 0; lb < LBW; ++lb) bx[pl][lb] = frag(xim, PX, pl, s, LBW * wl + lb);
      }
      constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};      // the six partial products of weight >= 2^-16
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int lb = 0; lb < LBW; ++lb) P[j][lb] = mfma32(a[PA[k]][j], bx[PB[k]][lb], P[j][lb]);
    }
    // fold into the slab's gradients; contract against W1_r (and b1_r in column L) for dh2
    const float* hq = h2s + (size_t)(b - b_lo) * kR * 64;
#pragma unroll
    for (int r = 0; r < kR; ++r)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const f32x4v qh = *reinterpret_cast<const f32x4v*>(hq + r * 64 + 16 * (2 * wh + j) + 4 * gq);
        f32x4v part = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int lb = 0; lb < LBW; ++lb) {
          dw[r][j][lb] += qh * P[j][lb];
#pragma unroll
          for (int i = 0; i < 4; ++i) part[i] = fmaf(P[j][lb][i], w1f[r][j][lb][i], part[i]);
        }
        // sum over the 16 columns of a block (the lanes r16); each column wave writes its own partial sum (4 parts in all)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s = part[i];
          s += __shfl_xor(s, 1);
          s += __shfl_xor(s, 2);
          s += __shfl_xor(s, 4);
          s += __shfl_xor(s, 8);
          part[i] = s;
        }
        const int h = h0 + 16 * (2 * wh + j) + 4 * gq;
        if (r16 == 0) {
          float* dst = p.part + (size_t)(2 * lh + wl) * p.B * kR * H + ((size_t)b * kR + r) * H + h;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (h + i < H) dst[i] = part[i];
        }
      }
    __syncthreads();
  }
  // slab[slab][r][h][l] (l < L) and dbslab[slab][r][h] (column L)
#pragma unroll
  for (int r = 0; r < kR; ++r) {
    float* dst = p.slab + ((size_t)slab * kR + r) * H * L;
    float* dbd = p.dbslab + ((size_t)slab * kR + r) * H;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb) {
        const int l = lbase + 16 * (LBW * wl + lb) + r16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int h = h0 + 16 * (2 * wh + j) + 4 * gq + i;
          if (h >= H) continue;
          if (l < L) dst[(size_t)h * L + l] = dw[r][j][lb][i];
          else if (l == L) dbd[h] = dw[r][j][lb][i];
        }
      }
  }
}

}  // namespace

// OPT-IN (VQA_K4_DW_SPLIT=1): measured at B = 512 the kernel takes ~150 us against the fp32 register-tile form's 86
// (docs/measured_negatives_r05.md: 730 VALU instructions per sample and wave -- the fold's accumulators live in AGPRs next to
// 80 registers of W1 and every FMA on them is three instructions -- and 185 LDS instructions for 120 MFMAs).
bool dw_split_supported(int B, int N, int L, int H, int R, int ldx) {
  if (!vqa::option_is("VQA_K4_DW_SPLIT", '1')) return false;
  const int lbw = (L + 1 + 63) / 64;
  // (N = 36: the staging's piece maps are built for it; B <= 512: the slab's question-side factors wait in LDS, 32 samples' worth)
  return R == kR && N == 36 && ldx == L && L % 2 == 0 && L >= 2 && H % 2 == 0 && lbw >= 1 && lbw <= 5 && H >= 16 && B >= 64 &&
         (B + kDwSplitSlabs - 1) / kDwSplitSlabs <= 32 && (size_t)B * N * H * 4 < (1ull << 32) && (size_t)B * N * L * 4 < (1ull << 32);
}

int dw_split_launch(const float* g, const float* x, const float* h2, const float* const* w1, const float* const* b1, float* slab,
                    float* dbslab, float* part, int B, int N, int L, int H, int R, hipStream_t s) {
  DwSplitArgs a{};
  a.g = g;
  a.x = x;
  a.h2 = h2;
  for (int r = 0; r < kR; ++r) {
    a.w[r] = w1[r];
    a.b[r] = b1[r];
  }
  a.slab = slab;
  a.dbslab = dbslab;
  a.part = part;
  a.B = B;
  a.N = N;
  a.L = L;
  a.H = H;
  a.sps = (B + kDwSplitSlabs - 1) / kDwSplitSlabs;
  const int lbw = (L + 1 + 63) / 64;
  const dim3 grid(kDwSplitSlabs * 2 * ((H + 63) / 64));
#define LAUNCH(LBW_)                                                                                                   \
  {                                                                                                                    \
    const size_t lds = 2 * ((size_t)3 * kRows * (64 * 2 + 16) + (size_t)3 * kRows * (32 * LBW_ * 2 + 16)) +          \
                       (size_t)a.sps * kR * 64 * 4;                                                                    \
    VQA_ENSURE_LDS((bilinear_dw_split_kernel<LBW_>), lds);                                                             \
    VQA_LAUNCH((bilinear_dw_split_kernel<LBW_>), grid, dim3(256), lds, s, a);                                          \
  }
  switch (lbw) {
    case 1: LAUNCH(1) break;
    case 2: LAUNCH(2) break;
    case 3: LAUNCH(3) break;
    case 4: LAUNCH(4) break;
    default: LAUNCH(5) break;
  }
#undef LAUNCH
  return check_launch("lowrank_bilinear_fusion_folded_bwd (dW, split engine)");
}

}  // namespace vqa
