// K4 forward and data gradient, rank-folded, on the register-tile engine (gemm_f32_rt.hpp): the sample-side contraction of
// putils.MutanFusion.forward (putils/__init__.py:232-238) against the per-sample weight  Weff_b = sum_r h2[b,r,:] (.) W1_r
// that exists only as MFMA operand fragments (bilinear_folded.hip has the derivation and the LDS-tile kernel this replaces
// at the shapes below).
//
//   forward        out[b,n,h] = sum_l Weff_b[h,l] x[b,n,l] + sum_r h2[b,r,h] b1_r[h]     rows o = h, contraction k = l
//   data gradient  dx[b,n,l]  = sum_h Weff_b[h,l] g[b,n,h]                               rows o = l, contraction k = h
//
// Both are  out_b[n][o] = sum_k A_b[o][k] X_b[n][k]  with  A_b[o][k] = sum_r f_r W_r[o][k]  where W_r is stored with k
// contiguous (W1_r itself forward, its transpose for the data gradient) and the fold factor f_r = h2[b,r,.] runs along
// the ROWS forward (one scalar per lane and row block) and along the CONTRACTION for the data gradient (a 16-byte
// fragment per chunk).  MFMA operands (v_mfma_f32_16x16x4_f32): A = the folded weight fragment (lane (r, g): row o0 +
// 16 i + r, steps k = 4 g + kb: the four components of ONE 16-byte load per rank), B = the sample's rows (lane (r, g):
// region 16 j + r, same k).  D block (i, j): lane (r, g) holds regions 16 j + r, rows o0 + 16 i + 4 g + t -- four
// consecutive output features: one 16-byte store.
//
// Shape: a workgroup owns S samples x (64 IB output rows over its 4 waves); a wave holds S x IB x 3 accumulator blocks
// (36 regions = 3 blocks of 16; forward S = 4, IB = 4: 192 registers; data gradient S = 2, IB = 5: 120) and reuses every
// weight fragment for its S samples -- at B = 512 both launches are exactly 256 workgroups, one per CU, one round.  Per
// 16-deep chunk and wave: 20 fragment loads, 8 VALU instructions of fold per (sample, row block) in front of its 12 MFMAs
// (the fold is inherent to the folded form: 0.67 VALU instructions per MFMA), no LDS, no barrier.  The forward's bias
// rides as contraction column L (x = 1, W = b1_r), part of the guarded tail chunk.
//
// Measured at B = 512, N = 36, L = 310, H = 510, R = 2 (rocprofv3, kernels launched back to back): forward 85 us, data
// gradient 81 us (the LDS-tile kernels: 99 / 100; in the step 90 / ~80 against 98 / 91).  With the fold placed by the compiler
// (three VALU instructions + a hazard nop in front of every group of three MFMAs) the same kernels took 94 / 96 us; without
// any fold arithmetic 81 / 77 -- what is left above the 56 us of pure MFMA issue at 2.2 GHz is the sustained clock under
// this load (~1.9 GHz) and the fixed cost of a one-round grid (prologue, the guarded tail chunk, the output stores: ~8 us).
// Tried and dropped: two samples per workgroup with two workgroups per CU (83 us forward); requesting the tail chunk
// before the main loop (80 more live registers: the forward spills).
#include <cstdlib>

#include "bilinear_folded.hpp"
#include "gemm_f32_rt.hpp"

namespace vqa {
namespace {

struct FoldRtArgs {
  const float* x;          // [B*N, ldx]  the sample rows (x forward, g for the data gradient)
  const float* w[2];       // R matrices [NO, ldw], contraction-contiguous
  const float* bias[2];    // forward: b1_r [NO]
  const float* h2;         // [B, R, ldh2]
  float* out;              // [B*N, ldo]
  int ldx, ldw, ldh2, ldo;
  int B, N, K, NO;         // samples, regions, contraction length, output rows
  int tiles_o;             // workgroup tiles of 64 IB output rows
};

constexpr int kNB = 3;     // region blocks of 16 (N <= 48)

// FG: (sample, row block) pairs whose weight fragments are folded in ONE burst of VALU instructions, ahead of their 12 FG MFMAs
template <bool FWD, int S, int IB, int R, int FG = 1>
__global__ __launch_bounds__(rt::kThreads, 1) void bilinear_fold_rt_kernel(FoldRtArgs p) {
  using rt::f32x4;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int b0 = (tile / p.tiles_o) * S;
  const int o0 = (tile % p.tiles_o) * (64 * IB) + wave * (16 * IB);
  const int B = p.B, N = p.N, K = p.K, NO = p.NO;
  if (o0 >= NO) return;                                   // (no barrier in this kernel)

  rt::rsrc_t Wb[R];
#pragma unroll
  for (int rk = 0; rk < R; ++rk) Wb[rk] = rt::make_rsrc(p.w[rk], ((size_t)(NO - 1) * p.ldw + K) * 4);
  const rt::rsrc_t Xb = rt::make_rsrc(p.x, ((size_t)(B * N - 1) * p.ldx + K) * 4);
  const rt::rsrc_t Hb = rt::make_rsrc(p.h2, (size_t)B * R * p.ldh2 * 4);
  uint32_t offW[IB], offX[S][kNB], offH[S];
#pragma unroll
  for (int i = 0; i < IB; ++i) offW[i] = ((uint32_t)min(o0 + 16 * i + r, NO - 1) * (uint32_t)p.ldw + 4u * g) * 4u;
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int b = min(b0 + s, B - 1);                     // (a group beyond B repeats the last sample; never stored)
#pragma unroll
    for (int j = 0; j < kNB; ++j) offX[s][j] = ((uint32_t)(b * N + min(16 * j + r, N - 1)) * (uint32_t)p.ldx + 4u * g) * 4u;
    offH[s] = (uint32_t)(b * R) * (uint32_t)p.ldh2 * 4u;
  }
  // forward: the fold factors of the lane's rows, one scalar per (sample, rank, row block)
  float hrow[FWD ? S : 1][R][FWD ? IB : 1];
  if constexpr (FWD) {
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int rk = 0; rk < R; ++rk)
#pragma unroll
        for (int i = 0; i < IB; ++i)
          hrow[s][rk][i] = rt::ldg4(Hb, offH[s] + (uint32_t)(rk * p.ldh2 + min(o0 + 16 * i + r, NO - 1)) * 4u, 0u);
  }

  f32x4 acc[S][IB][kNB];
#pragma unroll
  for (int s = 0; s < S; ++s)
#pragma unroll
    for (int i = 0; i < IB; ++i)
#pragma unroll
      for (int j = 0; j < kNB; ++j) acc[s][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Frag {
    f32x4 w[R][IB];                     // component kb = contraction step kb
    f32x4 x[S][kNB];
    f32x4 hk[FWD ? 1 : S][R];           // data gradient: the fold factors of the chunk's 4 contraction indices of the lane
  };
  auto load = [&](Frag& f, int c) {     // whole chunk c: k = 16 c .. 16 c + 15 < K
    const uint32_t so = (uint32_t)c * 64u;
#pragma unroll
    for (int rk = 0; rk < R; ++rk)
#pragma unroll
      for (int i = 0; i < IB; ++i) f.w[rk][i] = rt::ldg16(Wb[rk], offW[i], so);
#pragma unroll
    for (int s = 0; s < S; ++s) {
#pragma unroll
      for (int j = 0; j < kNB; ++j) f.x[s][j] = rt::ldg16(Xb, offX[s][j], so);
      if constexpr (!FWD) {
#pragma unroll
        for (int rk = 0; rk < R; ++rk) f.hk[s][rk] = rt::ldg16(Hb, offH[s] + (uint32_t)(rk * p.ldh2) * 4u + 16u * g, so);
      }
    }
  };
  // the folded weight fragment of (sample s, row block i): one packed multiply + one packed fma per pair of components
  auto fold = [&](const Frag& f, int s, int i) -> f32x4 {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 we;
    if constexpr (FWD) {
      const f32x2 h0 = f32x2{hrow[s][0][i], hrow[s][0][i]};
      f32x2 lo = f32x2{f.w[0][i][0], f.w[0][i][1]} * h0, hi = f32x2{f.w[0][i][2], f.w[0][i][3]} * h0;
      if constexpr (R == 2) {
        const f32x2 h1 = f32x2{hrow[s][1][i], hrow[s][1][i]};
        lo = __builtin_elementwise_fma(f32x2{f.w[1][i][0], f.w[1][i][1]}, h1, lo);
        hi = __builtin_elementwise_fma(f32x2{f.w[1][i][2], f.w[1][i][3]}, h1, hi);
      }
      we = f32x4{lo[0], lo[1], hi[0], hi[1]};
    } else {
      f32x2 lo = f32x2{f.w[0][i][0], f.w[0][i][1]} * f32x2{f.hk[s][0][0], f.hk[s][0][1]};
      f32x2 hi = f32x2{f.w[0][i][2], f.w[0][i][3]} * f32x2{f.hk[s][0][2], f.hk[s][0][3]};
      if constexpr (R == 2) {
        lo = __builtin_elementwise_fma(f32x2{f.w[1][i][0], f.w[1][i][1]}, f32x2{f.hk[s][1][0], f.hk[s][1][1]}, lo);
        hi = __builtin_elementwise_fma(f32x2{f.w[1][i][2], f.w[1][i][3]}, f32x2{f.hk[s][1][2], f.hk[s][1][3]}, hi);
      }
      we = f32x4{lo[0], lo[1], hi[0], hi[1]};
    }
    return we;
  };
  // One chunk.  The fold of row block q + 1 is issued as ONE burst in front of the 12 MFMAs of block q, which use the
  // fragment folded a block earlier: next to fp32 MFMAs a VALU instruction costs mostly the interruption of the MFMA
  // stream, and an MFMA that reads a register the VALU has just written waits for it -- folded right in front of each
  // group of three MFMAs (the compiler's own placement) the kernel ran at 60 % of the matrix pipe.
  // PIN: pin the order with sched_group_barrier (inside the pipeline step); LOADS = loads to interleave.
  auto compute = [&](const Frag& f, auto pin, auto nloads) {
    constexpr bool PIN = decltype(pin)::value;
    constexpr int NL = decltype(nloads)::value;
    constexpr int NQ = S * IB;                 // (sample, row block) pairs, q = s IB + i
    constexpr int NV = R == 2 ? 4 : 2;         // VALU instructions of one fold
    f32x4 we[FG], wn[FG];
#pragma unroll
    for (int t = 0; t < FG; ++t) we[t] = fold(f, min(t, NQ - 1) / IB, min(t, NQ - 1) % IB);
    if constexpr (PIN) __builtin_amdgcn_sched_group_barrier(0x002, NV * FG, 0);
    int issued = 0;
#pragma unroll
    for (int q0 = 0; q0 < NQ; q0 += FG) {
#pragma unroll
      for (int t = 0; t < FG; ++t) wn[t] = we[t];
      if (q0 + FG < NQ) {
#pragma unroll
        for (int t = 0; t < FG; ++t)
          if (q0 + FG + t < NQ) {
            wn[t] = fold(f, (q0 + FG + t) / IB, (q0 + FG + t) % IB);
            if constexpr (PIN) __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);     // (consecutive groups: one burst)
          }
      }
#pragma unroll
      for (int t = 0; t < FG; ++t) {
        const int q = q0 + t;
        if (q >= NQ) break;
        const int s = q / IB, i = q % IB;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int j = 0; j < kNB; ++j) {
            acc[s][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(we[t][kb], f.x[s][j][kb], acc[s][i][j], 0, 0, 0);
            if constexpr (PIN) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              // loads spread evenly over the chunk's MFMAs
              const int m = q * 4 * kNB + kb * kNB + j;
              if (issued < NL && (m + 1) * NL / (NQ * 4 * kNB) > issued) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                ++issued;
              }
            }
          }
      }
#pragma unroll
      for (int t = 0; t < FG; ++t) we[t] = wn[t];
    }
  };
  // one pipeline step (gemm_nt_kernel): chunk cn is requested in the shadow of chunk c's MFMAs
  auto step = [&](Frag& fn, int cn, const Frag& f) {
    constexpr int NL = R * IB + S * kNB + (FWD ? 0 : S * R);
    load(fn, cn);
    compute(f, std::true_type{}, std::integral_constant<int, NL>{});
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nfull = K >> 4;
  const int c_pairs = nfull & ~1;
  {
    Frag f0, f1;
    if (c_pairs > 0) {
      load(f0, 0);
      __builtin_amdgcn_sched_barrier(0);
      for (int c = 0; c < c_pairs; c += 2) {
        step(f1, c + 1, f0);
        step(f0, min(c + 2, c_pairs - 1), f1);            // (last pair: a harmless reload)
      }
    }
    if (c_pairs < nfull) {
      load(f0, c_pairs);
      compute(f0, std::false_type{}, std::integral_constant<int, 0>{});
    }
  }
  if ((K & 15) != 0 || FWD) {
    // tail chunk: the contraction indices 16 nfull + 4 g + kb, guarded element by element; forward: index K is the bias
    // column (W = b1_r, x = 1)
    Frag f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int k = 16 * nfull + 4 * g + kb;
      const bool in = k < K;
      const uint32_t kc = (uint32_t)min(k, K - 1);
#pragma unroll
      for (int rk = 0; rk < R; ++rk)
#pragma unroll
        for (int i = 0; i < IB; ++i) {
          float v = rt::ldg4(Wb[rk], offW[i] - 16u * g + kc * 4u, 0u);
          v = in ? v : 0.f;
          if constexpr (FWD) {
            if (k == K) v = p.bias[rk][min(o0 + 16 * i + r, NO - 1)];
          }
          f.w[rk][i][kb] = v;
        }
#pragma unroll
      for (int s = 0; s < S; ++s) {
#pragma unroll
        for (int j = 0; j < kNB; ++j) {
          float v = rt::ldg4(Xb, offX[s][j] - 16u * g + kc * 4u, 0u);
          v = in ? v : 0.f;
          if constexpr (FWD) {
            if (k == K) v = 1.f;
          }
          f.x[s][j][kb] = v;
        }
        if constexpr (!FWD) {
#pragma unroll
          for (int rk = 0; rk < R; ++rk) {
            const float v = rt::ldg4(Hb, offH[s] + (uint32_t)(rk * p.ldh2) * 4u + kc * 4u, 0u);
            f.hk[s][rk][kb] = in ? v : 0.f;
          }
        }
      }
    }
    compute(f, std::false_type{}, std::integral_constant<int, 0>{});
  }

  // ---- store: lane (r, g) holds regions 16 j + r, output rows o0 + 16 i + 4 g + t ----
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int b = b0 + s;
    if (b >= B) break;
#pragma unroll
    for (int j = 0; j < kNB; ++j) {
      const int n = 16 * j + r;
      if (n >= N) continue;
      float* __restrict__ row = p.out + (size_t)(b * N + n) * p.ldo;
#pragma unroll
      for (int i = 0; i < IB; ++i) {
        const int o = o0 + 16 * i + 4 * g;
        if (o + 3 < NO) {
          *reinterpret_cast<f32x4*>(row + o) = acc[s][i][j];      // (8-byte aligned at least: ldo and o are even)
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (o + t < NO) row[o + t] = acc[s][i][j][t];
        }
      }
    }
  }
}

template <bool FWD>
int launch(const FoldRtArgs& a0, int R, hipStream_t s) {
  FoldRtArgs a = a0;
  constexpr int S = FWD ? 4 : 2, IB = FWD ? 4 : 5;
  a.tiles_o = (a.NO + 64 * IB - 1) / (64 * IB);
  const dim3 grid((unsigned)(((a.B + S - 1) / S) * a.tiles_o));
  // fold-burst size: the folds of FG (sample, row block) pairs issued together, ahead of their 12 FG MFMAs -- fewer, longer
  // interruptions of the MFMA stream (the K5 mask finding of round 4).  Measured in the step at B = 512 (us per launch,
  // FG = 1 / 2 / 4): forward 91.2 / 91.0 / 88.9, data gradient + weight side 193.7 / 189.6 / 196.6 (more live fragments than
  // its 120 accumulators leave room for).  VQA_K4_FG overrides (measurement knob).
  const char* fg_opt = vqa::option("VQA_K4_FG");
  const int fg = fg_opt != nullptr ? std::atoi(fg_opt) : (FWD ? 4 : 2);
  if (R == 1)
    VQA_LAUNCH((bilinear_fold_rt_kernel<FWD, S, IB, 1>), grid, dim3(rt::kThreads), 0, s, a);
  else if (fg == 2)
    VQA_LAUNCH((bilinear_fold_rt_kernel<FWD, S, IB, 2, 2>), grid, dim3(rt::kThreads), 0, s, a);
  else if (fg == 4)
    VQA_LAUNCH((bilinear_fold_rt_kernel<FWD, S, IB, 2, 4>), grid, dim3(rt::kThreads), 0, s, a);
  else
    VQA_LAUNCH((bilinear_fold_rt_kernel<FWD, S, IB, 2>), grid, dim3(rt::kThreads), 0, s, a);
  return check_launch(FWD ? "lowrank_bilinear_fusion_folded_fwd (register-tile)" : "lowrank_bilinear_fusion_folded_bwd (dx, register-tile)");
}

}  // namespace

// Shapes the register-tile kernels take: R <= 2, 17 .. 48 regions (three blocks of 16; fewer regions waste them), a batch
// that fills the chip, 4-byte-addressable tensors below 4 GiB.  VQA_K4_RT=0 keeps the LDS-tile kernels.
bool fold_rt_supported(int B, int N, int K, int NO, int R, int ldx, int ldw, int ldo) {
  const bool off = vqa::option_is("VQA_K4_RT", '0');
  return !off && R >= 1 && R <= 2 && N > 16 && N <= 48 && B >= 256 && K >= 32 && NO >= 64 && ldx % 2 == 0 && ldw % 2 == 0 &&
         ldo % 2 == 0 && (size_t)B * N * (size_t)(ldx > ldo ? ldx : ldo) * 4 < (1ull << 32) && (size_t)NO * ldw * 4 < (1ull << 32);
}

int fold_rt_forward(const float* x, int ldx, const float* const* w1, const float* const* b1, const float* h2, float* out, int B,
                    int N, int L, int H, int R, hipStream_t s) {
  FoldRtArgs a{};
  a.x = x;
  for (int r = 0; r < R; ++r) {
    a.w[r] = w1[r];
    a.bias[r] = b1[r];
  }
  a.h2 = h2;
  a.out = out;
  a.ldx = ldx;
  a.ldw = L;
  a.ldh2 = H;
  a.ldo = H;
  a.B = B;
  a.N = N;
  a.K = L;
  a.NO = H;
  return launch<true>(a, R, s);
}

int fold_rt_data_gradient(const float* g, const float* const* w1t, const float* h2, float* d_x, int B, int N, int L, int H, int R,
                          hipStream_t s) {
  FoldRtArgs a{};
  a.x = g;
  for (int r = 0; r < R; ++r) a.w[r] = w1t[r];
  a.h2 = h2;
  a.out = d_x;
  a.ldx = H;
  a.ldw = H;
  a.ldh2 = H;
  a.ldo = L;
  a.B = B;
  a.N = N;
  a.K = H;
  a.NO = L;
  return launch<false>(a, R, s);
}

}  // namespace vqa
