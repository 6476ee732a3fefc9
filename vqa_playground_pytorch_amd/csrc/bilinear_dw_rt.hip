// K4 backward, weight side, register-tile form: dW1_r, db1_r and dh2 of the rank-folded low-rank bilinear fusion
// (putils.MutanFusion.forward, putils/__init__.py:232-238) from the per-sample product P_b = g_b^T x_b.
//
//   P_b[h,l]     = sum_n g[b,n,h] x[b,n,l]                       (N = 36 regions: NINE k = 4 steps of v_mfma_f32_16x16x4_f32,
//                                                                 no padding of the sample to 48 rows)
//   dW1_r[h,l]   = sum_b h2[b,r,h] P_b[h,l]                      (rank fold: R FMAs per element of P_b)
//   dh2[b,r,h]   = sum_l W1_r[h,l] P_b[h,l] + b1_r[h] sum_n g[b,n,h]
//   db1_r[h]     = sum_b h2[b,r,h] sum_n g[b,n,h]
//
// The two bias terms ride along as ONE extra column: x is given a column L of ones (so P_b[h,L] = sum_n g[b,n,h]) and
// W1_r a column L holding b1_r -- column L of the folded accumulator then IS db1_r and the dh2 contraction covers its
// second term.  L = 310 leaves that column free inside the fifth 64-column span.
//
// Shape of the kernel (same engine idea as gemm_f32_rt.hpp: one wave per SIMD, operands straight from L2 into registers):
// a workgroup owns ONE block of 16 features h and two slices of the samples; its 4 waves = 2 column halves x 2 slices.  The
// L + 1 columns are 5 spans of 64; a lane holds 4 consecutive columns of a span (gemm_f32_rt.hpp's TN trick: component e
// of the lane's x value feeds accumulator block e) and a wave takes the components {0, 1} or {2, 3} of every span -- 10
// accumulator blocks for P_b (40 registers) and 10 R for the folded dW (80 at R = 2).  (One wave holding all 20 blocks
// needs 160 accumulator registers that the VALU fold must reach, i.e. arch VGPRs: it spilled 42 of them and the fold ran
// at scratch-memory latency -- 130 us, of which 70 us fold.)  Per contraction step (4 region rows): one dword load of g
// (lane = feature) and five 8-byte loads of x, 10 MFMAs.  Loads run a ring of D steps ahead, across sample boundaries
// (the rows of consecutive samples are consecutive rows of g and x); a ring slot is refilled under the MFMAs of the step
// AFTER the one that consumed it.  After a sample's last step the wave folds its half of P_b into the rank accumulators
// and contracts it against W1_r (16 x 320 per rank, in LDS, shared by the 4 waves); the P accumulators restart from a
// zero C operand, not from a clearing pass.  The two slices meet in LDS at the end (8 slabs of dW for 16 slices); the two
// column halves of dh2 go to two partial buffers, added by bilinear_dh2_reduce_kernel.
//
// The 64 x 64 LDS-tile kernel this replaces (bilinear_fusion.hip, bilinear_dw_dh2_kernel) ran at 35 % matrix-pipe occupancy
// (135 us at B = 512; profiles/r02_a_pmc_mfma): 48-row padded samples, a barrier per 40-row stage, operands through LDS.
//
// Measured at B = 512, N = 36, L = 310, H = 510, R = 2 (stamps of the diagnostic build, per wave): 2880 MFMAs in 98.5 k
// shader cycles (34.2 per MFMA; 32 is the pipe's rate), the fold 1070 cycles per sample (VALU-issue bound: 80 reads of P
// out of the accumulator registers, 320 FMAs, a reduce-scatter over the quad + two row rotations for dh2, ONE guarded store
// per rank), prologue 3.6 us (the W tile comes through buffer loads with an out-of-range offset for the padding: a
// conditional load per element serialised 22 memory latencies, 7.6 us), 85 us per launch at the ~1.8-1.95 GHz the part
// sustains under this load.
// (Tried: four sample slices per workgroup = two waves per SIMD, 3-step ring to stay within 256 registers -- 81 us, no change.
//  The waves are not waiting for latency: every one of the 32 feature blocks streams all of x through its CU, 0.73 GB of
//  L2 -> register traffic per launch, ~9.6 TB/s at 80 us.  More reuse per workgroup -- 32 features, i.e. twice the
//  accumulators -- is what would move it, not occupancy.)
#include <cstdlib>

#include "bilinear_folded.hpp"
#include "gemm_f32_rt.hpp"

namespace vqa {

namespace {

constexpr int kSpans = 5;            // 64-column spans of the (L + 1)-wide row (256 < L < 320: column L sits in the last one)
constexpr int kCols = 64 * kSpans;   // 320
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned int))));

struct DwRtArgs {
  const float* g;     // [B*N, H]
  const float* x;     // [B*N, L]
  const float* h2;    // [B, R, H]
  const float* w[2];  // W1_r [H, L]
  const float* b[2];  // b1_r [H]
  float* slab;        // [SG][R][H*L]
  float* dbslab;      // [SG][R][H]
  float* part;        // [2][B*R*H]: dh2 partial sums of the two column halves
  int B, N, L, H;
  int HB;             // feature blocks = ceil(H / 16)
  int spl;            // samples per slice (2 slices per workgroup)
  unsigned long long* stamps;   // timing experiments only (TUNE & 4)
};

// TUNE (diagnostic builds; 0 in production): bit 2 = stamp the shader cycles of the MFMA steps and of the fold, and the 100 MHz
// ticks of prologue / loop / whole wave into p.stamps (VQA_K4_DW_TUNE=4 prints their means after 20 launches)
template <int R, int NS, int D, int TUNE = 0>
__global__ __launch_bounds__(rt::kThreads, 1) void bilinear_dw_rt_kernel(DwRtArgs p) {
  using rt::f32x4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* Ws = reinterpret_cast<float*>(smem_raw);   // [R][16][kCols]; reused for the cross-slice sum at the end
  const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, gq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int eh = wave & 1, sl = wave >> 1;            // column half (components 2 eh, 2 eh + 1) and sample slice
  const int id = xcd_remap(blockIdx.x, gridDim.x);    // an XCD gets the 32 feature blocks of ONE sample group: its L2
  const int hb = id % p.HB, sg = id / p.HB;           // serves that group's x and g to all of them
  const int h0 = hb * 16;
  unsigned long long tk_in = 0;
  if constexpr ((TUNE & 4) != 0) tk_in = __builtin_amdgcn_s_memrealtime();
  const int slice = sg * 2 + sl;
  const int b_lo = min(slice * p.spl, p.B), b_hi = min(p.B, b_lo + p.spl);
  const int L = p.L, H = p.H, N = p.N;

  const rt::rsrc_t Gb = rt::make_rsrc(p.g, (size_t)p.B * N * H * 4);
  const rt::rsrc_t Xb = rt::make_rsrc(p.x, (size_t)p.B * N * L * 4);
  const rt::rsrc_t Hb = rt::make_rsrc(p.h2, (size_t)p.B * R * H * 4);
  const uint32_t offG = ((uint32_t)gq * (uint32_t)H + (uint32_t)min(h0 + r, H - 1)) * 4u;
  uint32_t offX[kSpans];
#pragma unroll
  for (int q = 0; q < kSpans; ++q) offX[q] = ((uint32_t)gq * (uint32_t)L + (uint32_t)(64 * q + 4 * r + 2 * eh)) * 4u;
  const uint32_t offH = (uint32_t)(h0 + 4 * gq) * 4u;
  // the ones column: element L (even) of a row sits in the last span, lane (L % 64) / 4, component L % 4 -- local
  // component 0 of the half (L % 4) / 2; its neighbour L + 1 is forced to 0
  const bool one_lane = (r == (L % 64) / 4) && (eh == (L % 4) / 2);

  f32x4 acc[R][kSpans][2];   // folded dW (+ db in column L): [rank][span][local component] over the 4 rows t
#pragma unroll
  for (int rk = 0; rk < R; ++rk)
#pragma unroll
    for (int q = 0; q < kSpans; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[rk][q][e] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Step {
    float a;
    f32x2 xb[kSpans];
  };
  const int row_hi = b_hi * N;                          // end of the rows of g / x of this wave's samples
  auto load_step = [&](Step& st, int row) {             // rows row .. row + 3 (lane group gq takes row + gq)
    const uint32_t rw = (uint32_t)min(row, max(row_hi - 4, 0));   // (past the slice: a harmless reload of its last step)
    st.a = rt::ldg4(Gb, offG, rw * (uint32_t)H * 4u);
#pragma unroll
    for (int q = 0; q < kSpans; ++q)
      st.xb[q] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(Xb, (int)offX[q], (int)(rw * (uint32_t)L * 4u), 0));
  };
  auto fix_ones = [&](Step& st) {                       // x[.., L] = 1, x[.., L + 1] = 0 (the bias column)
    const float v0 = st.xb[kSpans - 1][0], v1 = st.xb[kSpans - 1][1];
    st.xb[kSpans - 1][0] = one_lane ? 1.f : v0;
    st.xb[kSpans - 1][1] = one_lane ? 0.f : v1;
  };

  // the first D steps of the row stream are requested before the W tile is staged: their latency runs under it.  (An
  // empty slice re-reads clamped rows it never uses.)
  Step ring[D];
#pragma unroll
  for (int d = 0; d < D; ++d) load_step(ring[d], b_lo * N + 4 * d);
  f32x4 h2v[R];
#pragma unroll
  for (int rk = 0; rk < R; ++rk) h2v[rk] = rt::ldg16(Hb, offH, (uint32_t)((min(b_lo, p.B - 1) * R + rk) * H) * 4u);

  // W1_r rows of this feature block, column L = b1_r, zero beyond (and zero rows beyond H).  16 threads per row, float2
  // loads 32 columns apart, all of a thread's loads in flight together (one wave per SIMD: nothing else hides latency)
  {
    const int hl = tid >> 4, c0 = (tid & 15) * 2;
    const int h = h0 + hl;
    f32x2 wv[R][kCols / 32];
    float bv[R];
    constexpr uint32_t kOut = 0x80000000u;                 // beyond the descriptor's range: the load returns 0, no branch
#pragma unroll
    for (int rk = 0; rk < R; ++rk) {
      const rt::rsrc_t Wb = rt::make_rsrc(p.w[rk], (size_t)H * L * 4);
      const rt::rsrc_t Bb = rt::make_rsrc(p.b[rk], (size_t)H * 4);
      bv[rk] = rt::ldg4(Bb, h < H ? (uint32_t)h * 4u : kOut, 0u);
#pragma unroll
      for (int k = 0; k < kCols / 32; ++k) {
        const int col = c0 + 32 * k;                       // even; L is even: the pair is inside the row or col >= L
        const uint32_t off = (h < H && col < L) ? ((uint32_t)h * (uint32_t)L + (uint32_t)col) * 4u : kOut;
        wv[rk][k] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(Wb, (int)off, 0, 0));
      }
    }
#pragma unroll
    for (int rk = 0; rk < R; ++rk)
#pragma unroll
      for (int k = 0; k < kCols / 32; ++k)
        if (c0 + 32 * k == L) wv[rk][k][0] = bv[rk];
#pragma unroll
    for (int rk = 0; rk < R; ++rk)
#pragma unroll
      for (int k = 0; k < kCols / 32; ++k)
        *reinterpret_cast<f32x2*>(Ws + (rk * 16 + hl) * kCols + c0 + 32 * k) = wv[rk][k];
  }
  __syncthreads();

  if (b_lo < b_hi) {

    f32x4 P[kSpans][2];
    unsigned long long cyc_steps = 0, cyc_fold = 0, tk0 = 0;
    if constexpr ((TUNE & 4) != 0) tk0 = __builtin_amdgcn_s_memrealtime();
    for (int b = b_lo; b < b_hi; ++b) {
      const int row0 = b * N;
      unsigned long long c0 = 0, c1 = 0;
      if constexpr ((TUNE & 4) != 0) c0 = __builtin_amdgcn_s_memtime();
      // question-side factors of the NEXT sample, a whole sample ahead of their use
      f32x4 h2n[R];
#pragma unroll
      for (int rk = 0; rk < R; ++rk) h2n[rk] = rt::ldg16(Hb, offH, (uint32_t)((min(b + 1, b_hi - 1) * R + rk) * H) * 4u);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        static_assert(NS % D == 0, "a ring slot must mean the same step in every sample");
        Step& st = ring[s % D];
        fix_ones(st);
#pragma unroll
        for (int q = 0; q < kSpans; ++q)
#pragma unroll
          for (int e = 0; e < 2; ++e)
            P[q][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(st.a, st.xb[q][e], s == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : P[q][e], 0, 0, 0);
        // refill the slot the PREVIOUS step consumed (step s - 1 + D of the row stream), under this step's MFMAs.  (At the
        // very first step of the slice that is a redundant reload of slot D - 1 with the rows it already holds.)
        load_step(ring[(s + D - 1) % D], row0 + 4 * (s - 1 + D));
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                                      // the ones-column selects
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int t = 0; t < kSpans + 1; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                    // VMEM read
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    // MFMA
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * kSpans - 1 - (kSpans + 1), 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr ((TUNE & 4) != 0) {
        c1 = __builtin_amdgcn_s_memtime();
        cyc_steps += c1 - c0;
      }
      // ---- fold this half of P_b into the rank accumulators; contract it against W1_r for dh2 ----
      f32x4 dh[R];
#pragma unroll
      for (int rk = 0; rk < R; ++rk) dh[rk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int q = 0; q < kSpans; ++q) {
        f32x2 wq[R][4];   // W1_r[h0 + 4 gq + t][64 q + 4 r + 2 eh, + 1]
#pragma unroll
        for (int rk = 0; rk < R; ++rk)
#pragma unroll
          for (int t = 0; t < 4; ++t)
            wq[rk][t] = *reinterpret_cast<const f32x2*>(Ws + (rk * 16 + 4 * gq + t) * kCols + 64 * q + 4 * r + 2 * eh);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const f32x4 pv = P[q][e];
#pragma unroll
          for (int rk = 0; rk < R; ++rk) {
            acc[rk][q][e] += h2v[rk] * pv;
#pragma unroll
            for (int t = 0; t < 4; ++t) dh[rk][t] = fmaf(wq[rk][t][e], pv[t], dh[rk][t]);
          }
        }
      }
      // this half's share of dh2[b][rk][h0 + 4 gq + t]: sum over the 16 lanes of a row (the columns).  A reduce-scatter
      // over the quad (lane r ends up with row t = r & 3), then two row rotations by 4 and 8 lanes
      {
        const int h = h0 + 4 * gq + (r & 3);
        const bool odd = (r & 1) != 0, hi = (r & 2) != 0;
#pragma unroll
        for (int rk = 0; rk < R; ++rk) {
          float k01 = odd ? dh[rk][1] : dh[rk][0], s01 = odd ? dh[rk][0] : dh[rk][1];
          float k23 = odd ? dh[rk][3] : dh[rk][2], s23 = odd ? dh[rk][2] : dh[rk][3];
          k01 += dpp_mov<0xB1>(s01);                       // lane ^ 1
          k23 += dpp_mov<0xB1>(s23);
          float v = hi ? k23 : k01;
          const float sv = hi ? k01 : k23;
          v += dpp_mov<0x4E>(sv);                          // lane ^ 2
          v += dpp_mov<0x124>(v);                          // row_ror:4
          v += dpp_mov<0x128>(v);                          // row_ror:8
          if (r < 4 && h < H) p.part[(size_t)eh * p.B * R * H + ((size_t)b * R + rk) * H + h] = v;
        }
      }
#pragma unroll
      for (int rk = 0; rk < R; ++rk) h2v[rk] = h2n[rk];
      if constexpr ((TUNE & 4) != 0) cyc_fold += __builtin_amdgcn_s_memtime() - c1;
    }
    if constexpr ((TUNE & 4) != 0) {
      if (lane == 0 && p.stamps != nullptr) {
        unsigned long long* st = p.stamps + (size_t)(blockIdx.x * 4 + wave) * 5;
        st[0] = cyc_steps;
        st[1] = cyc_fold;
        st[2] = __builtin_amdgcn_s_memrealtime() - tk0;
        st[3] = tk0 - tk_in;
      }
    }
  }

  // ---- the two sample slices of the workgroup meet in LDS: slice 1 -> box of its column half; slice 0 adds ----
  constexpr int NBLK = R * kSpans * 2;
  f32x4* box = reinterpret_cast<f32x4*>(smem_raw) + (size_t)eh * NBLK * 64 + lane;
  __syncthreads();            // everyone is done with Ws
  if (sl == 1) {
#pragma unroll
    for (int rk = 0; rk < R; ++rk)
#pragma unroll
      for (int q = 0; q < kSpans; ++q)
#pragma unroll
        for (int e = 0; e < 2; ++e) box[(size_t)((rk * kSpans + q) * 2 + e) * 64] = acc[rk][q][e];
  }
  __syncthreads();
  if (sl == 1) return;
#pragma unroll
  for (int rk = 0; rk < R; ++rk)
#pragma unroll
    for (int q = 0; q < kSpans; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) acc[rk][q][e] += box[(size_t)((rk * kSpans + q) * 2 + e) * 64];

  // slab[sg][rk][h][l] (l < L) and dbslab[sg][rk][h] (column L); this wave holds the columns 64 q + 4 r + 2 eh, + 1
#pragma unroll
  for (int rk = 0; rk < R; ++rk) {
    float* __restrict__ dst = p.slab + ((size_t)sg * R + rk) * H * L;
    float* __restrict__ dbd = p.dbslab + ((size_t)sg * R + rk) * H;
#pragma unroll
    for (int q = 0; q < kSpans; ++q) {
      const int col = 64 * q + 4 * r + 2 * eh;     // even, and L is even: the pair is either inside the row or col == L
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int h = h0 + 4 * gq + t;
        if (h >= H) continue;
        if (col + 1 < L)
          st2(dst + (size_t)h * L + col, make_float2(acc[rk][q][0][t], acc[rk][q][1][t]));
        else if (col == L)
          dbd[h] = acc[rk][q][0][t];
      }
    }
  }
  if constexpr ((TUNE & 4) != 0) {
    __builtin_amdgcn_s_waitcnt(0);
    if (lane == 0 && p.stamps != nullptr) p.stamps[(size_t)(blockIdx.x * 4 + wave) * 5 + 4] = __builtin_amdgcn_s_memrealtime() - tk_in;
  }
}

}  // namespace

bool dw_rt_supported(int B, int N, int L, int H, int R, int ldx) {
  const bool off = vqa::option_is("VQA_K4_DW_RT", '0');
  return !off && (R == 1 || R == 2) && (N == 36 || N == 100) && L > 256 && L < kCols && ldx == L && L % 2 == 0 && H >= 16 &&
         B >= 64 && (size_t)B * N * H * 4 < (1ull << 32);
}

// slab: [kDwRtGroups][R][H*L], dbslab: [kDwRtGroups][R][H] (summed by bilinear_dw_reduce_kernel); part: [2][B*R*H], the dh2
// partial sums of the two column halves (summed by bilinear_dh2_reduce_kernel)
int dw_rt_launch(const float* g, const float* x, const float* h2, const float* const* w1, const float* const* b1, float* slab,
                 float* dbslab, float* part, int B, int N, int L, int H, int R, hipStream_t s) {
  DwRtArgs a{};
  a.g = g;
  a.x = x;
  a.h2 = h2;
  for (int r = 0; r < R; ++r) {
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
  a.HB = (H + 15) / 16;
  a.spl = (B + 2 * kDwRtGroups - 1) / (2 * kDwRtGroups);
  const dim3 grid(a.HB * kDwRtGroups);
  const int tune = vqa::option("VQA_K4_DW_TUNE") ? std::atoi(vqa::option("VQA_K4_DW_TUNE")) : 0;   // (diagnostic only)
#define LAUNCH(R_, NS_)                                                                                         \
  {                                                                                                             \
    /* LDS: the W tile [R][16][320] floats, then 2 boxes of R x 10 blocks x 64 float4 for the cross-slice sum */ \
    const size_t lds_w = (size_t)R_ * 16 * kCols * 4, lds_b = (size_t)2 * R_ * kSpans * 2 * 64 * 16;            \
    const size_t lds = lds_w > lds_b ? lds_w : lds_b;                                                           \
    constexpr int D_ = NS_ == 9 ? 9 : 5;   /* ring depth in steps (divides the steps of a sample) */            \
    VQA_ENSURE_LDS((bilinear_dw_rt_kernel<R_, NS_, D_>), lds);                                                  \
    if (tune == 4 && R_ == 2 && NS_ == 9) {                                                                \
      static unsigned long long* st = nullptr;                                                                  \
      if (st == nullptr) (void)hipMalloc(&st, 1024 * 5 * 8);                                                    \
      a.stamps = st;                                                                                            \
      VQA_LAUNCH((bilinear_dw_rt_kernel<2, 9, 9, 4>), grid, dim3(rt::kThreads), lds, s, a);             \
      static int shown = 0;                                                                                     \
      if (++shown == 20) {                                                                                      \
        unsigned long long h[1024 * 5];                                                                         \
        (void)hipDeviceSynchronize();                                                                           \
        (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);                                               \
        double c0 = 0, c1 = 0, tk = 0, pro = 0, tot = 0, tkmax = 0, totmax = 0;                                 \
        int n0 = 0;                                                                                             \
        for (int i = 0; i < 1024; ++i) {                                                                        \
          c0 += h[5 * i]; c1 += h[5 * i + 1]; tk += h[5 * i + 2]; pro += h[5 * i + 3];                          \
          if ((double)h[5 * i + 2] > tkmax) tkmax = (double)h[5 * i + 2];                                       \
          if ((i & 3) < 2) { tot += h[5 * i + 4]; ++n0; if ((double)h[5 * i + 4] > totmax) totmax = (double)h[5 * i + 4]; } \
        }                                                                                                       \
        fprintf(stderr, "dw_rt stamps: steps %.0f cyc, fold %.0f cyc, loop %.2f us (max %.2f), prologue %.2f us, whole wave %.2f us (max %.2f)\n", \
                c0 / 1024, c1 / 1024, tk / 1024 / 100.0, tkmax / 100.0, pro / 1024 / 100.0, tot / n0 / 100.0, totmax / 100.0); \
      }                                                                                                         \
    }                                                                                                           \
    else                                                                                                        \
      VQA_LAUNCH((bilinear_dw_rt_kernel<R_, NS_, D_>), grid, dim3(rt::kThreads), lds, s, a);            \
  }
  if (R == 1) {
    if (N == 36) LAUNCH(1, 9) else LAUNCH(1, 25)
  } else {
    if (N == 36) LAUNCH(2, 9) else LAUNCH(2, 25)
  }
#undef LAUNCH
  return check_launch("lowrank_bilinear_fusion_folded_bwd (dW, register-tile)");
}

}  // namespace vqa
