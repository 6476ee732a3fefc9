// K2 -- object-difference attention logits (ODA).
//
// Replaces config/ODA.py:216-222 (a 36x36 python loop of (vi - vj) * q_low, torch.stack of 1296 [B,L]
// tensors, transpose, view -> vq [B,N,N*L], 1.6 MB per sample) and the dropout(0.5) + 1x1 conv over
// the 11160 channels of vq (config/ODA.py:149 with MyConv1d :89-105):
//
//   logits[b,i,g] = bias[g] + sum_{j,d} w[g,j*L+d] * keep(b,i,j,d) * (T[b,i,d] - T[b,j,d]),   T = vl * ql
//
// Neither vq nor its Bernoulli mask is ever written: every lane owns one feature d, builds the
// differences in registers, draws the mask from a counter-based hash of (seed, element index) -- one
// 32-bit word serves the four regions i = 4*iq .. 4*iq+3, one byte each, so p_drop is realised as
// round(p*256)/256 -- and contracts against the four filter rows immediately.  Backward regenerates
// the same mask from the same (seed, index).
//
// VALU-bound (N*N*L = 401 760 mask elements per sample); compulsory HBM traffic is only vl, ql, logits per sample plus
// the 178 KB filter, which stays in L2.
//
// Two mask layouts.  p = 0.5 (the rate the model uses, p8 == 128): ONE BIT per element -- keep iff bit (i & 31) of the
// hash word of counter ((b*NI + i/32)*N + j)*L + d, NI = ceil(N/32); a word serves up to 32 regions i of one (j, d), the
// kept difference is AND-ed with the sign-extended bit (2 lane-ops) and the factor 2 is applied once per output: ~8
// lane-ops per element (1 subtract, 2 mask, G = 4 FMAs, hash amortised).  Any other rate: one BYTE per element, four
// regions per word (~14 lane-ops per element: byte extract, compare, select, multiply + a hash per 4 elements).
#include <cstdlib>

#include <type_traits>
#include <utility>

#include "gemm_f32_rt.hpp"

namespace vqa {

typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kOdaMaxG = 8;
constexpr int kIC = 12;  // regions i (forward, dT) or j (dW) kept in registers per workgroup pass

// element (b, i, j, d): word index ((b*NQ + i/4)*N + j)*L + d, byte i%4
__device__ __forceinline__ uint64_t mask_counter(int b, int iq, int j, int d, int NQ, int N, int L) {
  return (((uint64_t)b * NQ + iq) * N + j) * L + d;
}
__device__ __forceinline__ float keep_scale(uint32_t word, int k, uint32_t p8, float scale) {
  return ((word >> (8 * k)) & 255u) >= p8 ? scale : 0.f;
}

// ---- p = 0.5: one bit per element --------------------------------------------------------------------------------
constexpr int kIB = 18;   // regions i per workgroup pass of the forward (two passes cover the reference's 36)
// the mask bits of regions i0 .. i0 + 31 of one (b, j, d), bit k <-> region i0 + k: word i0/32 shifted down, topped up
// from word i0/32 + 1 when the range straddles it (`straddle` is workgroup-uniform)
__device__ __forceinline__ uint32_t oda_bits(uint32_t cnt_lo, uint32_t word_stride, int sh, bool straddle, uint32_t key) {
  uint32_t bits = mask_word32(cnt_lo, key) >> sh;
  if (straddle) bits |= mask_word32(cnt_lo + word_stride, key) << (32 - sh);   // (straddle implies sh > 0)
  return bits;
}
// v & (bit K of bits ? ~0 : 0): v_bfe_i32 (bit K sign-extended) + v_and_b32, unscaled.  Inline assembly on purpose: written
// in C (0 - ((bits >> K) & 1), or the sbfe builtin) the compiler canonicalises the AND with a sign splat into a select and
// emits v_and + v_cmp + v_cndmask through VCC -- four issue slots and a hazard nop per element instead of two.
template <int K>
__device__ __forceinline__ float keep_bit(float v, uint32_t bits) {
  uint32_t m;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "n"(K));
  return __uint_as_float(__float_as_uint(v) & m);
}
// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop whose index is a constant expression
template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

// forward, bit mask: grid (ceil(N/kIB), B), lane <-> feature d
template <int G>
__global__ void oda_fwd_bits_kernel(const float* __restrict__ vl, const float* __restrict__ ql, const float* __restrict__ w,
                                    const float* __restrict__ bias, float* __restrict__ logits, DropCfg dc, int N, int L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red_s = reinterpret_cast<float*>(smem);  // [nwaves][kIB*G]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int b = blockIdx.y, i0 = blockIdx.x * kIB;
  const int NI = (N + 31) >> 5;
  const uint32_t key = drop_key(dc);
  const int sh = i0 & 31;
  const bool straddle = ((min(i0 + kIB, N) - 1) >> 5) != (i0 >> 5);
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const uint32_t base = ((uint32_t)b * NI + (uint32_t)(i0 >> 5)) * stride;
  const float* vlb = vl + (size_t)b * N * L;
  float acc[kIB][G];
#pragma unroll
  for (int ic = 0; ic < kIB; ++ic)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[ic][g] = 0.f;

  for (int d = tid; d < L; d += blockDim.x) {
    const float qd = ql[(size_t)b * L + d];
    float Ti[kIB];
#pragma unroll
    for (int ic = 0; ic < kIB; ++ic) Ti[ic] = (i0 + ic < N) ? vlb[(size_t)(i0 + ic) * L + d] * qd : 0.f;
    // region j + 1's operands are requested before region j's 18 x 7 VALU instructions (left to the compiler the loads sit
    // right in front of their use: with at most four waves per SIMD -- 117 registers -- an L2 round trip per region shows)
    float vn = vlb[d];
    float wn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) wn[g] = w[(size_t)g * N * L + d];
    for (int j = 0; j < N; ++j) {
      const float Tj = vn * qd;
      float wv[G];
#pragma unroll
      for (int g = 0; g < G; ++g) wv[g] = wn[g];
      const int jn = min(j + 1, N - 1);
      vn = vlb[(size_t)jn * L + d];
#pragma unroll
      for (int g = 0; g < G; ++g) wn[g] = w[((size_t)g * N + jn) * L + d];
      const uint32_t bits = oda_bits(base + (uint32_t)(j * L + d), stride, sh, straddle, key);
      static_for<kIB>([&](auto ic_) {
        constexpr int ic = decltype(ic_)::value;
        const float val = keep_bit<ic>(Ti[ic] - Tj, bits);
#pragma unroll
        for (int g = 0; g < G; ++g) acc[ic][g] = fmaf(wv[g], val, acc[ic][g]);
      });
    }
  }
#pragma unroll
  for (int ic = 0; ic < kIB; ++ic)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float r = wave_sum(acc[ic][g]);
      if (lane == 0) red_s[wave * (kIB * G) + ic * G + g] = r;
    }
  __syncthreads();
  for (int t = tid; t < kIB * G; t += blockDim.x) {
    const int ic = t / G, g = t % G;
    if (i0 + ic < N) {
      float sum = 0.f;
      for (int wv = 0; wv < nwaves; ++wv) sum += red_s[wv * (kIB * G) + t];
      logits[((size_t)b * N + i0 + ic) * G + g] = fmaf(dc.scale, sum, bias[g]);     // the kept values' factor 2
    }
  }
}

// ---- forward on the 4x4 matrix instruction ------------------------------------------------------------------------
// The G = 4 glimpse FMAs of every masked difference are a 4x4 outer product: v_mfma_f32_4x4x1_16B_f32 does sixteen of them
// per instruction (blocks k = lane >> 2; D[k][i][g] += A[k][i] B[k][g]; lane (k, r) supplies A[k][r] and B[k][r], 8 cycles),
// so the VALU is left with the 3 instructions that MAKE the masked difference (subtract, v_bfe_i32, v_and) instead of 7.
//   block k <-> feature d = 16 ds + k of the wave's current set of 16 features,  row r <-> region i = 4 ig + r of region
//   group ig,  column r <-> glimpse g = r:   acc[ig] += (keep (T_i[d] - T_j[d])) (outer) w[g][j][d]   for j = 0 .. N-1.
// A lane keeps T_i[d] of its 9 regions (N <= 36) in registers for a whole feature set; per j it loads T_j[d] and w[r][j][d].
// Mask words: ONE 32-bit hash word holds the bits of regions 0..31 of a (j, d) (a second one regions 32..35); the four
// lanes of a block need the same words, so lane r hashes for j = j0 + r and the quad exchanges them by DPP (quad_perm).
// A workgroup = one sample, its waves share the feature sets; the 16 blocks of a wave and the waves meet at the end.
typedef float oda_f32x4 __attribute__((ext_vector_type(4)));
constexpr int kOdaIG = 9;    // region groups of 4 (N <= 36)

template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xF, 0xF, true);
}

// Work units: (feature set, slice of the regions j) -- the logits are sums over both, so any split is exact.  JT slices of NJ
// regions each (NJ a multiple of 4); wave w of the workgroup takes units w, w + nwaves, ...
template <bool MASK, bool PK = false>
__global__ __launch_bounds__(512) void oda_fwd_mfma_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ logits, DropCfg dc, int N, int L, int G,
                                                           int JT, int NJ) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red_s = reinterpret_cast<float*>(smem);      // [nwaves][4 kOdaIG][4]
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k = lane >> 2, r = lane & 3;
  const int b = blockIdx.x;
  const int NI = (N + 31) >> 5;
  const uint32_t key = MASK ? drop_key(dc) : 0u;
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const float* vlb = vl + (size_t)b * N * L;
  oda_f32x4 acc[kOdaIG];
#pragma unroll
  for (int ig = 0; ig < kOdaIG; ++ig) acc[ig] = oda_f32x4{0.f, 0.f, 0.f, 0.f};

  // buffer addressing: a per-lane byte offset (feature column; glimpse row of w) + a scalar offset for the region row
  const rt::rsrc_t Vb = rt::make_rsrc(vlb, (size_t)N * L * 4);
  const rt::rsrc_t Wb = rt::make_rsrc(w, (size_t)G * N * L * 4);
  const int nsets = (L + 15) >> 4;
  // per feature set: the lane's column, its T_i (raw v values; q is multiplied in when the set starts) and the first group of
  // regions j -- all requested one SET ahead, so that only the wave's first set waits for memory
  struct SetRegs {
    float ti[kOdaIG], tj[4], wv[4], qd;
  };
  auto set_offsets = [&](int ds, uint32_t& vo, uint32_t& wo, int& dcl, bool& dok) {
    const int d = 16 * min(ds, nsets - 1) + k;
    dok = d < L && ds < nsets;
    dcl = d < L ? d : 0;
    vo = (uint32_t)dcl * 4u;
    wo = ((uint32_t)min(r, G - 1) * stride + (uint32_t)dcl) * 4u;
  };
  auto load_group = [&](float (&t)[4], float (&wg)[4], uint32_t vo, uint32_t wo, int j0) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const uint32_t so = (uint32_t)min(j0 + jj, N - 1) * (uint32_t)L * 4u;
      t[jj] = rt::ldg4(Vb, vo, so);
      wg[jj] = rt::ldg4(Wb, wo, so);
    }
  };
  const int nunits = nsets * JT;
  auto load_set = [&](SetRegs& sr, int u) {
    uint32_t vo, wo;
    int dcl;
    bool dok;
    const int ds = u / JT;
    set_offsets(ds, vo, wo, dcl, dok);
    sr.qd = ql[(size_t)b * L + dcl];
#pragma unroll
    for (int ig = 0; ig < kOdaIG; ++ig) sr.ti[ig] = rt::ldg4(Vb, vo + (uint32_t)min(4 * ig + r, N - 1) * (uint32_t)L * 4u, 0u);
    load_group(sr.tj, sr.wv, vo, wo, min((u - ds * JT) * NJ, N - 1));
  };
  SetRegs nx;
  if (wave < nunits) load_set(nx, wave);
  for (int u = wave; u < nunits; u += nwaves) {
    uint32_t vo, wo;
    int dcl;
    bool dok;
    const int ds = u / JT;
    const int jlo = (u - ds * JT) * NJ, jhi = min(N, jlo + NJ);
    set_offsets(ds, vo, wo, dcl, dok);
    const float qd = dok ? nx.qd : 0.f;
    float Ti[kOdaIG], tj[4], wv[4];
#pragma unroll
    for (int ig = 0; ig < kOdaIG; ++ig) Ti[ig] = 4 * ig + r < N ? nx.ti[ig] * qd : 0.f;      // (qd = 0 beyond L)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      tj[jj] = nx.tj[jj];
      wv[jj] = nx.wv[jj];
    }
    load_set(nx, u + nwaves);                                // (past the last unit: a clamped reload nobody uses)
    const bool gok = dok && r < G;
    const uint32_t cnt_d = (uint32_t)b * (uint32_t)NI * stride + (uint32_t)dcl;   // + j L (+ stride for the second word)
    for (int j0 = jlo; j0 < jhi; j0 += 4) {
      float tn[4], wn[4];
      load_group(tn, wn, vo, wo, min(j0 + 4, N - 1));
      uint32_t hw0 = 0u, hw1 = 0u;
      if constexpr (MASK) {
        const uint32_t cnt = cnt_d + (uint32_t)min(j0 + r, N - 1) * (uint32_t)L;
        hw0 = mask_word32(cnt, key);
        if (NI > 1) hw1 = mask_word32(cnt + stride, key);
      }
      static_for<4>([&](auto jj_) {
        constexpr int jj = decltype(jj_)::value;
        constexpr int kCtrl = jj * 0x55;                     // quad_perm [jj, jj, jj, jj]
        const uint32_t b0 = dpp_u32<kCtrl>(hw0) >> r, b1 = dpp_u32<kCtrl>(hw1) >> r;   // bit 4 ig <-> region 4 ig + r
        const float Tj = tj[jj] * qd;
        const float wg = (gok && j0 + jj < N) ? wv[jj] : 0.f;                             // (a region beyond N adds nothing)
        // the nine masked differences of this j in one burst, then nine independent MFMAs back to back: an MFMA that reads
        // a register the VALU has just written stalls on it (two wait states each in the interleaved order)
        float x[kOdaIG];
        if constexpr (PK) {      // the nine differences as four v_pk_add_f32 + one v_sub_f32 (fewer issue slots)
          const f32x2 tj2 = f32x2{Tj, Tj};
#pragma unroll
          for (int h = 0; h < kOdaIG / 2; ++h) {
            const f32x2 d2 = f32x2{Ti[2 * h], Ti[2 * h + 1]} - tj2;
            x[2 * h] = d2[0];
            x[2 * h + 1] = d2[1];
          }
          x[kOdaIG - 1] = Ti[kOdaIG - 1] - Tj;
        }
        static_for<kOdaIG>([&](auto ig_) {
          constexpr int ig = decltype(ig_)::value;
          if constexpr (!PK) x[ig] = Ti[ig] - Tj;
          if constexpr (MASK) x[ig] = ig < 8 ? keep_bit<(4 * ig) & 31>(x[ig], b0) : keep_bit<0>(x[ig], b1);
        });
        __builtin_amdgcn_sched_group_barrier(0x002, MASK ? 3 * kOdaIG + 6 : kOdaIG + 2, 0);
        static_for<kOdaIG>([&](auto ig_) {
          constexpr int ig = decltype(ig_)::value;
          acc[ig] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[ig], wg, acc[ig], 0, 0, 0);
        });
        __builtin_amdgcn_sched_group_barrier(0x008, kOdaIG, 0);
      });
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        tj[jj] = tn[jj];
        wv[jj] = wn[jj];
      }
    }
  }
  // the 16 blocks of the wave: blocks 0..3 of a 16-lane row by two rotations, the four rows lane-wise
#pragma unroll
  for (int ig = 0; ig < kOdaIG; ++ig)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float v = acc[ig][t];
      v += dpp_mov<0x124>(v);      // row_ror:4
      v += dpp_mov<0x128>(v);      // row_ror:8
      v = rows_sum(v);
      if (lane < 4) red_s[(wave * 4 * kOdaIG + 4 * ig + t) * 4 + lane] = v;       // region 4 ig + t, glimpse lane
    }
  __syncthreads();
  for (int t = tid; t < N * G; t += blockDim.x) {
    const int i = t / G, g = t % G;
    float sum = 0.f;
    for (int wv_ = 0; wv_ < nwaves; ++wv_) sum += red_s[(wv_ * 4 * kOdaIG + i) * 4 + g];
    logits[((size_t)b * N + i) * G + g] = fmaf(MASK ? dc.scale : 1.f, sum, bias[g]);    // the kept values' factor 2
  }
}

// ---- weight gradient on the 4x4 matrix instruction ------------------------------------------------------------------
//   dw[g][j][d] = scale sum_{b,i} dS[b,i,g] keep(b,i,j,d) (T_i[d] - T_j[d])
// block k <-> feature d = 16 ds + k, row r <-> glimpse g = r (A = dS[b][i][r]), column r <-> region j = 4 jg + r of region group
// jg (B = the masked difference of the lane's own j):  acc[jg] += dS[b][i][.] (outer) X[i][4 jg + .][d]  for every sample b
// of the group and region i.  A lane keeps T_j[d] of its 9 regions and their mask words for the sample (a word holds the
// bits of all regions i of one (j, d): hashed once per sample, the bit picked per i with a register offset).
// Workgroup = 4 waves = one group of samples, the 20 feature sets of L = 310 split 5 per wave: 256 groups of 2 samples at
// B = 512 = one workgroup per CU, the same work for every wave.  slab[sg][g][j][d], summed by oda_reduce_kernel.
template <bool MASK, bool PK = false>
__global__ __launch_bounds__(512) void oda_bwd_weight_mfma_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                                                  const float* __restrict__ dS, float* __restrict__ slab,
                                                                  DropCfg dc, int B, int N, int L, int G,
                                                                  int samples_per_group) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ex_s = reinterpret_cast<float*>(smem);            // [nwaves][4 kOdaIG][64]: the odd half's accumulators of a set
  // 2 nwaves waves: wave = (sample slot sw, feature-set quarter): the two slots take alternate samples of the group and meet
  // in LDS after every set -- two waves per SIMD (one alone shows every dependent-issue latency: 89 us against 5x us)
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 7;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform for the compiler too: scalar offsets below)
  const int wave = wid % nwaves, sw = wid / nwaves;
  const int k = lane >> 2, r = lane & 3;
  const int sg = blockIdx.x;
  const int NI = (N + 31) >> 5;
  const uint32_t key = MASK ? drop_key(dc) : 0u;
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const int b_lo = sg * samples_per_group, b_hi = min(B, b_lo + samples_per_group);
  const rt::rsrc_t Vb = rt::make_rsrc(vl, (size_t)B * N * L * 4);
  const rt::rsrc_t Sb = rt::make_rsrc(dS, (size_t)B * N * G * 4);
  const int nsets = (L + 15) >> 4;
  const int nb_all = b_hi - b_lo;
  const int nb = (nb_all - sw + 1) / 2;                      // samples b_lo + sw, b_lo + sw + 2, ... of this slot
  oda_f32x4 acc[kOdaIG];
  const int max_sets = (nsets + nwaves - 1) / nwaves;       // every wave runs the same number of set rounds (barriers inside)
  for (int si = 0; si < max_sets; ++si) {
    const int ds = wave + si * nwaves;
    const int d = 16 * ds + k;
    const bool set_ok = ds < nsets;
    const bool dok = set_ok && d < L;
    const int dcl = dok ? d : 0;
    const uint32_t vo = (uint32_t)dcl * 4u;
#pragma unroll
    for (int jg = 0; jg < kOdaIG; ++jg) acc[jg] = oda_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int bi = 0; bi < (set_ok ? nb : 0); ++bi) {
      const int b = b_lo + sw + 2 * bi;
      const uint32_t row0 = (uint32_t)b * stride * 4u;                    // byte offset of vl[b][0][0]
      const float qd = dok ? ql[(size_t)b * L + dcl] : 0.f;
      // regions i in groups of four: T_i[d] (one value per block) and the glimpse gradients dS[b][i][r], a group ahead
      float ta[4], sa[4];
      auto load_group = [&](float (&t)[4], float (&a)[4], int i0) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int ic = min(i0 + ii, N - 1);
          t[ii] = rt::ldg4(Vb, vo, row0 + (uint32_t)ic * (uint32_t)L * 4u);
          a[ii] = rt::ldg4(Sb, (uint32_t)min(r, G - 1) * 4u, (uint32_t)(b * N + ic) * (uint32_t)G * 4u);
        }
      };
      load_group(ta, sa, 0);
      float Tj[kOdaIG];
      uint32_t w0[kOdaIG], w1[kOdaIG];
#pragma unroll
      for (int jg = 0; jg < kOdaIG; ++jg) {
        const int j = min(4 * jg + r, N - 1);
        Tj[jg] = rt::ldg4(Vb, vo + (uint32_t)j * (uint32_t)L * 4u, row0) * qd;
        asm("" : "+v"(Tj[jg]));             // (a ROUNDED product on both sides of T_i - T_j -- no fma contraction --: T_i - T_i is
                                            //  exactly zero, as in the reference)
        if constexpr (MASK) {
          const uint32_t cnt = (uint32_t)b * (uint32_t)NI * stride + (uint32_t)j * (uint32_t)L + (uint32_t)dcl;
          w0[jg] = mask_word32(cnt, key);
          w1[jg] = NI > 1 ? mask_word32(cnt + stride, key) : 0u;
        }
      }
      // (a loop, not nine unrolled groups: with constant bit indices the code is 9x as long and ran 76 us against 72)
      for (int i0 = 0; i0 < N; i0 += 4) {
        float tn[4], sn[4];
        load_group(tn, sn, min(i0 + 4, N - 1));
        // the mask words of this group of regions, shifted so that region i0 + ii is bit ii
        uint32_t ws[kOdaIG];
        if constexpr (MASK) {
          const bool hi = i0 >= 32;                                        // (uniform)
          const uint32_t sh = (uint32_t)(i0 & 31);
#pragma unroll
          for (int jg = 0; jg < kOdaIG; ++jg) ws[jg] = (hi ? w1[jg] : w0[jg]) >> sh;
        }
        static_for<4>([&](auto ii_) {
          constexpr int ii = decltype(ii_)::value;
          float Ti = ta[ii] * qd;
          asm("" : "+v"(Ti));
          const float av = (r < G && dok && i0 + ii < N) ? sa[ii] : 0.f;   // (zero beyond N, G, L: such terms add nothing)
          float x[kOdaIG];
          if constexpr (PK) {     // the nine differences as four v_pk_add_f32 + one v_sub_f32 (see the forward kernel)
            const f32x2 ti2 = f32x2{Ti, Ti};
#pragma unroll
            for (int h = 0; h < kOdaIG / 2; ++h) {
              const f32x2 d2 = ti2 - f32x2{Tj[2 * h], Tj[2 * h + 1]};
              x[2 * h] = d2[0];
              x[2 * h + 1] = d2[1];
            }
            x[kOdaIG - 1] = Ti - Tj[kOdaIG - 1];
          }
          static_for<kOdaIG>([&](auto jg_) {
            constexpr int jg = decltype(jg_)::value;
            if constexpr (!PK) x[jg] = Ti - Tj[jg];
            if constexpr (MASK) x[jg] = keep_bit<ii>(x[jg], ws[jg]);
          });
          __builtin_amdgcn_sched_group_barrier(0x002, MASK ? 3 * kOdaIG + 2 : kOdaIG + 2, 0);
#pragma unroll
          for (int jg = 0; jg < kOdaIG; ++jg) acc[jg] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, x[jg], acc[jg], 0, 0, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, kOdaIG, 0);
        });
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          ta[ii] = tn[ii];
          sa[ii] = sn[ii];
        }
      }
    }
    // the two sample slots of a set meet: slot 1 hands its accumulators over, slot 0 adds and stores
    float* ex = ex_s + (size_t)wave * (4 * kOdaIG) * 64 + lane;
    if (sw == 1) {
#pragma unroll
      for (int jg = 0; jg < kOdaIG; ++jg)
#pragma unroll
        for (int t = 0; t < 4; ++t) ex[(4 * jg + t) * 64] = acc[jg][t];
    }
    __syncthreads();
    if (sw == 0 && dok) {
      // acc[jg][t]: glimpse t, region 4 jg + r, feature d
      const float sc = MASK ? dc.scale : 1.f;
#pragma unroll
      for (int jg = 0; jg < kOdaIG; ++jg) {
        const int j = 4 * jg + r;
        if (j < N) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < G) slab[(((size_t)sg * G + t) * N + j) * L + d] = (acc[jg][t] + ex[(4 * jg + t) * 64]) * sc;
        }
      }
    }
    __syncthreads();
  }
}

// The same product with NO memory access inside the region loop (round 6).  The kernel above fetches T_i[d] and dS[b][i][.] of the
// next four regions with eight buffer loads per pass of its i loop; the compiler sinks them to the end of the pass and waits
// for them at the top of the next -- every pass pays a first-touch latency (a wave reads its own 64-byte piece of 36 rows
// of vl per set: nothing it touches was touched before) and the counters show it: 37 % of the SIMD cycles issue a VALU
// instruction, 14 % an MFMA.  But the wave already HOLDS those values: lane (k, r) keeps T_j[d] of regions j = 4 jg + r -- all
// 36 regions of the wave's 16 features between the four lanes of a block.  It writes them once per (set, sample) to a
// wave-private piece of LDS, [16 features][kOdaTS regions], next to the sample's dS as [4 glimpses][kOdaTS] (zero beyond N
// and G), and the region loop reads four T_i and four dS with two ds_read_b128 (same address across a quad: broadcast).
// The raw vl / dS / ql values of the NEXT (set, sample) are loaded while the current one is multiplied.  Same operands,
// same order of accumulation: bit-identical to the kernel above (tests/test_gpu_oda.py).
constexpr int kOdaTS = 40;   // floats per row of the staged tiles (36 regions + 4: rows stay 16-byte aligned)
// v & (bit `off` of bits ? ~0 : 0), the bit index in a scalar register (wave-uniform): the region loop hands the index of its
// region instead of shifting nine mask words per pass
__device__ __forceinline__ float keep_bit_at(float v, uint32_t bits, int off) {
  uint32_t m;
  asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(bits), "s"(off));
  return __uint_as_float(__float_as_uint(v) & m);
}
// JH = 2: the nine region groups of a lane split 5 + 4 over two waves -- 16 waves per workgroup, four per SIMD, under 128
// registers each (36 -> 20 accumulators); every wave still stages the whole T tile for itself.  Two waves per SIMD issue
// during half of their cycles only (counters: SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES), whatever is staged.
template <bool MASK, int JH>
__global__ __launch_bounds__(512 * JH) void oda_bwd_weight_mfma_staged_kernel(const float* __restrict__ vl,
                                                                              const float* __restrict__ ql,
                                                                              const float* __restrict__ dS, float* __restrict__ slab,
                                                                              DropCfg dc, int B, int N, int L, int G,
                                                                              int samples_per_group) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ex_s = reinterpret_cast<float*>(smem);            // [2][nwaves][4 kOdaIG][64]: the odd slot's accumulators of a set,
                                                           // two buffers in turn (one barrier per set)
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x / (128 * JH);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wid % nwaves, sw = (wid / nwaves) & 1, jh = wid / (2 * nwaves);
  const int k = lane >> 2, r = lane & 3;
  float* Tw = ex_s + (size_t)2 * nwaves * (4 * kOdaIG) * 64 + (size_t)wid * (20 * kOdaTS);   // [16][kOdaTS], wave-private
  float* Sw = Tw + 16 * kOdaTS;                                                               // [4][kOdaTS]
  const int sg = blockIdx.x;
  const int NI = (N + 31) >> 5;
  const uint32_t key = MASK ? drop_key(dc) : 0u;
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const int b_lo = sg * samples_per_group, b_hi = min(B, b_lo + samples_per_group);
  const rt::rsrc_t Vb = rt::make_rsrc(vl, (size_t)B * N * L * 4);
  const rt::rsrc_t Sb = rt::make_rsrc(dS, (size_t)B * N * G * 4);
  const rt::rsrc_t Qb = rt::make_rsrc(ql, (size_t)B * L * 4);
  const rt::rsrc_t Wb = rt::make_rsrc(slab + (size_t)sg * G * N * L, (size_t)G * N * L * 4);   // this group's slab
  const int nsets = (L + 15) >> 4;
  const int nb = (b_hi - b_lo - sw + 1) / 2;                  // samples b_lo + sw, b_lo + sw + 2, ... of this slot
  const int max_sets = (nsets + nwaves - 1) / nwaves;         // every wave runs the same number of set rounds (barriers inside)
  // what a (set, sample) needs from memory: vl of the lane's nine regions, the lane's three entries of the [36][4] dS tile, ql[d]
  struct Raw {
    float v[kOdaIG], s[3], q;
  };
  auto fetch = [&](Raw& o, int si, int bi) {
    const int ds = wave + si * nwaves;
    const int d = 16 * ds + k;
    const uint32_t vo = (uint32_t)((ds < nsets && d < L) ? d : 0) * 4u;
    const int b = b_lo + sw + 2 * bi;
    const uint32_t row0 = (uint32_t)b * stride * 4u;
#pragma unroll
    for (int jg = 0; jg < kOdaIG; ++jg) o.v[jg] = rt::ldg4(Vb, vo + (uint32_t)min(4 * jg + r, N - 1) * (uint32_t)L * 4u, row0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int e = lane + 64 * t, i = e >> 2, g = e & 3;
      const float a = rt::ldg4(Sb, (uint32_t)(min(i, N - 1) * G + min(g, G - 1)) * 4u, (uint32_t)b * (uint32_t)(N * G) * 4u);
      o.s[t] = (i < N && g < G) ? a : 0.f;
    }
    o.q = rt::ldg4(Qb, vo, (uint32_t)b * (uint32_t)L * 4u);
  };
  auto arrived = [&](Raw& o) {
#pragma unroll
    for (int jg = 0; jg < kOdaIG; ++jg) asm volatile("" : "+v"(o.v[jg]));
    asm volatile("" : "+v"(o.s[0]), "+v"(o.s[1]), "+v"(o.s[2]), "+v"(o.q));
  };
  // region groups J0 .. J1 - 1 of the lane (all nine, or this wave's half)
  auto run = [&](auto j0_, auto j1_) {
    constexpr int J0 = decltype(j0_)::value, J1 = decltype(j1_)::value, NJ = J1 - J0;
    Raw cur;
    if (nb > 0 && wave < nsets) {
      fetch(cur, 0, 0);
      arrived(cur);      // (before the loops start: a wait for them inside would also wait for the stores of the set before)
    }
    oda_f32x4 acc[NJ];
    for (int si = 0; si < max_sets; ++si) {
      const int ds = wave + si * nwaves;
      const int d = 16 * ds + k;
      const bool set_ok = ds < nsets;
      const bool dok = set_ok && d < L;
      const int dcl = dok ? d : 0;
#pragma unroll
      for (int jg = 0; jg < NJ; ++jg) acc[jg] = oda_f32x4{0.f, 0.f, 0.f, 0.f};
      for (int bi = 0; bi < (set_ok ? nb : 0); ++bi) {
        const int b = b_lo + sw + 2 * bi;
        // the next (set, sample) of this wave, in flight while this one is multiplied
        Raw nxt;
        {
          const bool last_b = bi + 1 >= nb;
          const int si2 = last_b ? si + 1 : si, bi2 = last_b ? 0 : bi + 1;
          if (si2 < max_sets && wave + si2 * nwaves < nsets) fetch(nxt, si2, bi2);
        }
        float Tj[NJ];
        uint32_t w0[NJ], w1[NJ];
#pragma unroll
        for (int jg = 0; jg < kOdaIG; ++jg) {
          float t = cur.v[jg] * cur.q;
          asm("" : "+v"(t));                  // (a ROUNDED product on both sides of T_i - T_j, as above)
          Tw[k * kOdaTS + 4 * jg + r] = t;    // (regions beyond N: copies of region N - 1, finite, times dS = 0)
          if (jg >= J0 && jg < J1) {
            Tj[jg - J0] = t;
            if constexpr (MASK) {
              const int j = min(4 * jg + r, N - 1);
              const uint32_t cnt = (uint32_t)b * (uint32_t)NI * stride + (uint32_t)j * (uint32_t)L + (uint32_t)dcl;
              w0[jg - J0] = mask_word32(cnt, key);
              w1[jg - J0] = NI > 1 ? mask_word32(cnt + stride, key) : 0u;
            }
          }
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int e = lane + 64 * t;
          if (e < 4 * 36) Sw[(e & 3) * kOdaTS + (e >> 2)] = cur.s[t];
        }
        // (wave-private tiles: the wave's own LDS operations complete in order, no barrier)
        oda_f32x4 ta = *reinterpret_cast<const oda_f32x4*>(Tw + k * kOdaTS);
        oda_f32x4 sa = *reinterpret_cast<const oda_f32x4*>(Sw + r * kOdaTS);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), once per (set, sample): see the end of a pass
        // four regions i0 .. i0 + 3; region i0 + ii is bit bit0 + ii of the words wv
        auto pass = [&](int i0, const uint32_t (&wv)[NJ], int bit0) {
          const int in = min(i0 + 4, 32);
          const oda_f32x4 tn = *reinterpret_cast<const oda_f32x4*>(Tw + k * kOdaTS + in);
          const oda_f32x4 sn = *reinterpret_cast<const oda_f32x4*>(Sw + r * kOdaTS + in);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);      // (the two reads FIRST: left alone they sink to the end of the
                                                                  //  pass and every pass waits for them)
          static_for<4>([&](auto ii_) {
            constexpr int ii = decltype(ii_)::value;
            const float Ti = ta[ii];
            const float av = sa[ii];          // (zero beyond N and G; a lane beyond L accumulates finite values nobody stores)
            float x[NJ];
            const f32x2 ti2 = f32x2{Ti, Ti};
#pragma unroll
            for (int h = 0; h < NJ / 2; ++h) {
              const f32x2 d2 = ti2 - f32x2{Tj[2 * h], Tj[2 * h + 1]};
              x[2 * h] = d2[0];
              x[2 * h + 1] = d2[1];
            }
            if constexpr (NJ & 1) x[NJ - 1] = Ti - Tj[NJ - 1];
            if constexpr (MASK) {
#pragma unroll
              for (int jg = 0; jg < NJ; ++jg) x[jg] = keep_bit_at(x[jg], wv[jg], bit0 + ii);
            }
            __builtin_amdgcn_sched_group_barrier(0x002, (MASK ? 2 * NJ : 0) + NJ / 2 + (NJ & 1), 0);
#pragma unroll
            for (int jg = 0; jg < NJ; ++jg) acc[jg] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, x[jg], acc[jg], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, NJ, 0);
          });
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the two reads, issued a pass ago, for the compiler's bookkeeping too
                                               // (it cannot count across the back edge and would wait at the top, behind the
                                               //  reads it has just issued)
          ta = tn;
          sa = sn;
        };
        const int n_lo = min(N, 32);
        for (int i0 = 0; i0 < n_lo; i0 += 4) pass(i0, w0, i0);
        if (N > 32) pass(32, w1, 0);
        cur = nxt;
        // (in registers HERE, a whole (set, sample) after the loads were issued: left to the compiler the wait moves behind
        //  the stores of the set's epilogue and waits for them as well)
        arrived(cur);
      }
      // the two sample slots of a set meet: slot 1 hands its accumulators over, slot 0 adds and stores
      float* ex = ex_s + ((size_t)(si & 1) * nwaves + wave) * (4 * kOdaIG) * 64 + lane;
      if (sw == 1) {
#pragma unroll
        for (int jg = 0; jg < NJ; ++jg)
#pragma unroll
          for (int t = 0; t < 4; ++t) ex[(4 * (J0 + jg) + t) * 64] = acc[jg][t];
      }
      __syncthreads();
      if (sw == 0 && dok) {
        const float sc = MASK ? dc.scale : 1.f;
        float e[NJ][4];
#pragma unroll
        for (int jg = 0; jg < NJ; ++jg)
#pragma unroll
          for (int t = 0; t < 4; ++t) e[jg][t] = ex[(4 * (J0 + jg) + t) * 64];      // (all reads in flight, then the stores)
#pragma unroll
        for (int jg = 0; jg < NJ; ++jg) {
          const int j = 4 * (J0 + jg) + r;
          if (j < N) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
              if (t < G)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (acc[jg][t] + e[jg][t]) * sc), Wb,
                                                      (int)(((uint32_t)(t * N + j) * (uint32_t)L + (uint32_t)d) * 4u), 0, 0);
          }
        }
      }
      // (no second barrier: the next set hands over in the other buffer, and its barrier is behind these reads)
    }
  };
  using std::integral_constant;
  if constexpr (JH == 1) {
    run(integral_constant<int, 0>{}, integral_constant<int, kOdaIG>{});
  } else {
    constexpr int kHalf = (kOdaIG + 1) / 2;
    if (jh == 0) run(integral_constant<int, 0>{}, integral_constant<int, kHalf>{});
    else run(integral_constant<int, kHalf>{}, integral_constant<int, kOdaIG>{});
  }
}

// ---- the whole backward in ONE pass over the mask (round 6) ---------------------------------------------------------------
// The data gradient (oda_bwd_data_bits_split_kernel) and the weight gradient above each regenerate the mask and each walk
// the 401 760 (i, j, d) elements of a sample; both are VALU-issue bound (counters: 88 % and 64-99 % of the SIMD cycles).  One
// kernel in the weight gradient's layout does both: a lane that holds the masked difference x = keep (T_i - T_j) of its
// (j, d) for the matrix instruction also forms u = sum_g dS[b,i,g] w[g,j,d] (the filter rows of its nine regions live in 36
// registers for a whole feature set, dS[b][i][0..3] comes from LDS as one broadcast read) and masks it with the SAME
// sign-extended bit:
//   dT[j][d] -= keep u  -- a register per region of the lane, summed over i;
//   dT[i][d] += keep u  -- summed over the lane's nine regions in registers, over the four lanes of the block by DPP, one value
//                          per lane and pass written to a wave-private LDS row;
// and a (set, sample) ends with d_vl = dT ql (gated) and d_ql = sum_n dT[n] vl[n] for its 16 features -- complete, no other wave
// contributes.  Per element: 62 / 9 lane-operations + the MFMA against 23 / 9 + MFMA here and 6.7 in the data kernel.
// The weight gradient's arithmetic and order are those of the staged kernel (bit-identical); the data gradient sums in a
// different (fixed) order than the kernel it replaces.  MEASURED SLOWER than the two kernels (154 against 140 us for the
// backward at B = 512: 256 registers, two waves per SIMD, ~60 % issue) -- kept behind VQA_K2_FUSED=1 as the record of the probe.
template <bool MASK>
__global__ __launch_bounds__(512) void oda_bwd_fused_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                                            const float* __restrict__ w, const float* __restrict__ dS,
                                                            float* __restrict__ d_vl, float* __restrict__ d_ql,
                                                            float* __restrict__ slab, DropCfg dc, int B, int N, int L, int G,
                                                            int samples_per_group, int gate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ex_s = reinterpret_cast<float*>(smem);            // [2][nwaves][4 kOdaIG][64]: see the staged kernel
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 7;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wid % nwaves, sw = wid / nwaves;
  const int k = lane >> 2, r = lane & 3;
  constexpr int kWaveTile = (16 + 4 + 16 + 16) * kOdaTS + 36 * 4;
  float* Tw = ex_s + (size_t)2 * nwaves * (4 * kOdaIG) * 64 + (size_t)wid * kWaveTile;   // [16][kOdaTS]  T_i[d]
  float* Sw = Tw + 16 * kOdaTS;                                                          // [4][kOdaTS]   dS[i][g] as [g][i]
  float* Dw = Sw + 4 * kOdaTS;                                                           // [16][kOdaTS]  sum_j keep u, per i
  float* Vw = Dw + 16 * kOdaTS;                                                          // [16][kOdaTS]  vl[n][d] as loaded
  float* Sa = Vw + 16 * kOdaTS;                                                          // [36][4]       dS[i][0..3]
  const int sg = blockIdx.x;
  const int NI = (N + 31) >> 5;
  const uint32_t key = MASK ? drop_key(dc) : 0u;
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const int b_lo = sg * samples_per_group, b_hi = min(B, b_lo + samples_per_group);
  const rt::rsrc_t Vb = rt::make_rsrc(vl, (size_t)B * N * L * 4);
  const rt::rsrc_t Sb = rt::make_rsrc(dS, (size_t)B * N * G * 4);
  const rt::rsrc_t Qb = rt::make_rsrc(ql, (size_t)B * L * 4);
  const rt::rsrc_t Fb = rt::make_rsrc(w, (size_t)G * N * L * 4);
  const rt::rsrc_t Wb = rt::make_rsrc(slab + (size_t)sg * G * N * L, (size_t)G * N * L * 4);   // this group's slab
  const rt::rsrc_t DVb = rt::make_rsrc(d_vl, (size_t)B * N * L * 4);
  const int nsets = (L + 15) >> 4;
  const int nb = (b_hi - b_lo - sw + 1) / 2;                  // samples b_lo + sw, b_lo + sw + 2, ... of this slot
  const int max_sets = (nsets + nwaves - 1) / nwaves;
  struct Raw {
    float v[kOdaIG], s[3], q;
  };
  auto fetch = [&](Raw& o, int si, int bi) {
    const int ds = wave + si * nwaves;
    const int d = 16 * ds + k;
    const uint32_t vo = (uint32_t)((ds < nsets && d < L) ? d : 0) * 4u;
    const int b = b_lo + sw + 2 * bi;
    const uint32_t row0 = (uint32_t)b * stride * 4u;
#pragma unroll
    for (int jg = 0; jg < kOdaIG; ++jg) o.v[jg] = rt::ldg4(Vb, vo + (uint32_t)min(4 * jg + r, N - 1) * (uint32_t)L * 4u, row0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int e = lane + 64 * t, i = e >> 2, g = e & 3;
      const float a = rt::ldg4(Sb, (uint32_t)(min(i, N - 1) * G + min(g, G - 1)) * 4u, (uint32_t)b * (uint32_t)(N * G) * 4u);
      o.s[t] = (i < N && g < G) ? a : 0.f;
    }
    o.q = rt::ldg4(Qb, vo, (uint32_t)b * (uint32_t)L * 4u);
  };
  auto arrived = [&](Raw& o) {
#pragma unroll
    for (int jg = 0; jg < kOdaIG; ++jg) asm volatile("" : "+v"(o.v[jg]));
    asm volatile("" : "+v"(o.s[0]), "+v"(o.s[1]), "+v"(o.s[2]), "+v"(o.q));
  };
  constexpr int NP = kOdaIG / 2;       // pairs of region groups (packed operations) + one single
  Raw cur;
  if (nb > 0 && wave < nsets) {
    fetch(cur, 0, 0);
    arrived(cur);
  }
  oda_f32x4 acc[kOdaIG];
  for (int si = 0; si < max_sets; ++si) {
    const int ds = wave + si * nwaves;
    const int d = 16 * ds + k;
    const bool set_ok = ds < nsets;
    const bool dok = set_ok && d < L;
    const int dcl = dok ? d : 0;
#pragma unroll
    for (int jg = 0; jg < kOdaIG; ++jg) acc[jg] = oda_f32x4{0.f, 0.f, 0.f, 0.f};
    // the filter rows of the lane's regions, this feature set: F2[g][h] = (w[g][8h + r], w[g][8h + 4 + r]), F1[g] = w[g][32 + r];
    // zero for a region beyond N and a glimpse beyond G (such u are zero and add nothing to either sum)
    f32x2 F2[4][NP];
    float F1[4];
    if (set_ok && nb > 0) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float f[kOdaIG];
#pragma unroll
        for (int jg = 0; jg < kOdaIG; ++jg) {
          const int j = 4 * jg + r;
          const float x = rt::ldg4(Fb, (uint32_t)((min(g, G - 1) * N + min(j, N - 1)) * L + dcl) * 4u, 0u);
          f[jg] = (g < G && j < N) ? x : 0.f;
        }
#pragma unroll
        for (int h = 0; h < NP; ++h) F2[g][h] = f32x2{f[2 * h], f[2 * h + 1]};
        F1[g] = f[kOdaIG - 1];
      }
    }
    for (int bi = 0; bi < (set_ok ? nb : 0); ++bi) {
      const int b = b_lo + sw + 2 * bi;
      Raw nxt;
      {
        const bool last_b = bi + 1 >= nb;
        const int si2 = last_b ? si + 1 : si, bi2 = last_b ? 0 : bi + 1;
        if (si2 < max_sets && wave + si2 * nwaves < nsets) fetch(nxt, si2, bi2);
      }
      f32x2 Tj2[NP], dTj2[NP];
      float Tj1, dTj1 = 0.f;
      uint32_t w0[kOdaIG], w1[kOdaIG];
      const float qd = cur.q * (MASK ? dc.scale : 1.f);       // (the kept values' factor 2 rides on q, as in the data kernel)
#pragma unroll
      for (int jg = 0; jg < kOdaIG; ++jg) {
        Vw[k * kOdaTS + 4 * jg + r] = cur.v[jg];
        float t = cur.v[jg] * cur.q;
        asm("" : "+v"(t));                  // (a ROUNDED product on both sides of T_i - T_j)
        Tw[k * kOdaTS + 4 * jg + r] = t;
        if (jg == kOdaIG - 1) Tj1 = t;
        else if (jg & 1) Tj2[jg >> 1].y = t;
        else Tj2[jg >> 1].x = t;
        if constexpr (MASK) {
          const int j = min(4 * jg + r, N - 1);
          const uint32_t cnt = (uint32_t)b * (uint32_t)NI * stride + (uint32_t)j * (uint32_t)L + (uint32_t)dcl;
          w0[jg] = mask_word32(cnt, key);
          w1[jg] = NI > 1 ? mask_word32(cnt + stride, key) : 0u;
        }
      }
#pragma unroll
      for (int h = 0; h < NP; ++h) dTj2[h] = f32x2{0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int e = lane + 64 * t;
        if (e < 4 * 36) {
          Sw[(e & 3) * kOdaTS + (e >> 2)] = cur.s[t];
          Sa[e] = cur.s[t];
        }
      }
      oda_f32x4 ta = *reinterpret_cast<const oda_f32x4*>(Tw + k * kOdaTS);
      oda_f32x4 sa = *reinterpret_cast<const oda_f32x4*>(Sw + r * kOdaTS);
      oda_f32x4 a_cur = *reinterpret_cast<const oda_f32x4*>(Sa);          // dS[i][0..3] of the region in hand, one read ahead
      __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0), once per (set, sample)
      auto pass = [&](int i0, const uint32_t (&wv)[kOdaIG], int bit0) {
        const int in = min(i0 + 4, 32);
        const oda_f32x4 tn = *reinterpret_cast<const oda_f32x4*>(Tw + k * kOdaTS + in);
        const oda_f32x4 sn = *reinterpret_cast<const oda_f32x4*>(Sw + r * kOdaTS + in);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        // (the reads FIRST)
        float s4[4];
        static_for<4>([&](auto ii_) {
          constexpr int ii = decltype(ii_)::value;
          const float Ti = ta[ii];
          const float av = sa[ii];
          const oda_f32x4 a = a_cur;
          a_cur = *reinterpret_cast<const oda_f32x4*>(Sa + 4 * (ii < 3 ? i0 + ii + 1 : in));
          const f32x2 ti2 = f32x2{Ti, Ti};
          f32x2 x2[NP], u2[NP];
          float x1, u1;
#pragma unroll
          for (int h = 0; h < NP; ++h) {
            x2[h] = ti2 - Tj2[h];
            u2[h] = F2[0][h] * f32x2{a[0], a[0]};
          }
          x1 = Ti - Tj1;
          u1 = F1[0] * a[0];
#pragma unroll
          for (int g = 1; g < 4; ++g) {
#pragma unroll
            for (int h = 0; h < NP; ++h) u2[h] = __builtin_elementwise_fma(F2[g][h], f32x2{a[g], a[g]}, u2[h]);
            u1 = fmaf(F1[g], a[g], u1);
          }
          if constexpr (MASK) {
#pragma unroll
            for (int h = 0; h < NP; ++h) {
              uint32_t m0, m1;
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m0) : "v"(wv[2 * h]), "s"(bit0 + ii));
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m1) : "v"(wv[2 * h + 1]), "s"(bit0 + ii));
              x2[h] = f32x2{__uint_as_float(__float_as_uint(x2[h].x) & m0), __uint_as_float(__float_as_uint(x2[h].y) & m1)};
              u2[h] = f32x2{__uint_as_float(__float_as_uint(u2[h].x) & m0), __uint_as_float(__float_as_uint(u2[h].y) & m1)};
            }
            uint32_t m8;
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m8) : "v"(wv[kOdaIG - 1]), "s"(bit0 + ii));
            x1 = __uint_as_float(__float_as_uint(x1) & m8);
            u1 = __uint_as_float(__float_as_uint(u1) & m8);
          }
          f32x2 t2 = u2[0];
#pragma unroll
          for (int h = 0; h < NP; ++h) {
            dTj2[h] += u2[h];
            if (h > 0) t2 += u2[h];
          }
          dTj1 += u1;
          s4[ii] = (t2.x + t2.y) + u1;
#pragma unroll
          for (int h = 0; h < NP; ++h) {
            acc[2 * h] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, x2[h].x, acc[2 * h], 0, 0, 0);
            acc[2 * h + 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, x2[h].y, acc[2 * h + 1], 0, 0, 0);
          }
          acc[kOdaIG - 1] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, x1, acc[kOdaIG - 1], 0, 0, 0);
        });
        // the four lanes of a block meet: lane r leaves with the sum of region i0 + r
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          s4[ii] += dpp_mov<0xB1>(s4[ii]);   // quad_perm [1,0,3,2]
          s4[ii] += dpp_mov<0x4E>(s4[ii]);   // quad_perm [2,3,0,1]
        }
        const float mine = r == 0 ? s4[0] : r == 1 ? s4[1] : r == 2 ? s4[2] : s4[3];
        Dw[k * kOdaTS + i0 + r] = mine;
        __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0): see the staged kernel
        ta = tn;
        sa = sn;
      };
      const int n_lo = min(N, 32);
      for (int i0 = 0; i0 < n_lo; i0 += 4) pass(i0, w0, i0);
      if (N > 32) pass(32, w1, 0);
      // the data gradient of this sample's 16 features: dT[n] = (sum over j, from the LDS row) - (sum over i, in registers)
      {
        float dq = 0.f;
#pragma unroll
        for (int jg = 0; jg < kOdaIG; ++jg) {
          const int n = 4 * jg + r;
          const float over_i = jg == kOdaIG - 1 ? dTj1 : (jg & 1) ? dTj2[jg >> 1].y : dTj2[jg >> 1].x;
          const float dT = Dw[k * kOdaTS + n] - over_i;         // (a row beyond N may never have been written: not used)
          const float vn = Vw[k * kOdaTS + n];
          if (dok && n < N) {
            const float out = (gate != 0 && !(vn > 0.f)) ? 0.f : dT * qd;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, out), DVb,
                                                  (int)(((uint32_t)(b * N + n) * (uint32_t)L + (uint32_t)d) * 4u), 0, 0);
            dq = fmaf(dT, vn, dq);
          }
        }
        dq += dpp_mov<0xB1>(dq);
        dq += dpp_mov<0x4E>(dq);
        if (dok && r == 0) d_ql[(size_t)b * L + d] = dq * (MASK ? dc.scale : 1.f);
      }
      cur = nxt;
      arrived(cur);
    }
    float* ex = ex_s + ((size_t)(si & 1) * nwaves + wave) * (4 * kOdaIG) * 64 + lane;
    if (sw == 1) {
#pragma unroll
      for (int jg = 0; jg < kOdaIG; ++jg)
#pragma unroll
        for (int t = 0; t < 4; ++t) ex[(4 * jg + t) * 64] = acc[jg][t];
    }
    __syncthreads();
    if (sw == 0 && dok) {
      const float sc = MASK ? dc.scale : 1.f;
      float e[kOdaIG][4];
#pragma unroll
      for (int jg = 0; jg < kOdaIG; ++jg)
#pragma unroll
        for (int t = 0; t < 4; ++t) e[jg][t] = ex[(4 * jg + t) * 64];
#pragma unroll
      for (int jg = 0; jg < kOdaIG; ++jg) {
        const int j = 4 * jg + r;
        if (j < N) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
            if (t < G)
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, (acc[jg][t] + e[jg][t]) * sc), Wb,
                                                    (int)(((uint32_t)(t * N + j) * (uint32_t)L + (uint32_t)d) * 4u), 0, 0);
        }
      }
    }
  }
}

// (Measured on the data-gradient kernel below, B = 512: 14 us of prologue + epilogue, 60 us in the j loop, 4 us of it the hash;
//  removing a third of its lane-operations (packed pairs, independent chains) or prefetching the filter rows one or two
//  regions ahead moves it by 2 us -- the SIMDs that hold 3 of the 10 waves of a CU set the time.)
// (The data gradient was tried on the same instruction too -- u[i][j][d] = sum_g dS[b,i,g] w[g,j,d] as four chained 4x4x1 MFMAs
//  per pair of region groups, masked and summed along rows and columns by the VALU: 77-79 us, the same as the VALU kernel
//  below, whose 8 lane-operations per element it only reshuffles (4 MFMA reads + 2 mask + 2 adds).  Not kept.)
// backward d_vl, d_ql, bit mask: as oda_bwd_data_kernel
template <int G>
__global__ void oda_bwd_data_bits_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                         const float* __restrict__ w, const float* __restrict__ dS,
                                         float* __restrict__ d_vl, float* __restrict__ d_ql, DropCfg dc, int N, int L,
                                         int gate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dT_s = reinterpret_cast<float*>(smem);        // [N][blockDim]
  float* dS_s = dT_s + (size_t)N * blockDim.x;          // [N][G]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.x;
  const int NI = (N + 31) >> 5;
  const uint32_t key = drop_key(dc);
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const float* vlb = vl + (size_t)b * N * L;
  for (int t = tid; t < (N + kIC) * G; t += nt) dS_s[t] = t < N * G ? dS[(size_t)b * N * G + t] : 0.f;   // (zero rows pad the last chunk)
  __syncthreads();
  for (int d = tid; d < L; d += nt) {
    for (int n = 0; n < N; ++n) dT_s[n * nt + tid] = 0.f;
    for (int i0 = 0; i0 < N; i0 += kIC) {
      const int sh = i0 & 31;
      const bool straddle = ((min(i0 + kIC, N) - 1) >> 5) != (i0 >> 5);
      const uint32_t base = ((uint32_t)b * NI + (uint32_t)(i0 >> 5)) * stride;
      // two regions i per packed lane-operation (v_pk_fma_f32 / v_pk_add_f32); the kIC/2 accumulation chains of one filter
      // row are independent, so no dependent-issue bubbles
      f32x2 ds[kIC / 2][G], dTi[kIC / 2];
#pragma unroll
      for (int ip = 0; ip < kIC / 2; ++ip) {
        dTi[ip] = f32x2{0.f, 0.f};
#pragma unroll
        for (int g = 0; g < G; ++g) ds[ip][g] = f32x2{dS_s[(i0 + 2 * ip) * G + g], dS_s[(i0 + 2 * ip + 1) * G + g]};
      }
      // the filter rows of regions j + 1, j + 2 are in flight under the arithmetic of region j (L2 latency > one j): three
      // named buffers taken in turn, so no register rotation makes the compiler wait for the newest load
      auto load_w = [&](float (&wr)[G], int j) {
        const int jc = min(j, N - 1);
#pragma unroll
        for (int g = 0; g < G; ++g) wr[g] = w[((size_t)g * N + jc) * L + d];
      };
      auto one_j = [&](int j, const float (&wv)[G]) {
        const float dtj = dT_s[j * nt + tid];
        const uint32_t bits = oda_bits(base + (uint32_t)(j * L + d), stride, sh, straddle, key);
        f32x2 u[kIC / 2];
#pragma unroll
        for (int ip = 0; ip < kIC / 2; ++ip) u[ip] = ds[ip][0] * f32x2{wv[0], wv[0]};
#pragma unroll
        for (int g = 1; g < G; ++g)
#pragma unroll
          for (int ip = 0; ip < kIC / 2; ++ip) u[ip] = __builtin_elementwise_fma(ds[ip][g], f32x2{wv[g], wv[g]}, u[ip]);
        f32x2 pj = f32x2{0.f, 0.f};
        static_for<kIC / 2>([&](auto ip_) {
          constexpr int ip = decltype(ip_)::value;
          const f32x2 m = f32x2{keep_bit<2 * ip>(u[ip].x, bits), keep_bit<2 * ip + 1>(u[ip].y, bits)};
          dTi[ip] += m;
          pj += m;
        });
        dT_s[j * nt + tid] = dtj - (pj.x + pj.y);
      };
      float wa[G], wb[G], wc[G];
      load_w(wa, 0);
      load_w(wb, 1);
      load_w(wc, 2);
      for (int j = 0; j < N; j += 3) {
        one_j(j, wa);
        load_w(wa, j + 3);
        if (j + 1 < N) one_j(j + 1, wb);
        load_w(wb, j + 4);
        if (j + 2 < N) one_j(j + 2, wc);
        load_w(wc, j + 5);
      }
#pragma unroll
      for (int ip = 0; ip < kIC / 2; ++ip) {
        if (i0 + 2 * ip < N) dT_s[(i0 + 2 * ip) * nt + tid] += dTi[ip].x;
        if (i0 + 2 * ip + 1 < N) dT_s[(i0 + 2 * ip + 1) * nt + tid] += dTi[ip].y;
      }
    }
    const float qd = ql[(size_t)b * L + d] * dc.scale;      // (the kept values' factor 2 rides on q here)
    float dq = 0.f;
    for (int n = 0; n < N; ++n) {
      const float t = dT_s[n * nt + tid];
      const float vn = vlb[(size_t)n * L + d];
      d_vl[((size_t)b * N + n) * L + d] = (gate != 0 && !(vn > 0.f)) ? 0.f : t * qd;   // gate: relu gradient of vl's producer
      dq = fmaf(t, vn, dq);
    }
    d_ql[(size_t)b * L + d] = dq * dc.scale;
  }
}

// The same with the region chunks of a sample spread over the waves of a workgroup: grid (ceil(L/64), B), wave c = chunk
// i0 = c kIC of the regions i, lane = feature d of the block.  The kernel above runs 5 waves per sample (one feature per lane,
// three chunks each in turn): a CU's 10 waves sit 3/3/2/2 on its SIMDs and the busiest SIMD sets the time (VALU-issue bound:
// >= 65 % of the busy SIMD-cycles).  Here a wave is a third of that work and a CU holds 15 of them: the same instruction
// stream in pieces that spread evenly.  Every wave writes its own plane of partial sums (no read-modify-write in the j loop);
// the planes meet in LDS, and the N output rows are split over the waves again.
template <int G>
__global__ void oda_bwd_data_bits_split_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                               const float* __restrict__ w, const float* __restrict__ dS,
                                               float* __restrict__ d_vl, float* __restrict__ d_ql, DropCfg dc, int N, int L,
                                               int gate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NC = blockDim.x >> 6;                       // chunk waves = ceil(N / kIC)
  float* dT_s = reinterpret_cast<float*>(smem);         // [NC][N][64]
  float* dq_s = dT_s + (size_t)NC * N * 64;             // [NC][64]
  float* dS_s = dq_s + NC * 64;                         // [N + kIC][G]
  const int tid = threadIdx.x, lane = tid & 63;
  const int c = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.y;
  const int d_raw = blockIdx.x * 64 + lane;
  const bool active = d_raw < L;
  const int d = active ? d_raw : L - 1;
  const int NI = (N + 31) >> 5;
  const uint32_t key = drop_key(dc);
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const float* vlb = vl + (size_t)b * N * L;
  for (int t = tid; t < (N + kIC) * G; t += blockDim.x) dS_s[t] = t < N * G ? dS[(size_t)b * N * G + t] : 0.f;
  __syncthreads();
  float* mine = dT_s + (size_t)c * N * 64 + lane;       // this wave's plane: row n at mine[n * 64]
  float vn_pre[kIC], ql_pre;      // vl of the output rows this wave finishes with (n = c, c + NC, ...) and ql: see below
  {
    const int i0 = c * kIC;
    const int sh = i0 & 31;
    const bool straddle = ((min(i0 + kIC, N) - 1) >> 5) != (i0 >> 5);
    const uint32_t base = ((uint32_t)b * NI + (uint32_t)(i0 >> 5)) * stride;
    f32x2 ds[kIC / 2][G], dTi[kIC / 2];
#pragma unroll
    for (int ip = 0; ip < kIC / 2; ++ip) {
      dTi[ip] = f32x2{0.f, 0.f};
#pragma unroll
      for (int g = 0; g < G; ++g) ds[ip][g] = f32x2{dS_s[(i0 + 2 * ip) * G + g], dS_s[(i0 + 2 * ip + 1) * G + g]};
    }
    auto load_w = [&](float (&wr)[G], int j) {
      const int jc = min(j, N - 1);
#pragma unroll
      for (int g = 0; g < G; ++g) wr[g] = w[((size_t)g * N + jc) * L + d];
    };
    auto one_j = [&](int j, const float (&wv)[G]) {
      const uint32_t bits = oda_bits(base + (uint32_t)(j * L + d), stride, sh, straddle, key);
      f32x2 u[kIC / 2];
#pragma unroll
      for (int ip = 0; ip < kIC / 2; ++ip) u[ip] = ds[ip][0] * f32x2{wv[0], wv[0]};
#pragma unroll
      for (int g = 1; g < G; ++g)
#pragma unroll
        for (int ip = 0; ip < kIC / 2; ++ip) u[ip] = __builtin_elementwise_fma(ds[ip][g], f32x2{wv[g], wv[g]}, u[ip]);
      f32x2 pj = f32x2{0.f, 0.f};
      static_for<kIC / 2>([&](auto ip_) {
        constexpr int ip = decltype(ip_)::value;
        const f32x2 m = f32x2{keep_bit<2 * ip>(u[ip].x, bits), keep_bit<2 * ip + 1>(u[ip].y, bits)};
        dTi[ip] += m;
        pj += m;
      });
      mine[j * 64] = -(pj.x + pj.y);
    };
    float wa[G], wb[G], wc[G];
    load_w(wa, 0);
    load_w(wb, 1);
    load_w(wc, 2);
    // vl of the wave's output rows and ql: asked for HERE, behind the first filter rows (loads return in order), used behind
    // the region loop.  (Loaded where they are used, one row at a time, each was a first touch of HBM with nothing else to do:
    // 38 % of the wave cycles inside s_waitcnt, round 6.)
#pragma unroll
    for (int q = 0; q < kIC; ++q) vn_pre[q] = vlb[(size_t)min(c + q * NC, N - 1) * L + d];
    ql_pre = ql[(size_t)b * L + d];
    // The row a pass starts with must have arrived, the two behind it stay in flight.  Said before the loop and at the end
    // of a pass: the compiler does not count loads across the back edge and would wait for vmcnt(0) at the top -- for the
    // rows it had issued a few instructions earlier.
    auto arrived = [&](float (&wr)[G]) {
#pragma unroll
      for (int g = 0; g < G; ++g) asm volatile("" : "+v"(wr[g]));
    };
    arrived(wa);
    for (int j = 0; j < N; j += 3) {
      one_j(j, wa);
      load_w(wa, j + 3);
      if (j + 1 < N) one_j(j + 1, wb);
      load_w(wb, j + 4);
      if (j + 2 < N) one_j(j + 2, wc);
      load_w(wc, j + 5);
      arrived(wa);
    }
#pragma unroll
    for (int ip = 0; ip < kIC / 2; ++ip) {      // (the wave's own column of its own plane: written above, by this lane)
      if (i0 + 2 * ip < N) mine[(i0 + 2 * ip) * 64] += dTi[ip].x;
      if (i0 + 2 * ip + 1 < N) mine[(i0 + 2 * ip + 1) * 64] += dTi[ip].y;
    }
  }
  __syncthreads();
  const float qd = ql_pre * dc.scale;                     // (the kept values' factor 2 rides on q here)
  float dq = 0.f;
#pragma unroll
  for (int qn = 0; qn < kIC; ++qn) {                      // the output rows, dealt over the waves
    const int n = c + qn * NC;
    if (n < N) {
      float t = 0.f;
      for (int q = 0; q < NC; ++q) t += dT_s[((size_t)q * N + n) * 64 + lane];    // fixed order
      const float vn = vn_pre[qn];
      if (active) d_vl[((size_t)b * N + n) * L + d] = (gate != 0 && !(vn > 0.f)) ? 0.f : t * qd;
      dq = fmaf(t, vn, dq);
    }
  }
  dq_s[c * 64 + lane] = dq;
  __syncthreads();
  if (c == 0 && active) {
    float s = 0.f;
    for (int q = 0; q < NC; ++q) s += dq_s[q * 64 + lane];
    d_ql[(size_t)b * L + d] = s * dc.scale;
  }
}

// backward d_w slabs, bit mask: as oda_bwd_weight_kernel; the hash words of the chunk's kIC regions j are drawn once per
// 32 regions i
template <int G>
__global__ void oda_bwd_weight_bits_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                           const float* __restrict__ dS, float* __restrict__ slab, DropCfg dc, int B, int N,
                                           int L, int samples_per_group) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dS_s = reinterpret_cast<float*>(smem);  // [N + 3][G] (zero padded to a multiple of 4 rows)
  const int tid = threadIdx.x, nt = blockDim.x;
  const int j0 = blockIdx.x * kIC, sg = blockIdx.y;
  const int NQ = (N + 3) >> 2, NI = (N + 31) >> 5;
  const uint32_t key = drop_key(dc);
  const uint32_t stride = (uint32_t)N * (uint32_t)L;
  const int b_lo = sg * samples_per_group, b_hi = min(B, b_lo + samples_per_group);
  const int d = tid;  // the launcher guarantees blockDim >= L, so one feature per lane
  const bool active = d < L;
  float dw[kIC][G];
#pragma unroll
  for (int jc = 0; jc < kIC; ++jc)
#pragma unroll
    for (int g = 0; g < G; ++g) dw[jc][g] = 0.f;
  for (int b = b_lo; b < b_hi; ++b) {
    __syncthreads();
    for (int t = tid; t < NQ * 4 * G; t += nt) dS_s[t] = (t < N * G) ? dS[(size_t)b * N * G + t] : 0.f;
    __syncthreads();
    if (!active) continue;
    const float* vlb = vl + (size_t)b * N * L;
    const float qd = ql[(size_t)b * L + d];
    float Tj[kIC];
#pragma unroll
    for (int jc = 0; jc < kIC; ++jc) Tj[jc] = (j0 + jc < N) ? vlb[(size_t)(j0 + jc) * L + d] * qd : 0.f;
    for (int iw = 0; iw < NI; ++iw) {
      uint32_t words[kIC];
#pragma unroll
      for (int jc = 0; jc < kIC; ++jc)
        words[jc] = mask_word32(((uint32_t)b * NI + iw) * stride + (uint32_t)(min(j0 + jc, N - 1) * L + d), key);
      const int q_hi = min(NQ, (iw + 1) * 8);
      for (int iq = iw * 8; iq < q_hi; ++iq) {
        const int sh = (iq & 7) * 4;
        float Ti[4], ds[4][G];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int i = iq * 4 + k;
          Ti[k] = (i < N) ? vlb[(size_t)i * L + d] * qd : 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) ds[k][g] = dS_s[i * G + g];  // zero rows beyond N
        }
#pragma unroll
        for (int jc = 0; jc < kIC; ++jc) {
          const uint32_t bits = words[jc] >> sh;
          static_for<4>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            const float val = keep_bit<k>(Ti[k] - Tj[jc], bits);
#pragma unroll
            for (int g = 0; g < G; ++g) dw[jc][g] = fmaf(ds[k][g], val, dw[jc][g]);
          });
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int jc = 0; jc < kIC; ++jc)
    if (j0 + jc < N)
#pragma unroll
      for (int g = 0; g < G; ++g) slab[(((size_t)sg * G + g) * N + j0 + jc) * L + d] = dw[jc][g] * dc.scale;
}

// ------------------------------------------------------------------------------------------ forward
// grid (ceil(N/kIC), B); block = round_up(min(L,1024), 64) threads, lane <-> feature d.
template <int G, bool DROP>
__global__ void oda_fwd_kernel(const float* __restrict__ vl, const float* __restrict__ ql, const float* __restrict__ w,
                               const float* __restrict__ bias, float* __restrict__ logits, DropCfg dc, int N, int L) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* red_s = reinterpret_cast<float*>(smem);  // [nwaves][kIC*G]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int b = blockIdx.y, i0 = blockIdx.x * kIC;
  const int NQ = (N + 3) >> 2;
  const uint64_t seed_eff = dc.effective();
  const float* vlb = vl + (size_t)b * N * L;
  float acc[kIC][G];
#pragma unroll
  for (int ic = 0; ic < kIC; ++ic)
#pragma unroll
    for (int g = 0; g < G; ++g) acc[ic][g] = 0.f;

  for (int d = tid; d < L; d += blockDim.x) {
    const float qd = ql[(size_t)b * L + d];
    float Ti[kIC];
#pragma unroll
    for (int ic = 0; ic < kIC; ++ic) Ti[ic] = (i0 + ic < N) ? vlb[(size_t)(i0 + ic) * L + d] * qd : 0.f;
#pragma unroll 2
    for (int j = 0; j < N; ++j) {
      const float Tj = vlb[(size_t)j * L + d] * qd;
      float wv[G];
#pragma unroll
      for (int g = 0; g < G; ++g) wv[g] = w[((size_t)g * N + j) * L + d];
#pragma unroll
      for (int q = 0; q < kIC / 4; ++q) {
        uint32_t word = 0;
        if (DROP) word = mask_word(mask_counter(b, (i0 >> 2) + q, j, d, NQ, N, L), seed_eff);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int ic = q * 4 + k;
          float val = Ti[ic] - Tj;
          if (DROP) val *= keep_scale(word, k, dc.p8, dc.scale);
#pragma unroll
          for (int g = 0; g < G; ++g) acc[ic][g] = fmaf(wv[g], val, acc[ic][g]);
        }
      }
    }
  }
  // reduce over d: wave64 shuffles, then across waves through LDS
#pragma unroll
  for (int ic = 0; ic < kIC; ++ic)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float r = wave_sum(acc[ic][g]);
      if (lane == 0) red_s[wave * (kIC * G) + ic * G + g] = r;
    }
  __syncthreads();
  for (int t = tid; t < kIC * G; t += blockDim.x) {
    const int ic = t / G, g = t % G;
    if (i0 + ic < N) {
      float s = bias[g];
      for (int wv = 0; wv < nwaves; ++wv) s += red_s[wv * (kIC * G) + t];
      logits[((size_t)b * N + i0 + ic) * G + g] = s;
    }
  }
}

// ----------------------------------------------------------------------------- backward: d_vl, d_ql
// One workgroup per sample, lane <-> feature d.  dT[n][d] is accumulated in an LDS column private to
// the lane:  dT[i] += sum_j m*U[i,j],  dT[j] -= sum_i m*U[i,j],  U[i,j,d] = sum_g dS[i,g] w[g,j,d].
template <int G, bool DROP>
__global__ void oda_bwd_data_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                    const float* __restrict__ w, const float* __restrict__ dS, float* __restrict__ d_vl,
                                    float* __restrict__ d_ql, DropCfg dc, int N, int L, int gate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dT_s = reinterpret_cast<float*>(smem);        // [N][blockDim]
  float* dS_s = dT_s + (size_t)N * blockDim.x;          // [N][G]
  const int tid = threadIdx.x, nt = blockDim.x;
  const int b = blockIdx.x;
  const int NQ = (N + 3) >> 2;
  const uint64_t seed_eff = dc.effective();
  const float* vlb = vl + (size_t)b * N * L;
  for (int t = tid; t < N * G; t += nt) dS_s[t] = dS[(size_t)b * N * G + t];
  __syncthreads();
  for (int d = tid; d < L; d += nt) {
    for (int n = 0; n < N; ++n) dT_s[n * nt + tid] = 0.f;
    for (int i0 = 0; i0 < N; i0 += kIC) {
      float ds[kIC][G], dTi[kIC];
#pragma unroll
      for (int ic = 0; ic < kIC; ++ic) {
        dTi[ic] = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) ds[ic][g] = (i0 + ic < N) ? dS_s[(i0 + ic) * G + g] : 0.f;
      }
#pragma unroll 2
      for (int j = 0; j < N; ++j) {
        float wv[G];
#pragma unroll
        for (int g = 0; g < G; ++g) wv[g] = w[((size_t)g * N + j) * L + d];
        float pj = 0.f;
#pragma unroll
        for (int q = 0; q < kIC / 4; ++q) {
          uint32_t word = 0;
          if (DROP) word = mask_word(mask_counter(b, (i0 >> 2) + q, j, d, NQ, N, L), seed_eff);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int ic = q * 4 + k;
            float u = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) u = fmaf(ds[ic][g], wv[g], u);
            if (DROP) u *= keep_scale(word, k, dc.p8, dc.scale);
            dTi[ic] += u;
            pj += u;
          }
        }
        dT_s[j * nt + tid] -= pj;
      }
#pragma unroll
      for (int ic = 0; ic < kIC; ++ic)
        if (i0 + ic < N) dT_s[(i0 + ic) * nt + tid] += dTi[ic];
    }
    const float qd = ql[(size_t)b * L + d];
    float dq = 0.f;
    for (int n = 0; n < N; ++n) {
      const float t = dT_s[n * nt + tid];
      const float vn = vlb[(size_t)n * L + d];
      d_vl[((size_t)b * N + n) * L + d] = (gate != 0 && !(vn > 0.f)) ? 0.f : t * qd;   // gate: relu gradient of vl's producer
      dq = fmaf(t, vn, dq);
    }
    d_ql[(size_t)b * L + d] = dq;
  }
}

// --------------------------------------------------------------------------------- backward: d_w slabs
// grid (ceil(N/kIC) j-chunks, SG sample groups); lane <-> d; dw[jc][g] for the chunk lives in registers
// across the group's samples:  dw[g,j,d] += sum_i dS[b,i,g] * m(b,i,j,d) * (T[b,i,d] - T[b,j,d]).
template <int G, bool DROP>
__global__ void oda_bwd_weight_kernel(const float* __restrict__ vl, const float* __restrict__ ql,
                                      const float* __restrict__ dS, float* __restrict__ slab, DropCfg dc, int B, int N,
                                      int L, int samples_per_group) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* dS_s = reinterpret_cast<float*>(smem);  // [N + 3][G] (zero padded to a multiple of 4 rows)
  const int tid = threadIdx.x, nt = blockDim.x;
  const int j0 = blockIdx.x * kIC, sg = blockIdx.y;
  const int NQ = (N + 3) >> 2;
  const uint64_t seed_eff = dc.effective();
  const int b_lo = sg * samples_per_group, b_hi = min(B, b_lo + samples_per_group);
  const int d = tid;  // the launcher guarantees blockDim >= L, so one feature per lane
  const bool active = d < L;
  float dw[kIC][G];
#pragma unroll
  for (int jc = 0; jc < kIC; ++jc)
#pragma unroll
    for (int g = 0; g < G; ++g) dw[jc][g] = 0.f;
  for (int b = b_lo; b < b_hi; ++b) {
    __syncthreads();
    for (int t = tid; t < NQ * 4 * G; t += nt) dS_s[t] = (t < N * G) ? dS[(size_t)b * N * G + t] : 0.f;
    __syncthreads();
    if (!active) continue;
    const float* vlb = vl + (size_t)b * N * L;
    const float qd = ql[(size_t)b * L + d];
    float Tj[kIC];
#pragma unroll
    for (int jc = 0; jc < kIC; ++jc) Tj[jc] = (j0 + jc < N) ? vlb[(size_t)(j0 + jc) * L + d] * qd : 0.f;
    for (int iq = 0; iq < NQ; ++iq) {
      float Ti[4], ds[4][G];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = iq * 4 + k;
        Ti[k] = (i < N) ? vlb[(size_t)i * L + d] * qd : 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) ds[k][g] = dS_s[i * G + g];  // zero rows beyond N
      }
#pragma unroll
      for (int jc = 0; jc < kIC; ++jc) {
        uint32_t word = 0;
        if (DROP) word = mask_word(mask_counter(b, iq, j0 + jc, d, NQ, N, L), seed_eff);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float val = Ti[k] - Tj[jc];
          if (DROP) val *= keep_scale(word, k, dc.p8, dc.scale);
#pragma unroll
          for (int g = 0; g < G; ++g) dw[jc][g] = fmaf(ds[k][g], val, dw[jc][g]);
        }
      }
    }
  }
  if (!active) return;
#pragma unroll
  for (int jc = 0; jc < kIC; ++jc)
    if (j0 + jc < N)
#pragma unroll
      for (int g = 0; g < G; ++g) slab[(((size_t)sg * G + g) * N + j0 + jc) * L + d] = dw[jc][g];
}

// d_w[e] = sum_sg slab[sg][e]  (fixed order), e over G*N*L.  256 lanes = 64 elements x 4 quarters of the slab list, eight
// loads in flight per lane, the quarters added in order through LDS (one lane walking all 128 slabs with one load in
// flight took 32 us for 23 MB).
__global__ __launch_bounds__(256) void oda_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                         size_t n, int S) {
  __shared__ float part[3][64];
  const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
  const size_t e = (size_t)blockIdx.x * 64 + c;
  const size_t ec = e < n ? e : n - 1;
  const int per = (S + 3) / 4, s_lo = q * per, s_hi = min(S, s_lo + per);
  float a = 0.f;
  for (int s0 = s_lo; s0 < s_hi; s0 += 8) {
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = slab[(size_t)min(s0 + k, S - 1) * n + ec];
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (s0 + k < s_hi) a += v[k];
  }
  if (q > 0) part[q - 1][c] = a;
  __syncthreads();
  if (q == 0 && e < n) out[e] = ((a + part[0][c]) + part[1][c]) + part[2][c];
}

// d_bias[g] = sum_{b,i} dS[b,i,g]; one workgroup of 1024 lanes, fixed order.  A lane takes every 1024th row and keeps
// eight rows (8*G independent loads) in flight: the tensor is tiny (B*N*G floats), the kernel is a chain of load
// latencies -- with one dependent load at a time it took 44 us at B*N = 18432, more than a third of the data pass.
template <int G>
__global__ __launch_bounds__(1024) void oda_dbias_kernel(const float* __restrict__ dS, float* __restrict__ d_bias, int rows) {
  __shared__ float part[16][G];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float acc[G];
#pragma unroll
  for (int g = 0; g < G; ++g) acc[g] = 0.f;
  for (int r0 = tid; r0 < rows; r0 += 8 * 1024) {
    float v[8][G];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int r = min(r0 + 1024 * k, rows - 1);
#pragma unroll
      for (int g = 0; g < G; ++g) v[k][g] = dS[(size_t)r * G + g];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (r0 + 1024 * k < rows) {
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] += v[k][g];
      }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const float a = wave_sum(acc[g]);
    if (lane == 0) part[wave][g] = a;
  }
  __syncthreads();
  if (tid < G) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += part[w][tid];
    d_bias[tid] = t;
  }
}

// mask[b][i][j*L+d] = keep/(1-p) exactly as the fused kernels draw it (test / debugging aid)
__global__ __launch_bounds__(256) void oda_mask_kernel(float* __restrict__ mask, DropCfg dc, int N, int L, bool bits_mode) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;  // j*L + d
  if (e >= N * L) return;
  const int j = e / L, d = e % L;
  const int NQ = (N + 3) >> 2;
  const uint64_t seed_eff = dc.effective();
  float m = 1.f;
  if (bits_mode) {
    const int NI = (N + 31) >> 5;
    const uint32_t cnt = (((uint32_t)b * NI + (uint32_t)(i >> 5)) * N + j) * L + d;
    m = ((mask_word32(cnt, drop_key(dc)) >> (i & 31)) & 1u) != 0u ? dc.scale : 0.f;
  } else if (dc.p8 > 0) {
    m = keep_scale(mask_word(mask_counter(b, i >> 2, j, d, NQ, N, L), seed_eff), i & 3, dc.p8, dc.scale);
  }
  mask[((size_t)b * N + i) * N * L + e] = m;
}

// p = 0.5 and 32-bit element counters: the one-bit-per-element layout (VQA_K2_BYTE_MASK=1 keeps the byte layout)
static bool oda_bits_mode(const DropCfg& dc, int B, int N, int L) {
  return dc.p8 == kDropHalf && (size_t)B * ((N + 31) / 32) * N * L < (1ull << 32) && vqa::option("VQA_K2_BYTE_MASK") == nullptr;
}

static int oda_threads(int L) {
  int t = (L < 1024 ? L : 1024);
  return (t + 63) / 64 * 64;
}
static int oda_groups(int B) { return B < 128 ? B : 128; }
// the 4x4-MFMA kernels (forward, weight gradient): G <= 4 glimpses, N <= 36 regions, no dropout or the one-bit p = 0.5 mask
static bool oda_mfma_ok(const DropCfg& dc, int B, int N, int L, int G) {
  const bool off = vqa::option_is("VQA_K2_MFMA", '0');
  return !off && G <= 4 && N <= 4 * kOdaIG && (dc.p8 == 0 || oda_bits_mode(dc, B, N, L)) && (size_t)B * N * L * 4 < (1ull << 32);
}
static int oda_mfma_groups(int B) { return B < 256 ? B : 256; }   // sample groups of the MFMA weight gradient: one workgroup per CU

template <int G>
static int launch_fwd(const float* vl, const float* ql, const float* w, const float* bias, float* logits, DropCfg dc,
                      int B, int N, int L, hipStream_t s) {
  const int nt = oda_threads(L);
  {
    // the 4x4-MFMA form: G <= 4 glimpses, N <= 36 regions, no dropout or the one-bit p = 0.5 mask (VQA_K2_MFMA=0: VALU kernels)
    const bool bits = oda_bits_mode(dc, B, N, L);
    if (oda_mfma_ok(dc, B, N, L, G)) {
      const int nsets = (L + 15) / 16;
      // waves per sample: 8 = two per SIMD and workgroup, four per SIMD with the two workgroups a CU holds at B = 512
      // (the 20 feature sets of L = 310 go 3,3,3,3,2,2,2,2).  Measured on the ODA attention op, forward: 4 waves 54.9 us,
      // 8 waves 50.8; 5 or 6 waves 63-66 (a CU's 10 / 12 waves do not spread evenly over its SIMDs); slicing the region
      // axis j as well (units of (set, 12 regions): the same work for every wave) 55.2-69.6 -- every unit pays its own
      // nine T_i loads and pipeline fill.  The kernel is VALU-issue bound: what helps is an even wave count per SIMD, not
      // more waves.
      int nw = nsets < 8 ? (nsets < 4 ? nsets : 4) : 8;
      const int jt = 1, nj = (N + 3) / 4 * 4;   // (one slice of the region axis j: see the kernel's note on work units)
      const size_t lds_m = (size_t)nw * 4 * kOdaIG * 4 * sizeof(float);
      // packed subtracts (round 4): the nine differences of a region j as four v_pk_add_f32 + one v_sub_f32 -- the kernel is
      // VALU-issue bound, 52.1 -> 50.5 us at B = 512 (the weight gradient likewise); VQA_K2_PK=0 keeps the scalar form for A/B
      if (bits && !vqa::option_is("VQA_K2_PK", '0')) {
        VQA_LAUNCH((oda_fwd_mfma_kernel<true, true>), dim3(B), dim3(64 * nw), lds_m, s, vl, ql, w, bias, logits, dc, N, L, G, jt, nj);
      } else if (bits) {
        VQA_LAUNCH(oda_fwd_mfma_kernel<true>, dim3(B), dim3(64 * nw), lds_m, s, vl, ql, w, bias, logits, dc, N, L, G, jt, nj);
      } else {
        VQA_LAUNCH(oda_fwd_mfma_kernel<false>, dim3(B), dim3(64 * nw), lds_m, s, vl, ql, w, bias, logits, dc, N, L, G, jt, nj);
      }
      return check_launch("object_difference_attention_fwd");
    }
  }
  if (oda_bits_mode(dc, B, N, L)) {
    const size_t lds_b = (size_t)(nt / 64) * kIB * G * sizeof(float);
    VQA_LAUNCH((oda_fwd_bits_kernel<G>), dim3((N + kIB - 1) / kIB, B), dim3(nt), lds_b, s, vl, ql, w, bias, logits, dc, N,
                       L);
    return check_launch("object_difference_attention_fwd");
  }
  const size_t lds = (size_t)(nt / 64) * kIC * G * sizeof(float);
  dim3 grid((N + kIC - 1) / kIC, B);
  if (dc.p8 > 0)
    VQA_LAUNCH((oda_fwd_kernel<G, true>), grid, dim3(nt), lds, s, vl, ql, w, bias, logits, dc, N, L);
  else
    VQA_LAUNCH((oda_fwd_kernel<G, false>), grid, dim3(nt), lds, s, vl, ql, w, bias, logits, dc, N, L);
  return check_launch("object_difference_attention_fwd");
}

template <int G>
static int launch_bwd(const float* vl, const float* ql, const float* w, const float* dS, float* d_vl, float* d_ql,
                      float* d_w, float* d_bias, float* slab, DropCfg dc, int B, int N, int L, int gate_dvl, hipStream_t s) {
  const int nt = oda_threads(L);
  // One pass over the mask for both gradients: VQA_K2_FUSED=1.  NOT the default -- measured on one box, alternating
  // (tools/ab_knob.sh, B = 512): the backward 154 us fused against 140 us with the two kernels below; the fused kernel needs
  // all 256 registers (two waves per SIMD) and issues during ~60 % of its cycles, the data-gradient kernel below during 88 %
  // (docs/measured_negatives_r06.md).
  if (oda_mfma_ok(dc, B, N, L, G) && vqa::option_is("VQA_K2_FUSED", '1')) {
    int SG = oda_mfma_groups(B);
    const int spg = (B + SG - 1) / SG;
    SG = (B + spg - 1) / spg;
    const int nsets = (L + 15) / 16, nw = nsets < 4 ? nsets : 4;
    const size_t lds_f = ((size_t)2 * nw * 4 * kOdaIG * 64 + (size_t)2 * nw * ((16 + 4 + 16 + 16) * kOdaTS + 36 * 4)) * sizeof(float);
    if (dc.p8 > 0) {
      VQA_ENSURE_LDS((oda_bwd_fused_kernel<true>), lds_f);
      VQA_LAUNCH((oda_bwd_fused_kernel<true>), dim3(SG), dim3(128 * nw), lds_f, s, vl, ql, w, dS, d_vl, d_ql, slab, dc, B, N, L, G,
                 spg, gate_dvl);
    } else {
      VQA_ENSURE_LDS((oda_bwd_fused_kernel<false>), lds_f);
      VQA_LAUNCH((oda_bwd_fused_kernel<false>), dim3(SG), dim3(128 * nw), lds_f, s, vl, ql, w, dS, d_vl, d_ql, slab, dc, B, N, L, G,
                 spg, gate_dvl);
    }
    const size_t n = (size_t)G * N * L;
    VQA_LAUNCH(oda_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, slab, d_w, n, SG);
    VQA_LAUNCH(oda_dbias_kernel<G>, dim3(1), dim3(1024), 0, s, dS, d_bias, B * N);
    return check_launch("object_difference_attention_bwd");
  }
  {
    const size_t lds = ((size_t)N * nt + (size_t)(N + kIC) * G) * sizeof(float);   // (+ kIC zero rows of dS, bit-mask kernel)
    VQA_REQUIRE(lds <= 160 * 1024, VQA_E_UNSUPPORTED, "object_difference_attention_bwd: N=%d L=%d need %zu B of LDS", N, L, lds);
    const bool split_off = vqa::option_is("VQA_K2_DATA_SPLIT", '0');
    const int nc = (N + kIC - 1) / kIC;
    if (oda_bits_mode(dc, B, N, L) && nc <= 4 && !split_off && (long)B * ((L + 63) / 64) >= 512) {
      const size_t lds_s = ((size_t)nc * N * 64 + (size_t)nc * 64 + (size_t)(N + kIC) * G) * sizeof(float);
      VQA_ENSURE_LDS((oda_bwd_data_bits_split_kernel<G>), lds_s);
      VQA_LAUNCH((oda_bwd_data_bits_split_kernel<G>), dim3((L + 63) / 64, B), dim3(64 * nc), lds_s, s, vl, ql, w, dS,
                         d_vl, d_ql, dc, N, L, gate_dvl);
    } else if (oda_bits_mode(dc, B, N, L)) {
      VQA_ENSURE_LDS((oda_bwd_data_bits_kernel<G>), lds);
      VQA_LAUNCH((oda_bwd_data_bits_kernel<G>), dim3(B), dim3(nt), lds, s, vl, ql, w, dS, d_vl, d_ql, dc, N, L, gate_dvl);
    } else if (dc.p8 > 0) {
      VQA_ENSURE_LDS((oda_bwd_data_kernel<G, true>), lds);
      VQA_LAUNCH((oda_bwd_data_kernel<G, true>), dim3(B), dim3(nt), lds, s, vl, ql, w, dS, d_vl, d_ql, dc, N, L, gate_dvl);
    } else {
      VQA_ENSURE_LDS((oda_bwd_data_kernel<G, false>), lds);
      VQA_LAUNCH((oda_bwd_data_kernel<G, false>), dim3(B), dim3(nt), lds, s, vl, ql, w, dS, d_vl, d_ql, dc, N, L, gate_dvl);
    }
  }
  {
    int SG = oda_groups(B);
    int spg = (B + SG - 1) / SG;
    const size_t lds = (size_t)(N + 3) * G * sizeof(float);
    dim3 grid((N + kIC - 1) / kIC, SG);
    if (oda_mfma_ok(dc, B, N, L, G)) {
      SG = oda_mfma_groups(B);
      spg = (B + SG - 1) / SG;
      SG = (B + spg - 1) / spg;
      const int nsets = (L + 15) / 16, nw = nsets < 4 ? nsets : 4;
      const size_t lds_m = (size_t)nw * 4 * kOdaIG * 64 * sizeof(float);
      const bool pk = !vqa::option_is("VQA_K2_PK", '0');     // packed subtracts (round 4; "0" = the scalar form, for A/B)
      const bool staged = pk && !vqa::option_is("VQA_K2_WSTAGE", '0');   // operands of the region loop from LDS (round 6)
      const bool halves = vqa::option_is("VQA_K2_WSTAGE", '2');          // "2": 16 waves per workgroup (measured: the same time)
      const size_t lds_st = 2 * lds_m + (size_t)2 * nw * (halves ? 2 : 1) * 20 * kOdaTS * sizeof(float);
#define VQA_K2_STAGED(MASK_, JH_)                                                                                          \
  do {                                                                                                                     \
    VQA_ENSURE_LDS((oda_bwd_weight_mfma_staged_kernel<MASK_, JH_>), lds_st);                                               \
    VQA_LAUNCH((oda_bwd_weight_mfma_staged_kernel<MASK_, JH_>), dim3(SG), dim3(128 * nw * JH_), lds_st, s, vl, ql, dS, slab, \
               dc, B, N, L, G, spg);                                                                                       \
  } while (0)
      if (staged && dc.p8 > 0) {
        if (halves) VQA_K2_STAGED(true, 2); else VQA_K2_STAGED(true, 1);
      } else if (staged) {
        if (halves) VQA_K2_STAGED(false, 2); else VQA_K2_STAGED(false, 1);
#undef VQA_K2_STAGED
      } else if (dc.p8 > 0 && pk) {
        VQA_LAUNCH((oda_bwd_weight_mfma_kernel<true, true>), dim3(SG), dim3(128 * nw), lds_m, s, vl, ql, dS, slab, dc, B, N, L, G, spg);
      } else if (dc.p8 > 0) {
        VQA_LAUNCH(oda_bwd_weight_mfma_kernel<true>, dim3(SG), dim3(128 * nw), lds_m, s, vl, ql, dS, slab, dc, B, N, L, G, spg);
      } else {
        VQA_LAUNCH(oda_bwd_weight_mfma_kernel<false>, dim3(SG), dim3(128 * nw), lds_m, s, vl, ql, dS, slab, dc, B, N, L, G, spg);
      }
    } else if (oda_bits_mode(dc, B, N, L))
      VQA_LAUNCH((oda_bwd_weight_bits_kernel<G>), grid, dim3(nt), lds, s, vl, ql, dS, slab, dc, B, N, L, spg);
    else if (dc.p8 > 0)
      VQA_LAUNCH((oda_bwd_weight_kernel<G, true>), grid, dim3(nt), lds, s, vl, ql, dS, slab, dc, B, N, L, spg);
    else
      VQA_LAUNCH((oda_bwd_weight_kernel<G, false>), grid, dim3(nt), lds, s, vl, ql, dS, slab, dc, B, N, L, spg);
    const size_t n = (size_t)G * N * L;
    VQA_LAUNCH(oda_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, slab, d_w, n, SG);
  }
  VQA_LAUNCH(oda_dbias_kernel<G>, dim3(1), dim3(1024), 0, s, dS, d_bias, B * N);
  return check_launch("object_difference_attention_bwd");
}

}  // namespace vqa

using namespace vqa;

static int oda_check(const char* who, int B, int N, int L, int G, float p) {
  VQA_REQUIRE(B > 0 && N > 0 && L > 0 && G > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d L=%d G=%d", who, B, N, L, G);
  VQA_REQUIRE(G <= kOdaMaxG && N <= 128 && L <= 1024, VQA_E_UNSUPPORTED, "%s: needs G <= 8, N <= 128, L <= 1024 (G=%d N=%d L=%d)",
              who, G, N, L);
  VQA_REQUIRE(p >= 0.f && p < 1.f, VQA_E_BADARG, "%s: p_drop=%f outside [0,1)", who, (double)p);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "%s: B=%d exceeds 65535", who, B);
  return VQA_OK;
}

#define VQA_G_SWITCH(G, CALL) \
  switch (G) {                \
    case 1: return CALL(1);   \
    case 2: return CALL(2);   \
    case 3: return CALL(3);   \
    case 4: return CALL(4);   \
    case 5: return CALL(5);   \
    case 6: return CALL(6);   \
    case 7: return CALL(7);   \
    default: return CALL(8);  \
  }

extern "C" int vqa_object_difference_attention_fwd(const float* vl, const float* ql, const float* w, const float* bias,
                                                   float* logits, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B,
                                                   int N, int L, int G, vqa_stream_t stream) {
  VQA_REQUIRE(vl && ql && w && bias && logits, VQA_E_BADARG, "object_difference_attention_fwd: null pointer");
  int rc = oda_check("object_difference_attention_fwd", B, N, L, G, p_drop);
  if (rc != VQA_OK) return rc;
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  hipStream_t s = static_cast<hipStream_t>(stream);
#define CALL(G_) launch_fwd<G_>(vl, ql, w, bias, logits, dc, B, N, L, s)
  VQA_G_SWITCH(G, CALL)
#undef CALL
}

extern "C" size_t vqa_object_difference_attention_bwd_workspace_bytes(int B, int N, int L, int G) {
  if (B <= 0 || N <= 0 || L <= 0 || G <= 0) return 0;
  return (size_t)(oda_mfma_groups(B) > oda_groups(B) ? oda_mfma_groups(B) : oda_groups(B)) * G * N * L * sizeof(float);
}

extern "C" int vqa_object_difference_attention_bwd(const float* vl, const float* ql, const float* w,
                                                   const float* d_logits, float* d_vl, float* d_ql, float* d_w,
                                                   float* d_bias, void* workspace, size_t workspace_bytes, float p_drop,
                                                   uint64_t seed, const uint64_t* seed_ptr, int B, int N, int L, int G,
                                                   int gate_dvl, vqa_stream_t stream) {
  VQA_REQUIRE(vl && ql && w && d_logits && d_vl && d_ql && d_w && d_bias && workspace, VQA_E_BADARG,
              "object_difference_attention_bwd: null pointer");
  int rc = oda_check("object_difference_attention_bwd", B, N, L, G, p_drop);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(workspace_bytes >= vqa_object_difference_attention_bwd_workspace_bytes(B, N, L, G), VQA_E_BADARG,
              "object_difference_attention_bwd: workspace of %zu B is too small", workspace_bytes);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* slab = static_cast<float*>(workspace);
#define CALL(G_) launch_bwd<G_>(vl, ql, w, d_logits, d_vl, d_ql, d_w, d_bias, slab, dc, B, N, L, gate_dvl, s)
  VQA_G_SWITCH(G, CALL)
#undef CALL
}

extern "C" int vqa_object_difference_dropout_mask(float* mask, float p_drop, uint64_t seed, const uint64_t* seed_ptr,
                                                  int B, int N, int L,
                                                  vqa_stream_t stream) {
  VQA_REQUIRE(mask, VQA_E_BADARG, "object_difference_dropout_mask: null pointer");
  int rc = oda_check("object_difference_dropout_mask", B, N, L, 1, p_drop);
  if (rc != VQA_OK) return rc;
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  VQA_LAUNCH(oda_mask_kernel, dim3((N * L + 255) / 256, N, B), dim3(256), 0, static_cast<hipStream_t>(stream), mask,
                     dc, N, L, oda_bits_mode(dc, B, N, L));
  return check_launch("object_difference_dropout_mask");
}
