// K6 on the SPLIT engine: vqa_grouped_gemm_split -- the grouped launch of csrc/grouped_gemm.hip (every GEMM of a phase of the
// [B, .]-sized layers: MyLinear / putils.Linear / MutanFusion's vector sides, config/CoR2.py:94-122,133-134,170,180-189,
// putils/__init__.py:16-33,232-238; same VqaGemmProblem table, same forms, same slab / direct-output contract) with every fp32
// product formed from exact three-way bf16 splits of BOTH operands, six partial products on v_mfma_f32_16x16x32_bf16, fp32
// accumulation (csrc/gemm_f32_split.hpp: an fp32 GEMM's rounding, 2.67 x fewer matrix-pipe cycles than v_mfma_f32_32x32x2_f32).
//
// M = 512 rows leave no room for the 144 x 80 register tiles of the region-sized split kernels, and the three operand
// orientations (forward NT, data gradient NN, weight gradient TN) rule out one pre-packed plane image per weight.  So both
// operands are split WHILE THEY ARE STAGED into LDS, once per workgroup tile:
//   workgroup = 128 x BN outputs (BN = 128 or 160 per problem: whichever pads its N less -- 310 = 2 x 160, 2048 = 16 x 128) of one
//               contraction part, 8 waves of two kinds, one of each per SIMD:
//     waves 4..7 stage: per 32-deep step 256 threads fetch (128 + BN) x 32 fp32 (buffer loads: out-of-range rows, columns and k
//               come back as zeros; the loads of step s + 2 go out when step s + 1 has been staged), split them (9 VALU
//               instructions per pair of elements) and write the three bf16 planes of step s + 1 into the LDS stage that is not
//               being read;
//     waves 0..3 multiply: 2 (64-row halves: 4 blocks of 16) x 2 (BN / 2 columns: CB = 4 or 5 blocks), 16 CB accumulators each;
//               per step 3 (4 + CB) fragment reads and 24 CB MFMAs out of the other stage.  One barrier per step.
//   K-contiguous operands (A of NT / NN, B of NT) are stored in fragment order -- [16-row block][plane][lane][8 bf16] -- so a
//               fragment is one contiguous KiB (ds_read_b128); operands whose contraction index is the ROW in memory (B of NN,
//               both of TN) are stored as they lie -- [plane][k][mn] bf16 -- and come back as fragments through gfx950's
//               transposing read (ds_read_b64_tr_b16: four consecutive k of one column per lane), as in bilinear_dw_split.hip.
// LDS traffic per step is (128 + BN) x 32 x 6 bytes written + 4 x (4 + CB) x 3 KiB read = 163 KiB for 1920 matrix-pipe
// cycles (BN = 160): two thirds of the LDS rate, which is what bounds the tile from below -- a 64 x 64 tile would need 1.5 x it.
//
// Measured (tools/gs_probe.py, 256 tiles, one per CU): 2.3 us per step = 135-145 TFLOP/s at K = 2560 against 88-90 on the fp32
// MFMA kernel, and 9-12 us per work item that do not depend on K (launch, two memory latencies, 80 KiB of slab per item).  The
// step costs 1.47 us without the staging work and the difference is the staging's VALU instructions at their full issue time --
// whether they run in the multiplying waves' own instruction stream, interleaved MFMA by MFMA (the first form of this kernel:
// 2.23 us), or in other waves (this form): with every CU multiplying the chip runs at its power limit and work adds up.  With
// ~10-step parts (what fills 256 CUs from M = 512 rows) the fixed cost weighs as much as the loop: the phases of the CoR2 head
// come out between 0.77 and 1.36 x the fp32 MFMA launch's time, and head.py runs on this kernel the ones that win
// (Phase.SPLIT_PHASES: -15 us per step).
//
// Domain: all of fp32, by the repair path of gemm_f32_split.hpp -- an accumulator that comes out non-finite (an operand was
// Inf / NaN or within half a bf16 ulp of FLT_MAX) is recomputed as a plain fp32 dot product of the original operands.
#include "gemm_f32_split.hpp"

namespace vqa {
namespace {

using sp::f32x2;
using sp::f32x4;
using sp::u32x4;
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int kGsThreads = 512;   // 4 multiplying + 4 staging waves
constexpr int kGsBM = 128;
constexpr int kGsMaxGemms = VQA_GROUPED_GEMM_MAX;
// one LDS stage: A region (K-contiguous: 8 blocks x 3 KiB = 24576; row-contraction: 96 rows x 288) + B region (10 x 3 KiB = 30720;
// 96 x 352)
constexpr int kGsRegionA = 27648, kGsRegionB = 33792, kGsStage = kGsRegionA + kGsRegionB, kGsLds = 2 * kGsStage;

struct GsProbs {
  VqaGemmProblem p[kGsMaxGemms];
  int first[kGsMaxGemms + 1];  // first work item of each problem
  int n;
};

template <class T>
__device__ __forceinline__ const T& gs_kernarg() {   // (see grouped_gemm.hip: the table stays in the kernarg segment, scalar loads)
  return *(const T*)__builtin_amdgcn_kernarg_segment_ptr();
}

__device__ __forceinline__ DropCfg gs_drop(float p, uint64_t seed, const uint64_t* seed_ptr) {
  int p8 = (int)(p * 256.f + 0.5f);
  p8 = p8 < 0 ? 0 : (p8 > 255 ? 255 : p8);
  return DropCfg{(uint32_t)p8, 256.f / (256.f - (float)p8), seed, seed_ptr};
}
__device__ __forceinline__ float gs_act(float z, int act) {
  if (act == 1) return fmaxf(z, 0.f);
  if (act == 2) return 1.f / (1.f + expf(-z));
  return z;
}

// One operand of a tile.  KC: X[mn][k] (rows K-contiguous); MC: X[k][mn] (the contraction index is the row).  ROWS = tile extent
// along mn (128, or 32 CB).  An item is what one thread fetches with one load: two consecutive elements along the contiguous
// axis (SCALAR: two 4-byte loads, any alignment, odd extents); a thread holds ITEMS of them per step.
//   KC  item q: row 32 (q / 2) + t / 8, k = 4 (t % 8) + 2 (q % 2) + {0, 1}   -> 8 lanes cover a row's 128-byte line
//   MC  item q: e = 256 q + t, k = e / (ROWS / 2), mn = 2 (e % (ROWS / 2)) + {0, 1}   -> a wave covers 512 contiguous bytes
// Loads are buffer loads: the item's byte offset at k = 0 is a per-lane constant (kOob when its row / column lies outside the
// operand: the load then returns zero), the step's advance rides in the scalar offset -- no address arithmetic per step, no
// select on the loaded value; what a step costs on the VALU is the test "is this k still inside the part" (the last step's).
constexpr uint32_t kOob = 0x80000000u;   // (operands are below 2^31 bytes)
template <bool KC, int ROWS, bool SCALAR>
struct Operand {
  static constexpr int ITEMS = ROWS / 16;             // (ROWS x 32 elements / 256 threads / 2)
  static constexpr int HALF = ROWS / 2;               // MC: pairs per k row
  static constexpr int PITCH = ROWS == 128 ? 288 : 352;   // MC: bytes per k row in LDS (2 ROWS + pad; pitch mod 128 = 32 or 96:
                                                          // the 8 rows a transposing read touches per half-wave fall on all banks)
  rt::rsrc_t rs;
  uint32_t step_bytes;              // bytes per unit of k in the scalar offset
  uint32_t off0[ITEMS];             // byte offset of the item's first element at k = 0, or kOob
  uint32_t off1[SCALAR ? ITEMS : 1];   // SCALAR: of its second element
  f32x2 raw[ITEMS];

  __device__ __forceinline__ int kpos(int q, int t) const {   // k offset of item q inside a step
    if constexpr (KC) return 4 * (t & 7) + 2 * (q & 1);
    else return (256 * q + t) / HALF;
  }
  // p: the operand; mn_valid / k_valid: its extents; tile0: the tile's first row (KC) / column (MC)
  __device__ __forceinline__ void init(const float* p, int ld, int mn_valid, int k_valid, int tile0, int t) {
    rs = rt::make_rsrc(p, KC ? ((size_t)(mn_valid - 1) * ld + k_valid) * 4 : ((size_t)(k_valid - 1) * ld + mn_valid) * 4);
    step_bytes = KC ? 4u : 4u * (uint32_t)ld;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      if constexpr (KC) {
        const int row = tile0 + 32 * (q >> 1) + (t >> 3);
        const uint32_t o = ((uint32_t)row * (uint32_t)ld + (uint32_t)kpos(q, t)) * 4u;
        off0[q] = row < mn_valid ? o : kOob;
        if constexpr (SCALAR) off1[q] = row < mn_valid ? o + 4u : kOob;
      } else {
        const int e = 256 * q + t, col = tile0 + 2 * (e % HALF);
        const uint32_t o = ((uint32_t)(e / HALF) * (uint32_t)ld + (uint32_t)col) * 4u;
        off0[q] = col < mn_valid ? o : kOob;
        if constexpr (SCALAR) off1[q] = col + 1 < mn_valid ? o + 4u : kOob;
      }
    }
  }
  // request item q of the step at contraction offset kc into `dst`; `left` = how many k of it are still inside the part (and
  // the operand)
  __device__ __forceinline__ void fetch_item(f32x2& dst, int q, uint32_t soff, int left, int t) const {
    const int k = kpos(q, t);
    if constexpr (SCALAR) {
      const bool in1 = KC ? k + 1 < left : k < left;
      dst[0] = rt::ldg4(rs, k < left ? off0[q] : kOob, soff);
      dst[1] = rt::ldg4(rs, in1 ? off1[q] : kOob, soff);
    } else {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(k < left ? off0[q] : kOob), (int)soff, 0);   // (even extents)
      dst = __builtin_bit_cast(f32x2, v);
    }
  }
  __device__ __forceinline__ void fetch(f32x2 (&dst)[ITEMS], int kc, int lim, int t) const {
    const uint32_t soff = (uint32_t)kc * step_bytes;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) fetch_item(dst[q], q, soff, lim - kc, t);
  }
  // split item q (value v) and write its planes into `dst` (this operand's region of an LDS stage).  KC: an even item's planes
  // wait in `keep` for the odd item that completes the 8-byte slot.
  __device__ __forceinline__ void stage_item(f32x2 v, int q, char* dst, int t, uint32_t (&keep)[3]) const {
    uint32_t w[3];
    sp::split_pair<false>(v, w[0], w[1], w[2]);
    if constexpr (KC) {
      if ((q & 1) == 0) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) keep[pl] = w[pl];
      } else {
        const int rr = t >> 3, kq = t & 7;
        char* base = dst + ((rr >> 4) * 3) * 1024 + ((rr & 15) + 16 * (kq >> 1)) * 16 + 8 * (kq & 1);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(base + ((q - 1) * 3 + pl) * 1024) = make_uint2(keep[pl], w[pl]);
      }
    } else {
      const int e = 256 * q + t;
      char* at = dst + (e / HALF) * PITCH + 4 * (e % HALF);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint32_t*>(at + pl * 32 * PITCH) = w[pl];
    }
  }
  __device__ __forceinline__ void stage(const f32x2 (&src)[ITEMS], char* dst, int t) const {
    uint32_t keep[3];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) stage_item(src[q], q, dst, t, keep);
  }
  // fragment (plane, 16-wide block) of the staged step: lane (r, g) gets mn = 16 block + r, k = 8 g .. 8 g + 7
  __device__ __forceinline__ u32x4 frag(const char* src, int plane, int block, int lane) const {
    if constexpr (KC) {
      return *reinterpret_cast<const u32x4*>(src + (block * 3 + plane) * 1024 + lane * 16);
    } else {
      const int r16 = lane & 15, gq = lane >> 4;
      const char* at = src + (plane * 32 + 8 * gq + (r16 >> 2)) * PITCH + (16 * block + 4 * (r16 & 3)) * 2;
      const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(at));
      const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(at + 4 * PITCH));
      const uint2 a = __builtin_bit_cast(uint2, lo), b = __builtin_bit_cast(uint2, hi);
      return u32x4{a.x, a.y, b.x, b.y};
    }
  }
};

// element (mn, k) of an operand as the fp32 original, zero outside its extents: the repair path's operand
template <bool KC>
__device__ __forceinline__ float gs_elem(const float* p, int ld, int mn_valid, int k_valid, int mn, int k) {
  if (mn >= mn_valid || k >= k_valid) return 0.f;
  return KC ? p[(size_t)mn * ld + k] : p[(size_t)k * ld + mn];
}

template <int CB, bool A_KC, bool B_KC, bool A_SCALAR>
__device__ __forceinline__ void gs_tile(const VqaGemmProblem& pr, const int m0, const int n0, const int split, char* smem) {
  constexpr int BN = 32 * CB;
  const int k_begin = split * pr.ksplit, k_end = min(pr.K, k_begin + pr.ksplit);
  const int lim_a = min(k_end, pr.Ka), lim_b = min(k_end, pr.Kb);
  const int steps = (k_end - k_begin + 31) >> 5;
  const int mn_a = A_KC ? pr.M : pr.Ma, mn_b = B_KC ? pr.N : pr.Nb;
  const bool want_colsum = !A_KC && (pr.colsum != nullptr || pr.colsum_out != nullptr) && n0 == 0;   // (wave-uniform)
  // Two kinds of waves, one of each on every SIMD (the role is a scalar: no divergence, the buffer descriptors stay SGPR quads):
  //   waves 4..7 STAGE: fetch (128 + BN) x 32 fp32 per step, split them, write the bf16 planes of step s + 1 into the LDS stage
  //              the MFMA waves are not reading, request step s + 2 -- ~240 VALU instructions per step and wave;
  //   waves 0..3 MULTIPLY: 3 (4 + CB) fragment reads and 24 CB MFMAs per step out of the other stage.
  // A wave's own VALU work does not hide behind its own MFMAs (measured: every split instruction added its issue time to the
  // step -- 2.23 us per step against 1.47 without the staging); another wave's does.  One barrier per step.
  const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  if (role != 0) {
    const int t = (int)threadIdx.x - 256, lane = t & 63, wave = t >> 6;
    Operand<A_KC, kGsBM, A_SCALAR> oa;
    Operand<B_KC, BN, false> ob;
    oa.init(pr.A, pr.lda, mn_a, pr.Ka, m0, t);
    ob.init(pr.B, pr.ldb, mn_b, pr.Kb, n0, t);
    constexpr int IA = Operand<A_KC, kGsBM, A_SCALAR>::ITEMS, IB = Operand<B_KC, BN, false>::ITEMS;
    const float colsum_on = want_colsum ? 1.f : 0.f;
    f32x2 colsum = f32x2{0.f, 0.f};
    {
      // the first two steps are requested together (one memory latency instead of two ahead of the first MFMA)
      f32x2 first_a[IA], first_b[IB];
      oa.fetch(first_a, k_begin, lim_a, t);
      ob.fetch(first_b, k_begin, lim_b, t);
      oa.fetch(oa.raw, k_begin + 32, lim_a, t);
      ob.fetch(ob.raw, k_begin + 32, lim_b, t);
      if constexpr (!A_KC) {
#pragma unroll
        for (int q = 0; q < IA; ++q) colsum += first_a[q] * colsum_on;
      }
      oa.stage(first_a, smem, t);
      ob.stage(first_b, smem + kGsRegionA, t);
    }
    __syncthreads();
    // Past the last step the loads are out of range (zeros) and the stage written is never read: no branch in the loop.
    for (int s = 0; s < steps; ++s) {
      char* nxt = smem + ((s + 1) & 1) * kGsStage;
      const int kc2 = k_begin + 32 * (s + 2);
      const uint32_t soff_a = (uint32_t)kc2 * oa.step_bytes, soff_b = (uint32_t)kc2 * ob.step_bytes;
      uint32_t keep[3];
#pragma unroll
      for (int q = 0; q < IA; ++q) {
        if constexpr (!A_KC) colsum += oa.raw[q] * colsum_on;
        oa.stage_item(oa.raw[q], q, nxt, t, keep);
        oa.fetch_item(oa.raw[q], q, soff_a, lim_a - kc2, t);
      }
#pragma unroll
      for (int q = 0; q < IB; ++q) {
        ob.stage_item(ob.raw[q], q, nxt + kGsRegionA, t, keep);
        ob.fetch_item(ob.raw[q], q, soff_b, lim_b - kc2, t);
      }
      __syncthreads();
    }
    // a thread summed the elements A[k][m0 + 2 (t % 64) + {0, 1}] over the k rows 4 q + wave of every step: add the four waves
    float* red = reinterpret_cast<float*>(smem);   // (the last step's barrier has passed: the stages are free)
    if (want_colsum) {
      red[wave * 128 + 2 * lane] = colsum[0];
      red[wave * 128 + 2 * lane + 1] = colsum[1];
    }
    __syncthreads();
    if (want_colsum && t < 128) {
      const float total = (red[t] + red[128 + t]) + (red[256 + t] + red[384 + t]);
      float* cs = pr.colsum_out != nullptr ? pr.colsum_out : pr.colsum + (size_t)(pr.slab_base + split) * pr.M;
      if (m0 + t < pr.M) cs[m0 + t] = total;
    }
    return;
  }

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, wm = wave >> 1, wn = wave & 1, r16 = lane & 15, gq = lane >> 4;
  const Operand<A_KC, kGsBM, A_SCALAR> oa{};   // (fragment addressing only)
  const Operand<B_KC, BN, false> ob{};
  f32x4 acc[4][CB];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (int s = 0; s < steps; ++s) {
    const char* cur = smem + (s & 1) * kGsStage;
    u32x4 a[3][4], b[3][CB];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < 4; ++i) a[pl][i] = oa.frag(cur, pl, 4 * wm + i, lane);
      b[pl][0] = ob.frag(cur + kGsRegionA, pl, CB * wn, lane);
    }
#pragma unroll
    for (int j = 1; j < CB; ++j)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b[pl][j] = ob.frag(cur + kGsRegionA, pl, CB * wn + j, lane);
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};      // the six partial products of weight >= 2^-16
    // (j, product, i): consecutive products into one accumulator stay four MFMAs apart
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][j] = sp::mfma_bf16(b[PB[k]][j], a[PA[k]][i], acc[i][j]);   // D^T: a lane holds 4 columns of a row
    __syncthreads();
  }
  __syncthreads();   // (the staging waves' column-sum exchange)

  // ---- outputs: lane (r16, gq) holds C[m0 + 64 wm + 16 i + r16][n0 + 16 (CB wn + j) + 4 gq + 0..3]
  uint32_t top = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j) top = max(top, sp::abs_bits_max(acc[i][j]));
  if (top >= 0x7F800000u) {   // the repair path (never taken on finite, well-scaled data)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < CB; ++j) {
        f32x4 fixed = f32x4{0.f, 0.f, 0.f, 0.f};
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (the accumulator arrays are indexed by the rolled loop's counters through a select chain: no scratch)
          float v = 0.f;
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < CB; ++jj)
              if (ii == i && jj == j) v = acc[ii][jj][e];
          if (sp::nonfinite(v)) {
            const int m = m0 + 64 * wm + 16 * i + r16, n = n0 + 16 * (CB * wn + j) + 4 * gq + e;
            float s = 0.f;
            for (int k = k_begin; k < k_end; ++k)
              s = fmaf(gs_elem<A_KC>(pr.A, pr.lda, mn_a, pr.Ka, m, k), gs_elem<B_KC>(pr.B, pr.ldb, mn_b, pr.Kb, n, k), s);
            v = s;
            any = true;
          }
          fixed[e] = v;
        }
        if (any) {
#pragma unroll
          for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int jj = 0; jj < CB; ++jj)
              if (ii == i && jj == j) acc[ii][jj] = fixed;
        }
      }
  }

  if (pr.out != nullptr) {
    // direct output: the layer's epilogue on the accumulators (bias, activation, gate of the layer in front, dropout)
    const bool drop = pr.p_drop > 0.f;
    DropCfg dc{};
    if (drop) dc = gs_drop(pr.p_drop, pr.seed, pr.seed_ptr);
    const bool vec = (pr.N & 1) == 0 && (pr.ldo & 1) == 0 && (reinterpret_cast<uintptr_t>(pr.out) & 7u) == 0;
#pragma unroll
    for (int j = 0; j < CB; ++j) {
      const int col = n0 + 16 * (CB * wn + j) + 4 * gq;
      float bv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[e] = (pr.bias != nullptr && col + e < pr.N) ? pr.bias[col + e] : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = m0 + 64 * wm + 16 * i + r16;
        if (row < pr.M && col < pr.N) {
          float z[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            z[e] = gs_act(acc[i][j][e] + bv[e], pr.act);
            if (col + e < pr.N) {
              if (pr.gate != 0) {
                const float y = pr.gate_y[(size_t)row * pr.ld_gate + col + e];
                z[e] = pr.gate == 1 ? (y > 0.f ? z[e] * pr.gate_scale : 0.f) : z[e] * y * (1.f - y);
              }
              if (drop) z[e] *= drop_one(pr.drop_base + (uint32_t)row * pr.drop_ld + (uint32_t)(col + e), dc);
            }
          }
          float* o = pr.out + (size_t)row * pr.ldo + col;
          if (vec) {
            st2(o, make_float2(z[0], z[1]));
            if (col + 2 < pr.N) st2(o + 2, make_float2(z[2], z[3]));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < pr.N) o[e] = z[e];
          }
        }
      }
    }
  } else {
    float* __restrict__ dst = pr.slab + (size_t)(pr.slab_base + split) * pr.slab_stride;
    const bool vec = (pr.N & 1) == 0 && (pr.slab_stride & 1) == 0 && (reinterpret_cast<uintptr_t>(pr.slab) & 7u) == 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = m0 + 64 * wm + 16 * i + r16;
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const int col = n0 + 16 * (CB * wn + j) + 4 * gq;
        if (row < pr.M && col < pr.N) {
          float* o = dst + (size_t)row * pr.N + col;
          if (vec) {
            st2(o, make_float2(acc[i][j][0], acc[i][j][1]));
            if (col + 2 < pr.N) st2(o + 2, make_float2(acc[i][j][2], acc[i][j][3]));
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (col + e < pr.N) o[e] = acc[i][j][e];
          }
        }
      }
    }
  }
}

__host__ __device__ inline int gs_tile_cols(int N) {   // the tile width that pads N less (ties: the wider tile, fewer items)
  const int w128 = (N + 127) / 128 * 128, w160 = (N + 159) / 160 * 160;
  return w160 <= w128 ? 160 : 128;
}

template <int CB>
__device__ __forceinline__ void gs_forms(const VqaGemmProblem& pr, int m0, int n0, int split, char* smem) {
  if (pr.form == 0) gs_tile<CB, true, true, false>(pr, m0, n0, split, smem);          // NT
  else if (pr.form == 1) gs_tile<CB, true, false, false>(pr, m0, n0, split, smem);    // NN
  else if (pr.form == 2) gs_tile<CB, false, false, false>(pr, m0, n0, split, smem);   // TN
  else if (pr.form == 3) gs_tile<CB, true, false, true>(pr, m0, n0, split, smem);     // NN, A 4-byte aligned / odd extents
  else gs_tile<CB, false, false, true>(pr, m0, n0, split, smem);                      // TN, A 4-byte aligned / odd extents
}

__global__ __launch_bounds__(kGsThreads, 1) void grouped_gemm_split_kernel(GsProbs g_arg, int items) {
  const GsProbs& g = gs_kernarg<GsProbs>();
  extern __shared__ __attribute__((aligned(16))) char gs_smem[];
  const int bid = xcd_remap(blockIdx.x, items);
  int p = 0;
  while (p + 1 < g.n && g.first[p + 1] <= bid) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const VqaGemmProblem& pr = g.p[p];
  const int local = bid - g.first[p];
  const int bn = gs_tile_cols(pr.N);
  const int tiles_n = (pr.N + bn - 1) / bn, tiles_m = (pr.M + kGsBM - 1) / kGsBM;
  const int split = local / (tiles_m * tiles_n), tl = local % (tiles_m * tiles_n);
  const int m0 = (tl / tiles_n) * kGsBM, n0 = (tl % tiles_n) * bn;
  if (bn == 160) gs_forms<5>(pr, m0, n0, split, gs_smem);
  else gs_forms<4>(pr, m0, n0, split, gs_smem);
}

}  // namespace
}  // namespace vqa

using namespace vqa;

extern "C" int vqa_grouped_gemm_split_tile_cols(int N) { return N > 0 ? gs_tile_cols(N) : 0; }

extern "C" int vqa_grouped_gemm_split(const VqaGemmProblem* problems, int n, vqa_stream_t stream) {
  VQA_REQUIRE(problems != nullptr && n >= 1 && n <= kGsMaxGemms, VQA_E_BADARG, "grouped_gemm_split: 1..%d problems (got %d)", kGsMaxGemms, n);
  GsProbs g{};
  g.n = n;
  int items = 0;
  for (int i = 0; i < n; ++i) {
    VqaGemmProblem p = problems[i];
    VQA_REQUIRE(p.A && p.B && (p.slab || p.out), VQA_E_BADARG, "grouped_gemm_split[%d]: null pointer", i);
    VQA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.form >= 0 && p.form <= 4, VQA_E_BADARG,
                "grouped_gemm_split[%d]: bad sizes M=%d N=%d K=%d form=%d", i, p.M, p.N, p.K, p.form);
    VQA_REQUIRE(p.ksplit > 0 && p.ksplit % 32 == 0 && p.slab_base >= 0 && (p.out != nullptr || p.slab_stride >= (long long)p.M * p.N),
                VQA_E_BADARG, "grouped_gemm_split[%d]: ksplit %d must be a positive multiple of 32, slab_stride >= M*N", i, p.ksplit);
    if (p.out != nullptr) {
      VQA_REQUIRE(p.ksplit >= p.K, VQA_E_BADARG, "grouped_gemm_split[%d]: a direct output needs the contraction in one part", i);
      VQA_REQUIRE(p.ldo >= p.N && p.act >= 0 && p.act <= 2 && p.gate >= 0 && p.gate <= 2 && (p.gate == 0 || p.gate_y != nullptr) &&
                      p.p_drop >= 0.f && p.p_drop < 1.f,
                  VQA_E_BADARG, "grouped_gemm_split[%d]: bad direct-output epilogue", i);
    } else {
      VQA_REQUIRE(p.colsum_out == nullptr, VQA_E_BADARG, "grouped_gemm_split[%d]: colsum_out needs a direct output", i);
    }
    if (p.Ka <= 0) p.Ka = p.K;
    if (p.Kb <= 0) p.Kb = p.K;
    if (p.Ma <= 0) p.Ma = p.M;
    if (p.Nb <= 0) p.Nb = p.N;
    // 8-byte operand loads, as vqa_grouped_gemm: even leading dimensions, 8-byte aligned bases, even extents along the
    // contiguous axis (forms 3 / 4 read A with 4-byte loads: no requirement on A)
    const bool a_kc = p.form != 2 && p.form != 4, b_kc = p.form == 0, a_free = p.form >= 3;
    VQA_REQUIRE(aligned(p.A, 4) && aligned(p.B, 8) && p.ldb % 2 == 0 && (b_kc ? p.Kb : p.Nb) % 2 == 0 && (b_kc ? p.Kb : p.Nb) >= 2 &&
                    (a_free || (p.lda % 2 == 0 && aligned(p.A, 8) && (a_kc ? p.Ka : p.Ma) % 2 == 0 && (a_kc ? p.Ka : p.Ma) >= 2)),
                VQA_E_UNSUPPORTED,
                "grouped_gemm_split[%d]: operands need even leading dimensions / contiguous extents and 8-byte aligned bases "
                "(form %d lda=%d ldb=%d)", i, p.form, p.lda, p.ldb);
    VQA_REQUIRE((p.colsum == nullptr && p.colsum_out == nullptr) || p.form == 2 || p.form == 4, VQA_E_BADARG,
                "grouped_gemm_split[%d]: column sums exist for the TN forms only", i);
    // element indices are 32-bit inside the kernel
    const long long rows_a = a_kc ? p.M : p.Ka, rows_b = b_kc ? p.N : p.Kb;
    VQA_REQUIRE(rows_a * p.lda < (1LL << 29) && rows_b * p.ldb < (1LL << 29), VQA_E_UNSUPPORTED,
                "grouped_gemm_split[%d]: operand too large", i);
    g.p[i] = p;
    g.first[i] = items;
    const int bn = gs_tile_cols(p.N);
    const long tiles = (long)((p.M + kGsBM - 1) / kGsBM) * ((p.N + bn - 1) / bn) * ((p.K + p.ksplit - 1) / p.ksplit);
    VQA_REQUIRE(items + tiles < (1L << 24), VQA_E_UNSUPPORTED, "grouped_gemm_split: too many tiles");
    items += (int)tiles;
  }
  g.first[n] = items;
  VQA_ENSURE_LDS(grouped_gemm_split_kernel, kGsLds);
  VQA_LAUNCH(grouped_gemm_split_kernel, dim3(items), dim3(kGsThreads), kGsLds, static_cast<hipStream_t>(stream), g, items);
  return check_launch("grouped_gemm_split");
}
