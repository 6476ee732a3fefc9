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
#include "gemm_f32_rt.hpp"

namespace vqa {

constexpr int kGMaxProbs = VQA_GROUPED_MAX;
constexpr int kGMaxGemms = VQA_GROUPED_GEMM_MAX;

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

// ================================================================================================ DMA tile engine
// Round 4.  What bound the 64x64 LDS-tile form above (ablations in gemm_f32_mfma.hpp: the store side of the staging pass and
// the wait for its loads, not the MFMAs) is removed rather than tuned:
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass, no VALU);
//     measured on the MI355X (tools/dma_probe.hip): the 16-byte DMA takes 4-byte aligned sources at full rate, and a lane
//     whose offset is out of the descriptor's range writes ZERO to its LDS slot (range check per dword) -- that is the
//     contraction tail and the odd / 8-byte-aligned row pitches of these layers (310, 510, 155 floats) for free;
//   * v_mfma_f32_16x16x4_f32 with every fragment a 16-byte LDS read (the register-tile engine's operand maps,
//     gemm_f32_rt.hpp): a K-contiguous operand row gives a lane the 4 contraction steps of a 16-deep chunk in one read; an
//     MN-contiguous row gives it one step of FOUR accumulator blocks (the output index is permuted: lane r owns columns
//     4 r .. 4 r + 3 of a 64-wide span, which come back as 16-byte stores).  6-8 reads per 32-64 MFMAs;
//   * 128 x 64 (NT, NN) / 128 x 128 (TN) workgroup tiles, 32-deep stages in a two-stage ring, ONE barrier per stage; two
//     workgroups per CU (two waves per SIMD) cover each other's barrier and DMA waits.
// LDS images: a K-contiguous operand tile is [rows][32] with the eight 16-byte pieces of a row XOR-swizzled by (row >> 1) & 7
// (the DMA writes lane-linear, so the swizzle is applied to the SOURCE piece a lane fetches): the ds_read_b128 fragment reads
// -- 16 rows x one piece per 16-lane group -- are conflict-free.  An MN-contiguous tile is [32][W] as it lies in memory
// (W = 64 / 128: a fragment read covers whole 256-byte bank rows).
namespace gg {
using rt::f32x4;
using rt::rsrc_t;
constexpr int kBK = 32;                      // contraction depth of a stage
constexpr uint32_t kOOB = 0x80000000u;       // a byte offset no descriptor of ours reaches: the DMA writes zeros
constexpr int kStageFloats = 2 * 32 * 128;   // largest form (TN: two [32][128] images)
constexpr size_t kLdsBytes = 2 * kStageFloats * sizeof(float);
typedef __attribute__((address_space(3))) void* lds_t;

__device__ __forceinline__ void dma16(rsrc_t rs, float* wave_dst, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_t)wave_dst, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ f32x4 lds16(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// K-contiguous operand: tile rows row0 .. row0 + ROWS - 1, contraction k0 .. k0 + 31 -> [ROWS][32] (swizzled pieces).
// One DMA instruction = 8 rows x 128 B; wave w issues instructions w, w + 4, ...
template <int ROWS>
struct StageKC {
  static constexpr int NI = ROWS / 32;
  uint32_t voff[NI];
  int kloc[NI];      // first contraction index (within the stage) of the piece this lane fetches
  __device__ __forceinline__ void setup(int row0, int ld, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int rl = 8 * (wave + 4 * q) + (lane >> 3);
      const int piece = (lane & 7) ^ ((rl >> 1) & 7);
      voff[q] = ((uint32_t)(row0 + rl) * (uint32_t)ld + 4u * (uint32_t)piece) * 4u;
      kloc[q] = 4 * piece;
    }
  }
  // rem: valid contraction steps of this stage (>= 32: all of them)
  __device__ __forceinline__ void issue(rsrc_t rs, float* img, uint32_t k0, int rem, int wave) const {
#pragma unroll
    for (int q = 0; q < NI; ++q)
      dma16(rs, img + (wave + 4 * q) * 256, rem >= kBK || kloc[q] < rem ? voff[q] : kOOB, k0 * 4u);
  }
  // a piece that straddles the end of the contraction holds 1..3 values of the next row: zero them (after the DMA landed)
  __device__ __forceinline__ void fix(float* img, int rem, int wave, int lane) const {
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      const int left = rem - kloc[q];
      if (left > 0 && left < 4) {
        float* slot = img + (wave + 4 * q) * 256 + 4 * lane;
        for (int e = left; e < 4; ++e) slot[e] = 0.f;
      }
    }
  }
};
// MN-contiguous operand: contraction rows k0 .. k0 + 31, columns col0 .. col0 + W - 1 -> [32][W].  One DMA instruction =
// 1024 / (4 W) rows.  Columns past the operand's extent read whatever follows (zero past the buffer): they only reach
// accumulators that are never stored.
template <int W>
struct StageMC {
  static constexpr int LPR = W / 4, RPI = 64 / LPR, NI = kBK / RPI / 4;
  uint32_t voff[NI];
  int kloc[NI];
  __device__ __forceinline__ void setup(int col0, int ld, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      kloc[q] = (wave + 4 * q) * RPI + lane / LPR;
      voff[q] = ((uint32_t)kloc[q] * (uint32_t)ld + (uint32_t)col0 + 4u * (uint32_t)(lane % LPR)) * 4u;
    }
  }
  __device__ __forceinline__ void issue(rsrc_t rs, float* img, uint32_t k0, int ld, int rem, int wave) const {
#pragma unroll
    for (int q = 0; q < NI; ++q)
      dma16(rs, img + (wave + 4 * q) * 256, kloc[q] < rem ? voff[q] : kOOB, k0 * (uint32_t)ld * 4u);
  }
};

__device__ __forceinline__ rsrc_t operand_rsrc(const float* base, int rows, int ld, int extent) {
  return rt::make_rsrc(base, ((size_t)(rows - 1) * (size_t)ld + (size_t)extent) * sizeof(float));
}

// What happens to a finished accumulator element / 4-vector: slab store, or the layer's epilogue (direct output).
struct Sink {
  const VqaGemmProblem& pr;
  float* dst;       // slab of this split (row pitch N) or the direct output (row pitch ldo)
  int ld;
  bool direct, drop;
  DropCfg dc;
  __device__ __forceinline__ Sink(const VqaGemmProblem& p, int split) : pr(p), dc{} {
    direct = p.out != nullptr;
    dst = direct ? p.out : p.slab + (size_t)(p.slab_base + split) * p.slab_stride;
    ld = direct ? p.ldo : p.N;
    drop = direct && p.p_drop > 0.f;
    if (drop) dc = make_drop_dev(p.p_drop, p.seed, p.seed_ptr);
  }
  __device__ __forceinline__ float finish(int row, int col, float z) const {
    z = act_fwd_g(z + (pr.bias != nullptr ? pr.bias[col] : 0.f), pr.act);
    if (pr.gate != 0) {
      const float y = pr.gate_y[(size_t)row * pr.ld_gate + col];
      z = pr.gate == 1 ? (y > 0.f ? z * pr.gate_scale : 0.f) : z * y * (1.f - y);
    }
    if (drop) z *= drop_one(pr.drop_base + (uint32_t)row * pr.drop_ld + (uint32_t)col, dc);
    return z;
  }
  __device__ __forceinline__ void one(int row, int col, float v) const {
    if (row < pr.M && col < pr.N) dst[(size_t)row * ld + col] = direct ? finish(row, col, v) : v;
  }
  __device__ __forceinline__ void four(int row, int col, f32x4 v) const {   // columns col .. col + 3
    if (row >= pr.M) return;
    float* p = dst + (size_t)row * ld + col;
    if (col + 3 < pr.N) {
      if (direct) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ve = v[e];
          v[e] = finish(row, col + e, ve);
        }
      }
      *reinterpret_cast<f32x4*>(p) = v;      // (16-byte stores take 4-byte aligned addresses, like the loads)
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (col + e < pr.N) {
          const float ve = v[e];
          p[e] = direct ? finish(row, col + e, ve) : ve;
        }
    }
  }
};

// The stage loop shared by the three forms: `issue(s, buf)` starts the DMAs of stage s, `fix(s, buf)` repairs a tail stage
// after they landed, `compute(buf)` runs the MFMAs of a stage.  Stage s + 1 is in flight under the MFMAs of stage s.
// tune (VQA_GG_TUNE, ablation builds of the measurement only): 1 = no DMA after the first stage, 2 = no MFMA stage work.
template <class Issue, class Fix, class Compute>
__device__ __forceinline__ void stage_loop(int nst, float* smem, int tune, Issue issue, Fix fix, Compute compute) {
  issue(0, smem);
  for (int s = 0; s < nst; ++s) {
    float* buf = smem + (s & 1) * kStageFloats;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fix(s, buf);
    __syncthreads();       // stage s is complete for every wave, and every wave is done reading the other buffer
    if (s + 1 < nst && (tune & 1) == 0) issue(s + 1, smem + ((s + 1) & 1) * kStageFloats);
    if ((tune & 2) == 0) compute(buf);
  }
}

// ---- NT: C[m][n] = sum_k A[m][k] B[n][k]; tile 128 x 64, waves 2 x 2 of 64 x 32.  The B rows are fed as the MFMA's first operand:
// the accumulator block is then the TRANSPOSED 16 x 16 tile and a lane holds 4 consecutive output columns of one row (16-byte stores).
__device__ __forceinline__ void tile_nt(const VqaGemmProblem& pr, int m0, int n0, int split, float* smem, int tune) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
  const int k_begin = split * pr.ksplit, k_end = min(pr.K, k_begin + pr.ksplit);
  const int limA = min(pr.Ka, k_end), limB = min(pr.Kb, k_end);
  const rsrc_t Ab = operand_rsrc(pr.A, pr.M, pr.lda, pr.Ka), Bb = operand_rsrc(pr.B, pr.N, pr.ldb, pr.Kb);
  StageKC<128> sa;
  StageKC<64> sb;
  sa.setup(m0, pr.lda, wave, lane);
  sb.setup(n0, pr.ldb, wave, lane);
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sw = (r >> 1) & 7;
  const float* a_rd = smem + (wm * 64 + r) * kBK;              // + 16 i rows, + piece * 4
  const float* b_rd = smem + 128 * kBK + (wn * 32 + r) * kBK;
  stage_loop((k_end - k_begin + kBK - 1) / kBK, smem, tune,
             [&](int s, float* buf) {
               const int k0 = k_begin + s * kBK;
               sa.issue(Ab, buf, (uint32_t)k0, limA - k0, wave);
               sb.issue(Bb, buf + 128 * kBK, (uint32_t)k0, limB - k0, wave);
             },
             [&](int s, float* buf) {
               const int k0 = k_begin + s * kBK;
               if (limA - k0 < kBK) sa.fix(buf, limA - k0, wave, lane);
               if (limB - k0 < kBK) sb.fix(buf + 128 * kBK, limB - k0, wave, lane);
             },
             [&](float* buf) {
               const int o = (int)(buf - smem);
#pragma unroll
               for (int kc = 0; kc < 2; ++kc) {
                 const int pc = 4 * (((4 * kc + g) ^ sw));
                 f32x4 a[4], b[2];
#pragma unroll
                 for (int i = 0; i < 4; ++i) a[i] = lds16(a_rd + o + 16 * i * kBK + pc);
#pragma unroll
                 for (int j = 0; j < 2; ++j) b[j] = lds16(b_rd + o + 16 * j * kBK + pc);
#pragma unroll
                 for (int kb = 0; kb < 4; ++kb)
#pragma unroll
                   for (int i = 0; i < 4; ++i)
#pragma unroll
                     for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][kb], a[i][kb], acc[i][j], 0, 0, 0);
               }
             });
  const Sink sink(pr, split);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      sink.four(m0 + wm * 64 + 16 * i + r, n0 + wn * 32 + 16 * j + 4 * g, acc[i][j]);
}

// ---- NN: C[m][n] = sum_k A[m][k] B[k][n]; tile 128 x 64, waves 4 x 1 of 32 x 64 (block c of a wave = columns 4 r + c)
__device__ __forceinline__ void tile_nn(const VqaGemmProblem& pr, int m0, int n0, int split, float* smem, int tune) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int k_begin = split * pr.ksplit, k_end = min(pr.K, k_begin + pr.ksplit);
  const int limA = min(pr.Ka, k_end), limB = min(pr.Kb, k_end);
  const rsrc_t Ab = operand_rsrc(pr.A, pr.M, pr.lda, pr.Ka), Bb = operand_rsrc(pr.B, pr.Kb, pr.ldb, pr.Nb);
  StageKC<128> sa;
  StageMC<64> sb;
  sa.setup(m0, pr.lda, wave, lane);
  sb.setup(n0, pr.ldb, wave, lane);
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[i][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int sw = (r >> 1) & 7;
  const float* a_rd = smem + (wave * 32 + r) * kBK;
  const float* b_rd = smem + 128 * kBK + (4 * g) * 64 + 4 * r;    // + (16 kc + s) rows
  stage_loop((k_end - k_begin + kBK - 1) / kBK, smem, tune,
             [&](int s, float* buf) {
               const int k0 = k_begin + s * kBK;
               sa.issue(Ab, buf, (uint32_t)k0, limA - k0, wave);
               sb.issue(Bb, buf + 128 * kBK, (uint32_t)k0, pr.ldb, limB - k0, wave);
             },
             [&](int s, float* buf) {
               const int k0 = k_begin + s * kBK;
               if (limA - k0 < kBK) sa.fix(buf, limA - k0, wave, lane);
             },
             [&](float* buf) {
               const int o = (int)(buf - smem);
#pragma unroll
               for (int kc = 0; kc < 2; ++kc) {
                 const int pc = 4 * (((4 * kc + g) ^ sw));
                 f32x4 a[2], b[4];
#pragma unroll
                 for (int i = 0; i < 2; ++i) a[i] = lds16(a_rd + o + 16 * i * kBK + pc);
#pragma unroll
                 for (int st = 0; st < 4; ++st) b[st] = lds16(b_rd + o + (16 * kc + st) * 64);
#pragma unroll
                 for (int st = 0; st < 4; ++st)
#pragma unroll
                   for (int i = 0; i < 2; ++i)
#pragma unroll
                     for (int c = 0; c < 4; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i][st], b[st][c], acc[i][c], 0, 0, 0);
               }
             });
  const Sink sink(pr, split);
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      sink.four(m0 + wave * 32 + 16 * i + 4 * g + t, n0 + 4 * r, f32x4{acc[i][0][t], acc[i][1][t], acc[i][2][t], acc[i][3][t]});
}

// ---- TN: C[m][n] = sum_k A[k][m] B[k][n]; tile 128 x 128, waves 2 x 2 of 64 x 64; both output indices permuted: block
// (ca, cb) = rows 4 i + ca (i the MFMA row index), columns 4 r + cb.  Column sums of A (the bias gradient) from the fragments.
__device__ __forceinline__ void tile_tn(const VqaGemmProblem& pr, int m0, int n0, int split, float* smem, int tune) {
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
  const int k_begin = split * pr.ksplit, k_end = min(pr.K, k_begin + pr.ksplit);
  const int limA = min(pr.Ka, k_end), limB = min(pr.Kb, k_end);
  const rsrc_t Ab = operand_rsrc(pr.A, pr.Ka, pr.lda, pr.Ma), Bb = operand_rsrc(pr.B, pr.Kb, pr.ldb, pr.Nb);
  StageMC<128> sa, sb;
  sa.setup(m0, pr.lda, wave, lane);
  sb.setup(n0, pr.ldb, wave, lane);
  f32x4 acc[4][4];
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) acc[ca][cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool want_colsum = (pr.colsum != nullptr || pr.colsum_out != nullptr) && n0 == 0 && wn == 0;   // (wave-uniform)
  f32x4 cs = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* a_rd = smem + (4 * g) * 128 + 64 * wm + 4 * r;
  const float* b_rd = smem + kBK * 128 + (4 * g) * 128 + 64 * wn + 4 * r;
  stage_loop((k_end - k_begin + kBK - 1) / kBK, smem, tune,
             [&](int s, float* buf) {
               const int k0 = k_begin + s * kBK;
               sa.issue(Ab, buf, (uint32_t)k0, pr.lda, limA - k0, wave);
               sb.issue(Bb, buf + kBK * 128, (uint32_t)k0, pr.ldb, limB - k0, wave);
             },
             [&](int, float*) {},
             [&](float* buf) {
               const int o = (int)(buf - smem);
#pragma unroll
               for (int kc = 0; kc < 2; ++kc) {
                 f32x4 a[4], b[4];
#pragma unroll
                 for (int st = 0; st < 4; ++st) {
                   a[st] = lds16(a_rd + o + (16 * kc + st) * 128);
                   b[st] = lds16(b_rd + o + (16 * kc + st) * 128);
                 }
                 if (want_colsum) cs += (a[0] + a[1]) + (a[2] + a[3]);
#pragma unroll
                 for (int st = 0; st < 4; ++st)
#pragma unroll
                   for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                     for (int cb = 0; cb < 4; ++cb)
                       acc[ca][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[st][ca], b[st][cb], acc[ca][cb], 0, 0, 0);
               }
             });
  const Sink sink(pr, split);
#pragma unroll
  for (int ca = 0; ca < 4; ++ca)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      sink.four(m0 + 64 * wm + 16 * g + 4 * t + ca, n0 + 64 * wn + 4 * r,
                f32x4{acc[ca][0][t], acc[ca][1][t], acc[ca][2][t], acc[ca][3][t]});
  if (want_colsum) {
    // lane (r, g) holds the sums over its contraction rows (k = 4 g + step mod 16) of columns 4 r .. 4 r + 3 of the A tile
    float* dst = pr.colsum_out != nullptr ? pr.colsum_out : pr.colsum + (size_t)(pr.slab_base + split) * pr.M;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = cs[e];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int m = m0 + 64 * wm + 4 * r + e;
      if (g == 0 && m < pr.M) dst[m] = v;
    }
  }
}

__host__ __device__ inline int tile_cols(int form) { return (form == 2 || form == 4) ? 128 : 64; }
constexpr int kTileRows = 128;

__global__ __launch_bounds__(256, 2) void grouped_gemm_dma_kernel(GProbs g_arg, int items, int tune) {
  const GProbs& g = kernarg_table<GProbs>();
  extern __shared__ __attribute__((aligned(16))) float smem_gg[];
  const int bid = xcd_remap(blockIdx.x, items);
  int p = 0;
  while (p + 1 < g.n && g.first[p + 1] <= bid) ++p;
  p = __builtin_amdgcn_readfirstlane(p);
  const VqaGemmProblem& pr = g.p[p];
  const int local = bid - g.first[p];
  const int bn = tile_cols(pr.form);
  const int tiles_n = (pr.N + bn - 1) / bn, tiles_m = (pr.M + kTileRows - 1) / kTileRows;
  const int split = local / (tiles_m * tiles_n), t = local % (tiles_m * tiles_n);
  const int m0 = (t / tiles_n) * kTileRows, n0 = (t % tiles_n) * bn;
  if (pr.form == 0) tile_nt(pr, m0, n0, split, smem_gg, tune);
  else if (pr.form == 1 || pr.form == 3) tile_nn(pr, m0, n0, split, smem_gg, tune);
  else tile_tn(pr, m0, n0, split, smem_gg, tune);
}
}  // namespace gg

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

extern "C" int vqa_grouped_tile(int form, int* rows, int* cols, int* depth) {
  VQA_REQUIRE(form >= 0 && form <= 4 && rows && cols && depth, VQA_E_BADARG, "grouped_tile: form 0..4, non-null outputs");
  *rows = gg::kTileRows;
  *cols = gg::tile_cols(form);
  *depth = gg::kBK;
  return VQA_OK;
}

extern "C" int vqa_grouped_gemm(const VqaGemmProblem* problems, int n, vqa_stream_t stream) {
  VQA_REQUIRE(problems != nullptr && n >= 1 && n <= kGMaxGemms, VQA_E_BADARG, "grouped_gemm: 1..%d problems (got %d)", kGMaxGemms, n);
  GProbs g{};
  g.n = n;
  int items = 0;
  for (int i = 0; i < n; ++i) {
    VqaGemmProblem p = problems[i];
    VQA_REQUIRE(p.A && p.B && (p.slab || p.out), VQA_E_BADARG, "grouped_gemm[%d]: null pointer", i);
    VQA_REQUIRE(p.M > 0 && p.N > 0 && p.K > 0 && p.form >= 0 && p.form <= 4, VQA_E_BADARG,
                "grouped_gemm[%d]: bad sizes M=%d N=%d K=%d form=%d", i, p.M, p.N, p.K, p.form);
    VQA_REQUIRE(p.ksplit > 0 && p.ksplit % gg::kBK == 0 && p.slab_base >= 0 && (p.out != nullptr || p.slab_stride >= (long long)p.M * p.N),
                VQA_E_BADARG, "grouped_gemm[%d]: ksplit %d must be a positive multiple of %d, slab_stride >= M*N", i, p.ksplit, gg::kBK);
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
    // the DMA takes any 4-byte aligned source (tools/dma_probe.hip): no condition on pitches or extents beyond the
    // descriptor's 32-bit byte offsets
    const bool a_kc = p.form != 2 && p.form != 4, b_kc = p.form == 0;
    const long long a_bytes = ((long long)((a_kc ? p.M : p.Ka) - 1) * p.lda + (a_kc ? p.Ka : p.Ma)) * 4;
    const long long b_bytes = ((long long)((b_kc ? p.N : p.Kb) - 1) * p.ldb + (b_kc ? p.Kb : p.Nb)) * 4;
    VQA_REQUIRE(p.Ma >= p.M && p.Nb >= p.N, VQA_E_BADARG, "grouped_gemm[%d]: Ma / Nb must cover the output extent", i);
    VQA_REQUIRE(aligned(p.A, 4) && aligned(p.B, 4) && p.lda >= (a_kc ? p.Ka : p.Ma) && p.ldb >= (b_kc ? p.Kb : p.Nb) &&
                    a_bytes < (1LL << 30) && b_bytes < (1LL << 30),
                VQA_E_UNSUPPORTED, "grouped_gemm[%d]: operand pitch below its extent, or an operand of 1 GiB or more (form %d lda=%d ldb=%d)",
                i, p.form, p.lda, p.ldb);
    VQA_REQUIRE((p.colsum == nullptr && p.colsum_out == nullptr) || p.form == 2 || p.form == 4, VQA_E_BADARG,
                "grouped_gemm[%d]: column sums exist for the TN forms only", i);
    g.p[i] = p;
    g.first[i] = items;
    const long tiles = (long)((p.M + gg::kTileRows - 1) / gg::kTileRows) * ((p.N + gg::tile_cols(p.form) - 1) / gg::tile_cols(p.form)) * split_of(p);
    VQA_REQUIRE(items + tiles < (1L << 24), VQA_E_UNSUPPORTED, "grouped_gemm: too many tiles");
    items += (int)tiles;
  }
  g.first[n] = items;
  VQA_ENSURE_LDS(gg::grouped_gemm_dma_kernel, gg::kLdsBytes);
  const char* tune = vqa::option("VQA_GG_TUNE");      // (ablations of the measurement: see stage_loop)
  VQA_LAUNCH(gg::grouped_gemm_dma_kernel, dim3(items), dim3(256), gg::kLdsBytes, static_cast<hipStream_t>(stream), g, items,
             tune != nullptr ? std::atoi(tune) : 0);
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
