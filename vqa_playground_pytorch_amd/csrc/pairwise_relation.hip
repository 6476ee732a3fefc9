// K1 -- pairwise relation build + alpha-weighted reduce (CoR2 step 2).
//
// Replaces config/CoR2.py:191-199 (decare_cat) + :216 of the reference, which materialise five
// [B,N,N,D] fp32 tensors (5.4 GB each at B=512).  Here the [N,N,D] relation tensor of a sample never
// exists: a workgroup stages the [N x 4*NT] region tile of one sample in LDS with coalesced 16-byte
// loads (each lane owns one float4 column and its N LDS slots, so the tile needs no barrier and its
// ds_read_b128 / ds_write_b128 are conflict-free: consecutive lanes, consecutive 16-byte slots) and
// produces the N output rows straight from it.
//
// HBM-bound.  Algorithmic bytes per sample (fp32): forward (2*N*D + 2*D + N)*4 = 606 352 B at
// N=36, D=2048; backward (v, g read once, dq1/dq2/dalpha written) ~ 622 880 B.
//
// Every kernel is a template on the storage type T of the region tensors (v, v2, g, d_v): float, or bf16 for the
// mixed-precision path (BASELINE configs[4]; half the bytes, all arithmetic still fp32 in registers).  q1, q2,
// alpha and their gradients are fp32 in both.
#include <cstdlib>

#include "common.hpp"

namespace vqa {

template <typename T, int NT>
__global__ __launch_bounds__(NT) void pairwise_fwd_kernel(const T* __restrict__ v, const float* __restrict__ q1,
                                                          const float* __restrict__ q2, const float* __restrict__ alpha,
                                                          int astride, T* __restrict__ v2, int N, int D, int mode) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float4* tile = reinterpret_cast<float4*>(smem);                  // [N][NT]
  float* alpha_s = reinterpret_cast<float*>(tile + (size_t)N * NT);  // [N]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + tid) * 4;
  const bool active = d < D;
  for (int i = tid; i < N; i += NT) alpha_s[i] = alpha[((size_t)b * N + i) * astride];
  const size_t base = (size_t)b * N * D + d;
  float4 q1v = make_float4(0.f, 0.f, 0.f, 0.f), q2v = q1v;
  if (active) {
    q1v = ld4(q1 + (size_t)b * D + d);
    q2v = ld4(q2 + (size_t)b * D + d);
#pragma unroll 12
    for (int i = 0; i < N; ++i) tile[i * NT + tid] = ld4(v + base + (size_t)i * D);
  }
  __syncthreads();  // alpha_s; the tile column is private to this lane
  if (!active) return;
  if (mode == 1) {
    // factored: v2_j = q1 * (sum_i a_i v_i) + (sum_i a_i) * q2 * v_j
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float asum = 0.f;
#pragma unroll 4
    for (int i = 0; i < N; ++i) {
      const float a = alpha_s[i];
      s = fma4(a, tile[i * NT + tid], s);
      asum += a;
    }
    const float4 s1 = mul4(q1v, s);
    const float4 c2 = scale4(asum, q2v);
#pragma unroll 4
    for (int j = 0; j < N; ++j) {
      const float4 t = tile[j * NT + tid];
      st4(v2 + base + (size_t)j * D,
          make_float4(fmaf(c2.x, t.x, s1.x), fmaf(c2.y, t.y, s1.y), fmaf(c2.z, t.z, s1.z), fmaf(c2.w, t.w, s1.w)));
    }
  } else {
    // pairwise: every (i, j) term of the relation tensor is formed and weighted, as the reference sums it
    for (int j = 0; j < N; ++j) {
      const float4 t = mul4(tile[j * NT + tid], q2v);
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int i = 0; i < N; ++i) {
        const float4 vi = tile[i * NT + tid];
        const float a = alpha_s[i];
        acc.x = fmaf(a, fmaf(vi.x, q1v.x, t.x), acc.x);
        acc.y = fmaf(a, fmaf(vi.y, q1v.y, t.y), acc.y);
        acc.z = fmaf(a, fmaf(vi.z, q1v.z, t.z), acc.z);
        acc.w = fmaf(a, fmaf(vi.w, q1v.w, t.w), acc.w);
      }
      st4(v2 + base + (size_t)j * D, acc);
    }
  }
}

// ---- register-tile variants (N <= kRegN, the reference's 36 regions) ------------------------------------------
// One wave per workgroup, lane = one float4 column, the N region rows of that column held in VGPRs (144 for N=36):
// no LDS, so occupancy is set by registers (3 waves per SIMD, 12 per CU, against 4 per CU for the 73 KB LDS tile)
// and every lane has N independent 16-byte loads in flight.  Loads are unconditional from clamped rows (a load
// under a branch would serialise on vmcnt(0), see gemm_f32_mfma.hpp); rows >= N are zeroed by a select.
constexpr int kRegN = 36;

template <typename T>
__global__ __launch_bounds__(64) void pairwise_fwd_reg_kernel(const T* __restrict__ v, const float* __restrict__ q1,
                                                              const float* __restrict__ q2, const float* __restrict__ alpha,
                                                              int astride, T* __restrict__ v2, int N, int D) {
  const int b = blockIdx.y;
  const int d = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  const size_t base = (size_t)b * N * D + d;
  float4 r[kRegN];
#pragma unroll
  for (int i = 0; i < kRegN; ++i) r[i] = ld4(v + base + (size_t)min(i, N - 1) * D);
  const float4 q1v = ld4(q1 + (size_t)b * D + d), q2v = ld4(q2 + (size_t)b * D + d);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float asum = 0.f;
#pragma unroll
  for (int i = 0; i < kRegN; ++i) {
    const float a = i < N ? alpha[((size_t)b * N + i) * astride] : 0.f;  // wave-uniform address: scalar load
    s = fma4(a, r[i], s);
    asum += a;
  }
  const float4 s1 = mul4(q1v, s);
  const float4 c2 = scale4(asum, q2v);
#pragma unroll
  for (int j = 0; j < kRegN; ++j) {
    if (j < N)
      st4(v2 + base + (size_t)j * D, make_float4(fmaf(c2.x, r[j].x, s1.x), fmaf(c2.y, r[j].y, s1.y),
                                                 fmaf(c2.z, r[j].z, s1.z), fmaf(c2.w, r[j].w, s1.w)));
  }
}

// Pairwise form from registers (mode 0, N <= kRegN): the kernel north_star describes -- every (i, j) term of the relation
// tensor v_i*q1 + v_j*q2 is formed and weighted by alpha_i, as the reference sums it (config/CoR2.py:191-199, :216) -- with
// the region rows of a lane's float4 column held in VGPRs instead of an LDS tile.  The LDS version issued N*N
// ds_read_b128 per lane (24 TB/s of LDS reads chip-wide at B = 512: 225 us, 0.17 of the HBM roofline); from registers
// the inner sum is 2*N*N packed FMAs per lane (v_pk_fma_f32: two components per instruction) -- ~20 us of VALU at the
// chip's rate, under the ~50 us the 302 MB of v and v2 take at HBM speed.  alpha is wave-uniform (one sample per
// workgroup row): scalar loads, broadcast operands.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T>
__global__ __launch_bounds__(64, 3) void pairwise_fwd_pairs_reg_kernel(const T* __restrict__ v, const float* __restrict__ q1,
                                                                    const float* __restrict__ q2,
                                                                    const float* __restrict__ alpha, int astride,
                                                                    T* __restrict__ v2, int N, int D) {
  const int b = blockIdx.y;
  const int d = (blockIdx.x * 64 + threadIdx.x) * 4;
  if (d >= D) return;
  const size_t base = (size_t)b * N * D + d;
  f32x2 lo[kRegN], hi[kRegN];   // components (x, y) and (z, w) of the N rows of this lane's column
#pragma unroll
  for (int i = 0; i < kRegN; ++i) {
    const float4 t = ld4(v + base + (size_t)min(i, N - 1) * D);
    lo[i] = f32x2{t.x, t.y};
    hi[i] = f32x2{t.z, t.w};
  }
  const float4 q1v = ld4(q1 + (size_t)b * D + d), q2v = ld4(q2 + (size_t)b * D + d);
  const f32x2 q1l = {q1v.x, q1v.y}, q1h = {q1v.z, q1v.w}, q2l = {q2v.x, q2v.y}, q2h = {q2v.z, q2v.w};
  float a[kRegN];
#pragma unroll
  for (int i = 0; i < kRegN; ++i) a[i] = i < N ? alpha[((size_t)b * N + i) * astride] : 0.f;   // rows >= N weigh nothing
  // The j loop is not unrolled (36 x 36 x 4 FMAs of straight-line code would be 40 KB), so row j cannot be picked from
  // the register file by a constant index: it is loaded again (an L1 / L2 hit -- this lane fetched it microseconds ago).
  float4 n1 = ld4(v + base), n2 = ld4(v + base + (size_t)min(1, N - 1) * D);   // two ahead: see pairwise_fwd_pairs_reg2_kernel
  for (int j = 0; j < N; ++j) {
    const float4 vj = n1;
    n1 = n2;
    n2 = ld4(v + base + (size_t)min(j + 2, N - 1) * D);
    const f32x2 tl = f32x2{vj.x, vj.y} * q2l, th = f32x2{vj.z, vj.w} * q2h;
    f32x2 accl = {0.f, 0.f}, acch = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kRegN; ++i) {
      const f32x2 ai = {a[i], a[i]};
      accl = __builtin_elementwise_fma(ai, __builtin_elementwise_fma(lo[i], q1l, tl), accl);
      acch = __builtin_elementwise_fma(ai, __builtin_elementwise_fma(hi[i], q1h, th), acch);
    }
    st4(v2 + base + (size_t)j * D, make_float4(accl.x, accl.y, acch.x, acch.y));
  }
}

// The same with a float2 column per lane (fp32 storage): 72 row registers instead of 144, so 5-6 waves share a SIMD instead
// of 3 and twice as many, half-sized waves pass through it.  A wave loads its rows, runs its N x N sum, stores -- with few
// large waves the whole chip is in the same phase at the same time and the 302 MB of traffic do not overlap with the
// 2 N N packed FMAs per lane (99 us); with many small ones the phases interleave.
__global__ __launch_bounds__(64, 5) void pairwise_fwd_pairs_reg2_kernel(const float* __restrict__ v, const float* __restrict__ q1,
                                                                     const float* __restrict__ q2,
                                                                     const float* __restrict__ alpha, int astride,
                                                                     float* __restrict__ v2, int N, int D, DropCfg dc) {
  const int b = blockIdx.y;
  const int d = (blockIdx.x * 64 + threadIdx.x) * 2;
  if (d >= D) return;
  const size_t base = (size_t)b * N * D + d;
  const uint32_t key = dc.p8 > 0 ? drop_key(dc) : 0u;      // (dc.p8 > 0: the consumer's input dropout in the store, as relation_apply)
  f32x2 r[kRegN];
#pragma unroll
  for (int i = 0; i < kRegN; ++i) r[i] = *reinterpret_cast<const f32x2*>(v + base + (size_t)min(i, N - 1) * D);
  const f32x2 q1v = *reinterpret_cast<const f32x2*>(q1 + (size_t)b * D + d);
  const f32x2 q2v = *reinterpret_cast<const f32x2*>(q2 + (size_t)b * D + d);
  float a[kRegN];
#pragma unroll
  for (int i = 0; i < kRegN; ++i) a[i] = i < N ? alpha[((size_t)b * N + i) * astride] : 0.f;
  // Row j again (an L1 / L2 hit), TWO iterations ahead of its use: vmcnt counts loads and stores in issue order, so a
  // reload issued right behind the store of row j - 1 could only be waited for with vmcnt(0) -- i.e. together with that
  // store's whole HBM round trip, once per row (88 us).  Two ahead, the wait leaves the younger store in flight.
  f32x2 n1 = *reinterpret_cast<const f32x2*>(v + base);
  f32x2 n2 = *reinterpret_cast<const f32x2*>(v + base + (size_t)min(1, N - 1) * D);
  for (int j = 0; j < N; ++j) {
    const f32x2 tj = n1 * q2v;
    n1 = n2;
    n2 = *reinterpret_cast<const f32x2*>(v + base + (size_t)min(j + 2, N - 1) * D);
    f32x2 acc = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kRegN; ++i) {
      const f32x2 ai = {a[i], a[i]};
      acc = __builtin_elementwise_fma(ai, __builtin_elementwise_fma(r[i], q1v, tj), acc);
    }
    if (dc.p8 > 0) {
      const uint32_t e = (uint32_t)(base + (size_t)j * D);   // even
      if (dc.p8 == kDropHalf) {
        const uint32_t w = mask_word32(e >> 5, key) >> (e & 31u);
        acc *= f32x2{(w & 1u) ? 2.f : 0.f, (w & 2u) ? 2.f : 0.f};
      } else {
        const uint32_t w = mask_word32(e >> 2, key) >> (8 * (e & 3));
        acc *= f32x2{(w & 255u) >= dc.p8 ? dc.scale : 0.f, ((w >> 8) & 255u) >= dc.p8 ? dc.scale : 0.f};
      }
    }
    *reinterpret_cast<f32x2*>(v2 + base + (size_t)j * D) = acc;
  }
}

// Forward, factored form for N > kRegN (e.g. the dense 100-region configuration): two streaming passes at full
// occupancy instead of a 100 KB LDS tile per 64 lanes.  Pass 1 accumulates s = sum_i alpha_i v_i per column; pass 2
// re-reads the rows this workgroup has just streamed (L2 / Infinity Cache) and writes v2_j = q1*s + asum*q2*v_j.
template <typename T, int NT>
__global__ __launch_bounds__(NT) void pairwise_fwd_stream_kernel(const T* __restrict__ v, const float* __restrict__ q1,
                                                                 const float* __restrict__ q2,
                                                                 const float* __restrict__ alpha, int astride,
                                                                 T* __restrict__ v2, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);  // [N]
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + tid) * 4;
  for (int i = tid; i < N; i += NT) alpha_s[i] = alpha[((size_t)b * N + i) * astride];
  __syncthreads();
  if (d >= D) return;
  const size_t base = (size_t)b * N * D + d;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  float asum = 0.f;
#pragma unroll 8
  for (int i = 0; i < N; ++i) {
    const float a = alpha_s[i];
    s = fma4(a, ld4(v + base + (size_t)i * D), s);
    asum += a;
  }
  const float4 s1 = mul4(ld4(q1 + (size_t)b * D + d), s);
  const float4 c2 = scale4(asum, ld4(q2 + (size_t)b * D + d));
#pragma unroll 8
  for (int j = 0; j < N; ++j) {
    const float4 t = ld4(v + base + (size_t)j * D);
    st4(v2 + base + (size_t)j * D,
        make_float4(fmaf(c2.x, t.x, s1.x), fmaf(c2.y, t.y, s1.y), fmaf(c2.z, t.z, s1.z), fmaf(c2.w, t.w, s1.w)));
  }
}

// Backward.  g = dL/dv2.
//   gsum = sum_j g_j ; gv = sum_j g_j*v_j ; pooled = sum_i a_i v_i ; asum = sum_i a_i
//   dq1 = pooled*gsum ; dq2 = asum*gv ; dalpha_i = <v_i, q1*gsum> + <gv, q2> ; dv_i = a_i*q1*gsum + asum*q2*g_i
// Two-pass streaming form (any N): pass 1 streams (v_j, g_j) once and keeps only the three column
// accumulators (sum g, sum g*v, sum alpha*v), so the kernel runs at full occupancy; pass 2 re-reads the v rows this
// workgroup has just streamed (36 KB per wave, served by L2 / Infinity Cache, not HBM) for dalpha_i = <v_i, q1*sum g>.
// dalpha partials: wave64 shuffles -> LDS -> one float atomic per (workgroup, region).
// TWO: the gradient arrives in two tensors (v2 has two consumers -- the second-step compress and the second-step pooling
// -- and is returned to autograd as two aliases, so the two gradients are added here in registers instead of by a
// 3 x B*N*D-element add kernel in front of this one).
// RS (row split, as in attention_pool.hip): RS = NT/64 makes the waves of a workgroup share NT*4/RS columns and take
// every RS-th row each -- four times the lanes for a small batch; the three column accumulators then meet in LDS.
template <typename T, int NT, bool TWO, int RS>
__global__ __launch_bounds__(NT) void pairwise_bwd_stream_kernel(const T* __restrict__ v, const float* __restrict__ q1,
                                                                 const float* __restrict__ q2,
                                                                 const float* __restrict__ alpha, int astride,
                                                                 const T* __restrict__ g, const T* __restrict__ g_b,
                                                                 float* __restrict__ d_alpha,
                                                                 float* __restrict__ d_q1, float* __restrict__ d_q2,
                                                                 T* __restrict__ d_v, int N, int D) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* alpha_s = reinterpret_cast<float*>(smem);  // [N]
  float* red_s = alpha_s + N;                       // [N]
  const int tid = threadIdx.x, lane = tid & 63;
  const int b = blockIdx.y;
  constexpr int COLS = NT / RS;  // float4 columns per workgroup
  const int rs = tid / COLS, ct = tid % COLS;
  const int d = (blockIdx.x * COLS + ct) * 4;
  const bool active = d < D;
  const int dc = active ? d : 0;
  for (int i = tid; i < N; i += NT) {
    alpha_s[i] = alpha[((size_t)b * N + i) * astride];
    red_s[i] = 0.f;
  }
  __syncthreads();
  const size_t base = (size_t)b * N * D + dc;
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 gsum = z, gv = z, pooled = z;
  float asum = 0.f;
#pragma unroll 6
  for (int j = rs; j < N; j += RS) {
    const float4 vj = ld4(v + base + (size_t)j * D);
    float4 gj = ld4(g + base + (size_t)j * D);
    if constexpr (TWO) gj = add4(gj, ld4(g_b + base + (size_t)j * D));
    gsum = add4(gsum, gj);
    gv = add4(gv, mul4(gj, vj));
    pooled = fma4(alpha_s[j], vj, pooled);
  }
  for (int j = 0; j < N; ++j) asum += alpha_s[j];
  if constexpr (RS > 1) {  // column accumulators of the RS row slices meet in LDS (after alpha_s / red_s)
    float4* comb = reinterpret_cast<float4*>(smem + (((size_t)N * 8 + 15) / 16) * 16);  // [RS][3][COLS]
    comb[(rs * 3 + 0) * COLS + ct] = gsum;
    comb[(rs * 3 + 1) * COLS + ct] = gv;
    comb[(rs * 3 + 2) * COLS + ct] = pooled;
    __syncthreads();
    gsum = gv = pooled = z;
#pragma unroll
    for (int q = 0; q < RS; ++q) {
      gsum = add4(gsum, comb[(q * 3 + 0) * COLS + ct]);
      gv = add4(gv, comb[(q * 3 + 1) * COLS + ct]);
      pooled = add4(pooled, comb[(q * 3 + 2) * COLS + ct]);
    }
  }
  const float4 q1v = ld4(q1 + (size_t)b * D + dc), q2v = ld4(q2 + (size_t)b * D + dc);
  const float4 u = active ? mul4(q1v, gsum) : z;
  if (active && rs == 0) {
    st4(d_q1 + (size_t)b * D + d, mul4(pooled, gsum));
    st4(d_q2 + (size_t)b * D + d, scale4(asum, gv));
  }
  const float cpart = active ? dot4(gv, q2v) : 0.f;
  const float4 c2 = scale4(asum, q2v);
  constexpr int RB = 6;  // rows per batch: RB independent 16-byte loads in flight per lane, then RB wave reductions
  for (int i0 = rs; i0 < N; i0 += RB * RS) {
    float4 vi[RB];
#pragma unroll
    for (int k = 0; k < RB; ++k) vi[k] = ld4(v + base + (size_t)min(i0 + k * RS, N - 1) * D);
#pragma unroll
    for (int k = 0; k < RB; ++k) {
      const int i = i0 + k * RS;
      if (i < N) {
        if (d_v != nullptr && active) {
          float4 gi = ld4(g + base + (size_t)i * D);
          if constexpr (TWO) gi = add4(gi, ld4(g_b + base + (size_t)i * D));
          st4(d_v + base + (size_t)i * D, add4(scale4(alpha_s[i], u), mul4(c2, gi)));
        }
        const float p = wave_sum(dot4(vi[k], u) + cpart);
        if (lane == 0) atomicAdd(&red_s[i], p);
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < N; i += NT) atomicAdd(&d_alpha[(size_t)b * N + i], red_s[i]);
}

// ---- closed form with the pooled feature given ("relation apply") ------------------------------------------------------
// With s = sum_i alpha_i v_i already available -- it IS glimpse 0 of the first attention's pooled output, and a softmax
// alpha sums to 1 -- the whole relation step is a per-sample affine map of the regions,
//     v2[b,n,:] = t[b,:] + c2[b,:] * v[b,n,:],     t = q1 * s,  c2 = (sum_i alpha_i) * q2,
// and the only consumer that needs it materialised is the second compress layer, which wants it dropped out
// (config/CoR2.py:72-75 at :218): out = keep * v2 in ONE pass over v (keep = the counter-hash dropout of common.hpp keyed
// by the element index (b*N+n)*D+d, i.e. the mask vqa_linear_dropout_mask(B*N, D, ...) exports).  The second attention
// pools v itself (pooled2 = t + c2 * sum_n alpha2_n v_n), so v2 is never read back and its gradient arrives in one
// tensor.  Backward: d_t = sum_n keep*g, d_c2 = sum_n keep*g*v, optional d_v = c2*keep*g -- one pass over (v, g), no
// second pass, no atomics.
__device__ __forceinline__ float4 keep4(uint32_t e, const DropCfg& dc) {
  if (dc.p8 == 0) return make_float4(1.f, 1.f, 1.f, 1.f);
  const float2 a = drop_pair(e, dc), b = drop_pair(e + 2, dc);
  return make_float4(a.x, a.y, b.x, b.y);
}

template <typename T, int NT>
__global__ __launch_bounds__(NT) void relation_apply_fwd_kernel(const T* __restrict__ v, const float* __restrict__ t,
                                                                const float* __restrict__ c2, T* __restrict__ out, int N,
                                                                int D, int rows_per_block, DropCfg dc) {
  const int b = blockIdx.y;
  const int d = (blockIdx.x * NT + threadIdx.x) * 4;
  if (d >= D) return;
  const float4 tv = ld4(t + (size_t)b * D + d), cv = ld4(c2 + (size_t)b * D + d);
  const int n_lo = blockIdx.z * rows_per_block, n_hi = min(N, n_lo + rows_per_block);
  const size_t base = (size_t)b * N * D + d;
#pragma unroll 6
  for (int n = n_lo; n < n_hi; ++n) {
    const float4 x = ld4(v + base + (size_t)n * D);
    const float4 k = keep4((uint32_t)(base + (size_t)n * D), dc);
    st4(out + base + (size_t)n * D,
        make_float4(k.x * fmaf(cv.x, x.x, tv.x), k.y * fmaf(cv.y, x.y, tv.y), k.z * fmaf(cv.z, x.z, tv.z),
                    k.w * fmaf(cv.w, x.w, tv.w)));
  }
}

template <typename T, int NT, int RS>
__global__ __launch_bounds__(NT) void relation_apply_bwd_kernel(const T* __restrict__ v, const float* __restrict__ c2,
                                                                const T* __restrict__ g, float* __restrict__ d_t,
                                                                float* __restrict__ d_c2, T* __restrict__ d_v, int N, int D,
                                                                DropCfg dc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, b = blockIdx.y;
  constexpr int COLS = NT / RS;
  const int rs = tid / COLS, ct = tid % COLS;
  const int d = (blockIdx.x * COLS + ct) * 4;
  const bool active = d < D;
  const int dc_ = active ? d : 0;
  const size_t base = (size_t)b * N * D + dc_;
  const float4 cv = ld4(c2 + (size_t)b * D + dc_);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 st = z, sc = z;
#pragma unroll 6
  for (int n = rs; n < N; n += RS) {
    const float4 x = ld4(v + base + (size_t)n * D);
    const float4 gk = mul4(ld4(g + base + (size_t)n * D), keep4((uint32_t)(base + (size_t)n * D), dc));
    st = add4(st, gk);
    sc = add4(sc, mul4(gk, x));
    if (d_v != nullptr && active) st4(d_v + base + (size_t)n * D, mul4(cv, gk));
  }
  if constexpr (RS > 1) {
    float4* comb = reinterpret_cast<float4*>(smem);  // [RS][2][COLS]
    comb[(rs * 2 + 0) * COLS + ct] = st;
    comb[(rs * 2 + 1) * COLS + ct] = sc;
    __syncthreads();
    st = sc = z;
#pragma unroll
    for (int q = 0; q < RS; ++q) {
      st = add4(st, comb[(q * 2 + 0) * COLS + ct]);
      sc = add4(sc, comb[(q * 2 + 1) * COLS + ct]);
    }
  }
  if (active && rs == 0) {
    st4(d_t + (size_t)b * D + d, st);
    st4(d_c2 + (size_t)b * D + d, sc);
  }
}

// bf16 storage, 16-byte accesses (8 columns per lane), no d_v: 256 lanes = 256 / RS column groups x RS row slices.  The 8-byte
// form above moves the 104 MB of (v, g) at B = 128, N = 100 in 34 us (3.1 TB/s): too few bytes in flight per lane.
template <int RS>
__global__ __launch_bounds__(256) void relation_apply_bwd8_bf16_kernel(const bf16* __restrict__ v, const bf16* __restrict__ g,
                                                                       float* __restrict__ d_t, float* __restrict__ d_c2,
                                                                       int N, int D, DropCfg dc) {
  constexpr int COLS = 256 / RS;
  __shared__ float comb[RS][2][COLS][8];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int rs = tid / COLS, ct = tid % COLS;
  const int d = (blockIdx.x * COLS + ct) * 8;
  const bool active = d < D;
  const int dcl = active ? d : 0;
  const size_t base = (size_t)b * N * D + dcl;
  float st[8], sc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) st[j] = sc[j] = 0.f;
  const uint32_t key = drop_key(dc);
  const auto unpack = [](uint4 w, float (&x)[8]) {
    x[0] = bf16_lo(w.x), x[1] = bf16_hi(w.x), x[2] = bf16_lo(w.y), x[3] = bf16_hi(w.y);
    x[4] = bf16_lo(w.z), x[5] = bf16_hi(w.z), x[6] = bf16_lo(w.w), x[7] = bf16_hi(w.w);
  };
#pragma unroll 4
  for (int n = rs; n < N; n += RS) {
    const size_t e = base + (size_t)n * D;
    float x[8], gk[8];
    unpack(*reinterpret_cast<const uint4*>(v + e), x);
    unpack(*reinterpret_cast<const uint4*>(g + e), gk);
    if (dc.p8 == kDropHalf) {          // one bit per element: the 8 elements (e % 8 == 0) sit in one hash word
      const uint32_t bits = mask_word32((uint32_t)(e >> 5), key) >> ((uint32_t)e & 31u);
#pragma unroll
      for (int j = 0; j < 8; ++j) gk[j] = ((bits >> j) & 1u) != 0u ? 2.f * gk[j] : 0.f;
    } else if (dc.p8 != 0) {
      const float4 k0 = keep4((uint32_t)e, dc), k1 = keep4((uint32_t)e + 4u, dc);
      gk[0] *= k0.x, gk[1] *= k0.y, gk[2] *= k0.z, gk[3] *= k0.w, gk[4] *= k1.x, gk[5] *= k1.y, gk[6] *= k1.z, gk[7] *= k1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      st[j] += gk[j];
      sc[j] = fmaf(gk[j], x[j], sc[j]);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    comb[rs][0][ct][j] = st[j];
    comb[rs][1][ct][j] = sc[j];
  }
  __syncthreads();
  if (rs == 0 && active) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.f, c = 0.f;
#pragma unroll
      for (int q = 0; q < RS; ++q) {     // fixed order
        a += comb[q][0][ct][j];
        c += comb[q][1][ct][j];
      }
      d_t[(size_t)b * D + d + j] = a;
      d_c2[(size_t)b * D + d + j] = c;
    }
  }
}

static int pick_threads(int N) { return N <= 36 ? 128 : 64; }

template <typename T>
static int pairwise_fwd_impl(const char* who, const T* v, const float* q1, const float* q2, const float* alpha,
                             int alpha_stride, T* v2, int B, int N, int D, int mode, vqa_stream_t stream) {
  constexpr size_t kAlign = 4 * sizeof(T);  // one lane access = 4 elements
  VQA_REQUIRE(v && q1 && q2 && alpha && v2, VQA_E_BADARG, "%s: null pointer", who);
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && alpha_stride > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d D=%d alpha_stride=%d", who,
              B, N, D, alpha_stride);
  VQA_REQUIRE(mode == 0 || mode == 1, VQA_E_BADARG, "%s: mode must be 0 or 1, got %d", who, mode);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, kAlign) && aligned(q1, 16) && aligned(q2, 16) && aligned(v2, kAlign),
              VQA_E_UNSUPPORTED, "%s: needs D %% 4 == 0, 16-byte aligned q1/q2 and %zu-byte aligned v/v2 (D=%d)", who, kAlign,
              D);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "%s: B=%d exceeds 65535", who, B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (mode == 1 && N <= kRegN) {
    VQA_LAUNCH(pairwise_fwd_reg_kernel<T>, dim3((D / 4 + 63) / 64, B), dim3(64), 0, s, v, q1, q2, alpha, alpha_stride,
                       v2, N, D);
    return check_launch(who);
  }
  if (mode == 1) {
    VQA_REQUIRE(N <= 4096, VQA_E_UNSUPPORTED, "%s: N=%d exceeds 4096", who, N);
    constexpr int NT = 256;
    VQA_LAUNCH((pairwise_fwd_stream_kernel<T, NT>), dim3((D / 4 + NT - 1) / NT, B), dim3(NT), (size_t)N * 4, s, v, q1, q2,
                       alpha, alpha_stride, v2, N, D);
    return check_launch(who);
  }
  if (N <= kRegN && vqa::option("VQA_K1_PAIRWISE_LDS") == nullptr) {   // (the env knob keeps the LDS-tile form reachable)
    const int tiles_d = (D / 4 + 63) / 64;
    if constexpr (sizeof(T) == 4) {
      if ((long)B * D >= (1L << 19) && vqa::option("VQA_K1_PAIRWISE_REG4") == nullptr) {   // enough columns for half-sized waves
        VQA_LAUNCH(pairwise_fwd_pairs_reg2_kernel, dim3((D / 2 + 63) / 64, B), dim3(64), 0, s, v, q1, q2, alpha,
                           alpha_stride, v2, N, D, make_drop(0.f, 0));
        return check_launch(who);
      }
    }
    VQA_LAUNCH(pairwise_fwd_pairs_reg_kernel<T>, dim3(tiles_d, B), dim3(64), 0, s, v, q1, q2, alpha, alpha_stride, v2,
                       N, D);
    return check_launch(who);
  }
  VQA_REQUIRE(N <= 144, VQA_E_UNSUPPORTED, "%s: N=%d exceeds the LDS tile limit 144 of mode 0", who, N);
  const int nt = pick_threads(N);
  const size_t lds = (size_t)N * nt * 16 + (size_t)N * 4;
  dim3 grid((D / 4 + nt - 1) / nt, B);
  if (nt == 128) {
    VQA_ENSURE_LDS((pairwise_fwd_kernel<T, 128>), lds);
    VQA_LAUNCH((pairwise_fwd_kernel<T, 128>), grid, dim3(128), lds, s, v, q1, q2, alpha, alpha_stride, v2, N, D, mode);
  } else {
    VQA_ENSURE_LDS((pairwise_fwd_kernel<T, 64>), lds);
    VQA_LAUNCH((pairwise_fwd_kernel<T, 64>), grid, dim3(64), lds, s, v, q1, q2, alpha, alpha_stride, v2, N, D, mode);
  }
  return check_launch(who);
}

template <typename T>
static int pairwise_bwd_impl(const char* who, const T* v, const float* q1, const float* q2, const float* alpha,
                             int alpha_stride, const T* g_v2, const T* g_v2_b, float* d_alpha, float* d_q1, float* d_q2,
                             T* d_v, int B, int N, int D, vqa_stream_t stream) {
  constexpr size_t kAlign = 4 * sizeof(T);
  VQA_REQUIRE(v && q1 && q2 && alpha && g_v2 && d_alpha && d_q1 && d_q2, VQA_E_BADARG, "%s: null pointer", who);
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && alpha_stride > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d D=%d alpha_stride=%d", who,
              B, N, D, alpha_stride);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, kAlign) && aligned(q1, 16) && aligned(q2, 16) && aligned(g_v2, kAlign) &&
                  aligned(d_q1, 16) && aligned(d_q2, 16) && (d_v == nullptr || aligned(d_v, kAlign)) &&
                  (g_v2_b == nullptr || aligned(g_v2_b, kAlign)),
              VQA_E_UNSUPPORTED, "%s: needs D %% 4 == 0, 16-byte aligned fp32 and %zu-byte aligned region tensors (D=%d)", who,
              kAlign, D);
  VQA_REQUIRE(N <= 4096, VQA_E_UNSUPPORTED, "%s: N=%d exceeds 4096", who, N);
  VQA_REQUIRE(B <= 65535, VQA_E_UNSUPPORTED, "%s: B=%d exceeds 65535", who, B);
  hipStream_t s = static_cast<hipStream_t>(stream);
  int rc = zero_async(d_alpha, (size_t)B * N * sizeof(float), s);
  if (rc != VQA_OK) return rc;
  constexpr int NT = 256;
  if ((long)B * D / 4 < 4 * 65536) {  // fewer than 4 waves per CU worth of lanes: split the rows over the waves
    constexpr int RS = NT / 64;
    const size_t lds = (((size_t)N * 8 + 15) / 16) * 16 + (size_t)RS * 3 * (NT / RS) * sizeof(float4);
    const dim3 grid((D / 4 + 63) / 64, B);
    if (g_v2_b != nullptr) {
      VQA_LAUNCH((pairwise_bwd_stream_kernel<T, NT, true, RS>), grid, dim3(NT), lds, s, v, q1, q2, alpha, alpha_stride,
                         g_v2, g_v2_b, d_alpha, d_q1, d_q2, d_v, N, D);
    } else {
      VQA_LAUNCH((pairwise_bwd_stream_kernel<T, NT, false, RS>), grid, dim3(NT), lds, s, v, q1, q2, alpha, alpha_stride,
                         g_v2, g_v2_b, d_alpha, d_q1, d_q2, d_v, N, D);
    }
    return check_launch(who);
  }
  if (g_v2_b != nullptr) {
    VQA_LAUNCH((pairwise_bwd_stream_kernel<T, NT, true, 1>), dim3((D / 4 + NT - 1) / NT, B), dim3(NT), (size_t)N * 8, s,
                       v, q1, q2, alpha, alpha_stride, g_v2, g_v2_b, d_alpha, d_q1, d_q2, d_v, N, D);
  } else {
    VQA_LAUNCH((pairwise_bwd_stream_kernel<T, NT, false, 1>), dim3((D / 4 + NT - 1) / NT, B), dim3(NT), (size_t)N * 8, s,
                       v, q1, q2, alpha, alpha_stride, g_v2, g_v2_b, d_alpha, d_q1, d_q2, d_v, N, D);
  }
  return check_launch(who);
}

template <typename T>
static int relation_check(const char* who, const T* v, int B, int N, int D, float p) {
  constexpr size_t kAlign = 4 * sizeof(T);
  VQA_REQUIRE(B > 0 && N > 0 && D > 0, VQA_E_BADARG, "%s: bad sizes B=%d N=%d D=%d", who, B, N, D);
  VQA_REQUIRE(p >= 0.f && p < 1.f, VQA_E_BADARG, "%s: p_drop=%f outside [0,1)", who, (double)p);
  VQA_REQUIRE(D % 4 == 0 && aligned(v, kAlign), VQA_E_UNSUPPORTED, "%s: needs D %% 4 == 0 and %zu-byte aligned region tensors", who,
              kAlign);
  VQA_REQUIRE(B <= 65535 && (long)B * N * D < (1L << 32), VQA_E_UNSUPPORTED, "%s: B*N*D must stay below 2^32", who);
  return VQA_OK;
}

template <typename T>
static int relation_apply_fwd_impl(const char* who, const T* v, const float* t, const float* c2, T* out, float p_drop,
                                   uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D, vqa_stream_t stream) {
  VQA_REQUIRE(v && t && c2 && out, VQA_E_BADARG, "%s: null pointer", who);
  int rc = relation_check(who, v, B, N, D, p_drop);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(aligned(t, 16) && aligned(c2, 16) && aligned(out, 4 * sizeof(T)), VQA_E_UNSUPPORTED, "%s: unaligned pointer", who);
  constexpr int NT = 256;
  const int col_blocks = (D / 4 + NT - 1) / NT;
  int zsplit = (2048 + col_blocks * B - 1) / (col_blocks * B);  // >= 2048 workgroups when the batch alone gives fewer
  if (zsplit > (N + 3) / 4) zsplit = (N + 3) / 4;
  if (zsplit < 1) zsplit = 1;
  const int rows_per_block = (N + zsplit - 1) / zsplit;
  VQA_LAUNCH((relation_apply_fwd_kernel<T, NT>), dim3(col_blocks, B, (N + rows_per_block - 1) / rows_per_block), dim3(NT),
                     0, static_cast<hipStream_t>(stream), v, t, c2, out, N, D, rows_per_block, make_drop(p_drop, seed, seed_ptr));
  return check_launch(who);
}

template <typename T>
static int relation_apply_bwd_impl(const char* who, const T* v, const float* c2, const T* g, float* d_t, float* d_c2, T* d_v,
                                   float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D,
                                   vqa_stream_t stream) {
  VQA_REQUIRE(v && c2 && g && d_t && d_c2, VQA_E_BADARG, "%s: null pointer", who);
  int rc = relation_check(who, v, B, N, D, p_drop);
  if (rc != VQA_OK) return rc;
  VQA_REQUIRE(aligned(c2, 16) && aligned(d_t, 16) && aligned(d_c2, 16) && aligned(g, 4 * sizeof(T)) &&
                  (d_v == nullptr || aligned(d_v, 4 * sizeof(T))),
              VQA_E_UNSUPPORTED, "%s: unaligned pointer", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  constexpr int NT = 256;
  if constexpr (sizeof(T) == 2) {
    if (d_v == nullptr && D % 8 == 0 && aligned(v, 16) && aligned(g, 16) && (long)B * N * D < (1L << 32) &&
        (long)B * D / 4 < 4 * 65536) {
      constexpr int RS = 8;
      VQA_LAUNCH(relation_apply_bwd8_bf16_kernel<RS>, dim3((D / 8 + 256 / RS - 1) / (256 / RS), B), dim3(256), 0, s,
                         reinterpret_cast<const bf16*>(v), reinterpret_cast<const bf16*>(g), d_t, d_c2, N, D, dc);
      return check_launch(who);
    }
  }
  if ((long)B * D / 4 < 4 * 65536) {  // small batch: the 4 waves of a workgroup share 256 columns and split the rows
    constexpr int RS = NT / 64;
    VQA_LAUNCH((relation_apply_bwd_kernel<T, NT, RS>), dim3((D / 4 + 63) / 64, B), dim3(NT),
                       (size_t)RS * 2 * (NT / RS) * sizeof(float4), s, v, c2, g, d_t, d_c2, d_v, N, D, dc);
  } else {
    VQA_LAUNCH((relation_apply_bwd_kernel<T, NT, 1>), dim3((D / 4 + NT - 1) / NT, B), dim3(NT), 0, s, v, c2, g, d_t,
                       d_c2, d_v, N, D, dc);
  }
  return check_launch(who);
}

}  // namespace vqa

using namespace vqa;

extern "C" int vqa_pairwise_relation_reduce_fwd(const float* v, const float* q1, const float* q2, const float* alpha,
                                                int alpha_stride, float* v2, int B, int N, int D, int mode,
                                                vqa_stream_t stream) {
  return pairwise_fwd_impl<float>("pairwise_relation_reduce_fwd", v, q1, q2, alpha, alpha_stride, v2, B, N, D, mode, stream);
}

extern "C" int vqa_pairwise_relation_reduce_fwd_bf16(const vqa_bf16_t* v, const float* q1, const float* q2,
                                                     const float* alpha, int alpha_stride, vqa_bf16_t* v2, int B, int N,
                                                     int D, int mode, vqa_stream_t stream) {
  return pairwise_fwd_impl<bf16>("pairwise_relation_reduce_fwd_bf16", reinterpret_cast<const bf16*>(v), q1, q2, alpha,
                                 alpha_stride, reinterpret_cast<bf16*>(v2), B, N, D, mode, stream);
}

extern "C" int vqa_pairwise_relation_reduce_bwd(const float* v, const float* q1, const float* q2, const float* alpha,
                                                int alpha_stride, const float* g_v2, const float* g_v2_b, float* d_alpha,
                                                float* d_q1, float* d_q2, float* d_v, int B, int N, int D,
                                                vqa_stream_t stream) {
  return pairwise_bwd_impl<float>("pairwise_relation_reduce_bwd", v, q1, q2, alpha, alpha_stride, g_v2, g_v2_b, d_alpha, d_q1,
                                  d_q2, d_v, B, N, D, stream);
}

extern "C" int vqa_pairwise_relation_reduce_bwd_bf16(const vqa_bf16_t* v, const float* q1, const float* q2,
                                                     const float* alpha, int alpha_stride, const vqa_bf16_t* g_v2,
                                                     const vqa_bf16_t* g_v2_b, float* d_alpha, float* d_q1, float* d_q2,
                                                     vqa_bf16_t* d_v, int B, int N, int D, vqa_stream_t stream) {
  return pairwise_bwd_impl<bf16>("pairwise_relation_reduce_bwd_bf16", reinterpret_cast<const bf16*>(v), q1, q2, alpha,
                                 alpha_stride, reinterpret_cast<const bf16*>(g_v2), reinterpret_cast<const bf16*>(g_v2_b),
                                 d_alpha, d_q1, d_q2, reinterpret_cast<bf16*>(d_v), B, N, D, stream);
}

extern "C" int vqa_relation_apply_fwd(const float* v, const float* t, const float* c2, float* out, float p_drop, uint64_t seed,
                                      const uint64_t* seed_ptr, int B, int N, int D, vqa_stream_t stream) {
  return relation_apply_fwd_impl<float>("relation_apply_fwd", v, t, c2, out, p_drop, seed, seed_ptr, B, N, D, stream);
}
extern "C" int vqa_relation_apply_fwd_bf16(const vqa_bf16_t* v, const float* t, const float* c2, vqa_bf16_t* out, float p_drop,
                                           uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D, vqa_stream_t stream) {
  return relation_apply_fwd_impl<bf16>("relation_apply_fwd_bf16", reinterpret_cast<const bf16*>(v), t, c2,
                                       reinterpret_cast<bf16*>(out), p_drop, seed, seed_ptr, B, N, D, stream);
}
extern "C" int vqa_relation_apply_bwd(const float* v, const float* c2, const float* g, float* d_t, float* d_c2, float* d_v,
                                      float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D,
                                      vqa_stream_t stream) {
  return relation_apply_bwd_impl<float>("relation_apply_bwd", v, c2, g, d_t, d_c2, d_v, p_drop, seed, seed_ptr, B, N, D, stream);
}
extern "C" int vqa_relation_apply_bwd_bf16(const vqa_bf16_t* v, const float* c2, const vqa_bf16_t* g, float* d_t, float* d_c2,
                                           vqa_bf16_t* d_v, float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N,
                                           int D, vqa_stream_t stream) {
  return relation_apply_bwd_impl<bf16>("relation_apply_bwd_bf16", reinterpret_cast<const bf16*>(v), c2,
                                       reinterpret_cast<const bf16*>(g), d_t, d_c2, reinterpret_cast<bf16*>(d_v), p_drop, seed,
                                       seed_ptr, B, N, D, stream);
}

// The pairwise forward with the consumer's input dropout in its store (mode 0 of the fused relation + projection node:
// ops.relation_projection(..., pairwise=...)): v2[b,j,:] = keep * sum_i alpha_i (v_i q1 + v_j q2), keep over the element
// index (b N + j) D + d as vqa_relation_apply_fwd draws it.
extern "C" int vqa_pairwise_relation_reduce_drop_supported(int B, int N, int D) {
  return N >= 1 && N <= kRegN && D % 4 == 0 && (long)B * D >= (1L << 19) && B <= 65535 && (size_t)B * N * D < (1ull << 32);
}
extern "C" int vqa_pairwise_relation_reduce_drop_fwd(const float* v, const float* q1, const float* q2, const float* alpha,
                                                     int alpha_stride, float* v2, float p_drop, uint64_t seed,
                                                     const uint64_t* seed_ptr, int B, int N, int D, vqa_stream_t stream) {
  VQA_REQUIRE(v && q1 && q2 && alpha && v2, VQA_E_BADARG, "pairwise_relation_reduce_drop_fwd: null pointer");
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && alpha_stride > 0, VQA_E_BADARG, "pairwise_relation_reduce_drop_fwd: bad sizes");
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "pairwise_relation_reduce_drop_fwd: p_drop=%f outside [0,1)", (double)p_drop);
  VQA_REQUIRE(vqa_pairwise_relation_reduce_drop_supported(B, N, D) && aligned(v, 16) && aligned(q1, 16) && aligned(q2, 16) &&
                  aligned(v2, 16),
              VQA_E_UNSUPPORTED, "pairwise_relation_reduce_drop_fwd: needs N <= %d, D %% 4 == 0, B*D >= 2^19, aligned tensors", kRegN);
  VQA_LAUNCH(pairwise_fwd_pairs_reg2_kernel, dim3((D / 2 + 63) / 64, B), dim3(64), 0, static_cast<hipStream_t>(stream), v, q1,
                     q2, alpha, alpha_stride, v2, N, D, make_drop(p_drop, seed, seed_ptr));
  return check_launch("pairwise_relation_reduce_drop_fwd");
}
