// K4 (MutanFusion, putils/__init__.py:205-241; config/CoR2.py:172,176) in bf16, RANK-FOLDED -- the form round 1 built for fp32
// (bilinear_folded.hip), now for the mixed-precision configuration (BASELINE configs[4]: 100 regions, 128 samples per GPU).
//
//     out[b,n,h] = sum_r (x[b,n,:] . W1_r[h,:] + b1_r[h]) * h2_r[b,h]
//                = x[b,n,:] . Wb[h,:] + c[b,h],      Wb = sum_r diag(h2_r[b]) W1_r   (per sample, [H,L]),   c = sum_r h2_r b1_r
//
// The R-GEMM form (bf16_path.hip) runs R products per fusion, saves h1 = x W1_r^T + b1_r [M,R,H] for the backward (26 MB
// written per fusion at the config's size), and its backward first materialises g * h2_r [M,R,H] (another 26 MB) for two
// GEMMs over K = R H.  Folded, every product is ONE GEMM per sample with a weight that is built on the fly:
//   forward   out_b = x_b Wb^T + c            Wb folded in registers from the bf16 shadows of W1_r, rounded to bf16, MFMA operand
//   d x       dx_b  = g_b Wb                  the same fold from the transposed shadow w1t
//   d W, d h2 P_b   = g_b^T x_b  [H,L]        one product per sample; dW1_r += diag(h2_r[b]) P_b (fp32 registers, over the samples
//                                             of a slab), dh2_r[b,h] = sum_l P_b[h,l] W1_r[h,l] + b1_r[h] sum_n g[b,n,h]
// -- half the MFMA work of the R = 2 form, no h1, no g * h2 tensor, no prep pass.  At 128 x 100 rows the products are small
// (4.2 GFLOP each); the kernels are built for latency, not for the matrix pipe's peak: operands go straight from L2 to
// registers in MFMA fragment order (forward, d x: no LDS, no barrier), one wave per SIMD.
//   v_mfma_f32_16x16x32_bf16:  A lane (m = lane & 15, k = 8 (lane >> 4) .. + 7), B likewise, D lane (col = lane & 15, rows
//   4 (lane >> 4) + i).  Forward and d x compute the TRANSPOSED product (rows = features, columns = regions), so that a lane ends
//   up with four consecutive features of one region: 8-byte stores.
//   d W: both operands are contracted over the region index, which is their ROW index in memory: the tiles are staged in LDS
//   as they lie in memory and read as fragments by ds_read_b64_tr_b16 (gfx950's transposing LDS read; cdna_hip_programming.md T10).
// Rounding points (oracle/mixed_precision.py restates them): the bf16 shadows of W1_r, Wb in bf16, the stored out / d x in bf16;
// everything else fp32.
#include "common.hpp"

namespace vqa {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));

constexpr int kR = 2;            // ranks (every region fusion of the models)
constexpr int kMaxNB = 8;        // region blocks of 16: N <= 128
constexpr int kSlabs = 16;       // sample slabs of the weight gradient

__device__ __forceinline__ f32x2 unpack2(uint32_t u) { return f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)}; }
__device__ __forceinline__ uint32_t pack2(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2v)); }
__device__ __forceinline__ f32x4v mfma(const u32x4& a, const u32x4& b, const f32x4v& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 ldg16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }

// ------------------------------------------------------------------------------------------------------------------ forward
// One wave = one sample x 64 features (4 blocks) x all regions (NB blocks of 16).  Grid: B * H / 64 waves, 4 per workgroup.
template <int NB>
__global__ __launch_bounds__(256) void fold_fwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w1, const float* __restrict__ b1,
                                                       const float* __restrict__ h2, bf16* __restrict__ out, int B, int N, int L,
                                                       int H, int Hin) {
  const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
  const int wg = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int wps = H / 64;
  const int b = wg / wps, h0 = (wg % wps) * 64;
  if (b >= B) return;
  float q[kR][4];
  const bf16* wa[kR][4];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const int h = h0 + 16 * hb + r16;
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      q[r][hb] = h < Hin ? h2[((size_t)b * kR + r) * Hin + h] : 0.f;
      wa[r][hb] = w1 + ((size_t)r * H + h) * L + 8 * g;
    }
  }
  const bf16* xb[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) xb[nb] = x + ((size_t)b * N + min(16 * nb + r16, N - 1)) * L + 8 * g;
  f32x4v acc[4][NB];
#pragma unroll
  for (int hb = 0; hb < 4; ++hb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[hb][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // One wave per SIMD and operands straight from L2: nothing hides a load but the loop itself -- the raw fragments of chunk c + 1
  // are requested before chunk c is folded and multiplied (two register sets, L / 32 is even)
  struct Raw {
    u32x4 w[kR][4], x[NB];
  };
  auto load = [&](Raw& o, int c) {
#pragma unroll
    for (int hb = 0; hb < 4; ++hb)
#pragma unroll
      for (int r = 0; r < kR; ++r) o.w[r][hb] = ldg16(wa[r][hb] + 32 * c);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) o.x[nb] = ldg16(xb[nb] + 32 * c);
  };
  auto compute = [&](const Raw& in) {
    u32x4 a[4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x2 s = unpack2(in.w[0][hb][e]) * q[0][hb];
#pragma unroll
        for (int r = 1; r < kR; ++r) s += unpack2(in.w[r][hb][e]) * q[r][hb];
        a[hb][e] = pack2(s);
      }
#pragma unroll
    for (int hb = 0; hb < 4; ++hb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[hb][nb] = mfma(a[hb], in.x[nb], acc[hb][nb]);
  };
  const int chunks = L / 32;
  Raw r0, r1;
  load(r0, 0);
  for (int c = 0; c < chunks; c += 2) {
    load(r1, c + 1);
    compute(r0);
    load(r0, min(c + 2, chunks - 1));
    compute(r1);
  }
  // D of block (hb, nb): rows = features h0 + 16 hb + 4 g + i, column = region 16 nb + r16
#pragma unroll
  for (int hb = 0; hb < 4; ++hb) {
    const int h = h0 + 16 * hb + 4 * g;
    f32x4v cb = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (h + i < Hin) {
#pragma unroll
        for (int r = 0; r < kR; ++r) cb[i] = fmaf(h2[((size_t)b * kR + r) * Hin + h + i], b1[(size_t)r * H + h + i], cb[i]);
      }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = 16 * nb + r16;
      if (n < N) {
        const f32x4v v = acc[hb][nb] + cb;
        uint2 o = make_uint2(pack2(f32x2{v[0], v[1]}), pack2(f32x2{v[2], v[3]}));
        *reinterpret_cast<uint2*>(out + ((size_t)b * N + n) * H + h) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ d x
// One wave = one sample x 64 input features l (4 blocks) x all regions; contraction over h (H / 32 chunks).  The fold reads the
// transposed shadow w1t [L, R H] (8 consecutive h of one l = 16 bytes) and the sample's h2 rows, which the wave keeps in LDS
// (2 x H floats, zero past Hin).  gate: d x is multiplied by (x > 0), the relu gradient of the layer in front.
template <int NB>
__global__ __launch_bounds__(256) void fold_dx_kernel(const bf16* __restrict__ g_, const bf16* __restrict__ w1t, const float* __restrict__ h2,
                                                      const bf16* __restrict__ x, bf16* __restrict__ dx, int B, int N, int L, int H,
                                                      int Hin, int gate) {
  extern __shared__ __attribute__((aligned(16))) char fold_smem[];
  const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const int wg = blockIdx.x * 4 + wave;
  const int wps = L / 64;
  const int b = min(wg / wps, B - 1), l0 = (wg % wps) * 64;
  const bool live = wg / wps < B;
  float* hs = reinterpret_cast<float*>(fold_smem) + (size_t)wave * kR * H;   // [R][H] of this wave's sample
  for (int t = lane; t < kR * H; t += 64) {
    const int r = t / H, h = t - r * H;
    hs[t] = h < Hin ? h2[((size_t)b * kR + r) * Hin + h] : 0.f;
  }
  __builtin_amdgcn_s_waitcnt(0);      // (a wave's own LDS writes, read back by the same wave: no workgroup barrier needed,
  __builtin_amdgcn_wave_barrier();    //  but the writes must have landed)
  if (!live) return;
  const bf16* wa[4];
#pragma unroll
  for (int lb = 0; lb < 4; ++lb) wa[lb] = w1t + (size_t)(l0 + 16 * lb + r16) * (kR * H) + 8 * g;
  const bf16* gb[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) gb[nb] = g_ + ((size_t)b * N + min(16 * nb + r16, N - 1)) * H + 8 * g;
  f32x4v acc[4][NB];
#pragma unroll
  for (int lb = 0; lb < 4; ++lb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[lb][nb] = f32x4v{0.f, 0.f, 0.f, 0.f};
  struct Raw {
    u32x4 w[kR][4], x[NB];
  };
  auto load = [&](Raw& o, int c) {
#pragma unroll
    for (int lb = 0; lb < 4; ++lb)
#pragma unroll
      for (int r = 0; r < kR; ++r) o.w[r][lb] = ldg16(wa[lb] + (size_t)r * H + 32 * c);
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) o.x[nb] = ldg16(gb[nb] + 32 * c);
  };
  auto compute = [&](const Raw& in, int c) {
    u32x4 a[4];
    f32x2 qv[kR][4];      // h2_r[b, 32 c + 8 g + 2 e .. + 1]
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      const f32x4v lo = *reinterpret_cast<const f32x4v*>(hs + r * H + 32 * c + 8 * g);
      const f32x4v hi = *reinterpret_cast<const f32x4v*>(hs + r * H + 32 * c + 8 * g + 4);
      qv[r][0] = f32x2{lo[0], lo[1]};
      qv[r][1] = f32x2{lo[2], lo[3]};
      qv[r][2] = f32x2{hi[0], hi[1]};
      qv[r][3] = f32x2{hi[2], hi[3]};
    }
#pragma unroll
    for (int lb = 0; lb < 4; ++lb)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f32x2 s = unpack2(in.w[0][lb][e]) * qv[0][e];
#pragma unroll
        for (int r = 1; r < kR; ++r) s += unpack2(in.w[r][lb][e]) * qv[r][e];
        a[lb][e] = pack2(s);
      }
#pragma unroll
    for (int lb = 0; lb < 4; ++lb)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) acc[lb][nb] = mfma(a[lb], in.x[nb], acc[lb][nb]);
  };
  const int chunks = H / 32;      // (even: H % 64 == 0)
  Raw r0, r1;
  load(r0, 0);
  for (int c = 0; c < chunks; c += 2) {
    load(r1, c + 1);
    compute(r0, c);
    load(r0, min(c + 2, chunks - 1));
    compute(r1, c + 1);
  }
#pragma unroll
  for (int lb = 0; lb < 4; ++lb) {
    const int l = l0 + 16 * lb + 4 * g;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = 16 * nb + r16;
      if (n < N) {
        const size_t off = ((size_t)b * N + n) * L + l;
        f32x4v v = acc[lb][nb];
        if (gate) {
          const uint2 xv = *reinterpret_cast<const uint2*>(x + off);
          const f32x2 x0 = unpack2(xv.x), x1 = unpack2(xv.y);
          v[0] = x0[0] > 0.f ? v[0] : 0.f;
          v[1] = x0[1] > 0.f ? v[1] : 0.f;
          v[2] = x1[0] > 0.f ? v[2] : 0.f;
          v[3] = x1[1] > 0.f ? v[3] : 0.f;
        }
        *reinterpret_cast<uint2*>(dx + off) = make_uint2(pack2(f32x2{v[0], v[1]}), pack2(f32x2{v[2], v[3]}));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ d W, d h2
// Workgroup = 64 features h x half of the input features (L / 2 columns) x one slab of samples; 4 waves = 2 (h halves of 2
// blocks) x 2 (l quarters of LBW = L / 64 blocks).  Per sample: the tiles g_b[:, h tile] and x_b[:, l half] go to LDS as they
// lie in memory (rows = regions, zero rows up to 32 NCH), the fragments of P_b = g_b^T x_b come back through the transposing
// read, and P_b is folded into the slab's dW1_r accumulators and contracted against W1_r (registers, D layout) for d h2.
struct DwArgs {
  const bf16* g;        // [B N, H]
  const bf16* x;        // [B N, L]
  const bf16* w1;       // [R, H, L] bf16 shadow
  const float* h2;      // [B, R, Hin]
  float* slab;          // [kSlabs][R][H][L]
  float* dh2p;          // [2][B][R][H]   partial sums over the two l halves
  float* gsum;          // [B][H]         sum_n g[b,n,h]
  int B, N, L, H, Hin;
  int sps;              // samples per slab
};
template <int LBW, int NCH>
__global__ __launch_bounds__(256, 1) void fold_dw_kernel(DwArgs p) {
  extern __shared__ __attribute__((aligned(16))) char fold_smem[];
  constexpr int ROWS = 32 * NCH;
  const int L = p.L, H = p.H, N = p.N, Lh = L / 2;
  const int pitch_x = Lh * 2 + 16, pitch_g = 64 * 2 + 16;
  char* xs = fold_smem;
  char* gs = fold_smem + (size_t)ROWS * pitch_x;
  float* red = reinterpret_cast<float*>(gs + (size_t)ROWS * pitch_g);   // [2 l-waves][R][64]  +  [4][64] for the region sums
  const int lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
  const int wh = wave >> 1, wl = wave & 1;
  const int ht = blockIdx.x % (H / 64), lh = (blockIdx.x / (H / 64)) & 1, slab = blockIdx.x / (2 * (H / 64));
  const int h0 = ht * 64, lbase = lh * Lh;
  // W1_r in D layout: block (j, lb): rows h0 + 16 (2 wh + j) + 4 g + i, column lbase + 16 (LBW wl + lb) + r16
  float w1f[kR][2][LBW][4];
#pragma unroll
  for (int r = 0; r < kR; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          w1f[r][j][lb][i] = (float)p.w1[((size_t)r * H + h0 + 16 * (2 * wh + j) + 4 * g + i) * L + lbase + 16 * (LBW * wl + lb) + r16];
  f32x4v dw[kR][2][LBW];
#pragma unroll
  for (int r = 0; r < kR; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb) dw[r][j][lb] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // transposing reads: lane 4 q + p of a 16-lane group addresses row q, columns 4 p .. 4 p + 3 of a 4 x 16 block
  const int tq = r16 >> 2, tp = r16 & 3;
  const int b_lo = slab * p.sps, b_hi = min(p.B, b_lo + p.sps);
  // staging: 16-byte pieces, rows = regions (zero past N); the next sample's pieces are requested while this one is multiplied
  constexpr int XP = (ROWS * (LBW * 4) + 255) / 256, GP = ROWS * 8 / 256;      // pieces per thread (Lh / 8 = 4 LBW per row)
  u32x4 sx[XP], sg[GP];
  auto fetch = [&](int b) {
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int t = threadIdx.x + 256 * k;
      const int row = t / (LBW * 4), pc = t - row * (LBW * 4);
      sx[k] = u32x4{0u, 0u, 0u, 0u};
      if (row < N) sx[k] = ldg16(p.x + ((size_t)b * N + row) * L + lbase + 8 * pc);
    }
#pragma unroll
    for (int k = 0; k < GP; ++k) {
      const int t = threadIdx.x + 256 * k;
      const int row = t >> 3, pc = t & 7;
      sg[k] = u32x4{0u, 0u, 0u, 0u};
      if (row < N) sg[k] = ldg16(p.g + ((size_t)b * N + row) * H + h0 + 8 * pc);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int k = 0; k < XP; ++k) {
      const int t = threadIdx.x + 256 * k;
      const int row = t / (LBW * 4), pc = t - row * (LBW * 4);
      if (row < ROWS) *reinterpret_cast<u32x4*>(xs + (size_t)row * pitch_x + 16 * pc) = sx[k];
    }
#pragma unroll
    for (int k = 0; k < GP; ++k) {
      const int t = threadIdx.x + 256 * k;
      *reinterpret_cast<u32x4*>(gs + (size_t)(t >> 3) * pitch_g + 16 * (t & 7)) = sg[k];
    }
  };
  if (b_lo < b_hi) fetch(b_lo);
  for (int b = b_lo; b < b_hi; ++b) {
    __syncthreads();      // the previous sample's fragments have been read
    stage();
    __syncthreads();
    if (b + 1 < b_hi) fetch(b + 1);
    f32x4v P[2][LBW];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb) P[j][lb] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      u32x4 a[2], bx[LBW];
      const int row = 32 * c + 8 * g + tq;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const char* src = gs + (size_t)row * pitch_g + (16 * (2 * wh + j) + 4 * tp) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(src));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(src + 4 * pitch_g));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2v = __builtin_bit_cast(uint2, hi);
        a[j] = u32x4{l2.x, l2.y, h2v.x, h2v.y};
      }
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb) {
        const char* src = xs + (size_t)row * pitch_x + (16 * (LBW * wl + lb) + 4 * tp) * 2;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(src));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(src + 4 * pitch_x));
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2v = __builtin_bit_cast(uint2, hi);
        bx[lb] = u32x4{l2.x, l2.y, h2v.x, h2v.y};
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int lb = 0; lb < LBW; ++lb) P[j][lb] = mfma(a[j], bx[lb], P[j][lb]);
    }
    // fold into the slab's weight gradients; contract against W1_r for d h2
    float part[kR][2][4];
#pragma unroll
    for (int r = 0; r < kR; ++r)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4v qh;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int h = h0 + 16 * (2 * wh + j) + 4 * g + i;
          qh[i] = h < p.Hin ? p.h2[((size_t)b * kR + r) * p.Hin + h] : 0.f;
          part[r][j][i] = 0.f;
        }
#pragma unroll
        for (int lb = 0; lb < LBW; ++lb) {
          dw[r][j][lb] += qh * P[j][lb];
#pragma unroll
          for (int i = 0; i < 4; ++i) part[r][j][i] = fmaf(P[j][lb][i], w1f[r][j][lb][i], part[r][j][i]);
        }
      }
    // sum over the 16 columns of a block (lanes r16), then over the two l-waves through LDS
#pragma unroll
    for (int r = 0; r < kR; ++r)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s = part[r][j][i];
          s += __shfl_xor(s, 1);
          s += __shfl_xor(s, 2);
          s += __shfl_xor(s, 4);
          s += __shfl_xor(s, 8);
          if (r16 == 0) red[(wl * kR + r) * 64 + 16 * (2 * wh + j) + 4 * g + i] = s;
        }
    if (lh == 0) {        // sum_n g[b,n,h] for the bias terms: thread (h = t & 63, row quarter t >> 6)
      const int h = threadIdx.x & 63, qr = threadIdx.x >> 6;
      float s = 0.f;
      for (int row = qr * (ROWS / 4); row < (qr + 1) * (ROWS / 4); ++row)
        s += (float)*reinterpret_cast<const bf16*>(gs + (size_t)row * pitch_g + 2 * h);
      red[2 * kR * 64 + qr * 64 + h] = s;
    }
    __syncthreads();
    if (threadIdx.x < kR * 64) {
      const int r = threadIdx.x >> 6, h = threadIdx.x & 63;
      p.dh2p[(((size_t)lh * p.B + b) * kR + r) * H + h0 + h] = red[(0 * kR + r) * 64 + h] + red[(1 * kR + r) * 64 + h];
    } else if (lh == 0 && threadIdx.x < kR * 64 + 64) {
      const int h = threadIdx.x & 63;
      const float* q4 = red + 2 * kR * 64 + h;
      p.gsum[(size_t)b * H + h0 + h] = ((q4[0] + q4[64]) + q4[128]) + q4[192];
    }
  }
  float* out = p.slab + (size_t)slab * kR * H * L;
#pragma unroll
  for (int r = 0; r < kR; ++r)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int lb = 0; lb < LBW; ++lb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          out[((size_t)r * H + h0 + 16 * (2 * wh + j) + 4 * g + i) * L + lbase + 16 * (LBW * wl + lb) + r16] = dw[r][j][lb][i];
}

// one launch behind it: (A) dW1_r = fixed-order sum of the slabs, cropped to the master shape; (B) d h2 = the two l halves +
// b1_r[h] sum_n g; (C) d b1_r[h] = sum_b h2_r[b,h] sum_n g[b,n,h]  (64 columns x 4 sample slices per block, slices meet in LDS)
struct FinishArgs {
  const float* slab;
  const float* dh2p;
  const float* gsum;
  const float* h2;
  const float* b1;      // [R, H] padded shadow
  float* dw[kR];
  float* db[kR];
  float* dh2;           // [B, R, Hin]
  int B, L, H, Hin, Lout, slabs;
  int blocks_a, blocks_b;
};
__global__ __launch_bounds__(256) void fold_finish_kernel(FinishArgs p) {
  const int bid = blockIdx.x;
  if (bid < p.blocks_a) {
    const int e = bid * 256 + threadIdx.x;
    if (e >= kR * p.Hin * p.Lout) return;
    const int r = e / (p.Hin * p.Lout), u = e - r * (p.Hin * p.Lout);
    const int h = u / p.Lout, l = u - h * p.Lout;
    const float* src = p.slab + ((size_t)r * p.H + h) * p.L + l;
    const size_t stride = (size_t)kR * p.H * p.L;
    float s = 0.f;
    for (int k = 0; k < p.slabs; ++k) s += src[k * stride];
    p.dw[r][(size_t)h * p.Lout + l] = s;
  } else if (bid < p.blocks_a + p.blocks_b) {
    const int e = (bid - p.blocks_a) * 256 + threadIdx.x;
    if (e >= p.B * kR * p.Hin) return;
    const int b = e / (kR * p.Hin), u = e - b * (kR * p.Hin);
    const int r = u / p.Hin, h = u - r * p.Hin;
    const size_t o = ((size_t)b * kR + r) * p.H + h;
    p.dh2[e] = (p.dh2p[o] + p.dh2p[(size_t)p.B * kR * p.H + o]) + p.b1[(size_t)r * p.H + h] * p.gsum[(size_t)b * p.H + h];
  } else {
    __shared__ float part[3][64];
    const int blk = bid - p.blocks_a - p.blocks_b;           // over R * ceil(Hin / 64) column groups
    const int groups = (p.Hin + 63) / 64;
    const int r = blk / groups, h = (blk % groups) * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float s = 0.f;
    if (h < p.Hin)
      for (int b = sl; b < p.B; b += 4) s = fmaf(p.h2[((size_t)b * kR + r) * p.Hin + h], p.gsum[(size_t)b * p.H + h], s);
    if (sl > 0) part[sl - 1][threadIdx.x & 63] = s;
    __syncthreads();
    if (sl == 0 && h < p.Hin) p.db[r][h] = ((s + part[0][threadIdx.x]) + part[1][threadIdx.x]) + part[2][threadIdx.x];
  }
}

size_t round256(size_t b) { return (b + 255) & ~(size_t)255; }
bool shape_ok(int B, int N, int L, int H, int R) {
  return R == kR && B >= 1 && N >= 1 && N <= 16 * kMaxNB && L % 64 == 0 && L >= 64 && L / 64 <= 5 && H % 64 == 0 && H >= 64 &&
         (size_t)B * N * H < (1ull << 31) && (size_t)B * N * L < (1ull << 31);
}

}  // namespace
}  // namespace vqa

using namespace vqa;

// include/vqa_mi355x.h
extern "C" int vqa_bilinear_fold_bf16_supported(int B, int N, int L, int H, int R) { return shape_ok(B, N, L, H, R) ? 1 : 0; }

extern "C" int vqa_bilinear_fold_fwd_bf16(const vqa_bf16_t* x, const vqa_bf16_t* w1, const float* b1, const float* h2, vqa_bf16_t* out,
                                          int B, int N, int L, int H, int R, int H_in, vqa_stream_t stream) {
  VQA_REQUIRE(x && w1 && b1 && h2 && out, VQA_E_BADARG, "bilinear_fold_fwd_bf16: null pointer");
  VQA_REQUIRE(shape_ok(B, N, L, H, R), VQA_E_UNSUPPORTED, "bilinear_fold_fwd_bf16: needs R = 2, N <= 128, L %% 64 == 0 (<= 320), H %% 64 == 0 (B=%d N=%d L=%d H=%d R=%d)",
              B, N, L, H, R);
  VQA_REQUIRE(H_in > 0 && H_in <= H, VQA_E_BADARG, "bilinear_fold_fwd_bf16: h2 width %d exceeds the padded H = %d", H_in, H);
  VQA_REQUIRE(aligned(x, 16) && aligned(w1, 16) && aligned(out, 8), VQA_E_UNSUPPORTED, "bilinear_fold_fwd_bf16: x, w1 must be 16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int waves = B * (H / 64);
  const dim3 grid((waves + 3) / 4);
  const bf16* xb = reinterpret_cast<const bf16*>(x);
  const bf16* wb = reinterpret_cast<const bf16*>(w1);
  bf16* ob = reinterpret_cast<bf16*>(out);
#define FWD(NB_) VQA_LAUNCH((fold_fwd_kernel<NB_>), grid, dim3(256), 0, s, xb, wb, b1, h2, ob, B, N, L, H, H_in)
  switch ((N + 15) / 16) {
    case 1: FWD(1); break;
    case 2: FWD(2); break;
    case 3: FWD(3); break;
    case 4: FWD(4); break;
    case 5: FWD(5); break;
    case 6: FWD(6); break;
    case 7: FWD(7); break;
    default: FWD(8); break;
  }
#undef FWD
  return check_launch("bilinear_fold_fwd_bf16");
}

extern "C" size_t vqa_bilinear_fold_bwd_bf16_workspace_bytes(int B, int N, int L, int H, int R) {
  if (!shape_ok(B, N, L, H, R)) return 0;
  return round256((size_t)kSlabs * kR * H * L * 4) + round256((size_t)2 * B * kR * H * 4) + round256((size_t)B * H * 4);
}

extern "C" int vqa_bilinear_fold_bwd_bf16(const vqa_bf16_t* x, const vqa_bf16_t* w1, const vqa_bf16_t* w1t, const float* b1, const float* h2,
                                          const vqa_bf16_t* g, vqa_bf16_t* d_x, float* const* d_w1, float* const* d_b1, float* d_h2,
                                          void* workspace, size_t workspace_bytes, int B, int N, int L, int H, int R, int H_out,
                                          int L_out, int gate_dx, vqa_stream_t stream) {
  VQA_REQUIRE(x && w1 && b1 && h2 && g && d_w1 && d_b1 && d_h2 && workspace, VQA_E_BADARG, "bilinear_fold_bwd_bf16: null pointer");
  VQA_REQUIRE(d_x == nullptr || w1t != nullptr, VQA_E_BADARG, "bilinear_fold_bwd_bf16: d_x needs w1t");
  VQA_REQUIRE(shape_ok(B, N, L, H, R), VQA_E_UNSUPPORTED, "bilinear_fold_bwd_bf16: needs R = 2, N <= 128, L %% 64 == 0 (<= 320), H %% 64 == 0 (B=%d N=%d L=%d H=%d R=%d)",
              B, N, L, H, R);
  VQA_REQUIRE(H_out > 0 && H_out <= H && L_out > 0 && L_out <= L, VQA_E_BADARG,
              "bilinear_fold_bwd_bf16: master shape [%d,%d] exceeds the padded one [%d,%d]", H_out, L_out, H, L);
  VQA_REQUIRE(workspace_bytes >= vqa_bilinear_fold_bwd_bf16_workspace_bytes(B, N, L, H, R), VQA_E_BADARG,
              "bilinear_fold_bwd_bf16: workspace of %zu B is too small", workspace_bytes);
  VQA_REQUIRE(aligned(x, 16) && aligned(g, 16) && aligned(w1, 16) && aligned(workspace, 256) &&
                  (d_x == nullptr || (aligned(d_x, 8) && aligned(w1t, 16))),
              VQA_E_UNSUPPORTED, "bilinear_fold_bwd_bf16: tensors must be 16-byte aligned (workspace 256)");
  for (int r = 0; r < R; ++r)
    VQA_REQUIRE(d_w1[r] != nullptr && d_b1[r] != nullptr, VQA_E_BADARG, "bilinear_fold_bwd_bf16: null gradient %d", r);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bf16* xb = reinterpret_cast<const bf16*>(x);
  const bf16* gb = reinterpret_cast<const bf16*>(g);
  char* ws = static_cast<char*>(workspace);
  float* slab = reinterpret_cast<float*>(ws);
  float* dh2p = reinterpret_cast<float*>(ws + round256((size_t)kSlabs * kR * H * L * 4));
  float* gsum = reinterpret_cast<float*>(ws + round256((size_t)kSlabs * kR * H * L * 4) + round256((size_t)2 * B * kR * H * 4));
  const int NB = (N + 15) / 16;
  if (d_x != nullptr) {
    const int waves = B * (L / 64);
    const dim3 grid((waves + 3) / 4);
    const size_t lds = (size_t)4 * kR * H * sizeof(float);
    const bf16* wt = reinterpret_cast<const bf16*>(w1t);
    bf16* dxb = reinterpret_cast<bf16*>(d_x);
#define DX(NB_) VQA_LAUNCH((fold_dx_kernel<NB_>), grid, dim3(256), lds, s, gb, wt, h2, xb, dxb, B, N, L, H, H_out, gate_dx)
    switch (NB) {
      case 1: DX(1); break;
      case 2: DX(2); break;
      case 3: DX(3); break;
      case 4: DX(4); break;
      case 5: DX(5); break;
      case 6: DX(6); break;
      case 7: DX(7); break;
      default: DX(8); break;
    }
#undef DX
  }
  DwArgs a{};
  a.g = gb;
  a.x = xb;
  a.w1 = reinterpret_cast<const bf16*>(w1);
  a.h2 = h2;
  a.slab = slab;
  a.dh2p = dh2p;
  a.gsum = gsum;
  a.B = B;
  a.N = N;
  a.L = L;
  a.H = H;
  a.Hin = H_out;
  a.sps = (B + kSlabs - 1) / kSlabs;
  const int nch = (N + 31) / 32, lbw = L / 64;
  const size_t lds = (size_t)32 * nch * ((L / 2) * 2 + 16) + (size_t)32 * nch * (64 * 2 + 16) + (2 * kR * 64 + 4 * 64) * sizeof(float);
  const dim3 grid(kSlabs * 2 * (H / 64));
  bool launched = false;
#define DW(LBW_, NCH_)                                                                       \
  if (lbw == LBW_ && nch == NCH_) {                                                          \
    VQA_ENSURE_LDS((fold_dw_kernel<LBW_, NCH_>), lds);                                       \
    VQA_LAUNCH((fold_dw_kernel<LBW_, NCH_>), grid, dim3(256), lds, s, a);                    \
    launched = true;                                                                         \
  }
#define DW_ALL(LBW_) DW(LBW_, 1) DW(LBW_, 2) DW(LBW_, 3) DW(LBW_, 4)
  DW_ALL(1) DW_ALL(2) DW_ALL(3) DW_ALL(4) DW_ALL(5)
#undef DW_ALL
#undef DW
  VQA_REQUIRE(launched, VQA_E_UNSUPPORTED, "bilinear_fold_bwd_bf16: no weight-gradient kernel for L=%d N=%d", L, N);
  FinishArgs f{};
  f.slab = slab;
  f.dh2p = dh2p;
  f.gsum = gsum;
  f.h2 = h2;
  f.b1 = b1;
  for (int r = 0; r < kR; ++r) {
    f.dw[r] = d_w1[r];
    f.db[r] = d_b1[r];
  }
  f.dh2 = d_h2;
  f.B = B;
  f.L = L;
  f.H = H;
  f.Hin = H_out;
  f.Lout = L_out;
  f.slabs = kSlabs;
  f.blocks_a = (kR * H_out * L_out + 255) / 256;
  f.blocks_b = (B * kR * H_out + 255) / 256;
  const int blocks_c = kR * ((H_out + 63) / 64);
  VQA_LAUNCH(fold_finish_kernel, dim3(f.blocks_a + f.blocks_b + blocks_c), dim3(256), 0, s, f);
  return check_launch("bilinear_fold_bwd_bf16");
}
