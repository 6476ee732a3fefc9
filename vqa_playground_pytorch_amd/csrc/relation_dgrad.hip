// K1 -> K5 fusion, backward: the data gradient of the second region projection (compress_v2, config/CoR2.py:218 on the
// relation tensor of :191-199,:216) reduced straight to what the relation step needs.
//
// In the closed form of the relation step (pairwise_relation.hip, "relation apply") compress_v2 reads
//     x[m,:] = keep(m,:) * (t[b,:] + c2[b,:] * v[m,:])            m = b N + n, keep = the layer's input dropout
// and nothing else does.  v is an input of the model, so the gradient of x is wanted for two per-sample vectors only:
//     d_t[b,:]  = sum_n keep(m,:) * dx[m,:]            d_c2[b,:] = sum_n keep(m,:) * dx[m,:] * v[m,:]
//     dx[m,:]   = sum_l gz[m,l] W[l,:]                  gz = dL/d(pre-activation) [M,L],  W [L,D] (nn.Conv1d weight, k = 1)
// As two launches that is a 23.4 GFLOP GEMM that WRITES dx (M x D x 4 = 151 MB at B = 512) and a streaming kernel that
// reads it back with v (relation_apply_bwd): 233 + 62 us.  Here the GEMM tile is 144 rows = FOUR WHOLE SAMPLES of 36
// regions (9 row blocks of 16: gemm_f32_rt.hpp's register-tile shape), and the tile never leaves the registers: the
// epilogue masks it, multiplies by v, adds the rows of each sample up and stores 4 x 64 floats of d_t and d_c2 per wave.
//
// Register-tile engine, NN form: A = gz, K-contiguous (one 16-byte load per row block and 16-deep chunk, component kb =
// contraction step kb); B = W, contraction index = row: lane (r, g) loads W[16 c + 4 g + kb][n0 + 4 r .. + 3] as one
// 16-byte load per step, component e feeds accumulator block e (the column permutation of gemm_tn_kernel).  A wave owns
// 144 rows x 64 columns (36 blocks, 144 accumulator registers), a workgroup 144 x 256; no barrier.
//
// Two forms of the main loop (VQA_RELDG_TUNE selects; template parameter TUNE):
//  * TUNE = 4, the default -- TWO workgroups per CU.  A third of this kernel is not matrix work (per tile: a prologue of
//    two memory latencies, and the epilogue's mask + multiply by v + per-sample sums), and with one wave per SIMD the
//    matrix pipe idles through all of it (72 % busy).  Within 256 registers (236 used: the 144 accumulators, ONE fragment
//    set, addresses) a wave refills its fragments in halves -- the A fragments of row blocks 5..8 land under the MFMAs of
//    blocks 0..4, the next chunk's B and A 0..4 under those of 5..8 -- and reads its tile of v straight from memory in
//    the epilogue; whatever latency that leaves is covered by the SIMD's other wave, whose main loop also runs under this
//    wave's epilogue.  199 us against 230 at B = 512.
//  * TUNE = 0 -- one workgroup per CU, double-buffered fragment sets (340 registers), the wave's tile of v (36 loads of
//    1 KB) streamed into LDS by buffer_load ... lds (no registers, no wait before the epilogue), two loads per chunk of the
//    main loop: requested in one burst at the top of the tile they cost 36 us (every wave of the chip asking HBM for 36 KB
//    at once, the first MFMA behind it: memory operations retire in order), and kept in registers a runtime row-block
//    index cannot address them.  TUNE = 1..3 are ablations of this form.
#include <cstdlib>

#include "gemm_f32_rt.hpp"

namespace vqa {
namespace {

struct RelDgradArgs {
  const float* gz;   // [M, L]
  const float* w;    // [L, D]
  const float* v;    // [M, D]
  float* d_t;        // [B, D]
  float* d_c2;       // [B, D]
  int B, M, L, D;
  int tiles_n;       // workgroup tiles of 256 columns
};

constexpr int kRegions = 36;                 // regions per sample: 4 samples = 144 rows = 9 row blocks
constexpr int kRB = 9, kSamples = 4, kBM = 16 * kRB;
constexpr int kVAux = 2;   // cache policy of the v stream: nt (read once: keep it from displacing W and gz in L2)

template <int TUNE>
__global__ __launch_bounds__(rt::kThreads, (TUNE & 4) ? 2 : 1) void relation_dgrad_kernel(RelDgradArgs p, DropCfg dc) {
  using rt::f32x4;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (tile / p.tiles_n) * kBM;
  const int n0 = (tile % p.tiles_n) * 256 + 64 * wave;
  if (n0 >= p.D) return;                     // (no barrier in this kernel)
  const int M = p.M, L = p.L, D = p.D;

  uint32_t offA[kRB];
#pragma unroll
  for (int i = 0; i < kRB; ++i) offA[i] = ((uint32_t)min(m0 + 16 * i + r, M - 1) * (uint32_t)L + 4u * g) * 4u;
  const uint32_t offB = ((uint32_t)(4 * g) * (uint32_t)D + (uint32_t)(n0 + 4 * r)) * 4u;
  const rt::rsrc_t Ab = rt::make_rsrc(p.gz, (size_t)M * L * 4);
  const rt::rsrc_t Bb = rt::make_rsrc(p.w, (size_t)L * D * 4);
  const rt::rsrc_t Vb = rt::make_rsrc(p.v, (size_t)M * D * 4);

  f32x4 acc[kRB][4];   // [row block][column class e]: rows 16 i + 4 g + t (register t), column n0 + 4 r + e
#pragma unroll
  for (int i = 0; i < kRB; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[i][e] = f32x4{0.f, 0.f, 0.f, 0.f};

  struct Frag {
    f32x4 a[kRB];      // component kb = contraction step kb
    f32x4 b[4];        // [kb]: component e = column class e
  };
  auto load = [&](Frag& f, int c) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) f.b[kb] = rt::ldg16(Bb, offB, (uint32_t)(16 * c + kb) * (uint32_t)D * 4u);
#pragma unroll
    for (int i = 0; i < kRB; ++i) f.a[i] = rt::ldg16(Ab, offA[i], (uint32_t)c * 64u);
  };
  // the wave's tile of v in LDS: slot idx = 4 i + t (1 KB: lane l at byte 16 l) = v[m0 + 16 i + 4 g + t][n0 + 4 r .. + 3]
  extern __shared__ __attribute__((aligned(16))) char rd_smem[];
  char* vslots = rd_smem + (size_t)wave * (4 * kRB) * 1024;
  auto load_v = [&](int idx) {               // idx is wave-uniform (a loop counter); rows clamped per lane
    const int row = min(m0 + 16 * (idx >> 2) + 4 * g + (idx & 3), M - 1);
    const uint32_t voff = ((uint32_t)row * (uint32_t)D + (uint32_t)(n0 + 4 * r)) * 4u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(Vb, (__attribute__((address_space(3))) void*)(vslots + idx * 1024), 16, (int)voff, 0, 0, kVAux);
  };
  auto mfmas = [&](const Frag& f) {
#pragma unroll
    for (int i = 0; i < kRB; ++i)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[i][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][kb], f.b[kb][e], acc[i][e], 0, 0, 0);
  };
  // one pipeline step (gemm_nt_kernel): chunk cn is requested in the shadow of the first MFMAs of chunk c, one load per
  // PER MFMAs; two slots of the v tile follow the fragment loads
  auto step = [&](Frag& fn, int cn, const Frag& f, int vidx) {
    load(fn, cn);
    if constexpr ((TUNE & 1) == 0) {
      if (vidx < 4 * kRB) {                  // (uniform; 4 kRB is even)
        load_v(vidx);
        load_v(vidx + 1);
      }
    }
    mfmas(f);
    constexpr int NL = kRB + 4 + 2, NM = 16 * kRB, PER = 5;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            // MFMA
      if (m % PER == PER - 1 && m / PER < NL) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nfull = L >> 4;                  // whole 16-deep chunks
  const int c_pairs = (TUNE & 4) ? 0 : (nfull & ~1);
  if constexpr ((TUNE & 4) != 0) {
    // two workgroups per CU (<= 256 registers), no v tile in LDS (v is read in the epilogue, under the other wave's main loop).
    // One fragment set, refilled in halves: the A fragments of row blocks 5..8 land under the MFMAs of row blocks 0..4 and
    // the next chunk's B and A 0..4 under those of 5..8; what latency is left is covered by the SIMD's other wave.
    constexpr int kLo = 5, kHi = kRB - kLo;
    f32x4 lo[kLo], hi[kHi], b0[4], b1[4];
    auto load_b = [&](f32x4 (&b)[4], int c) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) b[kb] = rt::ldg16(Bb, offB, (uint32_t)(16 * c + kb) * (uint32_t)D * 4u);
    };
    auto load_lo = [&](int c) {
#pragma unroll
      for (int i = 0; i < kLo; ++i) lo[i] = rt::ldg16(Ab, offA[i], (uint32_t)c * 64u);
    };
    auto load_hi = [&](int c) {
#pragma unroll
      for (int i = 0; i < kHi; ++i) hi[i] = rt::ldg16(Ab, offA[kLo + i], (uint32_t)c * 64u);
    };
    auto mf_lo = [&](const f32x4 (&b)[4]) {
#pragma unroll
      for (int i = 0; i < kLo; ++i)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(lo[i][kb], b[kb][e], acc[i][e], 0, 0, 0);
    };
    auto mf_hi = [&](const f32x4 (&b)[4]) {
#pragma unroll
      for (int i = 0; i < kHi; ++i)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            acc[kLo + i][e] = __builtin_amdgcn_mfma_f32_16x16x4f32(hi[i][kb], b[kb][e], acc[kLo + i][e], 0, 0, 0);
    };
    auto half = [&](const f32x4 (&b)[4], f32x4 (&bn)[4], int c) {
      load_hi(c);
      __builtin_amdgcn_sched_barrier(0);
      mf_lo(b);
      __builtin_amdgcn_sched_barrier(0);
      const int cn = min(c + 1, nfull - 1);          // (last chunk: a harmless reload)
      load_b(bn, cn);
      load_lo(cn);
      __builtin_amdgcn_sched_barrier(0);
      mf_hi(b);
      __builtin_amdgcn_sched_barrier(0);
    };
    if (nfull > 0) {
      load_b(b0, 0);
      load_lo(0);
      for (int c = 0; c < nfull; c += 2) {
        half(b0, b1, c);
        if (c + 1 < nfull) half(b1, b0, c + 1);
      }
    }
  } else {
    Frag f0, f1;
    int vidx = 0;
    if (c_pairs > 0) {
      load(f0, 0);
      __builtin_amdgcn_sched_barrier(0);
      for (int c = 0; c < c_pairs; c += 2) {
        step(f1, c + 1, f0, vidx);
        step(f0, min(c + 2, c_pairs - 1), f1, vidx + 2);     // (last pair: a harmless reload)
        vidx += 4;
      }
    }
    if constexpr ((TUNE & 1) == 0) {
      for (; vidx < 4 * kRB; ++vidx) load_v(vidx);           // (a contraction shorter than 18 chunks)
    }
    if (c_pairs < nfull) {
      load(f0, c_pairs);
      mfmas(f0);
    }
  }
  if ((L & 15) != 0) {
    // contraction tail (< 16): element-wise guarded loads of gz, rows of W beyond L read as zero through the select
    Frag f;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int k = 16 * nfull + 4 * g + kb;
      const bool ok = k < L;
      const uint32_t kc = (uint32_t)min(k, L - 1);
#pragma unroll
      for (int i = 0; i < kRB; ++i) {
        const float x = rt::ldg4(Ab, offA[i] - 16u * g + kc * 4u, 0u);
        f.a[i][kb] = ok ? x : 0.f;
      }
      const f32x4 wv = rt::ldg16(Bb, (uint32_t)(n0 + 4 * r) * 4u, kc * (uint32_t)D * 4u);
      f.b[kb] = ok ? wv : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    mfmas(f);
  }

  // ---- epilogue: mask, multiply by v, add the rows of each sample up ----
  __builtin_amdgcn_s_waitcnt(0);             // the LDS-bound loads of the v tile have landed (nobody else reads these slots)
  auto vt = [&](int i, int t) -> f32x4 {
    if constexpr ((TUNE & 1) != 0) return f32x4{1.f, 1.f, 1.f, 1.f};
    if constexpr ((TUNE & 4) != 0) {
      const int row = min(m0 + 16 * i + 4 * g + t, M - 1);
      return rt::ldg16(Vb, (uint32_t)(n0 + 4 * r) * 4u + (uint32_t)row * (uint32_t)D * 4u, 0u);
    }
    return *reinterpret_cast<const f32x4*>(vslots + (4 * i + t) * 1024 + lane * 16);
  };
  if constexpr ((TUNE & 2) != 0) {   // (experiment: no epilogue math)
    f32x4 sum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < kRB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) sum += acc[i][e] * vt(i, e);
    if (g == 0) *reinterpret_cast<f32x4*>(p.d_t + (size_t)(m0 / kRegions) * D + n0 + 4 * r) = sum;
    return;
  }
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
  // MODE: 0 no dropout, 1 the one-bit-per-element p = 0.5 mask, 2 a byte per element -- ONE uniform branch around the whole
  // reduction (a test per row left 70 branches in it and nothing for the scheduler to overlap)
  auto reduce = [&](auto mode) {
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
  // the four lane groups hold different rows of the same columns: lane-wise sum over the 16-lane rows of the wave
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

}  // namespace
}  // namespace vqa

using namespace vqa;

// include/vqa_mi355x.h
extern "C" int vqa_relation_projection_dgrad_supported(int B, int N, int D, int L) {
  const bool off = vqa::option_is("VQA_FUSE_RELATION_DGRAD", '0');
  return !off && N == kRegions && B >= 1 && D % 64 == 0 && D >= 64 && L >= 32 && L % 2 == 0 && (size_t)B * N * D * 4 < (1ull << 32) &&
         (size_t)L * D * 4 < (1ull << 32);
}

extern "C" int vqa_relation_projection_dgrad(const float* gz, const float* w, const float* v, float* d_t, float* d_c2,
                                             float p_drop, uint64_t seed, const uint64_t* seed_ptr, int B, int N, int D, int L,
                                             vqa_stream_t stream) {
  VQA_REQUIRE(gz && w && v && d_t && d_c2, VQA_E_BADARG, "relation_projection_dgrad: null pointer");
  VQA_REQUIRE(p_drop >= 0.f && p_drop < 1.f, VQA_E_BADARG, "relation_projection_dgrad: p_drop=%f outside [0,1)", (double)p_drop);
  VQA_REQUIRE(B > 0 && N > 0 && D > 0 && L > 0, VQA_E_BADARG, "relation_projection_dgrad: bad sizes B=%d N=%d D=%d L=%d", B, N, D, L);
  VQA_REQUIRE(vqa_relation_projection_dgrad_supported(B, N, D, L), VQA_E_UNSUPPORTED,
              "relation_projection_dgrad: needs N = 36 regions, D %% 64 == 0, even L >= 32 (N=%d D=%d L=%d)", N, D, L);
  VQA_REQUIRE(aligned(w, 16) && aligned(v, 16) && aligned(d_t, 16) && aligned(d_c2, 16) && aligned(gz, 8), VQA_E_UNSUPPORTED,
              "relation_projection_dgrad: w, v, d_t, d_c2 must be 16-byte aligned, gz 8-byte");
  RelDgradArgs a{};
  a.gz = gz;
  a.w = w;
  a.v = v;
  a.d_t = d_t;
  a.d_c2 = d_c2;
  a.B = B;
  a.M = B * N;
  a.L = L;
  a.D = D;
  a.tiles_n = (D + 255) / 256;
  const int tiles_m = (a.M + kBM - 1) / kBM;
  const DropCfg dc = make_drop(p_drop, seed, seed_ptr);
  // 4 (default): two workgroups per CU, fragments refilled in halves, v read in the epilogue; 0: one workgroup per CU, double-
  // buffered fragments, v streamed into LDS (the first form: 230 us against 199 at B = 512); 1..3: ablations of the latter
  const int tune = vqa::option("VQA_RELDG_TUNE") ? std::atoi(vqa::option("VQA_RELDG_TUNE")) : 4;
  const dim3 grid((unsigned)(tiles_m * a.tiles_n));
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr size_t lds = (size_t)4 * 4 * kRB * 1024;      // 4 waves x 36 slots of 1 KB
#define VQA_RD_LAUNCH(T_)                                                                \
  {                                                                                      \
    VQA_ENSURE_LDS(relation_dgrad_kernel<T_>, lds);                                      \
    VQA_LAUNCH(relation_dgrad_kernel<T_>, grid, dim3(rt::kThreads), lds, s, a, dc); \
  }
  if (tune == 1) VQA_RD_LAUNCH(1)
  else if (tune == 2) VQA_RD_LAUNCH(2)
  else if (tune == 3) VQA_RD_LAUNCH(3)
  else if (tune == 4) VQA_LAUNCH(relation_dgrad_kernel<4>, grid, dim3(rt::kThreads), 0, s, a, dc);
  else VQA_RD_LAUNCH(0)
#undef VQA_RD_LAUNCH
  return check_launch("relation_projection_dgrad");
}
