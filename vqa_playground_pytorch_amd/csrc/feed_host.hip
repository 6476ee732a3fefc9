// Host side of the feed (no device code): the per-sample row gather of the reference's loader -- `outer.data['img']['feature'][visual_index]`
// once per sample in Inner.__getitem__ (datasets.py:912-913), collated by the DataLoader's worker processes (:975-977) -- as ONE call
// that copies the batch's rows from the memory-mapped feature store into the (pinned) staging tensor on a few threads, optionally
// rounding them to bf16 (round to nearest even; the transport format of feed.store_batches(region_dtype=bfloat16)).
// Why native: from Python the same gather is one numpy.take per worker thread, and every one of those calls takes and returns the GIL --
// next to a training loop that holds it for milliseconds at a time, a 16-thread gather of 151 MB took 9-19 ms per batch instead of the
// 2-3 ms the copies need (tools/feed_bench.py --store).  A ctypes call drops the GIL once for the whole batch.
#include <cstring>
#include <thread>
#include <vector>

#include "common.hpp"

namespace {
inline uint16_t bf16_rne(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);   // NaN stays a (quiet) NaN
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
}  // namespace

extern "C" int vqa_host_gather_rows(const float* store, long store_rows, long row_floats, const long* idx, int n, void* out,
                                    int out_bf16, int threads) {
  VQA_REQUIRE(store && idx && out, VQA_E_BADARG, "host_gather_rows: null pointer");
  VQA_REQUIRE(store_rows > 0 && row_floats > 0 && n >= 0, VQA_E_BADARG, "host_gather_rows: bad sizes");
  for (int r = 0; r < n; ++r)
    VQA_REQUIRE(idx[r] >= 0 && idx[r] < store_rows, VQA_E_BADARG, "host_gather_rows: index %ld outside [0, %ld)", idx[r], store_rows);
  if (n == 0) return VQA_OK;
  int nt = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
  if (nt > n) nt = n;
  auto work = [=](int lo, int hi) {
    for (int r = lo; r < hi; ++r) {
      const float* src = store + (size_t)idx[r] * row_floats;
      if (out_bf16) {
        uint16_t* dst = static_cast<uint16_t*>(out) + (size_t)r * row_floats;
        for (long k = 0; k < row_floats; ++k) dst[k] = bf16_rne(src[k]);
      } else {
        std::memcpy(static_cast<float*>(out) + (size_t)r * row_floats, src, (size_t)row_floats * 4);
      }
    }
  };
  const int per = (n + nt - 1) / nt;
  std::vector<std::thread> pool;
  pool.reserve(nt);
  for (int t = 1; t < nt; ++t) pool.emplace_back(work, t * per, (t + 1) * per < n ? (t + 1) * per : n);
  work(0, per < n ? per : n);
  for (auto& th : pool) th.join();
  return VQA_OK;
}
