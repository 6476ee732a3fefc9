// Version, per-thread error text and the process-wide option table of libvqa_mi355x.so.
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>

#include <climits>

#include "common.hpp"

namespace vqa {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}

// Option table.  Every tuning / diagnostic knob of the kernels (VQA_K3_FUSED_MIN_B, VQA_K2_BYTE_MASK, VQA_RELDG_TUNE ...)
// is looked up here and nowhere else.  The environment is read ONCE per name, at the first query, and the answer is
// kept: a later os.environ change cannot make the backward of an op pick another kernel form or mask layout than its
// forward did, and no launch calls getenv (which is not safe beside another thread's putenv).  Tests and tools switch
// a knob explicitly through vqa_set_option().
namespace {
struct OptionTable {
  std::mutex mu;
  std::unordered_set<std::string> interned;                     // node-stable storage: a value handed out is never touched again
  std::unordered_map<std::string, const std::string*> values;   // name -> interned value, nullptr = unset
  const std::string* intern(const char* v) { return v != nullptr ? &*interned.emplace(v).first : nullptr; }
};
OptionTable& options() {
  static OptionTable* t = new OptionTable();  // never destroyed: launches may outlive static destruction order
  return *t;
}
struct LaunchLog {
  int n;
  unsigned long long items[VQA_LAUNCH_LOG_CAP];
  const char* kernel[VQA_LAUNCH_LOG_CAP];   // the kernel expression as written at the launch site (string literal)
};
LaunchLog& launch_log() {
  static thread_local LaunchLog log = {0, {0}, {nullptr}};
  return log;
}
}  // namespace

const char* option(const char* name) {
  OptionTable& t = options();
  std::lock_guard<std::mutex> lock(t.mu);
  auto it = t.values.find(name);
  if (it == t.values.end()) it = t.values.emplace(name, t.intern(std::getenv(name))).first;
  return it->second != nullptr ? it->second->c_str() : nullptr;
}

void note_launch(const char* kernel, dim3 grid, dim3 block) {
  LaunchLog& log = launch_log();
  if (log.n < VQA_LAUNCH_LOG_CAP) {
    log.items[log.n] = (unsigned long long)grid.x * grid.y * grid.z * block.x * block.y * block.z;
    log.kernel[log.n] = kernel;
    ++log.n;
  } else if (log.n < INT_MAX) {   // the count saturates: a long-lived process that never resets the log launches > 2^31 kernels
    ++log.n;                      // in a few hours, and a wrapped counter would pass the bound check above with a negative index
  }
}

// Zero-fill as a plain kernel.  hipMemsetAsync must not be used in this library: captured into a hipGraph (memset
// node) it is correct on the first replay only -- from the second replay on the K1 / K3 backward accumulators came
// back wrong (ROCm 7.2, measured with tools/graph_memset_check.py), which silently corrupted every graph-replayed step.
__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned char* __restrict__ p, size_t head, size_t words,
                                                        size_t tail) {
  uint32_t* body = reinterpret_cast<uint32_t*>(p + head);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) body[i] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    for (size_t i = 0; i < head; ++i) p[i] = 0;
    for (size_t i = 0; i < tail; ++i) p[head + words * 4 + i] = 0;
  }
}

int zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return VQA_OK;
  unsigned char* p = static_cast<unsigned char*>(ptr);
  size_t head = (4 - (reinterpret_cast<uintptr_t>(p) & 3)) & 3;
  if (head > bytes) head = bytes;
  const size_t words = (bytes - head) / 4, tail = bytes - head - words * 4;
  size_t blocks = (words + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  VQA_LAUNCH(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, head, words, tail);
  return check_launch("zero_fill");
}
}  // namespace vqa

extern "C" int vqa_version(void) { return VQA_ABI_VERSION; }
extern "C" const char* vqa_source_hash(void) {
  return
#include "build/source_hash.inc"
      ;
}
extern "C" int vqa_set_option(const char* name, const char* value) {
  VQA_REQUIRE(name != nullptr && name[0] != 0, VQA_E_BADARG, "set_option: empty name");
  vqa::OptionTable& t = vqa::options();
  std::lock_guard<std::mutex> lock(t.mu);
  t.values[name] = t.intern(value);
  return VQA_OK;
}
extern "C" void vqa_launch_log_reset(void) { vqa::launch_log().n = 0; }
extern "C" int vqa_launch_log(unsigned long long* items, int capacity) {
  const vqa::LaunchLog& log = vqa::launch_log();
  const int n = log.n < 0 ? 0 : (log.n < VQA_LAUNCH_LOG_CAP ? log.n : VQA_LAUNCH_LOG_CAP);
  for (int i = 0; i < n && i < capacity; ++i) items[i] = log.items[i];
  return log.n < 0 ? 0 : log.n;   // (saturates at INT_MAX)
}
extern "C" const char* vqa_launch_log_kernel(int i) {
  const vqa::LaunchLog& log = vqa::launch_log();
  return (i >= 0 && i < log.n && i < VQA_LAUNCH_LOG_CAP) ? log.kernel[i] : nullptr;
}
extern "C" const char* vqa_last_error(void) { return vqa::error_buffer(); }
