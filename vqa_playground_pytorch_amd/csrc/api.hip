// Version and per-thread error text of libvqa_mi355x.so.
#include "common.hpp"

namespace vqa {
char* error_buffer() {
  static thread_local char buf[512] = {0};
  return buf;
}
}  // namespace vqa

extern "C" int vqa_version(void) { return VQA_ABI_VERSION; }
extern "C" const char* vqa_last_error(void) { return vqa::error_buffer(); }
