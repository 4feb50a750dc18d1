// api.cpp -- process-level pieces of the C ABI (version, error slot, device probe).
#include "common.h"
#include "../../include/virnet_hip.h"

namespace virnet {

char* error_slot() {
  static thread_local char buf[512] = {0};
  return buf;
}

int set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_slot(), 512, fmt, ap);
  va_end(ap);
  return 1;
}

}  // namespace virnet

extern "C" int virnet_abi_version(void) { return VIRNET_ABI_VERSION; }

extern "C" const char* virnet_last_error(void) { return virnet::error_slot(); }

extern "C" int virnet_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    virnet::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return -1;
  }
  return n;
}
