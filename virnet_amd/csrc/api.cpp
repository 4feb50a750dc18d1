// api.cpp -- process-level pieces of the C ABI (version, error slot, device probe).
#include "common.h"
#include <cstdlib>
namespace virnet { int* range_flag_ptr(); int store_nt_for(size_t bytes); }
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

// (per host THREAD: forwards running in several threads, each on its own stream, keep separate flags -- engine.py registers one per
// (device, thread) on first use)
static thread_local int* g_range_flag[64] = {nullptr};

int* range_flag_ptr() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return nullptr;
  return g_range_flag[dev];
}

int store_nt_for(size_t bytes) {
  const char* e = getenv("VIRNET_NT_STORE_MB");       // (read per launch: A/B runs and tests flip it inside one process)
  const long limit = e ? atol(e) : 128;
  return limit > 0 && bytes > (size_t)limit * 1048576u;
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

extern "C" int virnet_set_range_flag(int* device_flag) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return virnet::set_error("virnet_set_range_flag: hipGetDevice: %s", hipGetErrorString(e));
  if (dev < 0 || dev > 63) return virnet::set_error("virnet_set_range_flag: device %d out of range", dev);
  virnet::g_range_flag[dev] = device_flag;
  return 0;
}
