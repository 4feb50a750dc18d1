// Shared host-side helpers for the C-ABI translation units (error slot, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>

namespace virnet {

// One error slot per host thread; virnet_last_error() reads it.
char* error_slot();
int set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error("%s: %s", what, hipGetErrorString(e));
  return 0;
}

// Per-device "already configured" flag for a kernel's function attributes (a process may drive several GPUs in turn).
inline bool first_use_on_device(unsigned long long& mask) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
  const unsigned long long bit = 1ull << dev;
  if (mask & bit) return false;
  mask |= bit;
  return true;
}

}  // namespace virnet

#define VIRNET_REQUIRE(cond, ...)                         \
  do {                                                    \
    if (!(cond)) return virnet::set_error(__VA_ARGS__);   \
  } while (0)
