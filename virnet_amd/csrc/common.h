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

}  // namespace virnet

#define VIRNET_REQUIRE(cond, ...)                         \
  do {                                                    \
    if (!(cond)) return virnet::set_error(__VA_ARGS__);   \
  } while (0)
