// wino_args.h -- argument block and small helpers shared by the Winograd kernel forms (wino.hip, wino_row.hip).
#pragma once
#include "common.h"
#include "../../include/virnet_hip.h"

namespace virnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WArgs {
  const float* x;
  const float* up;       // packed U: [slab][chunk][pos 16][half 2][32 cout][2]
  const float* bias;
  const float* res;
  const float* mul;
  const float* add;
  const float* in_mul;
  const float* in_add;
  const float* mask;
  float* y_raw;
  float* y_act;
  int N, H, W, Cin, Cout;
  int nux, nuy, nunits, units_per_xcd, n64, n32;
  int in_act;
  float in_slope, mask_slope, slope;
};

__device__ __forceinline__ f32x4 wino_lrelu4(f32x4 u, float s) {
  const f32x4 t = u * s;
  return f32x4{fmaxf(u.x, t.x), fmaxf(u.y, t.y), fmaxf(u.z, t.z), fmaxf(u.w, t.w)};
}

// Row-split kernel form (wino_row.hip); `sft` = per-(image, channel) scale/shift on the staged input.
int launch_wino_row(WArgs k, hipStream_t st, bool sft);

}  // namespace virnet
