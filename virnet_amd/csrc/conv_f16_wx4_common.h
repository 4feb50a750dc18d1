// conv_f16_wx4_common.h -- device helpers shared by the two tile forms of the Winograd-along-x split-fp16 convolution
// (conv_f16_wx4.hip: 16 x 32-pixel tiles, 8 waves, one workgroup per CU; conv_f16_wx4h.hip: 8 x 32-pixel tiles, 4 waves, two per CU).
#pragma once
#include "conv_f16_common.h"
#include <type_traits>

namespace virnet {

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

template <int J>
__device__ __forceinline__ f32x4 wx4_pos(const f32x4 (&d)[6]) {
  // rows of BT (Lavin & Gray, F(4,3), points 0, +-1, +-2, inf)
  if constexpr (J == 0) return 4.f * d[0] - 5.f * d[2] + d[4];
  if constexpr (J == 1) return (d[4] - 4.f * d[2]) + (d[3] - 4.f * d[1]);
  if constexpr (J == 2) return (d[4] - 4.f * d[2]) - (d[3] - 4.f * d[1]);
  if constexpr (J == 3) return (d[4] - d[2]) + 2.f * (d[3] - d[1]);
  if constexpr (J == 4) return (d[4] - d[2]) - 2.f * (d[3] - d[1]);
  return 4.f * d[1] - 5.f * d[3] + d[5];
}

// the same row of BT on six scalars (conv_wx4h stages its halo rows one (row, x-tile, channel) value pair per thread)
template <int J>
__device__ __forceinline__ float wx4_pos_s(const float (&d)[6]) {
  if constexpr (J == 0) return fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
  if constexpr (J == 1) return fmaf(-4.f, d[2], d[4]) + fmaf(-4.f, d[1], d[3]);
  if constexpr (J == 2) return fmaf(-4.f, d[2], d[4]) - fmaf(-4.f, d[1], d[3]);
  if constexpr (J == 3) return fmaf(2.f, d[3] - d[1], d[4] - d[2]);
  if constexpr (J == 4) return fmaf(-2.f, d[3] - d[1], d[4] - d[2]);
  return fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
}

// max of two floats as the bare instruction: fmaxf() on a value that comes straight from a load first copies it through a
// canonicalising v_max x, x (IEEE sNaN quieting), one VALU op per staged value
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// two fp32 -> packed fp16x2, round to nearest even (v_cvt_pk_f16_f32)
__device__ __forceinline__ unsigned cvtpk(float a, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 r = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, r);
}
// v - float(half HALF of hpk) in ONE VALU operation (v_fma_mix_f32 reads the fp16 half directly); exact like the subtraction
template <int HALF>
__device__ __forceinline__ float subhi(float v, unsigned hpk) {
  float r;
  if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
  return r;
}

// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. would wait for global loads in flight
__device__ __forceinline__ void wx_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct WxItem {
  unsigned voff;     // byte offset of the item's first input pixel (row, 4*xtile - 1) from the image base; may wrap (the load is masked)
  unsigned inb;      // bit b: pixel b lies inside the image
  int dst;           // byte offset of this item inside a V plane
};

// BT rows as coefficient vectors (the halo rows are staged one value per thread: v = sum_b c[b] * d[b])
__device__ __forceinline__ float wx4_coef(int j, int b) {
  constexpr float BT[6][6] = {{4.f, 0.f, -5.f, 0.f, 1.f, 0.f}, {0.f, -4.f, -4.f, 1.f, 1.f, 0.f}, {0.f, 4.f, -4.f, -1.f, 1.f, 0.f},
                              {0.f, -2.f, -1.f, 2.f, 1.f, 0.f}, {0.f, 2.f, -1.f, -2.f, 1.f, 0.f}, {0.f, 4.f, 0.f, -5.f, 0.f, 1.f}};
  return BT[j][b];
}

#define WX_I(n) std::integral_constant<int, n>{}

// launcher of the 8 x 32-pixel form (conv_f16_wx4h.hip); `k` filled as for conv_wx4's own launch
int launch_wx4h(FArgs k, int nrep, int epi, int pre, hipStream_t st);
int launch_wx4h_emit(FArgs k, int nrep, int epi, int pre, hipStream_t st);   // TE = 1 instantiations (pre 0 / 1, epi 0..3)

// the persistent, overlapped 16-row form (conv_f16_wx4p.hip; round 6): what it serves, and its launcher (`k` filled as for conv_wx4's own launch)
bool wx4p_serves(const FArgs& k, int nrep, int epi, int pre);
int launch_wx4p(FArgs k, int nrep, int epi, int pre, int n_cu, hipStream_t st);

}  // namespace virnet
