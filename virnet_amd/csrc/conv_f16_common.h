// conv_f16_common.h -- argument block and small device helpers shared by the split-fp16 convolution kernels
// (conv_f16.hip: stride 1; conv_f16_s2.hip: stride 2).  See conv_f16.hip for the arithmetic.
#pragma once
#include "common.h"
#include "../../include/virnet_hip.h"

namespace virnet {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)
// -DVIRNET_F16_TIMING: wave 0 of every workgroup logs s_memtime at start / after the prologue / after the K loop / at exit plus
// its HW_ID and XCC_ID into the buffer given to virnet_debug_timing_buffer (tools/f16_timeline.py reads it).
#ifdef VIRNET_F16_TIMING
#define TSTAMP(i) do { if (a.tlog && tid == 0) a.tlog[(size_t)blockIdx.x * 8 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TSTAMP(i) do { } while (0)
#endif

struct FArgs {
  const float* x;
  const char* wimg;        // [slab][chunk][tap'][hi|lo][lane][16 B]
  const float* inv_scale;  // [NP]
  const float* bias;
  const float* res;
  const float* mul;
  const float* add;
  const float* in_mul;
  const float* in_add;
  const float* mask;
  float* y_raw;
  float* y_act;
  int N, H, W, Cin;
  int OH, OW;              // output size (stride-2 form; the stride-1 kernel uses H, W)
  int NP, cout;            // NP: GEMM rows (output channels) covered by THIS launch, starting at slab `slab_base`
  int slab_base;
  int ntx, nty, ntiles, tiles_per_xcd;
  unsigned mg_ncb, mg_ntx, mg_tpi;   // floor(2^32 / d) + 1 for d = channel blocks per tile, tiles per row, tiles per image (conv_wx4: fast_div)
  int in_act;
  int nchw_op, crop_h, crop_w, res_sf;      // EPI 5 (planar store)
  float in_slope, mask_slope, slope, clamp_lo, clamp_hi;
  long long* tlog;
  int rev;                 // 1: tiles are walked in reverse order (conv_wx4 / conv_wx4h: alternating launches, see virnet_conv_wx4)
  int store_nt;            // 1: the stored tensor is larger than the Infinity Cache and is read back only after it has left it -- its stores carry the
                           // non-temporal policy bit (measured in J per launch, profiles/r05_probes.md 6: -1.5 % on a conv1-type 96-channel launch)
  int* range_flag;         // sticky device flag (virnet_set_range_flag) set when a staged operand leaves fp16's range, or NULL
  // T emission (training step; kernels instantiated with TE = 1): besides the NHWC tensor the epilogue writes the channel-major
  // fp16 hi|lo (or bf16) image wgrad_f16.hip contracts over -- T[n][H+2][t_cb][t_npl][t_nseg][32 ch][8 px], pixel x at index x + 8 --
  // of the STORED value (t_act: of lrelu(stored, t_slope), the staging transform of the conv that will consume it), and per
  // (workgroup, wave) channel sums of the stored value (the bias gradient's partial sums: t_col[(cb * t_nblk + tile * waves + wave) * 32 + ch]).
  char* t_out;
  float* t_col;
  int t_cb, t_npl, t_nseg, t_nblk, t_act;
  float t_slope;
  // Entry form (conv_f16.hip, ENT = 1): the image entry of virnet_pack_input folded into the conv's staging -- the 16-channel record
  // [image | per-image vector | per-pixel map | 0] of a pixel is gathered from the NCHW sources instead of being read from a packed tensor
  // (H, W above are the padded size hp x wp)
  virnet_pack_desc ent;
};

// One pixel's first 8 record channels as virnet_pack_input writes them (pack.hip: nearest up-sampling, bottom / right reflect pad, sqrt of
// the variance map, channel concat); (y, x) inside the padded image.  c0 + ev + em <= 8.
__device__ __forceinline__ void entry_pixel(const virnet_pack_desc& d, int n, int y, int x, f32x4& r0, f32x4& r1) {
  const int HU = d.h * d.sf, WU = d.w * d.sf;
  const int ry = y < HU ? y : 2 * HU - 2 - y, rx = x < WU ? x : 2 * WU - 2 - x;
  const bool dead = d.zero_pad && (y >= HU || x >= WU);
  // image-uniform bases (scalar 64-bit arithmetic), 32-bit offsets inside one image's planes
  const float* const xi = d.x + (size_t)n * d.c0 * d.h * d.w;
  const float* const mi = d.em ? d.map + (size_t)n * d.em * d.mh * d.mw : nullptr;
  const float* const vi = d.ev ? d.vec + (size_t)n * d.ev : nullptr;
  const unsigned xpix = (unsigned)((d.sf == 1 ? ry : ry / d.sf) * d.w + (d.sf == 1 ? rx : rx / d.sf)), xplane = (unsigned)(d.h * d.w);
  const unsigned mpix = d.em ? (unsigned)((d.msf == 1 ? ry : ry / d.msf) * d.mw + (d.msf == 1 ? rx : rx / d.msf)) : 0u, mplane = (unsigned)(d.mh * d.mw);
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    int c = k;
    float val = 0.f;
    if (c < d.c0) {
      val = xi[c * xplane + xpix];
    } else if ((c -= d.c0) < d.ev) {
      val = vi[c];
    } else if ((c -= d.ev) < d.em) {
      val = mi[c * mplane + mpix];
      if (d.map_sqrt) val = sqrtf(val);
    }
    v[k] = dead ? 0.f : val;
  }
  r0 = f32x4{v[0], v[1], v[2], v[3]};
  r1 = f32x4{v[4], v[5], v[6], v[7]};
}

// segments per row of T (wgrad_f16.hip: t_nseg -- the pixels rounded up to whole 64-pixel steps, 32 for narrow images, + one pad segment each side)
inline int t_nseg_of(int w) { return w <= 32 ? 6 : 8 * ((w + 63) / 64) + 2; }

// fills the T-emission fields of a kernel argument block from the public descriptor pair
inline void t_emit_args(FArgs& k, const virnet_t_emit* te, int w, int cout, int nblk) {
  k.t_out = static_cast<char*>(te->t_out); k.t_col = te->col;
  k.t_cb = cout / 32; k.t_npl = 2; k.t_nseg = t_nseg_of(w); k.t_nblk = nblk;      // (a bf16 image keeps the two-plane stride and leaves plane 1 unused: chsplit_kernel<1>)
  k.t_act = te->act; k.t_slope = te->slope;
}

// 8 fp32 -> the 16-byte T unit(s) of wgrad_f16.hip: fp16 hi / lo (chsplit_kernel's split) or one bf16 plane
__device__ __forceinline__ void t_units(const float (&v)[8], bool bf, u32x4& hi, u32x4& lo) {
  unsigned short h[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    if (bf) {
      const __bf16 hb = (__bf16)v[e];
      h[e] = __builtin_bit_cast(unsigned short, hb);
      l[e] = 0;
    } else {
      const _Float16 hh = (_Float16)v[e];
      const _Float16 ll = (_Float16)(v[e] - (float)hh);
      h[e] = __builtin_bit_cast(unsigned short, hh);
      l[e] = __builtin_bit_cast(unsigned short, ll);
    }
  }
  hi = u32x4{(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16), (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16)};
  lo = u32x4{(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16), (unsigned)l[4] | ((unsigned)l[5] << 16), (unsigned)l[6] | ((unsigned)l[7] << 16)};
}

// One 1-KB piece (64 lanes x 16 B) global -> LDS without a register round trip, as a MUBUF instruction (buffer_load_dwordx4 ... lds):
// descriptor over the weight image, lane offset in a register, piece offset scalar.  Against global_load_lds (a FLAT instruction with a
// 64-bit address per lane) it saves the per-piece address arithmetic, and hipcc's wait-count pass does not treat it as a pending FLAT
// operation (which turns every wait it inserts into vmcnt(0): profiles/r03_probes.md).
template <class Rsrc>
__device__ __forceinline__ void lds_dma16(Rsrc rs, char* lds_dst, int lane_off, int piece_off) {
#if defined(__HIP_DEVICE_COMPILE__)                // (the host pass drops a kernel's stub without a diagnostic when it meets this builtin)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_dst, 16, lane_off, piece_off, 0, 0);
#endif
}

// n / d for a launch-invariant d with the host's magic number m = floor(2^32 / d) + 1: exact while n * d < 2^32 (the launchers check).
// A runtime division on the scalar unit is a reciprocal on the VECTOR unit and back (~150 cycles of latency each).
// (d = 1 has no 32-bit magic number: 0 stands for it)
__device__ __forceinline__ int fast_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }
inline unsigned div_magic(int d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / (unsigned long long)d) + 1u; }

__device__ __forceinline__ f32x4 lrelu4(f32x4 u, float s) {
  const f32x4 t = u * s;
  return f32x4{fmaxf(u.x, t.x), fmaxf(u.y, t.y), fmaxf(u.z, t.z), fmaxf(u.w, t.w)};
}

// Range guard of the split-fp16 kernels: an operand of magnitude >= 65520 rounds to an fp16 infinity and the result turns Inf / NaN --
// loud in the tensors, but clamped away by exp(clamp(.)) / tanh epilogues (VIRNet.py:43, KNet.py:56-58).  Every kernel keeps the largest
// staged magnitude per thread and raises the sticky flag once, after its K loop; the host side re-runs the forward in the fp32 form.
__device__ __forceinline__ void range_note(float& amax, const f32x4& a, const f32x4& b) {
  amax = fmaxf(fmaxf(amax, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fmaxf(fabsf(a.z), fabsf(a.w)), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
}
// (a plain store through a GLOBAL-address-space pointer: an atomic on the generic pointer is a FLAT instruction, and one pending FLAT
// operation -- even on a path never taken -- makes hipcc turn the next wait of the epilogue into vmcnt(0) lgkmcnt(0))
__device__ __forceinline__ void range_report(int* flag, float amax) {
  if (flag != nullptr && amax >= 65520.f) *(__attribute__((address_space(1))) int*)flag = 1;
}

// v = hi + lo in fp16 (round to nearest even both times)
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, h8& hi, h8& lo) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (_Float16)v[e];
    lo[e] = (_Float16)(v[e] - (float)hi[e]);
  }
}


// 8 fp32 -> 8 bf16 (round to nearest even)
__device__ __forceinline__ b8 to_bf16x8(const f32x4& a, const f32x4& b) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  b8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (__bf16)v[e];
  return r;
}

// the flag registered for the current device (api.cpp), or NULL
int* range_flag_ptr();
// store policy of an output of `bytes` bytes: non-temporal above VIRNET_NT_STORE_MB (default 128: the 96- and 192-channel levels of a 32 x 256^2 step; 0 = never)
int store_nt_for(size_t bytes);

// stride-2 form (conv_f16_s2.hip): `k` filled as for the stride-1 launch, H/W = INPUT size, OH/OW = output size; nb = 32-channel slabs
int launch_f16_s2(FArgs k, int nb, hipStream_t st);
// transposed 2x2/s2 conv as a pointwise GEMM + depth-to-space store (conv_f16_pw.hip)
int launch_f16_convt(FArgs k, int cin_real, hipStream_t st);

}  // namespace virnet
