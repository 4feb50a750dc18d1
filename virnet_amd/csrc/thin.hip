// thin.hip -- 3x3 convolution to a FEW output channels (<= 4) with planar NCHW store: AttResUNet.tail + crop + `+ x_in`
// (networks/AttResUNet.py:139,173), DnCNN.conv_last with exp(clamp(.)) (networks/DnCNN.py:29,41; networks/VIRNet.py:43),
// KernelNet.tail conv (networks/KNet.py:49).
//
// These layers are 0.1 % of the FLOPs but read a full-width NHWC tensor: HBM/LDS-bound, not MFMA work (on the matrix cores the
// 3 real channels would ride in a 32-channel block, 10x wasted).  One thread = one output pixel, 256 threads = an 8x32 tile.
// The halo tile is staged through LDS in 16-channel chunks exactly like conv_mfma.hip (64-B pixel records, XOR slot swizzle,
// register-staged double buffer, zero fill = the conv's zero padding); weights are wave-uniform and come through the scalar
// cache ([chunk][tap][16 ch][4 co] floats), so the inner loop is ds_read_b128 + v_fma with an SGPR operand.
#include "common.h"
#include "../../include/virnet_hip.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int swz(int p) { return (p >> 2) & 3; }

constexpr int TH = 8, TW = 32, IH = TH + 2, IW = TW + 2, NPIX = IH * IW, NPIECE = NPIX * 4, PPT = (NPIECE + 255) / 256;
constexpr int IN_BYTES = NPIX * 64;

__global__ __launch_bounds__(256) void conv3x3_thin_kernel(const virnet_thin_desc d, int ntx, int nty) {
  __shared__ __attribute__((aligned(16))) char smem[2 * IN_BYTES];
  const int tile = blockIdx.x;
  const int tx = tile % ntx, ty = (tile / ntx) % nty, img = tile / (ntx * nty);
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int tid = threadIdx.x;
  const int px = tid & 31, py = tid >> 5;           // this thread's pixel inside the tile
  const float* const ximg = d.x + (size_t)img * d.h * d.w * d.c;
  const int nchunks = d.c >> 4;

  auto in_addr = [&](int k, int chunk, int& off, int& dst, bool& inb) {
    const int qq = k * 256 + tid;
    const bool has = qq < NPIECE;
    const int qc = has ? qq : 0;
    const int p = qc >> 2, s = qc & 3;
    const int iy = p / IW, ix = p - iy * IW;
    const int gy = oy0 - 1 + iy, gx = ox0 - 1 + ix;
    inb = has && (unsigned)gy < (unsigned)d.h && (unsigned)gx < (unsigned)d.w;
    const int gyc = min(max(gy, 0), d.h - 1), gxc = min(max(gx, 0), d.w - 1);
    off = (gyc * d.w + gxc) * d.c + chunk * 16 + s * 4;
    dst = has ? p * 64 + ((s ^ swz(p)) << 4) : -1;
  };

  f32x4 ireg[PPT];
  int idst[PPT];
  bool iinb[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    int off;
    in_addr(k, 0, off, idst[k], iinb[k]);
    ireg[k] = *reinterpret_cast<const f32x4*>(ximg + off);
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k)
    if (idst[k] >= 0) *reinterpret_cast<f32x4*>(smem + idst[k]) = iinb[k] ? ireg[k] : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nchunks; ++c) {
    const char* const cur = smem + (c & 1) * IN_BYTES;
    char* const nxt = smem + ((c + 1) & 1) * IN_BYTES;
    const bool more = c + 1 < nchunks;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {                 // prefetch the next chunk (clamped re-read on the last one)
      int off;
      in_addr(k, more ? c + 1 : c, off, idst[k], iinb[k]);
      ireg[k] = *reinterpret_cast<const f32x4*>(ximg + off);
    }
    const float* const wc = d.wpack + (size_t)c * 9 * 64;      // [tap][16 ch][4 co], wave-uniform -> scalar loads
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int p = (py + t / 3) * IW + px + t % 3;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(cur + p * 64 + ((s ^ swz(p)) << 4));
        const float* const w = wc + t * 64 + s * 16;
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int co = 0; co < 4; ++co) acc[co] = fmaf(v[e], w[e * 4 + co], acc[co]);
      }
    }
    if (more) {
#pragma unroll
      for (int k = 0; k < PPT; ++k)
        if (idst[k] >= 0) *reinterpret_cast<f32x4*>(nxt + idst[k]) = iinb[k] ? ireg[k] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
  }

  const int oy = oy0 + py, ox = ox0 + px;
  if (oy < d.crop_h && ox < d.crop_w) {
    const size_t plane = (size_t)d.crop_h * d.crop_w;
    for (int co = 0; co < d.cout; ++co) {
      const size_t o = ((size_t)img * d.cout + co) * plane + (size_t)oy * d.crop_w + ox;
      float v = acc[co] + (d.bias ? d.bias[co] : 0.f);
      if (d.op == VIRNET_NCHW_ADD) {
        if (d.res_sf > 1) {
          const int rw = d.crop_w / d.res_sf;
          v += d.res[((size_t)img * d.cout + co) * (size_t)(d.crop_h / d.res_sf) * rw + (size_t)(oy / d.res_sf) * rw + ox / d.res_sf];
        } else {
          v += d.res[o];
        }
      } else if (d.op == VIRNET_NCHW_EXPCLAMP) {
        v = expf(fminf(fmaxf(v, d.clamp_lo), d.clamp_hi));
      }
      d.y[o] = v;
    }
  }
}

// OIHW [cout][c][3][3] -> [c/16][tap][16][4] with zero-padded output channels
__global__ void pack_thin_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int c, int total) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int co = i & 3, ch16 = (i >> 2) & 15, t = (i >> 6) % 9, chunk = i / (64 * 9);
    const int ci = chunk * 16 + ch16;
    out[i] = (co < cout && ci < c) ? w[((size_t)co * c + ci) * 9 + t] : 0.f;
  }
}

}  // namespace

extern "C" size_t virnet_thin_weight_floats(int c_pad) { return (size_t)(c_pad / 16) * 9 * 64; }

extern "C" int virnet_pack_thin_weight(const float* w, int cout, int c, int c_pad, float* packed, void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_thin_weight: NULL pointer");
  VIRNET_REQUIRE(cout >= 1 && cout <= 4, "virnet_pack_thin_weight: cout=%d (1..4)", cout);
  VIRNET_REQUIRE(c_pad % 16 == 0 && c_pad >= c && c >= 1, "virnet_pack_thin_weight: c=%d c_pad=%d", c, c_pad);
  const int total = (int)virnet_thin_weight_floats(c_pad);
  hipLaunchKernelGGL(pack_thin_weight_kernel, dim3((total + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed,
                     cout, c, total);
  return virnet::check_launch("pack_thin_weight launch");
}

extern "C" int virnet_conv3x3_thin(const virnet_thin_desc* d, void* stream) {
  VIRNET_REQUIRE(d && d->x && d->wpack && d->y, "virnet_conv3x3_thin: NULL pointer");
  VIRNET_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv3x3_thin: empty input n=%d h=%d w=%d", d->n, d->h, d->w);
  VIRNET_REQUIRE(d->c > 0 && d->c % 16 == 0, "virnet_conv3x3_thin: c=%d is not a multiple of 16", d->c);
  VIRNET_REQUIRE(d->cout >= 1 && d->cout <= 4, "virnet_conv3x3_thin: cout=%d (1..4)", d->cout);
  VIRNET_REQUIRE(d->crop_h >= 1 && d->crop_h <= d->h && d->crop_w >= 1 && d->crop_w <= d->w,
                 "virnet_conv3x3_thin: crop %dx%d outside output %dx%d", d->crop_h, d->crop_w, d->h, d->w);
  VIRNET_REQUIRE(d->op >= VIRNET_NCHW_PLAIN && d->op <= VIRNET_NCHW_EXPCLAMP, "virnet_conv3x3_thin: op=%d", d->op);
  VIRNET_REQUIRE(d->op != VIRNET_NCHW_ADD || d->res, "virnet_conv3x3_thin: VIRNET_NCHW_ADD without res");
  VIRNET_REQUIRE(d->res_sf <= 1 || (d->crop_h % d->res_sf == 0 && d->crop_w % d->res_sf == 0),
                 "virnet_conv3x3_thin: crop %dx%d is not a multiple of res_sf=%d", d->crop_h, d->crop_w, d->res_sf);
  const int ntx = (d->crop_w + TW - 1) / TW, nty = (d->crop_h + TH - 1) / TH;   // tiles beyond the crop store nothing
  hipLaunchKernelGGL(conv3x3_thin_kernel, dim3((unsigned)(d->n * ntx * nty)), dim3(256), 0, static_cast<hipStream_t>(stream), *d,
                     ntx, nty);
  return virnet::check_launch("conv3x3_thin launch");
}
