// pack.hip -- layout kernels either side of the MFMA convolution.
//
//  * virnet_pack_weight : reference parameter layouts (nn.Conv2d OIHW, nn.ConvTranspose2d IOHW; networks/AttResUNet.py:43-46,
//    67,80; networks/DnCNN.py:22-29; networks/KNet.py:32-34,49) -> per-stage MFMA B-fragment images that conv_mfma.hip copies
//    linearly into LDS.
//  * virnet_pack_input  : NCHW image (+ conditioning) -> 16-channel NHWC pixel records, fusing nearest up-sampling
//    (networks/VIRNet.py:83,94), the bottom/right reflect pad (utils/util_net.py:20-25), sqrt of the variance map (VIRNet.py:44)
//    and the channel concat (networks/AttResUNet.py:153).  HBM-bound: one 64-B record written per pixel.
#include "common.h"
#include "../../include/virnet_hip.h"

namespace {

// Packed layout (matches conv_mfma.hip's load_w): [cb][chunk][tap][j][nr][lane][r], lane = lhi*32 + l31:
//   output channel n = cb*32*nrep + nr*32 + l31, input channel ci = chunk*16 + 8*j + 4*lhi + r.
// One half-step (cb, chunk, tap, j) is nrep contiguous 1-KB wave fragments.
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int kind, int cout, int cin, int ks,
                                   int cin_pad, int n_pad, int nrep, size_t total) {
  const int ntaps = ks * ks;
  const int nchunks = cin_pad / 16;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int rr = (int)(r & 3); r >>= 2;
    const int lane = (int)(r & 63); r >>= 6;
    const int nr = (int)(r % nrep); r /= nrep;
    const int j = (int)(r & 1); r >>= 1;
    const int t = (int)(r % ntaps); r /= ntaps;
    const int c = (int)(r % nchunks); r /= nchunks;
    const int cb = (int)r;
    const int ci = c * 16 + 8 * j + 4 * (lane >> 5) + rr;
    const int n = (cb * nrep + nr) * 32 + (lane & 31);
    float v = 0.f;
    if (kind == 0) {                                   // conv: rows = cout, contraction = cin
      if (ci < cin && n < cout) v = w[((size_t)n * cin + ci) * ntaps + t];
    } else if (kind == 1) {                            // transposed conv as 1x1 GEMM: rows = ab*cout + co
      const int ab = n / cout, co = n - ab * cout;
      if (ci < cin && ab < 4) v = w[((size_t)ci * cout + co) * 4 + ab];  // [Cin][Cout][a][b], ab = a*2+b
    } else if (kind == 2) {                            // dgrad of a 3x3 conv: rows = forward cin, contraction = forward cout, taps flipped
      if (ci < cout && n < cin) v = w[((size_t)ci * cin + n) * ntaps + (ntaps - 1 - t)];
    } else {                                           // dgrad of the transposed conv: rows = forward cin, contraction = ab*cout + co
      const int ab = ci / cout, co = ci - ab * cout;
      if (n < cin && ab < 4) v = w[((size_t)n * cout + co) * 4 + ab];
    }
    out[i] = v;
  }
}

__device__ __forceinline__ int reflect(int i, int n) { return i < n ? i : 2 * n - 2 - i; }

__global__ void pack_input_kernel(const virnet_pack_desc d) {
  const size_t npix = (size_t)d.n * d.hp * d.wp;
  const int HU = d.h * d.sf, WU = d.w * d.sf;  // size before the reflect pad
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * 4; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i & 3);         // which 16-B quarter of the record
    const size_t pix = i >> 2;
    const int x = (int)(pix % d.wp);
    const int y = (int)((pix / d.wp) % d.hp);
    const int n = (int)(pix / ((size_t)d.wp * d.hp));
    const int ry = reflect(y, HU), rx = reflect(x, WU);
    const bool dead = d.zero_pad && (y >= HU || x >= WU);
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int c = q * 4 + k;
      float val = 0.f;
      if (c < d.c0) {
        val = d.x[(((size_t)n * d.c0 + c) * d.h + ry / d.sf) * d.w + rx / d.sf];
      } else if ((c -= d.c0) < d.ev) {
        val = d.vec[(size_t)n * d.ev + c];
      } else if ((c -= d.ev) < d.em) {
        val = d.map[(((size_t)n * d.em + c) * d.mh + ry / d.msf) * d.mw + rx / d.msf];
        if (d.map_sqrt) val = sqrtf(val);
      }
      v[k] = dead ? 0.f : val;
    }
    reinterpret_cast<float4*>(d.out)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

__global__ void zero_stuff2_kernel(const float4* __restrict__ dy, float4* __restrict__ z, int h, int w, int c4, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % c4);
    size_t p = i / c4;
    const int x = (int)(p % (2 * w)); p /= 2 * w;
    const int y = (int)(p % (2 * h));
    const size_t n = p / (2 * h);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!((x | y) & 1)) v = dy[((n * h + (y >> 1)) * w + (x >> 1)) * c4 + q];
    z[i] = v;
  }
}

__global__ void space_to_depth2_kernel(const float4* __restrict__ dy, float4* __restrict__ out, int h, int w, int c4, size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % c4);
    size_t p = i / c4;
    const int ab = (int)(p & 3); p >>= 2;
    const int x = (int)(p % w); p /= w;
    const int y = (int)(p % h);
    const size_t n = p / h;
    out[i] = dy[((n * 2 * h + 2 * y + (ab >> 1)) * 2 * w + 2 * x + (ab & 1)) * c4 + q];
  }
}

__global__ void pack_input_backward_kernel(const float* __restrict__ drec, int crec, int chan, const float* __restrict__ map,
                                           float* __restrict__ dmap, int h, int w, int hp, int wp, int map_sqrt, int accumulate,
                                           size_t total) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % w), y = (int)((i / w) % h);
    const size_t n = i / ((size_t)w * h);
    // padded rows that read source row y: y itself and its mirror 2h-2-y when that lands inside [h, hp)
    const int ym = 2 * h - 2 - y, xm = 2 * w - 2 - x;
    const bool my = ym >= h && ym < hp, mx = xm >= w && xm < wp;
    const float* base = drec + (n * hp * (size_t)wp) * crec + chan;
    float g = base[((size_t)y * wp + x) * crec];
    if (my) g += base[((size_t)ym * wp + x) * crec];
    if (mx) g += base[((size_t)y * wp + xm) * crec];
    if (my && mx) g += base[((size_t)ym * wp + xm) * crec];
    if (map_sqrt) g *= 0.5f / sqrtf(map[i]);
    dmap[i] = accumulate ? dmap[i] + g : g;
  }
}

}  // namespace

static int grid_for(size_t total) { return (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256); }

extern "C" int virnet_zero_stuff2(const float* dy, float* z, int n, int h, int w, int c, void* stream) {
  VIRNET_REQUIRE(dy && z && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "virnet_zero_stuff2: bad arguments");
  const size_t total = (size_t)n * 2 * h * 2 * w * (c / 4);
  hipLaunchKernelGGL(zero_stuff2_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(dy), reinterpret_cast<float4*>(z), h, w, c / 4, total);
  return virnet::check_launch("zero_stuff2 launch");
}

extern "C" int virnet_space_to_depth2(const float* dy, float* out, int n, int h, int w, int c, void* stream) {
  VIRNET_REQUIRE(dy && out && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "virnet_space_to_depth2: bad arguments");
  const size_t total = (size_t)n * h * w * 4 * (c / 4);
  hipLaunchKernelGGL(space_to_depth2_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(dy), reinterpret_cast<float4*>(out), h, w, c / 4, total);
  return virnet::check_launch("space_to_depth2 launch");
}

extern "C" int virnet_pack_input_backward(const float* drec, int crec, int chan, const float* map, float* dmap, int n, int h, int w,
                                          int hp, int wp, int map_sqrt, int accumulate, void* stream) {
  VIRNET_REQUIRE(drec && dmap && n > 0 && h > 0 && w > 0 && hp >= h && wp >= w && hp - h < h && wp - w < w,
                 "virnet_pack_input_backward: bad shape %dx%d -> %dx%d", h, w, hp, wp);
  VIRNET_REQUIRE(chan >= 0 && chan < crec, "virnet_pack_input_backward: channel %d of %d", chan, crec);
  VIRNET_REQUIRE(!map_sqrt || map, "virnet_pack_input_backward: map_sqrt without map");
  const size_t total = (size_t)n * h * w;
  hipLaunchKernelGGL(pack_input_backward_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), drec, crec,
                     chan, map, dmap, h, w, hp, wp, map_sqrt, accumulate, total);
  return virnet::check_launch("pack_input_backward launch");
}

extern "C" size_t virnet_packed_weight_floats(int ks, int cin_pad, int n_pad) {
  return (size_t)ks * ks * cin_pad * n_pad;
}

extern "C" int virnet_pack_weight(const float* w, int kind, int cout, int cin, int ks, int cin_pad, int n_pad, int nrep,
                                  float* packed, void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_weight: NULL pointer");
  VIRNET_REQUIRE(kind >= 0 && kind <= 3, "virnet_pack_weight: kind=%d", kind);
  VIRNET_REQUIRE(nrep >= 1 && n_pad % (32 * nrep) == 0 && cin_pad % 16 == 0, "virnet_pack_weight: n_pad=%d nrep=%d cin_pad=%d", n_pad, nrep, cin_pad);
  const int rows = kind == 0 ? cout : kind == 1 ? 4 * cout : cin;          // GEMM rows the packing must cover
  const int contr = kind <= 1 ? cin : kind == 2 ? cout : 4 * cout;         // contraction channels
  VIRNET_REQUIRE(n_pad >= rows && cin_pad >= contr, "virnet_pack_weight: n_pad=%d / cin_pad=%d do not cover %d rows x %d channels (kind %d)",
                 n_pad, cin_pad, rows, contr, kind);
  VIRNET_REQUIRE((kind == 1 || kind == 3) ? ks == 2 : (ks == 3 || ks == 1), "virnet_pack_weight: ks=%d for kind %d", ks, kind);
  const int gemm_ks = (kind == 1 || kind == 3) ? 1 : ks;  // the transposed conv (and its dgrad) run as pointwise GEMMs
  const size_t total = virnet_packed_weight_floats(gemm_ks, cin_pad, n_pad);
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed, kind, cout, cin,
                     gemm_ks, cin_pad, n_pad, nrep, total);
  return virnet::check_launch("pack_weight launch");
}

extern "C" int virnet_pack_input(const virnet_pack_desc* d, void* stream) {
  VIRNET_REQUIRE(d && d->x && d->out, "virnet_pack_input: NULL pointer");
  VIRNET_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->sf >= 1, "virnet_pack_input: bad shape n=%d h=%d w=%d sf=%d", d->n, d->h,
                 d->w, d->sf);
  VIRNET_REQUIRE(d->c0 >= 1 && d->ev >= 0 && d->em >= 0 && d->c0 + d->ev + d->em <= 16,
                 "virnet_pack_input: %d+%d+%d channels do not fit a 16-channel record", d->c0, d->ev, d->em);
  VIRNET_REQUIRE(d->ev == 0 || d->vec, "virnet_pack_input: ev=%d without vec", d->ev);
  VIRNET_REQUIRE(d->em == 0 || (d->map && d->msf >= 1), "virnet_pack_input: em=%d without map/msf", d->em);
  const int HU = d->h * d->sf, WU = d->w * d->sf;
  // F.pad(mode='reflect') demands pad < dim (utils/util_net.py:24); mirror the reference's error.
  VIRNET_REQUIRE(d->hp >= HU && d->wp >= WU && d->hp - HU < HU && d->wp - WU < WU,
                 "virnet_pack_input: reflect pad %dx%d -> %dx%d needs pad < dim", HU, WU, d->hp, d->wp);
  VIRNET_REQUIRE(d->em == 0 || (d->mh * d->msf == HU && d->mw * d->msf == WU),
                 "virnet_pack_input: map %dx%d x%d does not match image %dx%d", d->mh, d->mw, d->msf, HU, WU);
  const size_t total = (size_t)d->n * d->hp * d->wp * 4;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_input_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), *d);
  return virnet::check_launch("pack_input launch");
}
