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
    if (ci < cin) {
      if (kind == 0) {
        if (n < cout) v = w[((size_t)n * cin + ci) * ntaps + t];
      } else {
        const int ab = n / cout, co = n - ab * cout;
        if (ab < 4) v = w[((size_t)ci * cout + co) * 4 + ab];  // [Cin][Cout][a][b], ab = a*2+b
      }
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
      v[k] = val;
    }
    reinterpret_cast<float4*>(d.out)[i] = make_float4(v[0], v[1], v[2], v[3]);
  }
}

}  // namespace

extern "C" size_t virnet_packed_weight_floats(int ks, int cin_pad, int n_pad) {
  return (size_t)ks * ks * cin_pad * n_pad;
}

extern "C" int virnet_pack_weight(const float* w, int kind, int cout, int cin, int ks, int cin_pad, int n_pad, int nrep,
                                  float* packed, void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_weight: NULL pointer");
  VIRNET_REQUIRE(kind == 0 || kind == 1, "virnet_pack_weight: kind=%d", kind);
  VIRNET_REQUIRE(cin_pad % 16 == 0 && cin_pad >= cin, "virnet_pack_weight: cin_pad=%d for cin=%d", cin_pad, cin);
  VIRNET_REQUIRE(nrep >= 1 && n_pad % (32 * nrep) == 0, "virnet_pack_weight: n_pad=%d nrep=%d", n_pad, nrep);
  VIRNET_REQUIRE(kind == 0 ? n_pad >= cout : (ks == 2 && n_pad == 4 * cout),
                 "virnet_pack_weight: n_pad=%d does not cover cout=%d (kind %d, ks %d)", n_pad, cout, kind, ks);
  const int gemm_ks = (kind == 1) ? 1 : ks;  // the transposed conv runs as a pointwise GEMM
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
