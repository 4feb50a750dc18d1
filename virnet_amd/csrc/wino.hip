// wino.hip -- the stride-1 3x3 convolution in Winograd F(2x2,3x3) form on the CDNA4 matrix cores (gfx950), fp32: weight packing
// and the C-ABI entry points.  The kernel itself is in wino_row.hip.
//
// Same call sites as conv_mfma.hip's KS=3,S=1 instantiations (AttResBlock.conv1/conv2, networks/AttResUNet.py:43,46,55,58;
// DnCNN mid convs, networks/DnCNN.py:25-28) and their input-gradient GEMMs: Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A with
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. 16 multiplies per 2x2 output tile and channel pair instead of 36.  The fp32 matrix pipe is the bound of this path
// (DESIGN.md 5), so the 2.25x cut in MFMA work is the lever; in fp32 the transform's rounding error is at the level of the direct
// form's own re-association error (measured against an fp64 run of the whole network: 1.0e-5 vs 0.9e-5 max-abs).
//
// The 16 transform positions are 16 independent GEMMs  M_p[cout][tile] = sum_ci U_p[cout][ci] * V_p[ci][tile];  U = G g G^T is
// packed here once per parameter update as [cout/32][cin/4][pos 16][k-half 2][32 cout][2] -- per 4-channel chunk exactly the LDS
// image the kernel streams in by DMA.
#include "wino_args.h"
#include <cstdlib>
#include <type_traits>

namespace {

using namespace virnet;

// U = G g G^T of every (output channel, input channel) pair, in the chunk-stage layout the kernel copies linearly.
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ out, int dgrad, int cout, int cin, int cin_pad,
                                 int n_pad) {
  const int nch = cin_pad >> 2;
  const long total = (long)(n_pad >> 5) * nch * 2048;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int s = idx & 1, col = (idx >> 1) & 31, half = (idx >> 6) & 1, pos = (idx >> 7) & 15;
  const long sc = idx >> 11;
  const int chunk = (int)(sc % nch), slab = (int)(sc / nch);
  const int co = slab * 32 + col, ci = chunk * 4 + half * 2 + s;
  // GEMM extents: forward (cout, cin); dgrad (cin_fwd, cout_fwd) with flipped taps
  const int rows = dgrad ? cin : cout, ks = dgrad ? cout : cin;
  float u = 0.f;
  if (co < rows && ci < ks) {
    double g[3][3];
    for (int ky = 0; ky < 3; ++ky)
      for (int kx = 0; kx < 3; ++kx)
        g[ky][kx] = dgrad ? (double)w[(((size_t)ci * cin + co) * 3 + (2 - ky)) * 3 + (2 - kx)]
                          : (double)w[(((size_t)co * cin + ci) * 3 + ky) * 3 + kx];
    const double G[4][3] = {{1, 0, 0}, {.5, .5, .5}, {.5, -.5, .5}, {0, 0, 1}};
    const int i = pos >> 2, j = pos & 3;
    double acc = 0;
    for (int p = 0; p < 3; ++p)
      for (int r = 0; r < 3; ++r) acc += G[i][p] * g[p][r] * G[j][r];
    u = (float)acc;
  }
  out[idx] = u;
}

}  // namespace

extern "C" size_t virnet_wino_weight_floats(int cin_pad, int n_pad) { return (size_t)n_pad * cin_pad * 16; }

extern "C" int virnet_pack_wino_weight(const float* w, int dgrad, int cout, int cin, int cin_pad, int n_pad, float* packed,
                                       void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_wino_weight: NULL pointer");
  VIRNET_REQUIRE(cout > 0 && cin > 0, "virnet_pack_wino_weight: bad extents cout=%d cin=%d", cout, cin);
  const int rows = dgrad ? cin : cout, ks = dgrad ? cout : cin;
  VIRNET_REQUIRE(cin_pad % 16 == 0 && cin_pad >= ks, "virnet_pack_wino_weight: cin_pad=%d does not cover %d contraction channels", cin_pad, ks);
  VIRNET_REQUIRE(n_pad % 32 == 0 && n_pad >= rows, "virnet_pack_wino_weight: n_pad=%d does not cover %d output channels", n_pad, rows);
  const long total = (long)n_pad * cin_pad * 16;
  hipLaunchKernelGGL(pack_wino_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), w, packed,
                     dgrad, cout, cin, cin_pad, n_pad);
  return virnet::check_launch("pack_wino launch");
}

extern "C" int virnet_conv_wino(const virnet_conv_desc* d, void* stream) {
  VIRNET_REQUIRE(d != nullptr, "virnet_conv_wino: desc is NULL");
  VIRNET_REQUIRE(d->x && d->wpack, "virnet_conv_wino: x / wpack is NULL");
  VIRNET_REQUIRE(d->ks == 3 && d->stride == 1 && d->epi == VIRNET_EPI_NHWC, "virnet_conv_wino: only the stride-1 3x3 NHWC conv (ks=%d stride=%d epi=%d)",
                 d->ks, d->stride, d->epi);
  VIRNET_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv_wino: empty input n=%d h=%d w=%d", d->n, d->h, d->w);
  VIRNET_REQUIRE(d->cin_pad >= 16 && d->cin_pad % 16 == 0, "virnet_conv_wino: cin_pad=%d is not a multiple of 16", d->cin_pad);
  VIRNET_REQUIRE(d->cout > 0 && d->cout % 32 == 0 && d->n_pad == d->cout, "virnet_conv_wino: cout=%d must be a multiple of 32 (n_pad=%d)", d->cout, d->n_pad);
  VIRNET_REQUIRE(d->y_raw || d->y_act, "virnet_conv_wino: no output pointer");
  VIRNET_REQUIRE((d->in_mul == nullptr) == (d->in_add == nullptr), "virnet_conv_wino: in_mul and in_add must be given together");
  VIRNET_REQUIRE(d->in_act || !d->in_mul, "virnet_conv_wino: in_mul/in_add without in_act");
  VIRNET_REQUIRE(!d->in_act || (d->in_slope >= 0.f && d->in_slope <= 1.f), "virnet_conv_wino: in_slope=%g outside [0,1]", d->in_slope);
  VIRNET_REQUIRE(!d->y_act || (d->slope >= 0.f && d->slope <= 1.f), "virnet_conv_wino: slope=%g outside [0,1]", d->slope);
  WArgs k{};
  k.x = d->x; k.up = d->wpack; k.bias = d->bias; k.res = d->res; k.mul = d->mul; k.add = d->add;
  k.in_mul = d->in_mul; k.in_add = d->in_add; k.mask = d->mask; k.y_raw = d->y_raw; k.y_act = d->y_act;
  k.N = d->n; k.H = d->h; k.W = d->w; k.Cin = d->cin_pad; k.Cout = d->cout;
  k.in_act = d->in_act; k.in_slope = d->in_slope; k.mask_slope = d->mask_slope; k.slope = d->slope;
  k.n64 = d->cout / 64; k.n32 = (d->cout % 64) / 32;
  return virnet::launch_wino_row(k, static_cast<hipStream_t>(stream), d->in_mul != nullptr);
}
