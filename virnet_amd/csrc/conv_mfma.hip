// conv_mfma.hip -- fp32 implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
// One kernel template covers every dense contraction on VIRNet's forward path:
//   KS=3,S=1  AttResBlock.conv1/conv2 (networks/AttResUNet.py:43,46,55,58), DnCNN convs (networks/DnCNN.py:22-29),
//             RB_Layer convs (networks/KNet.py:32,34), AttResUNet.head/tail (:117-119,:139), KernelNet.tail (KNet.py:49)
//   KS=3,S=2  DownBlock.downsampler (networks/AttResUNet.py:67)
//   KS=1,S=1  UpBlock.upsampler, ConvTranspose2d(k2,s2) == 1x1 GEMM to 4*Cout columns + depth-to-space (AttResUNet.py:80)
//
// GEMM view: M = output pixels, N = output channels, K = KS*KS*Cin.  v_mfma_f32_32x32x2_f32 is exact fp32 (an fmaf chain),
// so parity with the reference's fp32 conv is at re-association level (~1e-6), far inside the 1e-3 contract.
//
// Workgroup = 4 waves = (4*MREP) output rows x 32 output columns x NB=32*NREP output channels.
// Wave w owns rows [w*MREP, (w+1)*MREP); an MFMA "M" block is 32 consecutive pixels of one row.
// K is walked as 16-channel chunks x KS*KS taps ("stages"); per stage a wave issues MREP*NREP*8 MFMAs.
//
// LDS (all dynamic, 16-B aligned):
//   in[2] : halo tile of the current / next 16-channel chunk, 64-B pixel records, 16-B slot s of pixel p stored at
//           slot s ^ ((p>>2)&3)  -> ds_read_b128 of 16 consecutive pixels is bank-conflict free (256-B bank row)
//   w[2]  : weights of the current / next stage, [n][16 k] records with the same swizzle (pre-applied by the packer,
//           so staging is a linear 16-B copy)
// Global->LDS staging goes through registers: loads for stage s+1 are issued before the MFMAs of stage s and written to
// the other buffer after them (one barrier per stage).  Out-of-image halo pixels are written as zeros, which is exactly the
// conv's zero padding because producers store the ACTIVATED tensor (see y_act below) -- the "pad after activation" rule
// of the pre-activation block (AttResUNet.py:55,58).
//
// Epilogue (per 32x32 block a lane holds ONE output channel for 16 pixels, so stores are 128-B channel runs):
//   raw = acc + bias (+ residual)            -> y_raw
//   act = lrelu(raw * mul[n,c] + add[n,c])   -> y_act   (what the next pre-activation conv consumes)
//
#include "common.h"
#include "../../include/virnet_hip.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct KArgs {
  const float* x;
  const float* wp;
  const float* bias;
  const float* res;
  const float* mul;
  const float* add;
  float* y_raw;
  float* y_act;
  int N, H, W, Cin;        // input
  int OH, OW;              // GEMM-M spatial extent (conv output; for CONVT the INPUT grid)
  int NP;                  // padded GEMM-N
  int cout;                // real channels of the stored tensor
  int ntx, nty, ntiles, tiles_per_xcd;
  int epi, nchw_op, crop_h, crop_w, res_sf;
  float slope, clamp_lo, clamp_hi;
};

__device__ __forceinline__ int swz(int p) { return (p >> 2) & 3; }

template <int KS, int STRIDE, int MREP, int NREP>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const KArgs a) {
  constexpr int TH = 4 * MREP;
  constexpr int PAD = KS / 2;
  constexpr int IH = (TH - 1) * STRIDE + KS;
  constexpr int IW = 31 * STRIDE + KS;
  constexpr int NPIX = IH * IW;
  constexpr int NPIECE = NPIX * 4;                 // 16-B pieces of one input chunk
  constexpr int NTAPS = KS * KS;
  constexpr int PPT = (NPIECE + 255) / 256;        // input pieces per thread per chunk
  constexpr int LPT = (PPT + NTAPS - 1) / NTAPS;   // ... issued per tap stage
  constexpr int NB = 32 * NREP;
  constexpr int WPIECE = NB * 4;                   // 16-B pieces of one weight stage
  constexpr int WPT = (WPIECE + 255) / 256;
  constexpr int IN_BYTES = NPIX * 64;
  constexpr int W_BYTES = NB * 64;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const in_lds = smem;                       // [2][IN_BYTES]
  char* const w_lds = smem + 2 * IN_BYTES;         // [2][W_BYTES]

  // ---- workgroup -> (tile, channel block).  Block b runs on XCD b%8 (observed, speed only): give every XCD a contiguous
  // range of tiles so halo rows and the per-XCD weight working set stay in that XCD's L2, and keep the channel blocks of
  // one tile adjacent in time so the input tile is fetched from HBM once.
  const int ncb = a.NP / NB;
  const int xcd = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int cb = q % ncb;
  const int tile = xcd * a.tiles_per_xcd + q / ncb;
  if (q / ncb >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx = tile % a.ntx;
  const int ty = (tile / a.ntx) % a.nty;
  const int img = tile / (a.ntx * a.nty);
  const int oy0 = ty * TH, ox0 = tx * 32;
  const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int nchunks = a.Cin >> 4;
  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;
  const float* const wcb = a.wp + (size_t)cb * nchunks * NTAPS * (W_BYTES / 4);

  // ---- staging helpers -------------------------------------------------------------------------------------------
  // input piece k of this thread: q = k*256+tid -> (pixel p, slot s)
  auto in_piece = [&](int k, int chunk, f32x4& v, int& dst) {
    const int qq = k * 256 + tid;
    dst = -1;
    v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (k < PPT && qq < NPIECE) {
      const int p = qq >> 2, s = qq & 3;
      const int iy = p / IW, ix = p - iy * IW;
      const int gy = iy0 + iy, gx = ix0 + ix;
      dst = p * 64 + ((s ^ swz(p)) << 4);
      if ((unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W)
        v = *reinterpret_cast<const f32x4*>(ximg + ((size_t)(gy * a.W + gx) * a.Cin + chunk * 16 + s * 4));
    }
  };

  // ---- prologue: chunk 0 input + stage 0 weights ------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    f32x4 v; int dst;
    in_piece(k, 0, v, dst);
    if (dst >= 0) *reinterpret_cast<f32x4*>(in_lds + dst) = v;
  }
#pragma unroll
  for (int i = 0; i < WPT; ++i) {
    const int qq = i * 256 + tid;
    if (qq < WPIECE) *reinterpret_cast<f32x4*>(w_lds + qq * 16) = *reinterpret_cast<const f32x4*>(wcb + qq * 4);
  }
  __syncthreads();

  f32x16 acc[MREP][NREP];
#pragma unroll
  for (int mr = 0; mr < MREP; ++mr)
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.f;

  // B-fragment byte offsets inside a weight stage (lane constants): record n = nr*32+l31, slot (2j+lhi)^swz(n)
  const int bsw = swz(l31);
  const int boff0 = l31 * 64 + (((0 + lhi) ^ bsw) << 4);
  const int boff1 = l31 * 64 + (((2 + lhi) ^ bsw) << 4);

  int stage = 0;
  for (int c = 0; c < nchunks; ++c) {
    const char* const in_cur = in_lds + (c & 1) * IN_BYTES;
    char* const in_nxt = in_lds + ((c + 1) & 1) * IN_BYTES;
    const bool more_chunks = (c + 1 < nchunks);
#pragma unroll 1
    for (int t = 0; t < NTAPS; ++t, ++stage) {
      const bool more_stages = more_chunks || (t + 1 < NTAPS);
      // ---- issue global loads for the next stage (weights) and a slice of the next chunk (input)
      f32x4 wreg[WPT];
      if (more_stages) {
        const float* const wsrc = wcb + (size_t)(stage + 1) * (W_BYTES / 4);
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
          const int qq = i * 256 + tid;
          if (qq < WPIECE) wreg[i] = *reinterpret_cast<const f32x4*>(wsrc + qq * 4);
        }
      }
      f32x4 ireg[LPT];
      int idst[LPT];
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        idst[i] = -1;
        if (more_chunks) in_piece(t * LPT + i, c + 1, ireg[i], idst[i]);
      }

      // ---- MFMAs of this stage
      const int dy = (KS == 3) ? t / 3 : 0;
      const int dx = (KS == 3) ? t - dy * 3 : 0;
      const char* const w_cur = w_lds + (stage & 1) * W_BYTES;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 af[MREP], bf[NREP];
#pragma unroll
        for (int mr = 0; mr < MREP; ++mr) {
          const int p = ((wave * MREP + mr) * STRIDE + dy) * IW + l31 * STRIDE + dx;
          af[mr] = *reinterpret_cast<const f32x4*>(in_cur + p * 64 + (((2 * j + lhi) ^ swz(p)) << 4));
        }
#pragma unroll
        for (int nr = 0; nr < NREP; ++nr)
          bf[nr] = *reinterpret_cast<const f32x4*>(w_cur + nr * 2048 + (j ? boff1 : boff0));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mr = 0; mr < MREP; ++mr)
#pragma unroll
            for (int nr = 0; nr < NREP; ++nr)
              acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mr][r], bf[nr][r], acc[mr][nr], 0, 0, 0);
      }

      // ---- land the prefetched data in the other buffers
      if (more_stages) {
        char* const w_nxt = w_lds + ((stage + 1) & 1) * W_BYTES;
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
          const int qq = i * 256 + tid;
          if (qq < WPIECE) *reinterpret_cast<f32x4*>(w_nxt + qq * 16) = wreg[i];
        }
      }
#pragma unroll
      for (int i = 0; i < LPT; ++i)
        if (idst[i] >= 0) *reinterpret_cast<f32x4*>(in_nxt + idst[i]) = ireg[i];
      __syncthreads();
    }
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------
  const int nbase = cb * NB;
  if (a.epi == VIRNET_EPI_NHWC) {
    const int C = a.cout;
    const size_t img_off = (size_t)img * a.OH * a.OW * C;
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      const int n = nbase + nr * 32 + l31;
      const float bias = a.bias ? a.bias[n] : 0.f;
      const float mul = a.mul ? a.mul[(size_t)img * C + n] : 1.f;
      const float add = a.add ? a.add[(size_t)img * C + n] : 0.f;
#pragma unroll
      for (int mr = 0; mr < MREP; ++mr) {
        const int oy = oy0 + wave * MREP + mr;
        if (oy >= a.OH) continue;
        const size_t row_off = img_off + (size_t)oy * a.OW * C + n;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (ox < a.OW) {
            const size_t o = row_off + (size_t)ox * C;
            float v = acc[mr][nr][r] + bias;
            if (a.res) v += a.res[o];
            if (a.y_raw) a.y_raw[o] = v;
            if (a.y_act) {
              const float u = fmaf(v, mul, add);
              a.y_act[o] = u > 0.f ? u : u * a.slope;
            }
          }
        }
      }
    }
  } else if (a.epi == VIRNET_EPI_CONVT) {
    // GEMM column n' = ab*cout + co ; pixel (iy,ix) -> output (2*iy+a, 2*ix+b)
    const int C = a.cout;
    const int OH2 = 2 * a.OH, OW2 = 2 * a.OW;
    const size_t img_off = (size_t)img * OH2 * OW2 * C;
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      const int np = nbase + nr * 32 + l31;
      const int ab = np / C, co = np - ab * C;
      const int ua = ab >> 1, ub = ab & 1;
      const float bias = a.bias ? a.bias[co] : 0.f;
      const float mul = a.mul ? a.mul[(size_t)img * C + co] : 1.f;
      const float add = a.add ? a.add[(size_t)img * C + co] : 0.f;
#pragma unroll
      for (int mr = 0; mr < MREP; ++mr) {
        const int iy = oy0 + wave * MREP + mr;
        if (iy >= a.OH) continue;
        const size_t row_off = img_off + (size_t)(2 * iy + ua) * OW2 * C + co;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ix = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (ix < a.OW) {
            const size_t o = row_off + (size_t)(2 * ix + ub) * C;
            float v = acc[mr][nr][r] + bias;
            if (a.res) v += a.res[o];
            if (a.y_raw) a.y_raw[o] = v;
            if (a.y_act) {
              const float u = fmaf(v, mul, add);
              a.y_act[o] = u > 0.f ? u : u * a.slope;
            }
          }
        }
      }
    }
  } else {  // VIRNET_EPI_NCHW: few real channels (<= 32), planar store with crop
    const int n = nbase + l31;
    if (n < a.cout) {
      const float bias = a.bias ? a.bias[n] : 0.f;
      const size_t plane = (size_t)a.crop_h * a.crop_w;
      const size_t base = ((size_t)img * a.cout + n) * plane;
#pragma unroll
      for (int mr = 0; mr < MREP; ++mr) {
        const int oy = oy0 + wave * MREP + mr;
        if (oy >= a.crop_h) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
          if (ox < a.crop_w) {
            const size_t o = base + (size_t)oy * a.crop_w + ox;
            float v = acc[mr][0][r] + bias;
            if (a.nchw_op == VIRNET_NCHW_ADD) {
              if (a.res_sf > 1) {
                const int rw = a.crop_w / a.res_sf;
                v += a.res[((size_t)img * a.cout + n) * (size_t)(a.crop_h / a.res_sf) * rw + (size_t)(oy / a.res_sf) * rw + ox / a.res_sf];
              } else {
                v += a.res[o];
              }
            } else if (a.nchw_op == VIRNET_NCHW_EXPCLAMP) {
              v = expf(fminf(fmaxf(v, a.clamp_lo), a.clamp_hi));
            }
            a.y_raw[o] = v;
          }
        }
      }
    }
  }
}

template <int KS, int STRIDE, int MREP, int NREP>
int launch(const KArgs& ka, hipStream_t st) {
  constexpr int TH = 4 * MREP;
  constexpr int IH = (TH - 1) * STRIDE + KS, IW = 31 * STRIDE + KS;
  constexpr int LDS = 2 * IH * IW * 64 + 2 * 32 * NREP * 64;
  static bool attr_done = false;
  auto kern = conv_mfma_kernel<KS, STRIDE, MREP, NREP>;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_mfma): %s", hipGetErrorString(e));
    attr_done = true;
  }
  KArgs k = ka;
  k.nty = (k.OH + TH - 1) / TH;
  k.ntx = (k.OW + 31) / 32;
  k.ntiles = k.N * k.nty * k.ntx;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  const int ncb = k.NP / (32 * NREP);
  const unsigned grid = (unsigned)(8 * k.tiles_per_xcd * ncb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, k);
  return virnet::check_launch("conv_mfma launch");
}

int pick_nrep(int nblocks32) {
  static const int pref[] = {3, 4, 5, 2, 7, 1};
  for (int d : pref)
    if (nblocks32 % d == 0) return d;
  return 1;
}

// Small grids (deep U-Net levels, single images) use 4-row tiles so the 256 CUs still see >= 2 workgroups each.
int pick_mrep(const virnet_conv_desc* d) {
  if (d->stride == 2 || d->nrep == 7) return 1;
  const int oh = d->h, ow = d->w;
  const long wg8 = (long)d->n * ((oh + 7) / 8) * ((ow + 31) / 32) * (d->n_pad / (32 * d->nrep));
  return wg8 < 2048 ? 1 : 2;
}

}  // namespace

extern "C" int virnet_conv_mfma_variant(const virnet_conv_desc* d, int out[4]) {
  VIRNET_REQUIRE(d && out, "virnet_conv_mfma_variant: NULL pointer");
  out[0] = d->ks; out[1] = d->stride; out[2] = pick_mrep(d); out[3] = d->nrep;
  return 0;
}

extern "C" int virnet_conv_get_plan(int ks, int stride, int cin, int gemm_n, virnet_conv_plan* plan) {
  VIRNET_REQUIRE(plan != nullptr, "virnet_conv_get_plan: plan is NULL");
  VIRNET_REQUIRE((ks == 3 && (stride == 1 || stride == 2)) || (ks == 1 && stride == 1),
                 "virnet_conv_get_plan: unsupported ks=%d stride=%d", ks, stride);
  VIRNET_REQUIRE(cin > 0 && gemm_n > 0, "virnet_conv_get_plan: bad extents cin=%d n=%d", cin, gemm_n);
  plan->cin_pad = (cin + 15) / 16 * 16;
  const int nb32 = (gemm_n + 31) / 32;
  plan->nrep = pick_nrep(nb32);
  plan->n_pad = nb32 * 32;
  return 0;
}

extern "C" int virnet_conv_mfma(const virnet_conv_desc* d, void* stream) {
  VIRNET_REQUIRE(d != nullptr, "virnet_conv_mfma: desc is NULL");
  VIRNET_REQUIRE(d->x && d->wpack, "virnet_conv_mfma: x / wpack is NULL");
  VIRNET_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv_mfma: empty input n=%d h=%d w=%d", d->n, d->h, d->w);
  VIRNET_REQUIRE(d->cin_pad > 0 && d->cin_pad % 16 == 0, "virnet_conv_mfma: cin_pad=%d is not a multiple of 16", d->cin_pad);
  VIRNET_REQUIRE(d->nrep >= 1 && d->n_pad % (32 * d->nrep) == 0, "virnet_conv_mfma: n_pad=%d not a multiple of 32*nrep (nrep=%d)",
                 d->n_pad, d->nrep);
  KArgs k{};
  k.x = d->x; k.wp = d->wpack; k.bias = d->bias; k.res = d->res; k.mul = d->mul; k.add = d->add;
  k.y_raw = d->y_raw; k.y_act = d->y_act;
  k.N = d->n; k.H = d->h; k.W = d->w; k.Cin = d->cin_pad;
  k.NP = d->n_pad; k.cout = d->cout;
  k.epi = d->epi; k.nchw_op = d->nchw_op; k.crop_h = d->crop_h; k.crop_w = d->crop_w; k.res_sf = d->res_sf;
  k.slope = d->slope; k.clamp_lo = d->clamp_lo; k.clamp_hi = d->clamp_hi;
  if (d->stride == 2) {
    VIRNET_REQUIRE(d->h % 2 == 0 && d->w % 2 == 0, "virnet_conv_mfma: stride-2 input %dx%d must be even", d->h, d->w);
    k.OH = d->h / 2; k.OW = d->w / 2;
  } else {
    k.OH = d->h; k.OW = d->w;
  }
  switch (d->epi) {
    case VIRNET_EPI_NHWC:
      VIRNET_REQUIRE(d->n_pad == d->cout, "virnet_conv_mfma: NHWC store needs cout (%d) to be a multiple of 32", d->cout);
      VIRNET_REQUIRE(d->y_raw || d->y_act, "virnet_conv_mfma: no output pointer");
      break;
    case VIRNET_EPI_CONVT:
      VIRNET_REQUIRE(d->ks == 1 && d->n_pad == 4 * d->cout && d->cout % 32 == 0,
                     "virnet_conv_mfma: transposed-conv store needs ks=1, n_pad=4*cout, cout%%32==0 (cout=%d n_pad=%d)", d->cout, d->n_pad);
      VIRNET_REQUIRE(d->y_raw || d->y_act, "virnet_conv_mfma: no output pointer");
      break;
    case VIRNET_EPI_NCHW:
      VIRNET_REQUIRE(d->cout >= 1 && d->cout <= 32 && d->n_pad == 32 && d->nrep == 1,
                     "virnet_conv_mfma: planar store handles 1..32 channels (cout=%d)", d->cout);
      VIRNET_REQUIRE(d->y_raw, "virnet_conv_mfma: y_raw is NULL");
      VIRNET_REQUIRE(d->crop_h >= 1 && d->crop_h <= k.OH && d->crop_w >= 1 && d->crop_w <= k.OW,
                     "virnet_conv_mfma: crop %dx%d outside output %dx%d", d->crop_h, d->crop_w, k.OH, k.OW);
      VIRNET_REQUIRE(d->nchw_op != VIRNET_NCHW_ADD || d->res, "virnet_conv_mfma: VIRNET_NCHW_ADD without res");
      VIRNET_REQUIRE(d->res_sf <= 1 || (d->crop_h % d->res_sf == 0 && d->crop_w % d->res_sf == 0),
                     "virnet_conv_mfma: crop %dx%d is not a multiple of res_sf=%d", d->crop_h, d->crop_w, d->res_sf);
      break;
    default:
      return virnet::set_error("virnet_conv_mfma: unknown epilogue %d", d->epi);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool small = pick_mrep(d) == 1;
#define VIRNET_CASE(KS_, S_, M_, N_) return launch<KS_, S_, M_, N_>(k, st)
  if (d->ks == 3 && d->stride == 1) {
    switch (d->nrep) {
      case 1: if (small) VIRNET_CASE(3, 1, 1, 1); else VIRNET_CASE(3, 1, 2, 1);
      case 2: if (small) VIRNET_CASE(3, 1, 1, 2); else VIRNET_CASE(3, 1, 2, 2);
      case 3: if (small) VIRNET_CASE(3, 1, 1, 3); else VIRNET_CASE(3, 1, 2, 3);
      case 4: if (small) VIRNET_CASE(3, 1, 1, 4); else VIRNET_CASE(3, 1, 2, 4);
      case 5: if (small) VIRNET_CASE(3, 1, 1, 5); else VIRNET_CASE(3, 1, 2, 5);
      case 7: VIRNET_CASE(3, 1, 1, 7);
    }
  } else if (d->ks == 3 && d->stride == 2) {
    switch (d->nrep) {
      case 1: VIRNET_CASE(3, 2, 1, 1);
      case 2: VIRNET_CASE(3, 2, 1, 2);
      case 3: VIRNET_CASE(3, 2, 1, 3);
      case 4: VIRNET_CASE(3, 2, 1, 4);
      case 5: VIRNET_CASE(3, 2, 1, 5);
      case 7: VIRNET_CASE(3, 2, 1, 7);
    }
  } else if (d->ks == 1 && d->stride == 1) {
    switch (d->nrep) {
      case 1: if (small) VIRNET_CASE(1, 1, 1, 1); else VIRNET_CASE(1, 1, 2, 1);
      case 2: if (small) VIRNET_CASE(1, 1, 1, 2); else VIRNET_CASE(1, 1, 2, 2);
      case 3: if (small) VIRNET_CASE(1, 1, 1, 3); else VIRNET_CASE(1, 1, 2, 3);
      case 4: if (small) VIRNET_CASE(1, 1, 1, 4); else VIRNET_CASE(1, 1, 2, 4);
      case 5: if (small) VIRNET_CASE(1, 1, 1, 5); else VIRNET_CASE(1, 1, 2, 5);
      case 7: VIRNET_CASE(1, 1, 1, 7);
    }
  }
#undef VIRNET_CASE
  return virnet::set_error("virnet_conv_mfma: no kernel for ks=%d stride=%d nrep=%d", d->ks, d->stride, d->nrep);
}
