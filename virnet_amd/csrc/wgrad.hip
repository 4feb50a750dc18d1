// wgrad.hip -- weight and bias gradients of the convolutions on VIRNet's training step (SURVEY.md 8-f1:
// train_denoising_syn.py:176-179, the backward of networks/AttResUNet.py:43,46,67,80,117,139 and networks/DnCNN.py:22-29).
//
//   dW[co][ci][ky][kx] = sum over output pixels p of  dY[p][co] * A[p*S + (ky,kx) - pad][ci]
// is a GEMM whose contraction runs over PIXELS: D[co][ci] += dY^T[co][p] * A[p][ci], once per tap.  On the matrix cores
// (v_mfma_f32_32x32x2_f32, exact fp32) one instruction contracts 2 pixels for a 32x32 (co, ci) block, so a wave keeps ALL taps
// of one (co-block, ci-block) pair in registers (KS*KS accumulator blocks) and per 2-pixel step reads ONE dY fragment and KS*KS
// shifted A fragments from LDS (ds_read_b32, one channel per lane -> conflict free); the 3x3 stride-1 case pairs two ROWS per
// MFMA and slides a 3x3 register window along the columns, so only 1 + 3 fragments are read per 9 MFMAs.
//
// Workgroup = 4 waves = 4 output rows x TW columns of one (co-block, ci-block) pair; it walks a strided list of such tiles
// (split-K over the image set) accumulating in registers, then adds its partial sums into dW with fp32 atomics (dW is
// zero-initialised by the caller).  Tiles are staged through LDS: the next tile's 32-channel pixel slabs are fetched
// global->registers while the current tile is consumed, and written after a barrier; out-of-image pixels and channels beyond
// the tensor are zero-filled; the same pre-activation as the forward conv (lrelu(x*mul+add)) is applied to A on the way in.
#include "common.h"
#include "../../include/virnet_hip.h"
#include <cstdlib>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct WArgs {
  const float* x;       // NHWC [n][h][w][cx]   forward input (cx channels stored per pixel)
  const float* dy;      // NHWC [n][oh][ow][cy] output gradient
  const float* in_mul;  // [n][cx] or null
  const float* in_add;
  float* dw;            // [cout][cin][ks][ks] (conv) or [cin][cout][2][2] (transposed conv), += with atomics
  int* ctr;             // [ncob*ncib] zero-initialised tile counters (dynamic tile hand-out per channel-block pair)
  int n, h, w, cx, oh, ow, cy;
  int cin, cout;        // real channel counts of dw (rows of dy used: cout, channels of x used: cin)
  int ncob, ncib;       // 32-channel blocks
  int ntx, nty, ntiles;
  int transposed;       // 1: dy rows are (ab*cout_t + co) of a space-to-depth gradient, dw is [cin][cout_t][2][2]
  int cout_t;
  int in_act;
  float in_slope;
};

template <int KS, int STRIDE, int TW>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WArgs a) {
  constexpr int TH = 4;
  constexpr int PAD = KS / 2;
  constexpr int NTAPS = KS * KS;
  constexpr int IH = (TH - 1) * STRIDE + KS, IW = (TW - 1) * STRIDE + KS;
  constexpr int XPIX = IH * IW, YPIX = TH * TW;
  constexpr int XPIECE = XPIX * 8, YPIECE = YPIX * 8;          // 16-B pieces of the 128-B (32-channel) records
  constexpr int XPT = (XPIECE + 255) / 256, YPT = (YPIECE + 255) / 256;

  __shared__ __attribute__((aligned(16))) float xs[XPIX * 32];
  __shared__ __attribute__((aligned(16))) float ys[YPIX * 32];

  const int blk = blockIdx.x;                                   // (co-block, ci-block) pair
  const int cob = blk / a.ncib, cib = blk - cob * a.ncib;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int sub = tid & 7;                                      // which 16 B of a 128-B record this thread stages
  const int xc0 = cib * 32 + sub * 4, yc0 = cob * 32 + sub * 4; // first channel of that piece
  const bool xc_ok = xc0 < a.cx, yc_ok = yc0 < a.cy;            // (channel counts are multiples of 4)
  const float slope_eff = a.in_act ? a.in_slope : 1.f;

  f32x4 xr[XPT], yr[YPT];
  bool xin[XPT], yin[YPT];
  f32x4 xm = f32x4{1.f, 1.f, 1.f, 1.f}, xa = f32x4{0.f, 0.f, 0.f, 0.f};   // this thread's channel quad is fixed

  auto fetch = [&](int tile) {
    const int tx = tile % a.ntx, ty = (tile / a.ntx) % a.nty, img = tile / (a.ntx * a.nty);
    const int oy0 = ty * TH, ox0 = tx * TW;
    const float* const ximg = a.x + (size_t)img * a.h * a.w * a.cx;
    const float* const yimg = a.dy + (size_t)img * a.oh * a.ow * a.cy;
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int p = min((k * 256 + tid) >> 3, XPIX - 1);
      const int iy = p / IW, ix = p - iy * IW;
      const int gy = oy0 * STRIDE - PAD + iy, gx = ox0 * STRIDE - PAD + ix;
      xin[k] = xc_ok && (unsigned)gy < (unsigned)a.h && (unsigned)gx < (unsigned)a.w;
      const int gyc = min(max(gy, 0), a.h - 1), gxc = min(max(gx, 0), a.w - 1);
      xr[k] = *reinterpret_cast<const f32x4*>(ximg + ((size_t)gyc * a.w + gxc) * a.cx + (xc_ok ? xc0 : 0));
    }
    if (a.in_mul) {
      xm = *reinterpret_cast<const f32x4*>(a.in_mul + (size_t)img * a.cx + (xc_ok ? xc0 : 0));
      xa = *reinterpret_cast<const f32x4*>(a.in_add + (size_t)img * a.cx + (xc_ok ? xc0 : 0));
    }
#pragma unroll
    for (int k = 0; k < YPT; ++k) {
      const int p = min((k * 256 + tid) >> 3, YPIX - 1);
      const int iy = p / TW, ix = p - iy * TW;
      const int gy = oy0 + iy, gx = ox0 + ix;
      yin[k] = yc_ok && gy < a.oh && gx < a.ow;
      const int gyc = min(gy, a.oh - 1), gxc = min(gx, a.ow - 1);
      yr[k] = *reinterpret_cast<const f32x4*>(yimg + ((size_t)gyc * a.ow + gxc) * a.cy + (yc_ok ? yc0 : 0));
    }
  };
  auto land = [&]() {
#pragma unroll
    for (int k = 0; k < XPT; ++k) {
      const int q = k * 256 + tid;
      if (q < XPIECE) {
        f32x4 v = xr[k];
        v = v * xm + xa;
        const f32x4 t = v * slope_eff;
        v = f32x4{fmaxf(v.x, t.x), fmaxf(v.y, t.y), fmaxf(v.z, t.z), fmaxf(v.w, t.w)};
        *reinterpret_cast<f32x4*>(xs + q * 4) = xin[k] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int k = 0; k < YPT; ++k) {
      const int q = k * 256 + tid;
      if (q < YPIECE) *reinterpret_cast<f32x4*>(ys + q * 4) = yin[k] ? yr[k] : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x16 acc[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // Tiles are handed out dynamically: the workgroups of one (co, ci) pair draw tile indices from a shared counter, so a slow
  // workgroup takes fewer tiles instead of holding up the launch (static shares left 1.67 of 2 waves per SIMD resident).
  // Thread 0 draws the index two tiles ahead; it travels through LDS across the barriers the tile loop already has.
  __shared__ int tsel[2];
  int* const ctr = a.ctr + blk;
  if (tid == 0) {
    tsel[0] = atomicAdd(ctr, 1);
    tsel[1] = atomicAdd(ctr, 1);
  }
  __syncthreads();
  int tile = tsel[0], nxt = tsel[1];
  if (tile < a.ntiles) {
    fetch(tile);
    land();
  }
  __syncthreads();
  while (tile < a.ntiles) {
    int drawn = 0;
    if (tid == 0) drawn = atomicAdd(ctr, 1);                    // the tile after next; its value is only needed after the MFMAs
    if (nxt < a.ntiles) fetch(nxt);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KS == 3 && STRIDE == 1) {
      // Row-pair contraction with a sliding register window: the two pixels an MFMA contracts are (row, col) for lanes 0-31 and
      // (row+1, col) for lanes 32-63, so the nine taps of column c are the 3x3 window x[row+dy][c+dx] in BOTH halves; stepping
      // to column c+1 re-uses six of the nine fragments from registers: 1 dY + 3 new x fragments per 9 MFMAs (instead of 1 + 9).
      // Wave w: row pair (w&1), column half (w>>1) of the 4x32 tile.
      const int r0 = 2 * (wave & 1) + lhi, cbeg = (TW / 2) * (wave >> 1);
      const float* const xrow = xs + (r0 * IW + cbeg) * 32 + l31;
      const float* const yrow = ys + (r0 * TW + cbeg) * 32 + l31;
      float xw[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        xw[dy][0] = xrow[(dy * IW + 0) * 32];
        xw[dy][1] = xrow[(dy * IW + 1) * 32];
      }
#pragma unroll
      for (int c = 0; c < TW / 2; ++c) {
        const float yv = yrow[c * 32];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) xw[dy][2] = xrow[(dy * IW + c + 2) * 32];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv, xw[dy][dx], acc[dy * 3 + dx], 0, 0, 0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) { xw[dy][0] = xw[dy][1]; xw[dy][1] = xw[dy][2]; }
      }
    } else {
    // this wave's row: TW/2 two-pixel steps
#pragma unroll 4
    for (int s = 0; s < TW / 2; ++s) {
      const float yv = ys[(wave * TW + 2 * s + lhi) * 32 + l31];
      float xv[NTAPS];
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const int dy = (KS == 3) ? t / 3 : 0, dx = (KS == 3) ? t % 3 : 0;
        xv[t] = xs[((wave * STRIDE + dy) * IW + (2 * s + lhi) * STRIDE + dx) * 32 + l31];
      }
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(yv, xv[t], acc[t], 0, 0, 0);
    }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (tid == 0) tsel[0] = drawn;
    __syncthreads();
    if (nxt < a.ntiles) land();
    const int nn = tsel[0];
    __syncthreads();
    tile = nxt;
    nxt = nn;
  }

  // ---- partial sums -> dW.  The four waves hold partials of the SAME (co, ci, tap) entries (different rows of the tiles):
  // reduce them through LDS, tap by tap, then ONE fp32 atomic per entry per workgroup.
  // Accumulator element (r, lane): row co = cob*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), column ci = cib*32 + (lane&31).
  float* const red = xs;                             // 4 waves x 16 x 64 floats = 16 KB (<= the pixel tile)
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[t][r];
    __syncthreads();
#pragma unroll
    for (int e = tid; e < 1024; e += 256) {
      const int r = e >> 6, ln = e & 63;
      const float v = red[e] + red[1024 + e] + red[2048 + e] + red[3072 + e];
      const int co = cob * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
      const int ci = cib * 32 + (ln & 31);
      if (co < a.cout && ci < a.cin) {
        size_t o;
        if (a.transposed) {                          // row co = ab*cout_t + c  ->  dw[ci][c][a][b]
          const int ab = co / a.cout_t, c = co - ab * a.cout_t;
          o = ((size_t)ci * a.cout_t + c) * 4 + ab;
        } else {
          o = ((size_t)co * a.cin + ci) * NTAPS + t;
        }
        atomicAdd(a.dw + o, v);
      }
    }
    __syncthreads();
  }
}

// db[c] += sum over pixels of dy[p][c]   (NHWC, c multiple of 4); one block = 256 pixels-rows strided, LDS tree per channel quad
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ dy, float* __restrict__ db, size_t npix, int c,
                                                     int cvalid) {
  const int quads = c >> 2;
  __shared__ float4 red[256];
  for (int q0 = 0; q0 < quads; q0 += 64) {
    const int q = q0 + (threadIdx.x & 63);
    const int ph = threadIdx.x >> 6;                     // 4 pixel phases per block
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q < quads) {
      for (size_t p = (size_t)blockIdx.x * 4 + ph; p < npix; p += (size_t)gridDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(dy + p * c + q * 4);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < 64 && q < quads) {
      float4 t = red[threadIdx.x];
      for (int k = 1; k < 4; ++k) {
        const float4 u = red[threadIdx.x + 64 * k];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      const int c0 = q * 4;
      if (c0 + 0 < cvalid) atomicAdd(db + c0 + 0, t.x);
      if (c0 + 1 < cvalid) atomicAdd(db + c0 + 1, t.y);
      if (c0 + 2 < cvalid) atomicAdd(db + c0 + 2, t.z);
      if (c0 + 3 < cvalid) atomicAdd(db + c0 + 3, t.w);
    }
    __syncthreads();
  }
}

template <int KS, int STRIDE, int TW>
int launch_wgrad(WArgs k, hipStream_t st) {
  k.ntx = (k.ow + TW - 1) / TW;
  k.nty = (k.oh + 3) / 4;
  k.ntiles = k.n * k.nty * k.ntx;
  const int pairs = k.ncob * k.ncib;
  // Split-K factor: fill the chip exactly ONCE.  Every workgroup runs for the whole launch (its share of the tiles), so a grid
  // that is not a whole number of residency rounds leaves SIMD slots empty (r01 PMC: 1.50 of 2 resident waves per SIMD with a
  // 2.004-round grid).  Residency comes from the occupancy query (VGPR/LDS-limited, 2 workgroups per CU here).
  static int wg_per_cu = 0;
  if (!wg_per_cu) {
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&wg_per_cu, reinterpret_cast<const void*>(conv_wgrad_kernel<KS, STRIDE, TW>), 256, 0) !=
            hipSuccess || wg_per_cu < 1)
      wg_per_cu = 1;
  }
  static const int forced = [] { const char* e = getenv("VIRNET_WGRAD_WGS"); return e ? atoi(e) : 0; }();   // tuning knob
  const int slots = forced > 0 ? forced : 256 * wg_per_cu;
  int split = slots / pairs;
  if (split > k.ntiles) split = k.ntiles;
  if (split < 1) split = 1;
  hipLaunchKernelGGL((conv_wgrad_kernel<KS, STRIDE, TW>), dim3(pairs, split), dim3(256), 0, st, k);
  return virnet::check_launch("conv_wgrad launch");
}

}  // namespace

extern "C" int virnet_conv_wgrad(const virnet_wgrad_desc* d, void* stream) {
  VIRNET_REQUIRE(d && d->x && d->dy && d->dw && d->counters, "virnet_conv_wgrad: NULL pointer");
  VIRNET_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv_wgrad: empty input");
  VIRNET_REQUIRE(d->cx > 0 && d->cx % 4 == 0 && d->cy > 0 && d->cy % 4 == 0, "virnet_conv_wgrad: stored channel counts cx=%d cy=%d must be multiples of 4", d->cx, d->cy);
  VIRNET_REQUIRE(d->cin >= 1 && d->cin <= d->cx, "virnet_conv_wgrad: cin=%d > stored %d", d->cin, d->cx);
  VIRNET_REQUIRE((d->ks == 3 && (d->stride == 1 || d->stride == 2)) || (d->ks == 1 && d->stride == 1),
                 "virnet_conv_wgrad: unsupported ks=%d stride=%d", d->ks, d->stride);
  VIRNET_REQUIRE((d->in_mul == nullptr) == (d->in_add == nullptr), "virnet_conv_wgrad: in_mul and in_add go together");
  VIRNET_REQUIRE(!d->in_act || (d->in_slope >= 0.f && d->in_slope <= 1.f), "virnet_conv_wgrad: in_slope=%g outside [0,1]", d->in_slope);
  WArgs k{};
  k.x = d->x; k.dy = d->dy; k.in_mul = d->in_mul; k.in_add = d->in_add; k.dw = d->dw; k.ctr = d->counters;
  k.n = d->n; k.h = d->h; k.w = d->w; k.cx = d->cx; k.cy = d->cy; k.cin = d->cin;
  k.in_act = d->in_act; k.in_slope = d->in_slope;
  if (d->stride == 2) {
    VIRNET_REQUIRE(d->h % 2 == 0 && d->w % 2 == 0, "virnet_conv_wgrad: stride-2 input %dx%d must be even", d->h, d->w);
    k.oh = d->h / 2; k.ow = d->w / 2;
  } else {
    k.oh = d->h; k.ow = d->w;
  }
  k.transposed = d->transposed;
  if (d->transposed) {
    VIRNET_REQUIRE(d->ks == 1 && d->cy == 4 * d->cout, "virnet_conv_wgrad: transposed-conv gradient wants ks=1 and a space-to-depth dy with 4*cout=%d channels (got %d)", 4 * d->cout, d->cy);
    k.cout_t = d->cout;
    k.cout = 4 * d->cout;
  } else {
    VIRNET_REQUIRE(d->cout >= 1 && d->cout <= d->cy, "virnet_conv_wgrad: cout=%d > stored %d", d->cout, d->cy);
    k.cout = d->cout;
  }
  k.ncob = (k.cout + 31) / 32;
  k.ncib = (k.cin + 31) / 32;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d->ks == 3 && d->stride == 1) return launch_wgrad<3, 1, 32>(k, st);
  if (d->ks == 3 && d->stride == 2) return launch_wgrad<3, 2, 16>(k, st);
  return launch_wgrad<1, 1, 32>(k, st);
}

extern "C" int virnet_colsum(const float* dy, float* db, long npix, int c, int cvalid, void* stream) {
  VIRNET_REQUIRE(dy && db && npix > 0 && c > 0 && c % 4 == 0 && cvalid >= 1 && cvalid <= c, "virnet_colsum: bad arguments");
  long blocks = (npix + 1023) / 1024;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dy, db, (size_t)npix, c,
                     cvalid);
  return virnet::check_launch("colsum launch");
}
