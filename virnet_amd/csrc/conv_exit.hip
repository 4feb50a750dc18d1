// conv_exit.hip -- the few-output-channel 3x3 exit convolutions with planar (NCHW) store: AttResUNet.tail 96 -> 3 + crop + `+ x_in`
// (networks/AttResUNet.py:139,173), DnCNN.conv_last 64 -> 1|2 + exp(clamp(.)) (DnCNN.py:29,41; VIRNet.py:43), KernelNet.tail 64 -> 3
// (KNet.py:49).  gfx950, split-fp16 MFMA products as conv_f16.hip.
//
// Taps as GEMM ROWS.  With cout * 9 <= 32 the convolution is a POINTWISE GEMM to rows (c, tap) followed by a shift-add:
//     z[(c, dy, dx)][p] = sum_ci w[c][ci][dy][dx] * x[ci][p]            one 32-row MFMA block holds all 27 (or 9, 18) rows
//     out[c][y][x]      = bias[c] + sum_{dy,dx} z[(c, dy, dx)][y + dy - 1][x + dx - 1]
// conv_f16's planar form spends a 32-row block on 3 channels for EVERY tap (27 useful of 288 rows per pixel and chunk); here every MFMA row
// is a (channel, tap) pair: 9x fewer MFMAs, and -- the point -- no halo tile of the input in LDS at all: a pixel is the column of exactly
// one MFMA, so its B fragment (8 channels of a 16-channel chunk per lane) goes global -> registers -> split -> MFMA.  What remains is
// reading the input once: the kernel is HBM-bound (algorithmic bytes = N*H*W*Cin*4 in + 4 B per output value).
//
// Workgroup = 4 waves = one 8 x 32 output tile.  Its z is needed on the 10 x 34 halo = 340 pixels, walked as 11 blocks of 32 (flat
// index, 12 surplus columns masked); block b belongs to wave b % 4.  Per block: Cin/16 chunks x (2 x 16-byte buffer loads -- a lane
// outside the halo / the image reads zeros through an out-of-range offset -- exact hi/lo split, 3 MFMAs); two blocks per wave in flight;
// A fragments (2 KB per chunk) sit in LDS.  z goes to LDS RAW as [row][pixel] fp32; then thread = output pixel sums its 9 z values per
// channel times the row's inverse weight scale (a scalar), applies bias and the planar epilogue (VIRNET_NCHW_*) and stores 128-byte runs.
// Nothing is read from global memory between the first pixel request and the last store except the pixels: scales, biases and the
// residual's values are requested at the top (round 5, tools/exit_timeline.py: per-register scale loads inside the block loop and
// per-channel scale / bias / residual loads behind the previous channel's store were 16 + 3 dependent round trips per tile, 0.28 -> 0.22 ms).
#include "conv_f16_common.h"

namespace {
using namespace virnet;

constexpr int EX_TH = 8, EX_TW = 32;
constexpr int EX_HW = EX_TW + 2;                 // halo row length
constexpr int EX_HPX = (EX_TH + 2) * EX_HW;      // 340
constexpr int EX_BLK = (EX_HPX + 31) / 32;       // 11
constexpr int EX_ZS = EX_BLK * 32 + 4;           // z row stride (floats)

template <int NCH>      // 16-channel chunks of the input (0: runtime count)
__global__ __launch_bounds__(256) void conv_exit_kernel(const FArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nch = NCH ? NCH : (a.Cin >> 4);
  char* const a_lds = smem;                                      // [chunk][hi|lo][64 lanes][16 B]
  float* const z_lds = reinterpret_cast<float*>(smem + nch * 2048);   // [cout * 9 rows + 1 dummy][EX_ZS]

  const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
  const int tile = xcd * a.tiles_per_xcd + q;
  if (q >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int img = fast_div(tile, a.mg_tpi);
  const int trem = tile - img * (a.ntx * a.nty);
  const int ty = fast_div(trem, a.mg_ntx), tx = trem - ty * a.ntx;
  const int oy0 = ty * EX_TH, ox0 = tx * EX_TW;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nrows = a.cout * 9;
  TSTAMP(0);

  if constexpr (NCH != 0) {                                       // (all pieces requested before the first lands: one round trip)
    f32x4 wv[NCH / 2];
#pragma unroll
    for (int i = 0; i < NCH / 2; ++i) wv[i] = *reinterpret_cast<const f32x4*>(a.wimg + i * 4096 + tid * 16);
#pragma unroll
    for (int i = 0; i < NCH / 2; ++i) *reinterpret_cast<f32x4*>(a_lds + i * 4096 + tid * 16) = wv[i];
  } else {
    for (int i = tid * 16; i < nch * 2048; i += 256 * 16) *reinterpret_cast<f32x4*>(a_lds + i) = *reinterpret_cast<const f32x4*>(a.wimg + i);
  }
  __syncthreads();
  TSTAMP(1);

  // thread = output pixel of the epilogue.  The residual's values (cout <= 3) are requested HERE, before the pixel blocks: they have landed
  // long before the shift-add wants them (requested there, they were one more HBM round trip between the barrier and the stores).
  const int oyl = tid >> 5, oxl = tid & 31;
  const int oy = oy0 + oyl, ox = ox0 + oxl;
  const bool inside = oy < a.crop_h && ox < a.crop_w;
  const size_t plane = (size_t)a.crop_h * a.crop_w;
  const size_t o0 = (size_t)img * a.cout * plane + (size_t)oy * a.crop_w + ox;
  // ... and the rows' inverse scales and the biases are read NOW, as scalars: behind the first store the compiler can no longer prove them
  // unchanged and reads them per channel with vector loads whose wait also waits for the previous channel's store (one more round trip each)
  float scv[27], bsv[3];
#pragma unroll
  for (int i = 0; i < 27; ++i) scv[i] = a.inv_scale[i];          // (the packed image always holds 32 scales)
#pragma unroll
  for (int c = 0; c < 3; ++c) bsv[c] = (a.bias && c < a.cout) ? a.bias[c] : 0.f;
  float rv[3] = {0.f, 0.f, 0.f};
  if (a.nchw_op == VIRNET_NCHW_ADD && inside) {
    const int rw = a.crop_w / a.res_sf;
    const size_t rplane = (size_t)(a.crop_h / a.res_sf) * rw;
    const size_t r0 = (size_t)img * a.cout * rplane + (size_t)(oy / a.res_sf) * rw + ox / a.res_sf;
#pragma unroll
    for (int c = 0; c < 3; ++c)
      if (c < a.cout) rv[c] = a.res[a.res_sf > 1 ? r0 + c * rplane : o0 + c * plane];
  }

  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;
  float amax = 0.f;
  // block -> this lane's pixel pointer (NULL: outside the halo / the image: zeros)
  auto pixel_of = [&](int blk) -> const float* {
    const int p = blk * 32 + l31;
    const int hy = p / EX_HW, hx = p - hy * EX_HW;
    const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
    const bool valid = blk < EX_BLK && p < EX_HPX && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    return valid ? ximg + ((size_t)gy * a.W + gx) * a.Cin + lhi * 8 : nullptr;
  };
  auto mma = [&](f32x16& acc, int c, f32x4 v0, f32x4 v1) {
    if (a.in_act) { v0 = lrelu4(v0, a.in_slope); v1 = lrelu4(v1, a.in_slope); }
    range_note(amax, v0, v1);
    h8 bh, bl;
    split8(v0, v1, bh, bl);
    const h8 ah = *reinterpret_cast<const h8*>(a_lds + c * 2048 + lane * 16);
    const h8 al = *reinterpret_cast<const h8*>(a_lds + c * 2048 + 1024 + lane * 16);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  };
  auto put_z = [&](int blk, const f32x16& acc) {
    // z[row][p] = the RAW accumulator (the row's inverse weight scale, a power of two, is applied by the shift-add: there it is a scalar
    // operand).  Accumulator register r of lane (l31, lhi) is row 8*(r>>2) + 4*lhi + (r&3), column l31; rows beyond the cout * 9 real ones
    // land in ONE dummy row behind them -- sixteen unconditional ds_write_b32.  (Round 5: the scale used to be read from global memory
    // here, one load + s_waitcnt vmcnt(0) per register inside the block loop -- sixteen dependent round trips per block, each of which
    // also drained the NEXT block's prefetched pixels: 42 of a workgroup's 55 k cycles, tools/exit_timeline.py.)
    const int p = blk * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 8 * (r >> 2) + 4 * lhi + (r & 3);
      z_lds[min(row, nrows) * EX_ZS + p] = acc[r];
    }
  };
  if constexpr (NCH != 0) {
    // A wave owns blocks wave, wave + 4, wave + 8 (the last one: waves 0..2).  TWO blocks' pixels are in flight per wave (2 x NCH x 32 B per
    // lane, ping-pong register sets, no copies); buffer loads, so that a lane outside the halo / the image reads zeros through an
    // out-of-range offset -- no branch around the loads, and no wait for a load in flight before a masked lane's zero is written.
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, a.H * a.W * a.Cin * 4, 0x00020000);
    auto pixel_off = [&](int blk) -> unsigned {
      const int p = blk * 32 + l31;
      const int hy = p / EX_HW, hx = p - hy * EX_HW;
      const int gy = oy0 - 1 + hy, gx = ox0 - 1 + hx;
      const bool valid = p < EX_HPX && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
      return valid ? (unsigned)(((gy * a.W + gx) * a.Cin + lhi * 8) * 4) : 0x80000000u;
    };
    auto request = [&](unsigned off, f32x4 (&v0)[NCH], f32x4 (&v1)[NCH]) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        v0[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off + c * 64, 0, 0));
        v1[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off + c * 64 + 16, 0, 0));
      }
    };
    auto compute = [&](int blk, f32x4 (&v0)[NCH], f32x4 (&v1)[NCH]) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) mma(acc, c, v0[c], v1[c]);
      put_z(blk, acc);
    };
    static_assert(EX_BLK > 8 && EX_BLK <= 12, "three blocks per wave at most, two at least");
    f32x4 p0[NCH], p1[NCH], q0[NCH], q1[NCH];
    const bool third = wave + 8 < EX_BLK;                 // (wave-uniform)
    request(pixel_off(wave), p0, p1);
    request(pixel_off(wave + 4), q0, q1);
    compute(wave, p0, p1);
    if (third) request(pixel_off(wave + 8), p0, p1);
    compute(wave + 4, q0, q1);
    if (third) compute(wave + 8, p0, p1);
  } else {
    for (int blk = wave; blk < EX_BLK; blk += 4) {
      const float* const px = pixel_of(blk);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      // (runtime chunk count: left rolled -- "#pragma unroll 2" here could not be honoured and warned in every build)
      for (int c = 0; c < nch; ++c) {
        f32x4 v0 = f32x4{0.f, 0.f, 0.f, 0.f}, v1 = v0;
        if (px) {
          v0 = *reinterpret_cast<const f32x4*>(px + c * 16);
          v1 = *reinterpret_cast<const f32x4*>(px + c * 16 + 4);
        }
        mma(acc, c, v0, v1);
      }
      put_z(blk, acc);
    }
  }
  TSTAMP(2);
  range_report(a.range_flag, amax);
  __syncthreads();
  TSTAMP(3);
#ifdef VIRNET_F16_TIMING
  if (a.tlog && tid == 0) {
    a.tlog[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
    a.tlog[(size_t)blockIdx.x * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
  }
#endif

  // ---- shift-add + planar epilogue: thread = output pixel
  if (!inside) return;
  float outv[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = bsv[c];
    const float* const zc = z_lds + (c < a.cout ? c * 9 : 0) * EX_ZS + oyl * EX_HW + oxl;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) v += zc[(dy * 3 + dx) * EX_ZS + dy * EX_HW + dx] * scv[c * 9 + dy * 3 + dx];
    if (a.nchw_op == VIRNET_NCHW_ADD) v += rv[c];
    else if (a.nchw_op == VIRNET_NCHW_EXPCLAMP) v = expf(fminf(fmaxf(v, a.clamp_lo), a.clamp_hi));
    outv[c] = v;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
    if (c < a.cout) a.y_raw[o0 + c * plane] = outv[c];
#ifdef VIRNET_F16_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  TSTAMP(4);
}

// rows (c, tap = dy*3 + dx) of the pointwise GEMM: per-row power-of-two scale (largest scaled magnitude in [8192, 16384)), split image
// [chunk][hi|lo][lane = row + 32*(k>>3)][k&7] preceded by the 32 inverse scales.  One block.
__global__ void pack_exit_kernel(const float* __restrict__ w, int cout, int cin, int cin_pad, float* __restrict__ inv_scale, char* __restrict__ img) {
  const int nrows = cout * 9;
  __shared__ int eb[32];
  if (threadIdx.x < 32) {
    const int row = threadIdx.x;
    float m = 0.f;
    if (row < nrows) {
      const int c = row / 9, t = row - c * 9;
      for (int k = 0; k < cin; ++k) m = fmaxf(m, fabsf(w[((size_t)c * cin + k) * 9 + t]));
    }
    int e = 0;
    if (m > 0.f) { frexpf(m, &e); e = 14 - e; }
    e = max(-100, min(100, e));
    eb[row] = e;
    inv_scale[row] = ldexpf(1.f, -e);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * cin_pad; i += blockDim.x) {
    const int row = i / cin_pad, k = i - row * cin_pad;
    float v = 0.f;
    if (row < nrows && k < cin) {
      const int c = row / 9, t = row - c * 9;
      v = ldexpf(w[((size_t)c * cin + k) * 9 + t], eb[row]);
    }
    const int chunk = k >> 4, kk = k & 15;
    const size_t base = (size_t)chunk * 2048 + (size_t)(row + 32 * (kk >> 3)) * 16 + (kk & 7) * 2;
    const _Float16 hi = (_Float16)v;
    *reinterpret_cast<_Float16*>(img + base) = hi;
    *reinterpret_cast<_Float16*>(img + base + 1024) = (_Float16)(v - (float)hi);
  }
}

template <int NCH>
int launch_exit(FArgs k, hipStream_t st) {
  const int nch = k.Cin >> 4;
  const int lds = nch * 2048 + (k.cout * 9 + 1) * EX_ZS * 4;    // A fragments + the (channel, tap) rows of z + the dummy row
  static unsigned long long attr_done = 0;
  auto kern = conv_exit_kernel<NCH>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_exit): %s", hipGetErrorString(e));
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * k.tiles_per_xcd)), dim3(256), lds, st, k);
  return virnet::check_launch("conv_exit launch");
}

#ifdef VIRNET_F16_TIMING
long long* g_xlog = nullptr;
#endif
}  // namespace

#ifdef VIRNET_F16_TIMING
extern "C" void virnet_debug_exit_timing_buffer(void* p) { g_xlog = static_cast<long long*>(p); }   // tools/exit_timeline.py
#endif

extern "C" size_t virnet_exit_weight_floats(int cin_pad) { return 32 + (size_t)cin_pad * 32; }      // 32 scales + cin_pad/16 x 2 KB

extern "C" int virnet_pack_exit_weight(const float* w, int cout, int cin, int cin_pad, float* packed, void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_exit_weight: NULL pointer");
  VIRNET_REQUIRE(cout >= 1 && cout * 9 <= 32, "virnet_pack_exit_weight: cout=%d: the (channel, tap) rows must fit one 32-row block (cout <= 3)", cout);
  VIRNET_REQUIRE(cin >= 1 && cin_pad % 16 == 0 && cin_pad >= cin, "virnet_pack_exit_weight: cin_pad=%d does not cover cin=%d", cin_pad, cin);
  hipLaunchKernelGGL(pack_exit_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), w, cout, cin, cin_pad, packed,
                     reinterpret_cast<char*>(packed + 32));
  return virnet::check_launch("pack_exit launch");
}

extern "C" int virnet_conv_exit(const virnet_conv_desc* d, void* stream) {
  VIRNET_REQUIRE(d != nullptr, "virnet_conv_exit: desc is NULL");
  VIRNET_REQUIRE(d->x && d->wpack && d->y_raw, "virnet_conv_exit: x / wpack / y_raw is NULL");
  VIRNET_REQUIRE(d->ks == 3 && d->stride == 1 && d->epi == VIRNET_EPI_NCHW, "virnet_conv_exit: only the stride-1 3x3 conv with planar store (ks=%d stride=%d epi=%d)",
                 d->ks, d->stride, d->epi);
  VIRNET_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv_exit: empty input n=%d h=%d w=%d", d->n, d->h, d->w);
  VIRNET_REQUIRE(d->cout >= 1 && d->cout * 9 <= 32, "virnet_conv_exit: cout=%d (the (channel, tap) rows must fit one 32-row block: cout <= 3)", d->cout);
  VIRNET_REQUIRE(d->cin_pad >= 16 && d->cin_pad % 16 == 0 && d->cin_pad <= 1024, "virnet_conv_exit: cin_pad=%d", d->cin_pad);
  VIRNET_REQUIRE(d->crop_h >= 1 && d->crop_h <= d->h && d->crop_w >= 1 && d->crop_w <= d->w, "virnet_conv_exit: crop %d x %d outside %d x %d", d->crop_h, d->crop_w, d->h, d->w);
  VIRNET_REQUIRE(d->nchw_op != VIRNET_NCHW_ADD || d->res, "virnet_conv_exit: VIRNET_NCHW_ADD without res");
  VIRNET_REQUIRE(d->res_sf >= 1 && d->crop_h % d->res_sf == 0 && d->crop_w % d->res_sf == 0, "virnet_conv_exit: res_sf=%d does not divide the crop", d->res_sf);
  VIRNET_REQUIRE(!d->mask && !d->mul && !d->in_mul && !d->y_act, "virnet_conv_exit: plain planar epilogue only");
  VIRNET_REQUIRE((long)d->h * d->w * d->cin_pad * 4 < (1L << 31), "virnet_conv_exit: one image's input (%d x %d x %d fp32) must stay below 2 GB (32-bit buffer offsets)", d->h, d->w, d->cin_pad);
  FArgs k{};
  k.x = d->x; k.inv_scale = d->wpack; k.wimg = reinterpret_cast<const char*>(d->wpack + 32);
  k.bias = d->bias; k.res = d->res; k.y_raw = d->y_raw;
  k.N = d->n; k.H = d->h; k.W = d->w; k.Cin = d->cin_pad; k.cout = d->cout;
  k.in_act = d->in_act; k.in_slope = d->in_slope;
  k.nchw_op = d->nchw_op; k.crop_h = d->crop_h; k.crop_w = d->crop_w; k.res_sf = d->res_sf; k.clamp_lo = d->clamp_lo; k.clamp_hi = d->clamp_hi;
  k.range_flag = virnet::range_flag_ptr();
#ifdef VIRNET_F16_TIMING
  k.tlog = g_xlog;
#endif
  // only the cropped region is produced
  k.nty = (d->crop_h + EX_TH - 1) / EX_TH;
  k.ntx = (d->crop_w + EX_TW - 1) / EX_TW;
  k.ntiles = k.N * k.nty * k.ntx;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  VIRNET_REQUIRE((unsigned long long)k.ntiles * (unsigned)(k.ntx * k.nty) < (1ull << 32), "virnet_conv_exit: %d tiles exceed the index arithmetic of one launch", k.ntiles);
  k.mg_ntx = div_magic(k.ntx);
  k.mg_tpi = div_magic(k.ntx * k.nty);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nch = d->cin_pad >> 4;
  if (nch == 6) return launch_exit<6>(k, st);
  if (nch == 4) return launch_exit<4>(k, st);
  VIRNET_REQUIRE(nch * 2048 + 33 * EX_ZS * 4 <= 160 * 1024, "virnet_conv_exit: cin_pad=%d does not fit LDS", d->cin_pad);
  return launch_exit<0>(k, st);
}
