// conv_f16.hip -- the 3x3 convolution as an implicit GEMM on the CDNA4 f16 matrix pipe with SPLIT fp32 operands (gfx950).
//
// Same call sites as conv_mfma.hip / wino.hip (AttResBlock.conv1/conv2, networks/AttResUNet.py:43,46,55,58; DnCNN mid convs,
// networks/DnCNN.py:25-28; RB_Layer convs, networks/KNet.py:32,34) and their input-gradient GEMMs.
//
// Arithmetic.  Every fp32 operand is split exactly into two fp16 numbers, v = hi + lo (hi = rne16(v), lo = rne16(v - hi); the
// residual is <= 2^-23 |v|, or <= 2^-25 absolute once lo is subnormal -- gfx950's MFMA keeps fp16 subnormals, measured), and
//     w * x  ~=  w_lo * x_hi + w_hi * x_lo + w_hi * x_hi
// runs as three v_mfma_f32_32x32x16_f16 (fp16 products are exact in fp32; accumulation is fp32).  The dropped term w_lo * x_lo is
// <= 2^-22 |w x|.  Weights are pre-scaled per output channel by a power of two (exact) so their low halves stay normal; the
// epilogue multiplies by the inverse.  Measured on MI355X against fp64 (profiles/r02_probes.md): K = 864 dot products 6.8e-8 of
// sum|w x| (the fp32 MFMA chain: 1.3e-7); whole denoise-syn network 8.0e-6 max-abs on mu (fp32: 8.9e-6).  The f16 pipe retires
// 16x the MACs of the fp32 pipe per cycle, so three products cost 3/16 of the fp32 direct form (Winograd F(2x2,3x3): 7.1/16).
// Range: activations must stay below 65504 in magnitude (fp16 max); beyond it the result is Inf/NaN, never silently wrong.
//
// GEMM view: D[cout][pixel] += W[cout][k] * X[k][pixel], k = (tap, cin).  Workgroup = 4 waves = (4*MREP) output rows x 32 output
// columns x 32*NREP output channels; wave w owns rows [w*MREP, (w+1)*MREP) x all channels (MREP x NREP accumulator blocks of 32x32).
// K is walked as 16-channel chunks x 9 taps; one (tap, chunk) is ONE MFMA k-step, three products deep.
//   X (pixels): per chunk a halo tile in LDS as two planes (hi, lo) of 32-B pixel records (16 fp16 channels), 16-B slot s of pixel p
//               stored at s ^ ((p>>3)&1): ds_read_b128 B-fragments are bank-conflict free for every tap shift.  Pixels are fetched
//               global -> registers (fp32), pre-activated (AttResUNet.py:54-55: lrelu(x*mul+add), zero outside the image AFTER it),
//               split and written to the other buffer one third per tap group.
//   W (weights): pre-split, pre-scaled, packed as the exact LDS image [tap][slab][hi|lo][lane][16 B]; streamed global -> LDS by
//               DMA (buffer_load_dwordx4 ... lds, conv_f16_common.h: lds_dma16), three taps (one kernel column) per stage, double buffered.
//   Taps run column by column (dx outer): the MREP+2 input rows a wave needs for one dx are read ONCE and serve all three dy.
// One barrier per tap group (54 MFMAs per wave at 2x3 blocks); two workgroups per CU cover each other's barriers and epilogues.
// Epilogue: inverse scale, bias, mask, residual, activation as conv_mfma.hip, after a per-wave LDS turn-around of each 32-channel slab
// that makes every residual load and store a run of whole 128-B lines.
#include "conv_f16_common.h"
#include <cstdlib>
#include <type_traits>

namespace {
using namespace virnet;

// EPI specialises the epilogue so its loads are straight-line code the compiler can count (a runtime `if (ptr) load` merges into a
// vmcnt(0) before every store, which serialises the stores on their acknowledgements -- measured: 7 k cycles per slab):
//   bit 0 = residual, bit 1 = LeakyReLU-derivative mask (backward), one stored tensor; 4 = everything by runtime pointer (two
//   stored tensors, SFT on the output); 5 = planar NCHW store of <= 32 channels with crop and `+ x_in` / exp(clamp) (NREP = 1:
//   AttResUNet.tail, AttResUNet.py:139,173; DnCNN.conv_last, DnCNN.py:29 + VIRNet.py:43; KernelNet.tail, KNet.py:49).
// BF = 1: the bf16-operand variant (BASELINE configs[4]'s training precision): ONE product per MAC on v_mfma_f32_32x32x16_bf16, operands
// rounded to bf16 while they are staged (weights when they are packed), no low halves anywhere; everything else is the same kernel.
// TE = 1 (training step): the epilogue also EMITS the channel-major image of the stored tensor that the weight-gradient GEMM contracts
// over (FArgs::t_out; wgrad_f16.hip's T) and the tile's channel sums -- what a virnet_chsplit pass over the stored tensor would produce,
// without reading it back.  The reader's items become 8 CONSECUTIVE pixels (one 16-byte T unit per channel and plane) instead of 8
// pixels eight apart; the NHWC stores cover 128 contiguous bytes per 8 lanes either way.
// ENT = 1: the network entries (AttResUNet.head :153-155, DnCNN.conv1 :38): ONE 16-channel chunk whose pixels are gathered from the NCHW
// image (+ conditioning vector / map) while they are staged -- virnet_pack_input's record never exists in HBM (FArgs::ent).
template <int MREP, int NREP, int EPI, int BF = 0, int TE = 0, int ENT = 0>
__global__ __launch_bounds__(256, 2) void conv_f16_kernel(const FArgs a) {
  static_assert(!TE || (MREP == 2 && EPI < 4), "T emission: 8-row tiles (8 items per thread), single-store epilogues");
  static_assert(!ENT || (EPI == 0 && !BF && !TE), "entry form: plain single-store epilogue, split-fp16 operands");
  constexpr int TH = 4 * MREP, IH = TH + 2, IW = 34, NPIX = IH * IW;
  constexpr int NPIECE = NPIX * 2;                 // (pixel, 8-channel half) staging pieces of one chunk
  constexpr int PPT = (NPIECE + 255) / 256;
  static_assert(PPT >= 2 && PPT <= 3, "one staged piece per tap group; surplus threads redo piece k-1");
  constexpr int PLANE = NPIX * 32, XB = 2 * PLANE;
  // CS ("chunk stages", the single-slab forms NREP = 1 of under-filled launches): a weight stage is a whole CHUNK (nine taps)
  // and the workgroup meets at a barrier once per chunk instead of once per tap group.  With 9 MFMAs per wave between two barriers the
  // group was 1 250 cycles of barrier + exposed LDS latency for 288 cycles of matrix work (profiles/r05_probes.md 9: removing the weight
  // DMA and the pixel loads altogether changed 30.8 us into 28.2); same LDS as the three-slab form's two tap-group stages.
  constexpr bool CS = NREP == 1 && !ENT;
  constexpr int WGRP = (CS ? 9 : 3) * NREP * 2048;  // one weight stage: a tap group (three taps) of fragments, or the chunk's nine
  constexpr int NDMA = (CS ? 9 : 3) * NREP * 2;     // ... in 1-KB DMA pieces
  constexpr int NR = MREP + 2;                     // input rows per wave and kernel column
  constexpr int NB = 32 * NREP;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const x_lds = smem;                        // [2][hi plane | lo plane]
  char* const w_lds = smem + 2 * XB;               // [2][3 taps][NREP][hi|lo][1 KB]

  // ---- workgroup -> (tile, channel block): contiguous tile ranges per XCD (block b runs on XCD b%8), channel blocks adjacent
  const int ncb = a.NP / NB;
  const int xcd = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int cb = __builtin_amdgcn_readfirstlane(q % ncb);
  const int tile = __builtin_amdgcn_readfirstlane(xcd * a.tiles_per_xcd + q / ncb);
  if (q / ncb >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx = __builtin_amdgcn_readfirstlane(tile % a.ntx);
  const int ty = __builtin_amdgcn_readfirstlane((tile / a.ntx) % a.nty);
  const int img = __builtin_amdgcn_readfirstlane(tile / (a.ntx * a.nty));
  const int oy0 = ty * TH, ox0 = tx * 32;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;

  const int tid = threadIdx.x;
  TSTAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nch = a.Cin >> 4;
  const int nstages = nch * 3;
  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;

  // ---- pixel staging: piece k of this thread = (pixel p, half h = tid&1) -> 8 channels chunk*16 + 8h .. of pixel p
  unsigned soff[PPT];
  int sdst[PPT];
  bool sinb[PPT];
  [[maybe_unused]] int egy[PPT], egx[PPT];             // ENT: clamped pixel coordinates of the piece
  [[maybe_unused]] bool ehalf[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int qq = k * 256 + tid;
    const int qc = qq < NPIECE ? qq : qq - 256;      // surplus threads of the last piece redo their previous one (same bytes):
    const int p = qc >> 1, h = qc & 1;               // no divergent store, the staging code stays one basic block
    const int iy = p / IW, ix = p - iy * IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    sinb[k] = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
    egy[k] = gyc; egx[k] = gxc; ehalf[k] = h != 0;
    soff[k] = (unsigned)((gyc * a.W + gxc) * a.Cin + h * 8);
    sdst[k] = p * 32 + ((h ^ ((p >> 3) & 1)) << 4);
  }
  const bool in_sft = a.in_mul != nullptr;
  const float* const imul = in_sft ? a.in_mul + (size_t)img * a.Cin + (tid & 1) * 8 : nullptr;
  const float* const iadd = in_sft ? a.in_add + (size_t)img * a.Cin + (tid & 1) * 8 : nullptr;
  const float in_slope_eff = a.in_act ? a.in_slope : 1.f;
  float amax = 0.f;                                // range guard (conv_f16_common.h)
  auto stage_store = [&](char* xb, int chunk, int k, f32x4 r0, f32x4 r1) {
    if (in_sft) {
      const f32x4 m0 = *reinterpret_cast<const f32x4*>(imul + chunk * 16), m1 = *reinterpret_cast<const f32x4*>(imul + chunk * 16 + 4);
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(iadd + chunk * 16), a1 = *reinterpret_cast<const f32x4*>(iadd + chunk * 16 + 4);
      r0 = r0 * m0 + a0;
      r1 = r1 * m1 + a1;
    }
    r0 = lrelu4(r0, in_slope_eff);
    r1 = lrelu4(r1, in_slope_eff);
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    r0 = sinb[k] ? r0 : z;
    r1 = sinb[k] ? r1 : z;
    if constexpr (BF) {
      *reinterpret_cast<b8*>(xb + sdst[k]) = to_bf16x8(r0, r1);
    } else {
      h8 hi, lo;
      range_note(amax, r0, r1);
      split8(r0, r1, hi, lo);
      *reinterpret_cast<h8*>(xb + sdst[k]) = hi;
      *reinterpret_cast<h8*>(xb + PLANE + sdst[k]) = lo;
    }
  };

  // ---- weight DMA: piece q = (tap-in-group, slab, hi|lo), 1 KB = the fragment of one MFMA operand; wave w moves pieces w, w+4, ...
  const size_t slab_bytes = (size_t)nch * 9 * 2048;
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wimg + (size_t)(a.slab_base + cb * NREP) * slab_bytes), 0,
                                                     (int)(NREP * slab_bytes), 0x00020000);
  const int lane16w = lane * 16;
  auto dma_group = [&](int stage, char* wb) {
#pragma unroll
    for (int i = 0; i < (NDMA + 3) / 4; ++i) {
      const int qd = BF ? 2 * (i * 4 + wave) : i * 4 + wave;        // (bf16 variant: the hi pieces only)
      if (qd < NDMA) {
        const int tg = qd / (NREP * 2), rem = qd - tg * (NREP * 2);
        lds_dma16(wrs, wb + qd * 1024, lane16w, (rem >> 1) * (int)slab_bytes + ((stage * (CS ? 9 : 3) + tg) * 2 + (rem & 1)) * 1024);     // (CS: stage = chunk)
      }
    }
  };

  // ---- fragment addressing
  // B (pixels): row r of this wave (input row wave*MREP + r), kernel column dx: pixel p = (wave*MREP + r)*IW + l31 + dx
  int boff[NR][3];
#pragma unroll
  for (int r = 0; r < NR; ++r)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = (wave * MREP + r) * IW + l31 + dx;
      boff[r][dx] = p * 32 + ((lhi ^ ((p >> 3) & 1)) << 4);
    }
  const int aoff = lane * 16;

  f32x16 acc[MREP][NREP];
#pragma unroll
  for (int mr = 0; mr < MREP; ++mr)
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.f;

  // ---- prologue: weights of stage 0 by DMA, pixels of chunk 0 staged whole; chunk 1's pieces are requested by the groups of chunk 0
  dma_group(0, w_lds);
  f32x4 pr0[PPT], pr1[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    if constexpr (ENT) {
      pr0[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      pr1[k] = pr0[k];
      if (!ehalf[k]) entry_pixel(a.ent, img, egy[k], egx[k], pr0[k], pr1[k]);      // (channels 8..15 of an entry record are zero)
    } else {
      pr0[k] = *reinterpret_cast<const f32x4*>(ximg + soff[k]);
      pr1[k] = *reinterpret_cast<const f32x4*>(ximg + soff[k] + 4);
    }
  }
#pragma unroll
  for (int k = 0; k < PPT; ++k) stage_store(x_lds, 0, k, pr0[k], pr1[k]);
  __syncthreads();
  TSTAMP(1);

  h8 ah[2][NREP], al[2][NREP];      // A fragments (weights) of tap t and t+1
  h8 bh[NR], bl[NR];                // B fragments (pixels) of the current kernel column
  auto read_a = [&](const char* wb, int tg, h8 (&h)[NREP], h8 (&l)[NREP]) {
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      h[nr] = *reinterpret_cast<const h8*>(wb + ((tg * NREP + nr) * 2 + 0) * 1024 + aoff);
      if (!BF) l[nr] = *reinterpret_cast<const h8*>(wb + ((tg * NREP + nr) * 2 + 1) * 1024 + aoff);
    }
  };
  auto read_b = [&](const char* xb, int r, int dx) {
    bh[r] = *reinterpret_cast<const h8*>(xb + boff[r][dx]);
    if (!BF) bl[r] = *reinterpret_cast<const h8*>(xb + PLANE + boff[r][dx]);
  };

  // One tap group = kernel column g of chunk c (P = c&1: pixel buffer; weight buffer (P+g)&1; A register set (P + 3g + dy)&1).
  // A wave issues in order and an MFMA occupies the matrix pipe for 32 cycles: everything else (fragment reads for the next tap, the
  // staging arithmetic) is threaded BETWEEN the MFMAs with sched_group_barrier so a wave on its own keeps the pipe close to busy.
  auto group = [&](int c, auto pc, auto gc) {
    constexpr int P = decltype(pc)::value, g = decltype(gc)::value;
    const int stage = c * 3 + g;
    const char* const xb = x_lds + P * XB;
    char* const xn = x_lds + (P ^ 1) * XB;
    const char* const wb = CS ? w_lds + P * WGRP + g * (3 * NREP * 2048) : w_lds + ((P + g) & 1) * WGRP;
    char* const wn = CS ? w_lds + (P ^ 1) * WGRP : w_lds + ((P + g + 1) & 1) * WGRP;
    if constexpr (CS) {
      if (g == 0 && c + 1 < nch) dma_group(c + 1, wn);             // the next chunk's nine taps: three tap groups of time to land
    } else {
      if (stage + 1 < nstages) dma_group(stage + 1, wn);
    }
    // one piece of the next chunk's pixels (the last chunk re-stages itself into the idle buffer: no branch in the tap code)
    const int cn = min(c + 1, nch - 1);
    f32x4 s0 = f32x4{0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if constexpr (g < PPT && !ENT) {                   // (entry form: one chunk, gathered in the prologue; nothing to re-stage)
      const float* const src = ximg + soff[g] + cn * 16;
      s0 = *reinterpret_cast<const f32x4*>(src);
      s1 = *reinterpret_cast<const f32x4*>(src + 4);
    }
    // operands of this group's first tap (the weights only became visible at the barrier)
    read_a(wb, 0, ah[(P + 3 * g) & 1], al[(P + 3 * g) & 1]);
    if (g == 0) {
#pragma unroll
      for (int r = 0; r < MREP; ++r) read_b(xb, r, 0);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int cur = (P + 3 * g + dy) & 1;
      SB();
      // requests for the next tap: its weights (same group) and the input row it needs first; at the last tap of a column the
      // first rows of the next column (same chunk) -- those registers were last used by tap dy = 1 -- and this group's staging
      constexpr int NRD = BF ? NREP + 1 : 2 * NREP + 2;
      if (dy < 2) {
        read_a(wb, dy + 1, ah[cur ^ 1], al[cur ^ 1]);
        read_b(xb, MREP + dy, g);
      } else {
        if (g < 2) {
#pragma unroll
          for (int r = 0; r < MREP; ++r) read_b(xb, r, g + 1);
        }
        if constexpr (g < PPT && !ENT) stage_store(xn, cn, g, s0, s1);
      }
#pragma unroll
      for (int part = BF ? 2 : 0; part < 3; ++part)
#pragma unroll
        for (int mr = 0; mr < MREP; ++mr)
#pragma unroll
          for (int nr = 0; nr < NREP; ++nr) {
            const h8 wa = (part == 0) ? al[cur][nr] : ah[cur][nr];
            const h8 xv = (part == 1) ? bl[mr + dy] : bh[mr + dy];
            if constexpr (BF)
              acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, wa), __builtin_bit_cast(b8, xv), acc[mr][nr], 0, 0, 0);
            else
              acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xv, acc[mr][nr], 0, 0, 0);
          }
      // interleave: one MFMA, then one LDS read (taps 0, 1) or a handful of staging VALU ops (tap 2)
      constexpr int NM = (BF ? 1 : 3) * MREP * NREP;
      if (dy < 2) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < NRD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (g < 2 && i < (BF ? 1 : 2) * MREP) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (g < PPT) __builtin_amdgcn_sched_group_barrier(0x002, BF ? 8 : 5, 0);
        }
        if (g < PPT) __builtin_amdgcn_sched_group_barrier(0x200, BF ? 1 : 2, 0);
      }
    }
    SB();
    if constexpr (!CS || g == 2) __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  int c = 0;
  for (; c + 1 < nch; c += 2) {
    group(c, I0{}, I0{}); group(c, I0{}, I1{}); group(c, I0{}, I2{});
    group(c + 1, I1{}, I0{}); group(c + 1, I1{}, I1{}); group(c + 1, I1{}, I2{});
  }
  if (c < nch) { group(c, I0{}, I0{}); group(c, I0{}, I1{}); group(c, I0{}, I2{}); }
  TSTAMP(2);
  range_report(a.range_flag, amax);

  // ---- epilogue
  // (lane coordinates re-derived from the hardware lane count through an opaque copy: derived from `tid` they stay live across the K loop,
  // and at the 256-register budget of the 2 x 3-block tile the allocator parked them -- and with them two B-fragment addresses that are
  // re-loaded INSIDE the loop behind a vmcnt(0) -- in scratch)
  int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(lane_e));
#endif
  const int l31_e = lane_e & 31, lhi_e = lane_e >> 5;
  if constexpr (EPI == 5) {
    // planar store: lane = pixel (ox0 + l31_e), register r = channel (r&3) + 8*(r>>2) + 4*lhi_e; 32 consecutive x per channel = 128-B runs
    static_assert(NREP == 1, "planar store is for <= 32 channels");
    const int px = ox0 + l31_e;
    const size_t plane = (size_t)a.crop_h * a.crop_w;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * lhi_e;
      if (n >= a.cout) continue;
      const float bias = a.bias ? a.bias[n] : 0.f;
      const float inv = a.inv_scale[n];
      const size_t base = ((size_t)img * a.cout + n) * plane;
#pragma unroll
      for (int mr = 0; mr < MREP; ++mr) {
        const int oy = oy0 + wave * MREP + mr;
        if (oy < a.crop_h && px < a.crop_w) {
          const size_t o = base + (size_t)oy * a.crop_w + px;
          float v = acc[mr][0][r] * inv + bias;
          if (a.nchw_op == VIRNET_NCHW_ADD) {
            if (a.res_sf > 1) {
              const int rw = a.crop_w / a.res_sf;
              v += a.res[((size_t)img * a.cout + n) * (size_t)(a.crop_h / a.res_sf) * rw + (size_t)(oy / a.res_sf) * rw + px / a.res_sf];
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
    TSTAMP(3);
    return;
  }
  // Each wave turns its own MREP x 32 pixel x 32 channel slab around through a private LDS region ([pixel][32 channels], 16 B of
  // padding per pixel; the pixel / weight buffers are free after the last barrier): lane = (pixel j>>3 of 8, channel quad j&7), so
  // one store / residual-load instruction covers 8 pixels x 128 contiguous bytes instead of 32 scattered 32-B pieces.
  const int nbase = a.slab_base * 32 + cb * NB;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  const int C = a.cout;
  const size_t img_off = (size_t)img * a.H * a.W * C;
  constexpr int TPIX = 144, NIT = MREP * 4;          // 8 pixels per pass
  constexpr int TREG = MREP * 32 * TPIX + (TE ? 128 : 0);   // one slab of one wave; two regions per wave (ping-pong)
  static_assert(4 * 2 * TREG <= 81920, "turn-around regions: two workgroups per CU");   // (launch<> sizes LDS for the larger of the two)
  char* const tbuf = smem + wave * (2 * TREG);
  const int cq = lane_e & 7, psub = lane_e >> 3;
  // LDS slot of tile pixel p (TE: 16 more bytes per 8 pixels, so that the lanes of one read -- pixels 8 apart -- keep the 36-dword spacing)
  auto pixoff = [](int p) { return p * TPIX + (TE ? (p >> 3) * 16 : 0); };
  unsigned eoff[NIT];
  bool eok[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int pix = TE ? psub * 8 + it : it * 8 + psub;
    const int oy = oy0 + wave * MREP + (pix >> 5), ox = ox0 + (pix & 31);
    eok[it] = oy < a.H && ox < a.W;
    eoff[it] = (unsigned)(min(oy, a.H - 1) * a.W + min(ox, a.W - 1)) * (unsigned)C + (unsigned)(nbase + cq * 4);
  }
  auto turn_in = [&](int nr, int region = 0) {
#pragma unroll
    for (int mr = 0; mr < MREP; ++mr)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(tbuf + region * TREG + pixoff(mr * 32 + l31_e) + (8 * g + 4 * lhi_e) * 4) =
            f32x4{acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]};
  };
  auto mask4 = [&](f32x4 v, f32x4 m) {
    return f32x4{m.x > 0.f ? v.x : v.x * a.mask_slope, m.y > 0.f ? v.y : v.y * a.mask_slope,
                 m.z > 0.f ? v.z : v.z * a.mask_slope, m.w > 0.f ? v.w : v.w * a.mask_slope};
  };
  if constexpr (EPI < 4) {
    // ONE stored tensor, straight-line code: every load is requested before the stores that follow it and none sits behind a branch,
    // so the compiler's counted vmcnt waits never include a store acknowledgement.
    constexpr bool RES = (EPI & 1) != 0, MASK = (EPI & 2) != 0;
    const float* const rimg = a.res + img_off;
    const float* const mimg = a.mask + img_off;
    float* const y = (a.y_act ? a.y_act : a.y_raw) + img_off;
    const float slope_eff = a.y_act ? a.slope : 1.f;                        // max(v, 1*v) == v: raw store
    const float* const bp = a.bias ? a.bias : a.inv_scale;                  // (no bias: read something valid, scale it by 0)
    const float hb = a.bias ? 1.f : 0.f;
    // gfx950 counts loads and stores on ONE counter that the compiler must treat as out of order once both kinds are pending: a
    // load consumed after a store was issued costs a vmcnt(0), i.e. the store's acknowledgement.  So every load of the tile is
    // requested AND waited for before the first store (EPI 3 cannot hold two operand tiles: its mask comes slab by slab).
    constexpr bool HOIST = EPI == 1 || EPI == 2;
    f32x4 bias4[NREP], inv4[NREP], op1[HOIST ? NREP : 1][NIT];
    const float* const op1p = RES ? rimg : mimg;
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      inv4[nr] = *reinterpret_cast<const f32x4*>(a.inv_scale + nbase + nr * 32 + cq * 4);
      bias4[nr] = *reinterpret_cast<const f32x4*>(bp + nbase + nr * 32 + cq * 4);
      if (HOIST) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) op1[nr][it] = *reinterpret_cast<const f32x4*>(op1p + eoff[it] + nr * 32);
      }
    }
    TSTAMP(6);
    turn_in(0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      asm volatile("" ::"v"(inv4[nr]), "v"(bias4[nr]));
      if (HOIST) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) asm volatile("" ::"v"(op1[nr][it]));
      }
    }
#endif
    // stores through a buffer descriptor: out-of-image lanes get an out-of-range offset and are dropped by the bounds check, so the
    // slab loop has no branch (the tile's reads, arithmetic and stores stay one schedulable block)
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y, 0, a.H * a.W * C * 4, 0x00020000);
    unsigned yoff[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) yoff[it] = eok[it] ? eoff[it] * 4u : 0x80000000u;
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      f32x4 mv[NIT], rv[NIT];
      if (EPI == 3) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          mv[it] = *reinterpret_cast<const f32x4*>(mimg + eoff[it] + nr * 32);
          rv[it] = *reinterpret_cast<const f32x4*>(rimg + eoff[it] + nr * 32);
        }
      }
      if (nr + 1 < NREP) turn_in(nr + 1, (nr + 1) & 1);          // next slab into the other region while this one is read back
      const f32x4 b4 = bias4[nr] * hb;
      f32x4 tv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) tv[it] = *reinterpret_cast<const f32x4*>(tbuf + (nr & 1) * TREG + pixoff(TE ? psub * 8 + it : it * 8 + psub) + cq * 16);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        f32x4 v = tv[it] * inv4[nr] + b4;
        if (MASK) v = mask4(v, EPI == 3 ? mv[it] : op1[HOIST ? nr : 0][it]);
        if (RES) v += EPI == 3 ? rv[it] : op1[HOIST ? nr : 0][it];
        v = lrelu4(v, slope_eff);
        if (TE) tv[it] = v;
        // (the slab offset rides in the instruction's immediate, NOT in soffset: hipcc 7.2 schedules a v_pk_* write of the data
        // registers straight behind a 16-B buffer store with an SGPR offset -- the hazard that needs a wait state -- and the odd
        // elements of the stored quad come out wrong)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, yoff[it] + nr * 128, 0, 0);
      }
      if constexpr (TE) {
        // thread (cq, psub) holds 8 consecutive pixels of tile row wave*MREP + (psub >> 2), x-segment psub & 3, for 4 channels
        const int cbg = (nbase >> 5) + nr;                      // 32-channel block of the stored tensor
        f32x4 cs = zero4;
        u32x4 uh[4], ul[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float e8[8];
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const float sv = eok[it] ? tv[it][c] : 0.f;       // (tile pixels beyond the image are zero in T, as chsplit writes them)
            cs[c] += sv;
            e8[it] = a.t_act ? fmaxf(sv, sv * a.t_slope) : sv;
          }
          t_units(e8, BF != 0, uh[c], ul[c]);
        }
        // Re-coalesce the wave's units through ITS turn-around region of this slab (free: every lane has read its items; wave-private, so the
        // wave's in-order LDS pipe is the only ordering needed): written in T's order [row 2][plane][x-segment 4][32 ch], read back lane-linear --
        // a store instruction then covers 1 KB of contiguous T instead of 64 pieces of 16 B that sit 64 B apart (profiles/r04_probes.md 8).
        // (A DPP quad transpose with the same effect cost more than it saved: 24.9 -> 27.7 ms fp32-class, 17.2 -> 18.5 bf16.)
        {
          constexpr int NPL = BF ? 1 : 2;
          char* const ub = tbuf + (nr & 1) * TREG;
          const int urow = psub >> 2, uxq = psub & 3;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(ub + ((((urow * NPL + 0) * 4 + uxq) * 32 + 4 * cq + i) << 4)) = uh[i];
            if (!BF) *reinterpret_cast<u32x4*>(ub + ((((urow * NPL + 1) * 4 + uxq) * 32 + 4 * cq + i) << 4)) = ul[i];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          char* const t0 = a.t_out + ((((size_t)img * (a.H + 2) + oy0 + wave * MREP + 1) * a.t_cb + cbg) * a.t_npl * a.t_nseg + ((ox0 >> 3) + 1)) * 512;
          const size_t trow_bytes = (size_t)a.t_cb * a.t_npl * a.t_nseg * 512;
#pragma unroll
          for (int k = 0; k < 4 * NPL; ++k) {
            const int u = k * 64 + lane_e;                       // unit: channel u & 31, x-segment (u >> 5) & 3, (row, plane) = u >> 7
            const int rp = u >> 7, r = rp / NPL, pl = rp - r * NPL;
            const u32x4 v = *reinterpret_cast<const u32x4*>(ub + (u << 4));
            if (oy0 + wave * MREP + r < a.H)
              *reinterpret_cast<u32x4*>(t0 + r * trow_bytes + (size_t)pl * a.t_nseg * 512 + ((u >> 5) & 3) * 512 + (u & 31) * 16) = v;
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the region is the next-but-one slab's turn-around buffer)
        }
        if (a.t_col) {                                          // wave's channel sums: the 8 psub lanes of a channel quad, then one row per wave
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            cs[c] += __shfl_xor(cs[c], 8);
            cs[c] += __shfl_xor(cs[c], 16);
            cs[c] += __shfl_xor(cs[c], 32);
          }
          if (psub == 0) *reinterpret_cast<f32x4*>(a.t_col + ((size_t)cbg * a.t_nblk + (size_t)tile * 4 + wave) * 32 + cq * 4) = cs;
        }
      }
      if (nr == 0) TSTAMP(7);
    }
  } else {
    // generic form (two stored tensors and / or SFT on the output): optional operands by runtime pointer
    const float* const rimg = a.res ? a.res + img_off : nullptr;
    const float* const mimg = a.mask ? a.mask + img_off : nullptr;
    float* const yraw = a.y_raw ? a.y_raw + img_off : nullptr;
    float* const yact = a.y_act ? a.y_act + img_off : nullptr;
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      const int co = nbase + nr * 32 + cq * 4;
      const f32x4 inv4 = *reinterpret_cast<const f32x4*>(a.inv_scale + co);
      const f32x4 bias4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + co) : zero4;
      f32x4 mul4 = f32x4{1.f, 1.f, 1.f, 1.f}, add4 = zero4;
      if (a.mul) {                                           // SFT on the output (AttResUNet.py:57-58): SISR down path
        mul4 = *reinterpret_cast<const f32x4*>(a.mul + (size_t)img * C + co);
        add4 = *reinterpret_cast<const f32x4*>(a.add + (size_t)img * C + co);
      }
      f32x4 rv[NIT], mv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        rv[it] = rimg ? *reinterpret_cast<const f32x4*>(rimg + eoff[it] + nr * 32) : zero4;
        if (mimg) mv[it] = *reinterpret_cast<const f32x4*>(mimg + eoff[it] + nr * 32);
      }
      turn_in(nr);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        f32x4 v = *reinterpret_cast<const f32x4*>(tbuf + (it * 8 + psub) * TPIX + cq * 16) * inv4 + bias4;
        if (mimg) v = mask4(v, mv[it]);
        v += rv[it];
        if (eok[it]) {
          if (yraw) *reinterpret_cast<f32x4*>(yraw + eoff[it] + nr * 32) = v;
          if (yact) *reinterpret_cast<f32x4*>(yact + eoff[it] + nr * 32) = lrelu4(v * mul4 + add4, a.slope);
        }
      }
    }
  }
#ifdef VIRNET_F16_TIMING
  TSTAMP(3);
  if (a.tlog && tid == 0) {
    a.tlog[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
    a.tlog[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
  }
#endif
}

template <int MREP, int NREP, int EPI, int BF = 0, int TE = 0, int ENT = 0>
int launch(FArgs k, hipStream_t st) {
  constexpr int TH = 4 * MREP;
  constexpr int LDS_K = 2 * (2 * (TH + 2) * 34 * 32) + 2 * (((NREP == 1 && !ENT) ? 9 : 3) * NREP * 2048);       // K loop: pixel tiles + weight stages (CS: whole chunks)
  constexpr int LDS_E = (EPI == 5) ? 0 : 4 * 2 * (MREP * 32 * 144 + (TE ? 128 : 0));   // epilogue: two turn-around regions per wave
  constexpr int LDS = LDS_K > LDS_E ? LDS_K : LDS_E;
  static unsigned long long attr_done = 0;
  auto kern = conv_f16_kernel<MREP, NREP, EPI, BF, TE, ENT>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_f16): %s", hipGetErrorString(e));
  }
  k.nty = (k.H + TH - 1) / TH;
  k.ntx = (k.W + 31) / 32;
  k.ntiles = k.N * k.nty * k.ntx;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  const int ncb = k.NP / (32 * NREP);
  const unsigned grid = (unsigned)(8 * k.tiles_per_xcd * ncb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), LDS, st, k);
  return virnet::check_launch("conv_f16 launch");
}

// ---- weight packing -------------------------------------------------------------------------------------------------------
// One block per GEMM row (output channel): power-of-two scale from the row's largest magnitude, then the split image.
// kind 0: forward OIHW; kind 2: the layer's input-gradient GEMM (rows = forward cin, contraction = forward cout, flipped taps).
// bf = 1: the bf16-operand image (one bf16 value in the hi plane, scale 1, lo plane zero).
__global__ void pack_f16_kernel(const float* __restrict__ w, int kind, int cout, int cin, int cin_pad, int n_pad,
                                float* __restrict__ inv_scale, char* __restrict__ img, int bf) {
  const int row = blockIdx.x;
  const int rows = kind == 2 ? cin : cout, ks = kind == 2 ? cout : cin;
  const int nch = cin_pad >> 4;
  auto wval = [&](int k, int dy, int dx) -> float {
    if (row >= rows || k >= ks) return 0.f;
    return kind == 2 ? w[(((size_t)k * cin + row) * 3 + (2 - dy)) * 3 + (2 - dx)] : w[(((size_t)row * cin + k) * 3 + dy) * 3 + dx];
  };
  __shared__ float red[256];
  float m = 0.f;
  for (int i = threadIdx.x; i < ks * 9; i += blockDim.x) m = fmaxf(m, fabsf(wval(i / 9, (i % 9) / 3, i % 3)));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  m = red[0];
  // largest scaled magnitude in [8192, 16384): two bits of headroom below fp16's 65504, low halves normal down to 2^-17 of it
  int e = 0;
  if (m > 0.f && !bf) { frexpf(m, &e); e = 14 - e; }
  e = max(-100, min(100, e));
  const float scale = ldexpf(1.f, e);
  if (threadIdx.x == 0) inv_scale[row] = ldexpf(1.f, -e);
  const int slab = row >> 5, col = row & 31;
  for (int i = threadIdx.x; i < cin_pad * 9; i += blockDim.x) {
    const int k = i / 9, t = i % 9, dy = t / 3, dx = t % 3;
    const float v = wval(k, dy, dx) * scale;
    const int chunk = k >> 4, kk = k & 15;
    const size_t base = ((((size_t)slab * nch + chunk) * 9 + (dx * 3 + dy)) * 2) * 1024 + (size_t)(col + 32 * (kk >> 3)) * 16 + (kk & 7) * 2;
    if (bf) {
      *reinterpret_cast<__bf16*>(img + base) = (__bf16)v;
      *reinterpret_cast<unsigned short*>(img + base + 1024) = 0;
    } else {
      const _Float16 hi = (_Float16)v;
      *reinterpret_cast<_Float16*>(img + base) = hi;
      *reinterpret_cast<_Float16*>(img + base + 1024) = (_Float16)(v - (float)hi);
    }
  }
}

}  // namespace

#ifdef VIRNET_F16_TIMING
static long long* g_tlog = nullptr;
extern "C" void virnet_debug_timing_buffer(void* p) { g_tlog = static_cast<long long*>(p); }
long long* virnet_f16_tlog() { return g_tlog; }
#endif

extern "C" size_t virnet_f16_weight_floats(int cin_pad, int n_pad) { return (size_t)n_pad + (size_t)n_pad * cin_pad * 9; }

extern "C" int virnet_pack_f16_weight(const float* w, int dgrad, int cout, int cin, int cin_pad, int n_pad, float* packed, void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_f16_weight: NULL pointer");
  VIRNET_REQUIRE(cout > 0 && cin > 0, "virnet_pack_f16_weight: bad extents cout=%d cin=%d", cout, cin);
  const int rows = dgrad ? cin : cout, ks = dgrad ? cout : cin;
  VIRNET_REQUIRE(cin_pad % 16 == 0 && cin_pad >= ks, "virnet_pack_f16_weight: cin_pad=%d does not cover %d contraction channels", cin_pad, ks);
  VIRNET_REQUIRE(n_pad % 32 == 0 && n_pad >= rows, "virnet_pack_f16_weight: n_pad=%d does not cover %d output channels", n_pad, rows);
  hipLaunchKernelGGL(pack_f16_kernel, dim3((unsigned)n_pad), dim3(256), 0, static_cast<hipStream_t>(stream), w, dgrad ? 2 : 0, cout, cin,
                     cin_pad, n_pad, packed, reinterpret_cast<char*>(packed + n_pad), 0);
  return virnet::check_launch("pack_f16 launch");
}

extern "C" int virnet_pack_bf16_weight(const float* w, int dgrad, int cout, int cin, int cin_pad, int n_pad, float* packed, void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_bf16_weight: NULL pointer");
  VIRNET_REQUIRE(cout > 0 && cin > 0, "virnet_pack_bf16_weight: bad extents cout=%d cin=%d", cout, cin);
  const int rows = dgrad ? cin : cout, ks = dgrad ? cout : cin;
  VIRNET_REQUIRE(cin_pad % 16 == 0 && cin_pad >= ks, "virnet_pack_bf16_weight: cin_pad=%d does not cover %d contraction channels", cin_pad, ks);
  VIRNET_REQUIRE(n_pad % 32 == 0 && n_pad >= rows, "virnet_pack_bf16_weight: n_pad=%d does not cover %d output channels", n_pad, rows);
  hipLaunchKernelGGL(pack_f16_kernel, dim3((unsigned)n_pad), dim3(256), 0, static_cast<hipStream_t>(stream), w, dgrad ? 2 : 0, cout, cin,
                     cin_pad, n_pad, packed, reinterpret_cast<char*>(packed + n_pad), 1);
  return virnet::check_launch("pack_bf16 launch");
}

static int conv_f16_impl(const virnet_conv_desc* d, void* stream, int bf, const virnet_t_emit* te = nullptr, const virnet_pack_desc* ent = nullptr);

extern "C" int virnet_conv_f16(const virnet_conv_desc* d, void* stream) { return conv_f16_impl(d, stream, 0); }

static bool f16_emit_shape_ok(const virnet_conv_desc* d) {
  return d && d->ks == 3 && d->stride == 1 && d->epi == VIRNET_EPI_NHWC && d->cout > 0 && d->cout % 32 == 0 && d->n_pad == d->cout &&
         !d->mul && ((d->y_raw != nullptr) != (d->y_act != nullptr));
}

extern "C" int virnet_conv_f16_emit(const virnet_conv_desc* d, const virnet_t_emit* te, int bf16_operands, void* stream) {
  VIRNET_REQUIRE(d != nullptr && te != nullptr && te->t_out != nullptr, "virnet_conv_f16_emit: NULL descriptor / T buffer");
  VIRNET_REQUIRE(f16_emit_shape_ok(d), "virnet_conv_f16_emit: T emission needs the stride-1 3x3 NHWC conv with ONE stored tensor and no output SFT");
  VIRNET_REQUIRE((bf16_operands != 0) == (te->bf16 != 0), "virnet_conv_f16_emit: the T image follows the operand form (bf16 with bf16 operands)");
  VIRNET_REQUIRE(!te->act || (te->slope >= 0.f && te->slope <= 1.f), "virnet_conv_f16_emit: slope=%g outside [0,1]", te->slope);
  return conv_f16_impl(d, stream, bf16_operands ? 1 : 0, te);
}

// bf16-operand variant of the stride-1 3x3 NHWC convolution (one product per MAC, fp32 accumulation): wpack from virnet_pack_bf16_weight
extern "C" int virnet_conv_bf16(const virnet_conv_desc* d, void* stream) {
  VIRNET_REQUIRE(d != nullptr, "virnet_conv_bf16: desc is NULL");
  VIRNET_REQUIRE(d->ks == 3 && d->stride == 1 && d->epi == VIRNET_EPI_NHWC, "virnet_conv_bf16: only the stride-1 3x3 NHWC conv (ks=%d stride=%d epi=%d)",
                 d->ks, d->stride, d->epi);
  return conv_f16_impl(d, stream, 1);
}

extern "C" int virnet_conv_f16_entry(const virnet_conv_desc* d, const virnet_pack_desc* e, void* stream) {
  VIRNET_REQUIRE(d != nullptr && e != nullptr && e->x != nullptr, "virnet_conv_f16_entry: NULL descriptor / image");
  VIRNET_REQUIRE(d->ks == 3 && d->stride == 1 && d->epi == VIRNET_EPI_NHWC && d->cin_pad == 16, "virnet_conv_f16_entry: the stride-1 3x3 NHWC conv on ONE 16-channel chunk (cin_pad=%d)", d->cin_pad);
  VIRNET_REQUIRE(!d->res && !d->mask && !d->mul && !d->in_mul && !d->in_act && ((d->y_raw != nullptr) != (d->y_act != nullptr)),
                 "virnet_conv_f16_entry: plain single-store epilogue only (no residual / mask / SFT / input activation)");
  VIRNET_REQUIRE(e->n == d->n && e->hp == d->h && e->wp == d->w, "virnet_conv_f16_entry: the entry %d x %dx%d does not match the conv input %d x %dx%d",
                 e->n, e->hp, e->wp, d->n, d->h, d->w);
  VIRNET_REQUIRE(e->c0 >= 1 && e->ev >= 0 && e->em >= 0 && e->c0 + e->ev + e->em <= 8, "virnet_conv_f16_entry: %d+%d+%d channels (the fused entry holds 8)", e->c0, e->ev, e->em);
  VIRNET_REQUIRE(e->h > 0 && e->w > 0 && e->sf >= 1 && e->hp >= e->h * e->sf && e->wp >= e->w * e->sf && e->hp - e->h * e->sf < e->h * e->sf && e->wp - e->w * e->sf < e->w * e->sf,
                 "virnet_conv_f16_entry: bad geometry %dx%d x%d -> %dx%d (reflect pad must stay below the image size)", e->h, e->w, e->sf, e->hp, e->wp);
  VIRNET_REQUIRE(e->ev == 0 || e->vec, "virnet_conv_f16_entry: ev=%d without vec", e->ev);
  VIRNET_REQUIRE(e->em == 0 || (e->map && e->msf >= 1), "virnet_conv_f16_entry: em=%d without map/msf", e->em);
  virnet_conv_desc dd = *d;
  dd.x = e->x;                                             // (never dereferenced as NHWC: the staging gathers through `ent`)
  return conv_f16_impl(&dd, stream, 0, nullptr, e);
}

static int conv_f16_impl(const virnet_conv_desc* d, void* stream, int bf, const virnet_t_emit* te, const virnet_pack_desc* ent) {
  VIRNET_REQUIRE(d != nullptr, "virnet_conv_f16: desc is NULL");
  VIRNET_REQUIRE(d->x && d->wpack, "virnet_conv_f16: x / wpack is NULL");
  if (d->ks == 1 && d->epi == VIRNET_EPI_CONVT) {                // UpBlock.upsampler + bridge (AttResUNet.py:80,84-87): conv_f16_pw.hip
    VIRNET_REQUIRE(d->stride == 1 && d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv_f16: bad transposed-conv shape");
    VIRNET_REQUIRE(d->cout > 0 && d->cout % 32 == 0 && d->n_pad == 4 * d->cout, "virnet_conv_f16: transposed conv needs cout %% 32 == 0 and n_pad = 4*cout (cout=%d n_pad=%d)", d->cout, d->n_pad);
    VIRNET_REQUIRE(d->cin_pad >= 16 && d->cin_pad % 16 == 0, "virnet_conv_f16: cin_pad=%d is not a multiple of 16", d->cin_pad);
    VIRNET_REQUIRE((d->y_raw != nullptr) != (d->y_act != nullptr) && !d->mask && !d->mul && !d->in_mul, "virnet_conv_f16: the transposed form has the bias + bridge / single-store epilogue only");
    VIRNET_REQUIRE((long)d->h * d->w * d->cout * 16 * d->n < (1L << 40), "virnet_conv_f16: output too large");
    FArgs t{};
    t.x = d->x; t.inv_scale = d->wpack; t.wimg = reinterpret_cast<const char*>(d->wpack + d->n_pad);
    t.bias = d->bias; t.res = d->res; t.y_raw = d->y_raw; t.y_act = d->y_act;
    t.N = d->n; t.H = d->h; t.W = d->w; t.cout = d->cout; t.in_act = d->in_act; t.in_slope = d->in_slope; t.slope = d->slope;
    t.range_flag = virnet::range_flag_ptr();
    t.store_nt = virnet::store_nt_for((size_t)d->n * d->h * d->w * 4 * d->cout * 4);
    return virnet::launch_f16_convt(t, d->cin_pad, static_cast<hipStream_t>(stream));
  }
  VIRNET_REQUIRE(d->ks == 3 && ((d->stride == 1 && (d->epi == VIRNET_EPI_NHWC || d->epi == VIRNET_EPI_NCHW)) || (d->stride == 2 && d->epi == VIRNET_EPI_NHWC)),
                 "virnet_conv_f16: 3x3 conv, stride 1 (NHWC or planar store) or stride 2 (NHWC) (ks=%d stride=%d epi=%d)", d->ks, d->stride, d->epi);
  VIRNET_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv_f16: empty input n=%d h=%d w=%d", d->n, d->h, d->w);
  VIRNET_REQUIRE(d->cin_pad >= 16 && d->cin_pad % 16 == 0, "virnet_conv_f16: cin_pad=%d is not a multiple of 16", d->cin_pad);
  if (d->epi == VIRNET_EPI_NCHW) {
    VIRNET_REQUIRE(d->cout >= 1 && d->cout <= 32 && d->n_pad == 32, "virnet_conv_f16: planar store handles 1..32 channels (cout=%d n_pad=%d)", d->cout, d->n_pad);
    VIRNET_REQUIRE(d->y_raw && !d->y_act && !d->mask && !d->mul, "virnet_conv_f16: planar store takes y_raw only");
    VIRNET_REQUIRE(d->crop_h >= 1 && d->crop_h <= d->h && d->crop_w >= 1 && d->crop_w <= d->w, "virnet_conv_f16: crop %dx%d outside output %dx%d",
                   d->crop_h, d->crop_w, d->h, d->w);
    VIRNET_REQUIRE(d->nchw_op != VIRNET_NCHW_ADD || d->res, "virnet_conv_f16: VIRNET_NCHW_ADD without res");
    VIRNET_REQUIRE(d->res_sf <= 1 || (d->crop_h % d->res_sf == 0 && d->crop_w % d->res_sf == 0), "virnet_conv_f16: crop %dx%d is not a multiple of res_sf=%d",
                   d->crop_h, d->crop_w, d->res_sf);
  } else {
    VIRNET_REQUIRE(d->cout > 0 && d->cout % 32 == 0 && d->n_pad == d->cout, "virnet_conv_f16: cout=%d must be a multiple of 32 (n_pad=%d)", d->cout, d->n_pad);
  }
  VIRNET_REQUIRE(d->y_raw || d->y_act, "virnet_conv_f16: no output pointer");
  VIRNET_REQUIRE((long)d->h * d->w * d->n_pad * 4 < (1L << 31), "virnet_conv_f16: one image's output (%d x %d x %d fp32) must stay below 2 GB", d->h, d->w, d->n_pad);
  VIRNET_REQUIRE((d->in_mul == nullptr) == (d->in_add == nullptr), "virnet_conv_f16: in_mul and in_add must be given together");
  VIRNET_REQUIRE(d->in_act || !d->in_mul, "virnet_conv_f16: in_mul/in_add without in_act");
  VIRNET_REQUIRE(!d->in_act || (d->in_slope >= 0.f && d->in_slope <= 1.f), "virnet_conv_f16: in_slope=%g outside [0,1]", d->in_slope);
  VIRNET_REQUIRE(!d->y_act || (d->slope >= 0.f && d->slope <= 1.f), "virnet_conv_f16: slope=%g outside [0,1]", d->slope);
  FArgs k{};
  k.x = d->x; k.inv_scale = d->wpack; k.wimg = reinterpret_cast<const char*>(d->wpack + d->n_pad);
  k.bias = d->bias; k.res = d->res; k.mul = d->mul; k.add = d->add;
  k.in_mul = d->in_mul; k.in_add = d->in_add; k.mask = d->mask; k.y_raw = d->y_raw; k.y_act = d->y_act;
  k.N = d->n; k.H = d->h; k.W = d->w; k.Cin = d->cin_pad; k.NP = d->n_pad; k.cout = d->cout;
  k.in_act = d->in_act; k.in_slope = d->in_slope; k.mask_slope = d->mask_slope; k.slope = d->slope;
  k.nchw_op = d->nchw_op; k.crop_h = d->crop_h; k.crop_w = d->crop_w; k.res_sf = d->res_sf; k.clamp_lo = d->clamp_lo; k.clamp_hi = d->clamp_hi;
#ifdef VIRNET_F16_TIMING
  k.tlog = g_tlog;
#endif
  hipStream_t st = static_cast<hipStream_t>(stream);
  k.range_flag = bf ? nullptr : virnet::range_flag_ptr();
  k.OH = d->h; k.OW = d->w;
  if (d->stride == 2) {                                         // DownBlock.downsampler (AttResUNet.py:67): conv_f16_s2.hip
    VIRNET_REQUIRE(d->h % 2 == 0 && d->w % 2 == 0, "virnet_conv_f16: stride-2 input %dx%d must be even", d->h, d->w);
    VIRNET_REQUIRE(!d->res && !d->mask && !d->mul && !d->in_mul && !(d->y_raw && d->y_act),
                   "virnet_conv_f16: the stride-2 form has the bias / single-store epilogue only");
    k.OH = d->h / 2; k.OW = d->w / 2;
    k.store_nt = virnet::store_nt_for((size_t)d->n * k.OH * k.OW * d->n_pad * 4);
    return virnet::launch_f16_s2(k, d->n_pad / 32, st);
  }
  const int nb = d->n_pad / 32;
  const long tiles8 = (long)d->n * ((d->h + 7) / 8) * ((d->w + 31) / 32);
  const char* const env_m = getenv("VIRNET_F16_MREP");      // tuning / tests (read per call)
  const int forced_m = env_m ? atoi(env_m) : 0;
  if (d->epi == VIRNET_EPI_NCHW) {
    const int mrep = (forced_m == 1 || forced_m == 2) ? forced_m : (tiles8 >= 1024 ? 2 : 1);
    k.slab_base = 0; k.NP = 32;
    return mrep == 2 ? launch<2, 1, 5>(k, st) : launch<1, 1, 5>(k, st);
  }
  const int epi = (d->mul || (d->y_raw && d->y_act)) ? 4 : (d->res ? 1 : 0) | (d->mask ? 2 : 0);
  // Slabs per workgroup: 3 where the count allows, the remainder in 2s (160 channels = 3 + 2, 224 = 3 + 2 + 2: two launches, each
  // staging the pixel tile once per workgroup, instead of 5 / 7 single-slab workgroups per tile); a lone odd slab runs by itself.
  int n3 = nb / 3, rem = nb - 3 * n3;
  if (rem == 1 && n3 >= 1) { n3 -= 1; rem = 4; }
  int n2 = rem / 2, n1 = rem - 2 * n2;
  // Launches far from filling the chip (deep levels of single images: a 64x64 x 288-channel conv is 32 tiles x 3 channel blocks = 96
  // workgroups of 18 chunks each): one slab per workgroup triples the grid, shortens every workgroup and puts channel counts that are
  // not multiples of 96 (160 = 3 + 2, 224 = 3 + 2 + 2) into ONE launch.  No result bit depends on the grouping.  VIRNET_F16_SPLIT_WGS:
  // the largest 3-slab grid that is split (0 = never).
  {
    const char* const env_s = getenv("VIRNET_F16_SPLIT_WGS");
    // A MIXED grouping (160 = 3 + 2 slabs) is two launches one after the other, each on half of the chip when it has ~128 workgroups:
    // there the split pays up to twice the grid (SISR x4, one image, 160-channel level: 2 x 128 workgroups in 58 us -> 640 in ~30;
    // the forward 1.21 -> 1.08 ms, profiles/r05_probes.md 11)
    const bool mixed = (n3 > 0) + (n2 > 0) + (n1 > 0) >= 2;
    const long split_below = env_s ? atol(env_s) : (mixed ? 256 : 128);
    const long tiles4 = (long)d->n * ((d->h + 3) / 4) * ((d->w + 31) / 32);
    if (nb > 1 && tiles4 * (n3 + n2 + n1) <= split_below && !te) { n3 = 0; n2 = 0; n1 = nb; }
  }
  if (te) virnet::t_emit_args(k, te, d->w, d->cout, (int)(tiles8 * 4));      // (emission runs on 8-row tiles whatever the grid: 4 waves per tile)
  if (ent) { k.ent = *ent; k.ent.out = nullptr; }
  auto run = [&](int nrep, int slab_base, int groups) -> int {
    if (groups <= 0) return 0;
    FArgs kk = k;
    kk.slab_base = slab_base;
    kk.NP = groups * nrep * 32;
    int mrep = (tiles8 * groups >= 1024) ? 2 : 1;
    if (forced_m == 1 || forced_m == 2) mrep = forced_m;
    if (ent) {
      if (epi != 0 || bf) return virnet::set_error("virnet_conv_f16_entry: epilogue %d / bf16 operands have no entry form", epi);
#define VIRNET_F16_ENT(M_, N_) if (mrep == M_ && nrep == N_) return launch<M_, N_, 0, 0, 0, 1>(kk, st);
      VIRNET_F16_ENT(2, 3) VIRNET_F16_ENT(2, 2) VIRNET_F16_ENT(2, 1) VIRNET_F16_ENT(1, 3) VIRNET_F16_ENT(1, 2) VIRNET_F16_ENT(1, 1)
#undef VIRNET_F16_ENT
    }
    if (te) {
#define VIRNET_F16_TE(N_, E_) if (nrep == N_ && epi == E_) return bf ? launch<2, N_, E_, 1, 1>(kk, st) : launch<2, N_, E_, 0, 1>(kk, st);
#define VIRNET_F16_TEN(N_) VIRNET_F16_TE(N_, 0) VIRNET_F16_TE(N_, 1) VIRNET_F16_TE(N_, 2) VIRNET_F16_TE(N_, 3)
      VIRNET_F16_TEN(3) VIRNET_F16_TEN(2) VIRNET_F16_TEN(1)
#undef VIRNET_F16_TEN
#undef VIRNET_F16_TE
      return virnet::set_error("virnet_conv_f16_emit: no emitting kernel for nrep=%d epi=%d", nrep, epi);
    }
#define VIRNET_F16_CASE(M_, N_)                                          \
    if (mrep == M_ && nrep == N_) {                                      \
      if (bf) {                                                          \
        if (epi == 0) return launch<M_, N_, 0, 1>(kk, st);               \
        if (epi == 1) return launch<M_, N_, 1, 1>(kk, st);               \
        if (epi == 2) return launch<M_, N_, 2, 1>(kk, st);               \
        if (epi == 3) return launch<M_, N_, 3, 1>(kk, st);               \
        return launch<M_, N_, 4, 1>(kk, st);                             \
      }                                                                  \
      if (epi == 0) return launch<M_, N_, 0>(kk, st);                    \
      if (epi == 1) return launch<M_, N_, 1>(kk, st);                    \
      if (epi == 2) return launch<M_, N_, 2>(kk, st);                    \
      if (epi == 3) return launch<M_, N_, 3>(kk, st);                    \
      return launch<M_, N_, 4>(kk, st);                                  \
    }
    VIRNET_F16_CASE(2, 3) VIRNET_F16_CASE(2, 2) VIRNET_F16_CASE(2, 1)
    VIRNET_F16_CASE(1, 3) VIRNET_F16_CASE(1, 2) VIRNET_F16_CASE(1, 1)
#undef VIRNET_F16_CASE
    return virnet::set_error("virnet_conv_f16: no kernel for mrep=%d nrep=%d", mrep, nrep);
  };
  if (int rc = run(3, 0, n3)) return rc;
  if (int rc = run(2, 3 * n3, n2)) return rc;
  return run(1, 3 * n3 + 2 * n2, n1);
}

// form 0: conv_f16 / conv_bf16 (8-row tiles, 4 waves); form 1: conv_wx4 (16-row tiles, 8 waves); form 2: conv_wx4, 8-row tiles (4 waves)
extern "C" int virnet_conv_emit_ok(const virnet_conv_desc* d, int form, int* nblk) {
  if (!f16_emit_shape_ok(d) || form < 0 || form > 2) return 0;
  if (form >= 1 && (d->in_mul || d->cin_pad < 32)) return 0;
  const long th = form == 1 ? 16 : 8, nw = form == 1 ? 8 : 4;
  const long blocks = (long)d->n * ((d->h + th - 1) / th) * ((d->w + 31) / 32) * nw;
  if (blocks >= (1L << 30)) return 0;
  if (nblk) *nblk = (int)blocks;
  return 1;
}
