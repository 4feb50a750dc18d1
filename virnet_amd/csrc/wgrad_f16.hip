// wgrad_f16.hip -- weight gradient of the stride-1 3x3 convolutions on the f16 matrix pipe (training step, SURVEY.md 8-f1:
// the backward of networks/AttResUNet.py:43,46,117,139 and networks/DnCNN.py:22-29, train_denoising_syn.py:176-179).
//
//   dW[co][ci][ky][kx] = sum over pixels p of dY[p][co] * A[p + (ky-1, kx-1)][ci]          (A = the forward conv's staged input)
// contracts over PIXELS, so an MFMA k-step is 16 consecutive pixels of a row and both operands must be CHANNEL-major (8 consecutive
// pixels of one channel per lane), the transpose of the NHWC tensors the rest of the path uses.  Three kernels:
//   1. chsplit_kernel: NHWC fp32 -> channel-major fp16 planes T[n][h+2][c/32][hi|lo][seg][32 ch][8 px] (pixel x at index x+8, one
//      zero row above and below, zero pads left and right), applying the forward conv's staging transform (lrelu(x*mul+add)) and the
//      hi/lo split of conv_f16.hip on the way (bf16 variant: one bf16 plane).  Bandwidth-bound; the transpose goes through LDS.
//   2. conv_wgrad_f16_kernel: one workgroup per CU = NWV (<= 3) output-channel blocks x 4 k-step waves, ONE input-channel block, each
//      wave pair keeping all nine taps of its (co, ci) pair in registers (6 accumulator tiles = 96 VGPRs per wave: the even wave owns tap
//      rows {0,1}, the odd wave {2,1}, each does both of its rows on its own k-step and its outer row on the partner's).  A step = one
//      image row x 64 pixels; the
//      dY segments and the one new A row of a step are contiguous runs of T and come in by LDS-DMA with NO staging arithmetic, two
//      steps ahead; LDS image [seg][32 ch][16 B] -> conflict-free ds_read_b128 fragments.  The kx = 0 / 2 taps are the aligned window
//      shifted by one pixel = 2 bytes: v_alignbyte over the window's 4 + 2 dwords.  Three products per k-step as in conv_f16.hip
//      (bf16: one).  The pixel range is split across workgroups; their partial sums go to a scratch tensor with plain stores.
//   3. wgrad_reduce_kernel adds the partial sums in a fixed order and writes dW in OIHW: no atomics, bitwise reproducible.
// The bias gradient (sum of dY over pixels) is a by-product of pass 1 over dY (per-block channel sums + colpart_reduce_kernel).
// Accuracy: as conv_f16.hip for operands above fp16's subnormal range; gradients far below 6e-5 in magnitude lose relative precision
// (absolute error <= 3e-8 per element) -- DESIGN.md 6.
#include "conv_f16_common.h"
#include <cstdlib>

namespace {
using namespace virnet;

struct TGeom {
  int n, h, w, c;      // source tensor (c = stored channels)
  int cb;              // 32-channel blocks of T
  int nseg;            // 8-pixel segments per row of T
  int par2;            // column-phase mode (stride-2 layers): the source is [n][h][2w][c]; T has 2*ceil(c/32) blocks per row, block
                       // par*cb/2 + k holding the pixels x = 2*ox + par of source block k (T pixel ox), so that the taps of a stride-2
                       // window are aligned (even columns) or shifted by one T pixel (odd columns)
};

// segments per row of T: the pixels rounded up to whole steps (64 pixels; 32 for images of at most 32 columns) + one pad segment each side
__host__ __device__ inline int t_nseg(int w) { return w <= 32 ? 6 : 8 * ((w + 63) / 64) + 2; }

// ---- 1. NHWC fp32 -> T -------------------------------------------------------------------------------------------------------------
// block = (image, padded row, channel block, group of 8 segments); 256 threads
template <int BF>
__global__ __launch_bounds__(256) void chsplit_kernel(const float* __restrict__ x, const float* __restrict__ in_mul, const float* __restrict__ in_add,
                                                      int in_act, float in_slope, TGeom g, unsigned short* __restrict__ out,
                                                      float* __restrict__ colpart) {
  __shared__ unsigned short tile[2][64][34];             // [plane][T index in group][channel (+2 pad)]
  __shared__ float red[32][33];                          // per-pixel-thread channel sums (bias gradient by-product)
  const int sgs = (g.nseg + 7) / 8;
  int b = blockIdx.x;
  const int sg = b % sgs; b /= sgs;
  const int cb = b % g.cb; b /= g.cb;
  const int prow = b % (g.h + 2);
  const int img = b / (g.h + 2);
  const int tid = threadIdx.x;
  const int q = tid & 7;                                 // channel quad
  const int cbh = g.par2 ? g.cb / 2 : g.cb;              // source channel blocks
  const int par = g.par2 ? cb / cbh : 0;
  const int c0 = (cb - par * cbh) * 32 + q * 4;
  const int sw = g.par2 ? 2 * g.w : g.w;                 // source row length in pixels
  const float slope = in_act ? in_slope : 1.f;
  f32x4 m4 = f32x4{1.f, 1.f, 1.f, 1.f}, a4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (in_mul && c0 < g.c) {
    m4 = *reinterpret_cast<const f32x4*>(in_mul + (size_t)img * g.c + c0);
    a4 = *reinterpret_cast<const f32x4*>(in_add + (size_t)img * g.c + c0);
  }
  f32x4 csum = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int ti = it * 32 + (tid >> 3);                 // T index inside the group
    const int px = sg * 64 + ti - 8;
    const bool ok = prow >= 1 && prow <= g.h && px >= 0 && px < g.w && c0 < g.c;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ok) {
      v = *reinterpret_cast<const f32x4*>(x + (((size_t)img * g.h + (prow - 1)) * sw + (g.par2 ? 2 * px + par : px)) * g.c + c0);
      v = lrelu4(v * m4 + a4, slope);
    }
    csum += v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (BF) {
        const __bf16 hb = (__bf16)v[e];
        tile[0][ti][q * 4 + e] = __builtin_bit_cast(unsigned short, hb);
      } else {
        const _Float16 hi = (_Float16)v[e];
        const _Float16 lo = (_Float16)(v[e] - (float)hi);
        tile[0][ti][q * 4 + e] = __builtin_bit_cast(unsigned short, hi);
        tile[1][ti][q * 4 + e] = __builtin_bit_cast(unsigned short, lo);
      }
    }
  }
  if (colpart) {
#pragma unroll
    for (int e = 0; e < 4; ++e) red[tid >> 3][q * 4 + e] = csum[e];
  }
  __syncthreads();
  if (colpart && tid < 32) {                             // this block's sum over its 64 pixels, per channel: plain store, reduced later
    float t = 0.f;
#pragma unroll 8
    for (int p2 = 0; p2 < 32; ++p2) t += red[p2][tid];
    const size_t nblk = (size_t)g.n * (g.h + 2) * sgs;
    const size_t blk = ((size_t)img * (g.h + 2) + prow) * sgs + sg;
    colpart[((size_t)cb * nblk + blk) * 32 + tid] = t;
  }
  const int seg = tid >> 5, ch = tid & 31;               // 8 segments x 32 channels
  const int gseg = sg * 8 + seg;
  if (gseg < g.nseg) {
#pragma unroll
    for (int plane = 0; plane < (BF ? 1 : 2); ++plane) {
      unsigned short v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[plane][seg * 8 + e][ch];
      u32x4 pk = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16), (unsigned)v[4] | ((unsigned)v[5] << 16),
                       (unsigned)v[6] | ((unsigned)v[7] << 16)};
      const size_t o = ((((size_t)img * (g.h + 2) + prow) * g.cb + cb) * 2 + plane) * g.nseg + gseg;
      *reinterpret_cast<u32x4*>(out + (o * 32 + ch) * 8) = pk;
    }
  }
}

// db[cb*32 + ch] += sum over blocks of colpart[cb][blk][ch]; grid (cb, slices)
__device__ __forceinline__ void colpart_reduce_block(const float* __restrict__ part, float* __restrict__ db, long nblk, int cvalid, int cbh,
                                                     int cb, int slice, int nslices, float (*red)[32]) {
  const int j = threadIdx.x >> 5, ch = threadIdx.x & 31;
  float t = 0.f;
  for (long blk = (long)slice * 8 + j; blk < nblk; blk += (long)nslices * 8) t += part[((size_t)cb * nblk + blk) * 32 + ch];
  red[j][ch] = t;
  __syncthreads();
  if (threadIdx.x < 32) {
    float u = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) u += red[k][threadIdx.x];
    const int c = (cb % cbh) * 32 + threadIdx.x;
    if (c < cvalid) atomicAdd(db + c, u);
  }
}

__global__ __launch_bounds__(256) void colpart_reduce_kernel(const float* __restrict__ part, float* __restrict__ db, long nblk, int cvalid, int cbh) {
  __shared__ float red[8][32];
  const int cb = blockIdx.x, j = threadIdx.x >> 5, ch = threadIdx.x & 31;
  float t = 0.f;
  for (long blk = (long)blockIdx.y * 8 + j; blk < nblk; blk += (long)gridDim.y * 8) t += part[((size_t)cb * nblk + blk) * 32 + ch];
  red[j][ch] = t;
  __syncthreads();
  if (threadIdx.x < 32) {
    float u = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) u += red[k][threadIdx.x];
    const int c = (cb % cbh) * 32 + threadIdx.x;         // (phase mode: both column phases of a channel block add into it)
    if (c < cvalid) atomicAdd(db + c, u);
  }
}

// ---- 2. the GEMM ----------------------------------------------------------------------------------------------------------------
struct GArgs {
  const char* xt;   // T of the forward input   [n][h+2][ncib][2][nseg][32][8] fp16
  const char* yt;   // T of the output gradient [n][h+2][ncob][2][nseg][32][8]
  float* dw;        // partial sums part[run][9][ncob*32][ncib*32] (every element written: no zeroing, no atomics)
  int n, h, w, nseg;
  int cin, cout, ncib, ncob;
  int nsteps, nxs, nsplit, npairs, run;
  long long* tlog;    // -DVIRNET_F16_TIMING builds: per-workgroup cycle sums (tools/wgrad_timeline.py)
};

// Workgroup = NWV output-channel blocks x KG k-steps = KG*NWV waves (twelve for the 96/192/288-channel layers: exactly three per
// SIMD -- two six-wave workgroups per CU left most CUs with a 2/3/3/4 split), ONE input-channel block, one workgroup per CU.
// A step = one image row x 16*KG pixels, one MFMA k-step of 16 pixels per wave.  Steps walk DOWN a column strip, so of the three A
// rows a step reads only one is new.  Operands arrive by LDS-DMA DIST steps ahead of their use (a DMA round trip is ~3 us, a step
// ~1.5-2 us): a ring of DIST+5 A-row slots and DIST+1 dY stages; every wave issues the SAME number of 1-KB pieces per step so the wait
// before a step is a literal `s_waitcnt vmcnt(pieces of the later steps)` and never drains the prefetch.
template <int BF, int KG, int S = 1> struct WgCfg {
  static constexpr int NPL = BF ? 1 : 2;
  static constexpr int DIST = BF ? 4 : 2;
  // rows alive while step t is computed: its own three + S new ones for each of the DIST requested steps + (3 - S) more when one of
  // those starts a strip (it brings three rows)
  static constexpr int RING = 3 + DIST * S + (3 - S), STAGES = DIST + 1;      // (S = 1: DIST + 5)
  static constexpr int XPL = (2 * KG + 2) * 512, XROW = NPL * XPL;   // one A row: [plane][seg 2KG+2][32 ch][16 B]
  static constexpr int YPL = 2 * KG * 512, YW = NPL * YPL;          // dY of one channel block: [plane][seg 2KG][32 ch][16 B]
  static constexpr int lds(int nwv) { return RING * XROW + STAGES * nwv * YW; }
};

// a wave-uniform pointer, pinned to scalar registers so that `p + lane offset` takes the saddr + 32-bit voffset form (one VGPR)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v));
  const unsigned hi = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(v >> 32));
  return reinterpret_cast<const char*>((static_cast<unsigned long long>(hi) << 32) | lo);
}

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

#ifdef VIRNET_F16_TIMING
#define WT_NOW() ((long long)__builtin_amdgcn_s_memtime())
#define WT_ADD(var, t0v) do { const long long n_ = WT_NOW(); var += n_ - (t0v); (t0v) = n_; } while (0)
#else
#define WT_NOW() 0ll
#define WT_ADD(var, t0v) do { } while (0)
#endif

// S = 2, DXM: the stride-2 layers (DownBlock.downsampler AttResUNet.py:67, UpBlock.upsampler :80).  The A operand is a column-phase T
// (chsplit par2) of the HIGH-resolution tensor, the dY operand a plain T of the low-resolution one; a step (low-res row oy) reads the
// high-res rows 2oy-1, 2oy, 2oy+1 -- TWO new ring rows per step instead of one -- and DXM masks the column shifts that are computed
// (bit dx: shift dx-1): a stride-2 window never needs the +1 shift (0b011), the 2x2 transposed conv only the aligned one (0b010).
// Which (tap row, shift, column phase) is which weight tap is the reduction kernel's business (wgrad_reduce_s2_kernel).
template <int NWV, int BF, int KG, int S = 1, int DXM = 7>
__global__ __launch_bounds__(64 * KG * NWV) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_wgrad_f16_kernel(const GArgs a) {
  using Cfg = WgCfg<BF, KG, S>;
  constexpr int NW = KG * NWV, NPL = Cfg::NPL, DIST = Cfg::DIST, RING = Cfg::RING, STAGES = Cfg::STAGES;
  constexpr int XPL = Cfg::XPL, XROW = Cfg::XROW, YPL = Cfg::YPL, YW = Cfg::YW, YST = NWV * YW;
  constexpr int XQ = KG + 1;                             // 1-KB pieces (two segments) of one plane of an A row
  constexpr int NYQ = NWV * NPL * KG;                    // dY pieces of a step
  constexpr int NRQ = NPL * S * XQ + NYQ, NFQ = NPL * 3 * XQ + NYQ;   // pieces of a normal step (S new rows) / of the first step of a strip
  constexpr int PN = (NRQ + NW - 1) / NW, PF = (NFQ + NW - 1) / NW;   // ... per wave (the round-up repeats a piece)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const xs = smem;                                 // RING row slots
  char* const ys = smem + RING * XROW;                   // STAGES stages

  // workgroups that walk the same steps (same run, every (co group, ci block) pair) are made neighbours on ONE XCD, so the tiles they
  // share come from that XCD's L2 after the first reader: hardware block id -> XCD = id % 8
  const int total = a.npairs * a.nsplit;
  const int xcd = blockIdx.x & 7, q8 = total >> 3, r8 = total & 7;
  const int vid = xcd * q8 + min(xcd, r8) + (blockIdx.x >> 3);   // XCD x holds the virtual ids [x*q8 + min(x, r8), ...) in dispatch order
  const int pair = vid % a.npairs;                       // (co group, ci block)
  const int runi = vid / a.npairs;
  const int cgrp = pair / a.ncib, cib = pair - cgrp * a.ncib;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cw = wv % NWV, kg = wv / NWV;
  const int l31 = lane & 31, lhi = lane >> 5;
  const unsigned lane16 = lane * 16;
  const int cob = cgrp * NWV + cw;
  const bool active = cob < a.ncob;
  const size_t plx = (size_t)a.nseg * 512, rowx = (size_t)a.ncib * 2 * plx, rowy = (size_t)a.ncob * 2 * plx;

  // SIX accumulator tiles per wave, not nine.  The KG waves of a channel block work in pairs on two neighbouring k-steps: the even
  // wave owns tap rows {0, 1}, the odd wave rows {2, 1}; on its OWN k-step a wave does both of its rows, on its partner's k-step only
  // its outer row (0 or 2) -- every (row, k-step) is done once, each wave still issues 9 of the pair's 18 (row, k-step, dx-triple)
  // units, and 96 accumulator registers instead of 144 leave room to read window i+1 while the MFMAs of window i issue.
  // Tile L0 (acc[0..2]) = outer row r0 = 0 | 2, tile L1 (acc[3..5]) = row 1.  The partial sums of a tap row meet in the epilogue.
  const int par = kg & 1;
  const int km = kg, ko = kg ^ 1;                          // own / partner k-step
  const int r0 = 2 * par;
  f32x16 acc[6];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  [[maybe_unused]] long long tw = 0, ti = 0, tc = 0, tmark = WT_NOW();
  [[maybe_unused]] const long long tstart = tmark;
  const int t0 = runi * a.run, t1 = min(t0 + a.run, a.nsteps);
  if (t0 < t1) {
    // issue cursor (the step whose operands are requested next) and consume cursor (the step computed next)
    int iu = t0, iy = t0 % a.h, istrip = t0 / a.h, icnt = 0;
    int cy = iy, ccnt = 0;                               // ccnt: ring position of the consumed step's first row

    // Row pointers of the issue cursor (wave-uniform): A row iy of this ci block / dY row iy of the group's first block, at the strip's
    // first segment.  They advance by one row per step; only a strip change recomputes them.
    const char* xrow = nullptr;
    const char* yrow = nullptr;
    auto seek = [&]() {
      const int img = istrip / a.nxs, xsi = istrip - img * a.nxs;
      xrow = uniform_ptr(a.xt + ((size_t)img * (S * a.h + 2) + S * iy) * rowx + (size_t)cib * 2 * plx + (size_t)xsi * 2 * KG * 512);
      yrow = uniform_ptr(a.yt + ((size_t)img * (a.h + 2) + iy + 1) * rowy + (size_t)(xsi * 2 * KG + 1) * 512);
    };
    seek();
    // The pieces this wave issues on a NORMAL step never change (piece q = i * NW + wv): decode them once.
    unsigned nsrc[PN], ndst[PN];                         // source offset from xrow + 2 rows / yrow; LDS offset inside the slot / stage
    bool nisx[PN];
    [[maybe_unused]] bool nrr[PN];                       // S = 2: which of the two new rows
#pragma unroll
    for (int i = 0; i < PN; ++i) {
      int q = i * NW + wv;
      if (q >= NRQ) q -= NYQ;                            // round-up: repeat a dY piece
      nisx[i] = q < NPL * S * XQ;
      nrr[i] = false;
      if (nisx[i]) {
        const int rr = q / (NPL * XQ), q1 = q - rr * (NPL * XQ);
        const int plane = q1 / XQ, s2 = q1 - plane * XQ;
        nrr[i] = rr != 0;
        nsrc[i] = (unsigned)(plane * plx + s2 * 1024);
        ndst[i] = plane * XPL + s2 * 1024;
      } else {
        const int qy = q - NPL * S * XQ;
        const int w2 = qy / (NPL * KG), rem = qy - w2 * (NPL * KG), plane = rem / KG, s2 = rem - plane * KG;
        const int cb2 = min(cgrp * NWV + w2, a.ncob - 1);
        nsrc[i] = (unsigned)(((size_t)cb2 * 2 + plane) * plx + s2 * 1024);
        ndst[i] = RING * XROW + w2 * YW + plane * YPL + s2 * 1024;
      }
    }

    // requests the operands of step `iu`; every wave issues exactly PF (first step of a strip / of the run) or PN pieces
    auto issue = [&](bool run_start) -> bool {
      const bool first = run_start || iy == 0;
      const unsigned ystage = ((iu - t0) % STAGES) * YST;
      if (!first) {
        // the common case: one new A row (padded row iy + 2) + the step's dY; a handful of scalar instructions per piece
        const char* const xnew = xrow + (3 - S) * rowx;        // (S = 2: padded rows 2iy+1 and 2iy+2; row 2iy came with the previous step)
        const char* const xnew2 = xrow + 2 * rowx;
        const unsigned xslot = (icnt % RING) * XROW, xslot2 = ((icnt + 1) % RING) * XROW;
#pragma unroll
        for (int i = 0; i < PN; ++i) {
          const char* const src = (nisx[i] ? (S == 2 && nrr[i] ? xnew2 : xnew) : yrow) + nsrc[i];
          const unsigned dst = ndst[i] + (nisx[i] ? (S == 2 && nrr[i] ? xslot2 : xslot) : ystage);
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + lane16),
                                           (__attribute__((address_space(3))) void*)(smem + dst), 16, 0, 0);
        }
        icnt += S;
      } else {
        char* const ydst = ys + ystage;
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          int q = i * NW + wv;
          if (q >= NFQ) q -= NYQ;                        // round-up: repeat a dY piece
          if (q < NPL * 3 * XQ) {
            const int rr = q / (NPL * XQ), rem = q - rr * (NPL * XQ), plane = rem / XQ, s2 = rem - plane * XQ;
            const int slot = (icnt + rr) % RING;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uniform_ptr(xrow + (size_t)rr * rowx + plane * plx + s2 * 1024) + lane16),
                                             (__attribute__((address_space(3))) void*)(xs + slot * XROW + plane * XPL + s2 * 1024), 16, 0, 0);
          } else {
            const int qy = q - NPL * 3 * XQ;
            const int w2 = qy / (NPL * KG), rem = qy - w2 * (NPL * KG), plane = rem / KG, s2 = rem - plane * KG;
            const int cb2 = min(cgrp * NWV + w2, a.ncob - 1);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(uniform_ptr(yrow + ((size_t)cb2 * 2 + plane) * plx + s2 * 1024) + lane16),
                                             (__attribute__((address_space(3))) void*)(ydst + w2 * YW + plane * YPL + s2 * 1024), 16, 0, 0);
          }
        }
        icnt += 3;
      }
      ++iu; ++iy;
      xrow += S * rowx; yrow += rowy;
      if (iy == a.h) { iy = 0; ++istrip; if (iu < t1) seek(); }
      return first;
    };

    // prologue: steps t0 .. t0+DIST-1 requested; pend = pieces of the steps AFTER the one about to be computed, oldest first
    bool pf[DIST];                                        // was step (t + 1 + j) a first-of-strip request?
#pragma unroll
    for (int j = 0; j < DIST; ++j) pf[j] = false;
    int ahead = 0;                                        // requested steps beyond the current one
    issue(true);                                          // step t0 (always `first`)
#pragma unroll
    for (int j = 0; j < DIST - 1; ++j)
      if (iu < t1) { pf[j] = issue(false); ++ahead; }

    WT_ADD(ti, tmark);
#pragma clang loop unroll(disable)
    for (int t = t0; t < t1; ++t) {
      // wait for step t's operands: the pieces of the `ahead` later steps may stay in flight
      {
        bool anyf = false;
#pragma unroll
        for (int j = 0; j < DIST - 1; ++j) anyf |= pf[j];
        if (ahead == DIST - 1 && !anyf) wait_vm_barrier<(DIST - 1) * PN>();
        else wait_vm_barrier<0>();
      }
      WT_ADD(tw, tmark);
      // every wave is past step t-1: its slots are free for the request DIST-1 steps ahead
#pragma unroll
      for (int j = 0; j + 1 < DIST - 1; ++j) pf[j] = pf[j + 1];
      if (iu < t1) { pf[DIST - 2] = issue(false); } else { pf[DIST - 2] = false; --ahead; }
      WT_ADD(ti, tmark);
      if (active) {
        // one lane-dependent LDS offset for both operands; everything else is a scalar (slot / stage / plane) plus an immediate
        // (lane16 = lhi * 512 + l31 * 16 is also this lane's fragment offset inside a row / dY image; the k-step adds kg * 1024.)
        // The scalar parts are re-derived per read (opaque to CSE): hoisted per-row addresses would cost seven live registers.
        // A SIMD issues ONE instruction per ~4 cycles over all its waves: a 32-cycle MFMA leaves room for at most seven others
        // (MI355X_MICROARCH.md), so the loop is written for instruction COUNT.  The addresses of a step's windows are two VGPRs per
        // tap row set up once per step (row slot of the ring + this lane's fragment offset, for the own and the partner k-step);
        // planes, segments and the edge dwords are immediates off them: the base sits 4 bytes into the segment BEFORE the window, so
        // ds_read2_b32 reaches the two edge dwords (bytes 12 and 1024 of that segment: dword offsets 2 and 255) and ds_read_b128 the
        // window (byte 512 = offset 508).
        const unsigned slot0 = ((ccnt + r0) % RING) * XROW, slot1 = ((ccnt + 1) % RING) * XROW;
        const unsigned wa_m0 = lane16 + slot0 + km * 1024 + 4;      // (own k-step, row r0)
        const unsigned wa_m1 = lane16 + slot1 + km * 1024 + 4;      // (own k-step, row 1)
        const unsigned wa_o0 = lane16 + slot0 + ko * 1024 + 4;      // (partner k-step, row r0)
        const unsigned ya = lane16 + RING * XROW + ((t - t0) % STAGES) * YST + cw * YW;
        struct Win { u32x4 d; unsigned dm, dp; };
        auto load_win = [&](int ph, int i) -> Win {
          const char* const xb = smem + (i == 0 ? wa_m0 : i == 1 ? wa_m1 : wa_o0) + (ph == 2 ? XPL : 0);
          return Win{*reinterpret_cast<const u32x4*>(xb + 508), *reinterpret_cast<const unsigned*>(xb + 8), *reinterpret_cast<const unsigned*>(xb + 1020)};
        };
        auto load_af = [&](int ph, int k) -> h8 {
          return *reinterpret_cast<const h8*>(smem + ya + ((!BF && ph == 0) ? YPL : 0) + k * 1024);
        };
        constexpr int NPH = BF ? 1 : 3;
        h8 af_m = load_af(0, km), af_o = load_af(0, ko);
        Win cur = load_win(0, 0);
#pragma unroll
        for (int wi = 0; wi < NPH * 3; ++wi) {
          const int ph = wi / 3, i = wi % 3;
          Win nxt = cur;
          if (wi + 1 < NPH * 3) nxt = load_win((wi + 1) / 3, (wi + 1) % 3);
          h8 nm = af_m, no = af_o;
          if (i == 2 && ph + 1 < NPH) { nm = load_af(ph + 1, km); no = load_af(ph + 1, ko); }
          __builtin_amdgcn_sched_barrier(0);             // the requests go out BEFORE this window's MFMAs (left alone, the compiler
                                                         // sinks them to one MFMA before their wait)
          const h8 af = i == 2 ? af_o : af_m;
          const int tb = i == 1 ? 3 : 0;
          const u32x4 d = cur.d;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            if (!((DXM >> dx) & 1)) continue;
            u32x4 f = d;
            if (dx == 0)
              f = u32x4{__builtin_amdgcn_alignbyte(d.x, cur.dm, 2), __builtin_amdgcn_alignbyte(d.y, d.x, 2), __builtin_amdgcn_alignbyte(d.z, d.y, 2),
                        __builtin_amdgcn_alignbyte(d.w, d.z, 2)};
            if (dx == 2)
              f = u32x4{__builtin_amdgcn_alignbyte(d.y, d.x, 2), __builtin_amdgcn_alignbyte(d.z, d.y, 2), __builtin_amdgcn_alignbyte(d.w, d.z, 2),
                        __builtin_amdgcn_alignbyte(cur.dp, d.w, 2)};
            if (BF)
              acc[tb + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, af), __builtin_bit_cast(b8, f), acc[tb + dx], 0, 0, 0);
            else
              acc[tb + dx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(h8, f), acc[tb + dx], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);             // nothing beyond window i+1 is hoisted above these MFMAs
          cur = nxt; af_m = nm; af_o = no;
        }
      }
      WT_ADD(tc, tmark);
      ccnt += S; ++cy;
      if (cy == a.h) { cy = 0; ccnt += 3 - S; }          // the next step starts a strip: its request brought three new rows, not S
    }
  }

  // ---- epilogue: a tap row's partial sums sit in several waves of the channel block (row 0 in the even waves, row 2 in the odd ones,
  // row 1 in all of them): one row at a time they are added up in LDS in wave order, then the row's three taps go to the scratch tensor
  // part[run][tap][co][ci] with plain coalesced stores (lane = ci) -- no atomics, a fixed order: bitwise reproducible.
  __syncthreads();
  float* const red = reinterpret_cast<float*>(smem) + cw * (3 * 16 * 64);
  const int cop = a.ncob * 32, cip = a.ncib * 32;
  // D[i = co][j = ci]: lane = ci (l31), register r -> co = (r&3) + 8*(r>>2) + 4*lhi
  float* const part = a.dw + (size_t)runi * 9 * cop * cip + (size_t)(cob * 32 + 4 * lhi) * cip + cib * 32 + l31;
#pragma unroll
  for (int row = 0; row < 3; ++row) {
    const int tb = row == 1 ? 3 : 0;
#pragma unroll 1
    for (int j = 0; j < KG; ++j) {
      const bool contributes = row == 1 || (j & 1) == (row >> 1);
      const bool first = j == (row == 2 ? 1 : 0);
      if (kg == j && contributes) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float* const q = red + ((dx * 16 + r) * 64 + lane);
            *q = first ? acc[tb + dx][r] : *q + acc[tb + dx][r];
          }
      }
      __syncthreads();
    }
    if (active) {
      for (int dx = kg; dx < 3; dx += KG) {
#pragma unroll
        for (int r = 0; r < 16; ++r) part[((size_t)(row * 3 + dx) * cop + (r & 3) + 8 * (r >> 2)) * cip] = red[(dx * 16 + r) * 64 + lane];
      }
    }
    __syncthreads();
  }
#ifdef VIRNET_F16_TIMING
  if (a.tlog && tid == 0) {
    long long* o = a.tlog + (size_t)blockIdx.x * 8;
    o[0] = tstart; o[1] = tw; o[2] = ti; o[3] = tc; o[4] = WT_NOW(); o[5] = t1 - t0;
  }
  if (a.tlog && lane == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    atomicOr(reinterpret_cast<unsigned long long*>(a.tlog + (size_t)blockIdx.x * 8 + 6), (unsigned long long)(((hw >> 4) & 3) + 1) << (4 * wv));
    if (wv == 0) a.tlog[(size_t)blockIdx.x * 8 + 7] = ((long long)(xcc & 0xf) << 16) | (hw & 0xffff);
  }
#endif
}

// dw[co][ci][t] = sum over runs of part[run][t][co][ci]   (thread = one (t, co, ci); consecutive threads = consecutive ci).
// Latency-bound (one dependent chain of nrun loads per thread): sixteen loads in flight per thread; the partial sums are added in
// run order within each of the sixteen strided chains and the chains in a fixed tree -> bitwise reproducible.
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw, int nrun, int cop, int cip, int cout, int cin, int blk);

// the weight gradient's partial-sum reduction and (blocks behind it) the bias gradient's column-partial reduction in ONE launch: a training
// step makes ~45 of each, every one a few microseconds of work behind a launch
__global__ __launch_bounds__(256) void wgrad_reduce_db_kernel(const float* __restrict__ part, float* __restrict__ dw, int nrun, int cop, int cip, int cout, int cin,
                                                              int nred, const float* __restrict__ col, float* __restrict__ db, long nblk, int cvalid, int ncb, int nslices) {
  __shared__ float red[8][32];
  if ((int)blockIdx.x < nred) {
    wgrad_reduce_body(part, dw, nrun, cop, cip, cout, cin, blockIdx.x);
  } else {
    const int q = blockIdx.x - nred;
    colpart_reduce_block(col, db, nblk, cvalid, ncb, q / nslices, q % nslices, nslices, red);
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nrun, int cop, int cip, int cout, int cin) {
  wgrad_reduce_body(part, dw, nrun, cop, cip, cout, cin, blockIdx.x);
}

__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ part, float* __restrict__ dw, int nrun, int cop, int cip, int cout, int cin, int blk) {
  const int i = blk * 256 + threadIdx.x;
  const int per = 9 * cop * cip;
  if (i >= per) return;
  const int ci = i % cip, co = (i / cip) % cop, t = i / (cip * cop);
  if (co >= cout || ci >= cin) return;
  float s[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) s[k] = 0.f;
  int r = 0;
  for (; r + 16 <= nrun; r += 16) {
#pragma unroll
    for (int k = 0; k < 16; ++k) s[k] += part[(size_t)(r + k) * per + i];
  }
#pragma unroll
  for (int k = 0; k < 16; ++k)
    if (r + k < nrun) s[k] += part[(size_t)(r + k) * per + i];
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int k = 0; k < w; ++k) s[k] += s[k + w];
  dw[((size_t)co * cin + ci) * 9 + t] = s[0];
}

// The stride-2 forms (conv_wgrad_f16_kernel<.., S = 2, DXM>): part[run][t' = tap row * 3 + shift][m][n''], n'' = column phase * hp + channel.
//   mode 0, 3x3 stride-2 conv, dw[co][ci][ky][kx]: input column 2ox + kx - 1 is the even phase at ox (kx = 1: aligned), the odd phase at
//           ox - 1 (kx = 0: shift -1) or at ox (kx = 2: aligned); tap row ky as in the stride-1 kernel (row 2oy + ky - 1)
//   mode 1, 2x2 transposed conv, dw[ci][co][a][b] = sum_p x[p][ci] dy[2p + (a, b)][co]: row 2p + a is tap row a + 1, column phase b, aligned
// Runs are added in order: bitwise reproducible.
__global__ __launch_bounds__(256) void wgrad_reduce_s2_kernel(const float* __restrict__ part, float* __restrict__ dw, int nrun, int mp, int np, int hp,
                                                              int mreal, int nreal, int mode) {
  const int ntap = mode ? 4 : 9;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= mreal * nreal * ntap) return;
  const int tt = i % ntap, n = (i / ntap) % nreal, m = i / (ntap * nreal);
  int tp, par;
  if (mode == 0) {
    const int ky = tt / 3, kx = tt - 3 * ky;
    par = kx != 1;
    tp = ky * 3 + (kx == 0 ? 0 : 1);
  } else {
    par = tt & 1;
    tp = ((tt >> 1) + 1) * 3 + 1;
  }
  const size_t per = (size_t)9 * mp * np;
  const float* q = part + ((size_t)tp * mp + m) * np + (size_t)par * hp + n;
  float s4[4] = {0.f, 0.f, 0.f, 0.f};
  int r = 0;
  for (; r + 4 <= nrun; r += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) s4[k] += q[(size_t)(r + k) * per];
  }
  for (int k = 0; r + k < nrun; ++k) s4[k] += q[(size_t)(r + k) * per];
  dw[i] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
}

// k-steps (waves) per channel block: 64-pixel steps = twelve-wave workgroups, one per CU; images of at most 32 columns would waste
// half of every such step, they take 32-pixel steps = six-wave workgroups, two per CU (2/3/3/4 waves on the SIMDs, still the better deal).
struct Plan { int kg, nwv, pairs, split, run, nxs, nsteps; };
inline Plan make_plan(int n, int h, int w, int ncob, int ncib, int s = 1) {
  Plan p;
  static const int kg_env = getenv("VIRNET_WGRAD_KG") ? atoi(getenv("VIRNET_WGRAD_KG")) : 0;   // tuning override: 2 or 4
  p.kg = w <= 32 ? 2 : (kg_env == 2 || kg_env == 4 ? kg_env : 4);   // (T rows of narrow images only have room for 32-pixel steps)
  p.nwv = ncob >= 3 ? 3 : ncob;
  p.nxs = (w + 16 * p.kg - 1) / (16 * p.kg);
  p.nsteps = n * p.nxs * h;
  const int groups = (ncob + p.nwv - 1) / p.nwv;
  p.pairs = groups * ncib;
  int split = (p.kg == 2 && s == 1 ? 512 : 256) / p.pairs;   // workgroups per CU: the LDS rings take 142 KB (64-pixel steps) / 78 KB (32; stride-2 form: 84 KB)
  if (split > p.nsteps / 4) split = p.nsteps / 4;        // runs of at least four steps (each run primes three rows)
  if (split < 1) split = 1;
  p.run = (p.nsteps + split - 1) / split;
  p.split = split;
  return p;
}

template <int NWV, int BF, int KG, int S = 1, int DXM = 7>
int launch_g(GArgs k, const Plan& p, hipStream_t st) {
  constexpr int LDS = WgCfg<BF, KG, S>::lds(NWV) > 3 * 16 * 64 * 4 * NWV ? WgCfg<BF, KG, S>::lds(NWV) : 3 * 16 * 64 * 4 * NWV;   // K loop / epilogue exchange
  static_assert(LDS <= 160 * 1024, "conv_wgrad_f16: LDS over 160 KB");
  static unsigned long long attr_done = 0;
  auto kern = conv_wgrad_f16_kernel<NWV, BF, KG, S, DXM>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_wgrad_f16): %s", hipGetErrorString(e));
  }
  k.nxs = p.nxs; k.nsteps = p.nsteps; k.run = p.run; k.nsplit = p.split; k.npairs = p.pairs;
  hipLaunchKernelGGL(kern, dim3(p.pairs * p.split), dim3(64 * KG * NWV), LDS, st, k);
  return virnet::check_launch("conv_wgrad_f16 launch");
}

template <int BF, int KG, int S = 1, int DXM = 7>
int launch_nwv(const GArgs& k, const Plan& p, hipStream_t st) {
  return p.nwv == 3 ? launch_g<3, BF, KG, S, DXM>(k, p, st) : p.nwv == 2 ? launch_g<2, BF, KG, S, DXM>(k, p, st) : launch_g<1, BF, KG, S, DXM>(k, p, st);
}

static long long* g_wlog = nullptr;
}  // namespace

#ifdef VIRNET_F16_TIMING
extern "C" void virnet_debug_wgrad_timing_buffer(void* p) { g_wlog = static_cast<long long*>(p); }
#endif

extern "C" size_t virnet_chsplit_bytes(int n, int h, int w, int c) {
  return (size_t)n * (h + 2) * ((c + 31) / 32) * 2 * t_nseg(w) * 512;
}

extern "C" size_t virnet_chsplit_colsum_bytes(int n, int h, int w, int c) {
  return (size_t)n * (h + 2) * ((t_nseg(w) + 7) / 8) * ((c + 31) / 32) * 32 * sizeof(float);
}

extern "C" size_t virnet_chsplit_s2_bytes(int n, int h, int w, int c) {
  return (size_t)n * (h + 2) * 2 * ((c + 31) / 32) * 2 * t_nseg(w / 2) * 512;
}

extern "C" size_t virnet_chsplit_s2_colsum_bytes(int n, int h, int w, int c) {
  return (size_t)n * (h + 2) * ((t_nseg(w / 2) + 7) / 8) * 2 * ((c + 31) / 32) * 32 * sizeof(float);
}

static int chsplit_launch(const float* x, int n, int h, int w, int c, int in_act, float in_slope, const float* in_mul, const float* in_add,
                          int bf16, void* out, float* col_scratch, float* db, int cvalid, void* stream, int par2) {
  VIRNET_REQUIRE(x && out, "virnet_chsplit: NULL pointer");
  VIRNET_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "virnet_chsplit: bad shape n=%d h=%d w=%d c=%d (c %% 4 == 0)", n, h, w, c);
  VIRNET_REQUIRE(!par2 || (w % 2 == 0 && c % 32 == 0), "virnet_chsplit_s2: w=%d must be even and c=%d a multiple of 32", w, c);
  VIRNET_REQUIRE((in_mul == nullptr) == (in_add == nullptr), "virnet_chsplit: in_mul and in_add go together");
  VIRNET_REQUIRE(!in_act || (in_slope >= 0.f && in_slope <= 1.f), "virnet_chsplit: in_slope=%g outside [0,1]", in_slope);
  VIRNET_REQUIRE(!db || (col_scratch && cvalid >= 1 && cvalid <= c), "virnet_chsplit: db needs col_scratch and 1 <= cvalid=%d <= c=%d", cvalid, c);
  if (par2) w /= 2;                                       // geometry of T: one row per source row, half the columns, twice the blocks
  TGeom g{n, h, w, c, (par2 ? 2 : 1) * ((c + 31) / 32), t_nseg(w), par2};
  const int sgs = (g.nseg + 7) / 8;
  const long blocks = (long)n * (h + 2) * g.cb * sgs;
  VIRNET_REQUIRE(blocks < (1L << 31), "virnet_chsplit: tensor too large");
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* const cp = db ? col_scratch : nullptr;
  if (bf16)
    hipLaunchKernelGGL(chsplit_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, x, in_mul, in_add, in_act, in_slope, g, static_cast<unsigned short*>(out), cp);
  else
    hipLaunchKernelGGL(chsplit_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, st, x, in_mul, in_add, in_act, in_slope, g, static_cast<unsigned short*>(out), cp);
  if (int rc = virnet::check_launch("chsplit launch")) return rc;
  if (db) {
    const long nblk = (long)n * (h + 2) * sgs;
    const int slices = (int)(nblk / 64 < 1 ? 1 : nblk / 64 > 64 ? 64 : nblk / 64);
    hipLaunchKernelGGL(colpart_reduce_kernel, dim3(g.cb, slices), dim3(256), 0, st, col_scratch, db, nblk, cvalid, par2 ? g.cb / 2 : g.cb);
    return virnet::check_launch("colpart_reduce launch");
  }
  return 0;
}

extern "C" int virnet_chsplit(const float* x, int n, int h, int w, int c, int in_act, float in_slope, const float* in_mul, const float* in_add,
                              int bf16, void* out, float* col_scratch, float* db, int cvalid, void* stream) {
  return chsplit_launch(x, n, h, w, c, in_act, in_slope, in_mul, in_add, bf16, out, col_scratch, db, cvalid, stream, 0);
}

extern "C" int virnet_chsplit_s2(const float* x, int n, int h, int w, int c, int in_act, float in_slope, const float* in_mul, const float* in_add,
                                 int bf16, void* out, float* col_scratch, float* db, int cvalid, void* stream) {
  return chsplit_launch(x, n, h, w, c, in_act, in_slope, in_mul, in_add, bf16, out, col_scratch, db, cvalid, stream, 1);
}

static int conv_wgrad_f16_impl(const void* xt, const void* yt, float* dw, float* scratch, int n, int h, int w, int cx, int cy, int cin, int cout,
                               int bf16, const float* col, float* db, long nblk, int cvalid, void* stream);

extern "C" int virnet_conv_wgrad_f16(const void* xt, const void* yt, float* dw, float* scratch, int n, int h, int w, int cx, int cy, int cin, int cout,
                                     int bf16, void* stream) {
  return conv_wgrad_f16_impl(xt, yt, dw, scratch, n, h, w, cx, cy, cin, cout, bf16, nullptr, nullptr, 0, 0, stream);
}

extern "C" int virnet_conv_wgrad_f16_db(const void* xt, const void* yt, float* dw, float* scratch, int n, int h, int w, int cx, int cy, int cin, int cout,
                                        int bf16, const float* col, float* db, long nblk, int cvalid, void* stream) {
  VIRNET_REQUIRE(col && db && nblk > 0 && cvalid >= 1 && cvalid <= cy, "virnet_conv_wgrad_f16_db: bad column-partial arguments (nblk=%ld cvalid=%d)", nblk, cvalid);
  return conv_wgrad_f16_impl(xt, yt, dw, scratch, n, h, w, cx, cy, cin, cout, bf16, col, db, nblk, cvalid, stream);
}

static int conv_wgrad_f16_impl(const void* xt, const void* yt, float* dw, float* scratch, int n, int h, int w, int cx, int cy, int cin, int cout,
                               int bf16, const float* col, float* db, long nblk, int cvalid, void* stream) {
  VIRNET_REQUIRE(xt && yt && dw && scratch, "virnet_conv_wgrad_f16: NULL pointer");
  VIRNET_REQUIRE(n > 0 && h > 4 && w > 0, "virnet_conv_wgrad_f16: h=%d (the row ring needs h >= 5) or empty input", h);
  VIRNET_REQUIRE(cin >= 1 && cin <= cx && cout >= 1 && cout <= cy, "virnet_conv_wgrad_f16: cin=%d / cout=%d beyond the stored %d / %d channels", cin, cout, cx, cy);
  GArgs k{};
  k.xt = static_cast<const char*>(xt); k.yt = static_cast<const char*>(yt); k.dw = scratch;
  k.n = n; k.h = h; k.w = w; k.nseg = t_nseg(w);
  k.cin = cin; k.cout = cout;
  k.tlog = g_wlog;
  k.ncib = (cx + 31) / 32; k.ncob = (cy + 31) / 32;     // blocks of the T tensors (all stored channels)
  hipStream_t st = static_cast<hipStream_t>(stream);
  GArgs kk = k;
  const Plan p = make_plan(n, h, w, kk.ncob, kk.ncib);
  int rc;
  if (p.kg == 2) rc = bf16 ? launch_nwv<1, 2>(kk, p, st) : launch_nwv<0, 2>(kk, p, st);
  else rc = bf16 ? launch_nwv<1, 4>(kk, p, st) : launch_nwv<0, 4>(kk, p, st);
  if (rc) return rc;
  const int cop = kk.ncob * 32, cip = kk.ncib * 32;
  const int nred = (9 * cop * cip + 255) / 256;
  if (col) {
    const int slices = (int)(nblk / 64 < 1 ? 1 : nblk / 64 > 64 ? 64 : nblk / 64);
    hipLaunchKernelGGL(wgrad_reduce_db_kernel, dim3(nred + kk.ncob * slices), dim3(256), 0, st, scratch, dw, p.split, cop, cip, cout, cin, nred,
                       col, db, nblk, cvalid, kk.ncob, slices);
    return virnet::check_launch("wgrad_reduce_db launch");
  }
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nred), dim3(256), 0, st, scratch, dw, p.split, cop, cip, cout, cin);
  return virnet::check_launch("wgrad_reduce launch");
}

extern "C" size_t virnet_conv_wgrad_f16_scratch_bytes(int n, int h, int w, int cx, int cy) {
  const int ncib = (cx + 31) / 32, ncob = (cy + 31) / 32;
  const Plan p = make_plan(n, h, w, ncob, ncib);
  return (size_t)p.split * 9 * ncob * 32 * ncib * 32 * sizeof(float);
}

extern "C" int virnet_conv_wgrad_f16_s2(const void* hi_t, const void* lo_t, float* dw, float* scratch, int n, int oh, int ow, int chi, int clo,
                                        int cin, int cout, int mode, int bf16, void* stream) {
  VIRNET_REQUIRE(hi_t && lo_t && dw && scratch, "virnet_conv_wgrad_f16_s2: NULL pointer");
  VIRNET_REQUIRE(mode == 0 || mode == 1, "virnet_conv_wgrad_f16_s2: mode=%d (0: 3x3 stride-2 conv, 1: 2x2 transposed conv)", mode);
  VIRNET_REQUIRE(n > 0 && oh > 4 && ow > 0, "virnet_conv_wgrad_f16_s2: oh=%d (the row ring needs oh >= 5) or empty input", oh);
  VIRNET_REQUIRE(chi % 32 == 0, "virnet_conv_wgrad_f16_s2: the high-resolution tensor stores %d channels (multiple of 32 needed)", chi);
  const int mreal = mode ? cin : cout, nreal = mode ? cout : cin;     // rows come from the low-res tensor, columns from the high-res one
  VIRNET_REQUIRE(mreal >= 1 && mreal <= clo && nreal >= 1 && nreal <= chi, "virnet_conv_wgrad_f16_s2: cin=%d / cout=%d beyond the stored channels (%d low-res, %d high-res)",
                 cin, cout, clo, chi);
  GArgs k{};
  k.xt = static_cast<const char*>(hi_t); k.yt = static_cast<const char*>(lo_t); k.dw = scratch;
  k.n = n; k.h = oh; k.w = ow; k.nseg = t_nseg(ow);
  k.cin = nreal; k.cout = mreal;
  k.tlog = nullptr;
  k.ncib = 2 * (chi / 32); k.ncob = (clo + 31) / 32;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const Plan p = make_plan(n, oh, ow, k.ncob, k.ncib, 2);
  int rc;
  if (mode == 0) {
    if (p.kg == 2) rc = bf16 ? launch_nwv<1, 2, 2, 3>(k, p, st) : launch_nwv<0, 2, 2, 3>(k, p, st);
    else rc = bf16 ? launch_nwv<1, 4, 2, 3>(k, p, st) : launch_nwv<0, 4, 2, 3>(k, p, st);
  } else {
    if (p.kg == 2) rc = bf16 ? launch_nwv<1, 2, 2, 2>(k, p, st) : launch_nwv<0, 2, 2, 2>(k, p, st);
    else rc = bf16 ? launch_nwv<1, 4, 2, 2>(k, p, st) : launch_nwv<0, 4, 2, 2>(k, p, st);
  }
  if (rc) return rc;
  const int mp = k.ncob * 32, np = k.ncib * 32;
  const int total = mreal * nreal * (mode ? 4 : 9);
  hipLaunchKernelGGL(wgrad_reduce_s2_kernel, dim3((total + 255) / 256), dim3(256), 0, st, scratch, dw, p.split, mp, np, np / 2, mreal, nreal, mode);
  return virnet::check_launch("wgrad_reduce_s2 launch");
}

extern "C" size_t virnet_conv_wgrad_f16_s2_scratch_bytes(int n, int oh, int ow, int chi, int clo) {
  const int ncib = 2 * (chi / 32), ncob = (clo + 31) / 32;
  const Plan p = make_plan(n, oh, ow, ncob, ncib, 2);
  return (size_t)p.split * 9 * ncob * 32 * ncib * 32 * sizeof(float);
}

extern "C" int virnet_colpart_reduce(const float* col, float* db, long nblk, int ncb, int cvalid, void* stream) {
  VIRNET_REQUIRE(col && db && nblk > 0 && ncb > 0 && cvalid >= 1 && cvalid <= ncb * 32, "virnet_colpart_reduce: bad arguments (nblk=%ld ncb=%d cvalid=%d)", nblk, ncb, cvalid);
  const int slices = (int)(nblk / 64 < 1 ? 1 : nblk / 64 > 64 ? 64 : nblk / 64);
  hipLaunchKernelGGL(colpart_reduce_kernel, dim3(ncb, slices), dim3(256), 0, static_cast<hipStream_t>(stream), col, db, nblk, cvalid, ncb);
  return virnet::check_launch("colpart_reduce launch");
}
