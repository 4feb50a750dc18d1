// conv_f16_s2.hip -- the 3x3 STRIDE-2 convolution (DownBlock.downsampler, networks/AttResUNet.py:67,74) on the f16 matrix pipe with
// split fp32 operands.  Arithmetic, weight image and accuracy argument: conv_f16.hip (same packing, virnet_pack_f16_weight).
//
// What differs from the stride-1 kernel is geometry.  An output tile of 4 rows x 32 columns reads 9 x 65 input pixels, four times
// the pixels per output of the stride-1 halo tile, so the pixel tile (37 KB per 16-channel chunk, double buffered) leaves room for
// ONE workgroup per CU.  The workgroup is therefore made as wide as the CU:
//   * 8 waves (two per SIMD): wave = (output row 0..3, slab group 0..1); the workgroup covers 2*NREP slabs (192 channels at
//     NREP = 3), so the pixel tile is staged ONCE for all of them and every SIMD still has a partner wave to fill barrier and
//     epilogue gaps.  4-wave form (one slab group) for what is left over (96 of 288 channels; 160 = 3 + 2 slabs).
//   * input columns de-interleaved in LDS: a row is stored as [33 even-offset pixels | 32 odd-offset pixels] (offset c = 2*ox + dx
//     from the tile's first column), so the 32 lanes of a B fragment read CONSECUTIVE records for every dx and the stride-1
//     kernel's slot swizzle stays conflict free.
//   * no row reuse across taps (row = 2*oy + dy): three B fragments per kernel column, each prefetched one tap ahead.
// K loop, weight stages (three taps = one kernel column per stage, LDS-DMA, double buffered), MFMA/LDS/VALU interleave and the
// epilogue (per-wave LDS turn-around, loads before stores, buffer stores) follow conv_f16.hip; only the bias / single-store epilogue
// exists here (the down conv has no residual, mask or SFT: AttResUNet.py:74).
#include "conv_f16_common.h"
#ifndef S2_LEDGER
#define S2_LEDGER 0      // probe builds only (tools/build_ledger_s2.sh)
#endif
#include <cstdlib>
#include <type_traits>

namespace {
using namespace virnet;

template <int NG /* slab groups: waves = 4*NG */, int NREP>
__global__ __launch_bounds__(256 * NG, NG) void conv_f16_s2_kernel(const FArgs a) {
  constexpr int NT = 256 * NG, NWAVES = 4 * NG;
  constexpr int TH = 4, IH = 2 * TH + 1, ROWPX = 65, NPIX = IH * ROWPX;
  constexpr int NPIECE = NPIX * 2;
  constexpr int PPT = (NPIECE + NT - 1) / NT;          // 3 (8 waves) or 5 (4 waves)
  constexpr int PG = (PPT + 2) / 3;                    // pieces staged per tap group
  static_assert((PPT - 1) * NT <= NPIECE, "surplus threads redo piece k-1");
  constexpr int PLANE = NPIX * 32, XB = 2 * PLANE;
  constexpr int SLABS = NG * NREP;                     // slabs per workgroup
  constexpr int WGRP = 3 * SLABS * 2048;
  constexpr int NDMA = 3 * SLABS * 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const x_lds = smem;
  char* const w_lds = smem + 2 * XB;

  const int ncb = a.NP / (32 * SLABS);
  const int xcd = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int cb = __builtin_amdgcn_readfirstlane(q % ncb);
  const int tile = __builtin_amdgcn_readfirstlane(xcd * a.tiles_per_xcd + q / ncb);
  if (q / ncb >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx = __builtin_amdgcn_readfirstlane(tile % a.ntx);
  const int ty = __builtin_amdgcn_readfirstlane((tile / a.ntx) % a.nty);
  const int img = __builtin_amdgcn_readfirstlane(tile / (a.ntx * a.nty));
  const int oy0 = ty * TH, ox0 = tx * 32;
  const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int row = wv & 3, sg = wv >> 2;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nch = a.Cin >> 4;
  const int nstages = nch * 3;
  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;

  // ---- pixel staging: piece -> (record p, 8-channel half h); record p = input row iy, slot idx: [0,33) = even offsets c = 2*idx,
  // [33,65) = odd offsets c = 2*(idx-33)+1 from the tile's first input column
  unsigned soff[PPT];
  int sdst[PPT];
  bool sinb[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int qq = k * NT + tid;
    const int qc = qq < NPIECE ? qq : qq - NT;
    const int p = qc >> 1, h = qc & 1;
    const int iy = p / ROWPX, idx = p - iy * ROWPX;
    const int c = idx < 33 ? 2 * idx : 2 * (idx - 33) + 1;
    const int gy = iy0 + iy, gx = ix0 + c;
    sinb[k] = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
    soff[k] = (unsigned)((gyc * a.W + gxc) * a.Cin + h * 8);
    sdst[k] = p * 32 + ((h ^ ((p >> 3) & 1)) << 4);
  }
  const float in_slope_eff = a.in_act ? a.in_slope : 1.f;
  float amax = 0.f;                                // range guard (conv_f16_common.h)
  auto stage_store = [&](char* xb, int k, f32x4 r0, f32x4 r1) {
    r0 = lrelu4(r0, in_slope_eff);
    r1 = lrelu4(r1, in_slope_eff);
    const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
    r0 = sinb[k] ? r0 : z;
    r1 = sinb[k] ? r1 : z;
    h8 hi, lo;
    range_note(amax, r0, r1);
    split8(r0, r1, hi, lo);
    *reinterpret_cast<h8*>(xb + sdst[k]) = hi;
    *reinterpret_cast<h8*>(xb + PLANE + sdst[k]) = lo;
  };

  // ---- weight DMA: piece = (tap-in-group, slab of the workgroup, hi|lo), 1 KB; wave w moves pieces w, w + NWAVES, ...
  const size_t slab_bytes = (size_t)nch * 9 * 2048;
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wimg + (size_t)(a.slab_base + cb * SLABS) * slab_bytes), 0,
                                                     (int)(SLABS * slab_bytes), 0x00020000);
  const int lane16w = lane * 16;
  auto dma_group = [&](int stage, char* wb) {
#pragma unroll
    for (int i = 0; i < (NDMA + NWAVES - 1) / NWAVES; ++i) {
      const int qd = i * NWAVES + wv;
      if (qd < NDMA) {
        const int tg = qd / (SLABS * 2), rem = qd - tg * (SLABS * 2);
        lds_dma16(wrs, wb + qd * 1024, lane16w, (rem >> 1) * (int)slab_bytes + ((stage * 3 + tg) * 2 + (rem & 1)) * 1024);
      }
    }
  };

  // ---- fragment addressing: B of tap (dy, dx) = input row 2*row + dy, slots l31 + dx/2 (even offsets) or 33 + l31 (dx = 1)
  int boff[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int p = (2 * row + dy) * ROWPX + (dx == 1 ? 33 + l31 : l31 + (dx >> 1));
      boff[dy][dx] = p * 32 + ((lhi ^ ((p >> 3) & 1)) << 4);
    }
  const int aoff = sg * NREP * 2048 + lane * 16;          // this wave's slabs inside a tap's [slab][hi|lo] block

  f32x16 acc[NREP];
#pragma unroll
  for (int nr = 0; nr < NREP; ++nr)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nr][r] = 0.f;

  // ---- prologue
  dma_group(0, w_lds);
  {
    f32x4 r0[PPT], r1[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      r0[k] = *reinterpret_cast<const f32x4*>(ximg + soff[k]);
      r1[k] = *reinterpret_cast<const f32x4*>(ximg + soff[k] + 4);
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) stage_store(x_lds, k, r0[k], r1[k]);
  }
  __syncthreads();

  h8 ah[2][NREP], al[2][NREP];
  h8 bh[3], bl[3];
  auto read_a = [&](const char* wb, int tg, h8 (&h)[NREP], h8 (&l)[NREP]) {
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      h[nr] = *reinterpret_cast<const h8*>(wb + tg * (SLABS * 2048) + nr * 2048 + aoff);
      l[nr] = *reinterpret_cast<const h8*>(wb + tg * (SLABS * 2048) + nr * 2048 + 1024 + aoff);
    }
  };
  auto read_b = [&](const char* xb, int dy, int dx) {
    bh[dy] = *reinterpret_cast<const h8*>(xb + boff[dy][dx]);
    bl[dy] = *reinterpret_cast<const h8*>(xb + PLANE + boff[dy][dx]);
  };

  auto group = [&](int c, auto pc, auto gc) {
    constexpr int P = decltype(pc)::value, g = decltype(gc)::value;
    const int stage = c * 3 + g;
    const char* const xb = x_lds + P * XB;
    char* const xn = x_lds + (P ^ 1) * XB;
    const char* const wb = w_lds + ((P + g) & 1) * WGRP;
    char* const wn = w_lds + ((P + g + 1) & 1) * WGRP;
    if (stage + 1 < nstages) dma_group(stage + 1, wn);
    const int cn = min(c + 1, nch - 1);
    constexpr int K0 = g * PG, K1 = (g + 1) * PG < PPT ? (g + 1) * PG : PPT, NK = K1 > K0 ? K1 - K0 : 0;
    f32x4 s0[NK > 0 ? NK : 1], s1[NK > 0 ? NK : 1];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float* const src = ximg + soff[K0 + k] + cn * 16;
      s0[k] = *reinterpret_cast<const f32x4*>(src);
      s1[k] = *reinterpret_cast<const f32x4*>(src + 4);
    }
    read_a(wb, 0, ah[(P + 3 * g) & 1], al[(P + 3 * g) & 1]);
    if (g == 0) read_b(xb, 0, 0);
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int cur = (P + 3 * g + dy) & 1;
      SB();
      if (dy < 2) {
        read_a(wb, dy + 1, ah[cur ^ 1], al[cur ^ 1]);
        read_b(xb, dy + 1, g);
      } else {
        if (g < 2) read_b(xb, 0, g + 1);
#pragma unroll
        for (int k = 0; k < NK; ++k) stage_store(xn, K0 + k, s0[k], s1[k]);
      }
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int nr = 0; nr < NREP; ++nr) {
          const h8 wa = (part == 0) ? al[cur][nr] : ah[cur][nr];
          const h8 xv = (part == 1) ? bl[dy] : bh[dy];
#if S2_LEDGER
          // probe builds (tools/build_ledger_s2.sh, profiles/r06_probes.md 6): 1 = no MFMA, 2 = two of the three products, 3 = one -- what the
          // launch would cost with fewer products and everything else in place (the upper bound of a reduced-product form).  Never shipped.
          if (S2_LEDGER == 1 || (S2_LEDGER == 2 && part == 0) || (S2_LEDGER == 3 && part != 2)) {
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" ::"v"(wa), "v"(xv));
#endif
            continue;
          }
#endif
          acc[nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xv, acc[nr], 0, 0, 0);
        }
      constexpr int NM = 3 * NREP;
      if (dy < 2) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < 2 * NREP + 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (g < 2 && i < 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (NK > 0) __builtin_amdgcn_sched_group_barrier(0x002, 6 * NK, 0);
        }
        if (NK > 0) __builtin_amdgcn_sched_group_barrier(0x200, 2 * NK, 0);
      }
    }
    SB();
    __syncthreads();
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

  range_report(a.range_flag, amax);
  // ---- epilogue: per-wave LDS turn-around of each 32-channel slab, every load before the first store, branch-free buffer stores
  const int nbase = (a.slab_base + cb * SLABS + sg * NREP) * 32;
  const int C = a.cout;
  const size_t img_off = (size_t)img * a.OH * a.OW * C;
  constexpr int TPIX = 144, NIT = 4, TREG = 32 * TPIX;
  static_assert(NWAVES * 2 * TREG <= 2 * XB + 2 * WGRP, "turn-around regions fit the K loop's LDS");
  char* const tbuf = smem + wv * (2 * TREG);
  const int cq = lane & 7, psub = lane >> 3;
  float* const y = (a.y_act ? a.y_act : a.y_raw) + img_off;
  const float slope_eff = a.y_act ? a.slope : 1.f;
  const float* const bp = a.bias ? a.bias : a.inv_scale;
  const float hb = a.bias ? 1.f : 0.f;
  const int oy = oy0 + row;
  unsigned yoff[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int ox = ox0 + it * 8 + psub;
    const bool ok = oy < a.OH && ox < a.OW;
    yoff[it] = ok ? ((unsigned)(oy * a.OW + ox) * (unsigned)C + (unsigned)(nbase + cq * 4)) * 4u : 0x80000000u;
  }
  f32x4 bias4[NREP], inv4[NREP];
#pragma unroll
  for (int nr = 0; nr < NREP; ++nr) {
    inv4[nr] = *reinterpret_cast<const f32x4*>(a.inv_scale + nbase + nr * 32 + cq * 4);
    bias4[nr] = *reinterpret_cast<const f32x4*>(bp + nbase + nr * 32 + cq * 4);
  }
  auto turn_in = [&](int nr, int region) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(tbuf + region * TREG + l31 * TPIX + (8 * g + 4 * lhi) * 4) =
          f32x4{acc[nr][4 * g], acc[nr][4 * g + 1], acc[nr][4 * g + 2], acc[nr][4 * g + 3]};
  };
  turn_in(0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
  for (int nr = 0; nr < NREP; ++nr) asm volatile("" ::"v"(inv4[nr]), "v"(bias4[nr]));
#endif
  const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y, 0, a.OH * a.OW * C * 4, 0x00020000);
#pragma unroll
  for (int nr = 0; nr < NREP; ++nr) {
    if (nr + 1 < NREP) turn_in(nr + 1, (nr + 1) & 1);
    const f32x4 b4 = bias4[nr] * hb;
    f32x4 tv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) tv[it] = *reinterpret_cast<const f32x4*>(tbuf + (nr & 1) * TREG + (it * 8 + psub) * TPIX + cq * 16);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const f32x4 v = lrelu4(tv[it] * inv4[nr] + b4, slope_eff);
      // (immediate, not soffset: conv_f16.hip; non-temporal for tensors larger than the Infinity Cache: conv_f16_wx4.hip)
      if (a.store_nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, yoff[it] + nr * 128, 0, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, yoff[it] + nr * 128, 0, 0);
    }
  }
}

template <int NG, int NREP>
int launch(FArgs k, hipStream_t st) {
  constexpr int LDS = 2 * (2 * 9 * 65 * 32) + 2 * (3 * NG * NREP * 2048);
  static unsigned long long attr_done = 0;
  auto kern = conv_f16_s2_kernel<NG, NREP>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_f16_s2): %s", hipGetErrorString(e));
  }
  k.nty = (k.OH + 3) / 4;
  k.ntx = (k.OW + 31) / 32;
  k.ntiles = k.N * k.nty * k.ntx;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  const int ncb = k.NP / (32 * NG * NREP);
  const unsigned grid = (unsigned)(8 * k.tiles_per_xcd * ncb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * NG), LDS, st, k);
  return virnet::check_launch("conv_f16_s2 launch");
}

}  // namespace

// Slabs per workgroup: 6 (8 waves) where the count allows, then 3 / 2 / 1 with 4 waves (288 channels = 6 + 3, 160 = 3 + 2, 224 = 6 + ... 3 + 2 + 2).
int virnet::launch_f16_s2(FArgs k, int nb, hipStream_t st) {
  // 160 and 224 channels (SISR: 5 / 7 slabs) as ONE 4-wave launch with 5 / 7 slabs per workgroup (the pixel tile staged once) instead of
  // 3 + 2 / 3 + 2 + 2: the workgroup is alone on its CU either way (75 KB pixel tiles), so its 512 registers per wave are there
  static const bool wide_off = getenv("VIRNET_S2_WIDE") && getenv("VIRNET_S2_WIDE")[0] == '0';      // (A/B knob)
  // ... unless the launch is a few dozen tiles (SISR, one image: 160 -> 224 channels onto 64 x 64 = 32 tiles = 32 workgroups on 256 CUs,
  // 54 us): then one slab per workgroup, all slabs in ONE launch (32 x 7 = 224 workgroups).  VIRNET_S2_SPLIT_TILES: the largest such launch.
  const char* const env_t = getenv("VIRNET_S2_SPLIT_TILES");
  const long split_tiles = env_t ? atol(env_t) : 64;
  const long tiles_all = (long)k.N * ((k.OH + 3) / 4) * ((k.OW + 31) / 32);
  if (nb > 1 && tiles_all <= split_tiles) {
    FArgs kk = k;
    kk.slab_base = 0;
    kk.NP = nb * 32;
    return launch<1, 1>(kk, st);
  }
  if ((nb == 5 || nb == 7 || nb == 4) && !wide_off) {
    FArgs kk = k;
    kk.slab_base = 0;
    kk.NP = nb * 32;
    return nb == 5 ? launch<1, 5>(kk, st) : nb == 7 ? launch<1, 7>(kk, st) : launch<1, 4>(kk, st);
  }
  int n6 = nb / 6, rem = nb - 6 * n6;
  // Small launches (single images): with 6 slabs per workgroup a 64x64 output is 32 workgroups on a 256-CU chip (and 288 channels two
  // such launches back to back: 48 + 36 us measured); 3-slab workgroups triple the grid and put all slabs in ONE launch.  The slab
  // grouping does not change any result bit (channels are independent).
  {
    const long tiles = (long)k.N * ((k.OH + 3) / 4) * ((k.OW + 31) / 32);
    if (n6 > 0 && tiles * n6 < 192) { n6 = 0; rem = nb; }
  }
  if (rem == 1 && n6 >= 1) { n6 -= 1; rem = 7; }
  int n3 = rem / 3, rem2 = rem - 3 * n3;
  if (rem2 == 1 && n3 >= 1) { n3 -= 1; rem2 = 4; }
  const int n2 = rem2 / 2, n1 = rem2 - 2 * n2;
  int base = 0;
  auto run = [&](int ng, int nrep, int groups) -> int {
    if (groups <= 0) return 0;
    FArgs kk = k;
    kk.slab_base = base;
    kk.NP = groups * ng * nrep * 32;
    base += groups * ng * nrep;
    if (ng == 2 && nrep == 3) return launch<2, 3>(kk, st);
    if (ng == 1 && nrep == 3) return launch<1, 3>(kk, st);
    if (ng == 1 && nrep == 2) return launch<1, 2>(kk, st);
    return launch<1, 1>(kk, st);
  };
  if (int rc = run(2, 3, n6)) return rc;
  if (int rc = run(1, 3, n3)) return rc;
  if (int rc = run(1, 2, n2)) return rc;
  return run(1, 1, n1);
}
