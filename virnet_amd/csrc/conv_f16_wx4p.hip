// conv_f16_wx4p.hip -- the PERSISTENT, overlapped form of conv_wx4_kernel (conv_f16_wx4.hip; round 6).
//
// Same call sites (AttResBlock.conv1/conv2, networks/AttResUNet.py:43,46,55,58), same tile (16 x 32 pixels x 96 channels, 8 waves, one
// workgroup per CU), same K loop -- the generated stage schedules of conv_f16_wx4_sched.inc are used unchanged -- and the same arithmetic in
// the same order: results are BIT FOR BIT those of conv_wx4_kernel (tests/test_conv_wx4p_gpu.py).  What changes is what happens between two
// tiles.  conv_wx4_kernel runs one (tile, channel block) item per workgroup: a prologue (requests 2.3 k cycles after the start, pixels back and
// first position staged at 5.5 k, V written and the barrier passed at 9.5 k) and an epilogue whose LDS exchange (24 blocks, 112 KB) lies on
// top of V and both weight buffers, so nothing of the next tile can exist on the CU before the last store has left -- 10.9 of a tile's 71.4 k
// cycles under load in front of every K loop (profiles/r06_probes.md 2).  Here
//   * a workgroup is PERSISTENT: one per CU, it walks every wgs_per_xcd-th item of its XCD's contiguous item range; weights descriptor,
//     scale / bias table, thread constants are set up once;
//   * the K loop simply CONTINUES across items: the stages of an item's LAST chunk stage the NEXT item's first chunk (pixel requests in
//     stage 0, positions {0,3} in stage 1, {1,4} in stage 2 -- the normal stage code, pointed at the next tile's pixels) and stage 2's weight
//     DMA fetches the next item's stage 0.  No "final" stage variants, no prologue after the workgroup's first item;
//   * the EPILOGUE leaves V and weight buffer 0 alone: the inverse transform only crosses the two waves of a ROW BLOCK, so the exchange runs
//     quarter by quarter (one row block = 6 blocks = 28 KB) through two buffers placed on top of weight buffer 1 and the free LDS above it --
//     while all eight waves read quarter k, combine and store, the two waves of row block k+1 write the next quarter: 12 barriers per tile
//     (conv_wx4_kernel: 11), the same LDS bytes, every write phase hidden behind a read phase.  The chunk-0 pixels the next stage 0 still has
//     to turn into positions {2,5} wait in the staging registers (30 VGPRs) across the epilogue.
// LDS: V [0, 54 KB) | U0 [54, 90 KB) | U1 [90, 126 KB) with exchange buffer A on its upper 27.4 KB | exchange buffer B [126, 153.4 KB) |
// [inverse scale | bias] of every channel block of the launch (<= 2.3 KB).  U1 is dead from the last stage's barrier to the first DMA of the
// next item's stage 0, which is issued behind the barrier that closes the last read of buffer A.
// MEASURED (profiles/r06_probes.md 2): the prologue is gone and paid back -- with all 256 CUs streaming all the time every stage takes 6 % longer,
// the last chunk's stages and the quarter-pipelined exchange cost the rest: 69.2 k per item against 71.4 k per tile (conv2-type 80.2 k
// against 77.0 k), the slowest fixed 16-item walk 4-5 % above the median, +2...3 % per launch.  Opt-in (VIRNET_WX4_PERSIST=1).
// Residual / mask tiles (EPI 1 / 2): one slab (8 items) ahead in two register sets; gfx950 counts loads and stores on ONE vmcnt, so a wait
// for a tile is also a wait for every store issued before it -- the wait is therefore placed explicitly in front of a phase's stores, when
// the youngest outstanding store is a whole phase old (conv_f16.hip, profiles/r02_probes.md).
#include "conv_f16_wx4_common.h"
#include "conv_f16_wx4_sched.inc"
#include <cstdlib>
#include <type_traits>

#ifndef WX4P_RES_AUX
#define WX4P_RES_AUX 2        // residual / mask tile loads: nt (every byte is read exactly once), as conv_f16_wx4.hip
#endif

namespace {
using namespace virnet;

constexpr int WX_PLANE = 18 * 8 * 32;        // one (position, hi|lo) plane: [18 rows][8 x-tiles][32 B]
constexpr int WX_POS = 2 * WX_PLANE;
constexpr int WX_VBYTES = 6 * WX_POS;        // 55296
constexpr int WX_XBLK = 32 * 144 + 64;       // exchange block: [32 columns][32 channels + 16 B pad], skewed by 64 B against its neighbours
constexpr int WX_CHUNK_BYTES = 36 * 1024;    // one slab's weights of one 16-channel chunk: [6 positions][3 dy][hi|lo][1 KB]
constexpr int WXP_MAXNP = 288;               // output channels of one launch whose [inverse scale | bias] table fits beside the exchange

template <int NREP>
struct WxpLds {
  static constexpr int USTAGE = 12 * NREP * 1024;
  static constexpr int U0 = WX_VBYTES;
  static constexpr int U1 = U0 + USTAGE;
  static constexpr int XB = U1 + USTAGE;                 // exchange buffer B: above the weight buffers
  static constexpr int XA = XB - 6 * WX_XBLK;            // exchange buffer A: the top of weight buffer 1
  static constexpr int TAB = XB + 6 * WX_XBLK;           // [inverse scale | bias] of the launch's channels
  static constexpr int TOTAL = TAB + 2 * WXP_MAXNP * 4;
  static_assert(XA >= U1, "exchange buffer A must not reach into weight buffer 0");
  static_assert(TOTAL <= 160 * 1024, "LDS");
};

template <int NREP, int EPI, int PRE>
__global__ __launch_bounds__(512, 2) void conv_wx4p_kernel(const FArgs a, const int nitems, const int items_per_xcd, const int wgs_per_xcd) {
  static_assert(EPI <= 2 && PRE <= 1, "persistent form: plain / residual / mask epilogues, no SFT vectors");
  using L = WxpLds<NREP>;
  constexpr int NB = 32 * NREP;
  constexpr int NDMA = 12 * NREP;                  // 1-KB pieces of one weight stage: [jt][dy][slab][hi|lo]
  constexpr int USTAGE = L::USTAGE;
  constexpr int NDI = (NDMA + 7) / 8;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const v_lds = smem;
  char* const w_lds = smem + L::U0;
  float* const sb_lds = reinterpret_cast<float*>(smem + L::TAB);     // [inverse scale of the NP channels | bias of the NP channels]

  // ---- this workgroup's items: slot, slot + wgs_per_xcd, ... of its XCD's contiguous range; item = tile * ncb + channel block
  const int ncb = a.NP / NB;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int i_end = min(nitems, (xcd + 1) * items_per_xcd);
  int item = xcd * items_per_xcd + slot;
  if (slot >= wgs_per_xcd || item >= i_end) return;

  struct Tile { int img, oy0, ox0, cb; };
  auto decode = [&](int it) {
    Tile t;
    const int tile = fast_div(it, a.mg_ncb);
    t.cb = __builtin_amdgcn_readfirstlane(it - tile * ncb);
    t.img = __builtin_amdgcn_readfirstlane(fast_div(tile, a.mg_tpi));
    const int trem = tile - t.img * (a.ntx * a.nty);
    const int ty = fast_div(trem, a.mg_ntx);
    t.oy0 = __builtin_amdgcn_readfirstlane(ty * 16);
    t.ox0 = __builtin_amdgcn_readfirstlane((trem - ty * a.ntx) * 32);
    return t;
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  int lane16 = lane * 16;
  const int jt = wave & 1, rb = wave >> 1;
  const int nch = a.Cin >> 4;
  const int pxb = a.Cin * 4;                       // bytes per pixel

  // ---- staging.  Main item of a thread: (V row 0..15, x-tile, channel quad): 6 pixels x 4 channels -> 6 positions x 4 channels.
  // The tile's two halo rows (V rows 16, 17) are 2 x 8 x 16 x 6 values = 512 per position pair: ONE value per thread and stage.
  // voff / inb belong to the tile whose pixels are being REQUESTED (the next item's during an item's last chunk); dst never changes.
  const int sxt = (lane >> 2) & 7, sq = lane & 3, srow = tid >> 5;
  const int hp = wave & 1, hg = wave >> 1, hch = lane & 15;
  const int hrow = 16 + (hg >> 1), hxt = (hg & 1) * 4 + (lane >> 4);
  WxItem it0, ith;
  it0.dst = (srow * 8 + sxt) * 32 + ((((sq >> 1) ^ (srow & 1))) << 4) + (sq & 1) * 8;
  ith.dst = (hrow * 8 + hxt) * 32 + ((((hch >> 3) ^ (hrow & 1))) << 4) + (hch & 7) * 2;
  auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, a.H * a.W * a.Cin * 4, 0x00020000);
  auto aim = [&](const Tile& t) {
    // (thread coordinates re-derived from the wave index and the lane count through an opaque copy: derived from `tid` they would stay
    // live across the whole K loop)
    int t_ = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(t_));
#endif
    const int ln = t_ & 63;
    auto one = [&](WxItem& w, int row, int xt, int chan) {
      const int gy = t.oy0 - 1 + row, gx0 = t.ox0 - 1 + 4 * xt;
      const bool rin = (unsigned)gy < (unsigned)a.H;
      w.inb = 0;
#pragma unroll
      for (int b = 0; b < 6; ++b) w.inb |= ((rin && (unsigned)(gx0 + b) < (unsigned)a.W) ? 1u : 0u) << b;
      w.voff = (unsigned)(((gy * a.W + gx0) * a.Cin + chan) * 4);
    };
    one(it0, t_ >> 5, (ln >> 2) & 7, 4 * (ln & 3));
    const int hg_ = wave >> 1;
    one(ith, 16 + (hg_ >> 1), (hg_ & 1) * 4 + (ln >> 4), ln & 15);
    xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)t.img * a.H * a.W * a.Cin), 0, a.H * a.W * a.Cin * 4, 0x00020000);
  };
  float hc[3][6];                                  // (wave-uniform) coefficients of the halo value per stage-pair
#pragma unroll
  for (int jw = 0; jw < 3; ++jw)
#pragma unroll
    for (int b = 0; b < 6; ++b)                      // (readfirstlane: keeps the 18 coefficients in scalar registers)
      hc[jw][b] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, hp ? wx4_coef(jw + 3, b) : wx4_coef(jw, b))));
  char* const vh_lds = v_lds + hp * 3 * WX_POS;
  const float in_slope_eff = a.in_slope;

  // ---- micro-operations of the staging work (conv_f16_wx4.hip; tools/gen_wx4_sched.py places them between the MFMAs of a stage)
  f32x4 d0[6];
  float dh[6];
  int ld_so = 0;                                   // byte offset of the chunk whose pixels are being requested
  constexpr unsigned OOB = 0x80000000u;
  auto ldp = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    d0[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ((it0.inb >> b) & 1u) ? it0.voff + b * pxb : OOB, ld_so, 0));
  };
  auto ldh = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    dh[b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ((ith.inb >> b) & 1u) ? ith.voff + b * pxb : OOB, ld_so, 0));
  };
  auto rdsft = [&]() {}; (void)rdsft;
  auto pr = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    const f32x4 x = d0[b];
    const f32x4 t = x * in_slope_eff;
    d0[b] = f32x4{vmax(x.x, t.x), vmax(x.y, t.y), vmax(x.z, t.z), vmax(x.w, t.w)};
  };
  auto prH = [&](int b0) {
#pragma unroll
    for (int b = b0; b < b0 + 3; ++b) {
      float u = dh[b];
      if constexpr (PRE == 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(u));
#endif
      }
      if constexpr (PRE >= 1) u = vmax(u, u * in_slope_eff);
      dh[b] = u;
    }
  };
  auto prHa = [&]() { prH(0); };
  auto prHb = [&]() { prH(3); };
  struct PutCtx { f32x4 a, b, v; unsigned h0, h1, l0, l1; };
  PutCtx pc[2];
  auto pA = [&](auto xc, auto jc) {
    constexpr int X = decltype(xc)::value, J = decltype(jc)::value;
    if constexpr (J == 0) pc[X].a = 4.f * d0[0] + d0[4];
    else if constexpr (J == 5) pc[X].a = 4.f * d0[1] + d0[5];
    else if constexpr (J == 1 || J == 2) pc[X].a = d0[4] - 4.f * d0[2];
    else pc[X].a = d0[4] - d0[2];
  };
  auto pB = [&](auto xc, auto jc) {
    constexpr int X = decltype(xc)::value, J = decltype(jc)::value;
    if constexpr (J == 1 || J == 2) pc[X].b = d0[3] - 4.f * d0[1];
    else pc[X].b = d0[3] - d0[1];
  };
  auto pV = [&](auto xc, auto jc) {
    constexpr int X = decltype(xc)::value, J = decltype(jc)::value;
    if constexpr (J == 0) pc[X].v = pc[X].a - 5.f * d0[2];
    else if constexpr (J == 5) pc[X].v = pc[X].a - 5.f * d0[3];
    else if constexpr (J == 1) pc[X].v = pc[X].a + pc[X].b;
    else if constexpr (J == 2) pc[X].v = pc[X].a - pc[X].b;
    else if constexpr (J == 3) pc[X].v = pc[X].a + 2.f * pc[X].b;
    else pc[X].v = pc[X].a - 2.f * pc[X].b;
  };
  float amax = 0.f;                                // range guard (conv_f16_common.h): largest transformed magnitude this thread staged
  auto pHi = [&](auto xc) {
    constexpr int X = decltype(xc)::value;
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(pc[X].v.x), fabsf(pc[X].v.y))), fmaxf(fabsf(pc[X].v.z), fabsf(pc[X].v.w)));
    // (pinned: amax is only read after the workgroup's last item, and hipcc otherwise keeps the last stages' transformed values alive --
    // in scratch, across the epilogue -- to fold them into it later)
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(amax));
#endif
    pc[X].h0 = cvtpk(pc[X].v.x, pc[X].v.y);
    pc[X].h1 = cvtpk(pc[X].v.z, pc[X].v.w);
  };
  auto pSub = [&](auto xc) {
    constexpr int X = decltype(xc)::value;
    pc[X].a = f32x4{subhi<0>(pc[X].v.x, pc[X].h0), subhi<1>(pc[X].v.y, pc[X].h0), subhi<0>(pc[X].v.z, pc[X].h1), subhi<1>(pc[X].v.w, pc[X].h1)};
  };
  auto pLo = [&](auto xc) {
    constexpr int X = decltype(xc)::value;
    pc[X].l0 = cvtpk(pc[X].a.x, pc[X].a.y);
    pc[X].l1 = cvtpk(pc[X].a.z, pc[X].a.w);
  };
  // (the V stores are asm: hipcc orders every LDS store it can see behind all pending weight pieces -- conv_f16_wx4.hip)
  unsigned st_main = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(v_lds + it0.dst);
  auto pSt = [&](auto xc, auto jc) {
    constexpr int X = decltype(xc)::value, J = decltype(jc)::value;
    const uint2 hi = make_uint2(pc[X].h0, pc[X].h1), lo = make_uint2(pc[X].l0, pc[X].l1);
    const unsigned ad = st_main;                     // (asm operands of a generic lambda must be its own locals)
    static_assert(WX_POS % 512 == 0 && WX_PLANE % 512 == 0, "ds_write2st64_b64 offsets are in units of 512 bytes");
    asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(ad), "v"(hi), "v"(lo), "n"(J * WX_POS / 512),
                 "n"((J * WX_POS + WX_PLANE) / 512)
                 : "memory");
  };
  float hv = 0.f, hw = 0.f;
  _Float16 hhi = (_Float16)0.f, hlo = (_Float16)0.f;
  auto hSa = [&](auto jwc) {
    constexpr int JW = decltype(jwc)::value;
    hv = fmaf(hc[JW][3], dh[3], fmaf(hc[JW][1], dh[1], hc[JW][0] * dh[0]));
  };
  auto hSb = [&](auto jwc) {
    constexpr int JW = decltype(jwc)::value;
    hw = fmaf(hc[JW][5], dh[5], fmaf(hc[JW][4], dh[4], hc[JW][2] * dh[2]));
  };
  auto hV = [&]() { hv += hw; };
  auto hHi = [&]() {
    amax = fmaxf(amax, fabsf(hv));
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(amax));
#endif
    hhi = (_Float16)hv;
  };
  auto hSub = [&]() { hw = hv - (float)hhi; };
  auto hLo = [&]() { hlo = (_Float16)hw; };
  unsigned st_halo = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(vh_lds + ith.dst);
  auto hSt = [&](auto jwc) {
    constexpr int JW = decltype(jwc)::value;
    const unsigned h = __builtin_bit_cast(unsigned short, hhi), l = __builtin_bit_cast(unsigned short, hlo), ad = st_halo;
    asm volatile("ds_write_b16 %0, %1 offset:%3\n\tds_write_b16 %0, %2 offset:%4" ::"v"(ad), "v"(h), "v"(l), "n"(JW * WX_POS),
                 "n"(JW * WX_POS + WX_PLANE)
                 : "memory");
  };
  // ---- weight DMA: piece q = i*8 + wave -> (jt, dy, slab, hi|lo) in LDS order; source = [slab][chunk][position][dy][hi|lo][1 KB].
  // One buffer descriptor per channel block (`wrs`: the item being multiplied, `wrs_n`: the next item -- used by the last stage's pieces).
  const size_t slab_bytes = (size_t)nch * WX_CHUNK_BYTES;
  auto wdesc = [&](int cb) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wimg + (size_t)(a.slab_base + cb * NREP) * slab_bytes), 0, (int)(NREP * slab_bytes), 0x00020000);
  };
  int poff[NDI], pdst[NDI];
#pragma unroll
  for (int i = 0; i < NDI; ++i) {
    int qd = i * 8 + wave;
    if (qd >= NDMA) qd -= 8;
    const int jq = qd / (6 * NREP), r = qd - jq * (6 * NREP);
    const int dq = r / (2 * NREP), r2 = r - dq * (2 * NREP);
    poff[i] = __builtin_amdgcn_readfirstlane((r2 >> 1) * (int)slab_bytes + jq * (3 * 3 * 2048) + dq * 2048 + (r2 & 1) * 1024);
    pdst[i] = __builtin_amdgcn_readfirstlane(qd * 1024);
  }
  (void)lane16;
  auto dma_piece = [&](auto rs, int i, int src_off, char* wb) {
#if defined(__HIP_DEVICE_COMPILE__)                // (the host pass drops the kernel's stub without a diagnostic when it meets this builtin)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(wb + pdst[i]), 16, lane16, src_off + poff[i], 0, 0);
#else
    (void)rs; (void)i; (void)src_off; (void)wb;
#endif
  };

  // ---- fragment addressing
  int boff[3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int row = 4 * rb + dy + (l31 >> 3);
    boff[dy] = ((4 * rb + dy) * 8 + l31) * 32 + ((lhi ^ (row & 1)) << 4);
  }
  int a_base = jt * (3 * NREP * 2048) + lane * 16;
  const char* const vjt = v_lds + jt * 3 * WX_POS;
  // The thread's K-loop addresses, RE-DERIVED at the head of every item from the wave index and the lane count through an opaque copy:
  // held across the epilogue they are the registers hipcc parks in scratch (around the epilogue's peak), and a scratch reload behind the
  // epilogue waits vmcnt(0) -- for the tile's stores to be acknowledged (measured: 3.2 k cycles between two items, tools/wx4p_timeline.py)
  auto rethread = [&]() {
    int t_ = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(t_));
#endif
    const int ln = t_ & 63, l31_ = ln & 31, lhi_ = ln >> 5;
    lane16 = ln * 16;
    a_base = jt * (3 * NREP * 2048) + ln * 16;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int row = 4 * rb + dy + (l31_ >> 3);
      boff[dy] = ((4 * rb + dy) * 8 + l31_) * 32 + ((lhi_ ^ (row & 1)) << 4);
    }
    const int sxt_ = (ln >> 2) & 7, sq_ = ln & 3, srow_ = t_ >> 5, hg_ = wave >> 1, hch_ = ln & 15;
    const int hrow_ = 16 + (hg_ >> 1), hxt_ = (hg_ & 1) * 4 + (ln >> 4);
    st_main = (unsigned)(size_t)(__attribute__((address_space(3))) char*)v_lds
              + (unsigned)((srow_ * 8 + sxt_) * 32 + ((((sq_ >> 1) ^ (srow_ & 1))) << 4) + (sq_ & 1) * 8);
    st_halo = (unsigned)(size_t)(__attribute__((address_space(3))) char*)vh_lds
              + (unsigned)((hrow_ * 8 + hxt_) * 32 + ((((hch_ >> 3) ^ (hrow_ & 1))) << 4) + (hch_ & 7) * 2);
  };

  f32x16 acc[3][NREP];                              // (never cleared: the first product of every item into a block takes a zero C operand)

  // ---- the workgroup's ONE prologue: table of every channel block, weights of the first item's stage 0, its chunk 0 -> positions {0,3},
  // {1,4} ({2,5} are written by stage 0 itself)
  Tile cur = decode(item);
  auto wrs = wdesc(cur.cb);
  for (int i = tid; i < a.NP; i += 512) {
    sb_lds[i] = a.inv_scale[a.slab_base * 32 + i];
    sb_lds[a.NP + i] = a.bias ? a.bias[a.slab_base * 32 + i] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < NDI; ++i) dma_piece(wrs, i, 0, w_lds);
  aim(cur);
  ldp(WX_I(0)); ldp(WX_I(1)); ldp(WX_I(2)); ldp(WX_I(3)); ldp(WX_I(4)); ldp(WX_I(5));
  ldh(WX_I(0)); ldh(WX_I(1)); ldh(WX_I(2)); ldh(WX_I(3)); ldh(WX_I(4)); ldh(WX_I(5));
  if constexpr (PRE >= 1) {
    pr(WX_I(0)); pr(WX_I(1)); pr(WX_I(2)); pr(WX_I(3)); pr(WX_I(4)); pr(WX_I(5));
    prHa(); prHb();
  }
  pA(WX_I(0), WX_I(0)); pV(WX_I(0), WX_I(0)); pHi(WX_I(0)); pSub(WX_I(0)); pLo(WX_I(0)); pSt(WX_I(0), WX_I(0));
  pA(WX_I(1), WX_I(3)); pB(WX_I(1), WX_I(3)); pV(WX_I(1), WX_I(3)); pHi(WX_I(1)); pSub(WX_I(1)); pLo(WX_I(1)); pSt(WX_I(1), WX_I(3));
  pA(WX_I(0), WX_I(1)); pB(WX_I(0), WX_I(1)); pV(WX_I(0), WX_I(1)); pHi(WX_I(0)); pSub(WX_I(0)); pLo(WX_I(0)); pSt(WX_I(0), WX_I(1));
  pA(WX_I(1), WX_I(4)); pB(WX_I(1), WX_I(4)); pV(WX_I(1), WX_I(4)); pHi(WX_I(1)); pSub(WX_I(1)); pLo(WX_I(1)); pSt(WX_I(1), WX_I(4));
  hSa(WX_I(0)); hSb(WX_I(0)); hV(); hHi(); hSub(); hLo(); hSt(WX_I(0));
  hSa(WX_I(1)); hSb(WX_I(1)); hV(); hHi(); hSub(); hLo(); hSt(WX_I(1));
  // (positions {2,5} too: every item's first stage is the 0F form, which leaves them alone)
  pA(WX_I(0), WX_I(2)); pB(WX_I(0), WX_I(2)); pV(WX_I(0), WX_I(2)); pHi(WX_I(0)); pSub(WX_I(0)); pLo(WX_I(0)); pSt(WX_I(0), WX_I(2));
  pA(WX_I(1), WX_I(5)); pV(WX_I(1), WX_I(5)); pHi(WX_I(1)); pSub(WX_I(1)); pLo(WX_I(1)); pSt(WX_I(1), WX_I(5));
  hSa(WX_I(2)); hSb(WX_I(2)); hV(); hHi(); hSub(); hLo(); hSt(WX_I(2));
  __syncthreads();

  constexpr int NPX = 12;                           // VMEM instructions of one chunk's pixel loads
  auto wrs_n = wrs;                                 // channel block of the NEXT item (set at the head of every item's last chunk)
  // One stage = positions {ji, 3+ji} of chunk c (conv_f16_wx4.hip).  `lastc`: c is the item's last chunk -- the pixels requested in stage 0
  // and staged in stages 1 and 2 are the NEXT item's chunk 0 (it0 / ith / xrs point at it), stage 2's pieces are the next item's stage 0.
  // `firstc`: c is the item's first chunk -- its positions {2,5} are in V already (0F form of stage 0): the epilogue in front of it formed
  // them from the pixel registers behind its first barrier (when every wave had left the last stage, which reads planes {2,5}).
  auto stage = [&](int c, auto jic, auto firstc, auto lastc) {
    constexpr int ji = decltype(jic)::value;
    constexpr bool first = decltype(firstc)::value;
    constexpr bool last = decltype(lastc)::value;
    const int s = c * 3 + ji;
    const char* const wb = w_lds + (s & 1) * USTAGE + a_base;
    char* const wn = w_lds + ((s + 1) & 1) * USTAGE;
    const int src_off = ji < 2 ? c * WX_CHUNK_BYTES + (ji + 1) * 6144 : (last ? 0 : (c + 1) * WX_CHUNK_BYTES);
    const char* const vb = vjt + ji * WX_POS;
    if constexpr (ji == 0) ld_so = last ? 0 : (c + 1) * 64;
    h8 ah[3 * NREP], al[3 * NREP], bh[3], bl[3];
    auto rdA = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      ah[g] = *reinterpret_cast<const h8*>(wb + (g * 2 + 0) * 1024);
      al[g] = *reinterpret_cast<const h8*>(wb + (g * 2 + 1) * 1024);
    };
    auto rdB = [&](auto dc) {
      constexpr int dy = decltype(dc)::value;
      bh[dy] = *reinterpret_cast<const h8*>(vb + boff[dy]);
      bl[dy] = *reinterpret_cast<const h8*>(vb + WX_PLANE + boff[dy]);
    };
    auto dma = [&](auto ic) {
      if constexpr (ji == 2 && last) dma_piece(wrs_n, decltype(ic)::value, src_off, wn);
      else dma_piece(wrs, decltype(ic)::value, src_off, wn);
    };
    auto mfma = [&](auto gc, auto pc_) {
      constexpr int g = decltype(gc)::value, part = decltype(pc_)::value;
      constexpr int dy = g / NREP, nr = g - dy * NREP;
      const h8 wa = part == 0 ? al[g] : ah[g];
      const h8 xv = part == 1 ? bl[dy] : bh[dy];
      // an item's FIRST product into each accumulator block starts from zero (inline constant as the C operand) instead of from a block the
      // epilogue had to clear: 144 v_mov per wave and item left the exchange's writer phases (round 6, second pass)
      if constexpr (first && dy == 0 && part == 0) acc[ji][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xv, f32x16{}, 0, 0, 0);
      else acc[ji][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xv, acc[ji][nr], 0, 0, 0);
    };
#define WX_TS(g) do { } while (0)
#define WX_STAGE_CASE(N_, J_, P_) if constexpr (NREP == N_ && ji == J_ && PRE == P_ && !(J_ == 0 && first)) { WX4_STAGE_##N_##_##J_##_##P_ }
    WX_STAGE_CASE(3, 0, 0) WX_STAGE_CASE(3, 0, 1) WX_STAGE_CASE(3, 1, 0) WX_STAGE_CASE(3, 1, 1) WX_STAGE_CASE(3, 2, 0) WX_STAGE_CASE(3, 2, 1)
#undef WX_STAGE_CASE
    if constexpr (NREP == 3 && ji == 0 && first && PRE == 0) { WX4_STAGE0F_3_0 }
    if constexpr (NREP == 3 && ji == 0 && first && PRE == 1) { WX4_STAGE0F_3_1 }
#undef WX_TS
    // end of stage: this wave's DMA pieces have landed (they are older than the pixel loads of stage 0, which stay in flight), its LDS
    // writes are done; then the workgroup barrier.  (The waits are the BUILTIN: conv_f16_wx4.hip.)
    constexpr int WAIT_ALL = 0x0070;                                           // vmcnt(0) expcnt(7) lgkmcnt(0)
    constexpr int WAIT_PX = (NPX & 15) | 0x0070 | ((NPX >> 4) << 14);          // vmcnt(NPX) lgkmcnt(0)
    if constexpr (ji == 0) __builtin_amdgcn_s_waitcnt(WAIT_PX);
    else __builtin_amdgcn_s_waitcnt(WAIT_ALL);
    asm volatile("s_barrier" ::: "memory");
  };
  using No = std::false_type;
  using Yes = std::true_type;

  const int C = a.cout;
  float* const ybase = a.y_act ? a.y_act : a.y_raw;
  const float slope_eff = a.y_act ? a.slope : 1.f;
  constexpr bool RES = EPI == 1, MASK = EPI == 2, OPND = EPI != 0;
  char* const xa = smem + L::XA;
  char* const xbuf_b = smem + L::XB;

#ifdef VIRNET_F16_TIMING
  // timing build (tools/wx4p_timeline.py): thread 0 stamps s_memtime per item -- K loop start / last chunk start / K loop end / epilogue end
  int n_done = 0;
#define PSTAMP(i) do { if (a.tlog && tid == 0 && n_done < 64) a.tlog[((size_t)blockIdx.x * 64 + n_done) * 4 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
  for (;;) {
    PSTAMP(0);
    // ---- K loop of `cur`; its last chunk stages the next item (the workgroup's last item stages itself again: no request behind a branch)
    rethread();
    aim(cur);            // (the chunk-1 ... pixels of THIS item: re-aimed here instead of keeping the offsets alive through the epilogue)
    stage(0, WX_I(0), Yes{}, No{});
    stage(0, WX_I(1), Yes{}, No{});
    stage(0, WX_I(2), Yes{}, No{});
    for (int c = 1; c + 1 < nch; ++c) {
      stage(c, WX_I(0), No{}, No{});
      stage(c, WX_I(1), No{}, No{});
      stage(c, WX_I(2), No{}, No{});
    }
    PSTAMP(1);
    const int item_n = item + wgs_per_xcd;
    const bool more = item_n < i_end;
    const Tile nxt = decode(more ? item_n : item);
    aim(nxt);
    wrs_n = wdesc(nxt.cb);
    stage(nch - 1, WX_I(0), No{}, Yes{});
    stage(nch - 1, WX_I(1), No{}, Yes{});
    stage(nch - 1, WX_I(2), No{}, Yes{});
    PSTAMP(2);

    // ---- epilogue of `cur`.  Reader: thread = (pixel column px of the tile, row parity prow, channel quad cq), items it = row pairs;
    // quarter q = row block q = items 2q, 2q+1.  One 32-bit byte offset per item serves the operand loads and the stores (buffer
    // instructions; an item outside the image gets an out-of-range offset: loads 0, stores nothing).
    int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid_e));
#endif
    const int lane_e = tid_e & 63, l31_e = lane_e & 31, lhi_e = lane_e >> 5;
    const int cq = tid_e & 7, px = (tid_e >> 3) & 31, prow = tid_e >> 8;
    const int nb_item = cur.cb * NB;                   // first channel of the item inside the launch's NP
    const size_t img_off = (size_t)cur.img * a.H * a.W * C;
    // byte offset of the thread's item 0 (row oy0 + prow) and what an item adds; an item outside the image gets an out-of-range offset
    // (formed per use: eight offsets held in registers were the difference between a clean epilogue and spilled K-loop invariants)
    const unsigned yoff0 = (unsigned)(((cur.oy0 + prow) * a.W + cur.ox0 + px) * C + a.slab_base * 32 + nb_item + cq * 4) * 4u;
    const unsigned ystep = (unsigned)(2 * a.W * C) * 4u;
    const int rows_in = cur.ox0 + px < a.W ? a.H - cur.oy0 - prow : 0;           // item it is inside the image iff 2*it < rows_in
    auto yoff_of = [&](int it) { return 2 * it < rows_in ? yoff0 + (unsigned)it * ystep : 0x80000000u; };
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(ybase + img_off, 0, a.H * a.W * C * 4, 0x00020000);
    const auto ors = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((RES ? a.res : MASK ? a.mask : a.x) + (OPND ? img_off : 0)), 0,
                                                       OPND ? a.H * a.W * C * 4 : 0, 0x00020000);
    f32x4 op[2][OPND ? 8 : 1];                       // operand tiles of two slabs
    auto load_op = [&](int set, int nr) {
      if constexpr (OPND) {
#pragma unroll
        for (int it = 0; it < 8; ++it)
          op[set][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ors, yoff_of(it) + nr * 128, 0, WX4P_RES_AUX));
      }
    };
    // writer: wave (jt, rb) writes three pre-combined blocks of slab nr into the buffer of quarter rb
    //   jt = 0: A0 = M0+M1+M2, A1 = M1-M2, A2 = M1+M2        jt = 1: S = M3+M4, D = M3-M4, E = M5
    // and pixel k of an x-tile is  k=0: A0 + S   k=1: A1 + 2D   k=2: A2 + 4S   k=3: A1 + 8D + E   (rows of AT).
    const int wblk = (jt * 3) * WX_XBLK + l31_e * 144 + lhi_e * 16;
    // (four accumulator registers at a time, each branch storing its own blocks: no 48-register set of pre-combined values, no phi copies)
    auto xwrite = [&](int nr, char* xb) {
      auto q4 = [&](const f32x16& m, int g) { return f32x4{m[4 * g], m[4 * g + 1], m[4 * g + 2], m[4 * g + 3]}; };
      if (jt == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 m0 = q4(acc[0][nr], g), m1 = q4(acc[1][nr], g), m2 = q4(acc[2][nr], g);
          const f32x4 b2 = m1 + m2;
          *reinterpret_cast<f32x4*>(xb + wblk + 0 * WX_XBLK + g * 32) = m0 + b2;
          *reinterpret_cast<f32x4*>(xb + wblk + 1 * WX_XBLK + g * 32) = m1 - m2;
          *reinterpret_cast<f32x4*>(xb + wblk + 2 * WX_XBLK + g * 32) = b2;
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 m0 = q4(acc[0][nr], g), m1 = q4(acc[1][nr], g), m2 = q4(acc[2][nr], g);
          *reinterpret_cast<f32x4*>(xb + wblk + 0 * WX_XBLK + g * 32) = m0 + m1;
          *reinterpret_cast<f32x4*>(xb + wblk + 1 * WX_XBLK + g * 32) = m0 - m1;
          *reinterpret_cast<f32x4*>(xb + wblk + 2 * WX_XBLK + g * 32) = m2;
        }
      }
    };
    const int pk = px & 3, pxt = px >> 2;
    const int r_p = ((pk == 0) ? 0 : (pk == 2) ? 2 : 1) * WX_XBLK + pxt * 144 + cq * 16;
    const int r_q = (3 + (pk & 1)) * WX_XBLK + pxt * 144 + cq * 16;
    const int r_e = 5 * WX_XBLK + pxt * 144 + cq * 16;
    const float ck = (float)(1 << pk), ek = pk == 3 ? 1.f : 0.f;
    auto xread = [&](const char* xb, int j) {               // item j of the quarter: row-in-block 2j + prow
      const int base = ((j * 2 + prow) * 8) * 144;
      const f32x4 p = *reinterpret_cast<const f32x4*>(xb + base + r_p);
      const f32x4 qv = *reinterpret_cast<const f32x4*>(xb + base + r_q);
      const f32x4 e = *reinterpret_cast<const f32x4*>(xb + base + r_e);
      return p + ck * qv + ek * e;
    };
    auto mask4 = [&](f32x4 v, f32x4 m) {
      return f32x4{m.x > 0.f ? v.x : v.x * a.mask_slope, m.y > 0.f ? v.y : v.y * a.mask_slope,
                   m.z > 0.f ? v.z : v.z * a.mask_slope, m.w > 0.f ? v.w : v.w * a.mask_slope};
    };
    if (rb == 0) xwrite(0, xa);
#pragma unroll
    for (int k = 0; k < 4 * NREP; ++k) {
      const int nr = k >> 2, q = k & 3;
      char* const xb_r = (k & 1) ? xbuf_b : xa;
      char* const xb_w = (k & 1) ? xa : xbuf_b;
      wx_lds_barrier();
      if (k == 0) {
        // every wave has left the K loop's last stage (the last reader of planes {2,5}): positions {2,5} and halo pair 2 of the NEXT item's
        // first chunk, from the pixel registers -- which die here instead of living through the epilogue; then slab 0's operand tile
        pA(WX_I(0), WX_I(2)); pB(WX_I(0), WX_I(2)); pV(WX_I(0), WX_I(2)); pHi(WX_I(0)); pSub(WX_I(0)); pLo(WX_I(0)); pSt(WX_I(0), WX_I(2));
        pA(WX_I(1), WX_I(5)); pV(WX_I(1), WX_I(5)); pHi(WX_I(1)); pSub(WX_I(1)); pLo(WX_I(1)); pSt(WX_I(1), WX_I(5));
        hSa(WX_I(2)); hSb(WX_I(2)); hV(); hHi(); hSub(); hLo(); hSt(WX_I(2));
        load_op(0, 0);
      }
      if (k + 1 < 4 * NREP && rb == ((k + 1) & 3)) xwrite((k + 1) >> 2, xb_w);
      const f32x4 i4 = *reinterpret_cast<const f32x4*>(sb_lds + nb_item + nr * 32 + cq * 4);
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb_lds + a.NP + nb_item + nr * 32 + cq * 4);
      f32x4 tv[2];
      tv[0] = xread(xb_r, 0);
      tv[1] = xread(xb_r, 1);
      SB();
      // next slab's operand tile: requested in the phase in which the LAST row block has handed over this slab's accumulators (the
      // registers exist only from then on), in front of that phase's stores
      if (OPND && q == 2 && nr + 1 < NREP) load_op((nr + 1) & 1, nr + 1);
      if constexpr (OPND) {
        // the wait for the NEXT slab's operand tile, placed where the youngest outstanding store is a whole phase old (see the header)
        if (q == 3 && nr + 1 < NREP) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) only (expcnt 7, lgkmcnt 15: not waited for)
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 v = tv[j] * i4 + b4;
        if constexpr (MASK) v = mask4(v, op[nr & 1][2 * q + j]);
        if constexpr (RES) v += op[nr & 1][2 * q + j];
        tv[j] = lrelu4(v, slope_eff);
      }
      SB();
      // (slab offset in the instruction's immediate, cache policy too: two copies of the stores behind a uniform branch -- conv_f16_wx4.hip)
      if (a.store_nt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tv[j]), yrs, yoff_of(2 * q + j) + nr * 128, 0, 2);
      } else {
#pragma unroll
        for (int j = 0; j < 2; ++j) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tv[j]), yrs, yoff_of(2 * q + j) + nr * 128, 0, 0);
      }
    }
    PSTAMP(3);
#ifdef VIRNET_F16_TIMING
    ++n_done;
#endif
    if (!more) break;
    item = item_n;
    cur = nxt;
    wrs = wrs_n;
  }
  range_report(a.range_flag, amax);
}

template <int NREP, int EPI, int PRE>
int launch_wx4p_t(FArgs k, int n_cu, hipStream_t st) {
  using L = WxpLds<NREP>;
  static unsigned long long attr_done = 0;
  auto kern = conv_wx4p_kernel<NREP, EPI, PRE>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_wx4p): %s", hipGetErrorString(e));
  }
  k.nty = (k.H + 15) / 16;
  k.ntx = (k.W + 31) / 32;
  k.ntiles = k.N * k.nty * k.ntx;
  const int ncb = k.NP / (32 * NREP);
  const long nitems_l = (long)k.ntiles * ncb;
  if (nitems_l >= (1L << 31) || (unsigned long long)nitems_l * (unsigned)ncb >= (1ull << 32) ||
      (unsigned long long)k.ntiles * (unsigned)(k.ntx * k.nty) >= (1ull << 32))
    return virnet::set_error("virnet_conv_wx4 (persistent): %d tiles x %d channel blocks exceed the index arithmetic of one launch", k.ntiles, ncb);
  const int nitems = (int)nitems_l;
  // whole tiles per XCD (the channel blocks of a tile stay on one XCD: its L2 serves the tile's second and third read)
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  const int items_per_xcd = k.tiles_per_xcd * ncb;
  // (VIRNET_WX4_PERSIST_WGS: workgroups per XCD -- a test / probe knob: few workgroups walk many items each)
  const char* const wenv = getenv("VIRNET_WX4_PERSIST_WGS");
  const int wgs_per_xcd = std::max(1, std::min(items_per_xcd, wenv && atoi(wenv) > 0 ? atoi(wenv) : n_cu / 8));
  k.mg_ncb = div_magic(ncb);
  k.mg_ntx = div_magic(k.ntx);
  k.mg_tpi = div_magic(k.ntx * k.nty);
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * wgs_per_xcd)), dim3(512), L::TOTAL, st, k, nitems, items_per_xcd, wgs_per_xcd);
  return virnet::check_launch("conv_wx4p launch");
}

}  // namespace

namespace virnet {

// The persistent form serves: three-slab channel blocks, plain / residual / mask epilogues (EPI 0..2), PRE 0 / 1, no T emission, at most
// WXP_MAXNP output channels per launch, at least two chunks and an even number of stages per item (the weight buffers alternate per stage, an item
// starts in buffer 0; first and last chunk have their own stage forms).
bool wx4p_serves(const FArgs& k, int nrep, int epi, int pre) {
  return nrep == 3 && epi <= 2 && pre <= 1 && k.NP <= WXP_MAXNP && (k.Cin >> 4) >= 2 && ((k.Cin >> 4) * 3) % 2 == 0;
}

int launch_wx4p(FArgs k, int nrep, int epi, int pre, int n_cu, hipStream_t st) {
#define VIRNET_WX4P_CASE(E_) if (epi == E_) return pre == 1 ? launch_wx4p_t<3, E_, 1>(k, n_cu, st) : launch_wx4p_t<3, E_, 0>(k, n_cu, st);
  if (nrep == 3) { VIRNET_WX4P_CASE(0) VIRNET_WX4P_CASE(1) VIRNET_WX4P_CASE(2) }
#undef VIRNET_WX4P_CASE
  return virnet::set_error("virnet_conv_wx4 (persistent): no kernel for nrep=%d epi=%d pre=%d", nrep, epi, pre);
}

}  // namespace virnet
