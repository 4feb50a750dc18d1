// conv_f16_wx4h.hip -- conv_wx4 (conv_f16_wx4.hip: Winograd F(4,3) along x on split-fp16 MFMA products) on 8 x 32-pixel tiles with
// FOUR waves, so that TWO workgroups share a CU (gfx950).
//
// Same call sites (AttResBlock.conv1/conv2, networks/AttResUNet.py:43,46,55,58; DnCNN mid convs, networks/DnCNN.py:25-28; their
// input-gradient GEMMs), same packed weight image, same arithmetic per product, same epilogue forms -- conv_f16_wx4.hip's header
// derives the algorithm.  Why a second tile form: conv_wx4 holds one 16 x 32 x 96 tile per CU (LDS 126 KB, 8 waves x 256 VGPRs), so the
// matrix pipe idles while that one workgroup runs its prologue and epilogue -- 30 % of a 96-channel tile (profiles/r03_probes.md 3).
// Here a workgroup is half of that (rows 8, waves 4, LDS <= 80 KB) and the CU's second workgroup multiplies meanwhile.
//
// Workgroup = 4 waves = one 8 x 32 pixel tile x 32*NREP output channels; wave (jt, rb) owns positions {3jt..3jt+2} of row block rb
// (rb = 0,1), 3 x NREP accumulator blocks, exactly as a wave of conv_wx4.
//   V (LDS, 30 KB): per position a hi and a lo plane of [10 rows][8 x-tiles] 32-byte records; single buffered, planes {ji, 3+ji} are
//       replaced during the stage after they die (as conv_wx4).
//   U (LDS, RING of 4 x 4*NREP KB): a weight stage (12*NREP KB) no longer fits twice, so it is cut into its three row taps: group
//       (stage, dy) = [jt][slab][hi|lo] fragments of 1 KB.  A stage is three groups of 3*NREP MFMAs per wave, each closed by a
//       workgroup barrier; group n = 9*chunk + 3*ji + dy lives in ring slot n & 3.  During group n every wave issues its NREP DMA
//       pieces of group n+3 (the same dy of the NEXT stage) into the slot group n-1 has just left, and at the end of group n it waits
//       for its pieces of group n+1 (two groups old) with s_waitcnt vmcnt(K), K = the vector-memory operations issued since, which
//       tools/gen_wx4h_sched.py knows because it places them.  A fragments of a group are read behind the barrier that opens it.
//   Staging: main item as conv_wx4 (V row 0..7, x-tile, channel quad); the two halo rows are 2 x 8 x 16 (row, x-tile, channel)
//       triples = ONE per thread: six scalars -> positions jw and jw+3 per stage.
// Epilogue: conv_wx4's exchange (three pre-combined blocks per wave and slab, [pixel][channel] records) with 12 blocks per slab.
#include "conv_f16_wx4_common.h"
#include "conv_f16_wx4h_sched.inc"
#include <cstdlib>

namespace {
using namespace virnet;

constexpr int WH_PLANE = 10 * 8 * 32;        // one (position, hi|lo) plane: [10 rows][8 x-tiles][32 B]
constexpr int WH_POS = 2 * WH_PLANE;
constexpr int WH_VBYTES = 6 * WH_POS;        // 30720
constexpr int WH_XBLK = 32 * 144 + 64;       // exchange block: [32 columns][32 channels + 16 B pad], skewed by 64 B against its neighbours
constexpr int WH_CHUNK_BYTES = 36 * 1024;    // one slab's weights of one 16-channel chunk: [6 positions][3 dy][hi|lo][1 KB]

// TE = 1: T emission (conv_f16_wx4.hip / conv_f16.hip): epilogue items = 8 consecutive pixels of one row, thread = (row of 8, x-segment of 4,
// channel quad).  With two workgroups per CU the emitted image's stores run beside the other workgroup's K loop.
template <int NREP, int EPI, int PRE, int TE = 0>
__global__ __launch_bounds__(256, NREP > 3 ? 1 : 2) void conv_wx4h_kernel(const FArgs a) {
  // NREP = 5 (SISR's 160 channels as ONE launch instead of 3 + 2 slabs): one workgroup per CU, one wave per SIMD -- 512 registers per wave
  // hold the 15 accumulator blocks, 110 KB of LDS the 20-KB ring slots; the pixel tile is staged and transformed once for all five slabs
  static_assert(!TE || EPI < 4, "T emission: single-store epilogues");
  constexpr int NB = 32 * NREP;
  constexpr int GRP = 4 * NREP * 1024;             // one ring slot: [jt][slab][hi|lo][1 KB]

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const v_lds = smem;
  char* const w_lds = smem + WH_VBYTES;
  constexpr int LDS_MAIN = WH_VBYTES + 4 * GRP > 12 * WH_XBLK ? WH_VBYTES + 4 * GRP : 12 * WH_XBLK;
  float* const sft_lds = reinterpret_cast<float*>(smem + LDS_MAIN) + 2 * NB;   // PRE 2: [in_mul | in_add] of the image's Cin channels
  float* const sb_lds = reinterpret_cast<float*>(smem + LDS_MAIN);   // [inverse scale | bias] of the NB channels

  // ---- workgroup -> (tile, channel block): contiguous tile ranges per XCD (block b runs on XCD b%8), channel blocks adjacent
  const int ncb = a.NP / NB;
  const int xcd = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int qt = fast_div(q, a.mg_ncb);
  const int cb = __builtin_amdgcn_readfirstlane(q - qt * ncb);
  // (a.rev: the XCD walks its tile range backwards -- consecutive launches alternate, so a launch starts on the tiles its producer wrote last)
  const int tile = __builtin_amdgcn_readfirstlane(xcd * a.tiles_per_xcd + (a.rev ? a.tiles_per_xcd - 1 - qt : qt));
  if (qt >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int img = __builtin_amdgcn_readfirstlane(fast_div(tile, a.mg_tpi));
  const int trem = tile - img * (a.ntx * a.nty);
  const int ty = __builtin_amdgcn_readfirstlane(fast_div(trem, a.mg_ntx));
  const int tx = __builtin_amdgcn_readfirstlane(trem - ty * a.ntx);
  const int oy0 = ty * 8, ox0 = tx * 32;

  const int tid = threadIdx.x;
  TSTAMP(0);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int lane16 = lane * 16;
  const int jt = wave & 1, rb = wave >> 1;
  const int nch = a.Cin >> 4;
  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;
  const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, a.H * a.W * a.Cin * 4, 0x00020000);
  const int pxb = a.Cin * 4;                       // bytes per pixel

  // ---- staging items
  auto item_at = [&](int srow, int sxt, int chan, int dst) {
    WxItem it;
    const int gy = oy0 - 1 + srow, gx0 = ox0 - 1 + 4 * sxt;
    const bool rin = (unsigned)gy < (unsigned)a.H;
    it.inb = 0;
#pragma unroll
    for (int b = 0; b < 6; ++b) it.inb |= ((rin && (unsigned)(gx0 + b) < (unsigned)a.W) ? 1u : 0u) << b;
    it.voff = (unsigned)(((gy * a.W + gx0) * a.Cin + chan) * 4);
    it.dst = dst;
    return it;
  };
  const int sxt = (lane >> 2) & 7, sq = lane & 3, srow = tid >> 5;
  const WxItem it0 = item_at(srow, sxt, 4 * sq, (srow * 8 + sxt) * 32 + ((((sq >> 1) ^ (srow & 1))) << 4) + (sq & 1) * 8);
  // halo triple of this thread: V row 8 + (wave>>1), x-tile (wave&1)*4 + (lane>>4), channel lane&15
  const int hch = lane & 15;
  const int hrow = 8 + (wave >> 1), hxt = (wave & 1) * 4 + (lane >> 4);
  const WxItem ith = item_at(hrow, hxt, hch, (hrow * 8 + hxt) * 32 + ((((hch >> 3) ^ (hrow & 1))) << 4) + (hch & 7) * 2);

  const float* const imul = PRE == 2 ? a.in_mul + (size_t)img * a.Cin : nullptr;
  const float* const iadd = PRE == 2 ? a.in_add + (size_t)img * a.Cin : nullptr;
  const float in_slope_eff = a.in_slope;
  f32x4 sm = f32x4{1.f, 1.f, 1.f, 1.f}, sa = f32x4{0.f, 0.f, 0.f, 0.f};
  float smh = 1.f, sah = 0.f;
  // ---- micro-operations of the staging work (placed by tools/gen_wx4h_sched.py); see conv_f16_wx4.hip for the masking rules
  f32x4 d0[6];
  float dh[6];
  int ld_so = 0;
  constexpr unsigned OOB = 0x80000000u;
  auto ldp = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    d0[b] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, ((it0.inb >> b) & 1u) ? it0.voff + b * pxb : OOB, ld_so, 0));
  };
  auto ldh = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    dh[b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, ((ith.inb >> b) & 1u) ? ith.voff + b * pxb : OOB, ld_so, 0));
  };
  auto ldsft = [&]() {
    if constexpr (PRE == 2) {
      sm = *reinterpret_cast<const f32x4*>(imul + (ld_so >> 2) + 4 * sq);
      sa = *reinterpret_cast<const f32x4*>(iadd + (ld_so >> 2) + 4 * sq);
      smh = imul[(ld_so >> 2) + hch];
      sah = iadd[(ld_so >> 2) + hch];
    }
  };
  // K loop: the chunk's SFT vectors come from the LDS table the prologue filled, read at the point of use (stage 1)
  auto rdsft = [&]() {
    if constexpr (PRE == 2) {
      // the thread's offsets into the table are re-derived from lane16 (live for the weight DMA anyway) by two opaque VALU ops per
      // chunk: as loop invariants the four table addresses were parked in scratch and reloaded INSIDE the K loop behind a vmcnt(0)
      int o4 = 0, o1 = 0;                                   // bytes: 4*sq*4 = lane16 & 48;  hch*4 = (lane16 >> 2) & 60
#if defined(__HIP_DEVICE_COMPILE__)
      asm volatile("v_and_b32 %0, 48, %1" : "=v"(o4) : "v"(lane16));
      asm volatile("v_bfe_u32 %0, %1, 2, 6" : "=v"(o1) : "v"(lane16));
#endif
      const char* const tb = reinterpret_cast<const char*>(sft_lds) + ld_so;
      sm = *reinterpret_cast<const f32x4*>(tb + o4);
      sa = *reinterpret_cast<const f32x4*>(tb + a.Cin * 4 + o4);
      smh = *reinterpret_cast<const float*>(tb + o1);
      sah = *reinterpret_cast<const float*>(tb + a.Cin * 4 + o1);
    }
  };
  auto pr = [&](auto bc) {
    constexpr int b = decltype(bc)::value;
    f32x4 x = d0[b];
    if constexpr (PRE == 2) x = x * sm + sa;
    const f32x4 t = x * in_slope_eff;
    f32x4 v = f32x4{vmax(x.x, t.x), vmax(x.y, t.y), vmax(x.z, t.z), vmax(x.w, t.w)};
    if constexpr (PRE == 2) v = ((it0.inb >> b) & 1u) ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    d0[b] = v;
  };
  auto prH = [&]() {
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      float u = dh[b];
      if constexpr (PRE == 2) u = u * smh + sah;
      if constexpr (PRE >= 1) u = vmax(u, u * in_slope_eff);
      if constexpr (PRE == 2) u = ((ith.inb >> b) & 1u) ? u : 0.f;
      dh[b] = u;
    }
  };
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
  float amax = 0.f;                                // range guard: largest transformed magnitude this thread staged
  auto pHi = [&](auto xc) {
    constexpr int X = decltype(xc)::value;
    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(pc[X].v.x), fabsf(pc[X].v.y))), fmaxf(fabsf(pc[X].v.z), fabsf(pc[X].v.w)));
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
  // (V stores as asm: conv_f16_wx4.hip -- hipcc would order every visible LDS store behind all pending weight pieces)
  const unsigned st_main = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(v_lds + it0.dst);
  auto pSt = [&](auto xc, auto jc) {
    constexpr int X = decltype(xc)::value, J = decltype(jc)::value;
    const uint2 hi = make_uint2(pc[X].h0, pc[X].h1), lo = make_uint2(pc[X].l0, pc[X].l1);
    const unsigned ad = st_main;
    static_assert(WH_POS % 512 == 0 && WH_PLANE % 512 == 0, "ds_write2st64_b64 offsets are in units of 512 bytes");
    asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(ad), "v"(hi), "v"(lo), "n"(J * WH_POS / 512),
                 "n"((J * WH_POS + WH_PLANE) / 512)
                 : "memory");
  };
  // the thread's halo pair of stage JW: positions JW (low halves of the packed words) and JW + 3 (high halves)
  float hvA = 0.f, hvB = 0.f, hwA = 0.f, hwB = 0.f;
  unsigned hpk = 0u, lpk = 0u;
  auto hP = [&](auto jwc) {
    constexpr int JW = decltype(jwc)::value;
    hvA = wx4_pos_s<JW>(dh);
    hvB = wx4_pos_s<JW + 3>(dh);
  };
  auto hHi = [&]() {
    amax = fmaxf(amax, fmaxf(fabsf(hvA), fabsf(hvB)));
    hpk = cvtpk(hvA, hvB);
  };
  auto hSub = [&]() {
    hwA = subhi<0>(hvA, hpk);
    hwB = subhi<1>(hvB, hpk);
  };
  auto hLo = [&]() { lpk = cvtpk(hwA, hwB); };
  const unsigned st_halo = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(v_lds + ith.dst);
  auto hSt = [&](auto jwc) {
    constexpr int JW = decltype(jwc)::value;
    const unsigned h = hpk, l = lpk, ad = st_halo;
    asm volatile("ds_write_b16 %0, %1 offset:%3\n\tds_write_b16 %0, %2 offset:%4\n\t"
                 "ds_write_b16_d16_hi %0, %1 offset:%5\n\tds_write_b16_d16_hi %0, %2 offset:%6" ::"v"(ad), "v"(h), "v"(l),
                 "n"(JW * WH_POS), "n"(JW * WH_POS + WH_PLANE), "n"((JW + 3) * WH_POS), "n"((JW + 3) * WH_POS + WH_PLANE)
                 : "memory");
  };
  // ---- weight DMA: piece q = i*4 + wave of a group -> (jt, slab, hi|lo) in LDS order; source = [slab][chunk][position][dy][hi|lo][1 KB]
  const size_t slab_bytes = (size_t)nch * WH_CHUNK_BYTES;
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wimg + (size_t)(a.slab_base + cb * NREP) * slab_bytes), 0,
                                                     (int)(NREP * slab_bytes), 0x00020000);
  int poff[NREP], pdst[NREP];
#pragma unroll
  for (int i = 0; i < NREP; ++i) {
    const int qd = i * 4 + wave;
    const int jq = qd / (2 * NREP), r = qd - jq * (2 * NREP);
    poff[i] = __builtin_amdgcn_readfirstlane((r >> 1) * (int)slab_bytes + jq * (3 * 6144) + (r & 1) * 1024);
    pdst[i] = __builtin_amdgcn_readfirstlane(qd * 1024);
  }
  (void)wrs; (void)lane16;                          // (only the device pass uses them: the builtin below is compiled out of the host pass)
  auto dma_piece = [&](int i, int src_off, char* wb) {      // src_off = chunk * 36 KB + ji * 6 KB + dy * 2 KB
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(wb + pdst[i]), 16, lane16, src_off + poff[i], 0, 0);
#endif
  };

  // ---- fragment addressing
  int boff[3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int row = 4 * rb + dy + (l31 >> 3);
    boff[dy] = ((4 * rb + dy) * 8 + l31) * 32 + ((lhi ^ (row & 1)) << 4);
  }
  const int a_base = jt * (NREP * 2048) + lane * 16;
  const char* const vjt = v_lds + jt * 3 * WH_POS;

  f32x16 acc[3][NREP];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][nr][r] = 0.f;

  // ---- prologue: the three groups of stage 0 -> ring slots 0,1,2; chunk 0's pixels -> positions {0,3} and {1,4}
  const int nbase = a.slab_base * 32 + cb * NB;
  float sbv = 0.f;
  if (tid < NB) sbv = a.inv_scale[nbase + tid];
  else if (tid < 2 * NB && a.bias) sbv = a.bias[nbase + tid - NB];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int i = 0; i < NREP; ++i) dma_piece(i, dy * 2048, w_lds + dy * GRP);
  ldp(WX_I(0)); ldp(WX_I(1)); ldp(WX_I(2)); ldp(WX_I(3)); ldp(WX_I(4)); ldp(WX_I(5));
  ldh(WX_I(0)); ldh(WX_I(1)); ldh(WX_I(2)); ldh(WX_I(3)); ldh(WX_I(4)); ldh(WX_I(5));
  ldsft();
  if constexpr (PRE >= 1) {
    pr(WX_I(0)); pr(WX_I(1)); pr(WX_I(2)); pr(WX_I(3)); pr(WX_I(4)); pr(WX_I(5));
    prH();
  }
  pA(WX_I(0), WX_I(0)); pV(WX_I(0), WX_I(0)); pHi(WX_I(0)); pSub(WX_I(0)); pLo(WX_I(0)); pSt(WX_I(0), WX_I(0));
  pA(WX_I(1), WX_I(3)); pB(WX_I(1), WX_I(3)); pV(WX_I(1), WX_I(3)); pHi(WX_I(1)); pSub(WX_I(1)); pLo(WX_I(1)); pSt(WX_I(1), WX_I(3));
  pA(WX_I(0), WX_I(1)); pB(WX_I(0), WX_I(1)); pV(WX_I(0), WX_I(1)); pHi(WX_I(0)); pSub(WX_I(0)); pLo(WX_I(0)); pSt(WX_I(0), WX_I(1));
  pA(WX_I(1), WX_I(4)); pB(WX_I(1), WX_I(4)); pV(WX_I(1), WX_I(4)); pHi(WX_I(1)); pSub(WX_I(1)); pLo(WX_I(1)); pSt(WX_I(1), WX_I(4));
  hP(WX_I(0)); hHi(); hSub(); hLo(); hSt(WX_I(0));
  hP(WX_I(1)); hHi(); hSub(); hLo(); hSt(WX_I(1));
  if (tid < 2 * NB) sb_lds[tid] = sbv;
  if constexpr (2 * NB > 256) {                     // (five slabs: 320 table entries for 256 threads)
    const int t2 = tid + 256;
    if (t2 < 2 * NB) sb_lds[t2] = a.bias ? a.bias[nbase + t2 - NB] : 0.f;
  }
  if constexpr (PRE == 2) {
    for (int i = tid; i < a.Cin; i += 256) { sft_lds[i] = imul[i]; sft_lds[a.Cin + i] = iadd[i]; }
  }
  __syncthreads();
  TSTAMP(1);

  // One stage = positions {ji, 3+ji} of chunk c as three groups dy of 3*NREP MFMAs per wave; the body is the generated issue schedule
  // WX4H_STAGE_<NREP>_<ji>_<PRE> (conv_f16_wx4h_sched.inc).  Staging per stage as conv_wx4:
  //   stage 0: positions {2,5} of chunk c (+ halo pair 2), then the pixels of chunk c+1 are requested;  stage 1: pre-activation of
  //   chunk c+1, positions {0,3};   stage 2: positions {1,4}.
  constexpr int NIT = 8;                            // epilogue items of a thread: the tile's rows
  constexpr bool EPF = EPI == 1 || EPI == 2;        // ONE operand tile (residual or mask): prefetched slab by slab
  unsigned yoff[NIT];
  f32x4 op1[EPF ? NREP : 1][NIT];
  decltype(__builtin_amdgcn_make_buffer_rsrc((float*)nullptr, 0, 0, 0)) op1rs;
  auto epf = [&](auto ic) {
    constexpr int it = decltype(ic)::value;
    if constexpr (EPF) op1[0][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(op1rs, yoff[it], 0, 2));        // (nt: a residual / mask byte is read exactly once -- conv_f16_wx4.hip, WX4_RES_AUX)
  };
  auto stage = [&](int c, auto jic, auto finc) {
    constexpr int ji = decltype(jic)::value;
    constexpr bool fin = decltype(finc)::value;
    const int n0 = c + 3 * ji;                       // ring slot of group (c, ji, dy) = (9c + 3ji + dy) & 3 = (n0 + dy) & 3
    const char* const wb0 = w_lds + ((n0 + 0) & 3) * GRP + a_base;
    const char* const wb1 = w_lds + ((n0 + 1) & 3) * GRP + a_base;
    const char* const wb2 = w_lds + ((n0 + 2) & 3) * GRP + a_base;
    // the next stage's group dy goes to the slot of group n + 3
    char* const wn0 = w_lds + ((n0 + 3) & 3) * GRP;
    char* const wn1 = w_lds + ((n0 + 4) & 3) * GRP;
    char* const wn2 = w_lds + ((n0 + 5) & 3) * GRP;
    const int src_next = ji < 2 ? c * WH_CHUNK_BYTES + (ji + 1) * 6144 : (c + 1) * WH_CHUNK_BYTES;
    const char* const vb = vjt + ji * WH_POS;
    if constexpr (ji == 0 && !fin) ld_so = (c + 1) * 64;
    h8 ah[3 * NREP], al[3 * NREP], bh[3], bl[3];
    auto rdA = [&](auto gc) {
      constexpr int g = decltype(gc)::value;
      constexpr int dy = g / NREP, nr = g - dy * NREP;
      const char* const wb = dy == 0 ? wb0 : dy == 1 ? wb1 : wb2;
      ah[g] = *reinterpret_cast<const h8*>(wb + (nr * 2 + 0) * 1024);
      al[g] = *reinterpret_cast<const h8*>(wb + (nr * 2 + 1) * 1024);
    };
    auto rdB = [&](auto dc) {
      constexpr int dy = decltype(dc)::value;
      bh[dy] = *reinterpret_cast<const h8*>(vb + boff[dy]);
      bl[dy] = *reinterpret_cast<const h8*>(vb + WH_PLANE + boff[dy]);
    };
    auto dma = [&](auto dc, auto ic) {
      constexpr int dy = decltype(dc)::value;
      dma_piece(decltype(ic)::value, src_next + dy * 2048, dy == 0 ? wn0 : dy == 1 ? wn1 : wn2);
    };
    auto mfma = [&](auto gc, auto pc_) {
      constexpr int g = decltype(gc)::value, part = decltype(pc_)::value;
      constexpr int dy = g / NREP, nr = g - dy * NREP;
      const h8 wa = part == 0 ? al[g] : ah[g];
      const h8 xv = part == 1 ? bl[dy] : bh[dy];
      acc[ji][nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xv, acc[ji][nr], 0, 0, 0);
    };
    // end of group dy < 2: this wave's pieces of the NEXT group have landed (K younger vector-memory operations may stay in flight),
    // then the workgroup barrier publishes every wave's pieces and frees this group's ring slot.  The waits are the BUILTIN, which
    // hipcc's wait-count pass can see (conv_f16_wx4.hip).
    auto gbar = [&](auto, auto kc, auto ec) {
      constexpr int K = decltype(kc)::value - (EPF ? 0 : decltype(ec)::value);     // (ec: operand-tile loads the schedule counted)
      __builtin_amdgcn_s_waitcnt((K & 15) | 0x0F70 | ((K >> 4) << 14));          // vmcnt(K), lgkmcnt untouched
      asm volatile("s_barrier" ::: "memory");
    };
    // end of the stage: additionally this wave's V stores are done (lgkmcnt(0)); K < 0: the K loop's last stage, no piece pending
    auto gend = [&](auto kc) {
      constexpr int K = decltype(kc)::value;
      if constexpr (K < 0) __builtin_amdgcn_s_waitcnt(0xC07F);                   // lgkmcnt(0) only
      else __builtin_amdgcn_s_waitcnt((K & 15) | 0x0070 | ((K >> 4) << 14));     // vmcnt(K) lgkmcnt(0)
      asm volatile("s_barrier" ::: "memory");
    };
#define WX_TS(g) do { } while (0)
#define WXH_STAGE_CASE(N_, J_, P_) if constexpr (!fin && NREP == N_ && ji == J_ && PRE == P_) { WX4H_STAGE_##N_##_##J_##_##P_ }
#define WXH_STAGE_PRE(N_, J_) WXH_STAGE_CASE(N_, J_, 0) WXH_STAGE_CASE(N_, J_, 1) WXH_STAGE_CASE(N_, J_, 2) \
    if constexpr (fin && NREP == N_ && ji == J_) { WX4H_FINAL_##N_##_##J_ }
    WXH_STAGE_PRE(1, 0) WXH_STAGE_PRE(1, 1) WXH_STAGE_PRE(1, 2)
    WXH_STAGE_PRE(2, 0) WXH_STAGE_PRE(2, 1) WXH_STAGE_PRE(2, 2)
    WXH_STAGE_PRE(3, 0) WXH_STAGE_PRE(3, 1) WXH_STAGE_PRE(3, 2)
    WXH_STAGE_PRE(5, 0) WXH_STAGE_PRE(5, 1) WXH_STAGE_PRE(5, 2)
#undef WXH_STAGE_PRE
#undef WXH_STAGE_CASE
#undef WX_TS
  };
  using No = std::false_type;
  using Yes = std::true_type;
  for (int c = 0; c + 1 < nch; ++c) {
    stage(c, WX_I(0), No{});
    stage(c, WX_I(1), No{});
    stage(c, WX_I(2), No{});
  }
  // (thread coordinates re-derived from the scalar wave index + lane count through an opaque copy: derived from `tid` they stay live
  // across the K loop and the PRE 2 instantiations parked 10-11 of them in scratch -- conv_f16_wx4.hip has the same lines)
  int tid_e = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(tid_e));
#endif
  stage(nch - 1, WX_I(0), Yes{});
  stage(nch - 1, WX_I(1), Yes{});
  // epilogue reader: thread = (pixel column x of the tile, channel quad cq), items it = rows.  One 32-bit byte offset per item serves
  // the operand loads and the stores (an item outside the image gets an out-of-range offset: loads 0, stores nothing).
  const int lane_e = tid_e & 63, l31_e = lane_e & 31, lhi_e = lane_e >> 5;
  const int cq = tid_e & 7, px = tid_e >> 3;
  const int te_row = tid_e >> 5, te_xq = (tid_e >> 3) & 3;       // TE mapping: tile row 0..7, x-segment 0..3 (items = its 8 pixels)
  const int C = a.cout;
  const size_t img_off = (size_t)img * a.H * a.W * C;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int oy = TE ? oy0 + te_row : oy0 + it, ox = TE ? ox0 + te_xq * 8 + it : ox0 + px;
    yoff[it] = (oy < a.H && ox < a.W) ? (unsigned)((oy * a.W + ox) * C + nbase + cq * 4) * 4u : 0x80000000u;
  }
  if constexpr (EPF) op1rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((EPI == 1 ? a.res : a.mask) + img_off), 0, a.H * a.W * C * 4, 0x00020000);
  stage(nch - 1, WX_I(2), Yes{});
  TSTAMP(2);
#ifdef VIRNET_F16_TIMING
  if (a.tlog && tid == 0) {
    a.tlog[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);     // HW_REG_HW_ID
    a.tlog[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
  }
#endif
  range_report(a.range_flag, amax);

  // ---- epilogue (conv_f16_wx4.hip).  Per slab: wave (jt, rb) writes three blocks of [column = (row, x-tile)][32 channels] records
  //   jt = 0: A0 = M0+M1+M2, A1 = M1-M2, A2 = M1+M2        jt = 1: S = M3+M4, D = M3-M4, E = M5
  // and pixel k of an x-tile is  k=0: A0 + S   k=1: A1 + 2D   k=2: A2 + 4S   k=3: A1 + 8D + E   (rows of AT).
  char* const xb = smem;
  const int wblk = (rb * 6 + jt * 3) * WH_XBLK + l31_e * 144 + lhi_e * 16;
  auto put_block = [&](int which, const f32x16& m) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(xb + wblk + which * WH_XBLK + g * 32) = f32x4{m[4 * g], m[4 * g + 1], m[4 * g + 2], m[4 * g + 3]};
  };
  auto xwrite = [&](int nr) {
    f32x16 b0, b1, b2;
    if (jt == 0) {
      b2 = acc[1][nr] + acc[2][nr];
      b1 = acc[1][nr] - acc[2][nr];
      b0 = acc[0][nr] + b2;
    } else {
      b0 = acc[0][nr] + acc[1][nr];
      b1 = acc[0][nr] - acc[1][nr];
      b2 = acc[2][nr];
    }
    put_block(0, b0); put_block(1, b1); put_block(2, b2);
  };
  const int pk = px & 3, pxt = px >> 2;
  const int r_p = ((pk == 0) ? 0 : (pk == 2) ? 2 : 1) * WH_XBLK + pxt * 144 + cq * 16;
  const int r_q = (3 + (pk & 1)) * WH_XBLK + pxt * 144 + cq * 16;
  const int r_e = 5 * WH_XBLK + pxt * 144 + cq * 16;
  const float ck = (float)(1 << pk), ek = pk == 3 ? 1.f : 0.f;
  const int te_base = (te_row >> 2) * 6 * WH_XBLK + ((te_row & 3) * 8) * 144 + cq * 16;
  auto xread = [&](int it) {                                // row it: row block it>>2, row-in-block it&3
    if constexpr (TE) {                                     // pixel te_xq*8 + it of row te_row: x-tile te_xq*2 + (it>>2), pixel-in-tile it&3
      const int pkk = it & 3;
      const int b0 = te_base + (te_xq * 2 + (it >> 2)) * 144;
      const f32x4 p = *reinterpret_cast<const f32x4*>(xb + b0 + ((pkk == 0) ? 0 : (pkk == 2) ? 2 : 1) * WH_XBLK);
      const f32x4 qv = *reinterpret_cast<const f32x4*>(xb + b0 + (3 + (pkk & 1)) * WH_XBLK);
      f32x4 r = p + (float)(1 << pkk) * qv;
      if (pkk == 3) r += *reinterpret_cast<const f32x4*>(xb + b0 + 5 * WH_XBLK);
      return r;
    }
    const int base = (it >> 2) * 6 * WH_XBLK + ((it & 3) * 8) * 144;
    const f32x4 p = *reinterpret_cast<const f32x4*>(xb + base + r_p);
    const f32x4 qv = *reinterpret_cast<const f32x4*>(xb + base + r_q);
    const f32x4 e = *reinterpret_cast<const f32x4*>(xb + base + r_e);
    return p + ck * qv + ek * e;
  };
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  auto mask4 = [&](f32x4 v, f32x4 m) {
    return f32x4{m.x > 0.f ? v.x : v.x * a.mask_slope, m.y > 0.f ? v.y : v.y * a.mask_slope,
                 m.z > 0.f ? v.z : v.z * a.mask_slope, m.w > 0.f ? v.w : v.w * a.mask_slope};
  };
  auto inv_of = [&](int nr) { return *reinterpret_cast<const f32x4*>(sb_lds + nr * 32 + cq * 4); };
  auto bias_of = [&](int nr) { return *reinterpret_cast<const f32x4*>(sb_lds + NB + nr * 32 + cq * 4); };
  if constexpr (EPI < 4) {
    // ONE stored tensor; operand tiles requested a slab ahead and every load of a slab waited for before its first store (conv_f16_wx4.hip)
    constexpr bool RES = (EPI & 1) != 0, MASK = (EPI & 2) != 0;
    float* const y = (a.y_act ? a.y_act : a.y_raw) + img_off;
    const float slope_eff = a.y_act ? a.slope : 1.f;
    auto load_op1 = [&](int nr) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        op1[EPF ? nr : 0][it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(op1rs, yoff[it] + nr * 128, 0, 2));
    };
    xwrite(0);
    if (EPF && NREP > 1) load_op1(1);
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y, 0, a.H * a.W * C * 4, 0x00020000);
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      if (nr > 0) xwrite(nr);
      wx_lds_barrier();
      if (nr == 0) TSTAMP(6);
      f32x4 mv[NIT], rv[NIT];
      if constexpr (EPI == 3) {
        const auto mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.mask + img_off), 0, a.H * a.W * C * 4, 0x00020000);
        const auto rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res + img_off), 0, a.H * a.W * C * 4, 0x00020000);
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          mv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(mrs, yoff[it] + nr * 128, 0, 0));
          rv[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rrs, yoff[it] + nr * 128, 0, 0));
        }
      }
      const f32x4 i4 = inv_of(nr), b4 = bias_of(nr);
      f32x4 tv[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) tv[it] = xread(it);
      SB();
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        f32x4 v = tv[it] * i4 + b4;
        if (MASK) v = mask4(v, EPI == 3 ? mv[it] : op1[EPF ? nr : 0][it]);
        if (RES) v += EPI == 3 ? rv[it] : op1[EPF ? nr : 0][it];
        tv[it] = lrelu4(v, slope_eff);
      }
      SB();
      if (EPF && nr + 2 < NREP) load_op1(nr + 2);
      SB();
      if (a.store_nt) {                                    // (non-temporal stores for tensors larger than the Infinity Cache: conv_f16_wx4.hip)
#pragma unroll
        for (int it = 0; it < NIT; ++it) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tv[it]), yrs, yoff[it] + nr * 128, 0, 2);
      } else {
#pragma unroll
        for (int it = 0; it < NIT; ++it) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tv[it]), yrs, yoff[it] + nr * 128, 0, 0);
      }
      if constexpr (TE) {
        const int cbg = (nbase >> 5) + nr;                  // 32-channel block of the stored tensor
#ifdef VIRNET_TE_DIRECT
        const int trow = oy0 + te_row;
        char* const tb = a.t_out + ((((size_t)img * (a.H + 2) + trow + 1) * a.t_cb + cbg) * a.t_npl * a.t_nseg + ((ox0 >> 3) + te_xq + 1)) * 512 + cq * 64;
#endif
        f32x4 cs = zero4;
        u32x4 uh[4], ul[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float e8[8];
#pragma unroll
          for (int it = 0; it < NIT; ++it) {
            const float sv = yoff[it] != 0x80000000u ? tv[it][c] : 0.f;      // (tile pixels beyond the image are zero in T)
            cs[c] += sv;
            e8[it] = a.t_act ? fmaxf(sv, sv * a.t_slope) : sv;
          }
          t_units(e8, false, uh[c], ul[c]);
        }
#ifdef VIRNET_TE_DIRECT
        if (trow < a.H) {                                   // (A/B build: every lane stores its own 4 x 16 B per plane, 64 B apart from its neighbour's)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(tb + i * 16) = uh[i];
            *reinterpret_cast<u32x4*>(tb + (size_t)a.t_nseg * 512 + i * 16) = ul[i];
          }
        }
#else
        // Re-coalesce the slab's 2048 units through LDS (the exchange region is free once every thread has read its items): written in T's own
        // order [row][plane][x-segment][32 ch], read back lane-linear, so that a store instruction covers 1 KB of contiguous T instead of 64
        // pieces of 16 B that sit 64 B apart (the emitted image's stores were the whole cost of the emission: profiles/r04_probes.md 8)
        wx_lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          *reinterpret_cast<u32x4*>(xb + ((((te_row * 2 + 0) * 4 + te_xq) * 32 + 4 * cq + i) << 4)) = uh[i];
          *reinterpret_cast<u32x4*>(xb + ((((te_row * 2 + 1) * 4 + te_xq) * 32 + 4 * cq + i) << 4)) = ul[i];
        }
        wx_lds_barrier();
        {
          char* const tile0 = a.t_out + ((((size_t)img * (a.H + 2) + oy0 + 1) * a.t_cb + cbg) * a.t_npl * a.t_nseg + ((ox0 >> 3) + 1)) * 512;
          const size_t trow_bytes = (size_t)a.t_cb * a.t_npl * a.t_nseg * 512;
          const int uch = tid_e & 31, uxq = (tid_e >> 5) & 3, upl = tid_e >> 7;         // unit u = k*256 + tid: row k, plane tid>>7, x-segment, channel
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(xb + ((k * 256 + tid_e) << 4));
            if (oy0 + k < a.H)
              *reinterpret_cast<u32x4*>(tile0 + k * trow_bytes + (size_t)upl * a.t_nseg * 512 + uxq * 512 + uch * 16) = v;
          }
        }
#endif
        if (a.t_col) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            cs[c] += __shfl_xor(cs[c], 8);
            cs[c] += __shfl_xor(cs[c], 16);
            cs[c] += __shfl_xor(cs[c], 32);
          }
          if (lane_e < 8) *reinterpret_cast<f32x4*>(a.t_col + ((size_t)cbg * a.t_nblk + (size_t)tile * 4 + wave) * 32 + cq * 4) = cs;
        }
      }
      if (nr == 0) TSTAMP(7);
      if (nr + 1 < NREP) wx_lds_barrier();
    }
  } else {
    // generic form (two stored tensors and / or SFT on the output): optional operands by runtime pointer
    const char* const rimg = a.res ? reinterpret_cast<const char*>(a.res + img_off) : nullptr;
    const char* const mimg = a.mask ? reinterpret_cast<const char*>(a.mask + img_off) : nullptr;
    char* const yraw = a.y_raw ? reinterpret_cast<char*>(a.y_raw + img_off) : nullptr;
    char* const yact = a.y_act ? reinterpret_cast<char*>(a.y_act + img_off) : nullptr;
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      xwrite(nr);
      wx_lds_barrier();
      const int co = nbase + nr * 32 + cq * 4;
      const f32x4 inv4 = inv_of(nr), bias4 = bias_of(nr);
      f32x4 mul4 = f32x4{1.f, 1.f, 1.f, 1.f}, add4 = zero4;
      if (a.mul) {                                           // SFT on the output (AttResUNet.py:57-58): SISR down path
        mul4 = *reinterpret_cast<const f32x4*>(a.mul + (size_t)img * C + co);
        add4 = *reinterpret_cast<const f32x4*>(a.add + (size_t)img * C + co);
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        f32x4 v = xread(it) * inv4 + bias4;
        if (yoff[it] != 0x80000000u) {
          const unsigned o = yoff[it] + nr * 128;
          if (mimg) v = mask4(v, *reinterpret_cast<const f32x4*>(mimg + o));
          if (rimg) v += *reinterpret_cast<const f32x4*>(rimg + o);
          if (yraw) *reinterpret_cast<f32x4*>(yraw + o) = v;
          if (yact) *reinterpret_cast<f32x4*>(yact + o) = lrelu4(v * mul4 + add4, a.slope);
        }
      }
      if (nr + 1 < NREP) wx_lds_barrier();
    }
  }
  TSTAMP(3);
}

template <int NREP, int EPI, int PRE, int TE = 0>
int launch_wx4h_t(FArgs k, hipStream_t st) {
  constexpr int LDS_K = WH_VBYTES + 4 * 4 * NREP * 1024;
  constexpr int LDS_E = 12 * WH_XBLK;
  constexpr int LDS = (LDS_K > LDS_E ? LDS_K : LDS_E) + 2 * 32 * NREP * 4;      // + the channel block's inverse scales and biases
  static_assert(LDS <= (NREP > 3 ? 158 : 80) * 1024, "two workgroups per CU (five-slab form: one)");
  static unsigned long long attr_done = 0;
  auto kern = conv_wx4h_kernel<NREP, EPI, PRE, TE>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_wx4h): %s", hipGetErrorString(e));
  }
  k.nty = (k.H + 7) / 8;
  k.ntx = (k.W + 31) / 32;
  k.ntiles = k.N * k.nty * k.ntx;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  const int ncb = k.NP / (32 * NREP);
  const unsigned grid = (unsigned)(8 * k.tiles_per_xcd * ncb);
  if ((unsigned long long)grid * (unsigned)ncb >= (1ull << 32) || (unsigned long long)k.ntiles * (unsigned)(k.ntx * k.nty) >= (1ull << 32))
    return virnet::set_error("virnet_conv_wx4: %d tiles x %d channel blocks exceed the index arithmetic of one launch", k.ntiles, ncb);
  k.mg_ncb = div_magic(ncb);
  k.mg_ntx = div_magic(k.ntx);
  k.mg_tpi = div_magic(k.ntx * k.nty);
  const int lds = LDS + (PRE == 2 ? 2 * k.Cin * 4 : 0);                          // (+ the image's SFT vectors)
  if (lds > 80 * 1024 + 4096 && lds > 160 * 1024) return virnet::set_error("virnet_conv_wx4 (8-row tiles): %d input channels of SFT vectors do not fit LDS", k.Cin);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, k);
  return virnet::check_launch("conv_wx4h launch");
}

}  // namespace

namespace virnet {

int launch_wx4h_emit(FArgs k, int nrep, int epi, int pre, hipStream_t st) {
#define VIRNET_WX4H_TE(N_, E_) if (nrep == N_ && epi == E_) return pre == 1 ? launch_wx4h_t<N_, E_, 1, 1>(k, st) : launch_wx4h_t<N_, E_, 0, 1>(k, st);
#define VIRNET_WX4H_TEN(N_) VIRNET_WX4H_TE(N_, 0) VIRNET_WX4H_TE(N_, 1) VIRNET_WX4H_TE(N_, 2) VIRNET_WX4H_TE(N_, 3)
  VIRNET_WX4H_TEN(3) VIRNET_WX4H_TEN(2) VIRNET_WX4H_TEN(1)
#undef VIRNET_WX4H_TEN
#undef VIRNET_WX4H_TE
  return virnet::set_error("virnet_conv_wx4_emit (8-row tiles): no emitting kernel for nrep=%d epi=%d pre=%d", nrep, epi, pre);
}

int launch_wx4h(FArgs k, int nrep, int epi, int pre, hipStream_t st) {
#define VIRNET_WX4H_EPI(N_, E_)                                                                                               \
  if (epi == E_) return pre == 2 ? launch_wx4h_t<N_, E_, 2>(k, st) : pre == 1 ? launch_wx4h_t<N_, E_, 1>(k, st) : launch_wx4h_t<N_, E_, 0>(k, st);
#define VIRNET_WX4H_CASE(N_)                                                                             \
  if (nrep == N_) {                                                                                      \
    VIRNET_WX4H_EPI(N_, 0) VIRNET_WX4H_EPI(N_, 1) VIRNET_WX4H_EPI(N_, 2) VIRNET_WX4H_EPI(N_, 3) VIRNET_WX4H_EPI(N_, 4)                    \
  }
  VIRNET_WX4H_CASE(3) VIRNET_WX4H_CASE(2) VIRNET_WX4H_CASE(1) VIRNET_WX4H_CASE(5)
#undef VIRNET_WX4H_CASE
#undef VIRNET_WX4H_EPI
  return virnet::set_error("virnet_conv_wx4 (8-row tiles): no kernel for nrep=%d", nrep);
}

}  // namespace virnet
