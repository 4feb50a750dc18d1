// wino.hip -- the stride-1 3x3 convolution in Winograd F(2x2,3x3) form on the CDNA4 matrix cores (gfx950), fp32.
//
// Same call sites as conv_mfma.hip's KS=3,S=1 instantiations (AttResBlock.conv1/conv2, networks/AttResUNet.py:43,46,55,58;
// DnCNN mid convs, networks/DnCNN.py:25-28) and their input-gradient GEMMs: Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A with
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. 16 multiplies per 2x2 output tile and channel pair instead of 36.  The fp32 matrix pipe is the bound of this path
// (DESIGN.md 5), so the 2.25x cut in MFMA work is the lever; in fp32 the transform's rounding error is at the level of the direct
// form's own re-association error (measured against an fp64 run of the whole network: 1.0e-5 vs 0.9e-5 max-abs).
//
// The 16 transform positions are 16 independent GEMMs  M_p[cout][tile] = sum_ci U_p[cout][ci] * V_p[ci][tile].
// Workgroup = CB*TG*2 waves, one of two roles with the same number of accumulator blocks; two sizes are instantiated:
//   8 waves, one workgroup per CU : role A <CB=2,TG=2> 64 channels x  64 tiles (8 x 32 output pixels)
//                                   role B <CB=1,TG=4> 32 channels x 128 tiles (16 x 32 pixels; the 32-channel remainder of 96 / 288)
//   4 waves, two workgroups per CU: role A <2,1> 64 channels x 32 tiles, role B <1,2> 32 channels x 64 tiles (small grids)
// The weight image U is re-read per workgroup and per chunk, so the tile count per workgroup sets the weight traffic.
// A wave owns 32 channels x 32 tiles x 8 of the 16 positions (columns j in {0,1} or {2,3} of the 4x4 position grid):
// 128 accumulator registers.  The two waves of a pair reduce A^T M A over their column halves, exchange half of the result
// through LDS once per tile, and each finalises one output row of the 2x2 tiles.
//
// K loop: chunks of 4 input channels.  Per chunk, double buffered in LDS:
//   raw  : the (2*TR+2) x 34 pixel halo tile, 16 B per pixel, activated on the way in (pre-activation of AttResUNet.py:54-55,
//          zero outside the image AFTER the activation), even/odd columns split so the transform reads contiguous runs
//   V    : B^T d B, [pos][k-half][tile][2]  (one 8-B MFMA fragment pair per lane, conflict free)
//   U    : G g G^T from the packed global image, [pos][k-half][cout][2], copied global -> LDS by the DMA path
//          (global_load_lds_dwordx4: no VGPR round trip, which would cost matrix-pipe time -- profiles/r01_probes.md)
// Iteration c: MFMAs of chunk c read V/U[c&1]; the transform of chunk c+1 runs raw[(c+1)&1] -> V[(c+1)&1]; chunk c+2's pixels
// are fetched global -> registers before the MFMAs and landed after them, chunk c+1's weights stream into U[(c+1)&1]; ONE
// barrier per chunk.
// Fragments are read half a chunk ahead of the MFMAs that use them (positions 4..7 of chunk c-1 run after barrier c-1).
#include "common.h"
#include "../../include/virnet_hip.h"
#include <cstdlib>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct WArgs {
  const float* x;
  const float* up;       // packed U: [slab][chunk][pos 16][half 2][32 cout][2]
  const float* bias;
  const float* res;
  const float* mul;
  const float* add;
  const float* in_mul;
  const float* in_add;
  const float* mask;
  float* y_raw;
  float* y_act;
  int N, H, W, Cin, Cout;
  int nux, nuy, nunits, units_per_xcd, n64, n32;
  int in_act;
  float in_slope, mask_slope, slope;
};

__device__ __forceinline__ f32x4 lrelu4(f32x4 u, float s) {
  const f32x4 t = u * s;
  return f32x4{fmaxf(u.x, t.x), fmaxf(u.y, t.y), fmaxf(u.z, t.z), fmaxf(u.w, t.w)};
}

template <int CB, int TG>
struct Cfg {
  static constexpr int NW = CB * TG * 2;            // waves
  static constexpr int TR = 2 * TG;                 // tile rows
  static constexpr int OHT = 2 * TR;                // output rows
  static constexpr int IH = OHT + 2, IW = 34;
  static constexpr int NPIX = IH * IW;
  static constexpr int RAWB = IH * 2 * 17 * 16;     // [iy][parity][17][16 B]
  static constexpr int VHALF = TG * 256 + 128;      // +128 B: the two k-halves of one transform write land in disjoint banks
  static constexpr int VPOS = 2 * VHALF;
  static constexpr int VB = 16 * VPOS;
  static constexpr int UROW = CB * 256;             // [pos*2+half] rows of CB*32 channels x 8 B
  static constexpr int UB = 32 * UROW;
  static constexpr int STAGE = RAWB + VB + UB;
  static constexpr int XCH = NW * 8192;             // result exchange: 8 KB per wave
  static constexpr int TURN = OHT * 32 * (CB * 128 + 16);   // epilogue turn-around buffer [pixel][CB*32 channels + 16 B]
  static constexpr int LDS = (2 * STAGE > XCH + TURN) ? 2 * STAGE : XCH + TURN;
};

#define SB() __builtin_amdgcn_sched_barrier(0)

template <int CB, int TG, bool SFT>
__device__ __forceinline__ void wino_body(const WArgs& a, char* const smem, const int img, const int oy0, const int ox0,
                                          const int cout_base) {
  using K = Cfg<CB, TG>;
  constexpr int NW = K::NW, NT = NW * 64;
  constexpr int PPT = (K::NPIX + NT - 1) / NT;       // pixels per thread per chunk
  constexpr int UPW = 4 / TG;                        // 1-KB DMA pieces per wave per chunk (CB*8 KB / NW)
  constexpr int XP = 2 / CB;                         // transform items per thread per chunk
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nch = a.Cin >> 2;
#ifdef WINO_TIMING
  const long long tbeg = clock64();
#endif
  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;

  // ---- pixel staging: thread -> pixel(s) of the halo tile.  Loads are always issued from a clamped address; out-of-image
  // pixels become zero on their way into LDS (after the activation: the conv pads the ACTIVATED tensor).
  unsigned poff[PPT];
  int pdst[PPT];
  bool pinb[PPT];
  float pmsk[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = k * NT + tid;
    const bool has = p < K::NPIX;
    const int pc = has ? p : 0;
    const int iy = pc / K::IW, ix = pc - iy * K::IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    pinb[k] = has && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    pmsk[k] = pinb[k] ? 1.f : 0.f;
    const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
    poff[k] = (unsigned)((gyc * a.W + gxc) * a.Cin);
    pdst[k] = has ? ((iy * 2 + (ix & 1)) * 17 + (ix >> 1)) * 16 : -1;
  }
  const float* const imul = SFT ? a.in_mul + (size_t)img * a.Cin : nullptr;
  const float* const iadd = SFT ? a.in_add + (size_t)img * a.Cin : nullptr;
  const float in_slope_eff = a.in_act ? a.in_slope : 1.f;
  auto load_raw1 = [&](int chunk, int k) -> f32x4 { return *reinterpret_cast<const f32x4*>(ximg + chunk * 4 + poff[k]); };
  // lrelu(x*mul+add) then zero outside the image.  Without SFT the image mask is folded into the multiply: lrelu(0) == 0.
  auto store_raw1 = [&](char* dstb, int chunk, int k, f32x4 r) {
    f32x4 v;
    if (SFT) {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(imul + chunk * 4);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(iadd + chunk * 4);
      v = lrelu4(r * m4 + a4, in_slope_eff);
      v = pinb[k] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      v = lrelu4(r * pmsk[k], in_slope_eff);
    }
    if (pdst[k] >= 0) *reinterpret_cast<f32x4*>(dstb + pdst[k]) = v;
  };
  // ---- weight staging: 1-KB piece q of the chunk image [pos*2+half][CB*32 ch][8 B] <- packed [slab][chunk][pos*2+half][32][2].
  // The DMA writes LDS at (wave-uniform base) + lane*16, so lane l of piece q supplies the global address of byte q*1024 + l*16.
  unsigned uoff[UPW];
#pragma unroll
  for (int k = 0; k < UPW; ++k) {
    const int q16 = (wave * UPW + k) * 64 + lane;                // 16-B piece index within the chunk image
    const int row = q16 / (CB * 16), pc = q16 - row * (CB * 16);
    const int ch = pc * 2;                                     // channel pair within the CB*32 block
    const int slab = (cout_base >> 5) + (ch >> 5);
    uoff[k] = (unsigned)(slab * nch * 2048 + row * 64 + (ch & 31) * 2);
  }
  auto dma_one = [&](int chunk, char* dstb, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.up + chunk * 2048 + uoff[k]),
                                     (__attribute__((address_space(3))) void*)(dstb + (wave * UPW + k) * 1024), 16, 0, 0);
  };
  auto dma_u = [&](int chunk, char* dstb) {
#pragma unroll
    for (int k = 0; k < UPW; ++k) dma_one(chunk, dstb, k);
  };
  // ---- input transform: item = (tile, row i of B^T d B, k-half); i = wave%4 (uniform), lanes = (k-half, tile column, row parity)
  const int ti = wave & 3, tq = wave >> 2;
  const int ra = (ti == 0) ? 0 : (ti == 2) ? 2 : 1;
  const int rb = (ti == 3) ? 3 : (ti == 2) ? 1 : 2;
  const float sgn = (ti == 1) ? 1.f : -1.f;
  const int h2 = lane & 1, tcol = (lane >> 1) & 15;
  const int xr_a = (2 * (2 * tq + lhi) + ra) * (2 * 17 * 16) + tcol * 16 + h2 * 8;      // + pass * (NW/4) * 4 rows
  const int xr_b = (2 * (2 * tq + lhi) + rb) * (2 * 17 * 16) + tcol * 16 + h2 * 8;
  const int xw_o = (ti * 4) * K::VPOS + h2 * K::VHALF + ((2 * tq + lhi) * 16 + tcol) * 8;  // + pass * (NW/4) * 2 tile rows
  constexpr int XR_PASS = (NW / 4) * 4 * (2 * 17 * 16), XW_PASS = (NW / 4) * 2 * 16 * 8;
  auto xf_read1 = [&](const char* rawb, int pass, f32x2 (&da)[4], f32x2 (&db)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int o = (b & 1) * (17 * 16) + (b >> 1) * 16 + pass * XR_PASS;
      da[b] = *reinterpret_cast<const f32x2*>(rawb + xr_a + o);
      db[b] = *reinterpret_cast<const f32x2*>(rawb + xr_b + o);
    }
  };
  auto xf_write1 = [&](char* vb, int pass, const f32x2 (&da)[4], const f32x2 (&db)[4]) {
    f32x2 t[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) t[b] = da[b] + sgn * db[b];
    char* const dst = vb + xw_o + pass * XW_PASS;
    *reinterpret_cast<f32x2*>(dst) = t[0] - t[2];
    *reinterpret_cast<f32x2*>(dst + K::VPOS) = t[1] + t[2];
    *reinterpret_cast<f32x2*>(dst + 2 * K::VPOS) = t[2] - t[1];
    *reinterpret_cast<f32x2*>(dst + 3 * K::VPOS) = t[1] - t[3];
  };
  // ---- MFMA role of this wave
  const int cbw = wave % CB;
  const int tg = (wave / CB) % TG;
  const int ph = wave / (CB * TG);                             // position columns {0,1} or {2,3}
  const int a_off = (cbw * 32 + l31) * 8 + lhi * K::UROW + 2 * ph * 2 * K::UROW;   // + (4*i + jj) * 2 * UROW
  const int b_off = (tg * 32 + l31) * 8 + lhi * K::VHALF + 2 * ph * K::VPOS;       // + (4*i + jj) * VPOS
  auto read_frags = [&](const char* ub, const char* vb, int grp, f32x2 (&fa)[4], f32x2 (&fb)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int lp = grp * 4 + k;
      const int pos = (lp >> 1) * 4 + (lp & 1);
      fa[k] = *reinterpret_cast<const f32x2*>(ub + a_off + pos * 2 * K::UROW);
      fb[k] = *reinterpret_cast<const f32x2*>(vb + b_off + pos * K::VPOS);
    }
  };
  f32x16 acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  auto mf = [&](int lp, int ks, const f32x2 (&fa)[4], const f32x2 (&fb)[4]) {
    acc[lp] = __builtin_amdgcn_mfma_f32_32x32x2f32(ks ? fa[lp & 3].y : fa[lp & 3].x, ks ? fb[lp & 3].y : fb[lp & 3].x, acc[lp], 0, 0, 0);
  };

  constexpr int RAW0 = 0, V0 = K::RAWB, U0 = K::RAWB + K::VB;
  // ---- prologue: chunks 0 and 1 of the pixels into LDS (chunk 2 stays in registers: pixels are fetched two iterations before
  // they are landed, so no iteration waits on its own loads), chunk 0 of the weights; transform chunk 0
  f32x4 rrc[PPT];
  {
    dma_u(0, smem + U0);
    f32x4 r0[PPT], r1[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) { r0[k] = load_raw1(0, k); r1[k] = load_raw1(1, k); rrc[k] = load_raw1(2, k); }
#pragma unroll
    for (int k = 0; k < PPT; ++k) { store_raw1(smem + RAW0, 0, k, r0[k]); store_raw1(smem + K::STAGE + RAW0, 1, k, r1[k]); }
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < XP; ++pass) {
    f32x2 da[4], db[4];
    xf_read1(smem + RAW0, pass, da, db);
    xf_write1(smem + V0, pass, da, db);
  }
  __syncthreads();

  f32x2 fa1[4], fb1[4];                                        // fragments of positions 4..7, consumed one iteration later
#ifdef WINO_TIMING
  long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  long long tprev = clock64();
  const long long tstart = tprev;
#define TICK(i) { const long long tn = clock64(); tacc[i] += tn - tprev; tprev = tn; }
#else
#define TICK(i)
#endif
  // One chunk (buffers B compile-time: the loop is unrolled by two so every LDS address is a loop-invariant base + immediate).
  // A wave issues in order and blocks on the busy matrix pipe, so the staging pieces are threaded BETWEEN its 16 MFMAs: each piece
  // runs under the MFMA issued before it, and the other wave of the SIMD (the other workgroup's, in the 4-wave form) feeds the
  // pipe through the gaps.  (Tried and dropped for the 8-wave form: complementary orders for the two waves of a SIMD -- stage
  // first / multiply first -- measured 12 % slower: the staging burst of four waves at once queues up in LDS.)
  auto iteration = [&](int c, auto bsel, auto first) {
    constexpr int B = decltype(bsel)::value;
    constexpr bool FIRST = decltype(first)::value;
    char* const st_cur = smem + B * K::STAGE;
    char* const st_nxt = smem + (B ^ 1) * K::STAGE;
    const bool more1 = c + 1 < nch, more2 = c + 2 < nch, more3 = c + 3 < nch;
    f32x2 fa0[4], fb0[4];
    f32x4 rrn[PPT];
    f32x2 da[XP][4], db[XP][4];
    TICK(5)
    {
      if (!FIRST) mf(4, 0, fa1, fb1);
      SB();
      read_frags(st_cur + U0, st_cur + V0, 0, fa0, fb0);
      SB();
      if (!FIRST) mf(5, 0, fa1, fb1);
      SB();
#pragma unroll
      for (int k = 0; k < PPT; ++k) rrn[k] = load_raw1(more3 ? c + 3 : c, k);
      if (more1) dma_one(c + 1, st_nxt + U0, 0);
      SB();
      if (!FIRST) mf(6, 0, fa1, fb1);
      SB();
      xf_read1(st_nxt + RAW0, 0, da[0], db[0]);
      if (UPW >= 2 && more1) dma_one(c + 1, st_nxt + U0, UPW >= 2 ? 1 : 0);
      SB();
      if (!FIRST) mf(7, 0, fa1, fb1);
      SB();
      if (XP == 2) xf_read1(st_nxt + RAW0, XP - 1, da[XP - 1], db[XP - 1]);
      if (UPW == 4 && more1) dma_one(c + 1, st_nxt + U0, UPW - 2);
      SB();
      TICK(0)
      if (!FIRST) mf(4, 1, fa1, fb1);
      SB();
      if (UPW == 4 && more1) dma_one(c + 1, st_nxt + U0, UPW - 1);
      SB();
      if (!FIRST) mf(5, 1, fa1, fb1);
      SB();
      if (more1) xf_write1(st_nxt + V0, 0, da[0], db[0]);
      SB();
      if (!FIRST) { mf(6, 1, fa1, fb1); mf(7, 1, fa1, fb1); }
      SB();
      if (XP == 2 && more1) xf_write1(st_nxt + V0, XP - 1, da[XP - 1], db[XP - 1]);
      SB();
      TICK(1)
      mf(0, 0, fa0, fb0); mf(1, 0, fa0, fb0);
      SB();
      TICK(2)
      if (more2) store_raw1(st_cur + RAW0, c + 2, 0, rrc[0]);
      SB();
      mf(2, 0, fa0, fb0);
      SB();
      if (PPT == 2 && more2) store_raw1(st_cur + RAW0, c + 2, PPT - 1, rrc[PPT - 1]);
      SB();
      mf(3, 0, fa0, fb0);
      SB();
      read_frags(st_cur + U0, st_cur + V0, 1, fa1, fb1);
      SB();
      TICK(3)
      mf(0, 1, fa0, fb0); mf(1, 1, fa0, fb0); mf(2, 1, fa0, fb0); mf(3, 1, fa0, fb0);
      SB();
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) rrc[k] = rrn[k];
    TICK(4)
#ifdef WINO_TIMING
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    TICK(6)
#endif
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  iteration(0, I0{}, std::true_type{});
  for (int c = 1; c + 1 < nch; c += 2) {
    iteration(c, I1{}, std::false_type{});
    iteration(c + 1, I0{}, std::false_type{});
  }
  iteration(nch - 1, I1{}, std::false_type{});
#ifdef WINO_TIMING
  const long long tloop = clock64();
  long long te[5] = {0, 0, 0, 0, 0};
#define ETICK(i) te[i] = clock64();
#else
#define ETICK(i)
#endif

  // ---- epilogue.  Accumulator side: lane = tile (row tg*2 + l31/16, column l31%16); this wave finalises output row 2*trow + ph,
  // pixels 2*tcol + b; accumulator quad g = 4 consecutive channels 8g + 4*lhi + (0..3).  Stored that way a wave instruction
  // would write 64 separate 16-B pieces (measured: the stores of one tile took ~6k cycles to issue), so the finished tile is
  // turned around in LDS ([pixel][CB*32 channels], 16 B of padding per pixel against bank conflicts) and every thread handles
  // (pixel, channel quad) pieces: 16 (8) consecutive lanes cover one pixel's 256 (128) contiguous bytes in the residual / mask
  // loads and in the stores.  Those loads are issued before the last MFMAs so the exchange covers their latency.
  const int C = a.Cout;
  const size_t img_off = (size_t)img * a.H * a.W * C;
  const float* const rimg = a.res ? a.res + img_off : nullptr;
  const float* const mimg = a.mask ? a.mask + img_off : nullptr;
  float* const yraw = a.y_raw ? a.y_raw + img_off : nullptr;
  float* const yact = a.y_act ? a.y_act + img_off : nullptr;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int QPP = CB * 8;                        // channel quads per pixel in this workgroup's channel block
  constexpr int TPIX = CB * 128 + 16;                // bytes per pixel in the turn-around buffer
  constexpr int EPT = (K::OHT * 32 * QPP) / NT;      // pieces per thread (= 8)
  static_assert((K::OHT * 32 * QPP) % NT == 0 && NT % QPP == 0, "epilogue piece mapping");
  const int ecq = tid % QPP;                         // this thread's channel quad (the same for all its pieces)
  const int eco = cout_base + ecq * 4;
  unsigned eoff[EPT];
  bool eok[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int pix = (k * NT + tid) / QPP;            // pixel within the workgroup tile, row-major 32 wide
    const int oy = oy0 + (pix >> 5), ox = ox0 + (pix & 31);
    eok[k] = oy < a.H && ox < a.W;
    eoff[k] = (unsigned)(min(oy, a.H - 1) * a.W + min(ox, a.W - 1)) * (unsigned)C + (unsigned)eco;
  }
  const f32x4 bias4 = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + eco) : zero4;
  f32x4 rv[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) rv[k] = rimg ? *reinterpret_cast<const f32x4*>(rimg + eoff[k]) : zero4;
  SB();
#pragma unroll
  for (int lp = 4; lp < 8; ++lp) mf(lp, 0, fa1, fb1);
#pragma unroll
  for (int lp = 4; lp < 8; ++lp) mf(lp, 1, fa1, fb1);

  ETICK(0)
  // output transform.  acc[lp], lp = i*2 + jj, holds M[i][2*ph+jj].  T[a][jj] = (A^T M)[a][j]; the pair's halves of
  // Y[a][b] = sum_j T[a][j] A[j][b] are  ph 0: {T0+T1, T1}   ph 1: {T2, -T2-T3}.  This wave finalises output row a = ph.
  f32x16 keep[2], send[2];
  {
    f32x16 t0[2], t1[2];                                       // T[0][jj], T[1][jj]
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      t0[jj] = acc[0 + jj] + acc[2 + jj] + acc[4 + jj];
      t1[jj] = acc[2 + jj] - acc[4 + jj] - acc[6 + jj];
    }
    if (ph == 0) {
      keep[0] = t0[0] + t0[1]; keep[1] = t0[1];                // row a=0
      send[0] = t1[0] + t1[1]; send[1] = t1[1];                // row a=1 -> partner
    } else {
      keep[0] = t1[0]; keep[1] = -t1[0] - t1[1];               // row a=1
      send[0] = t0[0]; send[1] = -t0[0] - t0[1];               // row a=0 -> partner
    }
  }
  {
    char* const mine = smem + wave * 8192 + lane * 16;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(mine + (b * 4 + g) * 1024) =
            f32x4{send[b][4 * g], send[b][4 * g + 1], send[b][4 * g + 2], send[b][4 * g + 3]};
  }
  ETICK(1)
  __syncthreads();
  ETICK(2)
  {
    const char* const theirs = smem + (wave ^ (CB * TG)) * 8192 + lane * 16;
    char* const tbuf = smem + K::XCH;
    const int prow = 2 * (tg * 2 + (l31 >> 4)) + ph;           // output row within the workgroup tile
    char* const tdst = tbuf + (prow * 32 + 2 * (l31 & 15)) * TPIX + (cbw * 32 + 4 * lhi) * 4;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 y = *reinterpret_cast<const f32x4*>(theirs + (b * 4 + g) * 1024) +
                        f32x4{keep[b][4 * g], keep[b][4 * g + 1], keep[b][4 * g + 2], keep[b][4 * g + 3]};
        *reinterpret_cast<f32x4*>(tdst + b * TPIX + g * 32) = y;
      }
  }
  __syncthreads();
  {
    const char* const tsrc = smem + K::XCH + ecq * 16;
    f32x4 mul = f32x4{1.f, 1.f, 1.f, 1.f}, add = zero4;
    if (a.mul) {
      mul = *reinterpret_cast<const f32x4*>(a.mul + (size_t)img * C + eco);
      add = *reinterpret_cast<const f32x4*>(a.add + (size_t)img * C + eco);
    }
    f32x4 mv[EPT];
    if (mimg) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) mv[k] = *reinterpret_cast<const f32x4*>(mimg + eoff[k]);
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int pix = (k * NT + tid) / QPP;
      f32x4 v = *reinterpret_cast<const f32x4*>(tsrc + pix * TPIX) + bias4;
      if (mimg) {
        const f32x4 m = mv[k];
        v = f32x4{m.x > 0.f ? v.x : v.x * a.mask_slope, m.y > 0.f ? v.y : v.y * a.mask_slope,
                  m.z > 0.f ? v.z : v.z * a.mask_slope, m.w > 0.f ? v.w : v.w * a.mask_slope};
      }
      v += rv[k];
      if (eok[k]) {
        if (yraw) *reinterpret_cast<f32x4*>(yraw + eoff[k]) = v;
        if (yact) *reinterpret_cast<f32x4*>(yact + eoff[k]) = lrelu4(v * mul + add, a.slope);
      }
    }
  }
#ifdef WINO_TIMING
  ETICK(3)
  __syncthreads();
  if (lane == 0 && a.add) {      // probe builds: `add` carries a timing buffer, 12 x int64 per wave
    long long* o = reinterpret_cast<long long*>(const_cast<float*>(a.add)) + ((size_t)blockIdx.x * 8 + wave) * 16;
    for (int i = 0; i < 6; ++i) o[i] = tacc[i];
    o[6] = tloop - tstart; o[7] = clock64() - tloop; o[8] = tstart - tbeg; o[9] = CB; o[10] = tacc[6]; o[11] = wave + 1;
    o[12] = te[0] - tloop; o[13] = te[1] - te[0]; o[14] = te[2] - te[1]; o[15] = te[3] - te[2];
  }
#endif
}

// NW8 = true: 8-wave workgroups (roles <2,2> / <1,4>, unit = 16 x 32 output pixels); false: 4-wave workgroups (<2,1> / <1,2>,
// unit = 8 x 32).  Workgroup -> (unit, slot): slots 0..2*n64-1 are role A (64-channel block slot/2, upper/lower half of the unit),
// slot 2*n64 is role B (the 32-channel remainder over the whole unit).  Block b runs on XCD b%8: units are contiguous per XCD and
// the slots of a unit adjacent in time, so the halo tile is fetched from HBM once.
template <bool NW8>
struct Roles {
  static constexpr int TGA = NW8 ? 2 : 1, TGB = NW8 ? 4 : 2;
  static constexpr int UH = 4 * TGB;                // unit height in output rows
  static constexpr int LDS = (Cfg<2, TGA>::LDS > Cfg<1, TGB>::LDS) ? Cfg<2, TGA>::LDS : Cfg<1, TGB>::LDS;
};

// WPU = workgroups per unit (2*n64 + n32) as a compile-time constant for the network's channel counts (64: 2, 96: 3, 192: 6,
// 288: 9), 0 = read it from the arguments.  Besides the cheaper index arithmetic it gives each layer width its own kernel symbol,
// so rocprofv3's per-kernel averages line up with bench.py's per-width HIP-event groups.
template <bool NW8, bool SFT, int WPU>
__global__ __launch_bounds__(NW8 ? 512 : 256, NW8 ? 1 : 2) void conv_wino_kernel(const WArgs a) {
  using R = Roles<NW8>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wpu = WPU ? WPU : 2 * a.n64 + a.n32;
  const int xcd = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int slot = __builtin_amdgcn_readfirstlane(q % wpu);
  const int unit = __builtin_amdgcn_readfirstlane(xcd * a.units_per_xcd + q / wpu);
  if (q / wpu >= a.units_per_xcd || unit >= a.nunits) return;
  const int ux = __builtin_amdgcn_readfirstlane(unit % a.nux);
  const int uy = __builtin_amdgcn_readfirstlane((unit / a.nux) % a.nuy);
  const int img = __builtin_amdgcn_readfirstlane(unit / (a.nux * a.nuy));
  if (slot < 2 * a.n64) {
    const int oy0 = uy * R::UH + (slot & 1) * (R::UH / 2);
    if (oy0 >= a.H) return;
    wino_body<2, R::TGA, SFT>(a, smem, img, oy0, ux * 32, (slot >> 1) * 64);
  } else {
    wino_body<1, R::TGB, SFT>(a, smem, img, uy * R::UH, ux * 32, a.n64 * 64);
  }
}

template <bool NW8, bool SFT, int WPU>
int launch_wino(WArgs k, hipStream_t st) {
  using R = Roles<NW8>;
  static unsigned long long attr_done = 0;
  auto kern = conv_wino_kernel<NW8, SFT, WPU>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, R::LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_wino): %s", hipGetErrorString(e));
  }
  k.nux = (k.W + 31) / 32;
  k.nuy = (k.H + R::UH - 1) / R::UH;
  k.nunits = k.N * k.nux * k.nuy;
  k.units_per_xcd = (k.nunits + 7) / 8;
  const unsigned grid = (unsigned)(8 * k.units_per_xcd * (2 * k.n64 + k.n32));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW8 ? 512 : 256), R::LDS, st, k);
  return virnet::check_launch("conv_wino launch");
}

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
  hipStream_t st = static_cast<hipStream_t>(stream);
  // Two 4-wave workgroups per CU measure faster than one 8-wave workgroup at every network shape (independent workgroups fill
  // each other's barrier / prologue / epilogue gaps); VIRNET_WINO_NW=8 selects the 8-wave form for A/B runs.
  const char* const env_nw = getenv("VIRNET_WINO_NW");        // read per call (tests flip it)
  const bool nw8 = env_nw && atoi(env_nw) == 8;
  const bool sft = d->in_mul != nullptr;
  if (nw8) return sft ? launch_wino<true, true, 0>(k, st) : launch_wino<true, false, 0>(k, st);
  if (sft) return launch_wino<false, true, 0>(k, st);
  switch (2 * k.n64 + k.n32) {
    case 2: return launch_wino<false, false, 2>(k, st);
    case 3: return launch_wino<false, false, 3>(k, st);
    case 6: return launch_wino<false, false, 6>(k, st);
    case 9: return launch_wino<false, false, 9>(k, st);
    default: return launch_wino<false, false, 0>(k, st);
  }
}
