// wino_row.hip -- the Winograd F(2x2,3x3) convolution kernel (math, call sites and weight image: see wino.hip).
//
// GEMM view: 16 independent position GEMMs  M_p[cout][tile] += U_p[cout][ci] * V_p[ci][tile]  on v_mfma_f32_32x32x2_f32, then
// Y = A^T M A per 2x2 output tile.  What shaped the kernel (profiles/r01_probes.md has the measurements):
//   * a wave issues in order and blocks on the busy matrix pipe: only work placed BETWEEN its MFMAs overlaps with them;
//   * LDS returns into VGPRs cost matrix-pipe time, so fragments must not be read twice and intermediates must not bounce through LDS;
//   * the weight image is re-read per workgroup and chunk: its DMA (issue cost and bytes) is the largest single overhead.
//
// Workgroup = G groups of 4 waves.  Wave i of a group owns position ROW i of the 4x4 position grid for the group's whole block
// (role A: 64 channels x 32 tiles = 4 x 32 output pixels, role B: 32 channels x 64 tiles, the 32-channel remainder of 96 / 288;
// 8 accumulator blocks = 128 VGPRs either way):
//   * B fragments V[i][0..3] = (row i of B^T d) B need only TWO rows of the 4x4 input patch: 8 ds_read_b64 + 8 packed VALU ops per
//     32-tile block, formed in registers for the NEXT chunk while this chunk's MFMAs run -- no transformed-input buffer in LDS;
//   * A fragments are read once per (position, 32-channel block): 2.0 VGPR rows per MFMA in total (a column-pair split with a
//     V buffer took 3.0 and measured ~5 % slower);
//   * per 4-channel chunk LDS holds the raw halo tile (16 B per pixel, pre-activation of AttResUNet.py:54-55 applied on the way in,
//     zero outside the image AFTER it, even/odd columns split so patch reads are contiguous runs) and the weight image
//     [pos][k-half][cout][2], double buffered, ONE barrier per chunk; the weights stream global -> LDS by DMA
//     (global_load_lds_dwordx4: no VGPR round trip), the pixels are fetched two chunks before they are landed;
//   * the MFMAs of k-step 1 of a chunk run at the start of the next one, after the barrier, and cover its A-fragment reads; every
//     other piece (patch reads, B arithmetic, DMA issues, pixel stores) is threaded between MFMA pairs (sched_barrier fences);
//     the loop is unrolled by two so every LDS address is a loop-invariant base + immediate.
// G = 1: 4 waves, two workgroups per CU (each fills the other's barrier / prologue / epilogue gaps).  G = 2: 8 waves, one per CU,
// the two groups share the weight image (half the DMA per MFMA) -- faster once the K loop is long (>= 192 input channels).
// Grid: unit = 8*G x 32 output pixels; slots 0..2*n64-1 are role A (64-channel block slot/2, upper/lower half of the unit), slot
// 2*n64 is role B over the whole unit; units are contiguous per XCD (block b runs on XCD b%8) and the slots of a unit adjacent in
// time, so the halo tile comes from HBM once (measured traffic ~ algorithmic).
//
// Epilogue: wave i reduces its row over j, R_i[b] = sum_j M[i][j] A^T[b][j] (64 registers), the four R_i of a group go through LDS
// once, wave (a, block) sums Y[a][b] = sum_i A^T[a][i] R_i[b] for its 32x32 block, and the finished tile is turned around in LDS
// ([pixel][channels], 16 B of padding per pixel) so residual / mask loads and stores are pixel-contiguous 128/256-B runs; then
// bias, LeakyReLU-derivative mask (backward), residual, activation as in conv_mfma.hip.
#include "wino_args.h"
#include <cstdlib>
#include <type_traits>

namespace {
using namespace virnet;

#define SB() __builtin_amdgcn_sched_barrier(0)

// G = wave groups per workgroup (1: 4 waves, two workgroups per CU; 2: 8 waves, one per CU, twice the tiles per weight byte).
// Group g owns tile blocks [g*TG, (g+1)*TG) of the workgroup tile; the weight image is shared by the groups.
template <int CB, int TG, int G>
struct CfgR {
  static constexpr int NB = CB * TG;                // 32x32 blocks per position and wave (2)
  static constexpr int TR = 2 * TG * G, OHT = 2 * TR, IH = OHT + 2, IW = 34, NPIX = IH * IW;
  static constexpr int ROWB = 2 * 17 * 16;          // one pixel row: [parity][17][16 B]
  static constexpr int RAWB = IH * ROWB;
  static constexpr int UROW = CB * 256;             // [pos*2+half] rows of CB*32 channels x 8 B
  static constexpr int UB = 32 * UROW;
  static constexpr int STAGE = RAWB + UB;
  static constexpr int XCH = 4 * G * 16384;         // R exchange: 16 KB per wave
  static constexpr int TPIX = CB * 128 + 16;
  static constexpr int TURN = OHT * 32 * TPIX;      // aliases the exchange area (a barrier apart)
  static constexpr int LDS = (2 * STAGE > XCH) ? 2 * STAGE : XCH;
  static_assert(TURN <= XCH, "turn-around buffer must fit the exchange area");
};

template <int CB, int TG, int G, bool SFT>
__device__ __forceinline__ void wino_row_body(const WArgs& a, char* const smem, const int img, const int oy0, const int ox0,
                                              const int cout_base) {
  using K = CfgR<CB, TG, G>;
  constexpr int NT = 256 * G, NB = K::NB;
  static_assert(NB == 2, "two 32x32 blocks per position");
  constexpr int PPT = (K::NPIX + NT - 1) / NT;
  constexpr int UPW = CB * 2 / G;                     // 1-KB DMA pieces per wave per chunk (CB*8 KB / waves)
  static_assert(UPW >= 1, "weight pieces per wave");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wv & 3;                            // position row i
  const int grp = wv >> 2;                            // tile-block group
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nch = a.Cin >> 2;
  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;
  constexpr int RAW0 = 0, U0 = K::RAWB;

  // ---- pixel staging (as wino.hip): clamped loads two iterations ahead, activation + image mask on the way into LDS
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
    pdst[k] = has ? iy * K::ROWB + (ix & 1) * 272 + (ix >> 1) * 16 : -1;
  }
  const float* const imul = SFT ? a.in_mul + (size_t)img * a.Cin : nullptr;
  const float* const iadd = SFT ? a.in_add + (size_t)img * a.Cin : nullptr;
  const float in_slope_eff = a.in_act ? a.in_slope : 1.f;
  auto load_raw1 = [&](int chunk, int k) -> f32x4 { return *reinterpret_cast<const f32x4*>(ximg + chunk * 4 + poff[k]); };
  auto store_raw1 = [&](char* dstb, int chunk, int k, f32x4 r) {
    f32x4 v;
    if (SFT) {
      const f32x4 m4 = *reinterpret_cast<const f32x4*>(imul + chunk * 4);
      const f32x4 a4 = *reinterpret_cast<const f32x4*>(iadd + chunk * 4);
      v = wino_lrelu4(r * m4 + a4, in_slope_eff);
      v = pinb[k] ? v : f32x4{0.f, 0.f, 0.f, 0.f};
    } else {
      v = wino_lrelu4(r * pmsk[k], in_slope_eff);
    }
    if (pdst[k] >= 0) *reinterpret_cast<f32x4*>(dstb + pdst[k]) = v;
  };
  // ---- weight DMA (as wino.hip): lane l of 1-KB piece q supplies the global address of byte q*1024 + l*16 of the chunk image
  // (a 1-KB piece is 2 (CB=2) or 4 (CB=1) whole [pos*2+half] rows, so piece k of this wave sits k*UPSTEP floats after piece 0)
  constexpr int UPSTEP = (CB == 2) ? 128 : 256;
  unsigned uoff0;
  {
    const int q16 = wv * UPW * 64 + lane;
    const int row = q16 / (CB * 16), pc = q16 - row * (CB * 16);
    const int ch = pc * 2;
    const int slab = (cout_base >> 5) + (ch >> 5);
    uoff0 = (unsigned)(slab * nch * 2048 + row * 64 + (ch & 31) * 2);
  }
  auto dma_one = [&](int chunk, char* dstb, int k) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.up + (chunk * 2048 + k * UPSTEP) + uoff0),
                                     (__attribute__((address_space(3))) void*)(dstb + (wv * UPW + k) * 1024), 16, 0, 0);
  };

  // ---- this wave's position row i = wave: patch rows (ra, rb, sign) of u = (B^T d)[i]
  const int ra = (wave == 0) ? 0 : (wave == 2) ? 2 : 1;
  const int rb = (wave == 3) ? 3 : (wave == 2) ? 1 : 2;
  const float sgn = (wave == 1) ? 1.f : -1.f;
  // A fragments: U[(4*i + j)*2 + lhi][cb*32 + l31][2]
  const int a_off = (wave * 4 * 2 + lhi) * K::UROW + l31 * 8;          // + j * 2 * UROW + cb * 256
  auto read_a = [&](const char* ub, f32x2 (&fa)[4][CB]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) fa[j][cb] = *reinterpret_cast<const f32x2*>(ub + a_off + j * 2 * K::UROW + cb * 256);
  };
  // B fragments of tile block tb: lane = (tile tb*32 + l31: row tb*2 + l31/16, column l31%16; k-half lhi)
  int p_off[TG];
#pragma unroll
  for (int tb = 0; tb < TG; ++tb) p_off[tb] = (2 * ((grp * TG + tb) * 2 + (l31 >> 4))) * K::ROWB + (l31 & 15) * 16 + lhi * 8;
  auto read_patch = [&](const char* rawb, int tb, f32x2 (&da)[4], f32x2 (&db)[4]) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int o = (b & 1) * 272 + (b >> 1) * 16;
      da[b] = *reinterpret_cast<const f32x2*>(rawb + p_off[tb] + ra * K::ROWB + o);
      db[b] = *reinterpret_cast<const f32x2*>(rawb + p_off[tb] + rb * K::ROWB + o);
    }
  };
  auto make_b = [&](const f32x2 (&da)[4], const f32x2 (&db)[4], f32x2 (&fb)[4]) {
    f32x2 u[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) u[b] = da[b] + sgn * db[b];
    fb[0] = u[0] - u[2]; fb[1] = u[1] + u[2]; fb[2] = u[2] - u[1]; fb[3] = u[1] - u[3];
  };

  // accumulators: acc[j][blk], blk = cb (role A) or tb (role B)
  f32x16 acc[4][NB];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][n][r] = 0.f;

  // ---- prologue: pixels of chunks 0, 1 in LDS (chunk 2 carried in registers), weights of chunk 0, B of chunk 0 in registers
  f32x4 rrc[PPT];
  {
#pragma unroll
    for (int k = 0; k < UPW; ++k) dma_one(0, smem + U0, k);
    f32x4 r0[PPT], r1[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) { r0[k] = load_raw1(0, k); r1[k] = load_raw1(1, k); rrc[k] = load_raw1(2, k); }
#pragma unroll
    for (int k = 0; k < PPT; ++k) { store_raw1(smem + RAW0, 0, k, r0[k]); store_raw1(smem + K::STAGE + RAW0, 1, k, r1[k]); }
  }
  __syncthreads();
  f32x2 fb_cur[TG][4];
#pragma unroll
  for (int tb = 0; tb < TG; ++tb) {
    f32x2 da[4], db[4];
    read_patch(smem + RAW0, tb, da, db);
    make_b(da, db, fb_cur[tb]);
  }
  __syncthreads();                                     // iteration 0 lands chunk 2 in the buffer these reads came from
  float fa_oy[4][CB], fb_oy[TG][4];                    // k-step-1 operands of the previous chunk

  // MFMA (j, blk) of k-step 0 / 1
  auto mf0 = [&](int j, int n, const f32x2 (&fa)[4][CB], const f32x2 (&fb)[TG][4]) {
    const int cb = (CB == 2) ? n : 0, tb = (CB == 2) ? 0 : n;
    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j][cb].x, fb[tb][j].x, acc[j][n], 0, 0, 0);
  };
  auto mf1 = [&](int j, int n) {
    const int cb = (CB == 2) ? n : 0, tb = (CB == 2) ? 0 : n;
    acc[j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa_oy[j][cb], fb_oy[tb][j], acc[j][n], 0, 0, 0);
  };

  // One chunk (buffers B compile-time, loop unrolled by two).  Order: k-step 1 of the previous chunk (operands in registers; covers
  // the A reads issued right after the barrier), then k-step 0 of this chunk; the patch reads / B arithmetic of the next chunk,
  // the pixel loads and stores and the DMA issues are threaded between the MFMAs.
  auto iteration = [&](int c, auto bsel, auto first) {
    constexpr int B = decltype(bsel)::value;
    constexpr bool FIRST = decltype(first)::value;
    char* const st_cur = smem + B * K::STAGE;
    char* const st_nxt = smem + (B ^ 1) * K::STAGE;
    const bool more1 = c + 1 < nch, more2 = c + 2 < nch, more3 = c + 3 < nch;
    f32x2 fa[4][CB], fb_next[TG][4];
    f32x2 da[4], db[4];
    f32x4 rrn[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) rrn[k] = load_raw1(more3 ? c + 3 : c, k);
    SB();
    if (!FIRST) { mf1(0, 0); mf1(0, 1); }
    SB();
    read_patch(st_nxt + RAW0, 0, da, db);
    if (more1) dma_one(c + 1, st_nxt + U0, 0);
    SB();
    if (!FIRST) { mf1(1, 0); mf1(1, 1); }
    SB();
    if (UPW >= 2 && more1) dma_one(c + 1, st_nxt + U0, UPW >= 2 ? 1 : 0);
    SB();
    if (!FIRST) { mf1(2, 0); mf1(2, 1); }
    SB();
    make_b(da, db, fb_next[0]);
    if (TG == 2) read_patch(st_nxt + RAW0, TG - 1, da, db);
    if (UPW == 4 && more1) dma_one(c + 1, st_nxt + U0, UPW - 2);
    SB();
    if (!FIRST) { mf1(3, 0); mf1(3, 1); }
    SB();
    read_a(st_cur + U0, fa);                             // (the old k-step-1 operands are dead from here)
    if (UPW == 4 && more1) dma_one(c + 1, st_nxt + U0, UPW - 1);
    SB();
    mf0(0, 0, fa, fb_cur); mf0(0, 1, fa, fb_cur);
    SB();
    if (TG == 2) make_b(da, db, fb_next[TG - 1]);
    SB();
    mf0(1, 0, fa, fb_cur); mf0(1, 1, fa, fb_cur);
    SB();
    if (more2) store_raw1(st_cur + RAW0, c + 2, 0, rrc[0]);
    SB();
    mf0(2, 0, fa, fb_cur); mf0(2, 1, fa, fb_cur);
    SB();
    if (PPT == 2 && more2) store_raw1(st_cur + RAW0, c + 2, PPT - 1, rrc[PPT - 1]);
    SB();
    mf0(3, 0, fa, fb_cur); mf0(3, 1, fa, fb_cur);
    SB();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int cb = 0; cb < CB; ++cb) fa_oy[j][cb] = fa[j][cb].y;
#pragma unroll
      for (int tb = 0; tb < TG; ++tb) { fb_oy[tb][j] = fb_cur[tb][j].y; fb_cur[tb][j] = fb_next[tb][j]; }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) rrc[k] = rrn[k];
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

  // ---- epilogue
  const int C = a.Cout;
  const size_t img_off = (size_t)img * a.H * a.W * C;
  const float* const rimg = a.res ? a.res + img_off : nullptr;
  const float* const mimg = a.mask ? a.mask + img_off : nullptr;
  float* const yraw = a.y_raw ? a.y_raw + img_off : nullptr;
  float* const yact = a.y_act ? a.y_act + img_off : nullptr;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int QPP = CB * 8;                        // channel quads per pixel in this workgroup's channel block
  constexpr int TPIX = K::TPIX;
  constexpr int EPT = (K::OHT * 32 * QPP) / NT;      // (pixel, quad) pieces per thread (= 8)
  static_assert((K::OHT * 32 * QPP) % NT == 0 && NT % QPP == 0, "epilogue piece mapping");
  const int ecq = tid % QPP;
  const int eco = cout_base + ecq * 4;
  unsigned eoff[EPT];
  bool eok[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int pix = (k * NT + tid) / QPP;
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
  for (int j = 0; j < 4; ++j) { mf1(j, 0); mf1(j, 1); }

  // R_i[b][blk] = sum_j M[i][j] A^T[b][j]:  b = 0: M0 + M1 + M2,  b = 1: M1 - M2 - M3   -> LDS [wave][b*NB + blk][g][lane][16 B]
  {
    char* const mine = smem + wv * 16384 + lane * 16;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const f32x16 r0 = acc[0][n] + acc[1][n] + acc[2][n];
      const f32x16 r1 = acc[1][n] - acc[2][n] - acc[3][n];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<f32x4*>(mine + ((0 * NB + n) * 4 + g) * 1024) = f32x4{r0[4 * g], r0[4 * g + 1], r0[4 * g + 2], r0[4 * g + 3]};
        *reinterpret_cast<f32x4*>(mine + ((1 * NB + n) * 4 + g) * 1024) = f32x4{r1[4 * g], r1[4 * g + 1], r1[4 * g + 2], r1[4 * g + 3]};
      }
    }
  }
  __syncthreads();
  // wave (a = wave/2, blk = wave%2): Y[a][b] = sum_i A^T[a][i] R_i[b]:  a = 0: R0 + R1 + R2,  a = 1: R1 - R2 - R3
  const int ya = wave >> 1, yblk = wave & 1;
  f32x4 y[2][4];
  {
    const char* const base = smem + grp * 65536 + lane * 16;
    const int i0 = ya, s1 = ya ? -1 : 1;                 // rows i0, i0+1, i0+2 with signs (+, s1, s1*... ) below
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int o = ((b * NB + yblk) * 4 + g) * 1024;
        const f32x4 r_a = *reinterpret_cast<const f32x4*>(base + (i0 + 0) * 16384 + o);
        const f32x4 r_b = *reinterpret_cast<const f32x4*>(base + (i0 + 1) * 16384 + o);
        const f32x4 r_c = *reinterpret_cast<const f32x4*>(base + (i0 + 2) * 16384 + o);
        // a = 0: R0 + R1 + R2 ; a = 1: R1 - R2 - R3
        y[b][g] = ya ? (r_a - r_b - r_c) : (r_a + r_b + r_c);
      }
    (void)s1;
  }
  __syncthreads();                                      // every wave has read the exchange area: turn the tile around in it
  {
    const int cbw = (CB == 2) ? yblk : 0, tbw = (CB == 2) ? 0 : yblk;
    const int prow = 2 * ((grp * TG + tbw) * 2 + (l31 >> 4)) + ya;
    char* const tdst = smem + (prow * 32 + 2 * (l31 & 15)) * TPIX + (cbw * 32 + 4 * lhi) * 4;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(tdst + b * TPIX + g * 32) = y[b][g];
  }
  __syncthreads();
  {
    const char* const tsrc = smem + ecq * 16;
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
        if (yact) *reinterpret_cast<f32x4*>(yact + eoff[k]) = wino_lrelu4(v * mul + add, a.slope);
      }
    }
  }
}

template <int G>
struct RowRoles {
  static constexpr int LDS = (CfgR<2, 1, G>::LDS > CfgR<1, 2, G>::LDS) ? CfgR<2, 1, G>::LDS : CfgR<1, 2, G>::LDS;
  static constexpr int UH = 8 * G;                   // unit height in output rows
};

template <int G, bool SFT, int WPU>
__global__ __launch_bounds__(256 * G, G == 1 ? 2 : 1) void conv_wino_row_kernel(const WArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int UH = RowRoles<G>::UH;
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
    const int oy0 = uy * UH + (slot & 1) * (UH / 2);
    if (oy0 >= a.H) return;
    wino_row_body<2, 1, G, SFT>(a, smem, img, oy0, ux * 32, (slot >> 1) * 64);
  } else {
    wino_row_body<1, 2, G, SFT>(a, smem, img, uy * UH, ux * 32, a.n64 * 64);
  }
}

template <int G, bool SFT, int WPU>
int launch_row(WArgs k, hipStream_t st) {
  static unsigned long long attr_done = 0;
  auto kern = conv_wino_row_kernel<G, SFT, WPU>;
  constexpr int LDS = RowRoles<G>::LDS;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_wino_row): %s", hipGetErrorString(e));
  }
  k.nux = (k.W + 31) / 32;
  k.nuy = (k.H + RowRoles<G>::UH - 1) / RowRoles<G>::UH;
  k.nunits = k.N * k.nux * k.nuy;
  k.units_per_xcd = (k.nunits + 7) / 8;
  const unsigned grid = (unsigned)(8 * k.units_per_xcd * (2 * k.n64 + k.n32));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * G), LDS, st, k);
  return virnet::check_launch("conv_wino_row launch");
}

}  // namespace

int virnet::launch_wino_row(WArgs k, hipStream_t st, bool sft) {
  // 8-wave workgroups (shared weight image) once the K loop is long and the grid still gives every CU several workgroups;
  // VIRNET_WINO_NW=4|8 forces a form (read per call: tests flip it).
  const char* const env_nw = getenv("VIRNET_WINO_NW");
  const int forced = env_nw ? atoi(env_nw) : 0;
  const long wgs8 = (long)k.N * ((k.W + 31) / 32) * ((k.H + 15) / 16) * (2 * k.n64 + k.n32);
  const bool nw8 = forced ? forced == 8 : (k.Cin >= 192 && wgs8 >= 1024);
  if (nw8) {
    if (sft) return launch_row<2, true, 0>(k, st);
    switch (2 * k.n64 + k.n32) {
      case 6: return launch_row<2, false, 6>(k, st);
      case 9: return launch_row<2, false, 9>(k, st);
      default: return launch_row<2, false, 0>(k, st);
    }
  }
  if (sft) return launch_row<1, true, 0>(k, st);
  switch (2 * k.n64 + k.n32) {
    case 2: return launch_row<1, false, 2>(k, st);
    case 3: return launch_row<1, false, 3>(k, st);
    case 6: return launch_row<1, false, 6>(k, st);
    case 9: return launch_row<1, false, 9>(k, st);
    default: return launch_row<1, false, 0>(k, st);
  }
}
