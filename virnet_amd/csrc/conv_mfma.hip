// conv_mfma.hip -- fp32 implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
// One kernel template covers every dense contraction on VIRNet's forward path:
//   KS=3,S=1  AttResBlock.conv1/conv2 (networks/AttResUNet.py:43,46,55,58), DnCNN convs (networks/DnCNN.py:22-29),
//             RB_Layer convs (networks/KNet.py:32,34), AttResUNet.head/tail (:117-119,:139), KernelNet.tail (KNet.py:49)
//   KS=3,S=2  DownBlock.downsampler (networks/AttResUNet.py:67)
//   KS=1,S=1  UpBlock.upsampler, ConvTranspose2d(k2,s2) == 1x1 GEMM to 4*Cout columns + depth-to-space (AttResUNet.py:80)
//
// GEMM view: D[cout][pixel] += W[cout][k] * X[k][pixel], k = (tap, cin).  v_mfma_f32_32x32x2_f32 is exact fp32 (an fmaf
// chain), so parity with the reference's fp32 conv is at re-association level (~1e-6), far inside the 1e-3 contract.
//
// Workgroup = 4 waves = (4*MREP) output rows x 32 output columns x NB=32*NREP output channels; wave w owns rows
// [w*MREP, (w+1)*MREP) x all NB channels (MREP x NREP accumulator blocks of 32x32); one MFMA column block is 32 consecutive
// pixels of a row.  K is walked as 16-channel chunks x KS*KS taps x 2 half-steps of 8 channels; a wave issues MREP*NREP*4
// MFMAs per half-step.
//
// Operand paths.  Both MFMA operands are read from LDS as ds_read_b128 fragments, one half-step ahead of the MFMAs that use
// them.  (Round-1 profiling on MI355X: fragments loaded straight from global memory cost ~16 matrix-pipe cycles per returned
// VGPR row -- L1-hot or not, consumed or not -- against ~3-6 for LDS returns; see DESIGN.md "what the probes showed".)
//   X (pixels)  : halo tile of one 16-channel chunk, 64-B pixel records, 16-B slot s of pixel p stored at slot s ^ ((p>>2)&3)
//                 so a ds_read_b128 of 16 consecutive pixels is bank-conflict free.  Two buffers: the next chunk is fetched
//                 global->registers one 16-B piece per thread per tap and written to the other buffer.
//   W (weights) : a 3-slot ring of per-tap stages ([j][nr][lane][16 B], exactly the packed global image, so staging is a
//                 linear copy and fragment reads are conflict free).  Stage s+2 is fetched during tap s and landed at its end,
//                 so after the per-tap barrier the fragments of tap s+1 are already in LDS and already requested.
//   The staging loads are issued before a tap's MFMAs and consumed after them; nothing else writes their registers, so no
//   s_waitcnt separates them from the MFMAs that hide their latency.
// Pre-activation (AttResUNet.py:54-55): lrelu(x*mul+add) is applied to the pixel pieces on their way into LDS, out-of-image
// halo pixels are zeroed AFTER it -- the conv's zero padding applies to the activated tensor ("pad after activation").
//
// Epilogue: weights are the MFMA row operand, so a lane ends up with 4 CONSECUTIVE output channels of one pixel per
// accumulator quad -> residual loads and stores are 16-B accesses:
//   raw = acc + bias (+ residual)            -> y_raw
//   act = lrelu(raw * mul[n,c] + add[n,c])   -> y_act   (stored instead of raw when only the activated tensor is consumed)
//
#include "common.h"
#include "../../include/virnet_hip.h"
#include <cstdlib>

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
  const float* in_mul;
  const float* in_add;
  const float* mask;
  float* y_raw;
  float* y_act;
  int N, H, W, Cin;        // input
  int OH, OW;              // GEMM pixel grid (conv output; for CONVT the INPUT grid)
  int NP;                  // padded GEMM-N (output channel) extent
  int nrep_p;              // 32-channel slabs per channel block in the PACKED weights (a multiple of the kernel's NREP)
  int cout;                // real channels of the stored tensor
  int ntx, nty, ntiles, tiles_per_xcd;
  int epi, nchw_op, crop_h, crop_w, res_sf, in_act;
  float in_slope, mask_slope, slope, clamp_lo, clamp_hi;
};

__device__ __forceinline__ int swz(int p) { return (p >> 2) & 3; }
// LeakyReLU for slopes in [0, 1] (the reference uses 0.2 / 0.25): max(u, s*u) -- one multiply + one max, no compare/select.
__device__ __forceinline__ f32x4 lrelu4(f32x4 u, float s) {
  const f32x4 t = u * s;
  return f32x4{fmaxf(u.x, t.x), fmaxf(u.y, t.y), fmaxf(u.z, t.z), fmaxf(u.w, t.w)};
}

// NW = waves per workgroup (4, or 8 for the stride-2 tile whose 141 KB of pixel buffers leave room for only one workgroup per CU:
// eight waves keep two per SIMD).
template <int KS, int STRIDE, int MREP, int NREP, int NW>
__global__ __launch_bounds__(64 * NW, ((NW == 8) ? 2 : (NREP >= 7 || (NREP >= 5 && STRIDE == 2)) ? 1 : 2)) void conv_mfma_kernel(const KArgs a) {
  constexpr int NT = 64 * NW;
  constexpr int TH = NW * MREP;
  constexpr int PAD = KS / 2;
  constexpr int IH = (TH - 1) * STRIDE + KS;
  constexpr int IW = 31 * STRIDE + KS;
  constexpr int NPIX = IH * IW;
  constexpr int NPIECE = NPIX * 4;                 // 16-B pieces of one input chunk
  constexpr int NTAPS = KS * KS;
  constexpr int PPT = (NPIECE + NT - 1) / NT;        // input pieces per thread per chunk
  constexpr int LPT = (PPT + NTAPS - 1) / NTAPS;   // ... issued per tap
  constexpr int NB = 32 * NREP;
  constexpr int IN_BYTES = NPIX * 64;
  constexpr int WSTAGE = NREP * 2048;              // bytes of one tap's weight fragments: [j][nr][lane][16 B]
  constexpr int WPIECE = NREP * 128;               // ... in 16-B pieces
  constexpr int WPT = (WPIECE + NT - 1) / NT;
  constexpr int RING = 3;                          // weight ring slots (NTAPS % RING == 0 or NTAPS == 1)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const x_lds = smem;                        // [2][IN_BYTES]   pixel tile of chunk c / c+1
  char* const w_lds = smem + 2 * IN_BYTES;         // [RING][WSTAGE]  weight fragments of taps s, s+1, s+2

  // ---- workgroup -> (tile, channel block).  Block b runs on XCD b%8 (observed, speed only): give every XCD a contiguous
  // range of tiles so halo rows and the weight stream stay in that XCD's L2, and keep the channel blocks of one tile
  // adjacent in time so the input tile is fetched from HBM once.
  const int ncb = a.NP / NB;
  const int xcd = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  // (integer division runs on the VALU: pin the wave-uniform results back into SGPRs so every derived pointer stays scalar)
  const int cb = __builtin_amdgcn_readfirstlane(q % ncb);
  const int tile = __builtin_amdgcn_readfirstlane(xcd * a.tiles_per_xcd + q / ncb);
  if (q / ncb >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const int tx = __builtin_amdgcn_readfirstlane(tile % a.ntx);
  const int ty = __builtin_amdgcn_readfirstlane((tile / a.ntx) % a.nty);
  const int img = __builtin_amdgcn_readfirstlane(tile / (a.ntx * a.nty));
  const int oy0 = ty * TH, ox0 = tx * 32;
  const int iy0 = oy0 * STRIDE - PAD, ix0 = ox0 * STRIDE - PAD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lhi = lane >> 5;

  const int nchunks = a.Cin >> 4;
  const int nstages = nchunks * NTAPS;
  const float* const ximg = a.x + (size_t)img * a.H * a.W * a.Cin;
  // The packed weights hold nrep_p slabs per channel block; this instantiation owns NREP of them (NREP divides nrep_p), so a
  // small grid can be cut into more, shorter workgroups without repacking.
  const int slab0 = (cb * NREP) % a.nrep_p;
  const int wstage_p = a.nrep_p * 512;             // floats of one packed stage
  const float* const wcb = a.wp + (size_t)((cb * NREP) / a.nrep_p) * nstages * wstage_p;

  // Input piece k of this thread: q = k*256+tid -> (pixel p, 16-B slot s).  The global load is ALWAYS issued, from an address
  // clamped into the image; the zero fill of out-of-image halo pixels is applied when the registers are written to LDS, so
  // nothing but the load writes its destination registers and no wait is needed before the MFMAs that hide its latency.
  auto in_addr = [&](int k, int chunk, int& off, int& dst, bool& inb) {
    const int qq = k * NT + tid;
    const bool has = (k < PPT) && (qq < NPIECE);
    const int qc = has ? qq : 0;
    const int p = qc >> 2, s = qc & 3;
    const int iy = p / IW, ix = p - iy * IW;
    const int gy = iy0 + iy, gx = ix0 + ix;
    inb = has && (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
    const int gyc = min(max(gy, 0), a.H - 1), gxc = min(max(gx, 0), a.W - 1);
    off = (gyc * a.W + gxc) * a.Cin + chunk * 16 + s * 4;
    dst = has ? p * 64 + ((s ^ swz(p)) << 4) : -1;
  };
  // Pre-activation on the way into LDS (AttResUNet.py:54-55): v -> lrelu(v*mul+add), then zero for out-of-image pixels, so
  // the conv's zero padding applies to the ACTIVATED tensor.  A thread's pieces always sit in slot tid&3 of their pixel, so
  // its 4 channels of chunk c are c*16 + 4*(tid&3) + (0..3): one scale/shift quad per chunk.
  const bool in_sft = a.in_mul != nullptr;
  const float* const imul = in_sft ? a.in_mul + (size_t)img * a.Cin + 4 * (tid & 3) : nullptr;
  const float* const iadd = in_sft ? a.in_add + (size_t)img * a.Cin + 4 * (tid & 3) : nullptr;
  // Branch-free for every mode: without SFT m4 = 1, a4 = 0 (v*1+0 == v exactly); without activation the slope is 1
  // (max(v, v) == v).  ~12 VALU ops per staged piece in the tail of a tap.
  const float in_slope_eff = a.in_act ? a.in_slope : 1.f;
  auto stage_x = [&](f32x4 v, bool inb, const f32x4& m4, const f32x4& a4) -> f32x4 {
    v = lrelu4(v * m4 + a4, in_slope_eff);
    return inb ? v : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // weight stage `stage` -> registers (linear 16-B pieces; the packed global image IS the LDS image)
  auto load_w = [&](int stage, f32x4 (&r)[WPT]) {
    const float* const p = wcb + (size_t)stage * wstage_p;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int qq = min(i * NT + tid, WPIECE - 1);          // piece of THIS kernel's stage image [j][NREP][lane]
      const int j = qq / (NREP * 64), r64 = qq - j * (NREP * 64);
      r[i] = *reinterpret_cast<const f32x4*>(p + ((j * a.nrep_p + slab0) * 64 + r64) * 4);
    }
  };
  auto store_w = [&](int slot, const f32x4 (&r)[WPT]) {
#pragma unroll
    for (int i = 0; i < WPT; ++i)
      if (i * NT + tid < WPIECE) *reinterpret_cast<f32x4*>(w_lds + slot * WSTAGE + (i * NT + tid) * 16) = r[i];
  };
  // fragments of half-step (tap, j): weights of ring slot `slot`, pixels of tile buffer `buf`
  auto read_w = [&](int slot, int j, f32x4 (&w)[NREP]) {
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr)
      w[nr] = *reinterpret_cast<const f32x4*>(w_lds + slot * WSTAGE + (j * NREP + nr) * 1024 + lane * 16);
  };
  auto read_x = [&](const char* buf, int tap, int j, f32x4 (&x)[MREP]) {
    const int dy = (KS == 3) ? tap / 3 : 0, dx = (KS == 3) ? tap % 3 : 0;
#pragma unroll
    for (int mr = 0; mr < MREP; ++mr) {
      const int p = ((wave * MREP + mr) * STRIDE + dy) * IW + l31 * STRIDE + dx;
      x[mr] = *reinterpret_cast<const f32x4*>(buf + p * 64 + (((2 * j + lhi) ^ swz(p)) << 4));
    }
  };

  // ---- prologue: chunk 0 of the pixel tile and weight stages 0, 1 -> LDS -------------------------------------------------
  {
    f32x4 w0[WPT], w1[WPT];
    load_w(0, w0);
    load_w(min(1, nstages - 1), w1);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      int off, dst; bool inb;
      in_addr(k, 0, off, dst, inb);
      const f32x4 v = *reinterpret_cast<const f32x4*>(ximg + off);
      const f32x4 m4 = in_sft ? *reinterpret_cast<const f32x4*>(imul) : f32x4{1.f, 1.f, 1.f, 1.f};
      const f32x4 a4 = in_sft ? *reinterpret_cast<const f32x4*>(iadd) : f32x4{0.f, 0.f, 0.f, 0.f};
      if (dst >= 0) *reinterpret_cast<f32x4*>(x_lds + dst) = stage_x(v, inb, m4, a4);
    }
    store_w(0, w0);
    store_w(1 % RING, w1);
  }
  __syncthreads();

  f32x16 acc[MREP][NREP];
#pragma unroll
  for (int mr = 0; mr < MREP; ++mr)
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mr][nr][r] = 0.f;

  f32x4 wcur[NREP], xcur[MREP];
  read_w(0, 0, wcur);
  read_x(x_lds, 0, 0, xcur);

  int stage = 0;
  for (int c = 0; c < nchunks; ++c) {
    const char* const in_cur = x_lds + (c & 1) * IN_BYTES;
    char* const in_nxt = x_lds + ((c + 1) & 1) * IN_BYTES;
    const bool more_chunks = (c + 1 < nchunks);
    f32x4 m4 = f32x4{1.f, 1.f, 1.f, 1.f}, a4 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (in_sft && more_chunks) {
      m4 = *reinterpret_cast<const f32x4*>(imul + (c + 1) * 16);
      a4 = *reinterpret_cast<const f32x4*>(iadd + (c + 1) * 16);
    }
#pragma unroll
    for (int t = 0; t < NTAPS; ++t, ++stage) {
      // ring slots: NTAPS == 9 is a multiple of RING, so the slot of tap t is a compile-time t % RING
      const int slot = (NTAPS % RING == 0) ? t % RING : stage % RING;
      const int slot1 = (slot + 1) % RING, slot2 = (slot + 2) % RING;
      const bool last_tap = (t == NTAPS - 1);
      // ---- staging requests (global -> regs): weights of stage s+2, one slice of the next chunk's pixel tile.  Always
      // issued (the final ones re-read valid addresses) so the memory counters see straight-line code.
      f32x4 wreg[WPT];
      load_w(min(stage + 2, nstages - 1), wreg);
      f32x4 ireg[LPT];
      int idst[LPT];
      bool iinb[LPT];
#pragma unroll
      for (int i = 0; i < LPT; ++i) {
        int off;
        in_addr(t * LPT + i, more_chunks ? c + 1 : c, off, idst[i], iinb[i]);
        ireg[i] = *reinterpret_cast<const f32x4*>(ximg + off);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // ---- fragment requests (LDS -> regs) for the NEXT half-step, issued before this half-step's MFMAs
        f32x4 wnxt[NREP], xnxt[MREP];
        if (j == 0) {
          read_w(slot, 1, wnxt);
          read_x(in_cur, t, 1, xnxt);
        } else {
          read_w(slot1, 0, wnxt);                               // stage s+1 was landed one tap ago
          if (!last_tap) read_x(in_cur, t + 1, 0, xnxt);        // (next chunk's pixels become visible after the barrier)
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int mr = 0; mr < MREP; ++mr)
#pragma unroll
            for (int nr = 0; nr < NREP; ++nr)
              acc[mr][nr] = __builtin_amdgcn_mfma_f32_32x32x2f32(wcur[nr][r], xcur[mr][r], acc[mr][nr], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nr = 0; nr < NREP; ++nr) wcur[nr] = wnxt[nr];
        if (!(j == 1 && last_tap)) {
#pragma unroll
          for (int mr = 0; mr < MREP; ++mr) xcur[mr] = xnxt[mr];
        }
      }
      // ---- land the staged data: weights of stage s+2 into the slot last read during tap s-1, pixels into the other tile
      store_w(slot2, wreg);
      if (more_chunks) {
#pragma unroll
        for (int i = 0; i < LPT; ++i)
          if (idst[i] >= 0) *reinterpret_cast<f32x4*>(in_nxt + idst[i]) = stage_x(ireg[i], iinb[i], m4, a4);
      }
      __syncthreads();
      if (last_tap && more_chunks) read_x(in_nxt, 0, 0, xcur);
    }
  }

  // ---- epilogue: lane = pixel (ox0 + l31), accumulator quad g = 4 consecutive channels 8g + 4*lhi + (0..3) ----------
  // Addressing: a wave-uniform 64-bit image base (SGPRs) + ONE 32-bit offset per row block + compile-time immediates.
  const int nbase = cb * NB;
  const int px = ox0 + l31;
  const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
  if (KS == 1 || a.epi != VIRNET_EPI_NCHW) {
    // NHWC store, or (KS == 1) the transposed conv's depth-to-space store: GEMM row n' = ab*cout + co and input pixel
    // (iy,ix) -> output pixel (2*iy+a, 2*ix+b), channel co.
    const bool convt = (KS == 1) && a.epi == VIRNET_EPI_CONVT;
    const int C = a.cout;
    const int OWs = convt ? 2 * a.OW : a.OW;
    const size_t img_off = (size_t)img * (convt ? 4 : 1) * a.OH * a.OW * C;
    const float* const rimg = a.res ? a.res + img_off : nullptr;
    const float* const mimg = a.mask ? a.mask + img_off : nullptr;
    float* const yraw = a.y_raw ? a.y_raw + img_off : nullptr;
    float* const yact = a.y_act ? a.y_act + img_off : nullptr;
    const float* const mulp = a.mul ? a.mul + (size_t)img * C : nullptr;
    const float* const addp = a.add ? a.add + (size_t)img * C : nullptr;
    unsigned eo[MREP];
    bool ok[MREP];
#pragma unroll
    for (int mr = 0; mr < MREP; ++mr) {
      const int oy = oy0 + wave * MREP + mr;
      ok[mr] = oy < a.OH && px < a.OW;
      const int oyc = min(oy, a.OH - 1), pxc = min(px, a.OW - 1);      // clamped: loads are always issued, stores predicated
      eo[mr] = convt ? (unsigned)((2 * oyc) * OWs + 2 * pxc) * (unsigned)C : (unsigned)(oyc * OWs + pxc) * (unsigned)C;
    }
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      // per 32-channel slab: issue the 4 bias quads and all 4*MREP residual quads back to back (one memory round trip per
      // slab instead of one per quad), then combine and store.
      f32x4 bias[4], rv[4][MREP], mv[4][MREP];
      unsigned off[4][MREP];
      int cog[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int np = nbase + nr * 32 + 8 * g + 4 * lhi;      // GEMM row of this quad
        int co = np;
        unsigned shift = 0;
        if (convt) {
          const int ab = np / C;
          co = np - ab * C;
          shift = (unsigned)((ab >> 1) * OWs + (ab & 1)) * (unsigned)C;
        }
        cog[g] = co;
        bias[g] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + co) : zero4;
#pragma unroll
        for (int mr = 0; mr < MREP; ++mr) {
          off[g][mr] = eo[mr] + shift + (unsigned)co;
          rv[g][mr] = rimg ? *reinterpret_cast<const f32x4*>(rimg + off[g][mr]) : zero4;
          if (mimg) mv[g][mr] = *reinterpret_cast<const f32x4*>(mimg + off[g][mr]);
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 mul = f32x4{1.f, 1.f, 1.f, 1.f}, add = zero4;
        if (mulp) {
          mul = *reinterpret_cast<const f32x4*>(mulp + cog[g]);
          add = *reinterpret_cast<const f32x4*>(addp + cog[g]);
        }
#pragma unroll
        for (int mr = 0; mr < MREP; ++mr) {
          f32x4 v = f32x4{acc[mr][nr][4 * g], acc[mr][nr][4 * g + 1], acc[mr][nr][4 * g + 2], acc[mr][nr][4 * g + 3]} + bias[g];
          if (mimg) {                                   // backward of a LeakyReLU: multiply by its derivative at the saved tensor
            const f32x4 m = mv[g][mr];
            v = f32x4{m.x > 0.f ? v.x : v.x * a.mask_slope, m.y > 0.f ? v.y : v.y * a.mask_slope,
                      m.z > 0.f ? v.z : v.z * a.mask_slope, m.w > 0.f ? v.w : v.w * a.mask_slope};
          }
          v += rv[g][mr];
          if (ok[mr]) {
            if (yraw) *reinterpret_cast<f32x4*>(yraw + off[g][mr]) = v;
            if (yact) *reinterpret_cast<f32x4*>(yact + off[g][mr]) = lrelu4(v * mul + add, a.slope);
          }
        }
      }
    }
  } else if (NREP == 1) {
    // VIRNET_EPI_NCHW: few real channels (<= 32): planar store with crop; 32 consecutive x per channel = 128-B runs
    const size_t plane = (size_t)a.crop_h * a.crop_w;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
      if (n >= a.cout) continue;
      const float bias = a.bias ? a.bias[n] : 0.f;
      const size_t base = ((size_t)img * a.cout + n) * plane;
#pragma unroll
      for (int mr = 0; mr < MREP; ++mr) {
        const int oy = oy0 + wave * MREP + mr;
        if (oy < a.crop_h && px < a.crop_w) {
          const size_t o = base + (size_t)oy * a.crop_w + px;
          float v = acc[mr][0][r] + bias;
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
  }
}

template <int KS, int STRIDE, int MREP, int NREP, int NW = 4>
int launch(const KArgs& ka, hipStream_t st) {
  constexpr int TH = NW * MREP;
  constexpr int IH = (TH - 1) * STRIDE + KS, IW = 31 * STRIDE + KS;
  constexpr int LDS = 2 * IH * IW * 64 + 3 * NREP * 2048;
  static unsigned long long attr_done = 0;     // one bit per device
  auto kern = conv_mfma_kernel<KS, STRIDE, MREP, NREP, NW>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_mfma): %s", hipGetErrorString(e));
  }
  KArgs k = ka;
  k.nty = (k.OH + TH - 1) / TH;
  k.ntx = (k.OW + 31) / 32;
  k.ntiles = k.N * k.nty * k.ntx;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  const int ncb = k.NP / (32 * NREP);
  const unsigned grid = (unsigned)(8 * k.tiles_per_xcd * ncb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), LDS, st, k);
  return virnet::check_launch("conv_mfma launch");
}

int pick_nrep(int nblocks32) {
  static const int pref[] = {3, 4, 5, 2, 7, 1};
  for (int d : pref)
    if (nblocks32 % d == 0) return d;
  return 1;
}

// Tile choice.  Large grids: 8-row tiles x all packed slabs (fewest operand fetches per MFMA).  Smaller grids trade that for
// more, shorter workgroups: 4-row tiles, then single-slab workgroups (NREP=1 reads its slab out of the wider packing), so deep
// U-Net levels and single images still cover the 256 CUs and no workgroup's serial K loop dominates the launch.
struct TileSel { int mrep, nrep, nw; };
TileSel pick_tile(const virnet_conv_desc* d) {
  static const int forced = [] { const char* e = getenv("VIRNET_FORCE_MREP"); return e ? atoi(e) : 0; }();    // tuning knobs
  static const int forced_n = [] { const char* e = getenv("VIRNET_FORCE_NREP"); return e ? atoi(e) : 0; }();
  const long tiles8 = (long)d->n * ((d->h / d->stride + 7) / 8) * ((d->w / d->stride + 31) / 32);
  const long tiles4 = (long)d->n * ((d->h / d->stride + 3) / 4) * ((d->w / d->stride + 31) / 32);
  const long cbs = d->n_pad / (32 * d->nrep);
  TileSel t{1, d->nrep, 4};
  if (d->stride == 1 && d->nrep <= 3 && tiles8 * cbs >= 2048) t.mrep = 2;      // MREP=2 with >= 4 slabs would spill
  else if (tiles4 * cbs < 1024 && d->nrep > 1) t.nrep = 1;
  if (d->stride == 2 && d->nrep <= 4 && tiles8 * cbs >= 512) t.nw = 8;       // one workgroup per CU either way: give it 8 waves
  static const int forced_w = [] { const char* e = getenv("VIRNET_FORCE_NW"); return e ? atoi(e) : 0; }();
  if (forced_w == 4) t.nw = 4;
  if ((forced == 1 || forced == 2) && d->stride == 1 && d->nrep <= 3) t.mrep = forced;
  if (forced_n == 1 || forced_n == d->nrep) t.nrep = forced_n;
  if (t.nrep != d->nrep) t.nw = 4;
  return t;
}

}  // namespace

extern "C" int virnet_conv_mfma_variant(const virnet_conv_desc* d, int out[4]) {
  VIRNET_REQUIRE(d && out, "virnet_conv_mfma_variant: NULL pointer");
  const TileSel t = pick_tile(d);
  out[0] = d->ks; out[1] = d->stride; out[2] = t.mrep; out[3] = t.nrep;
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
  VIRNET_REQUIRE((d->in_mul == nullptr) == (d->in_add == nullptr), "virnet_conv_mfma: in_mul and in_add must be given together");
  VIRNET_REQUIRE(d->in_act || !d->in_mul, "virnet_conv_mfma: in_mul/in_add without in_act");
  VIRNET_REQUIRE(!d->in_act || (d->in_slope >= 0.f && d->in_slope <= 1.f), "virnet_conv_mfma: in_slope=%g outside [0,1]", d->in_slope);
  VIRNET_REQUIRE(!d->y_act || (d->slope >= 0.f && d->slope <= 1.f), "virnet_conv_mfma: slope=%g outside [0,1]", d->slope);
  KArgs k{};
  k.x = d->x; k.wp = d->wpack; k.bias = d->bias; k.res = d->res; k.mul = d->mul; k.add = d->add;
  k.in_mul = d->in_mul; k.in_add = d->in_add; k.in_act = d->in_act; k.in_slope = d->in_slope;
  k.mask = d->mask; k.mask_slope = d->mask_slope;
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
      VIRNET_REQUIRE(d->ks == 3 || d->ks == 1, "virnet_conv_mfma: NHWC store needs ks 1 or 3 (ks=%d)", d->ks);
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
  const TileSel ts = pick_tile(d);
  const int mrep = ts.mrep;
  k.nrep_p = d->nrep;
#define VIRNET_CASE(KS_, S_, M_, N_) \
  if (d->ks == KS_ && d->stride == S_ && mrep == M_ && ts.nrep == N_) return launch<KS_, S_, M_, N_>(k, st)
  VIRNET_CASE(3, 1, 1, 1); VIRNET_CASE(3, 1, 2, 1);
  VIRNET_CASE(3, 1, 1, 2); VIRNET_CASE(3, 1, 2, 2);
  VIRNET_CASE(3, 1, 1, 3); VIRNET_CASE(3, 1, 2, 3);
  VIRNET_CASE(3, 1, 1, 4); VIRNET_CASE(3, 1, 1, 5); VIRNET_CASE(3, 1, 1, 7);
  if (d->ks == 3 && d->stride == 2 && ts.nw == 8) {
    if (ts.nrep == 2) return launch<3, 2, 1, 2, 8>(k, st);
    if (ts.nrep == 3) return launch<3, 2, 1, 3, 8>(k, st);
    if (ts.nrep == 4) return launch<3, 2, 1, 4, 8>(k, st);
  }
  VIRNET_CASE(3, 2, 1, 1); VIRNET_CASE(3, 2, 1, 2); VIRNET_CASE(3, 2, 1, 3); VIRNET_CASE(3, 2, 1, 4); VIRNET_CASE(3, 2, 1, 5);
  VIRNET_CASE(3, 2, 1, 7);
  VIRNET_CASE(1, 1, 1, 1); VIRNET_CASE(1, 1, 2, 1); VIRNET_CASE(1, 1, 1, 2); VIRNET_CASE(1, 1, 2, 2); VIRNET_CASE(1, 1, 1, 3);
  VIRNET_CASE(1, 1, 2, 3); VIRNET_CASE(1, 1, 1, 4); VIRNET_CASE(1, 1, 1, 5); VIRNET_CASE(1, 1, 1, 7);
#undef VIRNET_CASE
  return virnet::set_error("virnet_conv_mfma: no kernel for ks=%d stride=%d mrep=%d nrep=%d", d->ks, d->stride, mrep, ts.nrep);
}
