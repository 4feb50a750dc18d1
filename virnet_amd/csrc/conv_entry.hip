// conv_entry.hip -- the network ENTRY convolutions as their own kernel: AttResUNet.head (networks/AttResUNet.py:117,153-155: cat(image,
// conditioning) -> n_feat[0]) and DnCNN.conv1 (networks/DnCNN.py:22,38: image -> 64), 3x3 stride-1 on <= 8 input channels gathered straight
// from the NCHW image / per-image vector / per-pixel map (virnet_pack_desc: nearest up-sampling VIRNet.py:83,94, bottom / right reflect
// pad utils/util_net.py:20-25, sqrt of the variance map VIRNet.py:44), NHWC store.  gfx950, split-fp16 products as conv_f16.hip.
//
// Why not conv_f16's ENT form (round 4): that kernel walks 9 taps x one 16-channel chunk of which 3..7 channels are real (27 MFMA groups
// per block for 9..12 useful K rows each), gathers its pixel records in a latency-bound prologue, and ran at 2.8 TB/s of stores where a
// plain fill reaches 6.9 TB/s on this chip (profiles/r05_probes.md 6).  The layer has ~0.1 FLOP per output byte: it is a STORE kernel.
//   K = (dy, dx, channel): one MFMA k-step holds a whole kernel ROW -- slot j = dx * C + ch of the 16 (C <= 5) or 32 (C <= 8) slots per dy --
//       so a block costs 3 (6) split-fp16 product triples instead of 9 x 3, and the B fragment of a pixel is gathered from a PLANAR fp32
//       copy of the (tile + halo) x C input in LDS: eight ds_read_b32 at per-lane addresses fixed for the whole kernel + immediates.
//   Workgroup = 4 waves = 8 x 32 output pixels x all output channels (NSLAB = cout / 32 <= 3); wave w owns rows 2w, 2w+1.
//   Epilogue = the point: every wave turns each of its rows around through a private LDS region ([pixel][cout] + 16 B pad, conflict-free
//       both ways) and stores it LANE-LINEAR: one global_store_dwordx4 = 1 KB of contiguous NHWC bytes (a tile row is 32 x cout x 4 B
//       contiguous), 12 instructions per row at 96 channels.
#include "conv_f16_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace {
using namespace virnet;

constexpr int EN_TH = 8, EN_TW = 32;
constexpr int EN_ROW = 36;                        // floats per LDS row of one plane (34 used)
constexpr int EN_PLANE = (EN_TH + 2) * EN_ROW;    // floats per channel plane
constexpr int EN_ZERO = 4 * EN_ROW;               // floats of the always-zero area the empty K slots read from (reached with the (dy + row) immediates)

// Where record channel c comes from, resolved on the host once per launch so that the kernel's gather is branch-free:
//   value(c, image n, pixel) = base[c][n * istride[c] + xpix * mulx[c] + mpix * mulm[c]]      (xpix / mpix: the pixel's index at the image's /
//   the map's own resolution; a per-image vector entry has both factors 0), sqrt'ed when sq[c]; channels beyond the record read base[0].
struct EntrySrc {
  const float* base[8];
  unsigned istride[8], mulx[8], mulm[8], sq[8];
};

template <int NSLAB, int NT>      // NT = k-steps per kernel row: 1 (<= 5 channels) or 2 (<= 8)
__global__ __launch_bounds__(256, (NSLAB == 3 && NT == 2) ? 1 : 2) void conv_entry_kernel(const FArgs a, const EntrySrc src, const int nchan, const int wgs_per_xcd) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = NSLAB * 128 + 16;                             // bytes of one pixel's turn-around record
  constexpr int WBYTES = 3 * NT * NSLAB * 2048;                    // A fragments: [dy][t][slab][hi|lo][64 lanes][16 B]
  constexpr int TBYTES = EN_TW * RB;                               // one wave's turn-around region (one tile row)
  constexpr int NB = NSLAB * 32;
  constexpr int MAXC = 8;
  char* const w_lds = smem;
  char* const t_lds = smem + WBYTES;                               // 4 regions
  float* const sb_lds = reinterpret_cast<float*>(smem + WBYTES + 4 * TBYTES);      // [inverse scale | bias] of the NSLAB*32 channels
  float* const z_lds = sb_lds + 2 * NB;                            // zero area
  float* const x_lds = z_lds + EN_ZERO;                            // [nchan][10][EN_ROW] (ONE buffer: with a second one the workgroup is 82 KB and alone on its CU)

  const virnet_pack_desc& e = a.ent;
  // PERSISTENT workgroups: two per CU, each walks every wgs_per_xcd-th tile of its XCD's contiguous tile range.  The layer moves 4 B in
  // per 128 B out and multiplies for ~1.5 us per tile: what a tile costs is latency, so the next tile's pixels are requested before this
  // tile's MFMAs and landed in the other LDS buffer before this tile's stores; weights, tables and the gather addresses are set up once.
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int t_lo = xcd * a.tiles_per_xcd, t_hi = min(a.ntiles, t_lo + a.tiles_per_xcd);
  int tile = t_lo + slot;
  if (slot >= wgs_per_xcd || tile >= t_hi) return;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;

  // ---- weights, scale / bias table, zero area: once per workgroup
  for (int i = tid * 16; i < WBYTES; i += 256 * 16) *reinterpret_cast<f32x4*>(w_lds + i) = *reinterpret_cast<const f32x4*>(a.wimg + i);
  // (the image's trailer names the channel count it was packed for -- slot j = dx*cin + ch is baked into it: a caller of the C ABI that
  // hands over an image packed for another count gets NaN, not a plausible picture; ADVICE r05.  The host cannot look: device memory.)
  const bool img_ok = *reinterpret_cast<const int*>(a.wimg + WBYTES) == nchan && *reinterpret_cast<const int*>(a.wimg + WBYTES + 4) == NT;
  if (tid < NB) sb_lds[tid] = img_ok ? a.inv_scale[tid] : __builtin_nanf("");
  else if (tid < 2 * NB) sb_lds[tid] = a.bias ? a.bias[tid - NB] : 0.f;
  if (tid < EN_ZERO) z_lds[tid] = 0.f;

  // ---- staging slots of this thread: halo pixels p = tid and tid + 256 of the (8 + 2) x (32 + 2) input tile (slots beyond its 340 pixels
  // are "dead" and land in a dummy word).  Everything below is branch-free: coordinates are clamped for the address, dead pixels are
  // zeroed by a select behind the load, unused record channels read base[0] and are never gathered.
  const int HU = e.h * e.sf, WU = e.w * e.sf;
  int shy[2], shx[2];
  bool slot_ok[2];
  unsigned ldst[2];                                                // float index of the slot in a pixel buffer (dead slots: the dummy word)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int p = tid + 256 * k;
    slot_ok[k] = p < (EN_TH + 2) * 34;
    shy[k] = p / 34;
    shx[k] = p - shy[k] * 34;
    ldst[k] = slot_ok[k] ? (unsigned)(shy[k] * EN_ROW + shx[k]) : (unsigned)(EN_TH + 2) * EN_ROW - 1u;      // (last pad column of the last row: never read)
  }
  float vin[2][MAXC];
  unsigned vdead = 0;
  auto request = [&](int tl) {
    const int im = fast_div(tl, a.mg_tpi);
    const int trem = tl - im * (a.ntx * a.nty);
    const int ty = fast_div(trem, a.mg_ntx), tx = trem - ty * a.ntx;
    vdead = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int y = ty * EN_TH - 1 + shy[k], x = tx * EN_TW - 1 + shx[k];
      const bool in = slot_ok[k] && (unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W;
      const int yc = min(max(y, 0), a.H - 1), xc = min(max(x, 0), a.W - 1);
      const int ry = yc < HU ? yc : 2 * HU - 2 - yc, rx = xc < WU ? xc : 2 * WU - 2 - xc;      // bottom / right reflect pad
#ifdef EN_PROBE_NOLOAD
      const bool dead = a.slope != 12345.f;
#else
      const bool dead = !in || (e.zero_pad && (y >= HU || x >= WU));
#endif
      vdead |= (dead ? 1u : 0u) << k;
      const unsigned xpix = (unsigned)((e.sf == 1 ? ry : ry / e.sf) * e.w + (e.sf == 1 ? rx : rx / e.sf));
      const unsigned mpix = (unsigned)((e.msf == 1 ? ry : ry / e.msf) * e.mw + (e.msf == 1 ? rx : rx / e.msf));
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        if (NT == 1 && c >= 5) { vin[k][c] = 0.f; continue; }
#ifdef EN_PROBE_NOLOAD
        vin[k][c] = dead ? 0.f : src.base[c][(size_t)im * src.istride[c] + xpix * src.mulx[c] + mpix * src.mulm[c]];
#else
        vin[k][c] = src.base[c][(size_t)im * src.istride[c] + xpix * src.mulx[c] + mpix * src.mulm[c]];
#endif
      }
    }
  };
  float amax = 0.f;
  auto land = [&]() {                                       // the requested values -> planar fp32 tile `buf` (dead pixels zero, sqrt of the map channels)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float* const dst = x_lds + ldst[k];
      const bool dead = (vdead >> k) & 1u;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        if (NT == 1 && c >= 5) continue;
        float v = dead ? 0.f : vin[k][c];
        if (src.sq[c]) v = sqrtf(v);                                  // (uniform per channel)
        amax = fmaxf(amax, fabsf(v));
        if (c < nchan) dst[c * EN_PLANE] = v;
      }
    }
  };

  // ---- per-lane gather addresses of the K slots (buffer 0): slot j = 16 t + 8 lhi + i  ->  (dx, ch) = (j / C, j % C), empty beyond 3 C
  unsigned gaddr[NT][8];
  const unsigned xbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(x_lds) + (unsigned)((2 * wave) * EN_ROW + l31) * 4u;
  const unsigned zbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)(z_lds);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = 16 * t + 8 * lhi + i;
      int dx = 0, ch = j;
      while (ch >= nchan && dx < 3) { ch -= nchan; ++dx; }
      gaddr[t][i] = dx < 3 ? xbase + (unsigned)(ch * EN_PLANE + dx) * 4u : zbase;
    }

  request(tile);
  land();
  __syncthreads();
  // Loads and stores share ONE in-order vmcnt on gfx950: a tile's pixels requested BEHIND the previous tile's stores cannot be waited for
  // without waiting for those stores' acknowledgements as well (an HBM write round trip under load).  So the request for tile i + 2 goes out
  // in FRONT of tile i's stores: when tile i + 1 has been multiplied, the wait for them leaves tile i's 24 stores per wave outstanding.
  int nxt = tile + wgs_per_xcd;
  if (nxt < t_hi) request(nxt);

#ifdef VIRNET_F16_TIMING
  long long tq_mma = 0, tq_land = 0, tq_epi = 0, tq_bar = 0, tq_mark = (long long)__builtin_amdgcn_s_memtime();
  const long long tq_start = tq_mark;
  int tq_tiles = 0;
#define EN_TADD(var) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); var += n_ - tq_mark; tq_mark = n_; } while (0)
#else
#define EN_TADD(var) do { } while (0)
#endif
  char* const tw = t_lds + wave * TBYTES;
  const float slope = a.y_act ? a.slope : 1.f;
  const bool act = a.y_act != nullptr && a.slope != 1.f;
  constexpr int UPP = NSLAB * 8;                                   // 16-byte units per pixel
  constexpr int NU = EN_TW * UPP / 64;                             // store instructions per row
  for (;;) {
    const bool more = nxt < t_hi;                                  // (its pixels are in flight)

    const int img = fast_div(tile, a.mg_tpi);
    const int trem = tile - img * (a.ntx * a.nty);
    const int ty = fast_div(trem, a.mg_ntx), tx = trem - ty * a.ntx;
    const int oy0 = ty * EN_TH, ox0 = tx * EN_TW;
    int wl = lane * 16;                                            // this lane's offset into an A fragment
    if constexpr (NT == 2) {                                       // (six k-steps x NSLAB fragments do not fit the registers as loop invariants: keep
#if defined(__HIP_DEVICE_COMPILE__)                                //  the optimizer from hoisting their LDS reads out of the tile loop)
      asm volatile("" : "+v"(wl));
#endif
    }

    f32x16 acc[2][NSLAB];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (NT == 2) {
#pragma unroll
      for (int mr = 0; mr < 2; ++mr)
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) acc[mr][s] = zero16;
    }
    auto kernel_row = [&](int dy) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        h8 ah[NSLAB], al[NSLAB];
#pragma unroll
        for (int s = 0; s < NSLAB; ++s) {
          ah[s] = *reinterpret_cast<const h8*>(w_lds + (((dy * NT + t) * NSLAB + s) * 2 + 0) * 1024 + wl);
          al[s] = *reinterpret_cast<const h8*>(w_lds + (((dy * NT + t) * NSLAB + s) * 2 + 1) * 1024 + wl);
        }
#pragma unroll
        for (int mr = 0; mr < 2; ++mr) {
          f32x4 v0, v1;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const unsigned a0 = gaddr[t][i], a1 = gaddr[t][4 + i];
            v0[i] = *reinterpret_cast<const float __attribute__((address_space(3)))*>((size_t)(a0 + (unsigned)((dy + mr) * EN_ROW * 4)));
            v1[i] = *reinterpret_cast<const float __attribute__((address_space(3)))*>((size_t)(a1 + (unsigned)((dy + mr) * EN_ROW * 4)));
          }
          h8 bh, bl;
          split8(v0, v1, bh, bl);
#pragma unroll
          for (int s = 0; s < NSLAB; ++s) {
            // (NT == 1: the rows are unrolled, the very first product of an accumulator takes the literal zero as its C operand -- no clearing pass)
            acc[mr][s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[s], bh, (NT == 1 && dy == 0 && t == 0) ? zero16 : acc[mr][s], 0, 0, 0);
            acc[mr][s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bl, acc[mr][s], 0, 0, 0);
            acc[mr][s] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[s], bh, acc[mr][s], 0, 0, 0);
          }
        }
      }
    };
    if constexpr (NT == 1) {
      kernel_row(0); kernel_row(1); kernel_row(2);
    } else {                                                       // (two k-steps per row: rolled, or the fragments of all six steps crowd the registers)
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy) kernel_row(dy);
    }

    EN_TADD(tq_mma);
    const int nxt2 = nxt + wgs_per_xcd;
    if (more) {                                                    // the next tile's pixels are on the chip before this tile's stores leave
      __syncthreads();                                             // (every wave has gathered its last fragment of this tile)
      land();
      if (nxt2 < t_hi) request(nxt2);
    }
    EN_TADD(tq_land);

    // ---- epilogue: inverse scale, bias, optional LeakyReLU; per row a wave-private LDS turn-around, then lane-linear 1-KB stores.
    // Accumulator register r of lane (l31, lhi) = channel 8 (r >> 2) + 4 lhi + (r & 3) of pixel l31 (conv_f16.hip).
    float* const y = (a.y_act ? a.y_act : a.y_raw) + (size_t)img * a.H * a.W * a.cout;
    const auto yrs = __builtin_amdgcn_make_buffer_rsrc(y, 0, a.H * a.W * a.cout * 4, 0x00020000);      // (units outside the image: out-of-range offset, no branch)
    auto epilogue = [&](auto actc) {
#pragma unroll
    for (int mr = 0; mr < 2; ++mr) {
      const int oy = oy0 + 2 * wave + mr;
#pragma unroll
      for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 i4 = *reinterpret_cast<const f32x4*>(sb_lds + s * 32 + 8 * g + 4 * lhi);
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(sb_lds + NB + s * 32 + 8 * g + 4 * lhi);
          f32x4 v = f32x4{acc[mr][s][4 * g], acc[mr][s][4 * g + 1], acc[mr][s][4 * g + 2], acc[mr][s][4 * g + 3]} * i4 + b4;
          if constexpr (decltype(actc)::value) v = lrelu4(v, slope);
          *reinterpret_cast<f32x4*>(tw + l31 * RB + s * 128 + g * 32 + lhi * 16) = v;
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (wave-private region: the wave's own in-order LDS traffic is the only ordering needed)
      {
        const unsigned rowoff = (unsigned)((oy * a.W + ox0) * a.cout) * 4u;
        const int valid = oy < a.H ? min(EN_TW, a.W - ox0) * UPP : 0;      // 16-byte units of this row that lie inside the image
#pragma unroll
        for (int k = 0; k < NU; ++k) {
          const int u = k * 64 + lane;
          const int px = u / UPP, wi = u - px * UPP;
          const f32x4 v = *reinterpret_cast<const f32x4*>(tw + px * RB + wi * 16);
#ifdef EN_PROBE_NOSTORE
          const bool keep = u < valid && a.slope == 12345.f;
#else
          const bool keep = u < valid;
#endif
          if (a.store_nt) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, keep ? rowoff + (unsigned)u * 16u : 0x80000000u, 0, 2);
          else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), yrs, keep ? rowoff + (unsigned)u * 16u : 0x80000000u, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // the reads have returned before the next row overwrites the region
    }
    };
    if (act) epilogue(std::true_type{}); else epilogue(std::false_type{});      // (uniform: the head stores raw values)
    EN_TADD(tq_epi);
#ifdef VIRNET_F16_TIMING
    ++tq_tiles;
#endif
    if (!more) break;
    tile = nxt;
    nxt = nxt2;
    __syncthreads();          // every wave has landed its share of the next tile
    EN_TADD(tq_bar);
  }
#ifdef VIRNET_F16_TIMING
  if (a.tlog && tid == 0) {
    long long* const o = a.tlog + (size_t)blockIdx.x * 8;
    o[0] = tq_start; o[1] = tq_mma; o[2] = tq_land; o[3] = tq_epi; o[4] = tq_bar; o[5] = tq_tiles; o[6] = (long long)__builtin_amdgcn_s_memtime();
  }
#endif
  range_report(a.range_flag, amax);
}

// Weight image of the entry kernel: rows = output channels, K slots of kernel row dy: j = dx * cin + ch (j < 3 cin), NT k-steps of 16 slots.
// [dy][t][slab][hi|lo][lane = row + 32 * (k >> 3)][k & 7] preceded by the n_pad inverse scales; one block per output channel.
__global__ void pack_entry_kernel(const float* __restrict__ w, int cout, int cin, int nt, int n_pad, float* __restrict__ inv_scale,
                                  char* __restrict__ img) {
  const int row = blockIdx.x;
  __shared__ float red[64];
  float m = 0.f;
  if (row < cout)
    for (int i = threadIdx.x; i < cin * 9; i += 64) m = fmaxf(m, fabsf(w[(size_t)row * cin * 9 + i]));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 32; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  m = red[0];
  int ex = 0;
  if (m > 0.f) { frexpf(m, &ex); ex = 14 - ex; }          // largest scaled magnitude in [8192, 16384)
  ex = max(-100, min(100, ex));
  if (threadIdx.x == 0) inv_scale[row] = ldexpf(1.f, -ex);
  const int slab = row >> 5, col = row & 31, nslab = n_pad >> 5;
  for (int i = threadIdx.x; i < 3 * nt * 16; i += 64) {
    const int dy = i / (nt * 16), r = i - dy * (nt * 16), t = r >> 4, k = r & 15;
    const int j = 16 * t + k, dx = j / cin, ch = j - dx * cin;
    float v = 0.f;
    if (row < cout && dx < 3) v = ldexpf(w[(((size_t)row * cin + ch) * 3 + dy) * 3 + dx], ex);
    const size_t base = ((((size_t)(dy * nt + t) * nslab + slab) * 2) * 1024) + (size_t)(col + 32 * (k >> 3)) * 16 + (k & 7) * 2;
    const _Float16 hi = (_Float16)v;
    *reinterpret_cast<_Float16*>(img + base) = hi;
    *reinterpret_cast<_Float16*>(img + base + 1024) = (_Float16)(v - (float)hi);
  }
  if (row == 0 && threadIdx.x == 0) {                      // trailer behind the image: what it was packed for (checked by the kernel)
    int* const tr = reinterpret_cast<int*>(img + (size_t)3 * nt * nslab * 2048);
    tr[0] = cin; tr[1] = nt; tr[2] = n_pad; tr[3] = 0x454e5452;      // 'ENTR'
  }
}

// what the launcher asks the runtime once per (device, dynamic LDS size), not per launch (the single-image path is host-bound)
struct EntryOcc { int lds = -1, occ = 0, n_cu = 0; };

template <int NSLAB, int NT>
int launch_entry(FArgs k, int nchan, hipStream_t st) {
  constexpr int RB = NSLAB * 128 + 16;
  const int lds = 3 * NT * NSLAB * 2048 + 4 * EN_TW * RB + 2 * NSLAB * 32 * 4 + EN_ZERO * 4 + nchan * EN_PLANE * 4;
  static unsigned long long attr_done = 0;
  auto kern = conv_entry_kernel<NSLAB, NT>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_entry): %s", hipGetErrorString(e));
  }
  // two persistent workgroups per CU (occupancy: registers / 76-90 KB of LDS), fewer when the XCD's tile range is shorter.  Occupancy and
  // CU count are cached per device (a box with mixed devices) and dynamic-LDS size; the knob is read per launch (A/B runs flip it in-process)
  static thread_local EntryOcc cache[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  EntryOcc& c = cache[dev];
  if (c.lds != lds) {
    c.lds = lds;
    if (hipDeviceGetAttribute(&c.n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c.n_cu <= 0) c.n_cu = 256;
    // workgroups that really are co-resident on a CU (registers and this launch's LDS)
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&c.occ, kern, 256, (size_t)lds) != hipSuccess || c.occ <= 0) c.occ = 1;
  }
  const int n_cu = c.n_cu;
  const char* const env = getenv("VIRNET_ENTRY_WGS_PER_CU");
  const int per_cu = env && atoi(env) > 0 ? atoi(env) : std::min(3, c.occ);
  const int wgs_per_xcd = std::max(1, std::min(k.tiles_per_xcd, per_cu * n_cu / 8));
  const virnet_pack_desc& e = k.ent;
  EntrySrc src{};
  for (int c = 0; c < 8; ++c) {
    src.base[c] = e.x;
    if (c < e.c0) { src.base[c] = e.x + (size_t)c * e.h * e.w; src.istride[c] = (unsigned)(e.c0 * e.h * e.w); src.mulx[c] = 1; }
    else if (c < e.c0 + e.ev) { src.base[c] = e.vec + (c - e.c0); src.istride[c] = (unsigned)e.ev; }
    else if (c < nchan) { src.base[c] = e.map + (size_t)(c - e.c0 - e.ev) * e.mh * e.mw; src.istride[c] = (unsigned)(e.em * e.mh * e.mw); src.mulm[c] = 1; src.sq[c] = e.map_sqrt ? 1u : 0u; }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(8 * wgs_per_xcd)), dim3(256), lds, st, k, src, nchan, wgs_per_xcd);
  return virnet::check_launch("conv_entry launch");
}

}  // namespace

extern "C" size_t virnet_entry_weight_floats(int cin, int n_pad) {
  const int nt = 3 * cin <= 16 ? 1 : 2;
  return (size_t)n_pad + (size_t)3 * nt * (n_pad / 32) * 512 + 4;   // n_pad scales + [dy][t][slab] x 2 KB + trailer {cin, nt, n_pad, tag}
}

extern "C" int virnet_pack_entry_weight(const float* w, int cout, int cin, int n_pad, float* packed, void* stream) {
  VIRNET_REQUIRE(w && packed, "virnet_pack_entry_weight: NULL pointer");
  VIRNET_REQUIRE(cin >= 1 && cin <= 8, "virnet_pack_entry_weight: cin=%d (the entry kernel gathers <= 8 input channels)", cin);
  VIRNET_REQUIRE(cout >= 1 && n_pad % 32 == 0 && n_pad >= cout && n_pad <= 96, "virnet_pack_entry_weight: cout=%d n_pad=%d (<= 96 output channels)", cout, n_pad);
  const int nt = 3 * cin <= 16 ? 1 : 2;
  hipLaunchKernelGGL(pack_entry_kernel, dim3((unsigned)n_pad), dim3(64), 0, static_cast<hipStream_t>(stream), w, cout, cin, nt, n_pad, packed,
                     reinterpret_cast<char*>(packed + n_pad));
  return virnet::check_launch("pack_entry launch");
}

#ifdef VIRNET_F16_TIMING
static long long* g_elog = nullptr;
extern "C" void virnet_debug_entry_timing_buffer(void* p) { g_elog = static_cast<long long*>(p); }   // tools/entry_timeline.py
#endif

extern "C" int virnet_conv_entry(const virnet_conv_desc* d, const virnet_pack_desc* e, void* stream) {
  VIRNET_REQUIRE(d != nullptr && e != nullptr, "virnet_conv_entry: NULL descriptor");
  VIRNET_REQUIRE(e->x && d->wpack, "virnet_conv_entry: image / wpack is NULL");
  VIRNET_REQUIRE(d->ks == 3 && d->stride == 1 && d->epi == VIRNET_EPI_NHWC, "virnet_conv_entry: only the stride-1 3x3 NHWC conv (ks=%d stride=%d epi=%d)", d->ks, d->stride, d->epi);
  VIRNET_REQUIRE(d->h == e->hp && d->w == e->wp && d->n == e->n && d->n > 0 && d->h > 0 && d->w > 0, "virnet_conv_entry: conv size %d x %d x %d != entry size %d x %d x %d",
                 d->n, d->h, d->w, e->n, e->hp, e->wp);
  const int nchan = e->c0 + e->ev + e->em;
  VIRNET_REQUIRE(e->c0 >= 1 && e->ev >= 0 && e->em >= 0 && nchan <= 8, "virnet_conv_entry: %d + %d + %d channels (the entry kernel gathers <= 8)", e->c0, e->ev, e->em);
  VIRNET_REQUIRE(e->sf >= 1 && e->h * e->sf <= e->hp && e->w * e->sf <= e->wp && e->hp < 2 * e->h * e->sf && e->wp < 2 * e->w * e->sf,
                 "virnet_conv_entry: %d x %d (x%d) does not reflect-pad to %d x %d", e->h, e->w, e->sf, e->hp, e->wp);
  VIRNET_REQUIRE(e->ev == 0 || e->vec, "virnet_conv_entry: ev=%d without vec", e->ev);
  VIRNET_REQUIRE(e->em == 0 || (e->map && e->msf >= 1 && e->mh * e->msf >= e->h * e->sf && e->mw * e->msf >= e->w * e->sf), "virnet_conv_entry: map %d x %d (x%d) does not cover the image", e->mh, e->mw, e->msf);
  VIRNET_REQUIRE(d->cout % 32 == 0 && d->cout <= 96 && d->n_pad == d->cout, "virnet_conv_entry: cout=%d (multiples of 32 up to 96; n_pad=%d)", d->cout, d->n_pad);
  VIRNET_REQUIRE((d->y_raw != nullptr) != (d->y_act != nullptr), "virnet_conv_entry: exactly one of y_raw / y_act");
  VIRNET_REQUIRE(!d->res && !d->mask && !d->mul && !d->in_mul && !d->in_act, "virnet_conv_entry: plain epilogue only");
  VIRNET_REQUIRE(!d->y_act || (d->slope >= 0.f && d->slope <= 1.f), "virnet_conv_entry: slope=%g outside [0,1]", d->slope);
  VIRNET_REQUIRE((long)d->h * d->w * d->cout * 4 < (1L << 31), "virnet_conv_entry: one image's output (%d x %d x %d fp32) must stay below 2 GB", d->h, d->w, d->cout);
  FArgs k{};
  k.inv_scale = d->wpack; k.wimg = reinterpret_cast<const char*>(d->wpack + d->n_pad);
  k.bias = d->bias; k.y_raw = d->y_raw; k.y_act = d->y_act;
  k.N = d->n; k.H = d->h; k.W = d->w; k.Cin = nchan; k.cout = d->cout; k.NP = d->n_pad; k.slope = d->slope;
  k.ent = *e;
  k.range_flag = virnet::range_flag_ptr();
#ifdef VIRNET_F16_TIMING
  k.tlog = g_elog;
#endif
  k.store_nt = virnet::store_nt_for((size_t)d->n * d->h * d->w * d->cout * 4);
  k.nty = (d->h + EN_TH - 1) / EN_TH;
  k.ntx = (d->w + EN_TW - 1) / EN_TW;
  k.ntiles = k.N * k.nty * k.ntx;
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  VIRNET_REQUIRE((unsigned long long)k.ntiles * (unsigned)(k.ntx * k.nty) < (1ull << 32), "virnet_conv_entry: %d tiles exceed the index arithmetic of one launch", k.ntiles);
  k.mg_ntx = div_magic(k.ntx);
  k.mg_tpi = div_magic(k.ntx * k.nty);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int nslab = d->cout / 32, nt = 3 * nchan <= 16 ? 1 : 2;
#define VIRNET_ENTRY_CASE(S_, T_) if (nslab == S_ && nt == T_) return launch_entry<S_, T_>(k, nchan, st);
  VIRNET_ENTRY_CASE(3, 1) VIRNET_ENTRY_CASE(3, 2) VIRNET_ENTRY_CASE(2, 1) VIRNET_ENTRY_CASE(2, 2) VIRNET_ENTRY_CASE(1, 1) VIRNET_ENTRY_CASE(1, 2)
#undef VIRNET_ENTRY_CASE
  return virnet::set_error("virnet_conv_entry: no kernel for %d slabs / %d k-steps per row", nslab, nt);
}
