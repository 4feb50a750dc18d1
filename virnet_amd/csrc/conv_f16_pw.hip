// conv_f16_pw.hip -- the 2x2 stride-2 TRANSPOSED convolution (UpBlock.upsampler + bridge add, networks/AttResUNet.py:80,84-87) on the
// f16 matrix pipe with split fp32 operands.  Non-overlapping taps make it a pointwise GEMM to 4*Cout columns followed by a
// depth-to-space store:  D[(a*2+b)*Cout + co][pixel] = sum_ci Wt[ci][co][a][b] * X[ci][pixel]  ->  out[2y+a][2x+b][co].
// Arithmetic and accuracy argument: conv_f16.hip (hi/lo fp16 halves, three products, per-row power-of-two weight scale).
//
// Geometry.  No halo: a workgroup takes 128 CONSECUTIVE pixels of the flattened [N*H*W] list and 6 slabs (192 GEMM rows);
// 8 waves = (pixel block 0..3, slab group 0..1), each 1 x 3 accumulator blocks -- the shape of conv_f16_s2.hip.  K is walked in
// stages of three 16-channel chunks (one MFMA k-step each, three products deep): the pixel sub-tiles (hi/lo planes, swizzled 32-B
// records) and the weight fragments (LDS-DMA of the packed image) of stage s+1 are staged while stage s multiplies, both double
// buffered, one barrier per stage.  K is padded to a multiple of 48 in the weight image (zero rows), the pixel loads clamp the chunk.
// Epilogue: per-wave LDS turn-around of each 32-row slab (a slab lies inside one (a, b) phase because Cout % 32 == 0), bias + bridge
// (loaded before the first store), branch-free buffer stores of 128-B runs at the depth-to-space address.
#include "conv_f16_common.h"
#include <cstdlib>
#include <type_traits>

namespace {
using namespace virnet;

// KS = 16-channel chunks per K stage.  3 (round 2): 120 KB of LDS at 6 slabs, ONE workgroup per CU.  2 (round 4): 80 KB and <= 128
// VGPRs, so TWO workgroups share a CU -- the K loop is only Cin/32 stages and the epilogue moves 300 KB per workgroup, so a lone
// workgroup serialises latencies (profiles/r04_probes.md 5); with two, one's epilogue runs beside the other's loads and MFMAs.
template <int NG, int NREP, int KS>
__global__ __launch_bounds__(256 * NG, KS == 2 ? 2 * NG : NG) void conv_f16_pw_kernel(const FArgs a, const int nchr /* real 16-channel chunks */, const long npix) {
  constexpr int NT = 256 * NG, NWAVES = 4 * NG;
  constexpr int TP = 128;                              // pixels per workgroup
  constexpr int PLANE = TP * 32, KC = 2 * PLANE, XB = KS * KC;
  constexpr int NPIECE = KS * TP * 2;
  constexpr int PPT = (NPIECE + NT - 1) / NT;          // 2 (8 waves) or 3 (4 waves)
  static_assert((PPT - 1) * NT <= NPIECE, "surplus threads redo piece k-1");
  constexpr int SLABS = NG * NREP;
  constexpr int WGRP = KS * SLABS * 2048;
  constexpr int NDMA = KS * SLABS * 2;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const x_lds = smem;
  char* const w_lds = smem + 2 * XB;

  const int ncb = a.NP / (32 * SLABS);
  const int xcd = blockIdx.x & 7;
  const int q = blockIdx.x >> 3;
  const int cb = __builtin_amdgcn_readfirstlane(q % ncb);
  const int tile = __builtin_amdgcn_readfirstlane(xcd * a.tiles_per_xcd + q / ncb);
  if (q / ncb >= a.tiles_per_xcd || tile >= a.ntiles) return;
  const long p0 = (long)tile * TP;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pb = wv & 3, sg = wv >> 2;
  const int l31 = lane & 31, lhi = lane >> 5;
  const int nst = a.Cin / (16 * KS);                    // stages (Cin here = the PADDED contraction length, a multiple of 16 * KS)
  const int cx = nchr * 16;                             // channels of x

  // ---- pixel staging: piece -> (chunk-in-stage kc, pixel, half)
  size_t soff[PPT];
  int sdst[PPT], skc[PPT];
  bool sinb[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int qq = k * NT + tid;
    const int qc = qq < NPIECE ? qq : qq - NT;
    const int h = qc & 1, pl = (qc >> 1) % TP, kc = (qc >> 1) / TP;
    const long pg = p0 + pl;
    sinb[k] = pg < npix;
    soff[k] = (size_t)(pg < npix ? pg : npix - 1) * cx + h * 8;
    skc[k] = kc;
    sdst[k] = kc * KC + pl * 32 + ((h ^ ((pl >> 3) & 1)) << 4);
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
  auto stage_src = [&](int stage, int k) -> const float* {
    const int chunk = min(stage * KS + skc[k], nchr - 1);            // padded chunks re-read a valid one (their weights are zero)
    return a.x + soff[k] + chunk * 16;
  };

  // ---- weight DMA
  const size_t slab_bytes = (size_t)(a.Cin >> 4) * 2048;
  const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wimg + (size_t)(a.slab_base + cb * SLABS) * slab_bytes), 0,
                                                     (int)(SLABS * slab_bytes), 0x00020000);
  const int lane16w = lane * 16;
  auto dma_group = [&](int stage, char* wb) {
#pragma unroll
    for (int i = 0; i < (NDMA + NWAVES - 1) / NWAVES; ++i) {
      const int qd = i * NWAVES + wv;
      if (qd < NDMA) {
        const int tg = qd / (SLABS * 2), rem = qd - tg * (SLABS * 2);
        lds_dma16(wrs, wb + qd * 1024, lane16w, (rem >> 1) * (int)slab_bytes + ((stage * KS + tg) * 2 + (rem & 1)) * 1024);
      }
    }
  };

  const int pl_b = pb * 32 + l31;
  const int boff = pl_b * 32 + ((lhi ^ ((pl_b >> 3) & 1)) << 4);
  const int aoff = sg * NREP * 2048 + lane * 16;

  f32x16 acc[NREP];
#pragma unroll
  for (int nr = 0; nr < NREP; ++nr)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nr][r] = 0.f;

  // ---- prologue: stage 0
  dma_group(0, w_lds);
  {
    f32x4 r0[PPT], r1[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float* const src = stage_src(0, k);
      r0[k] = *reinterpret_cast<const f32x4*>(src);
      r1[k] = *reinterpret_cast<const f32x4*>(src + 4);
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) stage_store(x_lds, k, r0[k], r1[k]);
  }
  __syncthreads();

  h8 ah[2][NREP], al[2][NREP], bh[2], bl[2];
  auto read_ab = [&](const char* wb, const char* xb, int kc, int set) {
#pragma unroll
    for (int nr = 0; nr < NREP; ++nr) {
      ah[set][nr] = *reinterpret_cast<const h8*>(wb + kc * (SLABS * 2048) + nr * 2048 + aoff);
      al[set][nr] = *reinterpret_cast<const h8*>(wb + kc * (SLABS * 2048) + nr * 2048 + 1024 + aoff);
    }
    bh[set] = *reinterpret_cast<const h8*>(xb + kc * KC + boff);
    bl[set] = *reinterpret_cast<const h8*>(xb + kc * KC + PLANE + boff);
  };

  auto stage_fn = [&](int s, auto pc) {
    constexpr int P = decltype(pc)::value;
    const char* const xb = x_lds + P * XB;
    char* const xn = x_lds + (P ^ 1) * XB;
    const char* const wb = w_lds + P * WGRP;
    char* const wn = w_lds + (P ^ 1) * WGRP;
    if (s + 1 < nst) dma_group(s + 1, wn);
    const int sn = min(s + 1, nst - 1);
    f32x4 s0[PPT], s1[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const float* const src = stage_src(sn, k);
      s0[k] = *reinterpret_cast<const f32x4*>(src);
      s1[k] = *reinterpret_cast<const f32x4*>(src + 4);
    }
    read_ab(wb, xb, 0, KS == 3 ? P : 0);                    // (register set: KS 3: stage parity + k-step; KS 2: the k-step)
#pragma unroll
    for (int kc = 0; kc < KS; ++kc) {
      const int cur = KS == 3 ? (P + kc) & 1 : kc & 1;
      SB();
      if (kc < KS - 1) {
        read_ab(wb, xb, kc + 1, cur ^ 1);
      } else {
#pragma unroll
        for (int k = 0; k < PPT; ++k) stage_store(xn, k, s0[k], s1[k]);
      }
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int nr = 0; nr < NREP; ++nr) {
          const h8 wa = (part == 0) ? al[cur][nr] : ah[cur][nr];
          const h8 xv = (part == 1) ? bl[cur] : bh[cur];
          acc[nr] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa, xv, acc[nr], 0, 0, 0);
        }
      constexpr int NM = 3 * NREP;
      if (kc < KS - 1) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (i < 2 * NREP + 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 6 * PPT, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x200, 2 * PPT, 0);
      }
    }
    SB();
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  int s = 0;
  for (; s + 1 < nst; s += 2) { stage_fn(s, I0{}); stage_fn(s + 1, I1{}); }
  if (s < nst) stage_fn(s, I0{});

  range_report(a.range_flag, amax);
  // ---- epilogue: GEMM row slab -> phase (a, b) and channel offset; pixel -> (image, y, x) -> output (2y+a, 2x+b)
  const int C = a.cout;                                   // channels of the up-sampled tensor
  const int W2 = 2 * a.W;
  constexpr int TPIX = 144, NIT = 4, TREG = 32 * TPIX;
  static_assert(NWAVES * 2 * TREG <= 2 * XB + 2 * WGRP, "turn-around regions fit the K loop's LDS");
  char* const tbuf = smem + wv * (2 * TREG);
  const int cq = lane & 7, psub = lane >> 3;
  float* const y = a.y_act ? a.y_act : a.y_raw;
  const float slope_eff = a.y_act ? a.slope : 1.f;
  const float* const bp = a.bias ? a.bias : a.inv_scale;
  const float hb = a.bias ? 1.f : 0.f;
  const bool has_res = a.res != nullptr;
  // element offset of output pixel (2y, 2x), channel 0, for this lane's four pixels
  long pbase[NIT];
  bool pok[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const long pg = p0 + pb * 32 + it * 8 + psub;
    pok[it] = pg < npix;
    const long pc = pok[it] ? pg : npix - 1;
    const int xx = (int)(pc % a.W);
    const long t = pc / a.W;
    const int yy = (int)(t % a.H);
    const long im = t / a.H;
    pbase[it] = ((im * 2 * a.H + 2 * yy) * W2 + 2 * xx) * (long)C;
  }
  const int row0 = (a.slab_base + cb * SLABS + sg * NREP) * 32;       // first GEMM row of this wave
  constexpr bool RV_ALL = KS == 3;                       // KS 2 (128 VGPRs): the bridge values of ONE slab at a time
  f32x4 bias4[NREP], inv4[NREP], rv[RV_ALL ? NREP : 1][NIT];
  long eo[NREP];
#pragma unroll
  for (int nr = 0; nr < NREP; ++nr) {
    const int nrow = row0 + nr * 32;
    const int ab = nrow / C, co = nrow - ab * C + cq * 4;
    eo[nr] = ((long)(ab >> 1) * W2 + (ab & 1)) * C + co;
    inv4[nr] = *reinterpret_cast<const f32x4*>(a.inv_scale + nrow + cq * 4);
    bias4[nr] = *reinterpret_cast<const f32x4*>(bp + co);
    if (RV_ALL || nr == 0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        rv[RV_ALL ? nr : 0][it] = has_res ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.res + pbase[it] + eo[nr])) : f32x4{0.f, 0.f, 0.f, 0.f};      // (nt: a bridge byte is read once)
    }
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
  for (int nr = 0; nr < NREP; ++nr) {
    asm volatile("" ::"v"(inv4[nr]), "v"(bias4[nr]));
    if (RV_ALL || nr == 0) {
#pragma unroll
      for (int it = 0; it < NIT; ++it) asm volatile("" ::"v"(rv[RV_ALL ? nr : 0][it]));
    }
  }
#endif
#pragma unroll
  for (int nr = 0; nr < NREP; ++nr) {
    if (nr + 1 < NREP) turn_in(nr + 1, (nr + 1) & 1);
    const f32x4 b4 = bias4[nr] * hb;
    f32x4 tv[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) tv[it] = *reinterpret_cast<const f32x4*>(tbuf + (nr & 1) * TREG + (it * 8 + psub) * TPIX + cq * 16);
    f32x4 ov[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) ov[it] = lrelu4(tv[it] * inv4[nr] + b4 + rv[RV_ALL ? nr : 0][it], slope_eff);
    if (!RV_ALL && nr + 1 < NREP) {                      // next slab's bridge values: requested before this slab's stores (one vmcnt)
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        rv[0][it] = has_res ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.res + pbase[it] + eo[nr + 1])) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    if (a.store_nt) {                                    // (tensors larger than the Infinity Cache: non-temporal stores, conv_f16_wx4.hip)
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (pok[it]) __builtin_nontemporal_store(ov[it], reinterpret_cast<f32x4*>(y + pbase[it] + eo[nr]));
    } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (pok[it]) *reinterpret_cast<f32x4*>(y + pbase[it] + eo[nr]) = ov[it];
    }
  }
}

template <int NG, int NREP, int KS = 3>
int launch(FArgs k, int nchr, hipStream_t st) {
  constexpr int LDS_K = 2 * (KS * 2 * 128 * 32) + 2 * (KS * NG * NREP * 2048);
  constexpr int LDS_E = 4 * NG * 2 * 32 * 144;                            // the waves' turn-around regions
  constexpr int LDS = LDS_K > LDS_E ? LDS_K : LDS_E;
  static_assert(KS == 3 || LDS <= 80 * 1024, "KS 2: two workgroups per CU");
  static unsigned long long attr_done = 0;
  auto kern = conv_f16_pw_kernel<NG, NREP, KS>;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_f16_pw): %s", hipGetErrorString(e));
  }
  const long npix = (long)k.N * k.H * k.W;
  k.ntiles = (int)((npix + 127) / 128);
  k.tiles_per_xcd = (k.ntiles + 7) / 8;
  const int ncb = k.NP / (32 * NG * NREP);
  const unsigned grid = (unsigned)(8 * k.tiles_per_xcd * ncb);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256 * NG), LDS, st, k, nchr, npix);
  return virnet::check_launch("conv_f16_pw launch");
}

// Row r = (a*2+b)*cout + co of the pointwise GEMM of ConvTranspose2d(k2,s2) weights Wt[cin][cout][2][2]; contraction padded to k_pad.
__global__ void pack_f16_convt_kernel(const float* __restrict__ w, int cout, int cin, int k_pad, int n_pad, float* __restrict__ inv_scale,
                                      char* __restrict__ img) {
  const int row = blockIdx.x;
  const int rows = 4 * cout;
  const int ab = row / cout, co = row - ab * cout;
  auto wval = [&](int k) -> float { return (row < rows && k < cin) ? w[((size_t)k * cout + co) * 4 + ab] : 0.f; };
  __shared__ float red[256];
  float m = 0.f;
  for (int i = threadIdx.x; i < cin; i += blockDim.x) m = fmaxf(m, fabsf(wval(i)));
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  m = red[0];
  int e = 0;
  if (m > 0.f) { frexpf(m, &e); e = 14 - e; }
  e = max(-100, min(100, e));
  const float scale = ldexpf(1.f, e);
  if (threadIdx.x == 0) inv_scale[row] = ldexpf(1.f, -e);
  const int slab = row >> 5, col = row & 31, nch = k_pad >> 4;
  for (int k = threadIdx.x; k < k_pad; k += blockDim.x) {
    const float v = wval(k) * scale;
    const _Float16 hi = (_Float16)v;
    const _Float16 lo = (_Float16)(v - (float)hi);
    const int chunk = k >> 4, kk = k & 15;
    const size_t base = (((size_t)slab * nch + chunk) * 2) * 1024 + (size_t)(col + 32 * (kk >> 3)) * 16 + (kk & 7) * 2;
    *reinterpret_cast<_Float16*>(img + base) = hi;
    *reinterpret_cast<_Float16*>(img + base + 1024) = lo;
  }
}

}  // namespace

// padded contraction length of the weight image: the kernel walks K in stages of two 16-channel chunks when the channel count allows
// (two workgroups per CU), of three otherwise
static int convt_kpad(int cin) { return cin % 32 == 0 ? cin : (cin + 47) / 48 * 48; }

extern "C" size_t virnet_f16_convt_weight_floats(int cin, int cout) {
  const size_t k_pad = (size_t)convt_kpad(cin), n_pad = (size_t)4 * cout;
  return n_pad + n_pad * k_pad;
}

extern "C" int virnet_pack_f16_convt_weight(const float* w_iohw, int cout, int cin, float* packed, void* stream) {
  VIRNET_REQUIRE(w_iohw && packed, "virnet_pack_f16_convt_weight: NULL pointer");
  VIRNET_REQUIRE(cout > 0 && cout % 32 == 0 && cin > 0, "virnet_pack_f16_convt_weight: cout=%d must be a positive multiple of 32 (cin=%d)", cout, cin);
  const int k_pad = convt_kpad(cin), n_pad = 4 * cout;
  hipLaunchKernelGGL(pack_f16_convt_kernel, dim3((unsigned)n_pad), dim3(256), 0, static_cast<hipStream_t>(stream), w_iohw, cout, cin, k_pad,
                     n_pad, packed, reinterpret_cast<char*>(packed + n_pad));
  return virnet::check_launch("pack_f16_convt launch");
}

// `k`: x, wimg / inv_scale (from virnet_pack_f16_convt_weight), bias, res (bridge), y_raw | y_act, N, H, W (INPUT size), cout (channels of the
// up-sampled tensor), slope; cin_real = channels of x.
int virnet::launch_f16_convt(FArgs k, int cin_real, hipStream_t st) {
  const int nb = 4 * k.cout / 32;
  k.Cin = convt_kpad(cin_real);
  const int nchr = cin_real >> 4;
  int n6 = nb / 6, rem = nb - 6 * n6;
  // 2-chunk stages / two workgroups per CU for the 6-slab groups when the padded contraction length allows (VIRNET_CONVT_KS=3: round 2's form)
  const char* const ks_env = getenv("VIRNET_CONVT_KS");
  const bool ks2 = k.Cin % 32 == 0 && !(ks_env && atoi(ks_env) == 3 && k.Cin % 48 == 0);
  int max_group = 6;
  if (const char* f = getenv("VIRNET_CONVT_SLABS")) max_group = atoi(f);      // tuning aid: largest slab group per workgroup (6 default, 3, 2)
  if (max_group < 6) { n6 = 0; rem = nb; }
  if (rem == 1 && n6 >= 1) { n6 -= 1; rem = 7; }
  int n3 = rem / 3, rem2 = rem - 3 * n3;
  if (max_group < 3) { n3 = 0; rem2 = rem; }
  if (rem2 == 1 && n3 >= 1) { n3 -= 1; rem2 = 4; }
  const int n2 = rem2 / 2, n1 = rem2 - 2 * n2;
  int base = 0;
  auto run = [&](int ng, int nrep, int groups) -> int {
    if (groups <= 0) return 0;
    FArgs kk = k;
    kk.slab_base = base;
    kk.NP = groups * ng * nrep * 32;
    base += groups * ng * nrep;
    if (ng == 2 && nrep == 3) return ks2 ? launch<2, 3, 2>(kk, nchr, st) : launch<2, 3>(kk, nchr, st);
    if (ng == 1 && nrep == 3) return ks2 ? launch<1, 3, 2>(kk, nchr, st) : launch<1, 3>(kk, nchr, st);
    if (ng == 1 && nrep == 2) return ks2 ? launch<1, 2, 2>(kk, nchr, st) : launch<1, 2>(kk, nchr, st);
    return ks2 ? launch<1, 1, 2>(kk, nchr, st) : launch<1, 1>(kk, nchr, st);
  };
  if (int rc = run(2, 3, n6)) return rc;
  if (int rc = run(1, 3, n3)) return rc;
  if (int rc = run(1, 2, n2)) return rc;
  return run(1, 1, n1);
}
