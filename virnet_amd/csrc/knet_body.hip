// knet_body.hip -- KernelNet's eight RB_Layers (networks/KNet.py:28-39,46-48,54) as ONE persistent kernel: one workgroup per image
// keeps the 16 x 16 x 64 feature map on the CU for the whole body.
//
//   RB_Layer:  x + CA(conv2(lrelu_0.2(conv1(x))))     CA(h) = h * sigmoid(W2 lrelu_0.2(W1 mean_hw(h) + b1) + b2)      (KNet.py:15-26,32-38)
//
// The per-layer path (engine.knet_forward: conv_f16 x 2 + virnet_ca_scale_add per layer) is 24 launches of a few microseconds of work
// each on a map that fits one CU; here the map never leaves it:
//   * `cur` (fp32, the residual stream) lives in REGISTERS in the MFMA accumulator layout: wave w of 8 owns map rows 2w, 2w+1 (32 pixels
//     = the 32 columns of an MFMA) x 64 channels (two 32-row blocks) = 32 VGPRs; a conv's result lands in the same layout, so the
//     bias / LeakyReLU / gate / skip arithmetic is register-to-register.
//   * the conv INPUT is the split-fp16 image X in LDS: [hi|lo][4 chunks of 16 channels][18 x 18 padded pixels][32 B], zero ring written
//     once; a B fragment of tap (dy, dx) is one ds_read_b128 at the shifted pixel.
//   * weights: the layer's virnet_pack_f16_weight image ([slab][chunk][tap, column-major][hi|lo][1 KB], the same tensor conv_f16 reads)
//     streamed by LDS-DMA in stages of one (chunk, kernel column) = 12 KB, double buffered, one barrier per stage.
//   * arithmetic per product exactly as conv_f16.hip: three v_mfma_f32_32x32x16_f16 (w_lo x_hi + w_hi x_lo + w_hi x_hi), fp32
//     accumulation, per-row power-of-two weight scale undone in the epilogue -- 1 728 MFMAs per conv, 216 per wave.
//   * CALayer: per-channel sums by a fixed-order lane reduction + one LDS pass over the 8 waves (deterministic), the 64 -> cr -> 64 MLP by
//     64 threads, gate and skip applied in registers.
// Maps up to 16 x 16 (LR images up to 64 x 64 at the reference's stride-4 head); larger maps keep the per-layer path.
#include "conv_f16_common.h"

namespace {
using namespace virnet;

constexpr int KB_W = 18;                         // padded row length
constexpr int KB_PX = KB_W * KB_W;
constexpr int KB_CHUNK = KB_PX * 32;             // one (plane, chunk): [324 pixels][16 channels x fp16]
constexpr int KB_PLANE = 4 * KB_CHUNK;           // 41472
constexpr int KB_XBYTES = 2 * KB_PLANE;          // 82944
constexpr int KB_WSTAGE = 12 * 1024;             // [slab 2][dy 3][hi|lo] x 1 KB
constexpr int KB_SLAB_BYTES = 4 * 9 * 2048;      // one 32-row slab of the packed image: [chunk 4][tap 9][hi|lo][1 KB]
constexpr int KB_RING = 5;                       // weight stages in LDS: four in flight ahead of the one being multiplied
constexpr int KB_RED = (8 * 64 + 64 + 16 + 64) * 4;
// the layer's small parameters, staged once per layer (round 5: read from global memory where they are used, they were ~10 dependent
// round trips per layer behind s_waitcnt vmcnt(0) -- scale / bias vectors after each conv, the CALayer's two matrices in 16-float batches):
// [inv1 64 | b1 64 | inv2 64 | b2 64 | cab1 16 | cab2 64 | caw1 cr x 64 (<= 1024) | caw2 64 x cr (<= 1024)]
constexpr int KB_PAR_INV1 = 0, KB_PAR_B1 = 64, KB_PAR_INV2 = 128, KB_PAR_B2 = 192, KB_PAR_CAB1 = 256, KB_PAR_CAB2 = 272, KB_PAR_CAW1 = 336,
              KB_PAR_CAW2 = 336 + 1024, KB_PAR_FLOATS = 336 + 2048;
constexpr int KB_PAR_PT = (KB_PAR_FLOATS + 511) / 512;      // floats per thread
constexpr int KB_LDS = KB_XBYTES + KB_RING * KB_WSTAGE + KB_RED + KB_PAR_FLOATS * 4;
static_assert(KB_LDS <= 160 * 1024, "one workgroup per CU");
constexpr int KB_MAX_LAYERS = 8;

struct KnetArgs {
  const float* x;                                // NHWC [n][h][w][64]
  float* y;
  virnet_knet_layer L[KB_MAX_LAYERS];
  int nlayers, h, w, cr;
  int* range_flag;
  long long* tlog;                               // -DVIRNET_F16_TIMING builds (tools/knet_timeline.py)
};

__device__ __forceinline__ float kb_lrelu(float v, float s) { return v > 0.f ? v : v * s; }

__global__ __launch_bounds__(512, 2) void knet_body_kernel(const KnetArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const x_lds = smem;
  char* const w_lds = smem + KB_XBYTES;
  float* const part = reinterpret_cast<float*>(smem + KB_XBYTES + KB_RING * KB_WSTAGE);   // [8 waves][64]
  float* const mean = part + 8 * 64;
  float* const f1 = mean + 64;
  float* const gate = f1 + 16;
  float* const par = gate + 64;                                      // KB_PAR_FLOATS

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, lhi = lane >> 5;
  const int prow = 2 * wave + (l31 >> 4), pcol = l31 & 15;          // this lane's pixel (MFMA column l31)
  const bool inside = prow < a.h && pcol < a.w;
  const int img = blockIdx.x;
  const int pxo = ((prow + 1) * KB_W + (pcol + 1)) * 32;              // its record in a (plane, chunk)

  // ---- zero X once (the padding ring and the pixels outside an h x w < 16 x 16 map stay zero: "same" padding of every conv)
  for (int i = tid * 16; i < KB_XBYTES; i += 512 * 16) *reinterpret_cast<f32x4*>(x_lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();

  // ---- the residual stream in accumulator layout: cur[mb][4g + i] = channel 32 mb + 8 g + 4 lhi + i of the lane's pixel
  f32x16 cur[2];
  {
    const float* const px = a.x + (((size_t)img * a.h + prow) * a.w + pcol) * 64;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (inside) v = *reinterpret_cast<const f32x4*>(px + 32 * mb + 8 * g + 4 * lhi);
        cur[mb][4 * g] = v.x; cur[mb][4 * g + 1] = v.y; cur[mb][4 * g + 2] = v.z; cur[mb][4 * g + 3] = v.w;
      }
  }
  float amax = 0.f;
  // split an accumulator-layout tensor into X (hi / lo planes); every lane writes its own pixel's 64 channels
  auto put_split = [&](const f32x16 (&t)[2]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v0 = t[mb][4 * g], v1 = t[mb][4 * g + 1], v2 = t[mb][4 * g + 2], v3 = t[mb][4 * g + 3];
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v0), fabsf(v1))), fmaxf(fabsf(v2), fabsf(v3)));
        typedef _Float16 h4v __attribute__((ext_vector_type(4)));
        const h4v hi = {(_Float16)v0, (_Float16)v1, (_Float16)v2, (_Float16)v3};
        const h4v lo = {(_Float16)(v0 - (float)hi[0]), (_Float16)(v1 - (float)hi[1]), (_Float16)(v2 - (float)hi[2]), (_Float16)(v3 - (float)hi[3])};
        const int off = (2 * mb + (g >> 1)) * KB_CHUNK + pxo + (8 * (g & 1) + 4 * lhi) * 2;
        *reinterpret_cast<h4v*>(x_lds + off) = hi;
        *reinterpret_cast<h4v*>(x_lds + KB_PLANE + off) = lo;
      }
  };

#ifdef VIRNET_F16_TIMING
  long long kq_mma = 0, kq_wait = 0, kq_rest = 0, kq_mark = (long long)__builtin_amdgcn_s_memtime();
  const long long kq_start = kq_mark;
#define KB_TADD(var) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); var += n_ - kq_mark; kq_mark = n_; } while (0)
#else
#define KB_TADD(var) do { } while (0)
#endif
  const int lane16 = lane * 16;
  // ---- weight stream.  The 2 * nlayers convolutions are ONE sequence of stages gs = conv * 12 + (chunk * 3 + kernel column), 12 KB each
  // ([slab 2][dy 3][hi|lo] x 1 KB), living in a ring of KB_RING buffers; stage gs + KB_RING - 1 is requested when stage gs starts, so the
  // L2 latency of a piece (~1-2 us, several times the 0.25 us a stage multiplies) is covered five stages deep and the first stages of the
  // NEXT conv arrive while this conv's epilogue / CALayer runs.  Every wave issues exactly two pieces per stage (q = wave, wave + 8; the
  // four surplus ones repeat pieces 0..3: same bytes to the same place), so "stages gs + 1 and gs + 2 have landed" is the literal s_waitcnt
  // vmcnt(2 * (KB_RING - 3)) -- loads are retired in order, other loads in flight only make the wait longer.
  const int total_stages = a.nlayers * 24;
  auto issue = [&](int gs) {
    if (gs >= total_stages) return;
    const int cv = gs / 12, s = gs - cv * 12;
    const virnet_knet_layer& L = a.L[cv >> 1];
    const float* const pack = (cv & 1) ? L.w2pack : L.w1pack;
    const auto wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(pack + 64)), 0, 2 * KB_SLAB_BYTES, 0x00020000);
    char* const buf = w_lds + (gs % KB_RING) * KB_WSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int q = i * 8 + wave;
      if (q >= 12) q -= 12;
      const int slab = q / 6, r = q - slab * 6;
      const int src = slab * KB_SLAB_BYTES + (s / 3) * (9 * 2048) + (s % 3) * 6144 + r * 1024;
      (void)src; (void)wrs; (void)buf;
#if defined(__HIP_DEVICE_COMPILE__)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrs, (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, lane16, src, 0, 0);
#endif
    }
  };
  for (int gs = 0; gs < KB_RING - 1; ++gs) issue(gs);
  int gs0 = 0;                                     // first stage of the conv about to run
  // ---- one 64 -> 64 3x3 convolution of X: acc (accumulator layout) = W * X; returns with every wave past the last barrier
  // Fragments are read ONE STAGE AHEAD: at the barrier that closes stage gs the B fragments (X, static during a conv) and the first slab's
  // A fragments of stage gs + 1 are already in registers, so the first nine MFMAs of a stage start right behind the barrier and the second
  // slab's A reads return under them (before: eighteen ds_read_b128 behind every barrier, their LDS latency exposed 192 times).  For the A
  // fragments of stage gs + 1 to be readable DURING stage gs, the barrier that closes stage gs - 1 publishes stage gs + 1 as well: the wait in
  // front of it leaves 2 * (KB_RING - 3) pieces in flight instead of 2 * (KB_RING - 2).
  auto conv = [&](f32x16 (&acc)[2]) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
    auto x_base = [&](int s) {
      const int chunk = s / 3, dx = s - chunk * 3;
      return x_lds + chunk * KB_CHUNK + pxo + ((dx - 1) - KB_W) * 32 + lhi * 16;
    };
    auto w_base = [&](int gs) { return w_lds + (gs % KB_RING) * KB_WSTAGE + lane16; };
    // X complete (this wave's LDS stores) and stages gs0, gs0 + 1 landed (this wave's pieces; the barrier publishes the others')
    KB_TADD(kq_rest);
    if (gs0 + KB_RING - 1 <= total_stages) __builtin_amdgcn_s_waitcnt(((2 * (KB_RING - 3)) & 15) | 0x0070);
    else __builtin_amdgcn_s_waitcnt(0x0070);
    asm volatile("s_barrier" ::: "memory");
    KB_TADD(kq_wait);
    h8 bh[3], bl[3], a0h[3], a0l[3];
    {
      const char* const xb = x_base(0);
      const char* const wb = w_base(gs0);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        bh[dy] = *reinterpret_cast<const h8*>(xb + dy * (KB_W * 32));
        bl[dy] = *reinterpret_cast<const h8*>(xb + KB_PLANE + dy * (KB_W * 32));
        a0h[dy] = *reinterpret_cast<const h8*>(wb + (dy * 2 + 0) * 1024);
        a0l[dy] = *reinterpret_cast<const h8*>(wb + (dy * 2 + 1) * 1024);
      }
    }
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int gs = gs0 + s;
      const char* const wb = w_base(gs);
      issue(gs + KB_RING - 1);                     // into the buffer stage gs - 1 has just left (every wave is past its barrier)
      h8 a1h[3], a1l[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {             // second slab of THIS stage: under the first slab's MFMAs
        a1h[dy] = *reinterpret_cast<const h8*>(wb + (6 + dy * 2 + 0) * 1024);
        a1l[dy] = *reinterpret_cast<const h8*>(wb + (6 + dy * 2 + 1) * 1024);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l[dy], bh[dy], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h[dy], bl[dy], acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h[dy], bh[dy], acc[0], 0, 0, 0);
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l[dy], bh[dy], acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h[dy], bl[dy], acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h[dy], bh[dy], acc[1], 0, 0, 0);
      }
      if (s + 1 < 12) {                            // next stage's B and first-slab A fragments (stage gs + 1 was published a barrier ago)
        const char* const xb = x_base(s + 1);
        const char* const wn = w_base(gs + 1);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          bh[dy] = *reinterpret_cast<const h8*>(xb + dy * (KB_W * 32));
          bl[dy] = *reinterpret_cast<const h8*>(xb + KB_PLANE + dy * (KB_W * 32));
          a0h[dy] = *reinterpret_cast<const h8*>(wn + (dy * 2 + 0) * 1024);
          a0l[dy] = *reinterpret_cast<const h8*>(wn + (dy * 2 + 1) * 1024);
        }
      }
      // stages gs + 1 and gs + 2 landed (the 2 * (KB_RING - 3) pieces of the stages behind them may stay in flight; near the end of the
      // stream fewer were issued: wait for all), then the barrier: this stage's buffer is free, the next two are visible
      KB_TADD(kq_mma);
      if (gs + KB_RING <= total_stages) __builtin_amdgcn_s_waitcnt(((2 * (KB_RING - 3)) & 15) | 0x0070);
      else __builtin_amdgcn_s_waitcnt(0x0070);
      asm volatile("s_barrier" ::: "memory");
      KB_TADD(kq_wait);
    }
    gs0 += 12;
  };
  // acc * inverse row scale + bias, per channel of the accumulator layout
  auto scale_bias = [&](f32x16 (&acc)[2], const float* inv, const float* bias) {      // (both in the LDS parameter block)
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = 32 * mb + 8 * g + 4 * lhi;
        const f32x4 iv = *reinterpret_cast<const f32x4*>(inv + c);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
        acc[mb][4 * g] = fmaf(acc[mb][4 * g], iv.x, bv.x);
        acc[mb][4 * g + 1] = fmaf(acc[mb][4 * g + 1], iv.y, bv.y);
        acc[mb][4 * g + 2] = fmaf(acc[mb][4 * g + 2], iv.z, bv.z);
        acc[mb][4 * g + 3] = fmaf(acc[mb][4 * g + 3], iv.w, bv.w);
      }
  };

  const float inv_hw = 1.f / (float)(a.h * a.w);
  // A layer's small parameters travel global -> registers -> LDS one layer AHEAD: requested behind the previous layer's conv1 (the loads are in
  // flight under its epilogue and conv2, never the youngest entries of the vmcnt queue when a stage boundary waits), written to the LDS block
  // at the top of their layer -- behind the barrier that ended the previous layer's CALayer, the block's last reader.
  float pv[KB_PAR_PT];
  auto request_params = [&](int li) {
    const virnet_knet_layer& L = a.L[li];
#pragma unroll
    for (int i = 0; i < KB_PAR_PT; ++i) {
      const int j = i * 512 + tid;
      const float* src = nullptr;
      if (j < KB_PAR_B1) src = L.w1pack + j;
      else if (j < KB_PAR_INV2) src = L.b1 ? L.b1 + (j - KB_PAR_B1) : nullptr;
      else if (j < KB_PAR_B2) src = L.w2pack + (j - KB_PAR_INV2);
      else if (j < KB_PAR_CAB1) src = L.b2 ? L.b2 + (j - KB_PAR_B2) : nullptr;
      else if (j < KB_PAR_CAB2) src = j - KB_PAR_CAB1 < a.cr ? L.cab1 + (j - KB_PAR_CAB1) : nullptr;
      else if (j < KB_PAR_CAW1) src = L.cab2 + (j - KB_PAR_CAB2);
      else if (j < KB_PAR_CAW2) src = j - KB_PAR_CAW1 < a.cr * 64 ? L.caw1 + (j - KB_PAR_CAW1) : nullptr;
      else if (j < KB_PAR_FLOATS) src = j - KB_PAR_CAW2 < 64 * a.cr ? L.caw2 + (j - KB_PAR_CAW2) : nullptr;
      pv[i] = src ? *src : 0.f;
    }
  };
  request_params(0);
  for (int li = 0; li < a.nlayers; ++li) {
#pragma unroll
    for (int i = 0; i < KB_PAR_PT; ++i)
      if (i * 512 + tid < KB_PAR_FLOATS) par[i * 512 + tid] = pv[i];          // (published by conv1's barriers)
    put_split(cur);
    f32x16 t[2];
    conv(t);                                                          // KNet.py:32
    if (li + 1 < a.nlayers) request_params(li + 1);
    scale_bias(t, par + KB_PAR_INV1, par + KB_PAR_B1);
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) t[mb][r] = inside ? kb_lrelu(t[mb][r], 0.2f) : 0.f;   // KNet.py:33; outside the map: padding
    put_split(t);
    conv(t);                                                          // KNet.py:34
    scale_bias(t, par + KB_PAR_INV2, par + KB_PAR_B2);
    // ---- CALayer (KNet.py:15-26): channel means -> gate.  The per-channel sums over the wave's 32 pixels are a butterfly reduce-scatter
    // (lane bit 4, 3, .. 0 decides which half of its values a lane keeps and which it hands to its partner): 31 cross-lane moves per lane
    // instead of 5 x 32, and lane l31 ends up with the sum of value q = l31 (q = 16 mb + r) -- one channel per lane, fixed order.
    float rs[32];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = inside ? t[mb][r] : 0.f;
        t[mb][r] = v;
        rs[16 * mb + r] = v;
      }
#pragma unroll
    for (int half = 16; half >= 1; half >>= 1) {
      const bool up = (l31 & half) != 0;
#pragma unroll
      for (int i = 0; i < half; ++i) {
        const float lo_v = rs[i], hi_v = rs[i + half];
        const float give = up ? lo_v : hi_v, keep = up ? hi_v : lo_v;
        rs[i] = keep + __shfl_xor(give, half, 64);
      }
    }
    {
      const int r = l31 & 15;
      part[wave * 64 + 32 * (l31 >> 4) + 8 * (r >> 2) + 4 * lhi + (r & 3)] = rs[0];
    }
    __syncthreads();
    // 64 -> cr: wave w owns outputs w and w + 8; lane k forms caw1[j][k] * mean[k] (every wave sums the 8 partial sums of its lane's channel
    // itself), the 64 products meet in a lane reduction
    {
      float mk = 0.f;
#pragma unroll
      for (int wv = 0; wv < 8; ++wv) mk += part[wv * 64 + lane];
      mk *= inv_hw;
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        const int j = wave + 8 * jj;
        if (j < a.cr) {                                                // (wave-uniform)
          float pr = par[KB_PAR_CAW1 + j * 64 + lane] * mk;
#pragma unroll
          for (int m = 32; m >= 1; m >>= 1) pr += __shfl_xor(pr, m, 64);
          if (lane == 0) f1[j] = kb_lrelu(pr + par[KB_PAR_CAB1 + j], 0.2f);
        }
      }
    }
    __syncthreads();
    if (tid < 64) {
      float sg = par[KB_PAR_CAB2 + tid];
      for (int k = 0; k < a.cr; ++k) sg = fmaf(par[KB_PAR_CAW2 + tid * a.cr + k], f1[k], sg);
      gate[tid] = 1.f / (1.f + expf(-sg));
    }
    __syncthreads();
    // ---- x + CA(h)  (KNet.py:38), registers only
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gate + 32 * mb + 8 * g + 4 * lhi);
        cur[mb][4 * g] = fmaf(t[mb][4 * g], gv.x, cur[mb][4 * g]);
        cur[mb][4 * g + 1] = fmaf(t[mb][4 * g + 1], gv.y, cur[mb][4 * g + 1]);
        cur[mb][4 * g + 2] = fmaf(t[mb][4 * g + 2], gv.z, cur[mb][4 * g + 2]);
        cur[mb][4 * g + 3] = fmaf(t[mb][4 * g + 3], gv.w, cur[mb][4 * g + 3]);
      }
  }
  KB_TADD(kq_rest);
#ifdef VIRNET_F16_TIMING
  if (a.tlog && tid == 0) {
    long long* const o = a.tlog + (size_t)blockIdx.x * 8;
    o[0] = kq_start; o[1] = kq_mma; o[2] = kq_wait; o[3] = kq_rest; o[4] = (long long)__builtin_amdgcn_s_memtime();
  }
#endif
  range_report(a.range_flag, amax);
  if (inside) {
    float* const py = a.y + (((size_t)img * a.h + prow) * a.w + pcol) * 64;
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<f32x4*>(py + 32 * mb + 8 * g + 4 * lhi) = f32x4{cur[mb][4 * g], cur[mb][4 * g + 1], cur[mb][4 * g + 2], cur[mb][4 * g + 3]};
  }
}

#ifdef VIRNET_F16_TIMING
long long* g_kblog = nullptr;
#endif
}  // namespace

#ifdef VIRNET_F16_TIMING
extern "C" void virnet_debug_knet_timing_buffer(void* p) { g_kblog = static_cast<long long*>(p); }
#endif

extern "C" int virnet_knet_body(const float* x, float* y, const virnet_knet_layer* layers, int nlayers, int n, int h, int w, int c, int cr,
                                void* stream) {
  VIRNET_REQUIRE(x && y && layers, "virnet_knet_body: NULL pointer");
  VIRNET_REQUIRE(n > 0 && nlayers > 0, "virnet_knet_body: empty call n=%d nlayers=%d", n, nlayers);
  VIRNET_REQUIRE(c == 64, "virnet_knet_body: built for KernelNet's 64 feature channels, got %d", c);
  VIRNET_REQUIRE(h >= 1 && w >= 1 && h <= 16 && w <= 16, "virnet_knet_body: the map (%d x %d) must fit 16 x 16 (one workgroup holds it); use the per-layer kernels", h, w);
  VIRNET_REQUIRE(cr >= 1 && cr <= 16, "virnet_knet_body: cr=%d outside 1..16", cr);
  for (int i = 0; i < nlayers; ++i)
    VIRNET_REQUIRE(layers[i].w1pack && layers[i].w2pack && layers[i].caw1 && layers[i].cab1 && layers[i].caw2 && layers[i].cab2,
                   "virnet_knet_body: layer %d has a NULL weight pointer", i);
  static unsigned long long attr_done = 0;
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(knet_body_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, KB_LDS);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(knet_body): %s", hipGetErrorString(e));
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float* src = x;
  for (int l0 = 0; l0 < nlayers; l0 += KB_MAX_LAYERS) {             // (more than 8 RB_Layers: further launches, y -> y in place)
    KnetArgs a{};
    a.x = src; a.y = y; a.h = h; a.w = w; a.cr = cr;
    a.nlayers = nlayers - l0 < KB_MAX_LAYERS ? nlayers - l0 : KB_MAX_LAYERS;
    for (int i = 0; i < a.nlayers; ++i) a.L[i] = layers[l0 + i];
    a.range_flag = virnet::range_flag_ptr();
#ifdef VIRNET_F16_TIMING
    a.tlog = g_kblog;
#endif
    hipLaunchKernelGGL(knet_body_kernel, dim3((unsigned)n), dim3(512), KB_LDS, st, a);
    if (int rc = virnet::check_launch("knet_body launch")) return rc;
    src = y;
  }
  return 0;
}
