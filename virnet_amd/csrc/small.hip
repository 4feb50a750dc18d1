// small.hip -- the latency-bound pieces around the MFMA convolutions: KNet's strided head, global average pools with their
// finishing ops, the CALayer gate, and the SFT (AttLayer) generators.  None of these is a dense contraction worth MFMA tiles:
// they are reductions / tiny per-image or per-pixel MLPs, written as plain wave64 VALU kernels with LDS reductions.
#include "common.h"
#include "../../include/virnet_hip.h"

namespace {

__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ float sigmoidf(float v) { return 1.f / (1.f + expf(-v)); }

// ----------------------------------------------------------------------------------------------------------------
// KernelNet.head (networks/KNet.py:45,53): Conv2d(cin -> 64-multiple, k=9, s=4, p=4, no bias), NCHW in, NHWC out.
// Block: 256 threads = 4 output pixels x 64 output channels; the [K][64] weight tile sits transposed in LDS (row stride 65
// floats -> conflict-free for both the transposing write and the channel-parallel read), and the 4 pixels' K input samples next to it
// (zero where the 9 x 9 window leaves the image), so that a tap is two LDS reads and an FMA.  Round 5 (tools/probes/sisr_n1_trace.sh:
// 32 us for one 64 x 64 image): the weight tile used to be staged by a load -> wait -> store loop of K * 64 / 256 = 61 dependent round
// trips, and every tap read its input sample from global memory behind a bounds branch -- 243 more; now eight rows' loads are in flight
// at a time and the samples of a pixel group arrive in one.
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_head_s4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ out, int n, int cin, int h, int wd, int cout,
                                                           int oh, int ow, int pix_per_block) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [K][65] | [4][K]
  const int K = cin * 81;
  float* const xs = wl + K * 65;
  const int ctile = blockIdx.y;  // 64 output channels
  // weights: row co of the tile = K contiguous floats; thread t takes column k = t (+ 256 ..) of eight rows at a time
  for (int k = threadIdx.x; k < K; k += 256) {
    const float* const wp = w + (size_t)ctile * 64 * K + k;
#pragma unroll 1
    for (int c0 = 0; c0 < 64; c0 += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = wp[(size_t)(c0 + j) * K];
#pragma unroll
      for (int j = 0; j < 8; ++j) wl[k * 65 + c0 + j] = v[j];
    }
  }
  const int co = threadIdx.x & 63, ps = threadIdx.x >> 6;
  const long npix = (long)n * oh * ow;
  const long p0 = (long)blockIdx.x * pix_per_block;
  for (long pg = p0; pg < p0 + pix_per_block && pg < npix; pg += 4) {
    __syncthreads();                                        // (weights staged / the previous group's samples consumed)
    const long p = pg + ps;
    const bool live = p < npix && p < p0 + pix_per_block;
    const int ox = (int)(p % ow), oy = (int)((p / ow) % oh), img = (int)(p / ((long)ow * oh));
    // this wave's pixel: sample k = (ci, ky, kx), lane l takes k = l, l + 64, ..
    for (int k = co; k < K; k += 64) {
      const int ci = k / 81, r = k - ci * 81, ky = r / 9, kx = r - ky * 9;
      const int iy = oy * 4 - 4 + ky, ix = ox * 4 - 4 + kx;
      float v = 0.f;
      if (live && (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)wd) v = x[(((size_t)img * cin + ci) * h + iy) * wd + ix];
      xs[ps * K + k] = v;
    }
    __syncthreads();
    if (live) {
      float acc = 0.f;
      const float* const xp = xs + ps * K;
#pragma unroll 9
      for (int k = 0; k < K; ++k) acc = fmaf(xp[k], wl[k * 65 + co], acc);      // (a sample outside the image is a zero: fma(0, w, acc) = acc)
      out[(size_t)p * cout + ctile * 64 + co] = acc;
    }
  }
}

// Weight gradient of the same layer (SISR training step, train_SISR.py:207-224): dw[co][ci][ky][kx] = sum over images and output
// pixels of dy[n][oy][ox][co] * x[n][ci][4oy+ky-4][4ox+kx-4].  One block per (ci, ky, kx); a thread owns one output channel and a
// quarter of the (image, output row) pairs (the input sample is a wave-uniform broadcast, dy is read channel-contiguous), the four
// partial sums meet in LDS.  0.17 % of the SISR FLOPs: latency class.
__global__ __launch_bounds__(256) void conv_head_s4_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dw, int n, int cin, int h, int wd, int cout,
                                                                  int oh, int ow) {
  __shared__ float red[4][64];
  const int kx = blockIdx.x % 9, ky = (blockIdx.x / 9) % 9, ci = blockIdx.x / 81;
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  for (int co0 = 0; co0 < cout; co0 += 64) {
    const int co = co0 + lane;
    float acc0 = 0.f, acc1 = 0.f;
    if (co < cout) {
      for (int row = part; row < n * oh; row += 4) {         // (image, output row) pairs
        const int img = row / oh, oy = row - img * oh;
        const int iy = oy * 4 - 4 + ky;
        if ((unsigned)iy >= (unsigned)h) continue;
        const float* const xp = x + (((size_t)img * cin + ci) * h + iy) * wd;
        const float* const dp = dy + ((size_t)img * oh + oy) * ow * cout + co;
        int ox = 0;
        for (; ox + 1 < ow; ox += 2) {
          const int ix0 = ox * 4 - 4 + kx, ix1 = ix0 + 4;
          if ((unsigned)ix0 < (unsigned)wd) acc0 = fmaf(dp[(size_t)ox * cout], xp[ix0], acc0);
          if ((unsigned)ix1 < (unsigned)wd) acc1 = fmaf(dp[(size_t)(ox + 1) * cout], xp[ix1], acc1);
        }
        if (ox < ow) {
          const int ix = ox * 4 - 4 + kx;
          if ((unsigned)ix < (unsigned)wd) acc0 = fmaf(dp[(size_t)ox * cout], xp[ix], acc0);
        }
      }
    }
    red[part][lane] = acc0 + acc1;
    __syncthreads();
    if (part == 0 && co < cout) dw[(((size_t)co * cin + ci) * 9 + ky) * 9 + kx] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    __syncthreads();
  }
}

// ----------------------------------------------------------------------------------------------------------------
// block-wide sum of one float per thread (256 threads): wave64 shuffles, then 4 partials through LDS.
// ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats */) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// planar GAP: one block per (n, c) plane
__global__ __launch_bounds__(256) void gap_nchw_kernel(const float* __restrict__ x, float* __restrict__ out, int c, int hw,
                                                       int finish, float lo, float hi) {
  __shared__ float red[4];
  const float* p = x + (size_t)blockIdx.x * hw;
  float s = 0.f;
  for (int i = threadIdx.x; i < hw; i += 256) s += p[i];
  const float mean = block_sum_256(s, red) / (float)hw;
  if (threadIdx.x == 0) {
    const int ch = blockIdx.x % c;
    float v = mean;
    if (finish == VIRNET_GAP_EXPCLAMP || (finish == VIRNET_GAP_KINFO && ch < c - 1)) v = expf(fminf(fmaxf(mean, lo), hi));
    else if (finish == VIRNET_GAP_KINFO) v = tanhf(mean);
    out[blockIdx.x] = v;
  }
}

// CALayer gate: block per image. threads = (256/c pixel phases) x c channels
__global__ __launch_bounds__(256) void ca_gate_kernel(const float* __restrict__ x, const float* __restrict__ w1,
                                                      const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, float* __restrict__ gate, int hw, int c,
                                                      int cr) {
  __shared__ float part[256];
  __shared__ float mean[256];
  __shared__ float f1[64];
  const int img = blockIdx.x;
  const int ch = threadIdx.x % c, ph = threadIdx.x / c, nph = 256 / c;
  const float* p = x + (size_t)img * hw * c;
  float s = 0.f;
  for (int i = ph; i < hw; i += nph) s += p[(size_t)i * c + ch];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < c) {
    float t = 0.f;
    for (int k = 0; k < nph; ++k) t += part[k * c + threadIdx.x];
    mean[threadIdx.x] = t / (float)hw;
  }
  __syncthreads();
  if (threadIdx.x < cr) {
    float t = b1[threadIdx.x];
    for (int k = 0; k < c; ++k) t = fmaf(w1[threadIdx.x * c + k], mean[k], t);
    f1[threadIdx.x] = lrelu(t, 0.2f);
  }
  __syncthreads();
  if (threadIdx.x < c) {
    float t = b2[threadIdx.x];
    for (int k = 0; k < cr; ++k) t = fmaf(w2[threadIdx.x * cr + k], f1[k], t);
    gate[(size_t)img * c + threadIdx.x] = sigmoidf(t);
  }
}

__global__ __launch_bounds__(256) void scale_add_kernel(const float4* __restrict__ hcv, const float* __restrict__ gate,
                                                        const float4* __restrict__ skip, float4* __restrict__ out, int hw,
                                                        int c4, size_t total4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
    const int cq = (int)(i % c4);
    const size_t img = i / ((size_t)hw * c4);
    const float4 g = reinterpret_cast<const float4*>(gate + img * c4 * 4)[cq];
    const float4 a = hcv[i], b = skip[i];
    out[i] = make_float4(fmaf(a.x, g.x, b.x), fmaf(a.y, g.y, b.y), fmaf(a.z, g.z, b.z), fmaf(a.w, g.w, b.w));
  }
}

// CALayer + RB_Layer tail in ONE launch (KNet.py:15-26,38): out = hcv * gate(hcv) + skip for maps small enough that one workgroup
// can hold an image in registers (KernelNet's 16 x 16 x 64 map: 16 float4 per thread).  The two-kernel form (ca_gate + scale_add)
// reads every element through 64 dependent loads per thread to form the mean (18.7 us per launch on the SISR forward, 8 launches);
// here all loads of the image (and of the skip tensor) are in flight at once and the mean goes through one LDS reduction.
// thread t owns float4 items t, t + 1024, ...: its channel quad t % c4 is the same for all of them (c4 divides 1024).
template <int V>
__global__ __launch_bounds__(1024) void ca_scale_add_kernel(const float4* __restrict__ hcv, const float* __restrict__ w1,
                                                            const float* __restrict__ b1, const float* __restrict__ w2,
                                                            const float* __restrict__ b2, const float4* __restrict__ skip,
                                                            float4* __restrict__ out, int hw, int c, int cr) {
  __shared__ float4 part[1024];
  __shared__ float mean[256];
  __shared__ float f1[64];
  __shared__ float gate[256];
  const int tid = threadIdx.x, c4 = c >> 2;
  const int n4 = hw * c4;                              // float4 items of one image
  const size_t base = (size_t)blockIdx.x * n4;
  float4 v[V], sk[V];
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int i = k * 1024 + tid;
    v[k] = i < n4 ? hcv[base + i] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr bool SKIP_EARLY = V <= 8;                  // (16 items: the 128 registers of a 16-wave workgroup hold the image OR both tensors)
  if constexpr (SKIP_EARLY) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const int i = k * 1024 + tid;
      sk[k] = i < n4 ? skip[base + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) { s.x += v[k].x; s.y += v[k].y; s.z += v[k].z; s.w += v[k].w; }
  part[tid] = s;
  __syncthreads();
  if (tid < c) {                                       // channel tid = quad tid/4, component tid%4: sum over the 1024/c4 threads of the quad
    const int q = tid >> 2, e = tid & 3;
    float t = 0.f;
    for (int k = q; k < 1024; k += c4) t += reinterpret_cast<const float*>(&part[k])[e];
    mean[tid] = t / (float)hw;
  }
  __syncthreads();
  if (tid < cr) {
    float t = b1[tid];
    for (int k = 0; k < c; ++k) t = fmaf(w1[tid * c + k], mean[k], t);
    f1[tid] = lrelu(t, 0.2f);
  }
  __syncthreads();
  if (tid < c) {
    float t = b2[tid];
    for (int k = 0; k < cr; ++k) t = fmaf(w2[tid * cr + k], f1[k], t);
    gate[tid] = sigmoidf(t);
  }
  __syncthreads();
  const float4 g = reinterpret_cast<const float4*>(gate)[tid % c4];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int i = k * 1024 + tid;
    if constexpr (!SKIP_EARLY) sk[k] = i < n4 ? skip[base + i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) out[base + i] = make_float4(fmaf(v[k].x, g.x, sk[k].x), fmaf(v[k].y, g.y, sk[k].y), fmaf(v[k].z, g.z, sk[k].z), fmaf(v[k].w, g.w, sk[k].w));
  }
}

// Backward of the SFT pre-activation a = lrelu(u), u = x*mul + add with per-image vectors (AttResUNet.py:54-58; training step of the
// SISR model): from dA = dL/da,  du = dA * lrelu'(u);  dx = du * mul (+ res: the skip gradient);  dmul[n][c] += sum_p du * x;
// dadd[n][c] += sum_p du.  One pass over the two tensors; a block walks a run of pixels of ONE image with a fixed channel quad per
// thread, so the per-image sums stay in registers until one LDS reduction + 2*C atomics per block.  grid (chunks, n).
__global__ __launch_bounds__(256) void sft_backward_kernel(const float4* __restrict__ da, const float4* __restrict__ x, const float* __restrict__ mul,
                                                           const float* __restrict__ add, const float4* __restrict__ res, float slope,
                                                           float4* __restrict__ dx, float* __restrict__ dmul, float* __restrict__ dadd, long hw, int c4,
                                                           int chunk) {
  __shared__ float4 rm[256], ra[256];
  const int img = blockIdx.y;
  const int pl_n = 256 / c4;                               // pixel lanes per block
  const int q = threadIdx.x % c4, pl = threadIdx.x / c4;
  float4 sm = make_float4(0.f, 0.f, 0.f, 0.f), sa = sm;
  if (pl < pl_n) {
    const float4 m4 = reinterpret_cast<const float4*>(mul + (size_t)img * c4 * 4)[q];
    const float4 a4 = reinterpret_cast<const float4*>(add + (size_t)img * c4 * 4)[q];
    const long p0 = (long)blockIdx.x * chunk, p1 = p0 + chunk < hw ? p0 + chunk : hw;
    for (long p = p0 + pl; p < p1; p += pl_n) {
      const size_t i = ((size_t)img * hw + p) * c4 + q;
      const float4 g = da[i], v = x[i];
      float4 du;
      du.x = fmaf(v.x, m4.x, a4.x) > 0.f ? g.x : g.x * slope;
      du.y = fmaf(v.y, m4.y, a4.y) > 0.f ? g.y : g.y * slope;
      du.z = fmaf(v.z, m4.z, a4.z) > 0.f ? g.z : g.z * slope;
      du.w = fmaf(v.w, m4.w, a4.w) > 0.f ? g.w : g.w * slope;
      float4 o = make_float4(du.x * m4.x, du.y * m4.y, du.z * m4.z, du.w * m4.w);
      if (res) { const float4 r = res[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
      dx[i] = o;
      sm.x = fmaf(du.x, v.x, sm.x); sm.y = fmaf(du.y, v.y, sm.y); sm.z = fmaf(du.z, v.z, sm.z); sm.w = fmaf(du.w, v.w, sm.w);
      sa.x += du.x; sa.y += du.y; sa.z += du.z; sa.w += du.w;
    }
  }
  rm[threadIdx.x] = sm; ra[threadIdx.x] = sa;
  __syncthreads();
  if (threadIdx.x < c4) {
    float4 tm = rm[threadIdx.x], ta = ra[threadIdx.x];
    for (int k = 1; k < pl_n; ++k) {
      const float4 um = rm[threadIdx.x + k * c4], ua = ra[threadIdx.x + k * c4];
      tm.x += um.x; tm.y += um.y; tm.z += um.z; tm.w += um.w;
      ta.x += ua.x; ta.y += ua.y; ta.z += ua.z; ta.w += ua.w;
    }
    float* const om = dmul + ((size_t)img * c4 + threadIdx.x) * 4;
    float* const oa = dadd + ((size_t)img * c4 + threadIdx.x) * 4;
    atomicAdd(om + 0, tm.x); atomicAdd(om + 1, tm.y); atomicAdd(om + 2, tm.z); atomicAdd(om + 3, tm.w);
    atomicAdd(oa + 0, ta.x); atomicAdd(oa + 1, ta.y); atomicAdd(oa + 2, ta.z); atomicAdd(oa + 3, ta.w);
  }
}

// AttLayer on a per-image vector: block per image
__device__ __forceinline__ void sft_vec_body(const float* __restrict__ vec, const virnet_sft_weights& wt, float* __restrict__ mul,
                                             float* __restrict__ add, float* e, float* f1, float* f2) {
  const int img = blockIdx.x;
  if (threadIdx.x < wt.e) e[threadIdx.x] = vec[(size_t)img * wt.e + threadIdx.x];
  __syncthreads();
  for (int j = threadIdx.x; j < wt.nf1; j += 256) {
    float t = wt.b1[j];
    for (int k = 0; k < wt.e; ++k) t = fmaf(wt.w1[j * wt.e + k], e[k], t);
    f1[j] = lrelu(t, 0.2f);
  }
  __syncthreads();
  for (int j = threadIdx.x; j < wt.nf2; j += 256) {
    float t = wt.b2[j];
    for (int k = 0; k < wt.nf1; ++k) t = fmaf(wt.w2[j * wt.nf1 + k], f1[k], t);
    f2[j] = lrelu(t, 0.2f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < wt.nf; c += 256) {
    float tm = wt.bm[c], ta = wt.ba[c];
    for (int k = 0; k < wt.nf2; ++k) {
      tm = fmaf(wt.wm[c * wt.nf2 + k], f2[k], tm);
      ta = fmaf(wt.wa[c * wt.nf2 + k], f2[k], ta);
    }
    mul[(size_t)img * wt.nf + c] = sigmoidf(tm);
    add[(size_t)img * wt.nf + c] = ta;
  }
}

__global__ __launch_bounds__(256) void sft_vec_kernel(const float* __restrict__ vec, const virnet_sft_weights wt,
                                                      float* __restrict__ mul, float* __restrict__ add) {
  __shared__ float e[16];
  __shared__ float f1[64];
  __shared__ float f2[128];
  sft_vec_body(vec, wt, mul, add, e, f1, f2);
}

// ... of SEVERAL AttLayers on the same vector in one launch (grid.y = layer): every SFT layer of the down path depends on nothing but the
// conditioning vector, and a single-image SISR forward spent twelve dependent ~5-us launches on them (tools/probes/sisr_n1_trace.sh)
constexpr int SFT_MULTI_MAX = 16;
struct SftMulti {
  virnet_sft_weights wt[SFT_MULTI_MAX];
  float* mul[SFT_MULTI_MAX];
  float* add[SFT_MULTI_MAX];
};
__global__ __launch_bounds__(256) void sft_vec_multi_kernel(const float* __restrict__ vec, const SftMulti m) {
  __shared__ float e[16];
  __shared__ float f1[64];
  __shared__ float f2[128];
  const int l = blockIdx.y;
  sft_vec_body(vec, m.wt[l], m.mul[l], m.add[l], e, f1, f2);
}

// AttLayer per pixel + SFT + LeakyReLU: block = PT pixels; thread <-> output channel (looped), weights read once per block
constexpr int SFT_PT = 16;
__global__ __launch_bounds__(256) void sft_apply_kernel(const float* __restrict__ raw, const float* __restrict__ rec,
                                                        const virnet_sft_weights wt, float* __restrict__ act, int n, int h, int w,
                                                        int step, int chan0) {
  __shared__ float e[SFT_PT][16];
  __shared__ float f1[SFT_PT][64];
  __shared__ float f2[SFT_PT][128];
  const size_t npix = (size_t)n * h * w;
  const size_t p0 = (size_t)blockIdx.x * SFT_PT;
  const int hp = h * step, wp = w * step;
  for (int i = threadIdx.x; i < SFT_PT * wt.e; i += 256) {
    const int pi = i / wt.e, k = i - pi * wt.e;
    const size_t p = p0 + pi;
    float v = 0.f;
    if (p < npix) {
      const int x = (int)(p % w), y = (int)((p / w) % h), img = (int)(p / ((size_t)w * h));
      v = rec[(((size_t)img * hp + (size_t)y * step) * wp + (size_t)x * step) * 16 + chan0 + k];
    }
    e[pi][k] = v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SFT_PT * wt.nf1; i += 256) {
    const int pi = i / wt.nf1, j = i - pi * wt.nf1;
    float t = wt.b1[j];
    for (int k = 0; k < wt.e; ++k) t = fmaf(wt.w1[j * wt.e + k], e[pi][k], t);
    f1[pi][j] = lrelu(t, 0.2f);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SFT_PT * wt.nf2; i += 256) {
    const int pi = i / wt.nf2, j = i - pi * wt.nf2;
    float t = wt.b2[j];
    for (int k = 0; k < wt.nf1; ++k) t = fmaf(wt.w2[j * wt.nf1 + k], f1[pi][k], t);
    f2[pi][j] = lrelu(t, 0.2f);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < wt.nf; c += 256) {
    float tm[SFT_PT], ta[SFT_PT];
#pragma unroll
    for (int pi = 0; pi < SFT_PT; ++pi) { tm[pi] = wt.bm[c]; ta[pi] = wt.ba[c]; }
    for (int k = 0; k < wt.nf2; ++k) {
      const float wm = wt.wm[c * wt.nf2 + k], wa = wt.wa[c * wt.nf2 + k];
#pragma unroll
      for (int pi = 0; pi < SFT_PT; ++pi) {
        tm[pi] = fmaf(wm, f2[pi][k], tm[pi]);
        ta[pi] = fmaf(wa, f2[pi][k], ta[pi]);
      }
    }
#pragma unroll
    for (int pi = 0; pi < SFT_PT; ++pi) {
      const size_t p = p0 + pi;
      if (p < npix) {
        const size_t o = p * wt.nf + c;
        act[o] = lrelu(fmaf(raw[o], sigmoidf(tm[pi]), ta[pi]), 0.2f);
      }
    }
  }
}

int check_sft(const virnet_sft_weights* wt, const char* who) {
  VIRNET_REQUIRE(wt && wt->w1 && wt->b1 && wt->w2 && wt->b2 && wt->wm && wt->bm && wt->wa && wt->ba, "%s: NULL weights", who);
  VIRNET_REQUIRE(wt->e >= 1 && wt->e <= 16 && wt->nf1 >= 1 && wt->nf1 <= 64 && wt->nf2 >= 1 && wt->nf2 <= 128 && wt->nf >= 1,
                 "%s: unsupported widths e=%d nf1=%d nf2=%d nf=%d", who, wt->e, wt->nf1, wt->nf2, wt->nf);
  return 0;
}

}  // namespace

extern "C" int virnet_conv_head_s4(const float* x, const float* w, float* out, int n, int cin, int h, int w_, int cout,
                                   void* stream) {
  VIRNET_REQUIRE(x && w && out, "virnet_conv_head_s4: NULL pointer");
  VIRNET_REQUIRE(n > 0 && h > 0 && w_ > 0 && cin > 0, "virnet_conv_head_s4: bad shape");
  VIRNET_REQUIRE(cout % 64 == 0 && cout > 0, "virnet_conv_head_s4: cout=%d must be a multiple of 64", cout);
  const int K = cin * 81;
  const size_t lds = ((size_t)K * 65 + 4 * (size_t)K) * sizeof(float);
  VIRNET_REQUIRE(lds <= 160 * 1024, "virnet_conv_head_s4: cin=%d too large for the LDS weight tile", cin);
  static unsigned long long attr_done = 0;     // one bit per device
  if (virnet::first_use_on_device(attr_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_s4_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return virnet::set_error("hipFuncSetAttribute(conv_head_s4): %s", hipGetErrorString(e));
  }
  const int oh = (h - 1) / 4 + 1, ow = (w_ - 1) / 4 + 1;
  const long npix = (long)n * oh * ow;
  // 4 pixels per block (one per pixel slot) while the grid stays moderate: a block walks its pixels one after the other, 243 dependent
  // taps each, so 32 pixels per block left a single 64 x 64 image to 8 blocks and 130 us (the weight tile is re-staged per block from L2:
  // 62 KB, cheap next to that)
  int ppb = 4;
  while ((npix + ppb - 1) / ppb > 4096) ppb *= 2;
  const int gx = (int)((npix + ppb - 1) / ppb);
  hipLaunchKernelGGL(conv_head_s4_kernel, dim3(gx, cout / 64), dim3(256), lds, static_cast<hipStream_t>(stream), x, w, out, n,
                     cin, h, w_, cout, oh, ow, ppb);
  return virnet::check_launch("conv_head_s4 launch");
}

extern "C" int virnet_conv_head_s4_wgrad(const float* x, const float* dy, float* dw, int n, int cin, int h, int w_, int cout,
                                         void* stream) {
  VIRNET_REQUIRE(x && dy && dw, "virnet_conv_head_s4_wgrad: NULL pointer");
  VIRNET_REQUIRE(n > 0 && h > 0 && w_ > 0 && cin > 0 && cout > 0, "virnet_conv_head_s4_wgrad: bad shape");
  const int oh = (h - 1) / 4 + 1, ow = (w_ - 1) / 4 + 1;
  hipLaunchKernelGGL(conv_head_s4_wgrad_kernel, dim3(cin * 81), dim3(256), 0, static_cast<hipStream_t>(stream), x, dy, dw, n, cin, h,
                     w_, cout, oh, ow);
  return virnet::check_launch("conv_head_s4_wgrad launch");
}

extern "C" int virnet_gap_nchw(const float* x, float* out, int n, int c, int h, int w, int finish, float lo, float hi,
                               void* stream) {
  VIRNET_REQUIRE(x && out, "virnet_gap_nchw: NULL pointer");
  VIRNET_REQUIRE(n > 0 && c > 0 && h > 0 && w > 0, "virnet_gap_nchw: bad shape");
  VIRNET_REQUIRE(finish >= VIRNET_GAP_MEAN && finish <= VIRNET_GAP_KINFO, "virnet_gap_nchw: finish=%d", finish);
  hipLaunchKernelGGL(gap_nchw_kernel, dim3(n * c), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, c, h * w, finish, lo,
                     hi);
  return virnet::check_launch("gap_nchw launch");
}

extern "C" int virnet_ca_gate(const float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* gate,
                              int n, int h, int w, int c, int cr, void* stream) {
  VIRNET_REQUIRE(x && w1 && b1 && w2 && b2 && gate, "virnet_ca_gate: NULL pointer");
  VIRNET_REQUIRE(c >= 1 && c <= 256 && 256 % c == 0, "virnet_ca_gate: c=%d must divide 256", c);
  VIRNET_REQUIRE(cr >= 1 && cr <= 64, "virnet_ca_gate: cr=%d", cr);
  hipLaunchKernelGGL(ca_gate_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), x, w1, b1, w2, b2, gate, h * w, c,
                     cr);
  return virnet::check_launch("ca_gate launch");
}

extern "C" int virnet_scale_add(const float* hcv, const float* gate, const float* skip, float* out, int n, int hw, int c,
                                void* stream) {
  VIRNET_REQUIRE(hcv && gate && skip && out, "virnet_scale_add: NULL pointer");
  VIRNET_REQUIRE(c % 4 == 0 && c > 0, "virnet_scale_add: c=%d must be a multiple of 4", c);
  const size_t total4 = (size_t)n * hw * (c / 4);
  const int grid = (int)((total4 + 255) / 256 > 8192 ? 8192 : (total4 + 255) / 256);
  hipLaunchKernelGGL(scale_add_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(hcv), gate, reinterpret_cast<const float4*>(skip),
                     reinterpret_cast<float4*>(out), hw, c / 4, total4);
  return virnet::check_launch("scale_add launch");
}

extern "C" int virnet_ca_scale_add(const float* hcv, const float* w1, const float* b1, const float* w2, const float* b2, const float* skip,
                                   float* out, int n, int h, int w, int c, int cr, void* stream) {
  VIRNET_REQUIRE(hcv && w1 && b1 && w2 && b2 && skip && out, "virnet_ca_scale_add: NULL pointer");
  VIRNET_REQUIRE(n > 0 && h > 0 && w > 0, "virnet_ca_scale_add: bad shape n=%d h=%d w=%d", n, h, w);
  VIRNET_REQUIRE(c >= 4 && c <= 256 && 256 % c == 0, "virnet_ca_scale_add: c=%d must be a divisor of 256 and a multiple of 4", c);
  VIRNET_REQUIRE(cr >= 1 && cr <= 64, "virnet_ca_scale_add: cr=%d", cr);
  const long n4 = (long)h * w * (c / 4);
  VIRNET_REQUIRE(n4 <= 16 * 1024, "virnet_ca_scale_add: an image of %ld float4 items does not fit one workgroup's registers (use virnet_ca_gate + virnet_scale_add)", n4);
  const int vv = (int)((n4 + 1023) / 1024);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const float4* hp = reinterpret_cast<const float4*>(hcv);
  const float4* sp = reinterpret_cast<const float4*>(skip);
  float4* op = reinterpret_cast<float4*>(out);
  if (vv <= 4) hipLaunchKernelGGL(ca_scale_add_kernel<4>, dim3(n), dim3(1024), 0, st, hp, w1, b1, w2, b2, sp, op, h * w, c, cr);
  else if (vv <= 8) hipLaunchKernelGGL(ca_scale_add_kernel<8>, dim3(n), dim3(1024), 0, st, hp, w1, b1, w2, b2, sp, op, h * w, c, cr);
  else hipLaunchKernelGGL(ca_scale_add_kernel<16>, dim3(n), dim3(1024), 0, st, hp, w1, b1, w2, b2, sp, op, h * w, c, cr);
  return virnet::check_launch("ca_scale_add launch");
}

extern "C" int virnet_sft_backward(const float* da, const float* x, const float* mul, const float* add, const float* res, float slope, float* dx,
                                   float* dmul, float* dadd, int n, long hw, int c, void* stream) {
  VIRNET_REQUIRE(da && x && mul && add && dx && dmul && dadd, "virnet_sft_backward: NULL pointer");
  VIRNET_REQUIRE(n > 0 && hw > 0 && c > 0 && c % 4 == 0 && c <= 1024, "virnet_sft_backward: bad shape n=%d hw=%ld c=%d (c %% 4 == 0, c <= 1024)", n, hw, c);
  VIRNET_REQUIRE(slope >= 0.f && slope <= 1.f, "virnet_sft_backward: slope=%g outside [0,1]", slope);
  const int chunk = 1024;
  hipLaunchKernelGGL(sft_backward_kernel, dim3((unsigned)((hw + chunk - 1) / chunk), n), dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(da), reinterpret_cast<const float4*>(x), mul, add, reinterpret_cast<const float4*>(res), slope,
                     reinterpret_cast<float4*>(dx), dmul, dadd, hw, c / 4, chunk);
  return virnet::check_launch("sft_backward launch");
}

extern "C" int virnet_sft_vec(const float* vec, const virnet_sft_weights* wt, float* mul, float* add, int n, void* stream) {
  VIRNET_REQUIRE(vec && mul && add && n > 0, "virnet_sft_vec: bad arguments");
  if (int rc = check_sft(wt, "virnet_sft_vec")) return rc;
  hipLaunchKernelGGL(sft_vec_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), vec, *wt, mul, add);
  return virnet::check_launch("sft_vec launch");
}

extern "C" int virnet_sft_vec_multi(const float* vec, const virnet_sft_weights* wts, int nlayers, float* const* muls, float* const* adds, int n,
                                    void* stream) {
  VIRNET_REQUIRE(vec && wts && muls && adds && n > 0, "virnet_sft_vec_multi: bad arguments");
  VIRNET_REQUIRE(nlayers >= 1 && nlayers <= SFT_MULTI_MAX, "virnet_sft_vec_multi: %d layers (1..%d per call)", nlayers, SFT_MULTI_MAX);
  SftMulti m{};
  for (int l = 0; l < nlayers; ++l) {
    if (int rc = check_sft(wts + l, "virnet_sft_vec_multi")) return rc;
    VIRNET_REQUIRE(wts[l].e == wts[0].e, "virnet_sft_vec_multi: layer %d takes %d conditioning channels, layer 0 takes %d", l, wts[l].e, wts[0].e);
    VIRNET_REQUIRE(muls[l] && adds[l], "virnet_sft_vec_multi: layer %d has a NULL output", l);
    m.wt[l] = wts[l]; m.mul[l] = muls[l]; m.add[l] = adds[l];
  }
  hipLaunchKernelGGL(sft_vec_multi_kernel, dim3(n, nlayers), dim3(256), 0, static_cast<hipStream_t>(stream), vec, m);
  return virnet::check_launch("sft_vec_multi launch");
}

extern "C" int virnet_sft_apply(const float* raw, const float* rec, const virnet_sft_weights* wt, float* act, int n, int h,
                                int w, int step, int chan0, void* stream) {
  VIRNET_REQUIRE(raw && rec && act && n > 0 && h > 0 && w > 0 && step >= 1, "virnet_sft_apply: bad arguments");
  if (int rc = check_sft(wt, "virnet_sft_apply")) return rc;
  VIRNET_REQUIRE(chan0 >= 0 && chan0 + wt->e <= 16, "virnet_sft_apply: channels [%d,%d) outside the 16-channel record", chan0,
                 chan0 + wt->e);
  const size_t npix = (size_t)n * h * w;
  hipLaunchKernelGGL(sft_apply_kernel, dim3((unsigned)((npix + SFT_PT - 1) / SFT_PT)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), raw, rec, *wt, act, n, h, w, step, chan0);
  return virnet::check_launch("sft_apply launch");
}

// ---- range guard, device side of a replayed graph: when the sticky flag is up, overwrite an output with NaN so that a forward whose
// split-fp16 operands left fp16's range can never be mistaken for a result (the exp(clamp) / tanh heads would hide the Inf otherwise).
namespace {
__global__ void poison_on_flag_kernel(const int* __restrict__ flag, float* __restrict__ y, size_t n) {
  if (*flag == 0) return;
  const float nan = __builtin_nanf("");
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = nan;
}
}  // namespace

extern "C" int virnet_poison_on_flag(const int* device_flag, float* y, size_t n, void* stream) {
  VIRNET_REQUIRE(device_flag && y, "virnet_poison_on_flag: NULL pointer");
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
  hipLaunchKernelGGL(poison_on_flag_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), device_flag, y, n);
  return virnet::check_launch("poison_on_flag launch");
}
