// mfma_dep_probe.hip -- what a dependent chain of v_mfma_f32_32x32x16_f16 costs on gfx950 when two waves share a SIMD.
// Order A: three consecutive MFMAs into the same accumulator (a0 a0 a0 a1 a1 a1 a2 a2 a2); order B: rotating (a0 a1 a2 a0 a1 a2 ...).
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_dep_probe.hip -o tools/probes/bin/mfma_dep_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ORDER, int NVALU>
__global__ void k(float* out, int iters) {
  f32x16 a0 = {}, a1 = {}, a2 = {};
  h8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
  float v = threadIdx.x, v1 = v + 1, v2 = v + 2, v3 = v + 3;
  for (int it = 0; it < iters; ++it) {
#define W() for (int q = 0; q < NVALU / 12; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(v), "+v"(v1), "+v"(v2), "+v"(v3))
#define M(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc, 0, 0, 0)
#define V() for (int q = 0; q < NVALU; ++q) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v))
    if (ORDER == 0) { M(a0); M(a0); M(a0); V(); M(a1); M(a1); M(a1); V(); M(a2); M(a2); M(a2); V(); }
    else if (ORDER == 1) { M(a0); M(a1); M(a2); V(); M(a0); M(a1); M(a2); V(); M(a0); M(a1); M(a2); V(); }
    else if (ORDER == 2) { M(a0); W(); M(a0); W(); M(a0); W(); M(a1); W(); M(a1); W(); M(a1); W(); M(a2); W(); M(a2); W(); M(a2); W(); }
    else if (ORDER == 3) { M(a0); W(); M(a1); W(); M(a2); W(); M(a0); W(); M(a1); W(); M(a2); W(); M(a0); W(); M(a1); W(); M(a2); W(); }
    else { M(a0); M(a0); M(a0); W(); W(); W(); M(a1); M(a1); M(a1); W(); W(); W(); M(a2); M(a2); M(a2); W(); W(); W(); }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = v + v1 + v2 + v3;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int ORDER, int NVALU>
void run(int threads, const char* name) {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<ORDER, NVALU><<<256, threads>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<ORDER, NVALU><<<256, threads>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * 9 * (threads / 64) / 4;
  printf("%-28s threads %4d valu arg %2d: %.3f ms, %.1f ns per MFMA per SIMD (32 cycles @2.4GHz = 13.3 ns)\n", name, threads, NVALU, ms, ms * 1e6 / mfma_per_simd);
  hipFree(out);
}

int main() {
  run<0, 0>(256, "same-acc triples"); run<1, 0>(256, "rotating accs");
  run<0, 0>(512, "same-acc triples"); run<1, 0>(512, "rotating accs");
  run<0, 12>(512, "same-acc triples"); run<1, 12>(512, "rotating accs");
  run<0, 24>(512, "same-acc triples"); run<1, 24>(512, "rotating accs");
  // independent VALU (4 per MFMA at NVALU = 12 ... 12 per MFMA at 36): between every MFMA (same acc / rotating) or after each triple
  run<2, 12>(512, "same acc, valu between"); run<3, 12>(512, "rotating, valu between"); run<4, 12>(512, "triples, valu after");
  run<2, 24>(512, "same acc, valu between"); run<3, 24>(512, "rotating, valu between"); run<4, 24>(512, "triples, valu after");
  run<2, 36>(512, "same acc, valu between"); run<3, 36>(512, "rotating, valu between"); run<4, 36>(512, "triples, valu after");
  run<2, 24>(256, "same acc, valu between"); run<3, 24>(256, "rotating, valu between"); run<4, 24>(256, "triples, valu after");
  return 0;
}
