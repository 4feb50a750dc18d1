// mfma_lds_probe.hip -- what LDS fragment returns cost the matrix pipe on gfx950, with the accumulators in VGPRs or in AGPRs.
// Two waves per SIMD, each: 9 x v_mfma_f32_32x32x16_f16 per iteration (three accumulators, triples) + NRD ds_read_b128 spread between them.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_lds_probe.hip -o tools/probes/bin/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int AG, int NRD>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32768];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = i * 0.001f;
  __syncthreads();
  f32x16 a0 = {}, a1 = {}, a2 = {};
  h8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
  const unsigned addr = (threadIdx.x & 63) * 16 + (threadIdx.x >> 6) * 1024;
  f32x4 sink = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    f32x4 r[9];
#define M(acc) do { if (AG == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(x), "v"(y)); \
                    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y)); } while (0)
#define R(i) do { if ((i) < NRD) { if (AG == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(r[i]) : "v"(addr), "n"(((i) % 8) * 4096 % 32768)); \
                                    else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(((i) % 8) * 4096 % 32768)); } } while (0)
    M(a0); R(0); M(a0); R(1); M(a0); R(2); M(a1); R(3); M(a1); R(4); M(a1); R(5); M(a2); R(6); M(a2); R(7); M(a2); R(8);
    asm volatile("s_waitcnt lgkmcnt(0)");
    if (AG == 2) { for (int i = 0; i < 9; ++i) if (i < NRD) asm volatile("" :: "a"(r[i])); }
    else { for (int i = 0; i < 9; ++i) if (i < NRD) asm volatile("" :: "v"(r[i])); }
  }
  float s = sink.x + sink.y + sink.z + sink.w;
  if (AG == 1) { asm volatile("s_nop 7\n s_nop 7"); f32x16 t0, t1, t2;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t0[0]) : "a"(a0[0])); (void)t1; (void)t2; s += t0[0]; }
  else for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int AG, int NRD>
void run(const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<AG, NRD><<<256, 512>>>(out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<AG, NRD><<<256, 512>>>(out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-12s ds_read_b128 per 9 MFMAs: %d   %.3f ms, %.1f ns per MFMA per SIMD\n", name, NRD, ms, ms * 1e6 / (iters * 9.0 * 2));
  (void)hipFree(out);
}

int main() {
  run<0, 0>("acc in VGPR"); run<1, 0>("acc in AGPR");
  run<0, 3>("acc in VGPR"); run<1, 3>("acc in AGPR");
  run<0, 6>("acc in VGPR"); run<1, 6>("acc in AGPR");
  run<0, 9>("acc in VGPR"); run<1, 9>("acc in AGPR");
  run<2, 3>("frags->AGPR"); run<2, 6>("frags->AGPR"); run<2, 9>("frags->AGPR");
  return 0;
}
