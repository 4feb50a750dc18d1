// stage_probe.hip -- a synthetic replica of one conv_wx4_kernel stage (8 waves, 2 per SIMD, 27 MFMAs per wave between barriers) that
// adds the ingredients one at a time: which of them breaks the 64-cycle MFMA cadence of two waves sharing a SIMD?
//   bit 0: 4 independent VALU ops behind every MFMA          bit 1: fragment reads (24 ds_read_b128 per stage, one group ahead)
//   bit 2: s_barrier per stage                               bit 3: 5 LDS-DMA pieces per stage (1 KB each, from an L2-resident buffer)
//   bit 4: 6 ds_write_b64 per stage                          bit 5: MFMA operands come from the fragment reads (real dependence)
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/stage_probe.hip -o tools/probes/bin/stage_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F>
__global__ __launch_bounds__(512, 2) void k(float* out, const char* wsrc, int stages) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (i % 977) * 0.001f;
  __syncthreads();
  f32x16 a0 = {}, a1 = {}, a2 = {};
  h8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned addr = lane * 16 + (wave & 1) * 18432;
  float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
  const char* src = wsrc + lane * 16 + wave * 5120;
  for (int s = 0; s < stages; ++s) {
    h8 fa = x, fb = y, na, nb;
#pragma unroll
    for (int g = 0; g < 9; ++g) {
      if (F & 2) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(na) : "v"(addr), "n"((g * 2048) % 16384 + 65536 - 65536));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nb) : "v"(addr), "n"((g * 2048 + 1024) % 16384));
        if (g % 3 == 0) { h8 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(40960 + g * 512)); asm volatile("" :: "v"(t));
                          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(40960 + g * 512 + 256)); asm volatile("" :: "v"(t)); }
      }
      if ((F & 8) && g < 5) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + g * 1024 + (s & 7) * 65536),
                                         (__attribute__((address_space(3))) void*)(lds + 65536 + ((s & 1) * 40 + wave * 5 + g) * 1024), 16, 0, 0);
      }
#define M(acc, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
#define V4() do { if (F & 1) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)); } while (0)
      if (g % 3 == 0) { M(a0, fa, fb); V4(); M(a0, fb, fa); V4(); M(a0, fa, fa); V4(); }
      if (g % 3 == 1) { M(a1, fa, fb); V4(); M(a1, fb, fa); V4(); M(a1, fa, fa); V4(); }
      if (g % 3 == 2) { M(a2, fa, fb); V4(); M(a2, fb, fa); V4(); M(a2, fa, fa); V4(); }
      if ((F & 16) && g < 6) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v0), "n"(57344) : "memory");
      if (F & 2) {
        asm volatile("s_waitcnt lgkmcnt(0)");
        if (F & 32) { fa = na; fb = nb; } else { asm volatile("" :: "v"(na), "v"(nb)); }
      }
    }
    if (F & 4) { if (F & 8) asm volatile("s_waitcnt vmcnt(0)"); asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
  }
  float sres = v0 + v1 + v2 + v3;
  for (int r = 0; r < 16; ++r) sres += a0[r] + a1[r] + a2[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sres;
}

template <int F>
void run(const char* wsrc, const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  const int stages = 600;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<F>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<F><<<256, 512, 150 * 1024>>>(out, wsrc, stages);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<F><<<256, 512, 150 * 1024>>>(out, wsrc, stages);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-58s %.3f ms  %.0f ns per stage (54 MFMAs per SIMD; 994 ns = the bare pipe at this clock)\n", name, ms, ms * 1e6 / stages);
  (void)hipFree(out);
}

int main() {
  char* wsrc; (void)hipMalloc(&wsrc, 8 * 65536 + 65536); (void)hipMemset(wsrc, 1, 8 * 65536 + 65536);
  run<0>(wsrc, "MFMA only");
  run<1>(wsrc, "+ 4 VALU per MFMA");
  run<2>(wsrc, "+ fragment reads (not consumed)");
  run<2 | 32>(wsrc, "+ fragment reads feeding the MFMAs");
  run<1 | 2 | 32>(wsrc, "+ VALU + fragment reads feeding the MFMAs");
  run<1 | 2 | 32 | 4>(wsrc, "+ VALU + reads + barrier per stage");
  run<1 | 2 | 32 | 4 | 16>(wsrc, "+ VALU + reads + barrier + LDS writes");
  run<1 | 2 | 32 | 4 | 8>(wsrc, "+ VALU + reads + barrier + DMA");
  run<1 | 2 | 32 | 4 | 8 | 16>(wsrc, "+ everything");
  run<4 | 8>(wsrc, "MFMA + barrier + DMA");
  run<4>(wsrc, "MFMA + barrier");
  return 0;
}
