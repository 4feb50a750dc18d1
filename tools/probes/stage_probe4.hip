// stage_probe4.hip -- the synthetic stage of stage_probe.hip re-cut for ONE wave per SIMD (4 waves of 256 threads, 512 registers each):
// a wave owns two row blocks, so a stage is 54 MFMAs per wave, an A fragment pair feeds 6 MFMAs and a B pair 9 (30 fragment reads per
// stage instead of 2 x 24), with the same VALU / LDS-write / DMA work per SIMD as the 8-wave kernel.  Question: does one in-order
// wave per SIMD keep the matrix pipe as busy as two?
//   bit 0: 4 VALU per MFMA   bit 1: fragment reads, LA groups ahead   bit 2: barrier per stage   bit 3: 9 DMA pieces per wave and stage
//   bit 4: 12 ds_write per stage   bit 5: MFMA operands come from the reads
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/stage_probe4.hip -o tools/probes/bin/stage_probe4
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, int LA>
__global__ __launch_bounds__(256, 1) void k(float* out, const char* wsrc, int stages) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (i % 977) * 0.001f;
  __syncthreads();
  f32x16 acc[6] = {};
  h8 x, y;
  for (int e = 0; e < 8; ++e) { x[e] = (_Float16)(threadIdx.x * 0.001f + e); y[e] = (_Float16)(e * 0.5f - threadIdx.x * 0.002f); }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned addr = lane * 16 + (wave & 1) * 18432;
  float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wsrc), 0, 8 * 65536 + 65536, 0x00020000);
  for (int s = 0; s < stages; ++s) {
    // fragment ring: group g uses ring[g % (LA + 1)]; reads for group g + LA are issued in group g
    h8 fa[LA + 1], fb[LA + 1];
#pragma unroll
    for (int i = 0; i <= LA; ++i) { fa[i] = x; fb[i] = y; }
    auto rd = [&](int g) {
      if (!(F & 2)) return;
      const int i = g % (LA + 1);
      if (g % 2 == 0) {      // A pair every second group
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i]) : "v"(addr), "n"(0));
        h8 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(1024)); if (F & 32) fb[i] = t; else asm volatile("" :: "v"(t));
      }
      if (g % 3 == 0) {      // B pair every third group
        h8 t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(40960)); asm volatile("" :: "v"(t));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(addr), "n"(41216)); asm volatile("" :: "v"(t));
      }
    };
#pragma unroll
    for (int g = 0; g < LA; ++g) rd(g);
#pragma unroll
    for (int g = 0; g < 18; ++g) {
      if (g + LA < 18) rd(g + LA);
      if ((F & 8) && g < 9) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 65536 + ((s & 1) * 36 + wave * 9 + g) * 1024), 16, lane * 16,
                                                 (s & 7) * 65536 + wave * 9216 + g * 1024, 0, 0);
#endif
      }
      if (F & 2) {           // wait for this group's fragments only: everything issued for later groups stays in flight
        // reads in flight behind this group's: for groups g+1..g+LA (issued so far)
        int later = 0;
        for (int h = g + 1; h <= g + LA && h < 18; ++h) later += (h % 2 == 0 ? 2 : 0) + (h % 3 == 0 ? 2 : 0);
        if (later == 0) asm volatile("s_waitcnt lgkmcnt(0)");
        else if (later <= 2) asm volatile("s_waitcnt lgkmcnt(2)");
        else if (later <= 4) asm volatile("s_waitcnt lgkmcnt(4)");
        else if (later <= 6) asm volatile("s_waitcnt lgkmcnt(6)");
        else if (later <= 8) asm volatile("s_waitcnt lgkmcnt(8)");
        else asm volatile("s_waitcnt lgkmcnt(10)");
      }
      const int i = g % (LA + 1);
#define M(acc, A, B) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B))
#define V4() do { if (F & 1) asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3)); } while (0)
      M(acc[g % 6], fa[i], fb[i]); V4(); M(acc[g % 6], fb[i], fa[i]); V4(); M(acc[g % 6], fa[i], fa[i]); V4();
      if ((F & 16) && g < 12) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(addr), "v"(v0), "n"(57344) : "memory");
    }
    if (F & 4) { if (F & 8) asm volatile("s_waitcnt vmcnt(0)"); asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }
  }
  float sres = v0 + v1 + v2 + v3;
  for (int a = 0; a < 6; ++a) for (int r = 0; r < 16; ++r) sres += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sres;
}

template <int F, int LA>
void run(const char* wsrc, const char* name) {
  float* out; (void)hipMalloc(&out, 256 * 256 * 4);
  const int stages = 600;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<F, LA>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<F, LA><<<256, 256, 150 * 1024>>>(out, wsrc, stages);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<F, LA><<<256, 256, 150 * 1024>>>(out, wsrc, stages);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("4 waves, LA=%d: %-52s %.3f ms  %.0f ns per stage (54 MFMAs per SIMD)\n", LA, name, ms, ms * 1e6 / stages);
  (void)hipFree(out);
}

int main() {
  char* wsrc; (void)hipMalloc(&wsrc, 8 * 65536 + 65536); (void)hipMemset(wsrc, 1, 8 * 65536 + 65536);
  run<0, 1>(wsrc, "MFMA only");
  run<1, 1>(wsrc, "+ 4 VALU per MFMA");
  run<2 | 32, 1>(wsrc, "+ fragment reads feeding the MFMAs");
  run<2 | 32, 2>(wsrc, "+ fragment reads feeding the MFMAs");
  run<1 | 2 | 32, 2>(wsrc, "+ VALU + reads");
  run<1 | 2 | 32 | 4, 2>(wsrc, "+ VALU + reads + barrier");
  run<1 | 2 | 32 | 4 | 16, 2>(wsrc, "+ VALU + reads + barrier + LDS writes");
  run<1 | 2 | 32 | 4 | 8, 2>(wsrc, "+ VALU + reads + barrier + DMA");
  run<1 | 2 | 32 | 4 | 8 | 16, 1>(wsrc, "+ everything");
  run<1 | 2 | 32 | 4 | 8 | 16, 2>(wsrc, "+ everything");
  run<1 | 2 | 32 | 4 | 8 | 16, 3>(wsrc, "+ everything");
  return 0;
}
