// Which XCD does workgroup i of a 1-D launch land on?  (hardware register XCC_ID; gfx942+)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc & 0xf; out[2 * blockIdx.x + 1] = hwid; }
}
int main() {
  const int n = 64;
  int* d; hipMalloc(&d, 2 * n * sizeof(int));
  for (int threads : {64, 384}) {
    hipLaunchKernelGGL(probe, dim3(n), dim3(threads), 0, 0, d);
    int h[2 * n]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("threads %d  xcc:", threads);
    for (int i = 0; i < n; ++i) printf(" %d", h[2 * i]);
    printf("\n  cu_id:");
    for (int i = 0; i < n; ++i) printf(" %d", (h[2 * i + 1] >> 8) & 0xf);
    printf("\n  se_id:");
    for (int i = 0; i < n; ++i) printf(" %d", (h[2 * i + 1] >> 13) & 0x7);
    printf("\n");
  }
  return 0;
}
