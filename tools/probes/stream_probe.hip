// Stream rates of this chip on an 805 MB tensor (the level-0 activation of the metric's step): read-only, write-only, copy.
// hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(256) void rd(const f4* __restrict__ x, size_t n4, float* out) {
  f4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + (U - 1) * 256 < n4; i += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(x + i + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1.f;
}
template <int U>
__global__ __launch_bounds__(256) void wr(f4* __restrict__ y, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  const f4 v = {1, 2, 3, 4};
  for (; i + (U - 1) * 256 < n4; i += stride)
#pragma unroll
    for (int u = 0; u < U; ++u) y[i + u * 256] = v;
}
template <int U>
__global__ __launch_bounds__(256) void cp(const f4* __restrict__ x, f4* __restrict__ y, size_t n4) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + (U - 1) * 256 < n4; i += stride) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) y[i + u * 256] = v[u];
  }
}
int main() {
  const size_t bytes = 32ull * 256 * 256 * 96 * 4, n4 = bytes / 16;
  f4 *x, *y; float* o;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&o, 4);
  hipMemset(x, 0x3c, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto fn, double gb) {
    std::vector<float> ts;
    for (int it = 0; it < 25; ++it) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 5) ts.push_back(ms); }
    std::sort(ts.begin(), ts.end());
    printf("%-40s median %.3f ms  %.2f TB/s\n", name, ts[ts.size() / 2], gb / ts[ts.size() / 2]);
  };
  for (int g : {256 * 4, 256 * 8, 256 * 16, 256 * 32}) {
    char nm[64];
    snprintf(nm, 64, "read  U=8 grid=%d", g); run(nm, [&] { rd<8><<<g, 256>>>(x, n4, o); }, bytes / 1e9);
    snprintf(nm, 64, "write U=4 grid=%d", g); run(nm, [&] { wr<4><<<g, 256>>>(y, n4); }, bytes / 1e9);
    snprintf(nm, 64, "copy  U=4 grid=%d", g); run(nm, [&] { cp<4><<<g, 256>>>(x, y, n4); }, 2 * bytes / 1e9);
  }
  return 0;
}
