#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cmath>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned cvt_pk(float a, float b) {   // RNE pack of two fp32 into fp16x2
  h2 r = {(_Float16)a, (_Float16)b};
  return __builtin_bit_cast(unsigned, r);
}
template <int HALF> __device__ __forceinline__ float sub_hi(float v, unsigned hpk) {   // v - float(half HALF of hpk), one VALU op
  float r;
  if (HALF == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
  else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
  return r;
}
__global__ void k(const f32x4* p, uint2* oh, uint2* ol) {
  f32x4 v = p[threadIdx.x];
  unsigned h01 = cvt_pk(v.x, v.y), h23 = cvt_pk(v.z, v.w);
  unsigned l01 = cvt_pk(sub_hi<0>(v.x, h01), sub_hi<1>(v.y, h01)), l23 = cvt_pk(sub_hi<0>(v.z, h23), sub_hi<1>(v.w, h23));
  oh[threadIdx.x] = make_uint2(h01, h23); ol[threadIdx.x] = make_uint2(l01, l23);
}
int main() {
  f32x4* p; uint2 *oh, *ol; hipMallocManaged(&p, 64 * 16); hipMallocManaged(&oh, 64 * 8); hipMallocManaged(&ol, 64 * 8);
  for (int i = 0; i < 64; ++i) p[i] = f32x4{1.0f + i * 0.0123457f, -3.14159f * i, 1e-3f * i + 7e-5f, 1234.567f + i};
  k<<<1, 64>>>(p, oh, ol); hipDeviceSynchronize();
  double worst = 0;
  for (int i = 0; i < 64; ++i) for (int e = 0; e < 4; ++e) {
    unsigned hw = e < 2 ? oh[i].x : oh[i].y, lw = e < 2 ? ol[i].x : ol[i].y;
    unsigned short hb = (e & 1) ? hw >> 16 : hw & 0xffff, lb = (e & 1) ? lw >> 16 : lw & 0xffff;
    _Float16 hh, ll; memcpy(&hh, &hb, 2); memcpy(&ll, &lb, 2);
    float v = p[i][e]; _Float16 rh = (_Float16)v; _Float16 rl = (_Float16)(v - (float)rh);
    if (hb != *(unsigned short*)&rh || lb != *(unsigned short*)&rl) { printf("MISMATCH %d %d\n", i, e); return 1; }
    double err = fabs((double)v - ((double)(float)hh + (double)(float)ll)); if (v != 0) worst = fmax(worst, err / fabs(v));
  }
  printf("fma_mix split matches the reference split bit for bit; worst relative residual %.3g\n", worst);
  return 0;
}
