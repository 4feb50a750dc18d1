// f16_mfma_probe.hip -- hardware facts the split-fp16 convolution depends on (run on the GPU box, prints a small report):
//   1. does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (the low halves of small activations are subnormal)?
//   2. operand layout check (asymmetric data) of A[i][k], B[k][j], D[i][j]
//   3. accuracy of a K=864 dot product (one 96-channel 3x3 conv output) done as hi*hi + hi*lo + lo*hi on the f16 pipe
//      vs the same in fp32 on v_mfma_f32_32x32x2_f32, both against fp64.
// Build: hipcc --offload-arch=gfx950 -O2 -o f16_mfma_probe f16_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [32][K] row-major, B: [K][32] row-major, D: [32][32].  mode 0: f16x3 split, 1: fp32 MFMA, 2: f16 hi only, 3: f16x4
__global__ void probe(const float* A, const float* B, float* D, int K, int mode, float wscale) {
  const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (mode == 1) {
    for (int k = 0; k < K; k += 2) {
      const float a = A[l31 * K + k + lhi], b = B[(k + lhi) * 32 + l31];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < K; k += 16) {
      h8 ah, al, bh, bl;
      for (int e = 0; e < 8; ++e) {
        const float a = A[l31 * K + k + lhi * 8 + e] * wscale, b = B[(k + lhi * 8 + e) * 32 + l31];
        ah[e] = (_Float16)a; al[e] = (_Float16)(a - (float)ah[e]);
        bh[e] = (_Float16)b; bl[e] = (_Float16)(b - (float)bh[e]);
      }
      if (mode == 0 || mode == 3) {
        if (mode == 3) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
    D[row * 32 + l31] = acc[r] / (mode == 1 ? 1.f : wscale);
  }
}

static void run(const std::vector<float>& A, const std::vector<float>& B, int K, int mode, float ws, std::vector<float>& D) {
  float *dA, *dB, *dD;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, mode, ws);
  D.resize(1024);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  hipFree(dA); hipFree(dB); hipFree(dD);
}

static double urand() { return (double)rand() / RAND_MAX; }

int main() {
  // ---- 1. subnormal inputs
  {
    const int K = 16;
    std::vector<float> A(32 * K, 0.f), B(K * 32, 0.f), D;
    for (int i = 0; i < 32; ++i) A[i * K + 0] = ldexpf(1.f, -20);       // fp16 subnormal (min normal 2^-14)
    for (int j = 0; j < 32; ++j) B[0 * 32 + j] = 1.f;
    run(A, B, K, 2, 1.f, D);
    printf("subnormal A (2^-20) x 1.0 -> %.9g (expect %.9g): %s\n", D[0], ldexp(1.0, -20), D[0] == ldexpf(1.f, -20) ? "KEPT" : "FLUSHED");
    for (int i = 0; i < 32; ++i) A[i * K + 0] = 3.f;
    for (int j = 0; j < 32; ++j) B[0 * 32 + j] = ldexpf(1.f, -22) * 3.f;  // subnormal B
    run(A, B, K, 2, 1.f, D);
    printf("3.0 x subnormal B (3*2^-22) -> %.9g (expect %.9g): %s\n", D[0], 9.0 * ldexp(1.0, -22), D[0] == 9.f * ldexpf(1.f, -22) ? "KEPT" : "FLUSHED");
  }
  // ---- 2. layout (asymmetric integers, exact in fp16)
  {
    const int K = 32;
    std::vector<float> A(32 * K), B(K * 32), D;
    for (int i = 0; i < 32; ++i) for (int k = 0; k < K; ++k) A[i * K + k] = (float)((i * 7 + k * 3) % 11 - 5);
    for (int k = 0; k < K; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 13) % 9 - 4);
    for (int mode = 0; mode < 2; ++mode) {
      run(A, B, K, mode ? 1 : 2, 1.f, D);
      int bad = 0;
      for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * B[k * 32 + j];
        bad += D[i * 32 + j] != (float)s;
      }
      printf("layout check (%s): %d mismatches of 1024\n", mode ? "f32 32x32x2" : "f16 32x32x16", bad);
    }
  }
  // ---- 3. accuracy at K = 864 with conv-like operands: x in [0,1) mixed with small values, w ~ U(-b, b), b = sqrt(3/864)
  for (int trial = 0; trial < 2; ++trial) {
    const int K = 864;
    srand(1234 + trial);
    std::vector<float> A(32 * K), B(K * 32);
    const double b = sqrt(3.0 / K);
    for (auto& v : A) v = (float)((2 * urand() - 1) * b);
    for (auto& v : B) { double u = urand(); v = (float)(trial ? (u < 0.5 ? u * 1e-3 : (2 * u - 1) * 20.0) : u); }
    float amax = 0; for (auto v : A) amax = fmaxf(amax, fabsf(v));
    const float ws = ldexpf(1.f, (int)floor(log2(16384.0 / amax)));
    std::vector<double> ref(1024), mag(1024);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0, m = 0;
      for (int k = 0; k < K; ++k) { double p = (double)A[i * K + k] * B[k * 32 + j]; s += p; m += fabs(p); }
      ref[i * 32 + j] = s; mag[i * 32 + j] = m;
    }
    const char* names[4] = {"f16x3 (scaled w)", "f32 mfma", "f16 hi only", "f16x4 (scaled w)"};
    printf("trial %d (%s), weight scale 2^%d\n", trial, trial ? "x: half tiny (<1e-3), half U(-20,20)" : "x ~ U[0,1)", (int)log2f(ws));
    for (int mode : {1, 0, 3, 2}) {
      std::vector<float> D;
      run(A, B, K, mode, ws, D);
      double emax = 0, erms = 0, ebias = 0;
      for (int q = 0; q < 1024; ++q) {
        const double e = ((double)D[q] - ref[q]) / mag[q];      // relative to sum |a b|
        emax = fmax(emax, fabs(e)); erms += e * e; ebias += e;
      }
      printf("  %-18s err / sum|ab|: max %.3e  rms %.3e  mean %.3e\n", names[mode], emax, sqrt(erms / 1024), ebias / 1024);
    }
    if (trial == 0) {
      std::vector<float> D;
      run(A, B, K, 0, 1.f, D);
      double emax = 0;
      for (int q = 0; q < 1024; ++q) emax = fmax(emax, fabs(((double)D[q] - ref[q]) / mag[q]));
      printf("  %-18s err / sum|ab|: max %.3e\n", "f16x3 unscaled w", emax);
    }
  }
  return 0;
}
