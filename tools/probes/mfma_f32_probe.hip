// Probe of v_mfma_f32_32x32x2_f32 operand/result lane layout on gfx950.
// Build: hipcc --offload-arch=gfx950 -O2 mfma_f32_probe.hip -o mfma_f32_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// C[i][j] = sum_k A[i][k] * B[k][j], K = 8 (4 instructions), assumed layout:
//   lane l supplies A[l&31][2*t + (l>>5)] and B[2*t + (l>>5)][l&31] for instruction t; result reg r of lane l is
//   C[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
__global__ void probe(const float* A, const float* B, float* C, float* raw) {
  const int l = threadIdx.x, x = l & 31, hi = l >> 5;
  f32x16 c;
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  for (int t = 0; t < 4; ++t) {
    const int k = 2 * t + hi;
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(A[x * 8 + k], B[k * 32 + x], c, 0, 0, 0);
  }
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    C[row * 32 + x] = c[r];
    raw[l * 16 + r] = c[r];
  }
}

int main() {
  float hA[32 * 8], hB[8 * 32], hC[32 * 32], hraw[64 * 16], ref[32 * 32];
  for (int i = 0; i < 32; ++i) for (int k = 0; k < 8; ++k) hA[i * 8 + k] = (float)((i * 7 + k * 3) % 11) - 5.f;
  for (int k = 0; k < 8; ++k) for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (float)((j * 5 + k * 13) % 17) - 8.f;
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { float s = 0; for (int k = 0; k < 8; ++k) s += hA[i * 8 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
  float *dA, *dB, *dC, *draw;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC)); hipMalloc(&draw, sizeof(hraw));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, draw);
  hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost); hipMemcpy(hraw, draw, sizeof(hraw), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 1024; ++i) if (fabsf(hC[i] - ref[i]) > 1e-3f) ++bad;
  printf("mfma_f32_32x32x2 assumed-layout mismatches: %d / 1024\n", bad);
  if (bad) {
    // is it the transpose?
    int badT = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) if (fabsf(hC[i * 32 + j] - ref[j * 32 + i]) > 1e-3f) ++badT;
    printf("  vs transpose: %d\n", badT);
    for (int l = 0; l < 64; l += 21) { printf("  lane %2d raw:", l); for (int r = 0; r < 16; ++r) printf(" %g", hraw[l * 16 + r]); printf("\n"); }
    printf("  ref row0:"); for (int j = 0; j < 8; ++j) printf(" %g", ref[j]); printf("\n");
  }
  return 0;
}
