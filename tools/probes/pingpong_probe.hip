// Can the two waves of a SIMD overlap matrix-pipe and VALU work, and does phase-locking them help?
// Each "tile" of a wave = M phase (NM dependent-chain MFMAs over 4 accumulators) + V phase (NV VALU ops).
// Build: hipcc --offload-arch=gfx950 -O2 pingpong_probe.hip -o pingpong_probe      (timing: hipEvents, long runs)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#define VA(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k))
#define EX(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define M16 { MF(c0); MF(c1); MF(c2); MF(c3); MF(c0); MF(c1); MF(c2); MF(c3); MF(c0); MF(c1); MF(c2); MF(c3); MF(c0); MF(c1); MF(c2); MF(c3); }
// 64 fma + 32 exp  (about the VALU mix of a 64-key attention tile)
#define V8 { VA(x0); VA(x1); VA(x2); VA(x3); VA(x4); VA(x5); VA(x6); VA(x7); }
#define E8 { EX(x0); EX(x1); EX(x2); EX(x3); EX(x4); EX(x5); EX(x6); EX(x7); }
#define VPHASE { V8 E8 V8 V8 E8 V8 V8 E8 V8 V8 E8 V8 }
#define BAR __builtin_amdgcn_s_barrier()

// mode 0: M only   1: V only   2: M;V per wave, free running   3: M;V per wave, ping-pong barriers (group 1 offset by a phase)
// mode 4: role split (waves 0-3 M only, waves 4-7 V only)      5: fine interleave in one wave (4 MFMA, 24 VALU) x4
// mode 6: as 3 plus s_setprio raised in the M phase
__global__ void __launch_bounds__(512) probe(int mode, int iters, float* sink) {
  const int wave = threadIdx.x >> 6, grp = wave >> 2;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * e); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float x0 = 0.1f * threadIdx.x, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f, x4 = .5f, x5 = .6f, x6 = .7f, x7 = .8f, k = 0.999f;
  if (mode == 0) { for (int it = 0; it < iters; ++it) M16 }
  else if (mode == 1) { for (int it = 0; it < iters; ++it) VPHASE }
  else if (mode == 2) { for (int it = 0; it < iters; ++it) { M16 VPHASE } }
  else if (mode == 3 || mode == 6) {
    if (grp == 1) BAR;
    for (int it = 0; it < iters; ++it) {
      if (mode == 6) __builtin_amdgcn_s_setprio(2);
      M16
      if (mode == 6) __builtin_amdgcn_s_setprio(0);
      BAR;
      VPHASE
      BAR;
    }
    if (grp == 0) BAR;
  }
  else if (mode == 4) { if (grp == 0) { for (int it = 0; it < iters; ++it) M16 } else { for (int it = 0; it < iters; ++it) VPHASE } }
  else if (mode == 5) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int g = 0; g < 4; ++g) { MF(c0); MF(c1); MF(c2); MF(c3); V8 E8 V8 }
    }
  }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main() {
  float* s; hipMalloc(&s, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  struct { const char* name; int mode, threads; } cfg[] = {
    {"warm", 2, 512},
    {"M only, 1 wave/SIMD", 0, 256}, {"M only, 2 waves/SIMD", 0, 512},
    {"V only, 1 wave/SIMD", 1, 256}, {"V only, 2 waves/SIMD", 1, 512},
    {"M;V phased, 1 wave/SIMD", 2, 256}, {"M;V phased, 2 waves/SIMD free-running", 2, 512},
    {"M;V phased, 2 waves/SIMD ping-pong barriers", 3, 512}, {"  + setprio", 6, 512},
    {"role split: grp0 M only, grp1 V only", 4, 512},
    {"fine interleave, 1 wave/SIMD", 5, 256}, {"fine interleave, 2 waves/SIMD", 5, 512},
  };
  for (auto& c : cfg) {
    hipLaunchKernelGGL(probe, dim3(256), dim3(c.threads), 0, 0, c.mode, iters, s);     // untimed (clock ramp)
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe, dim3(256), dim3(c.threads), 0, 0, c.mode, iters, s);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-50s %8.3f ms   ns/tile/wave-slot = %7.1f\n", c.name, ms, ms * 1e6 / iters);
  }
  return 0;
}
