// Does the matrix pipe overlap with VALU / LDS work (a) inside one wave, (b) across two waves of one SIMD?
// Build: hipcc --offload-arch=gfx950 -O2 overlap_probe.hip -o overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)
#define VA(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k))
#define EX(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))

// mode bit0: MFMA work, bit1: VALU fma work, bit2: exp work, bit3: LDS b128 reads.  wsel: which waves do which:
//  split=0: every wave does everything (interleaved in program order); split=1: even waves MFMA only, odd waves the rest
__global__ void probe(int mode, int split, int iters, long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[16384];
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) ((int*)lds)[i] = i;
  __syncthreads();
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (threadIdx.x + e)); b[e] = (__bf16)(0.002f * e); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float x0 = 0.1f * threadIdx.x, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f, x4 = .5f, x5 = .6f, x6 = .7f, x7 = .8f, k = 0.999f;
  u32x4 acc = {0, 0, 0, 0};
  const bool do_m = (mode & 1) && (!split || (wave & 1) == 0);
  const bool do_v = (mode & 2) && (!split || (wave & 1) == 1);
  const bool do_e = (mode & 4) && (!split || (wave & 1) == 1);
  const bool do_l = (mode & 8) && (!split || (wave & 1) == 1);
  const char* lp = lds + (threadIdx.x & 63) * 16;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    // one "tile": 4 groups of {1..4 MFMA, 8 fma, 8 exp, 2 lds}
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (do_m) { MF(c0); MF(c1); MF(c2); MF(c3); }
      if (do_v) { VA(x0); VA(x1); VA(x2); VA(x3); VA(x4); VA(x5); VA(x6); VA(x7); VA(x0); VA(x1); VA(x2); VA(x3); VA(x4); VA(x5); VA(x6); VA(x7); }
      if (do_e) { EX(x0); EX(x1); EX(x2); EX(x3); EX(x4); EX(x5); EX(x6); EX(x7); }
      if (do_l) { acc += *(const u32x4*)(lp + g * 1024); acc += *(const u32x4*)(lp + g * 1024 + 4096); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0 || threadIdx.x == 64) out[blockIdx.x * 2 + (threadIdx.x >> 6)] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + acc[0];
}

int main() {
  long long* d; float* s; hipMalloc(&d, 4096 * sizeof(long long)); hipMalloc(&s, 1 << 22);
  long long h[8];
  const int iters = 2000;
  struct { const char* name; int mode, split, threads; } cfg[] = {
    {"1 wave/SIMD: MFMA only (16/iter)", 1, 0, 256}, {"1 wave/SIMD: fma only (64/iter)", 2, 0, 256}, {"1 wave/SIMD: exp only (32/iter)", 4, 0, 256},
    {"1 wave/SIMD: lds only (8 b128/iter)", 8, 0, 256}, {"1 wave/SIMD: MFMA+fma same wave", 3, 0, 256}, {"1 wave/SIMD: MFMA+fma+exp same wave", 7, 0, 256},
    {"1 wave/SIMD: all four same wave", 15, 0, 256},
    {"2 waves/SIMD: both MFMA only", 1, 0, 512}, {"2 waves/SIMD: both fma only", 2, 0, 512},
    {"2 waves/SIMD: both do MFMA+fma+exp", 7, 0, 512}, {"2 waves/SIMD: both do all four", 15, 0, 512},
  };
  for (auto& c : cfg) {
    hipLaunchKernelGGL(probe, dim3(256), dim3(c.threads), 0, 0, c.mode, c.split, iters, d, s);
    hipDeviceSynchronize();
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-44s cycles/iter = %8.1f\n", c.name, (double)h[0] / iters);
  }
  // split roles: with 512 threads waves 0..7 -> SIMD = wave % 4 presumably: waves w and w+4 share a SIMD; make wave parity differ within a SIMD
  return 0;
}
