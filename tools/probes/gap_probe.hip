// How many instructions of which kind fit into the gap between two v_mfma_f32_32x32x16_bf16 of the SAME wave before the
// matrix pipe starts to idle -- with one wave per SIMD (256-thread workgroup) and with two (512 threads)?  Decides between the
// narrow (32 positions per wave, two waves per SIMD) and the wide (64 positions per wave, one wave per SIMD) tile forms:
// the wide form halves the LDS instructions per MFMA but loses the partner wave that fills dependency bubbles.
// Per iteration: 16 MFMAs; each is followed by the listed fillers (asm volatile: the program order is exactly this order).
//   dep = 1: chains of four MFMAs on one accumulator (like the S / dP chains), dep = 0: four accumulators round robin.
// Build: hipcc --offload-arch=gfx950 -O2 gap_probe.hip -o gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct Mix { int exp, valu, cvt, b128, tr; };      // fillers per MFMA gap (fractions: numerator over 4 gaps)

// fillers per group of FOUR MFMA gaps (so that fractional per-gap loads like 1.25 LDS reads can be expressed)
template <int EXP4, int VALU4, int CVT4, int B1284, int TR4, int DEP, int NT>
__global__ void __launch_bounds__(NT) probe(int iters, long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((int*)lds)[i] = i;
  __syncthreads();
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * e); }
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  float x[8] = {0.1f * lane, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
  const float k = 0.999f;
  u32x4 acc = {0, 0, 0, 0};
  u32x4 dv[4] = {{0}, {0}, {0}, {0}};
  u32x2 dt[4] = {{0}, {0}, {0}, {0}};
  const unsigned lp = (unsigned)(size_t)(lds + lane * 16 + (wave & 3) * 8192);
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {            // 4 groups of 4 MFMAs
      int ne = 0, nv = 0, nc = 0, nb = 0, ntr = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int ai = DEP ? g : m;
        __builtin_amdgcn_sched_barrier(0);      // program order = this order (the MFMA is a builtin: hipcc pads its hazards)
        c[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[ai], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // spread each kind evenly over the four gaps of the group
#pragma unroll
        for (; ne < (m + 1) * EXP4 / 4; ++ne) asm volatile("v_exp_f32 %0, %0" : "+v"(x[ne & 3]));
#pragma unroll
        for (; nv < (m + 1) * VALU4 / 4; ++nv) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[4 + (nv & 3)]) : "v"(k));
#pragma unroll
        for (; nc < (m + 1) * CVT4 / 4; ++nc) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(acc[nc & 3]) : "v"(x[4 + ((nc + 2) & 3)]), "v"(x[4 + ((nc + 3) & 3)]));
#pragma unroll
        for (; nb < (m + 1) * B1284 / 4; ++nb) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dv[nb & 3]) : "v"(lp), "i"((nb & 7) * 1024));
#pragma unroll
        for (; ntr < (m + 1) * TR4 / 4; ++ntr) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "+v"(dt[ntr & 3]) : "v"(lp), "i"((ntr & 7) * 1024));
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][5];
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += dv[i][0] + dv[i][3] + dt[i][0] + dt[i][1];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s + acc[0] + acc[1] + acc[2] + acc[3];
}

template <int EXP4, int VALU4, int CVT4, int B1284, int TR4, int DEP>
void run(const char* name, long long* d, float* s) {
  const int iters = 1000;
  long long h[8];
  double r[2][2];
  for (int two = 0; two < 2; ++two) {
    // 96 KiB of dynamic LDS: one workgroup per CU whatever its size
    if (two) {
      auto kern = probe<EXP4, VALU4, CVT4, B1284, TR4, DEP, 512>;
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
      hipLaunchKernelGGL(kern, dim3(256), dim3(512), 98304, 0, iters, d, s);
    } else {
      auto kern = probe<EXP4, VALU4, CVT4, B1284, TR4, DEP, 256>;
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
      hipLaunchKernelGGL(kern, dim3(256), dim3(256), 98304, 0, iters, d, s);
    }
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d + 8 * 100, sizeof(h), hipMemcpyDeviceToHost);
    r[two][0] = (double)h[0] / iters / 16;                 // ticks per MFMA, oldest wave
    r[two][1] = (double)h[two ? 7 : 3] / iters / 16;       // youngest wave (finishes last: the SIMD's time for both streams)
  }
  // SIMD time per MFMA: one wave = its own time; two waves = the later finisher's time / 2 (both streams done)
  printf("%-44s dep%d | 1 wave/SIMD: %5.1f | 2 waves/SIMD: old %5.1f young %5.1f -> %5.1f per MFMA of the pair\n", name, DEP, r[0][1], r[1][0], r[1][1],
         r[1][1] / 2);
}

int main() {
  long long* d; float* s;
  (void)hipMalloc(&d, 256 * 8 * sizeof(long long)); (void)hipMalloc(&s, 256 * 512 * 4);
  printf("ticks per MFMA (32 = matrix pipe saturated).  Fillers are listed per MFMA gap.\n");
  // single kinds, per gap
  run<0, 0, 0, 0, 0, 1>("nothing", d, s);
  run<0, 0, 0, 0, 0, 0>("nothing", d, s);
  run<4, 0, 0, 0, 0, 1>("1 exp", d, s);
  run<8, 0, 0, 0, 0, 1>("2 exp", d, s);
  run<0, 8, 0, 0, 0, 1>("2 mul", d, s);
  run<0, 16, 0, 0, 0, 1>("4 mul", d, s);
  run<0, 24, 0, 0, 0, 1>("6 mul", d, s);
  run<0, 0, 8, 0, 0, 1>("2 cvt_pk", d, s);
  run<0, 0, 0, 4, 0, 1>("1 ds_read_b128", d, s);
  run<0, 0, 0, 8, 0, 1>("2 ds_read_b128", d, s);
  run<0, 0, 0, 0, 4, 1>("1 ds_read_b64_tr", d, s);
  run<0, 0, 0, 0, 8, 1>("2 ds_read_b64_tr", d, s);
  // the dK/dV tile: per MFMA 1 exp, 1 mul, 1 cvt_pk (16 exp, 16 mul, 16 cvt per 16 MFMAs) ...
  run<4, 4, 4, 4, 4, 1>("dkv narrow: 1 exp 1 mul 1 cvt 1 b128 1 tr", d, s);        // ... 16 b128 + 16 tr per 16 MFMAs
  run<4, 4, 4, 4, 4, 0>("dkv narrow: 1 exp 1 mul 1 cvt 1 b128 1 tr", d, s);
  run<4, 4, 4, 3, 2, 1>("dkv wide:   1 exp 1 mul 1 cvt .75 b128 .5 tr", d, s);     // 24 b128 + 16 tr per 32 MFMAs
  run<4, 4, 4, 3, 2, 0>("dkv wide:   1 exp 1 mul 1 cvt .75 b128 .5 tr", d, s);
  run<4, 4, 4, 2, 2, 1>("dkv wide, seeds once: .5 b128 .5 tr", d, s);
  // the forward tile: per 10 MFMAs 16 exp, 8 cvt, 4 b128, 8 tr  ->  per 4 gaps: 6.4 exp, 3.2 cvt, 1.6 b128, 3.2 tr
  run<6, 0, 3, 2, 3, 1>("fwd narrow: 1.5 exp .75 cvt .5 b128 .75 tr", d, s);
  run<6, 0, 3, 1, 2, 1>("fwd wide:   1.5 exp .75 cvt .25 b128 .5 tr", d, s);
  // the dQ tile: per 12 MFMAs 16 exp, 16 mul, 8 cvt, 8 b128, 8 tr -> per 4 gaps: 5.3 exp, 5.3 mul, 2.7 cvt, 2.7 b128, 2.7 tr
  run<5, 5, 3, 3, 3, 1>("dq narrow:  1.3 exp 1.3 mul .7 cvt .7 b128 .7 tr", d, s);
  run<5, 5, 3, 2, 3, 1>("dq wide:    1.3 exp 1.3 mul .7 cvt .4 b128 .7 tr", d, s);
  return 0;
}
