// What does the partner wave's work cost an MFMA stream on the same SIMD?  512-thread workgroups, one per CU: waves 0-3 run
// back-to-back v_mfma_f32_32x32x16_bf16 (4 independent accumulators), waves 4-7 (the SIMD partners) run a stream of ONE
// kind of instruction.  Prints cycles per MFMA of the matrix waves and cycles per instruction of the partner stream.
// Build: hipcc --offload-arch=gfx950 -O2 coissue_probe.hip -o coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define MF(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0)

// kind: 0 nothing, 1 ds_read_b128, 2 ds_read_b64_tr_b16, 3 v_fma_f32, 4 v_exp_f32, 5 v_cvt_pk_bf16_f32, 6 MFMA too, 7 ds_read_b32, 8 s_nop-free SALU
template <int kind, int both>
__global__ void __launch_bounds__(512) probe(int iters, long long* out, float* sink) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((int*)lds)[i] = i;
  __syncthreads();
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * e); }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  float x0 = 0.1f * lane, x1 = 0.2f, x2 = 0.3f, x3 = 0.4f, k = 0.999f;
  u32x4 acc = {0, 0, 0, 0};
  unsigned sacc = 0;
  const bool matrix = wave < 4 || both >= 2;
  const bool partner = wave >= 4 || both == 3;       // both = 3: every wave runs both streams, 4 MFMAs then the 8 others
  const char* lp = lds + lane * 16 + (wave & 3) * 8192;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (matrix) { MF(c0); MF(c1); MF(c2); MF(c3); }
      if (partner) {
        if (kind == 1) {
#pragma unroll
          for (int r = 0; r < 8; ++r) { u32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)lp), "i"(r * 1024)); acc += v; }
        } else if (kind == 2) {
#pragma unroll
          for (int r = 0; r < 8; ++r) { u32x2 v; asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)lp), "i"(r * 1024)); acc[0] += v[0]; }
        } else if (kind == 7) {
#pragma unroll
          for (int r = 0; r < 8; ++r) { unsigned v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)lp), "i"(r * 1024)); acc[0] += v; }
        } else if (kind == 3) {
#pragma unroll
          for (int r = 0; r < 2; ++r) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x0) : "v"(k)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x1) : "v"(k));
                                        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x2) : "v"(k)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x3) : "v"(k)); }
        } else if (kind == 4) {
#pragma unroll
          for (int r = 0; r < 2; ++r) { asm volatile("v_exp_f32 %0, %0" : "+v"(x0)); asm volatile("v_exp_f32 %0, %0" : "+v"(x1)); asm volatile("v_exp_f32 %0, %0" : "+v"(x2)); asm volatile("v_exp_f32 %0, %0" : "+v"(x3)); }
        } else if (kind == 5) {
#pragma unroll
          for (int r = 0; r < 2; ++r) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x0) : "v"(x1), "v"(x2)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x3) : "v"(x1), "v"(x2));
                                        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x0) : "v"(x1), "v"(x2)); asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(x3) : "v"(x1), "v"(x2)); }
        } else if (kind == 6) { MF(c0); MF(c1); MF(c2); MF(c3); MF(c0); MF(c1); MF(c2); MF(c3);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  long long t1 = __builtin_readcyclecounter();
  if (lane == 0) out[blockIdx.x * 8 + wave] = t1 - t0;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + acc[0] + acc[1] + acc[2] + acc[3] + sacc;
}

template <int kind, int both> void run(const char* name, int iters, long long* d, float* s) {
  long long h[8];
  hipLaunchKernelGGL((probe<kind, both>), dim3(256), dim3(512), 0, 0, iters, d, s);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, d + 8 * 100, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-34s waves 0-3: %7.1f ticks / group (4 MFMA%s)      waves 4-7: %7.1f ticks / group\n", name, (double)h[0] / iters / 4,
         both == 3 ? " + 8 others" : "", (double)h[4] / iters / 4);
}
template <int both> void sweep(int iters, long long* d, float* s) {
  run<0, both>("nothing else", iters, d, s);
  run<1, both>("8 ds_read_b128", iters, d, s);
  run<2, both>("8 ds_read_b64_tr_b16", iters, d, s);
  run<7, both>("8 ds_read_b32", iters, d, s);
  run<3, both>("8 v_fma_f32", iters, d, s);
  run<4, both>("8 v_exp_f32", iters, d, s);
  run<5, both>("8 v_cvt_pk_bf16_f32", iters, d, s);
  run<6, both>("8 MFMA", iters, d, s);
}
int main() {
  long long* d; float* s;
  (void)hipMalloc(&d, 256 * 8 * sizeof(long long)); (void)hipMalloc(&s, 256 * 512 * 4);
  const int iters = 2000;
  printf("--- waves 0-3: 4 MFMA per group; waves 4-7 (their SIMD partners): the listed stream only.  Solo MFMA pipe time = ticks / 4\n");
  sweep<0>(iters, d, s);
  printf("--- every wave: 4 MFMA then the listed 8 instructions per group (two identical waves per SIMD)\n");
  sweep<3>(iters, d, s);
  return 0;
}
