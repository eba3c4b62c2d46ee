// How expensive is pushing dQ partial tiles through global f32 atomics?  (SURVEY §7.2 option (b): one key-parallel
// backward kernel, dQ reduced with global_atomic_add_f32 -- the reference does that at cu:1610.)
//
// Pattern of the planned kernel at C3 (B4 H8 N4096 D64 causal): per (batch, head) a [4096 x 64] f32 dQ slab (1 MiB);
// every (256-key, 128-query) tile pair adds a [128 x 64] f32 tile = 32 KiB; 272 pairs per (b, h) -> 285 MB of atomic
// operands per launch, every address hit 8.5 times on average.  A wave instruction covers 2 rows x 32 features
// (2 x 128 contiguous bytes); a wave issues 16 of them per tile.  Blocks of one (b, h) sit on one XCD (block % 8).
//
// Variants: 0 plain stores (write-bandwidth reference), 1 agent-scope atomicAdd (what a correct kernel must use),
// 2 workgroup-scope atomic (executes in the XCD's own L2; only informative: not coherent across XCDs),
// 3 agent-scope atomics with 64 ALU-only "tiles" of MFMA work in between (does it hide?).
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics atomic_probe.hip -o atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512) push(float* dq, int n_bh, int q_tiles, int k_tiles, int mfma_per_tile, float* sink, int spread) {
  // block -> (bh, key tile): all key tiles of one bh on one XCD (spread = 1: consecutive blocks, i.e. one bh over all XCDs)
  const int id = blockIdx.x;
  const int xcd = id & 7, slot = id >> 3;
  const int bh = spread ? id / k_tiles : (slot / k_tiles) * 8 + xcd, kt = spread ? id % k_tiles : slot % k_tiles;
  if (bh >= n_bh) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* slab = dq + (size_t)bh * q_tiles * 128 * 64;
  // causal: key tile kt (256 keys) sees query tiles >= 2 * kt
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * e); }
  f32x16 c0 = {0}, c1 = {0};
  for (int qt = 2 * kt; qt < q_tiles; ++qt) {
    for (int m = 0; m < mfma_per_tile; m += 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
    // wave w owns rows 32*(w&3) .. +31, features 32*(w>>2) .. +31 of the [128 x 64] tile
    float* base = slab + ((size_t)qt * 128 + 32 * (wave & 3)) * 64 + 32 * (wave >> 2) + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* p = base + (size_t)(2 * r + (lane >> 5)) * 64;
      const float val = c0[r] + 1.0f;
      if (MODE == 3) { if (val == 54321.f) *p = val; }
      else if (MODE == 0) *p = val;
      else if (MODE == 2) __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(p, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (c1[3] == 12345.f) sink[0] = c1[3];
}

int main() {
  const int n_bh = 32, q_tiles = 32, k_tiles = 16;
  const size_t elems = (size_t)n_bh * q_tiles * 128 * 64;
  float *dq, *sink;
  hipMalloc(&dq, elems * 4);
  hipMalloc(&sink, 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  size_t pairs = 0;
  for (int kt = 0; kt < k_tiles; ++kt) pairs += q_tiles - 2 * kt;
  const double bytes = (double)pairs * n_bh * 128 * 64 * 4;
  printf("atomic operand bytes per launch: %.1f MB (slab %.1f MB)\n", bytes / 1e6, elems * 4 / 1e6);
  struct { const char* name; int mode, mfma, spread; } cfg[] = {
    {"plain stores, no compute", 0, 0, 0}, {"agent atomics, no compute", 1, 0, 0}, {"workgroup-scope atomics, no compute", 2, 0, 0},
    {"agent atomics, no compute, bh spread over XCDs", 1, 0, 1},
    {"compute only (80 MFMA / wave / tile)", 9, 80, 0}, {"agent atomics + 80 MFMA / wave / tile", 1, 80, 0},
    {"agent atomics + 80 MFMA, bh spread over XCDs", 1, 80, 1},
    {"workgroup atomics + 80 MFMA / wave / tile", 2, 80, 0}, {"plain stores + 80 MFMA / wave / tile", 0, 80, 0},
  };
  for (auto& c : cfg) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemsetAsync(dq, 0, elems * 4, 0);
      hipEventRecord(e0, 0);
      const dim3 grid(n_bh * k_tiles), blk(512);
      if (c.mode == 0) hipLaunchKernelGGL(push<0>, grid, blk, 0, 0, dq, n_bh, q_tiles, k_tiles, c.mfma, sink, c.spread);
      else if (c.mode == 1) hipLaunchKernelGGL(push<1>, grid, blk, 0, 0, dq, n_bh, q_tiles, k_tiles, c.mfma, sink, c.spread);
      else if (c.mode == 2) hipLaunchKernelGGL(push<2>, grid, blk, 0, 0, dq, n_bh, q_tiles, k_tiles, c.mfma, sink, c.spread);
      else hipLaunchKernelGGL(push<3>, grid, blk, 0, 0, dq, n_bh, q_tiles, k_tiles, c.mfma, sink, c.spread);   // MODE 3: no memory op
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-44s %8.1f us   %7.2f TB/s of operands\n", c.name, best * 1e3, c.mode == 9 ? 0.0 : bytes / (best * 1e-3) / 1e12);
  }
  // correctness of the agent-scope form: every element must equal its hit count
  for (int spread = 0; spread < 2; ++spread) {
    hipMemset(dq, 0, elems * 4);
    hipLaunchKernelGGL(push<1>, dim3(n_bh * k_tiles), dim3(512), 0, 0, dq, n_bh, q_tiles, k_tiles, 0, sink, spread);
    std::vector<float> h(elems);
    hipMemcpy(h.data(), dq, elems * 4, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int bh = 0; bh < n_bh; ++bh)
      for (int qt = 0; qt < q_tiles; ++qt) {
        const float want = (float)(qt / 2 + 1);
        for (int e = 0; e < 128 * 64; ++e) bad += h[((size_t)bh * q_tiles + qt) * 128 * 64 + e] != want;
      }
    printf("agent-scope result check (spread=%d): %zu wrong elements of %zu\n", spread, bad, elems);
  }
  return 0;
}
