// What would a ONE-kernel backward pay to pass its dQ partial tiles from key-tile workgroup to key-tile workgroup through L2?
// (round-4 review item 3 / DESIGN "Out of scope / next": the only backward form that had never been priced.)
//
// The form.  The dK/dV kernel (lanes = keys; a workgroup owns 256 keys of one (batch, head) and walks the 128-row query tiles at or
// below its diagonal) would also form dS^T . K for its keys: + 16 MFMAs per wave and query tile on top of the 64 it runs today, the dS
// tile transposed through the LDS (ds_write_b64 + ds_read_b64_tr_b16), and a [128 x 64] float32 partial of dQ per query tile -- 32 KiB,
// 4 KiB per wave -- that has to be summed over the key-tile workgroups of the head: C3 has 16 of them per head, eight chains of two in
// the causal pairing.  Without atomics (priced in round 2: 221 us) that is a systolic hand-off: workgroup c takes the running partial of
// query tile i from workgroup c + 1, adds its own and hands it to c - 1; the last one applies the l2norm backward and stores dq.  With
// the chain running from the HIGH key tiles to the low ones every workgroup can start at once (its diagonal tiles have no predecessor)
// and needs its predecessor to be two query tiles ahead of it in program order.
//
// What this probe measures (one workgroup per CU, 256 workgroups of 512 threads, chains of `CH` workgroups that share an XCD the way the
// kernels' block_to_work places a head's tiles: block % 8):
//   mode 0  compute only: `mfma` MFMAs per wave and step (v_mfma_f32_32x32x16_bf16, two accumulators per wave), `steps` steps
//   mode 1  + every workgroup WRITES its 32 KiB partial per step (16-byte write-through stores, drain, flag) -- nobody waits
//   mode 2  + the chain: poll the predecessor's flag for step s - LAG (one lane, relaxed agent-scope loads, s_sleep), read its 32 KiB with
//           sc1 loads (issued BEFORE the step's MFMAs, consumed after them), add, write-through, drain, flag
//   mode 3  mode 2 with the wait placed after the step's MFMAs (the predecessor gets a whole step of slack, the read is exposed)
// Flags are data-independent epochs (step + 1), zeroed before every launch; the payload is checked at the end of the chain (the sum of
// the chain's contributions), so a stale read fails the run instead of flattering it.
// Build: hipcc --offload-arch=gfx950 -O3 dq_handoff_probe.hip -o dq_handoff_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int SLOTS = 8;                 // ring of partial tiles per workgroup (the probe assumes the consumer lags < SLOTS - LAG steps)
constexpr int TILE_F = 128 * 64;         // floats per partial tile (32 KiB)
constexpr int LAG = 2;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

template <int MODE>
__global__ void __launch_bounds__(512) chain(float* slabs, unsigned* flags, float* out, int steps, int mfma, int CH, unsigned* timeouts) {
  const int id = blockIdx.x;
  const int xcd = id & 7, slot_id = id >> 3;                 // chain = CH consecutive slots of one XCD
  const int chain_id = (slot_id / CH) * 8 + xcd, c = slot_id % CH;      // position in the chain: CH - 1 = head (no predecessor), 0 = tail
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wg = chain_id * CH + c, pred = chain_id * CH + c + 1;
  float* my = slabs + (size_t)wg * SLOTS * TILE_F;
  const float* theirs = slabs + (size_t)pred * SLOTS * TILE_F;
  gu32* my_flag = (gu32*)(flags + wg);
  gu32* pred_flag = (gu32*)(flags + pred);
  const __amdgpu_buffer_rsrc_t wr = rsrc_of(my, SLOTS * TILE_F * 4), rd = rsrc_of(theirs, SLOTS * TILE_F * 4);

  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(0.001f * (lane + e)); b[e] = (__bf16)(0.002f * e); }
  f32x16 c0 = {0}, c1 = {0};
  const bool has_pred = MODE >= 2 && c + 1 < CH;
  // this wave's 4 KiB of the tile: 4 x 16 bytes per lane
  const unsigned lane_off = (unsigned)(wave * 1024 + lane * 4) * 4u;      // byte offset of the lane's first float4; pieces 1 KiB apart
  for (int s = 0; s < steps; ++s) {
    f32x4 in[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    const bool dep = has_pred && s >= LAG;
    const unsigned src_slot = (unsigned)((s - LAG) & (SLOTS - 1)) * TILE_F * 4u;
    auto wait_and_request = [&]() {
      if (wave == 0) {                                        // ONE wave polls ONE word, relaxed, agent scope
        unsigned spins = 0;
        while (__hip_atomic_load(pred_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(s - LAG + 1)) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(timeouts, 1u); break; }
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 4; ++i)                             // sc1 loads (aux 16): the producer stored write-through, no acquire fence needed
        in[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rd, src_slot + lane_off + (unsigned)i * 256u * 4u, 0, 16));
    };
    if (dep && MODE == 2) wait_and_request();
    for (int m = 0; m < mfma; m += 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
    }
    if (dep && MODE == 3) wait_and_request();
    if (MODE >= 1) {
      const unsigned dst_slot = (unsigned)(s & (SLOTS - 1)) * TILE_F * 4u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 v = in[i];
        v[0] += 1.0f; v[1] += 1.0f; v[2] += 1.0f; v[3] += 1.0f;      // "own contribution": the tail of a chain of CH must read CH (or LAG-clipped)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), wr, dst_slot + lane_off + (unsigned)i * 256u * 4u, 0, 16);
        if (c == 0 && s == steps - 1 && i == 0) out[(size_t)chain_id * 512 + threadIdx.x] = v[0];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // EVERY storing wave drains its write-through stores
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_store(my_flag, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (c0[3] + c1[5] == 12345.f) out[0] = c0[3];
}

int main(int argc, char** argv) {
  const int n_wg = 256, CH = 8;
  float *slabs, *out; unsigned *flags, *tmo;
  hipMalloc(&slabs, (size_t)(n_wg + 1) * SLOTS * TILE_F * 4);
  hipMalloc(&out, 32 * 512 * 4);
  hipMalloc(&flags, (n_wg + 8) * 4);
  hipMalloc(&tmo, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("dQ hand-off probe: %d workgroups x 512 threads, chains of %d on one XCD each, %d KiB partial per step and workgroup, lag %d steps\n", n_wg, CH, TILE_F * 4 / 1024, LAG);
  printf("%-8s %-6s %-6s %10s %12s %10s\n", "mode", "steps", "mfma", "us", "us/step", "check");
  const int step_list[] = {34};
  const int mfma_list[] = {96, 128, 160};      // per wave and step: the fused kernel would run 64 + 16 MFMAs plus its VALU work (~4.6 us per step)
  for (int steps : step_list)
    for (int mfma : mfma_list)
      for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f; float check = 0.f; unsigned to = 0;
        for (int rep = 0; rep < 6; ++rep) {
          hipMemsetAsync(flags, 0, (n_wg + 8) * 4, 0);
          hipMemsetAsync(tmo, 0, 4, 0);
          hipEventRecord(e0, 0);
          switch (mode) {
            case 0: hipLaunchKernelGGL(chain<0>, dim3(n_wg), dim3(512), 0, 0, slabs, flags, out, steps, mfma, CH, tmo); break;
            case 1: hipLaunchKernelGGL(chain<1>, dim3(n_wg), dim3(512), 0, 0, slabs, flags, out, steps, mfma, CH, tmo); break;
            case 2: hipLaunchKernelGGL(chain<2>, dim3(n_wg), dim3(512), 0, 0, slabs, flags, out, steps, mfma, CH, tmo); break;
            default: hipLaunchKernelGGL(chain<3>, dim3(n_wg), dim3(512), 0, 0, slabs, flags, out, steps, mfma, CH, tmo); break;
          }
          hipEventRecord(e1, 0);
          hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep > 0 && ms < best) best = ms;
        }
        std::vector<float> h(32 * 512);
        hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(&to, tmo, 4, hipMemcpyDeviceToHost);
        // tail of a chain at the last step: its own 1 + predecessor's value of step s - LAG (which holds 1 + ...): min(CH, 1 + (steps - 1) / LAG)
        float expect = mode >= 2 ? (float)((1 + (steps - 1) / LAG) < CH ? (1 + (steps - 1) / LAG) : CH) : 1.f;
        int bad = 0;
        if (mode >= 1) for (int ch = 0; ch < 32; ++ch) for (int t = 0; t < 512; ++t) bad += h[(size_t)ch * 512 + t] != expect;
        check = (float)bad;
        printf("%-8d %-6d %-6d %10.1f %12.3f %10s%s\n", mode, steps, mfma, best * 1e3, best * 1e3 / steps, mode == 0 ? "-" : (bad ? "STALE/WRONG" : "ok"), to ? "  (poll timeouts!)" : "");
        (void)check;
      }
  return 0;
}
