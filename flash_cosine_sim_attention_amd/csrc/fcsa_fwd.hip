// fcsa_fwd.hip -- forward kernel of fused cosine-similarity attention for gfx950 (f16 / bf16 / f32).
//
// Replaces forward_kernel (reference cu:1072-1247).  Math (cu:1204-1246):
//     S = scale * Qh Kh^T (+ bias);  P~ = valid ? exp(S - shift) : 0;  l = rowsum(P~);
//     O = (P~ V) / max(l, eps);      inv_l = 1 / max(l, eps)
// with NO running max / rescale (logits are bounded because q, k are l2-normalised).
//
// Decomposition (one workgroup = NW waves = 32*NW query rows; K/V tiles of BN = 64 keys):
//   * each wave keeps its 32 query rows' Q fragments in VGPRs for the whole key loop
//     (the reference re-reads the Q tile from global for every column tile, cu:1185-1189);
//   * K and V tiles are staged global -> VGPR -> LDS (buffer loads), double buffered, one barrier per tile placed in
//     the MIDDLE of the tile (after its last LDS read) so the next tile's K fragments are requested before the PV
//     products (see the pipeline comment in fwd_kernel);
//   * S^T = K Q^T on v_mfma_f32_32x32x16 (A = K rows via ds_read_b128, B = Q registers), so a
//     lane owns ONE query (column) and 16 keys (rows) of each 32x32 block;
//   * exp2 + masking stay in registers; P~ is packed to 16 bit in place and becomes the
//     B operand of O^T = V^T P~^T, whose A operand V^T comes from the row-major LDS V tile through
//     ds_read_b64_tr_b16 (no P~ round trip through shared memory, cf. cu:1222);
//   * 16-bit types: the row sum is the exact f32 sum of the ROUNDED P~ that feeds P~V, so O is a true convex combination of
//     V rows (the reference also sums the rounded tile, cu:1236): v_dot2c against packed ones, 8 per block and lane, one
//     lane^32 add at the end (round 4; rounds 1 - 3 spent two all-ones MFMAs per block on it, kernel comment below);
//     f32: per-lane adds plus one lane^32 add;
//   * float32 inputs run the same skeleton on v_mfma_f32_32x32x2_f32 (exact f32, 1/16 of the bf16
//     rate): P~ stays f32 and V^T comes from ds_read_b32 (fcsa_common.cuh), no transposed read;
//   * the key loop is split in two SEQUENTIAL loops, first the tiles that need no masking, then the
//     tiles that do (key mask / tail / causal diagonal).  Each loop has one straight-line body, so the
//     accumulators never cross an if/else join (which costs dozens of register copies per tile).  The masked
//     loop has two forms, fixed per kernel instantiation (KM): causal launches select per logit on their diagonal
//     tiles, the others take a key mask / ragged tail as a rank-1 MFMA per block (key_mask_rank1, fcsa_common.cuh).
//
// Variants chosen by launch_forward (all parity-tested through the normal dispatch):
//   fwd_kernel<.., NW = 4 | 8, ..>  two 128-row workgroups per CU, or one 256-row workgroup when the grid still covers the chip
//   fwd_kernel<.., DYN>             per-row exponent reference for logit ranges no constant shift can hold (kept online, one pass)
//   fwd_kernel, gridDim.y = splits  key range split over several workgroups + fwd_combine_kernel (grids that cannot fill the chip)
//   fwd_kernel<.., LEAN>            16-bit D = 96 / 128 on chip-covering grids: no cross-block prefetch, 256 registers, two waves per SIMD
//   fwd_kernel<.., KM>              non-causal launches: no causal pairing / diagonal logic, masked tiles in the rank-1 form
//   fwd_kernel<.., KSPLIT>          128-row workgroups of 8 waves whose halves split the keys (grids of <= one workgroup per CU; 16-bit D = 96 / 128
//                                   whenever the 256-row lean form cannot cover the chip)
//   fwd2_kernel                     64 rows per wave, slot-scheduled rotating pipeline (D <= 64, 16 bit, no bias, no key mask; see its header)
#include <cstdlib>
#include <type_traits>

#include "fcsa_common.cuh"
#include "fcsa_kernels.h"
#ifdef FCSA_VAR_SPLIT_ENV
#include "dev/fcsa_sweep_env.h"
#endif

namespace fcsa {
// 64-key tiles per LDS stage of fwd_kernel where the stages arrive by LDS-DMA.  One: this kernel's barrier sits in the MIDDLE of a
// tile (mid()), where the old wave of a SIMD waits less than at a tile end; two per stage measured +1.8 % time at C3.
constexpr int kFwdSub = 1;
// Row sums of the 16-bit forward forms.  Rounds 1 - 3 took them from the MATRIX pipe (an all-ones MFMA per 16-key step: exact f32 sums of
// the rounded P~, 2 of a block's 10 MFMAs at D = 64); round 4 takes them from the VALU with v_dot2c_f32_{bf16,f16} against packed ones
// (the same exact sums: 8 instructions per block and lane, one lane^32 add at the end).  Measured again because the kernels run at the
// chip's power limit, where a fifth less MFMA work weighs more than eight VALU slots: C3 forward 83.3 -> 77.7 us (-6.7 %), lean D = 128
// 136.9 -> 129.9, bias form 66.8 -> 63.2, online-reference form 93.8 -> 90.4 (profiles/r04_ab_rowsum_valu.txt).  (Round 2 measured
// the same swap at +1 % -- before LDS-DMA staging, with the MFMA group hints still counting the two row-sum MFMAs.)
// All 16-bit forms (prefetching, lean, generic / bias) use the VALU sums; f32 keeps its per-lane adds.
#ifdef FCSA_TRACE
__device__ unsigned long long g_trace_fwd[128];
#endif
#ifdef FCSA_TRACE_BAR      // see fcsa_bwd.hip
__device__ unsigned long long g_trace_bar_fwd[64];
#define FCSA_BAR_BEGIN(v) do { asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v)); } while (0)
#define FCSA_BAR_END(v, acc) do { unsigned long long e_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(e_)); acc += e_ - v; } while (0)
#else
#define FCSA_BAR_BEGIN(v) ((void)0)
#define FCSA_BAR_END(v, acc) ((void)0)
#endif
#ifdef FCSA_TRACE_WG
__device__ unsigned long long g_trace_wg_fwd[2048];      // per workgroup: [2 * id] = start time, [2 * id + 1] = end time (wave 0)
__device__ unsigned long long g_trace_pass_fwd[2560];     // per workgroup (first 256): [pass][5] pass marks of wave 0
#endif

// "Lean" form of the 32-rows-per-wave kernel for 16-bit rows of 129 .. 256 bytes (D = 96, 128) without bias: nothing is prefetched
// across blocks -- K row fragments and V transposed fragments are requested per 32-key block, next to their MFMAs -- so the wave
// fits 256 registers and TWO waves share a SIMD (eight waves per CU), the partner hiding the LDS latency the prefetches hid.
// The round-2 form prefetched a whole tile's K fragments and ran one wave per SIMD at these widths (397 registers at D = 128);
// measured on MI355X (C3 at D = 128, profiles/r03_*): see DESIGN.md section 6.
// It only pays when two waves per SIMD are actually resident -- an 8-wave workgroup per CU, or two 4-wave workgroups -- so it is a
// kernel template parameter chosen at launch (launch_fwd_b); small grids keep the prefetching one-wave form.
template <typename T, int D, bool BIAS> constexpr bool fwd_lean() {
  return Traits<T>::ES == 2 && !BIAS && D * Traits<T>::ES > 128 && D * Traits<T>::ES <= 256;
}

// Per-row exponent reference of the DYN kernels, kept ONLINE (one pass over the keys).  `x` holds one block's exponents of this lane's
// row (log2 units, relative to the row's current reference `mref`, masked positions at -inf).  The block is used as it is while its
// largest exponent stays inside what the rounded P~ can hold (2^kTau); otherwise -- and at the row's first block with a valid key,
// whatever its level -- the reference moves to the block's max: x -= d, and what the row has accumulated so far is rescaled by
// 2^-d (per LANE: the C layout gives every lane its own row, so there is no broadcast).  The test is one wave-uniform branch per
// block that is taken a handful of times per row tile; rows end with their max inside [0, kTau] of the reference, i.e. no weight
// that matters is lost to the 16-bit exponent range whatever scale, groups or bias are.  (Rounds 1 - 2: a first pass over all K
// tiles computed the row max -- +70 % forward time; the bias form read the bias twice.)
template <typename T> constexpr float online_tau() { return std::is_same<T, F16>::value ? 10.f : 64.f; }

FCSA_DEV float max16(const f32x16& x) {
  float m = fmaxf(fmaxf(x[0], x[1]), x[2]);
#pragma unroll
  for (int r = 3; r < 15; r += 2) m = fmaxf(fmaxf(m, x[r]), x[r + 1]);      // (v_max3_f32)
  return fmaxf(m, x[15]);
}

// bm: largest exponent of the block(s) about to be used, this lane's half of the row.  `relog(d)` re-expresses the logits that were
// accumulated against the old reference (x -= d) and whatever was derived from them.
template <typename T, int DB, typename Relog>
FCSA_DEV void online_recentre(float bm, float& mref, float& rmax, f32x16 (&o)[DB], float& l, f32x16& lacc, Relog&& relog) {
  {   // the lane pair of a row: v_permlane32_swap (VALU) -- __shfl_xor is a ds_bpermute round trip in front of the branch
    const uint32_t bits = __builtin_bit_cast(uint32_t, bm);
    const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
    bm = fmaxf(__builtin_bit_cast(float, (uint32_t)sw[0]), __builtin_bit_cast(float, (uint32_t)sw[1]));
  }
  const bool first = rmax == -INFINITY && bm > -INFINITY;
  const bool move = first || bm > online_tau<T>();
  if (__builtin_amdgcn_ballot_w64(move) != 0) {
    const float d = move ? bm : 0.f;
    // nothing is accumulated before the first valid key (and d may be far below zero there: 2^-d would overflow)
    const float f = first ? 1.f : fast_exp2(-d);
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] *= f;
    l *= f;
#pragma unroll
    for (int r = 0; r < 16; ++r) lacc[r] *= f;
    mref += d;
    rmax = first ? 0.f : rmax - d;
    bm -= d;
    relog(d);
  }
  rmax = fmaxf(rmax, bm);
}

// exp2 / mask / pack of one 32x32 block of logits (in place): s -> P~ (f32), pb = packed operand, l / lacc updated
template <typename T, int DB, int MODE, bool BIAS, bool ONL, typename Other>
FCSA_DEV void fwd_softmax_block(f32x16& s, SecondB<T>& pb, float& l, f32x16& lacc, const FwdParams& p, uint32_t w,
                                int jbase, const char* bias_row, float& mref, float& rmax, f32x16 (&o)[DB], Other&& other,
                                u32x4 (*braw)[2] = nullptr, bool use_raw = false, int next_blk = -1) {
  typedef Traits<T> TR;
  constexpr bool MASKED = MODE == 1;      // (MODE 2: the key mask is in the logits already, key_mask_rank1)
  float bv[16];
  if constexpr (BIAS && TR::ES == 2) {
    // raw chunks requested a tile ahead by the kernel (wave-uniform choices); the same registers then take the chunks of the
    // block at this position of the NEXT tile, a tile's worth of work ahead of their use
    if (use_raw) bias_raw_finish<T>(bv, *braw, p.bias_c);
    if (next_blk >= 0) bias_raw_request<T>(*braw, bias_row, next_blk, (jbase >> 2) & 1);
  }
  if (BIAS && (TR::ES != 2 || !use_raw)) {
    // loads from clamped (always valid) addresses; out-of-range positions are masked below or never stored, so their value is
    // irrelevant.  bias_row already points at a valid row.
    load_bias_block<T>(bv, bias_row, jbase, p.M, (p.M & 3) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0, p.bias_c,
                       (p.M & 7) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0, (jbase >> 2) & 1);
  }
  if constexpr (ONL) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if constexpr (BIAS) s[r] += bv[r];
      if constexpr (MASKED) s[r] = ((w >> crow(r, 0)) & 1u) ? s[r] : -INFINITY;      // 2^-inf == 0: the mask needs no second select
    }
    online_recentre<T, DB>(max16(s), mref, rmax, o, l, lacc, [&](float d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] -= d;
      other(d);
    });
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float e = fast_exp2(s[r]);
      if constexpr (TR::ES == 4) l += e;
      s[r] = e;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = s[r];                 // = c1 * qh.kh - c2 already: c1 rides on q, -c2 is the accumulator's initial value
      if constexpr (BIAS) x += bv[r];
      float e = fast_exp2(x);
      if constexpr (MASKED) e = ((w >> crow(r, 0)) & 1u) ? e : 0.f;
      if constexpr (TR::ES == 4) l += e;
      s[r] = e;
    }
  }
  pb.prep(s);
  if constexpr (TR::ES == 2) {      // row sum of this block's 32 keys, of the ROUNDED P~
#pragma unroll
    for (int e = 0; e < 8; ++e) l = TR::add_pair(pb.v[e >> 2][e & 3], l);
  }
}

// One 64-key tile for one wave.  Software-pipelined INSIDE the wave (in-order issue: the matrix pipe and the VALU
// only overlap if their instructions alternate in program order) and ACROSS tiles:
//     S0 chain | V requests | S1 chain  x exp(block 0) | mid() | PV(block 0) x exp(block 1) | PV(block 1)
// `kf` holds this tile's K row fragments, requested by the caller one phase earlier.  mid() is the workgroup
// barrier plus the K requests of the NEXT tile; it runs when every LDS read of this tile has been issued, so the
// next tile's fragments land during the PV products instead of being waited for at the top of the next tile
// (phase timing of the previous structure: 690 of 2490 cycles per tile were spent there, right after the barrier,
// with all four waves bursting their K reads at once).
// MODE: 0 = every pair valid, 1 = causal tiles on the diagonal (and their ragged tails): select per logit, 2 = key mask / ragged
// tail of a non-causal problem: rank-1 MFMA per block (key_mask_rank1)
template <typename T, int D, int MODE, bool BIAS, bool LEAN, bool ONL, typename Mid>
FCSA_DEV void fwd_tile(const char* vt, u32x4 (&kf)[2][TileGeom<D, Traits<T>::ES>::KS], const FragAddr<T, D>& fa,
                       const u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS], f32x16 (&o)[TileGeom<D, Traits<T>::ES>::DB],
                       float& l, f32x16& lacc, const FwdParams& p, float& c2row, float& rmax, uint64_t word, uint32_t ncm, int i, int j0, int diff,
                       const char* bias_row, Trace& ts, Mid&& mid, const char* knext, bool more_k, const char* kt,
                       u32x4 (*braw)[2][2] = nullptr, bool use_raw = false, int next_j0 = -1) {

  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr bool MASKED = MODE == 1, KEYM = MODE == 2;
  constexpr bool PREFETCH_K = !LEAN && D * TR::ES < 512;     // see fwd_kernel
  const int kx = fa.row_off / G::ROWB;      // lane & 31: the key row this lane holds in an A operand
  // validity bits of this lane's 16 keys per block.  Branch-free and BEFORE the MFMA chains on purpose: a runtime
  // branch between the last MFMA and the first read of its result gets too few wait states on the
  // taken path (hipcc 7.2 pads only the fall-through; seen with the 16-pass v_mfma_f32_32x32x2_f32).
  uint32_t w[2] = {0xffffffffu, 0xffffffffu};
  if constexpr (MASKED) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
      w[jb] = ((uint32_t)(word >> (32 * jb)) >> (4 * fa.hi)) & (le_mask(i + diff - (j0 + 32 * jb + 4 * fa.hi)) | ncm);
  }

  if constexpr (LEAN) {
    // per 32-key block: K fragments (4 k-steps at a time) -> S chain; V fragments of the block requested behind the chain, landing
    // during exp / pack; row sum on the matrix pipe; PV.  Everything is read from the LDS tile here (`kf` is unused).
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = -c2row;
      if constexpr (KEYM) s = key_mask_rank1<T>(s, (uint32_t)(word >> (32 * jb)), kx, fa.hi);
      constexpr int PF = 4;
#pragma unroll
      for (int k0 = 0; k0 < G::KS; k0 += PF) {
        u32x4 kfr[PF];
#pragma unroll
        for (int kk = 0; kk < PF; ++kk)
          if (k0 + kk < G::KS) kfr[kk] = fa.row_frag(kt, 32 * jb, k0 + kk);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < PF; ++kk)
          if (k0 + kk < G::KS) s = TR::mfma32(kfr[kk], qf[k0 + kk], s);
      }
      u32x4 vf[G::DB][2];
#pragma unroll
      for (int db = 0; db < G::DB; ++db) { vf[db][0] = fa.tr_frag(vt, 32 * jb, db); vf[db][1] = fa.tr_frag(vt, 32 * jb + 16, db); }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ONL) {
        if constexpr (MASKED) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = ((w[jb] >> crow(r, 0)) & 1u) ? s[r] : -INFINITY;
        }
        online_recentre<T, G::DB>(max16(s), c2row, rmax, o, l, lacc, [&](float d) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] -= d;
        });
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = fast_exp2(s[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float e = fast_exp2(s[r]);
          if constexpr (MASKED) e = ((w[jb] >> crow(r, 0)) & 1u) ? e : 0.f;
          s[r] = e;
        }
      }
      SecondB<T> pb;
      pb.prep(s);
#pragma unroll
      for (int e = 0; e < 8; ++e) l = TR::add_pair(pb.v[e >> 2][e & 3], l);
#pragma unroll
      for (int db = 0; db < G::DB; ++db) {
        o[db] = TR::mfma32(vf[db][0], pb.v[0], o[db]);
        o[db] = TR::mfma32(vf[db][1], pb.v[1], o[db]);
      }
    }
    mid();
  } else if constexpr (TR::ES == 2 && !BIAS) {
    constexpr int MFMA = 0x8, VALU = 0x2 | 0x400, DSR = 0x100;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = -c2row; s1[r] = -c2row; }   // exponent shift (static, or this row's max) as the accumulator's initial value
    if constexpr (KEYM) {
      s0 = key_mask_rank1<T>(s0, (uint32_t)word, kx, fa.hi);
      s1 = key_mask_rank1<T>(s1, (uint32_t)(word >> 32), kx, fa.hi);
    }
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s0 = TR::mfma32(kf[0][kk], qf[kk], s0);
    u32x4 vf0[G::DB][2], vf1[G::DB][2];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) { vf0[db][0] = fa.tr_frag(vt, 0, db); vf0[db][1] = fa.tr_frag(vt, 16, db); }
    __builtin_amdgcn_sched_barrier(0);
    FCSA_STAMP(ts, 3);
    // --- S1 chain interleaved with exp / pack of block 0
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s1 = TR::mfma32(kf[1][kk], qf[kk], s1);
    SecondB<T> pb0, pb1;
    // ONL: this tile's 64 keys are tested together, behind the barrier (which ends a scheduling region anyway, so the phases keep
    // their interleave): block 0 is exponentiated against the reference the tile started with, and where the test moves the
    // reference (rare) its S chain is simply run again -- the K fragments are still in registers -- and P~0 recomputed.
    float bm = -INFINITY;
    if constexpr (ONL) {
      if constexpr (MASKED) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = ((w[0] >> crow(r, 0)) & 1u) ? s0[r] : -INFINITY;
      }
      bm = max16(s0);
#pragma unroll
      for (int r = 0; r < 16; ++r) s0[r] = fast_exp2(s0[r]);
      pb0.prep(s0);
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float e = fast_exp2(s0[r]);
        if constexpr (MASKED) e = ((w[0] >> crow(r, 0)) & 1u) ? e : 0.f;
        s0[r] = e;
      }
      pb0.prep(s0);
    }
#pragma unroll
    for (int db = 0; db < G::DB; ++db) { vf1[db][0] = fa.tr_frag(vt, 32, db); vf1[db][1] = fa.tr_frag(vt, 48, db); }
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(VALU, (MASKED ? 48 : 26) / G::KS + 1, 0);
      __builtin_amdgcn_sched_group_barrier(DSR, (4 * G::DB + G::KS - 1) / G::KS, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    FCSA_STAMP(ts, 4);
    if constexpr (ONL) {
      if constexpr (MASKED) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s1[r] = ((w[1] >> crow(r, 0)) & 1u) ? s1[r] : -INFINITY;
      }
      bm = fmaxf(bm, max16(s1));
    }
    mid();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (ONL) {
      online_recentre<T, G::DB>(bm, c2row, rmax, o, l, lacc, [&](float d) {
        f32x16 t;
#pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = -c2row;          // (already the new reference)
        if constexpr (KEYM) t = key_mask_rank1<T>(t, (uint32_t)word, kx, fa.hi);
#pragma unroll
        for (int kk = 0; kk < G::KS; ++kk) t = TR::mfma32(kf[0][kk], qf[kk], t);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if constexpr (MASKED) t[r] = ((w[0] >> crow(r, 0)) & 1u) ? t[r] : -INFINITY;
          t[r] = fast_exp2(t[r]);
          s1[r] -= d;
        }
        pb0.prep(t);
      });
      __builtin_amdgcn_sched_barrier(0);
    }
    // --- K fragment requests of the NEXT tile, scheduled among the PV products of block 0 instead of as a burst right behind
    // the barrier (all eight waves bursting 8 ds_read_b128 there took 350 .. 770 ticks of a 2100-tick tile, phase trace; -1.5 %).
    // Branch-free: past the last tile the reads hit the idle buffer and are never used.  (A fully slot-fenced form of this
    // phase, as in fwd2_tile, measured no better than hipcc's own placement under these group hints.)
    if constexpr (PREFETCH_K) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kk = 0; kk < G::KS; ++kk) kf[jb][kk] = fa.row_frag(knext, 32 * jb, kk);
    }
    // --- PV of block 0 (row sum + DB output blocks) interleaved with exp / pack of block 1
#pragma unroll
    for (int e = 0; e < 8; ++e) l = TR::add_pair(pb0.v[e >> 2][e & 3], l);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      o[db] = TR::mfma32(vf0[db][0], pb0.v[0], o[db]);
      o[db] = TR::mfma32(vf0[db][1], pb0.v[1], o[db]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float e = fast_exp2(s1[r]);
      if constexpr (MASKED && !ONL) e = ((w[1] >> crow(r, 0)) & 1u) ? e : 0.f;      // (ONL: masked exponents are -inf already)
      s1[r] = e;
    }
    pb1.prep(s1);
    constexpr int NPV = 2 * G::DB;
#pragma unroll
    for (int m = 0; m < NPV; ++m) {
      __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(VALU, (MASKED ? 48 : 26) / NPV + 1, 0);
      if constexpr (PREFETCH_K) __builtin_amdgcn_sched_group_barrier(DSR, (2 * G::KS + NPV - 1) / NPV, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    FCSA_STAMP(ts, 8);
    // --- PV of block 1
#pragma unroll
    for (int e = 0; e < 8; ++e) l = TR::add_pair(pb1.v[e >> 2][e & 3], l);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      o[db] = TR::mfma32(vf1[db][0], pb1.v[0], o[db]);
      o[db] = TR::mfma32(vf1[db][1], pb1.v[1], o[db]);
    }
    FCSA_STAMP(ts, 9);
  } else {
    // generic order (f32, bias, online exponent reference): both S chains first, then per block: softmax, PV
    f32x16 s[2];
    auto shift_block1 = [&](float d) {        // block 1's logits were accumulated against the reference block 0 has just moved
#pragma unroll
      for (int r = 0; r < 16; ++r) s[1][r] -= d;
    };
    auto nothing = [](float) {};
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[jb][r] = -c2row;
      if constexpr (KEYM) s[jb] = key_mask_rank1<T>(s[jb], (uint32_t)(word >> (32 * jb)), kx, fa.hi);
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) s[jb] = TR::mfma32(kf[jb][kk], qf[kk], s[jb]);
    }
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      SecondB<T> pb;
      if constexpr (TR::ES == 2) {
        u32x4 vf[G::DB][2];
#pragma unroll
        for (int db = 0; db < G::DB; ++db) {
          vf[db][0] = fa.tr_frag(vt, 32 * jb, db);
          vf[db][1] = fa.tr_frag(vt, 32 * jb + 16, db);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (jb == 0) fwd_softmax_block<T, G::DB, MODE, BIAS, ONL>(s[0], pb, l, lacc, p, w[0], j0 + 4 * fa.hi, bias_row, c2row, rmax, o, shift_block1,
                                                                     braw ? &(*braw)[0] : nullptr, use_raw, next_j0);
        else fwd_softmax_block<T, G::DB, MODE, BIAS, ONL>(s[1], pb, l, lacc, p, w[1], j0 + 32 + 4 * fa.hi, bias_row, c2row, rmax, o, nothing,
                                                           braw ? &(*braw)[1] : nullptr, use_raw, next_j0 < 0 ? -1 : next_j0 + 32);
#pragma unroll
        for (int db = 0; db < G::DB; ++db) {
          o[db] = TR::mfma32(vf[db][0], pb.v[0], o[db]);
          o[db] = TR::mfma32(vf[db][1], pb.v[1], o[db]);
        }
      } else {
        if (jb == 0) fwd_softmax_block<T, G::DB, MODE, BIAS, ONL>(s[0], pb, l, lacc, p, w[0], j0 + 4 * fa.hi, bias_row, c2row, rmax, o, shift_block1,
                                                                     braw ? &(*braw)[0] : nullptr, use_raw, next_j0);
        else fwd_softmax_block<T, G::DB, MODE, BIAS, ONL>(s[1], pb, l, lacc, p, w[1], j0 + 32 + 4 * fa.hi, bias_row, c2row, rmax, o, nothing,
                                                           braw ? &(*braw)[1] : nullptr, use_raw, next_j0 < 0 ? -1 : next_j0 + 32);
#pragma unroll
        for (int db = 0; db < G::DB; ++db) o[db] = second_mma<T, D>(o[db], vt, 32 * jb, db, pb, fa);
      }
    }
    mid();
    if (PREFETCH_K && more_k) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kk = 0; kk < G::KS; ++kk) kf[jb][kk] = fa.row_frag(knext, 32 * jb, kk);
    }
  }
}

// (request_q_rows / finish_q_frags / load_q_frags: fcsa_common.cuh -- shared with fcsa_fwd3.hip)

// 64-key tiles per LDS stage of fwd_kernel (kernel and launcher must agree): kFwdSub with LDS-DMA staging and no dynamic-shift
// pre-pass (which stages single tiles through the same buffers), else 1
template <typename T, int D, bool DYN> constexpr int fwd_stage_tiles() {
  return (!DYN && Traits<T>::ES == 2 && (64 * TileGeom<D, Traits<T>::ES>::ROWB) % 1024 == 0) ? kFwdSub : 1;
}

// DYN: per-row exponent reference for logit ranges no constant shift can hold, kept online (online_recentre); the S accumulators
// start from -reference instead of the static shift and inv_l is saved as log2(1 / sum_j exp(S_ij)), i.e. for shift 0.
// KM: the launch is NOT causal (compile-time: a third tile loop in one kernel made hipcc spill 200 registers): tiles that need
// masking -- a key mask, the ragged last tile -- take the rank-1 MFMA form (fwd_tile MODE 2) instead of the per-logit select.
// KSPLIT (8 waves; the lean tile, or the generic tile where a bias rides along; static or online exponent reference): the workgroup
// owns 128 query rows and its two wave halves split the KEYS -- a stage is two 64-key
// tiles, waves 0-3 take the even tile, waves 4-7 the odd one, for the same four 32-row slices -- and add their (O, l) partials through
// the LDS at the end of the pass (plain sums; with the online reference the halves first meet at the larger one).  For grids whose 128-row workgroups cannot give
// every SIMD two waves: 256 four-wave workgroups (C2: 4 x 8 x 1024 rows; C5 at D = 128, where the four-wave form runs ONE wave per
// SIMD whatever the grid) are one wave per SIMD; this form keeps the grid and doubles the waves, and the partner wave hides what the
// lean form does not prefetch.
template <typename T, int D, int NW, bool BIAS, bool DYN, bool LEAN, bool KM, bool KSPLIT = false>
__global__ void __launch_bounds__(NW * 64, ((D * Traits<T>::ES <= 128 || LEAN) ? 2 : 1)) fwd_kernel(const FwdParams p) {
  static_assert(!KSPLIT || (NW == 8 && (LEAN != BIAS) && Traits<T>::ES == 2), "key-split form: 8 waves, 16 bit; lean tile, or the generic tile with a bias");
  const int causal = KM ? 0 : p.causal;      // (same type and value as p.causal: the causal instantiations compile to what they were)
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr int RWAVES = KSPLIT ? NW / 2 : NW;            // waves that own distinct row slices
  constexpr int BN = 64, BM = 32 * RWAVES, NT = NW * 64;
  constexpr int TILE_B = BN * G::ROWB;
  constexpr int SUB = KSPLIT ? 2 : fwd_stage_tiles<T, D, DYN>();       // 64-key tiles per stage
  constexpr int STAGE_B = 2 * SUB * TILE_B;                // K tiles | V tiles of one stage
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][SUB K tiles | SUB V tiles]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rwave = KSPLIT ? (wave & (RWAVES - 1)) : wave;      // row slice of this wave
  const int half = KSPLIT ? wave / RWAVES : 0;                  // KSPLIT: which tile of a stage
  FragAddr<T, D> fa;
  fa.init(lane);

  // Work of a row tile grows with its index under causal masking (2 .. 2*MT key tiles), and a workgroup runs
  // start to finish on one CU, so the makespan would be set by the heaviest tile.  Each workgroup therefore
  // takes a PAIR of row tiles (MT-1-pt, pt): constant work per workgroup.  Non-causal: one tile each.
  const int MT = (p.N + BM - 1) / BM;
  const int PT = causal ? (MT + 1) / 2 : MT;
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (causal && (MT - 1 - pt) != pt) ? 2 : 1;
  // split-key launches (gridDim.y = p.splits > 1, never bias / dynamic shift): this workgroup sees the keys [k_lo, k_lo + Mk) only and
  // writes un-normalised partials; everything below works on that sub-problem.  Causal launches (round 6) split the key range of EACH row
  // tile -- the keys up to its diagonal -- so the ranges are set per pass (below): the pair (MT-1-pt, pt) keeps its constant work, 1 / splits
  // of it per workgroup
  int k_lo = 0, Mk = p.M;
  if (p.splits > 1 && !causal) {
    const int tps = ((p.M + BN - 1) / BN + p.splits - 1) / p.splits;      // 64-key tiles per split
    k_lo = (int)blockIdx.y * tps * BN;
    Mk = max(0, min(p.M, k_lo + tps * BN) - k_lo);
  }
  int diff = p.M - p.N - k_lo;                    // cu:1097 seq_len_diff (in the sub-problem's key numbering)
  const uint32_t ncm = causal ? 0u : 0xffffffffu;   // OR-ed into the causal bit mask: all ones when not causal
  Trace ts;
  ts.reset();
#ifdef FCSA_TRACE_WG
  const unsigned long long trace_t0 = trace_now();
#endif
#ifdef FCSA_TRACE
  unsigned long long pass_marks[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
  unsigned long long first_iter[2][2] = {{0, 0}, {0, 0}};      // duration of the first iteration of the [pass][unmasked, masked] loop
#define FCSA_PASS_MARK(k) pass_marks[pass][k] = trace_now()
#elif defined(FCSA_TRACE_WG)
#define FCSA_PASS_MARK(k) do { if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 256) g_trace_pass_fwd[blockIdx.x * 10 + pass * 5 + (k)] = trace_now(); } while (0)
#else
#define FCSA_PASS_MARK(k) ((void)0)
#endif
#ifdef FCSA_TRACE_BAR
  unsigned long long bar_wait = 0, bar_loop = 0;
#endif
  for (int pass = 0; pass < npass; ++pass) {
  FCSA_PASS_MARK(0);
  const int mt = causal ? (pass == 0 ? MT - 1 - pt : pt) : pt;      // heavy tile first
  const int m0 = mt * BM;
  const int mw = m0 + rwave * 32;                 // first query row of this wave
  const int i = mw + (lane & 31);                 // this lane's query row
  if (p.splits > 1 && causal) {                   // this row tile's visible keys [0, vis), split over gridDim.y workgroups
    const int vis = max(0, min(p.M, m0 + BM + p.M - p.N));
    const int tps = max(1, ((vis + BN - 1) / BN + p.splits - 1) / p.splits);
    k_lo = min((int)blockIdx.y * tps * BN, p.M);
    Mk = max(0, min(p.M, k_lo + tps * BN) - k_lo);
    diff = p.M - p.N - k_lo;
  }

  // key tiles this workgroup needs
  int last_key = Mk - 1;
  if (causal) last_key = min(last_key, m0 + BM - 1 + diff);
  const int nt = last_key < 0 ? 0 : last_key / BN + 1;

  const char* kbase = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)k_lo * p.k.sn;
  const char* vbase = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh + (int64_t)k_lo * p.v.sn;
  // DMA form (16-bit types): tile t+1 goes global -> LDS by LDS-DMA (DmaStager: no staging registers, no ds_write passes), issued
  // at the top of tile t into the buffer whose last reads finished before the barrier of tile t-1, and waited for right before
  // the barrier of tile t.  The FIRST tile is issued here, ahead of the Q fragments and their fused l2norm, so that its
  // HBM / L2 latency hides under that work (the previous pass ended with a barrier: the buffers are free).
  constexpr bool DMA = TR::ES == 2 && (BN * G::ROWB) % 1024 == 0;
  constexpr bool EARLY = DMA;
  Stager<T, D, BN, NT> sk, sv;
  typedef DmaStager<T, D, DMA ? BN * SUB : 1024, NW> DS;
  DS dk_, dv_;
  typename DS::Stream stk, stv;       // K / V walked tile by tile from key k_lo: one descriptor per pass, one scalar add per tile
  uint32_t k_step = 0, v_step = 0, lds0 = 0;
  bool far = false;
  if constexpr (DMA) {
    dk_.init(p.k.sn, wave, lane);
    dv_.init(p.v.sn, wave, lane);
    stk = dk_.open(kbase, p.k.sn, Mk);
    stv = dv_.open(vbase, p.v.sn, Mk);
    k_step = (uint32_t)(BN * SUB * p.k.sn);
    v_step = (uint32_t)(BN * SUB * p.v.sn);
    far = BN * SUB * p.k.sn > (int64_t)DS::REBASE || BN * SUB * p.v.sn > (int64_t)DS::REBASE;
    lds0 = DS::lds_addr(smem);
    if (EARLY && nt > 0) {
      dk_.issue(stk, lds0, wave);
      dv_.issue(stv, lds0 + SUB * TILE_B, wave);
    }
  } else {
    sk.init(p.k.sn, tid);
    sv.init(p.v.sn, tid);
  }

  u32x4 qf[G::KS];
  request_q_rows<T, D>(p, b, h, i, fa.hi, qf);
  finish_q_frags<T, D, LEAN || (BIAS && DYN)>(p, b, h, i, fa, qf, half == 0);
  FCSA_PASS_MARK(1);

  f32x16 o[G::DB];
#pragma unroll
  for (int db = 0; db < G::DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float l = 0.f;          // f32: per-lane partial row sum
  f32x16 lacc;            // (rounds 1 - 3: row sums from an all-ones MFMA; stays zero since the 16-bit sums moved to the VALU, kept for the online-reference plumbing)
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;

  const uint8_t* mrow = p.mask ? p.mask + (int64_t)b * p.M + k_lo : nullptr;
  const char* bias_row = nullptr;                 // row min(i, N-1): always a valid address
  if constexpr (BIAS)
    bias_row = p.bias + ((int64_t)(p.bias_batch ? b : h) * p.N + min(i, p.N - 1)) * (int64_t)p.M * (int64_t)sizeof(typename TR::elem);

  // exponent reference of this lane's row: the problem's constant shift, or (DYN) a per-row value kept online by fwd_softmax_block
  float c2row = DYN ? 0.f : p.c2;
  float rmax = -INFINITY;     // DYN: largest exponent seen so far relative to c2row (-inf: no valid key yet)

  // Pipeline (per 64-key tile t; two LDS buffers, one register staging set, ONE barrier per tile):
  //   top of t : staging registers (tile t+1, loaded during t-1) -> LDS buffer (t+1)&1; global loads of tile t+2
  //   tile t   : S chains / exp / V requests from buffer t&1, K fragments already in registers
  //   mid()    : barrier (all LDS reads of tile t returned, tile t+1 visible), K fragment requests of tile t+1
  //   PV products of tile t
  // Buffer (t+1)&1 held tile t-1, whose last reads every wave completed before the barrier of t-1.
  uint8_t mb = 1;
  // BIAS, 16-bit: the raw bias chunks of a tile's two blocks are requested one tile ahead (bias_raw_request) where the whole 64-key
  // tile lies inside 16-byte-aligned bias rows (wave-uniform); other tiles load inside the block as before
  constexpr bool BIAS_AHEAD = BIAS && TR::ES == 2;
  u32x4 bnext[2][2];
  bool bnext_ok = false;
  const bool bias_rows16 = BIAS_AHEAD && (p.M & 7) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
  auto request_bias = [&](int j0n) {
    bnext_ok = bias_rows16 && j0n + BN <= p.M;
    if (bnext_ok) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb) bias_raw_request<T>(bnext[jb], bias_row, j0n + 32 * jb, fa.hi);
    }
  };
  u32x4 kf[2][G::KS];
  auto request_k = [&](const char* kt) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) kf[jb][kk] = fa.row_frag(kt, 32 * jb, kk);
  };
  if (nt > 0) {
    if constexpr (DMA) {
      if (!EARLY) {
        dk_.issue(stk, lds0, wave);
        dv_.issue(stv, lds0 + SUB * TILE_B, wave);
      }
    } else {
      sk.load(kbase, p.k.sn, Mk);
      sv.load(vbase, p.v.sn, Mk);
    }
    if (mrow) mb = half * BN + lane < Mk ? mrow[half * BN + lane] : (uint8_t)0;
    if constexpr (BIAS_AHEAD) request_bias(0);
  }
  // Every prologue load (Q fragments, first tile, mask byte) is complete here on the real path; say so on ALL
  // paths.  Otherwise hipcc's waitcnt model keeps the Q loads pending along the no-tile path, the loop-header
  // merge never clears that, and each iteration waits with vmcnt(0) at its first MFMA -- right after issuing
  // the prefetch of tile t+2, which serialises the prefetch with the compute meant to hide it.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched
  if constexpr (DMA) {
    dma_wait();
  } else if (nt > 0) {
    sk.store(smem, tid);
    sv.store(smem + TILE_B, tid);
    if (nt > 1) {
      sk.load(kbase + (int64_t)BN * p.k.sn, p.k.sn, Mk - BN);
      sv.load(vbase + (int64_t)BN * p.v.sn, p.v.sn, Mk - BN);
    }
  }
  __syncthreads();
  FCSA_PASS_MARK(2);
  // K fragments of a tile are 8 * KS registers; 512-byte rows (f32, D = 128) cannot hold them across the PV products
  constexpr bool PREFETCH_K = !LEAN && D * TR::ES < 512;      // (LEAN: K fragments are read per block inside the tile)
  if (PREFETCH_K && nt > 0) request_k(smem);

  // tiles [0, t_split) need no masking for THIS wave, tiles [t_split, nt) do (wave-uniform split; both
  // loops execute one barrier per tile, so waves of one workgroup may sit in different loops)
  int t_split = 0;
  if (!BIAS && mrow == nullptr) {
    t_split = Mk / BN;                                                 // tail tile (j0 + BN > Mk) is masked
    if (causal) t_split = min(t_split, max(0, mw + diff + 1) / BN);  // needs (t+1)*BN - 1 <= mw + diff
    t_split = min(t_split, nt);
  }

#ifdef FCSA_TRACE_BAR
  unsigned long long bar_t = 0, loop_t = 0;
  FCSA_BAR_BEGIN(loop_t);
#endif
  auto run = [&](auto masked_tag, int t_begin, int t_end) {
    constexpr int MODE = decltype(masked_tag)::value;      // fwd_tile: 0 all valid, 1 causal select, 2 key mask by rank-1 MFMA
    constexpr bool MASKED = MODE != 0;
    for (int t = t_begin; t < t_end; ++t) {
#ifdef FCSA_TRACE
      if (t == t_begin + 1) first_iter[pass][MASKED ? 1 : 0] = trace_now() - first_iter[pass][MASKED ? 1 : 0];
      if (t == t_begin) first_iter[pass][MASKED ? 1 : 0] = trace_now();
#endif
      const int j0 = t * BN;
      const int u = t / SUB, sub = t % SUB;                    // stage, tile inside the stage
      const char* vcur = smem + (u & 1) * STAGE_B + (SUB + sub) * TILE_B;
      char* knxt = smem + (sub + 1 < SUB ? (u & 1) * STAGE_B + (sub + 1) * TILE_B : ((u + 1) & 1) * STAGE_B);
      const bool last_of_stage = sub == SUB - 1 || t + 1 >= nt;   // workgroup-uniform
      FCSA_STAMP(ts, 0);
      if constexpr (kPrioFwd == 1 && NW == 8) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); }
      if constexpr (kPrioFwd == 2 && NW == 8) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }      // (mirror: favoured from the tile top to the barrier)
      uint64_t word = 0;
      if constexpr (MASKED) {
        // consume the mask byte loaded one tile ago BEFORE issuing new loads (its wait then covers nothing else)
        word = __ballot((j0 + lane) < Mk && mb != 0);                   // valid keys of this tile
        if (mrow && t + 1 < nt) {
          const int key = j0 + BN + lane;
          mb = key < Mk ? mrow[key] : (uint8_t)0;
        }
      }
      if constexpr (DMA) {
        // stage u + 1 is requested at the first tile of stage u: its buffer was last read in stage u - 1, which ended with a barrier
        if (sub == 0 && (u + 1) * SUB < nt) {
          stk.off += k_step;
          stv.off += v_step;
          if (far || (stk.off | stv.off) > DS::REBASE) {      // 32-bit offsets about to run out: re-open at this stage
            stk = dk_.open(kbase + (int64_t)(u + 1) * BN * SUB * p.k.sn, p.k.sn, Mk - (u + 1) * BN * SUB);
            stv = dv_.open(vbase + (int64_t)(u + 1) * BN * SUB * p.v.sn, p.v.sn, Mk - (u + 1) * BN * SUB);
          }
          const uint32_t lds_nxt = lds0 + ((u + 1) & 1) * STAGE_B;
          dk_.issue(stk, lds_nxt, wave);
          dv_.issue(stv, lds_nxt + SUB * TILE_B, wave);
        }
        FCSA_STAMP(ts, 1);
      } else {
        if (t + 1 < nt) {
          sk.store(knxt, tid);
          sv.store(knxt + TILE_B, tid);
        }
        FCSA_STAMP(ts, 1);
        if (t + 2 < nt) {
          sk.load(kbase + (int64_t)(j0 + 2 * BN) * p.k.sn, p.k.sn, Mk - (j0 + 2 * BN));
          sv.load(vbase + (int64_t)(j0 + 2 * BN) * p.v.sn, p.v.sn, Mk - (j0 + 2 * BN));
        }
      }
      FCSA_STAMP(ts, 2);
      auto mid = [&]() {          // every LDS read of this tile has been issued; at the last tile of a stage: publish the next stage
        FCSA_STAMP(ts, 5);
        if (last_of_stage) {
          if constexpr (DMA) dma_wait();
          FCSA_BAR_BEGIN(bar_t);
          __syncthreads();
          FCSA_BAR_END(bar_t, bar_wait);
          // (kPrioFwd: the younger half of the workgroup is favoured from the barrier to the end of the tile, the older half -- by age --
          //  from the top of the next tile to the barrier; see fcsa_common.cuh)
          if constexpr (kPrioFwd == 1 && NW == 8) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
          if constexpr (kPrioFwd == 2 && NW == 8) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); }
          if constexpr (kPrioFwd == 3 && NW == 8) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
        } else {
          if constexpr (kPrioFwd == 3 && NW == 8) {      // (multi-tile stages: the younger half is favoured for the first half of the interval)
            if (sub == SUB / 2 - 1) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); }
          }
        }
        FCSA_STAMP(ts, 6);
      };
      // (BIAS_AHEAD) this tile's raw bias chunks are in bnext if bcur_ok; each block re-requests its registers for the next tile
      const bool bcur_ok = bnext_ok;
      const bool bfut_ok = BIAS_AHEAD && t + 1 < nt && bias_rows16 && j0 + 2 * BN <= p.M;
      bool skip = false;
      if constexpr (MASKED) skip = causal && (j0 > mw + 31 + diff);              // no valid pair for this wave
      if (!LEAN && !PREFETCH_K && !skip) request_k(vcur - SUB * TILE_B);
      if (skip) {
        bnext_ok = false;                  // (nothing was requested for the next tile; once a wave skips it skips to the end of the pass)
        mid();
        if (PREFETCH_K && t + 1 < nt) request_k(knxt);
      } else {
        fwd_tile<T, D, tile_mode<MODE>(), BIAS, LEAN, DYN>(vcur, kf, fa, qf, o, l, lacc, p, c2row, rmax, word, ncm, i, j0, diff, bias_row, ts, mid, knxt, t + 1 < nt,
                                     vcur - SUB * TILE_B, BIAS_AHEAD ? &bnext : nullptr, bcur_ok, bfut_ok ? j0 + BN : -1);
        bnext_ok = bfut_ok;
      }
      FCSA_STAMP(ts, 10);
      if constexpr (!MASKED) ts.close(10);     // trace: unmasked tiles only
    }
  };
  if constexpr (!KSPLIT) {
    run(std::integral_constant<int, 0>{}, 0, t_split);
    run(std::integral_constant<int, KM ? 2 : 1>{}, t_split, nt);
  } else {
    // stage u = tiles 2u, 2u + 1; this wave's tile is t = 2u + half.  ONE barrier per stage for every wave, whatever its tile needs:
    // first the stages whose tile needs no masking, then the others (a stage past this wave's last tile is a bare barrier)
    const int ns = (nt + 1) / 2;
    const int u_split = min(ns, max(0, (t_split - half + 1) / 2));
    auto stage = [&](auto masked_tag, int u) {
      constexpr int MODE = decltype(masked_tag)::value;
      const int t = 2 * u + half, j0 = t * BN;
      const char* buf = smem + (u & 1) * STAGE_B;
      uint64_t word = 0;
      if constexpr (MODE != 0) {
        word = __ballot((j0 + lane) < Mk && mb != 0);
        if (mrow && t + 2 < nt) {
          const int key = j0 + 2 * BN + lane;
          mb = key < Mk ? mrow[key] : (uint8_t)0;
        }
      }
      if (2 * (u + 1) < nt) {      // stage u + 1: its buffer was last read in stage u - 1, which ended with a barrier
        stk.off += k_step;
        stv.off += v_step;
        if (far || (stk.off | stv.off) > DS::REBASE) {
          stk = dk_.open(kbase + (int64_t)(u + 1) * BN * SUB * p.k.sn, p.k.sn, Mk - (u + 1) * BN * SUB);
          stv = dv_.open(vbase + (int64_t)(u + 1) * BN * SUB * p.v.sn, p.v.sn, Mk - (u + 1) * BN * SUB);
        }
        const uint32_t lds_nxt = lds0 + ((u + 1) & 1) * STAGE_B;
        dk_.issue(stk, lds_nxt, wave);
        dv_.issue(stv, lds_nxt + SUB * TILE_B, wave);
      }
      auto mid = [&]() {
        dma_wait();
        FCSA_BAR_BEGIN(bar_t);
        __syncthreads();
        FCSA_BAR_END(bar_t, bar_wait);
      };
      bool skip = t >= nt;
      if constexpr (MODE == 1) skip = skip || (causal && j0 > mw + 31 + diff);
      if (skip) {
        mid();
        return;
      }
      if constexpr (!LEAN) request_k(buf + half * TILE_B);      // (bias form: the generic tile takes its K fragments from registers; nothing is prefetched across stages)
      fwd_tile<T, D, tile_mode<MODE>(), BIAS, LEAN, DYN>(buf + (SUB + half) * TILE_B, kf, fa, qf, o, l, lacc, p, c2row, rmax, word, ncm, i, j0, diff, bias_row, ts, mid,
                                           nullptr, false, buf + half * TILE_B);
    };
    for (int u = 0; u < u_split; ++u) stage(std::integral_constant<int, 0>{}, u);
    for (int u = u_split; u < ns; ++u) stage(std::integral_constant<int, KM ? 2 : 1>{}, u);
  }
  FCSA_PASS_MARK(3);
#ifdef FCSA_TRACE_BAR
  FCSA_BAR_END(loop_t, bar_loop);
#endif
  // no trailing barrier: every wave completed its last LDS read before the final mid() barrier, so the next
  // pass may overwrite buffer 0 in its prologue

  // epilogue: normalise and store.  Lane (i, hi) holds O[i][32*db + 8*rq + 4*hi + 0..3].
  float lt = (TR::ES == 2) ? lacc[0] + xhalf_sum(l) : xhalf_sum(l);
  if constexpr (KSPLIT) {
    // the odd-tile half hands its partials to the even-tile half of the same rows through the LDS (the staging buffers are free:
    // every wave's last LDS read came before the last stage barrier); 16-byte accesses, lane-contiguous
    // DYN: the halves kept their own per-row exponent references (c2row; a half that met no valid key has rmax == -inf and nothing
    // accumulated): the partials are brought to the larger reference before they are added -- the one rescale a running max costs here
    constexpr int NV = G::DB * 4 + 1;                                  // f32x4 per lane: O^T accumulators + (l, reference, valid, -)
    f32x4* ms = reinterpret_cast<f32x4*>(smem) + rwave * (NV * 64) + lane;
    if (half == 1) {
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {o[db][4 * g], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]};
          ms[(db * 4 + g) * 64] = v;
        }
      const f32x4 lv = {lt, c2row, rmax == -INFINITY ? 0.f : 1.f, 0.f};
      ms[(NV - 1) * 64] = lv;
    }
    __syncthreads();
    if (half == 0) {
      const f32x4 lv = ms[(NV - 1) * 64];
      float f0 = 1.f, f1 = 1.f;
      if constexpr (DYN) {
        const bool v0 = rmax != -INFINITY, v1 = lv[2] != 0.f;
        const float m = v0 ? (v1 ? fmaxf(c2row, lv[1]) : c2row) : lv[1];
        f0 = v0 ? fast_exp2(c2row - m) : 0.f;
        f1 = v1 ? fast_exp2(lv[1] - m) : 0.f;
        c2row = m;
        if (v1) rmax = 0.f;      // (only its -inf-ness is read below)
      }
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = ms[(db * 4 + g) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[db][4 * g + e] = DYN ? o[db][4 * g + e] * f0 + v[e] * f1 : o[db][4 * g + e] + v[e];
        }
      lt = DYN ? lt * f0 + lv[0] * f1 : lt + lv[0];
    }
    __syncthreads();                                                    // (the row epilogue's scratch overlays what was just read)
    if (half == 1) {
      if (pass + 1 < npass) __syncthreads();
      continue;
    }
  }
  if (p.splits > 1) {     // un-normalised partial of this key range; fwd_combine_kernel sums, clamps and normalises
    if (i < p.N) {
      const int64_t prow = ((int64_t)blockIdx.y * p.B * p.H + bh) * p.N + i;
      if (fa.hi == 0) p.ws_l[prow] = lt;
      store_row_tile<T, D>(reinterpret_cast<char*>(p.ws_o + prow * D), o, 1.f, fa.hi, true);
    }
    if (pass + 1 < npass) __syncthreads();      // (causal pairs: the barrier the key-split form's other half waits at, and the one that frees the LDS)
    continue;
  }
  const float inv = 1.f / fmaxf(lt, p.l_eps);     // cu:1239 (constants::eps, cu:83), rescaled with the shift
  // saved for the backward in the GLOBAL shift convention; DYN: log2(1 / sum_j exp(S_ij)) = log2(inv) - (row reference), which is what
  // the backward kernels seed their S accumulators with -- the normaliser itself may not fit f32 at such logit ranges
  if (i < p.N && p.inv_l != nullptr && fa.hi == 0)
    p.inv_l[((int64_t)b * p.H + h) * p.N + i] = DYN ? __builtin_amdgcn_logf(inv) - c2row : inv;
  // O rows through the LDS (RowEpilogue): every wave issued its last LDS read of the key loop before the final barrier, so the
  // staging buffers are free; the next pass's prologue must not overwrite the scratch while another wave still reads it.
  {
    typedef RowEpilogue<T, D> EP;
    if (p.N - mw > 0)
      EP::store(smem + rwave * EP::BYTES_NOX, o, inv, (LEAN || (BIAS && DYN)) ? opaque(lane) : lane, p.o.p + (int64_t)b * p.o.sb + (int64_t)h * p.o.sh + (int64_t)mw * p.o.sn, p.o.sn,
                p.N - mw, false, nullptr, 0, 1.f, nullptr, 1, 0, 1.f);
    if (pass + 1 < npass) __syncthreads();
  }
  FCSA_PASS_MARK(4);
  }   // pass
#ifdef FCSA_TRACE_BAR
  if (blockIdx.x == gridDim.x / 2 + 3 && blockIdx.y == 0 && lane == 0) { g_trace_bar_fwd[2 * wave] = bar_wait; g_trace_bar_fwd[2 * wave + 1] = bar_loop; }
#endif
#ifdef FCSA_TRACE_WG
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 1024) { g_trace_wg_fwd[2 * blockIdx.x] = trace_t0; g_trace_wg_fwd[2 * blockIdx.x + 1] = trace_now(); }
#endif
#ifdef FCSA_TRACE
  if (blockIdx.x == gridDim.x / 2 + 3 && (tid & 63) == 0 && (NW == 4 ? wave < 4 : (wave & 2) == 0)) {
    unsigned long long* out = g_trace_fwd + 32 * (NW == 4 ? wave : (wave & 1) + 2 * (wave >> 2));
    ts.dump(out, trace_now() - trace_t0);   // 8 waves: 0, 1, 4, 5
    for (int ps = 0; ps < 2; ++ps)
      for (int k = 0; k < 4; ++k) out[14 + 4 * ps + k] = pass_marks[ps][k + 1] - pass_marks[ps][k];   // Q frags | first tile | key loop | epilogue
    out[22] = pass_marks[0][0] - trace_t0;
    for (int ps = 0; ps < 2; ++ps) { out[23 + 2 * ps] = first_iter[ps][0]; out[24 + 2 * ps] = first_iter[ps][1]; }
  }
#endif
}

#ifdef FCSA_TRACE_BAR
}  // namespace fcsa
extern "C" int fcsa_trace_read_bar_fwd(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_bar_fwd), sizeof(unsigned long long) * 64);
}
namespace fcsa {
#endif
#ifdef FCSA_TRACE_WG
}  // namespace fcsa
extern "C" int fcsa_trace_read_pass_fwd(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_pass_fwd), sizeof(unsigned long long) * 2560);
}
extern "C" int fcsa_trace_read_wg_fwd(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_wg_fwd), sizeof(unsigned long long) * 2048);
}
namespace fcsa {
#endif
#ifdef FCSA_TRACE
}  // namespace fcsa
extern "C" int fcsa_trace_read_fwd(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_fwd), sizeof(unsigned long long) * 128);
}
namespace fcsa {
#endif

// =============================================================================================
// Wide forward kernel (16-bit types, no bias): every wave owns 64 query rows = two 32-row blocks that share each
// K / V fragment, i.e. two MFMAs per LDS fragment read.  Motivation (phase timing + PMC of the 32-row kernel): a
// 32-row wave moves 1 KiB through the LDS per 32x32x16 MFMA, so the CU's 128 B/clk LDS port is exactly as loaded
// as its matrix pipes and the fragment requests of the four waves queue up behind each other (8 ds_read_b128
// took ~400 cycles to ISSUE).  64 rows per wave halve that traffic; the two independent row blocks also give the
// in-order wave two accumulator chains to alternate.  One workgroup (4 waves, 256 rows) per CU, so the
// register budget is 512 per lane and nothing spills.  Row sums come from v_dot2c against packed ones on the
// VALU (exact f32 sums of the rounded P~, like the ones-MFMA of the narrow kernel, without its 25% extra MFMAs).
// Used when the grid still fills the chip (launch_forward); the narrow kernel covers everything else.
// =============================================================================================
// Rotating software pipeline of the wide kernel.  The wave issues in order, so the matrix pipe and the VALU overlap
// only if their instructions alternate in PROGRAM order; hipcc's list scheduler follows sched_group_barrier hints
// only loosely (it produced "4 MFMAs, then 22 v_exp"), so every MFMA gets its own fenced issue SLOT that also holds
// a fixed share of the VALU work and at most a couple of memory instructions.  Per 64-key tile t, four phases:
//
//     P1  S0(t)    | softmax 2nd half of block 1 of tile t-1 | V requests block 0, stage store t+1, stage loads t+2
//     P2  PV1(t-1) | softmax 1st half of block 0 of tile t
//     P3  S1(t)    | softmax 2nd half of block 0 of tile t   | V requests block 1
//     --  barrier (every LDS read of tile t has returned; tile t+1 is visible)
//     P4  PV0(t)   | softmax 1st half of block 1 of tile t   | K requests of tile t+1
//
// i.e. the exp / pack / row-sum work of a 32-key block (32 values per lane) is spread two values per slot over the
// 16 MFMAs that follow its S chain: 2 v_exp + v_cvt_pk + v_dot2c = 24 VALU cycles under a 32-cycle MFMA.  A block's
// PV product runs one phase after its softmax completes; PV1 of the last tile is drained after the loop.
template <typename T, int D> struct Fwd2State {
  typedef TileGeom<D, 2> G;
  f32x16 s0[2], s1[2];                  // logits [row block]
  SecondB<T> pb0[2], pb1[2];            // packed P~ [row block]
  u32x4 vf0[G::DB][2], vf1[G::DB][2];   // transposed V fragments [feature block][16-key step]
  uint32_t w1[2];                       // validity bits of key block 1 of the tile whose softmax is still in flight
};

// two logits of one block: exp, (mask), pack, row sum.  c = 0..7 selects accumulator registers 2c, 2c+1
template <typename T, bool MASKED>
FCSA_DEV void soft2(const f32x16& s, uint32_t wm, SecondB<T>& pb, float& lr, int c) {
  float e0 = fast_exp2(s[2 * c]), e1 = fast_exp2(s[2 * c + 1]);
  if constexpr (MASKED) {
    e0 = ((wm >> crow(2 * c, 0)) & 1u) ? e0 : 0.f;
    e1 = ((wm >> crow(2 * c + 1, 0)) & 1u) ? e1 : 0.f;
  }
  const uint32_t u = Traits<T>::pack2(e0, e1);
  pb.v[c >> 2][c & 3] = u;
  lr = Traits<T>::add_pair(u, lr);
}


template <typename T, int D, bool MASKED, typename StageStore, typename StageLoad, typename Mid>
FCSA_DEV void fwd2_tile(const char* vt, const char* knext, u32x4 (&kf)[2][TileGeom<D, 2>::KS], const FragAddr<T, D>& fa,
                        const u32x4 (&qf)[2][TileGeom<D, 2>::KS], f32x16 (&o)[2][TileGeom<D, 2>::DB], float (&l)[2],
                        Fwd2State<T, D>& st, const FwdParams& p, uint64_t word, uint32_t ncm, int i0, int j0, int diff,
                        Trace& ts, StageStore&& stage_store, StageLoad&& stage_load, Mid&& mid) {
  typedef TileGeom<D, 2> G;
  typedef Traits<T> TR;
  constexpr int NS = 2 * G::KS;          // MFMAs of an S phase
  constexpr int NPV = 4 * G::DB;         // MFMAs of a PV phase
  constexpr int NSTG = 2 * Stager<T, D, 64, 256>::PER;
  uint32_t w[2][2] = {{0xffffffffu, 0xffffffffu}, {0xffffffffu, 0xffffffffu}};     // [key block][row block]
  if constexpr (MASKED) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int r = 0; r < 2; ++r)
        w[jb][r] = ((uint32_t)(word >> (32 * jb)) >> (4 * fa.hi)) & (le_mask(i0 + 32 * r + diff - (j0 + 32 * jb + 4 * fa.hi)) | ncm);
  }
  f32x16 cinit;
#pragma unroll
  for (int e = 0; e < 16; ++e) cinit[e] = -p.c2;       // exponent shift as the accumulator's initial value
  FCSA_FENCE();
  FCSA_STAMP(ts, 0);
  // ---- P1: S0(t) | second half of softmax(block 1, tile t-1) | V requests block 0, stage store, stage loads
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    const int kk = m >> 1, r = m & 1;
    st.s0[r] = TR::mfma32(kf[0][kk], qf[r][kk], kk == 0 ? cinit : st.s0[r]);
    FCSA_SHARE(m, NS, 8, c) soft2<T, MASKED>(st.s1[c >> 2], st.w1[c >> 2], st.pb1[c >> 2], l[c >> 2], 4 + (c & 3));
    if (m < NS / 2) {
      FCSA_SHARE(m, NS / 2, 2 * G::DB, f) st.vf0[f >> 1][f & 1] = fa.tr_frag(vt, 16 * (f & 1), f >> 1);
    } else {
      FCSA_SHARE(m - NS / 2, NS - NS / 2, NSTG, x) stage_store(x);     // after the V requests: LDS writes do not move across reads
    }
    FCSA_FENCE();
  }
  FCSA_STAMP(ts, 1);
  // ---- P2: PV1(t-1) | first half of softmax(block 0, tile t) | stage loads of tile t+2 (after the stores: same registers)
#pragma unroll
  for (int m = 0; m < NPV; ++m) {
    const int ks = m / (2 * G::DB), db = (m >> 1) % G::DB, r = m & 1;
    o[r][db] = TR::mfma32(st.vf1[db][ks], st.pb1[r].v[ks], o[r][db]);
    FCSA_SHARE(m, NPV, 8, c) soft2<T, MASKED>(st.s0[c >> 2], w[0][c >> 2], st.pb0[c >> 2], l[c >> 2], c & 3);
    FCSA_SHARE(m, NPV, NSTG, x) stage_load(x);
    FCSA_FENCE();
  }
  FCSA_STAMP(ts, 2);
  // ---- P3: S1(t) | second half of softmax(block 0, tile t) | V requests block 1 (first half of the slots)
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    const int kk = m >> 1, r = m & 1;
    st.s1[r] = TR::mfma32(kf[1][kk], qf[r][kk], kk == 0 ? cinit : st.s1[r]);
    FCSA_SHARE(m, NS, 8, c) soft2<T, MASKED>(st.s0[c >> 2], w[0][c >> 2], st.pb0[c >> 2], l[c >> 2], 4 + (c & 3));
    if (m < NS / 2) {
      FCSA_SHARE(m, NS / 2, 2 * G::DB, f) st.vf1[f >> 1][f & 1] = fa.tr_frag(vt, 32 + 16 * (f & 1), f >> 1);
    }
    FCSA_FENCE();
  }
  st.w1[0] = w[1][0];
  st.w1[1] = w[1][1];
  FCSA_STAMP(ts, 3);
  mid();                                   // barrier
  FCSA_STAMP(ts, 4);
  // ---- P4: PV0(t) | first half of softmax(block 1, tile t) | K requests of tile t+1
#pragma unroll
  for (int m = 0; m < NPV; ++m) {
    const int ks = m / (2 * G::DB), db = (m >> 1) % G::DB, r = m & 1;
    o[r][db] = TR::mfma32(st.vf0[db][ks], st.pb0[r].v[ks], o[r][db]);
    FCSA_SHARE(m, NPV, 8, c) soft2<T, MASKED>(st.s1[c >> 2], w[1][c >> 2], st.pb1[c >> 2], l[c >> 2], c & 3);
    FCSA_SHARE(m, NPV, 2 * G::KS, f) kf[f / G::KS][f % G::KS] = fa.row_frag(knext, 32 * (f / G::KS), f % G::KS);
    FCSA_FENCE();
  }
  FCSA_STAMP(ts, 5);
}
template <typename T, int D, int NW>
__global__ void __launch_bounds__(NW * 64, 1) fwd2_kernel(const FwdParams p) {
  typedef TileGeom<D, 2> G;
  typedef Traits<T> TR;
  static_assert(TR::ES == 2, "16-bit types only");
  constexpr int BN = 64, RW = 64, BM = RW * NW, NT = NW * 64;
  constexpr int TILE_B = BN * G::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FragAddr<T, D> fa;
  fa.init(lane);

  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;                 // causal: pairs of row tiles (MT-1-pt, pt), constant work
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (p.causal && (MT - 1 - pt) != pt) ? 2 : 1;
  const int diff = p.M - p.N;
  const uint32_t ncm = p.causal ? 0u : 0xffffffffu;
  const char* kbase = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh;
  const char* vbase = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh;
  const uint8_t* mrow = p.mask ? p.mask + (int64_t)b * p.M : nullptr;
  Stager<T, D, BN, NT> sk, sv;
  sk.init(p.k.sn, tid);
  sv.init(p.v.sn, tid);
  Trace ts;
  ts.reset();
#ifdef FCSA_TRACE
  const unsigned long long trace_t0 = trace_now();
#endif

  for (int pass = 0; pass < npass; ++pass) {
    const int mt = p.causal ? (pass == 0 ? MT - 1 - pt : pt) : pt;      // heavy tile first
    const int m0 = mt * BM;
    const int mw = m0 + wave * RW;                  // first query row of this wave
    const int i0 = mw + (lane & 31);                // this lane's row in block 0; block 1 is i0 + 32
    int last_key = p.M - 1;
    if (p.causal) last_key = min(last_key, m0 + BM - 1 + diff);
    const int nt = last_key < 0 ? 0 : last_key / BN + 1;

    u32x4 qf[2][G::KS];
#pragma unroll
    for (int r = 0; r < 2; ++r) load_q_frags<T, D>(p, b, h, i0 + 32 * r, fa, qf[r]);
    f32x16 o[2][G::DB];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[r][db][e] = 0.f;
    float l[2] = {0.f, 0.f};          // per-lane partial row sums (this lane's 16 keys per block)

    // pipeline state "nothing in flight": the logits of tile -1 exponentiate to 0 and its V fragments are 0
    Fwd2State<T, D> st;
    auto reset_pipe = [&]() {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int e = 0; e < 16; ++e) st.s1[r][e] = -1e30f;
        const u32x4 z = {0u, 0u, 0u, 0u};
        st.pb1[r].v[0] = z;
        st.pb1[r].v[1] = z;
        st.w1[r] = 0xffffffffu;
      }
#pragma unroll
      for (int db = 0; db < G::DB; ++db) { const u32x4 z = {0u, 0u, 0u, 0u}; st.vf1[db][0] = z; st.vf1[db][1] = z; }
    };
    // second half of softmax(block 1) and PV1 of the last computed tile; leaves the pipeline empty
    auto drain = [&]() {
#pragma unroll
      for (int c = 0; c < 8; ++c) soft2<T, true>(st.s1[c >> 2], st.w1[c >> 2], st.pb1[c >> 2], l[c >> 2], 4 + (c & 3));
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int r = 0; r < 2; ++r) o[r][db] = TR::mfma32(st.vf1[db][ks], st.pb1[r].v[ks], o[r][db]);
      reset_pipe();
    };
    reset_pipe();

    // stage store of tile t+1 during tile t, global loads of tile t+2, one barrier per tile (see fwd_kernel)
    uint8_t mb = 1;
    u32x4 kf[2][G::KS];
    if (nt > 0) {
      sk.load(kbase, p.k.sn, p.M);
      sv.load(vbase, p.v.sn, p.M);
      if (mrow) mb = lane < p.M ? mrow[lane] : (uint8_t)0;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on ALL paths (see fwd_kernel)
    if (nt > 0) {
      sk.store(smem, tid);
      sv.store(smem + TILE_B, tid);
      sk.load(kbase + (int64_t)BN * p.k.sn, p.k.sn, nt > 1 ? p.M - BN : 0);
      sv.load(vbase + (int64_t)BN * p.v.sn, p.v.sn, nt > 1 ? p.M - BN : 0);
    }
    __syncthreads();
    if (nt > 0) {
#pragma unroll
      for (int jb = 0; jb < 2; ++jb)
#pragma unroll
        for (int kk = 0; kk < G::KS; ++kk) kf[jb][kk] = fa.row_frag(smem, 32 * jb, kk);
    }

    int t_split = 0;                      // tiles [0, t_split): no masking for this wave
    if (mrow == nullptr) {
      t_split = p.M / BN;
      if (p.causal) t_split = min(t_split, max(0, mw + diff + 1) / BN);
      t_split = min(t_split, nt);
    }
    auto run = [&](auto masked_tag, int t_begin, int t_end) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      for (int t = t_begin; t < t_end; ++t) {
        const int j0 = t * BN;
        const char* vcur = smem + (t & 1) * 2 * TILE_B + TILE_B;
        char* knxt = smem + ((t + 1) & 1) * 2 * TILE_B;
        uint64_t word = 0;
        if constexpr (MASKED) {
          word = __ballot((j0 + lane) < p.M && mb != 0);
          if (mrow && t + 1 < nt) {
            const int key = j0 + BN + lane;
            mb = key < p.M ? mrow[key] : (uint8_t)0;
          }
        }
        // All branch-free: a conditional would put the instruction into its own basic block, out of its issue slot.
        // Past the last tile the stores fill the LDS buffer nobody reads any more, the loads carry a zero-record
        // descriptor (no memory access, zeros returned) and the K requests read that unused buffer.
        const int rows_next = t + 2 < nt ? p.M - (j0 + 2 * BN) : 0;
        const __amdgpu_buffer_rsrc_t kd = Stager<T, D, BN, NT>::descriptor(kbase + (int64_t)(j0 + 2 * BN) * p.k.sn, p.k.sn, rows_next);
        const __amdgpu_buffer_rsrc_t vd = Stager<T, D, BN, NT>::descriptor(vbase + (int64_t)(j0 + 2 * BN) * p.v.sn, p.v.sn, rows_next);
        constexpr int PER = Stager<T, D, BN, NT>::PER;
        auto stage_store = [&](int x) { if (x < PER) sk.store_one(knxt, tid, x); else sv.store_one(knxt + TILE_B, tid, x - PER); };
        auto stage_load = [&](int x) { if (x < PER) sk.load_one(kd, x); else sv.load_one(vd, x - PER); };
        auto mid = [&]() { __syncthreads(); };
        if constexpr (MASKED) {
          if (p.causal && j0 > mw + RW - 1 + diff) {      // every key of this and all later tiles is in this wave's future:
            drain();                                      // finish what is in flight (idempotent), then only keep the
#pragma unroll
            for (int x = 0; x < 2 * PER; ++x) stage_store(x);   // workgroup's staging and barrier protocol going
#pragma unroll
            for (int x = 0; x < 2 * PER; ++x) stage_load(x);
            __syncthreads();
            continue;
          }
        }
        fwd2_tile<T, D, MASKED>(vcur, knxt, kf, fa, qf, o, l, st, p, word, ncm, i0, j0, diff, ts, stage_store, stage_load, mid);
        if constexpr (!MASKED) ts.close(5);
      }
    };
    run(std::false_type{}, 0, t_split);
    run(std::true_type{}, t_split, nt);
    drain();


#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i = i0 + 32 * r;
      const float inv = 1.f / fmaxf(xhalf_sum(l[r]), p.l_eps);
      if (i < p.N) {
        if (p.inv_l != nullptr && fa.hi == 0) p.inv_l[((int64_t)b * p.H + h) * p.N + i] = inv;
        char* orow = p.o.p + (int64_t)b * p.o.sb + (int64_t)h * p.o.sh + (int64_t)i * p.o.sn;
        store_row_tile<T, D>(orow, o[r], inv, fa.hi, false);
      }
    }
  }   // pass
#ifdef FCSA_TRACE
  if (blockIdx.x == gridDim.x / 2 + 3 && (tid & 63) == 0 && wave < 4) ts.dump(g_trace_fwd + 32 * wave, trace_now() - trace_t0);
#endif
}

// Wide or narrow forward kernel (measured on MI355X, bf16, B4 H8: tools/fwd_ab.py, tools/form_sweep.py).  The wide kernel needs enough
// 256-row workgroups to cover the 256 CUs.  Rounds 2 - 3 measured it ahead where the MFMA share of a tile is large or the sequence is
// long (D = 32 / 64 non-causal: +3..22%; causal N = 8192: +5%; D = 96 against the ONE-wave narrow kernel of round 2: +25..34%); with
// causal masking and short sequences its 256-row diagonal granularity costs more than the halved LDS traffic saves (N = 4096: -6%,
// N = 1024: -20%).  Re-measured in round 6, after the 32-row kernel's round-4 gains: see use_wide_fwd.
template <int D>
static bool use_wide_fwd(const FwdParams& p) {
  const int MT = (p.N + 255) / 256;
  const int64_t wgs = (int64_t)p.B * p.H * (p.causal ? (MT + 1) / 2 : MT);
  if (wgs < 224) return false;
  // Round 6 (tools/form_sweep.py, profiles/r06_form_sweep_*.txt): since round 4 (row sums on the VALU, LDS-DMA staging) the 32-row kernel
  // at two waves per SIMD beats this one at D = 64 -- non-causal (4,8,4096) 135 vs 156 us, (8,8,2048) 74 vs 88, causal (4,8,8192) 273 vs
  // 324 -- and at D = 16 (causal 8192: 155 vs 165); at D = 32 this kernel still wins on long key ranges (causal (4,8,8192) 183 vs 198,
  // non-causal (2,8,8192) 172 vs 178; level at 2048 - 4096 keys, 7 % behind at 1024).  D = 96: the lean two-wave kernel (round 3).
  if (D != 32) return false;
  return p.causal ? p.N >= 8192 : p.M >= 4096;
}

template <typename T, int D>
static hipError_t launch_fwd2(const FwdParams& p, hipStream_t s) {
  constexpr int NW = 4, BM = 64 * NW;
  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  const size_t lds = 4 * 64 * TileGeom<D, 2>::ROWB;
  auto kern = fwd2_kernel<T, D, NW>;
  static std::atomic<uint64_t> lds_ok{0};
  if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.H * PT)), dim3(NW * 64), lds, s, p);
  return hipGetLastError();
}

// Waves per workgroup of the row-tile kernels: 8 (one 256-row workgroup per CU) when that still gives every CU a
// workgroup, else 4 (two 128-row workgroups per CU).  Both keep two waves per SIMD; with 8 the K / V tiles are
// staged once per CU instead of twice, i.e. half the global loads and LDS stores per wave (C3: forward -6%).
// tail: also apply the last-round rule of rows <= 128 bytes (below)
static int row_tile_waves(int64_t batch_heads, int rows, bool causal, bool bits16 = false, bool tail = false) {
  const int MT = (rows + 255) / 256, cus = cu_count();
  const int64_t w256 = batch_heads * (causal ? (MT + 1) / 2 : MT);
  // More 256-row workgroups than CUs, 16-bit rows <= 128 bytes (round 6, profiles/r06_form_sweep_big*.txt): the LAST round decides.  A
  // last round that fills at most ~55 % of the CUs costs the 8-wave form a whole 256-row workgroup time; as 4-wave workgroups (two per
  // CU) the same tail is 128-row workgroups running alone on their CUs: 264 ... 384 and 544 ... 640 workgroups on 256 CUs -4 ... -11 %,
  // 800 and 1088 -6 ... -8 %.  Full or nearly full last rounds (C3: exactly 256) keep the 8-wave form (K / V staged once per CU).
  if (bits16 && tail && w256 > cus) {
    const int64_t rem = w256 % cus;
    return (rem != 0 && rem * 20 <= (int64_t)cus * 11) ? 4 : 8;
  }
  if (w256 >= cus * 7 / 8) return 8;
  // 16-bit types (round 6, tools/form_sweep.py): once the 128-row tiles outnumber the CUs -- where the key-split 8-wave form would need a
  // second round of workgroups -- the 256-row 8-wave workgroup wins from 132 workgroups on 256 CUs up, not only from 7/8 of the CUs:
  // rows <= 128 bytes against two 4-wave workgroups per CU -5 ... -9 % (profiles/r06_form_sweep_d64_b.txt), D = 96 / 128 (lean form)
  // against the key-split form -25 ... -35 % (profiles/r06_form_sweep_d128_b.txt)
  if (bits16) {
    const int MT4 = (rows + 127) / 128;
    if (batch_heads * (causal ? (MT4 + 1) / 2 : MT4) > cus) return 8;
  }
  return 4;
}

// Key-split form (fwd_kernel<.., KSPLIT>): 128-row workgroups of 8 waves.  Where the 128-row four-wave workgroups would leave the SIMDs
// with one wave each: rows wider than 128 bytes always (that form runs one wave per SIMD whatever the grid), narrower rows when the grid
// has fewer than ~1.5 workgroups per CU.
template <typename T, int D, bool BIAS> constexpr bool fwd_ksplit() {
  return Traits<T>::ES == 2 && (BIAS ? (D == 64 || D == 32 || D == 16) : (D == 64 || D == 96 || D == 128 || D == 32 || D == 16)) &&
         (64 * TileGeom<D, Traits<T>::ES>::ROWB) % 1024 == 0;
}
template <typename T, int D>
static bool use_ksplit_fwd(const FwdParams& p) {
  const int MT = (p.N + 127) / 128;
  const int64_t wgs = (int64_t)p.B * p.H * (p.causal ? (MT + 1) / 2 : MT) * (p.splits > 1 ? p.splits : 1);
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only (tools/split_sweep.py, dev/fcsa_sweep_env.h): FCSA_KSPLIT = 0 / 1 forces the form
  if (const int v = fcsa_dev::env_int("FCSA_KSPLIT"); v >= 0) return v != 0;
#endif
  if (D * Traits<T>::ES > 128) return true;
  return wgs <= cu_count();
}

// Split-key forward, second step: O = (sum_s partial P~V) / max(sum_s partial l, eps), inv_l alike.  One thread per
// (row, 8 features); the partials are a few MB and L2-resident.
template <typename T, int D>
__global__ void __launch_bounds__(256) fwd_combine_kernel(const FwdParams p) {
  constexpr int CH = D / 8;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t rows = (int64_t)p.B * p.H * p.N;
  if (idx >= rows * CH) return;
  const int64_t row = idx / CH;
  const int c = (int)(idx - row * CH);
  float acc[8], lt = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  // four splits' loads in flight before the first add (the loop over a run-time count left one dependent triple of loads in flight: the whole
  // kernel is this loop); same summation order, bit-identical (round 6)
  for (int s0 = 0; s0 < p.splits; s0 += 4) {
    f32x4 a[4], b2[4];
    float lv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t prow = (int64_t)min(s0 + u, p.splits - 1) * rows + row;      // (clamped: a repeated load, not added)
      lv[u] = p.ws_l[prow];
      a[u] = *reinterpret_cast<const f32x4*>(p.ws_o + prow * D + 8 * c);
      b2[u] = *reinterpret_cast<const f32x4*>(p.ws_o + prow * D + 8 * c + 4);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (s0 + u < p.splits) {
        lt += lv[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc[e] += a[u][e]; acc[4 + e] += b2[u][e]; }
      }
  }
  const float inv = 1.f / fmaxf(lt, p.l_eps);
  const int64_t bh = row / p.N;
  const int i = (int)(row - bh * p.N);
  const int b = (int)(bh / p.H), h = (int)(bh - (int64_t)b * p.H);
  char* orow = p.o.p + (int64_t)b * p.o.sb + (int64_t)h * p.o.sh + (int64_t)i * p.o.sn;
  if constexpr (Traits<T>::ES == 4) {
    f32x4 x = {acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv}, y = {acc[4] * inv, acc[5] * inv, acc[6] * inv, acc[7] * inv};
    reinterpret_cast<f32x4*>(orow + 32 * c)[0] = x;
    reinterpret_cast<f32x4*>(orow + 32 * c)[1] = y;
  } else {
    u32x4 u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = Traits<T>::pack2(acc[2 * e] * inv, acc[2 * e + 1] * inv);
    *reinterpret_cast<u32x4*>(orow + 16 * c) = u;
  }
  if (c == 0 && p.inv_l != nullptr) p.inv_l[row] = inv;
}

template <typename T, int D, bool BIAS, int NW, bool DYN, bool LEAN = false, bool KSPLIT = false>
static hipError_t launch_fwd_nw(const FwdParams& p, hipStream_t s) {
  constexpr int RWAVES = KSPLIT ? NW / 2 : NW, BM = 32 * RWAVES;
  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  size_t lds = 4 * 64 * (KSPLIT ? 2 : fwd_stage_tiles<T, D, DYN>()) * TileGeom<D, Traits<T>::ES>::ROWB;      // 2 buffers x (K + V tiles of a stage)
  if (lds < (size_t)RWAVES * RowEpilogue<T, D>::BYTES_NOX) lds = (size_t)RWAVES * RowEpilogue<T, D>::BYTES_NOX;   // epilogue scratch reuses the same bytes
  if (KSPLIT && lds < (size_t)RWAVES * 64 * 16 * (TileGeom<D, Traits<T>::ES>::DB * 4 + 1)) lds = (size_t)RWAVES * 64 * 16 * (TileGeom<D, Traits<T>::ES>::DB * 4 + 1);
  // two instantiations: causal launches (select per logit on the diagonal tiles) and the others (key masks as a rank-1 MFMA)
  const dim3 grid((unsigned)(p.B * p.H * PT), (unsigned)(p.splits > 1 ? p.splits : 1));
  if (p.causal) {
    auto kern = fwd_kernel<T, D, NW, BIAS, DYN, LEAN, false, KSPLIT>;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p);
  } else {
    auto kern = fwd_kernel<T, D, NW, BIAS, DYN, LEAN, true, KSPLIT>;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p);
  }
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (p.splits > 1) {
    const int64_t items = (int64_t)p.B * p.H * p.N * (D / 8);
    hipLaunchKernelGGL((fwd_combine_kernel<T, D>), dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

template <typename T, int D, bool BIAS>
static hipError_t launch_fwd_b(const FwdParams& p, hipStream_t s) {
  if (p.dyn) {                  // per-row exponent reference (online): the prefetching form, 8 waves where they fit two per SIMD
    if constexpr (D * Traits<T>::ES <= 128) {
      if (row_tile_waves((int64_t)p.B * p.H, p.N, p.causal, Traits<T>::ES == 2, true) == 8) return launch_fwd_nw<T, D, BIAS, 8, true>(p, s);
    } else if constexpr (fwd_lean<T, D, BIAS>()) {
      if (row_tile_waves((int64_t)p.B * p.H, p.N, p.causal, true) == 8) return launch_fwd_nw<T, D, BIAS, 8, true, true>(p, s);
    }
    if constexpr (fwd_ksplit<T, D, BIAS>()) {
      if (use_ksplit_fwd<T, D>(p)) return launch_fwd_nw<T, D, BIAS, 8, true, !BIAS, true>(p, s);
    }
    return launch_fwd_nw<T, D, BIAS, 4, true>(p, s);
  }
  if (p.splits > 1) {                                                   // split-key path: 128-row tiles x key ranges
    if constexpr (fwd_ksplit<T, D, BIAS>()) {
      if (use_ksplit_fwd<T, D>(p)) return launch_fwd_nw<T, D, BIAS, 8, false, !BIAS, true>(p, s);
    }
    return launch_fwd_nw<T, D, BIAS, 4, false>(p, s);
  }
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only (tools/form_sweep.py): FCSA_FWD_FORM = 1 row tiles of 8 waves, 2 key-split 8 waves, 3 four waves
  if constexpr (D * Traits<T>::ES <= 128 && fwd_ksplit<T, D, BIAS>()) {
    const int f = fcsa_dev::env_int("FCSA_FWD_FORM");
    if (f == 1) return launch_fwd_nw<T, D, BIAS, 8, false>(p, s);
    if (f == 2) return launch_fwd_nw<T, D, BIAS, 8, false, !BIAS, true>(p, s);
    if (f == 3) return launch_fwd_nw<T, D, BIAS, 4, false>(p, s);
  } else if constexpr (fwd_lean<T, D, BIAS>() && fwd_ksplit<T, D, BIAS>()) {      // 16-bit D = 96 / 128: 1 = lean 256-row tiles, 2 = lean key-split, 3 = four waves (one per SIMD)
    const int f = fcsa_dev::env_int("FCSA_FWD_FORM");
    if (f == 1) return launch_fwd_nw<T, D, BIAS, 8, false, true>(p, s);
    if (f == 2) return launch_fwd_nw<T, D, BIAS, 8, false, !BIAS, true>(p, s);
    if (f == 3) return launch_fwd_nw<T, D, BIAS, 4, false>(p, s);
  }
#endif
  if constexpr (D * Traits<T>::ES <= 128) {      // two waves per SIMD whatever the grid (<= 256 registers with all prefetches)
    if (row_tile_waves((int64_t)p.B * p.H, p.N, p.causal, Traits<T>::ES == 2, true) == 8) return launch_fwd_nw<T, D, BIAS, 8, false>(p, s);
  } else if constexpr (fwd_lean<T, D, BIAS>()) {
    // the lean form needs its partner wave: one 8-wave workgroup per CU (a grid with two 4-wave workgroups per CU always has that)
    if (row_tile_waves((int64_t)p.B * p.H, p.N, p.causal, true) == 8) return launch_fwd_nw<T, D, BIAS, 8, false, true>(p, s);
  }
  if constexpr (fwd_ksplit<T, D, BIAS>()) {
    if (use_ksplit_fwd<T, D>(p)) return launch_fwd_nw<T, D, BIAS, 8, false, !BIAS, true>(p, s);
  }
  return launch_fwd_nw<T, D, BIAS, 4, false>(p, s);
}

template <typename T, int D>
static hipError_t launch_fwd_t(const FwdParams& p, hipStream_t s) {
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only (tools/form_sweep.py): FCSA_FWD_FORM = 4 the 64-rows-per-wave kernel, any other value > 0 never
  if constexpr (Traits<T>::ES == 2 && D <= 64) {
    if (const int f = fcsa_dev::env_int("FCSA_FWD_FORM"); f > 0) {
      if (f == 4 && p.bias == nullptr && !p.dyn && p.splits <= 1 && p.mask == nullptr) return launch_fwd2<T, D>(p, s);
      return p.bias != nullptr ? launch_fwd_b<T, D, true>(p, s) : launch_fwd_b<T, D, false>(p, s);
    }
  }
#endif
  if constexpr (Traits<T>::ES == 2 && D == 32) {       // (use_wide_fwd: the only head dim fwd2_kernel still wins at -- nothing else instantiates it)
    if (p.bias == nullptr && !p.dyn && p.splits <= 1 && p.mask == nullptr && use_wide_fwd<D>(p)) return launch_fwd2<T, D>(p, s);
  }
  return p.bias != nullptr ? launch_fwd_b<T, D, true>(p, s) : launch_fwd_b<T, D, false>(p, s);
}

template <typename T>
static hipError_t launch_fwd_d(int D, const FwdParams& p, hipStream_t s) {
#ifdef FCSA_DEV_ONLY      // development builds: one instantiation (bf16, D = 64)
#ifndef FCSA_DEV_D
#define FCSA_DEV_D 64
#endif
  if constexpr (std::is_same<T, BF16>::value) { if (D == FCSA_DEV_D) return launch_fwd_t<T, FCSA_DEV_D>(p, s); }
  return hipErrorInvalidValue;
#else
  switch (D) {
    case 16:  return launch_fwd_t<T, 16>(p, s);
    case 32:  return launch_fwd_t<T, 32>(p, s);
    case 64:  return launch_fwd_t<T, 64>(p, s);
    case 96:  return launch_fwd_t<T, 96>(p, s);
    case 128: return launch_fwd_t<T, 128>(p, s);
    default:  return hipErrorInvalidValue;
  }
#endif
}

hipError_t launch_forward(int dtype, int D, const FwdParams& p, hipStream_t s) {
  if (p.B * p.H == 0 || p.N == 0) return hipSuccess;
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only: FCSA_FWD_FORM = 5 forces the D = 128 64-rows-per-wave kernel whatever the grid, 1 .. 4 never take it
  if (const int f = fcsa_dev::env_int("FCSA_FWD_FORM"); f > 0) {
    if (f == 5 && D == 128 && (dtype == 1 || dtype == 2) && p.bias == nullptr && p.mask == nullptr && !p.dyn && p.splits <= 1) return launch_forward_wide128(dtype, p, s);
  } else
#endif
  if (use_forward_wide128(dtype, D, p)) return launch_forward_wide128(dtype, p, s);
  if (dtype == 2) return launch_fwd_d<BF16>(D, p, s);
  if (dtype == 1) return launch_fwd_d<F16>(D, p, s);
  if (dtype == 0) return launch_fwd_d<F32>(D, p, s);
  return hipErrorInvalidValue;
}

}  // namespace fcsa
