// fcsa_common.cuh -- device building blocks shared by the gfx950 cosine-sim attention kernels.
//
// Everything here is written for CDNA4 (wave64, v_mfma_f32_32x32x16_{bf16,f16},
// v_mfma_f32_32x32x2_f32, ds_read_b64_tr_b16, 160 KiB LDS with 64 dword banks).  It replaces the
// reference's mem::shared_fragment (cu:89-258), mma::warp_tile (cu:604-1067), rowsum_accumulator
// (cu:262-316) and layout:: tables (cu:320-597) with a different decomposition:
//
//   * One wave owns 32 sequence positions ("row-per-lane"): every tile product is issued so
//     that the MFMA C operand has  column = lane & 31 = the wave's own sequence position  and
//     rows = the other index.  Per-row scalars (row sum, 1/l, delta, norms) are then per-lane
//     scalars and the row reductions are a single cross-half add (lane ^ 32).
//   * C layout of the 32x32 MFMAs (guide §3, dtype independent): value r of lane l is
//         row  crow(r, l>>5) = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),   col = l & 31.
//   * 16-bit operands (v_mfma_f32_32x32x16): lane l holds 8 consecutive k of row/col (l & 31):
//     k = 8 * (l >> 5) + e.  A 32x32 f32 result becomes the 16-bit B operand of the NEXT product
//     without any cross-lane traffic: registers r = 8*ks + e (e = 0..7) of a lane are exactly
//     k-slot (l>>5, e) of k-step ks if the other operand enumerates the contraction index as
//         idx(ks, hi, e) = 16*ks + 8*(e >> 2) + 4*hi + (e & 3)        ( = crow(8*ks + e, hi) ).
//     That other operand comes from a row-major LDS tile through two ds_read_b64_tr_b16
//     (rows idx(ks,hi,0..3) and idx(ks,hi,4..7)).
//   * f32 operands (v_mfma_f32_32x32x2, exact f32): lane l holds ONE k = l >> 5 per instruction.
//     A 16-byte row fragment (4 floats: features 8*kk + 4*hi + t) feeds 4 instructions (t = 0..3);
//     the second product takes register t of the f32 result directly as B and one ds_read_b32
//     (row crow(t, hi) of the LDS tile) as A -- no conversion, no transposed read.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fcsa_kernels.h"

namespace fcsa {

typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef float    f32x2  __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2  __attribute__((ext_vector_type(2)));
typedef short    s16x4  __attribute__((ext_vector_type(4)));
typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16   bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2  __attribute__((ext_vector_type(2)));
typedef _Float16 f16x4  __attribute__((ext_vector_type(4)));

#define FCSA_DEV __device__ __forceinline__

// NOTE: always pass vector ELEMENTS through this by-value helper.  `__builtin_bit_cast(float, v[t])`
// applied directly to an ext_vector element lvalue is miscompiled by hipcc 7.2 (it reads element 0
// for every t: observed as four v_mfma_f32_32x32x2_f32 with identical operand registers).
FCSA_DEV float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }

// ---------------------------------------------------------------------------------------------
// dtype traits
// ---------------------------------------------------------------------------------------------
struct BF16 {};
struct F16 {};
struct F32 {};

template <typename T> struct Traits;

template <> struct Traits<BF16> {
  typedef __bf16 elem;
  static constexpr int ES = 2;
  static FCSA_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static FCSA_DEV uint32_t pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
  }
  static FCSA_DEV float lo(uint32_t u) { return as_f32(u << 16); }
  static FCSA_DEV float hi(uint32_t u) { return as_f32(u & 0xffff0000u); }
  static constexpr uint32_t kOne2 = 0x3f803f80u;     // two packed 1.0
  // acc + lo(u) + hi(u): one v_dot2c_f32_bf16 against packed ones (exact f32 sum of the two ROUNDED values)
  static FCSA_DEV float add_pair(uint32_t u, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, u), __builtin_bit_cast(bf16x2, kOne2), acc, false);
  }
};

template <> struct Traits<F16> {
  typedef _Float16 elem;
  static constexpr int ES = 2;
  static FCSA_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static FCSA_DEV uint32_t pack2(float a, float b) {
    f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
  }
  static FCSA_DEV float lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
  static FCSA_DEV float hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
  static constexpr uint32_t kOne2 = 0x3c003c00u;     // two packed 1.0
  static FCSA_DEV float add_pair(uint32_t u, float acc) {      // v_dot2c_f32_f16 against packed ones
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, u), __builtin_bit_cast(f16x2, kOne2), acc, false);
  }
};

template <> struct Traits<F32> {
  typedef float elem;
  static constexpr int ES = 4;
  static constexpr uint32_t kOne2 = 0x3f800000u;     // (unused: the packed-ones tricks are 16-bit only)
  // 4 x v_mfma_f32_32x32x2_f32 over the 4 floats of a 16-byte fragment
  static FCSA_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(as_f32(a[t]), as_f32(b[t]), c, 0, 0, 0);
    return c;
  }
};

// Key mask as ONE small MFMA per 32-key block instead of a select per logit.  A key mask (and the zero-filled keys past the end of
// a ragged last tile) is the same for every query: S^T[j][i] += m_j with m_j = 0 (valid) or -inf (masked) is a rank-1 update, i.e.
// one more k-step of the S chain whose A operand carries m_j in k-slot 0 of key row j and whose B operand carries 1.0 in k-slot 0
// of every query column (v_mfma_f32_32x32x8_{bf16,f16}: 2 + 2 operand registers; f32: v_mfma_f32_32x32x2_f32).  2^-inf == 0 then
// removes the key from P~, from the row sum and from the running max without a single VALU instruction per logit -- the
// per-element form (bit test, compare, select, and a split cvt_pk + v_perm per pair) was 4 extra VALU instructions per logit:
// 181 instead of 37 per 64-key tile of the forward, 167 instead of 59 in dQ.  Causal masks are not rank-1 and keep the select
// (they only touch the tiles on the diagonal).  `valid_bits`: wave-uniform validity bits of the block's 32 keys; rows = keys
// (lane & 31) in the forward and dQ (S^T = K Q^T).
template <typename T> FCSA_DEV f32x16 key_mask_rank1(f32x16 c, uint32_t valid_bits, int x /* lane & 31 */, int hi /* lane >> 5 */) {
  const bool first = hi == 0;
  const bool m = first && ((valid_bits >> x) & 1u) == 0u;
  if constexpr (Traits<T>::ES == 4) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(m ? -INFINITY : 0.f, first ? 1.f : 0.f, c, 0, 0, 0);
  } else if constexpr (__is_same(typename Traits<T>::elem, _Float16)) {
    const f16x4 a = {m ? (_Float16)(-INFINITY) : (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const f16x4 b = {first ? (_Float16)1.f : (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
  } else {
    const s16x4 a = {(short)(m ? 0xFF80 : 0), 0, 0, 0};            // bf16 -inf
    const s16x4 b = {(short)(first ? 0x3F80 : 0), 0, 0, 0};        // bf16 1.0
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
  }
}

// The same for kernels whose lanes (C columns) are the keys (dK/dV: S = Q K^T): ones in the A operand's k-slot 0 (every query row),
// this lane's 0 / -inf in the B operand's.
template <typename T> FCSA_DEV f32x16 key_mask_rank1_cols(f32x16 c, bool key_masked, int hi /* lane >> 5 */) {
  const bool first = hi == 0;
  const bool m = first && key_masked;
  if constexpr (Traits<T>::ES == 4) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(first ? 1.f : 0.f, m ? -INFINITY : 0.f, c, 0, 0, 0);
  } else if constexpr (__is_same(typename Traits<T>::elem, _Float16)) {
    const f16x4 a = {first ? (_Float16)1.f : (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    const f16x4 b = {m ? (_Float16)(-INFINITY) : (_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
    return __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, c, 0, 0, 0);
  } else {
    const s16x4 a = {(short)(first ? 0x3F80 : 0), 0, 0, 0};
    const s16x4 b = {(short)(m ? 0xFF80 : 0), 0, 0, 0};
    return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a, b, c, 0, 0, 0);
  }
}

// Wave priority inside a barrier interval.  The two waves of a SIMD are arbitrated by priority, then AGE: the first-dispatched half of
// an 8-wave workgroup (waves 0-3) wins every conflict, finishes its share of a barrier interval early and then idles ~30 % of the tile
// loop at the barrier (-DFCSA_TRACE_BAR, tools/trace_bar.py: waves 0-3 wait 29 - 33 % of the dK/dV and dQ loops, waves 4-7 4 - 6 %) while
// its partner, which was held back, finishes alone at the one-wave rate.  kPrioBwd: the younger half runs at priority 1 during the
// FIRST half of its blocks of an interval and at 0 afterwards, so each half is favoured for about half of the interval and both reach
// the barrier together (waits 7 - 9 % / 6 %; tile loops of dK/dV -10.7 %, of dQ -4.1 % in clocks).  The forward keeps age order
// (kPrioFwd = 0): its interval is one 64-key tile with the barrier in its middle, and the balanced form measured +10 % clocks there --
// the old half running ahead is what puts its MFMA phases beside the young half's exponentials.
constexpr int kPrioLean = 0;      // the same idea in the lean (16-bit D = 96 / 128, two waves per SIMD) dK/dV form
constexpr int kPrioBwd = 1;
constexpr int kPrioFwd = 0;

// A value the optimiser must treat as freshly computed here: keeps per-lane address arithmetic of prologues / epilogues from being
// hoisted out of the pass loop, where it would stay live across the tile loops and push the kernels over their register budget
// (observed: 10 hoisted address pairs spilled to scratch and reloaded -- a memory round trip each -- in the dQ epilogue).
FCSA_DEV int opaque(int v) { asm volatile("" : "+v"(v)); return v; }

// Bias values of one 32-key block for the row this lane owns (kernels whose lanes are query rows: forward, dQ): register r <->
// key jbase + crow(r, 0), i.e. four groups of 4 consecutive keys.  `row` points at key 0 of a valid bias row, keys >= m_lim are
// clamped (their logits are masked by the caller).  A group that lies inside the row is ONE 8- / 16-byte load when the rows keep
// that alignment (m % 4 == 0: jbase is a multiple of 4) -- the element-wise form issued 16 two-byte loads per block and lane and
// made the bias path three times slower than the plain one.
// The 16-bit fast path in two halves, so that a kernel can issue the loads of the NEXT tile's blocks a tile ahead (forward: the L2
// round trip of a bias row chunk is longer than one block's S chain, and a one-wave-per-SIMD kernel has nothing else to cover it):
// bias_raw_request = the two 16-byte loads (keys blk + 16 * hi .. + 15 of this lane's row), bias_raw_finish = exchange + convert.
template <typename T> FCSA_DEV void bias_raw_request(u32x4 (&raw)[2], const char* row, int blk, int hi) {
  const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const typename Traits<T>::elem*>(row) + blk + 16 * hi);
  raw[0] = src[0];
  raw[1] = src[1];
}
template <typename T> FCSA_DEV void bias_raw_finish(float (&bv)[16], const u32x4 (&raw)[2], float mul) {
  uint32_t w[4][2];      // [rq][dword]: keys 8 * rq + 4 * hi + (0, 1 | 2, 3)
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const auto sa = __builtin_amdgcn_permlane32_swap(raw[0][d], raw[0][2 + d], false, false);
    const auto sb = __builtin_amdgcn_permlane32_swap(raw[1][d], raw[1][2 + d], false, false);
    w[0][d] = sa[0];
    w[2][d] = sa[1];
    w[1][d] = sb[0];
    w[3][d] = sb[1];
  }
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    bv[4 * rq + 0] = Traits<T>::lo(w[rq][0]) * mul;
    bv[4 * rq + 1] = Traits<T>::hi(w[rq][0]) * mul;
    bv[4 * rq + 2] = Traits<T>::lo(w[rq][1]) * mul;
    bv[4 * rq + 3] = Traits<T>::hi(w[rq][1]) * mul;
  }
}

template <typename T> FCSA_DEV void load_bias_block(float (&bv)[16], const char* row, int jbase, int m_lim, bool rows_aligned, float mul,
                                                    bool rows_aligned8 = false, int hi = 0) {
  typedef typename Traits<T>::elem E;
  const E* r0 = reinterpret_cast<const E*>(row);
  if constexpr (Traits<T>::ES == 2) {
    // 16-bit rows that keep 16-byte alignment (m % 8 == 0), whole 32-key block inside the row (wave-uniform test; needs every
    // lane active): lane (row, hi) fetches the 16 CONSECUTIVE keys 16 * hi .. + 15 of the block as two 16-byte loads -- half the
    // memory instructions, and 32 instead of 16 useful bytes per row and instruction -- and the lane pair (lane, lane ^ 32)
    // exchanges the halves it fetched for each other with four v_permlane32_swap (lanes 32..63 of the first register <-> lanes
    // 0..31 of the second).
    const int jb = jbase - 4 * hi;
    if (rows_aligned8 && jb + 31 < m_lim) {
      u32x4 raw[2];
      bias_raw_request<T>(raw, row, jb, hi);
      bias_raw_finish<T>(bv, raw, mul);
      return;
    }
  }
#pragma unroll
  for (int rq = 0; rq < 4; ++rq) {
    const int j = jbase + 8 * rq;
    if (rows_aligned && j + 3 < m_lim) {
      if constexpr (Traits<T>::ES == 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(r0 + j);
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[4 * rq + e] = v[e] * mul;
      } else {
        const u32x2 v = *reinterpret_cast<const u32x2*>(r0 + j);
        bv[4 * rq + 0] = Traits<T>::lo(v[0]) * mul;
        bv[4 * rq + 1] = Traits<T>::hi(v[0]) * mul;
        bv[4 * rq + 2] = Traits<T>::lo(v[1]) * mul;
        bv[4 * rq + 3] = Traits<T>::hi(v[1]) * mul;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) bv[4 * rq + e] = (float)r0[min(j + e, m_lim - 1)] * mul;
    }
  }
}

// Bias values of one block for kernels whose LANES ARE KEYS (dK/dV): register r of lane (key j, hi) needs bias[query crow(r, hi)][j]
// -- sixteen different rows, two bytes each.  Loaded that way it is 16 global loads per block and lane (the bias form of the dKV
// kernel ran 5x slower than the plain one).  Here the wave fetches the [32 queries x 32 keys] block with lanes as QUERY rows
// (lane (query, hi) takes the 16 keys 16 * hi .. + 15 of its row: 16-byte loads), writes it to a private LDS scratch and reads it
// back transposed.  The raw loads are issued one block ahead (request), so their latency overlaps a block of work.
template <typename T> struct BiasBlock {
  static constexpr int ES = Traits<T>::ES;
  static constexpr int KB = 32 * ES;                 // bytes of 32 keys
  static constexpr int PITCH = KB + KB / 4;          // 16 bit: 80, f32: 160 -> the two lane halves (rows 4 apart) hit disjoint banks
  static constexpr int BYTES = 32 * PITCH;           // scratch per wave
  static constexpr int NV = KB / 32;                 // 16-byte loads per lane
  u32x4 raw[NV];
  // blk = &bias[0][first key of the wave]; row = this lane's (clamped, valid) query row; hi = lane >> 5
  FCSA_DEV void request(const char* blk, int64_t row, int64_t row_pitch, int hi) {
    const char* src = blk + row * row_pitch + hi * (KB / 2);
#pragma unroll
    for (int v = 0; v < NV; ++v) raw[v] = *reinterpret_cast<const u32x4*>(src + 16 * v);
  }
  FCSA_DEV void stage(char* scr, int lane_) const {
    const int lane = opaque(lane_);      // (address recomputed per block: hoisted, it is one more register that lives across the tile loops)
    char* dst = scr + (lane & 31) * PITCH + (lane >> 5) * (KB / 2);
#pragma unroll
    for (int v = 0; v < NV; ++v) *reinterpret_cast<u32x4*>(dst + 16 * v) = raw[v];
  }
};

// row index (0..31) of accumulator register r for lane half hi
FCSA_DEV constexpr int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// <a, b> over the elements of two 16-byte fragments
template <typename T> FCSA_DEV float dot_frag(const u32x4& a, const u32x4& b) {
  float s = 0.f;
  if constexpr (Traits<T>::ES == 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) s += as_f32(a[e]) * as_f32(b[e]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) s += Traits<T>::lo(a[e]) * Traits<T>::lo(b[e]) + Traits<T>::hi(a[e]) * Traits<T>::hi(b[e]);
  }
  return s;
}

// multiply the elements of a 16-byte fragment by a scalar (prologue-only work)
template <typename T> FCSA_DEV u32x4 scale_frag(const u32x4& a, float c) {
  u32x4 o;
  if constexpr (Traits<T>::ES == 4) {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = __builtin_bit_cast(uint32_t, as_f32(a[e]) * c);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = Traits<T>::pack2(Traits<T>::lo(a[e]) * c, Traits<T>::hi(a[e]) * c);
  }
  return o;
}

// ---------------------------------------------------------------------------------------------
// LDS tile geometry: row-major [rows][D] tile of ES-byte elements, 16-byte chunks XOR-swizzled per
// row so that
//   (a) ds_read_b128 of one chunk column by 16 different rows      (MFMA row fragments) and
//   (b) ds_read_b64_tr_b16 of 4 consecutive rows x 64 contiguous B (transposed fragments, 16 bit)
// are both bank-conflict free (64 banks x 4 B; derivation in DESIGN.md §LDS).  ds_read_b32 of 32
// consecutive floats of one row (f32 second product) is conflict free under any chunk permutation.
// ---------------------------------------------------------------------------------------------
template <int D, int ES> struct TileGeom {
  static_assert(D == 16 || D == 32 || D == 64 || D == 96 || D == 128, "dim_head");
  static constexpr int CPR = D * ES / 16;            // valid 16-byte chunks per row
  static constexpr int KS = CPR / 2;                 // row-fragment steps over the feature dim (2 chunks each)
  static constexpr int DB = (D + 31) / 32;           // 32-wide feature blocks of an output tile
  static constexpr int RSC = CPR <= 2 ? 2 : CPR <= 4 ? 4 : CPR <= 8 ? 8 : CPR <= 16 ? 16 : 32;   // chunks per LDS row
  static constexpr int ROWB = RSC * 16;              // LDS row pitch in bytes

  static FCSA_DEV int swz(int row) {
    if constexpr (RSC == 2) return (row >> 3) & 1;
    else if constexpr (RSC == 4) return (row >> 2) & 3;
    else if constexpr (RSC == 8) { int t = (row >> 1) & 7; return ((t & 1) << 2) | (t >> 1); }
    else if constexpr (RSC == 16) return ((row & 3) << 2) | ((row >> 2) & 3);
    else return row & 15;
  }
  // byte offset of 16-byte chunk `chunk` of row `row`
  static FCSA_DEV int off(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
};

// Per-lane offsets for reading fragments out of a TileGeom tile.
//   row fragment (ds_read_b128): lane (x = l&31, hi) reads row (rbase + x), 16-byte chunk 2*kk + hi
//   transposed fragment, 16 bit (2 x ds_read_b64_tr_b16): lane (x, hi) receives, for feature column
//   c = 32*db + x, the 8 rows rbase + idx(0, hi, e), e = 0..7
//   scalar, f32 (ds_read_b32): lane (x, hi) reads element (rbase + crow(t, hi), 32*db + x)
template <typename T, int D> struct FragAddr {
  static constexpr int ES = Traits<T>::ES;
  typedef TileGeom<D, ES> G;
  int row_off;        // (l&31) * ROWB
  int row_swz;        // swz(l&31)
  int tr_off[2];      // 16 bit: byte offset (excluding rbase/db terms) for the two 4-row halves
  int tr_swz[2];      // 16 bit: swizzle of the row this lane ADDRESSES in each half
  int tr_col;         // 16 bit: chunk contribution of this lane: 2*((l>>4)&1) + ((l&3)>>1)   (+4*db at use)
  int sc_chunk;       // f32: chunk of column x within a 32-column block: x >> 2 (+8*db at use)
  int sc_byte;        // f32: byte inside the chunk: (x & 3) * 4
  int hi;

  FCSA_DEV void init(int lane) {
    const int x = lane & 31;
    hi = lane >> 5;
    row_off = x * G::ROWB;
    row_swz = G::swz(x);
    const int t = lane & 15;                 // position inside the 16-lane transpose group
    const int cg = (lane >> 4) & 1;          // which 16-column half of the 32-column block
    // this lane supplies the address of row (t>>2) of the 4x16 block, 8-byte piece (t&3)
    tr_col = ((D >= 32) ? 2 * cg : 0) + ((t & 3) >> 1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int r = 8 * half + 4 * hi + (t >> 2);      // row within a 16-row k-step
      tr_off[half] = r * G::ROWB + ((t & 1) << 3);
      tr_swz[half] = G::swz(r);                          // independent of the 16-row step and the 32-row block
    }
    const int xc = (D >= 32) ? x : (x & 15);             // D = 16: columns 16..31 do not exist, re-read 0..15
    sc_chunk = xc >> 2;
    sc_byte = (xc & 3) << 2;
  }
  // row fragment of row (rbase + x): rbase must be a multiple of 32
  FCSA_DEV u32x4 row_frag(const char* tile, int rbase, int kk) const {
    const int chunk = 2 * kk + hi;
    return *reinterpret_cast<const u32x4*>(tile + rbase * G::ROWB + row_off + ((chunk ^ row_swz) << 4));
  }
  // 16 bit: transposed fragment, rows rbase + idx(0, hi, e) (rbase multiple of 16), feature block db
  FCSA_DEV u32x4 tr_frag(const char* tile, int rbase, int db) const {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    u32x4 out;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int chunk = 4 * db + tr_col;
      const char* p = tile + rbase * G::ROWB + tr_off[half] + ((chunk ^ tr_swz[half]) << 4);
      s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
      u32x2 u = __builtin_bit_cast(u32x2, v);
      out[2 * half] = u[0];
      out[2 * half + 1] = u[1];
    }
    return out;
  }
  // f32: element (rbase + crow(t, hi), 32*db + x); rbase multiple of 32
  FCSA_DEV float scalar(const char* tile, int rbase, int t, int db) const {
    const int row = crow(t, 0) + 4 * hi;
    return *reinterpret_cast<const float*>(tile + (rbase + row) * G::ROWB + (((8 * db + sc_chunk) ^ G::swz(row)) << 4) + sc_byte);
  }
};

// ---------------------------------------------------------------------------------------------
// Second product of a pair:  acc[32 features x 32 own positions] += tile^T[features][rows] * P[rows][own]
// where P is a 32x32 f32 result held in the C layout.  16 bit: P is packed once (SecondB) and the tile
// operand comes through transposed reads; f32: P registers are used as they are.
// ---------------------------------------------------------------------------------------------
template <typename T> struct SecondB {
  u32x4 v[2];
  FCSA_DEV void prep(const f32x16& p) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[ks][e] = Traits<T>::pack2(p[8 * ks + 2 * e], p[8 * ks + 2 * e + 1]);
  }
};
template <> struct SecondB<F32> {
  f32x16 p;
  FCSA_DEV void prep(const f32x16& x) { p = x; }
};

template <typename T, int D>
FCSA_DEV f32x16 second_mma(f32x16 acc, const char* tile, int rbase, int db, const SecondB<T>& b, const FragAddr<T, D>& fa) {
  if constexpr (Traits<T>::ES == 4) {
#pragma unroll
    for (int t = 0; t < 16; ++t)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.scalar(tile, rbase, t, db), b.p[t], acc, 0, 0, 0);
  } else {
    acc = Traits<T>::mfma32(fa.tr_frag(tile, rbase, db), b.v[0], acc);
    acc = Traits<T>::mfma32(fa.tr_frag(tile, rbase + 16, db), b.v[1], acc);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------
// global -> registers -> LDS staging of a [ROWS][D] tile by NT threads (split issue / write, so the
// HBM/L2 latency hides under the MFMA phase in between: guide T14).
//
// Loads are buffer loads (guide T8): a 128-bit descriptor in SGPRs rebuilt per tile on the scalar unit
// (base = first row of the tile, num_records = bytes that remain in the tensor slice) plus a per-lane
// byte offset that is computed ONCE and stays live.  Two reasons:
//   * with flat/global loads the address VGPR pair is recomputed per tile and then recycled; hipcc guards
//     that write-after-read with s_waitcnt vmcnt(0) right before the first MFMA of the tile, i.e. the
//     prefetch it was meant to overlap is waited for immediately (seen in the ISA, ~1/3 of wave cycles);
//   * the hardware range check returns zeros past num_records, which is exactly the zero fill needed for
//     rows beyond the end of the sequence -- no predication, no clamping.
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int ROWS, int NT> struct Stager {
  typedef TileGeom<D, Traits<T>::ES> G;
  static constexpr int NCH = ROWS * G::CPR;
  static constexpr int PER = (NCH + NT - 1) / NT;
  static constexpr int ROW_BYTES = D * Traits<T>::ES;
  u32x4 r[PER];
  int voff[PER];       // loop-invariant byte offset of this lane's chunk i inside a tile

  FCSA_DEV void init(int64_t pitch, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      const int row = c / G::CPR, ch = c % G::CPR;
      voff[i] = (NCH % NT == 0 || c < NCH) ? (int)(row * pitch + ch * 16) : 0x7ffffff0;   // beyond any num_records
    }
  }
  // g: address of (row 0, feature 0) of the tile (wave-uniform); rows >= rows_valid read as zero
  FCSA_DEV void load(const char* g, int64_t pitch, int rows_valid) {
    int64_t bytes = rows_valid > 0 ? (int64_t)(rows_valid - 1) * pitch + ROW_BYTES : 0;
    if (bytes > 0x7fffffff) bytes = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, (int)bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < PER; ++i) r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], 0, 0);
  }
  FCSA_DEV void store(char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) store_one(tile, tid, i);
  }
  // one chunk at a time, for kernels that place each memory instruction in a chosen issue slot
  FCSA_DEV static __amdgpu_buffer_rsrc_t descriptor(const char* g, int64_t pitch, int rows_valid) {
    int64_t bytes = rows_valid > 0 ? (int64_t)(rows_valid - 1) * pitch + ROW_BYTES : 0;
    if (bytes > 0x7fffffff) bytes = 0x7fffffff;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(g), 0, (int)bytes, 0x00020000);
  }
  FCSA_DEV void load_one(__amdgpu_buffer_rsrc_t rsrc, int i) { r[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff[i], 0, 0); }
  FCSA_DEV void store_one(char* tile, int tid, int i) const {
    const int c = tid + i * NT;
    const int row = c / G::CPR, ch = c % G::CPR;
    if (NCH % NT == 0 || c < NCH) *reinterpret_cast<u32x4*>(tile + G::off(row, ch)) = r[i];
  }
};

// ---------------------------------------------------------------------------------------------
// global -> LDS staging WITHOUT registers: buffer_load_dwordx4 ... lds (LDS-DMA).  A wave instruction moves 64 x 16 bytes to
// 1 KiB of CONSECUTIVE LDS bytes (address = M0 base + lane * 16: no per-lane scatter), so the XOR swizzle of TileGeom moves
// to the SOURCE side: the lane that fills physical chunk c' of row r fetches logical chunk c' ^ swz(r) of that row (same
// 128/256-byte row, so the coalescing of the global read is unchanged).  Rows past the end of the tensor slice read as zero
// through the descriptor's range check, like the register path.  A piece is 1 KiB = 1024 / ROWB tile rows; wave w of NW owns
// pieces w, w + NW, ...  Completion is tracked by vmcnt; hipcc's __syncthreads() waits for it (vmcnt(0)) by itself.
// Compared with Stager: no staging registers (PER * 4 per tensor), no ds_write_b128 passes.
// ---------------------------------------------------------------------------------------------
template <typename T, int D, int ROWS, int NW> struct DmaStager {
  typedef TileGeom<D, Traits<T>::ES> G;
  static constexpr int NPIECE = ROWS * G::ROWB / 1024;                  // 1 KiB pieces per tile
  static_assert(ROWS * G::ROWB % 1024 == 0, "tile must be a whole number of 1 KiB pieces");
  static constexpr int PER = (NPIECE + NW - 1) / NW;                    // pieces per wave
  static constexpr int ROW_BYTES = D * Traits<T>::ES;
  // Source byte offset of this lane inside a tile, per piece.  Piece i of a wave starts NW * 1024 / ROWB rows below piece i - 1; when
  // that is a multiple of the swizzle period (16 rows for every TileGeom) the lane fetches the same chunk column in every piece and
  // the offsets are voff0 + i * vstep with a wave-uniform step: ONE register instead of PER (the wider kernels sit at their
  // register budget: the dQ kernel at D = 128 with two waves per SIMD spilled its offsets).
  static constexpr int ROWS_PER_STEP = NW * 1024 / G::ROWB;
  static constexpr bool UNIFORM = (NW * 1024) % G::ROWB == 0 && ROWS_PER_STEP % 16 == 0;
  int voff[UNIFORM ? 1 : PER];
  int vstep;             // UNIFORM: byte distance between consecutive pieces of this wave (wave-uniform)

  FCSA_DEV void init(int64_t pitch, int wave, int lane) {
    vstep = __builtin_amdgcn_readfirstlane((int)(ROWS_PER_STEP * pitch));
#pragma unroll
    for (int i = 0; i < (UNIFORM ? 1 : PER); ++i) {
      const int lds_off = (wave + i * NW) * 1024 + lane * 16;           // byte offset inside the LDS tile
      const int row = lds_off / G::ROWB, cp = (lds_off % G::ROWB) >> 4;
      const int c = cp ^ G::swz(row);
      // padding chunks of a row (D = 96: 12 of 16) are never read: point them at chunk 0 (valid, already fetched)
      voff[i] = (int)(row * pitch + (c < G::CPR ? c : 0) * 16);
    }
  }
  FCSA_DEV int piece_offset(int i) const { return UNIFORM ? voff[0] + i * vstep : voff[UNIFORM ? 0 : i]; }
  // The DMA is issued from inline asm on purpose: hipcc tracks a builtin LDS-DMA as a pending LDS write that may alias any later
  // ds_read of the same __shared__ array and drains it (s_waitcnt vmcnt(0)) in front of the next tile's first fragment read,
  // i.e. right after issuing it.  Hidden from the compiler, the transfer overlaps the whole tile; the caller waits for it with
  // dma_wait() before the barrier that publishes the buffer.
  // A Stream is the wave-uniform (SGPR) state of one tensor slice that is walked tile by tile: a buffer descriptor of
  // [g0, g0 + bytes) made ONCE (per pass) and the byte offset of the current tile from g0.  Per tile the kernel only adds the
  // tile step to `off` (one SALU instruction) -- the scalar unit is shared by all waves of a CU, and rebuilding a descriptor per
  // tile (64-bit multiplies, range clamp, readfirstlanes: ~35 SALU instructions per tensor, times 8 waves, all right behind the
  // tile barrier) showed up as ~700 clocks per tile in the dKV phase trace.  Offsets are 32 bit: the kernel re-opens the stream
  // (rebase) when `off` passes 1 GiB; num_records is clamped to 2 GiB - 1.
  struct Stream { u32x4 rs; uint32_t off; };
  static constexpr uint32_t REBASE = 0x3fffffffu;
  FCSA_DEV Stream open(const char* g, int64_t pitch, int rows_valid) const {
    int64_t bytes = rows_valid > 0 ? (int64_t)(rows_valid - 1) * pitch + ROW_BYTES : 0;
    if (bytes > 0x7fffffff) bytes = 0x7fffffff;
    const uint64_t ga = reinterpret_cast<uint64_t>(g);
    Stream st;
    st.rs[0] = __builtin_amdgcn_readfirstlane((uint32_t)ga);
    st.rs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(ga >> 32) & 0xffffu);      // base[47:32], stride 0
    st.rs[2] = __builtin_amdgcn_readfirstlane((uint32_t)bytes);                       // num_records
    st.rs[3] = 0x00020000u;                                                           // raw buffer, 32-bit data format
    st.off = 0u;
    return st;
  }
  // LDS byte address of a tile (low 32 bits of the flat shared address), wave-uniform
  static FCSA_DEV uint32_t lds_addr(const char* tile) { return __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tile); }
  // piece i of this wave (i < PER) of the tile at st.off -> LDS tile at byte address lds_tile.  The DMA is issued from inline
  // asm on purpose (see above); M0 (LDS base of the DMA) is saved and restored around the statement.
  FCSA_DEV void issue_piece(const Stream& st, uint32_t lds_tile, int i, int wave) const {
    if (NPIECE % NW == 0 || wave + i * NW < NPIECE) {
      // (readfirstlane: no-ops where hipcc already knows the stream state to be wave-uniform; where a merge of two assignments
      //  made it lose track, they bring the descriptor back into SGPRs, which the instruction requires)
      u32x4 rs;
#pragma unroll
      for (int e = 0; e < 4; ++e) rs[e] = __builtin_amdgcn_readfirstlane(st.rs[e]);
      const uint32_t off = __builtin_amdgcn_readfirstlane(st.off), m0v = __builtin_amdgcn_readfirstlane(lds_tile + (uint32_t)(wave + i * NW) * 1024u);
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 4\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "s"(m0v), "v"(piece_offset(i) + off), "s"(rs) : "memory");
    }
  }
  FCSA_DEV void issue(const Stream& st, uint32_t lds_tile, int wave) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) issue_piece(st, lds_tile, i, wave);
  }
  // one-off form: g = address of (row 0, feature 0) of the tile (wave-uniform); rows >= rows_valid are zero-filled
  FCSA_DEV void issue(const char* g, int64_t pitch, int rows_valid, char* tile, int wave) const {
    issue(open(g, pitch, rows_valid), lds_addr(tile), wave);
  }
};
// all LDS-DMA transfers of this wave have landed (vmcnt also counts the compiler's own loads and stores: conservative)
FCSA_DEV void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Store the C-layout accumulators of a [32 own positions x D] tile: lane (row, hi) holds features
// 32*db + 8*rq + 4*hi + 0..3.  Output element type: float (as_f32 or T = F32) or the 16-bit T.
template <typename T, int D>
FCSA_DEV void store_row_tile(char* row, const f32x16 (&acc)[TileGeom<D, Traits<T>::ES>::DB], float mul, int hi, bool as_f32) {
#pragma unroll
  for (int db = 0; db < TileGeom<D, Traits<T>::ES>::DB; ++db)
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      if (32 * db + 8 * rq < D) {                 // compile-time after unrolling (D % 8 == 0)
        const int d0 = 32 * db + 8 * rq + 4 * hi;
        if (Traits<T>::ES == 4 || as_f32) {
          f32x4 v = {acc[db][4 * rq] * mul, acc[db][4 * rq + 1] * mul, acc[db][4 * rq + 2] * mul, acc[db][4 * rq + 3] * mul};
          *reinterpret_cast<f32x4*>(row + d0 * 4) = v;
        } else if constexpr (Traits<T>::ES == 2) {
          u32x2 v;
          v[0] = Traits<T>::pack2(acc[db][4 * rq] * mul, acc[db][4 * rq + 1] * mul);
          v[1] = Traits<T>::pack2(acc[db][4 * rq + 2] * mul, acc[db][4 * rq + 3] * mul);
          *reinterpret_cast<u32x2*>(row + d0 * 2) = v;
        }
      }
    }
}

// sum of a per-lane value over the two half-waves (lane ^ 32)
FCSA_DEV float xhalf_sum(float x) { return x + __shfl_xor(x, 32, 64); }

// ---------------------------------------------------------------------------------------------
// Row epilogue through LDS.  The C layout gives a lane 4 consecutive features of ITS row per (db, rq): stored directly that is
// 8 (or 16) row-strided 8-byte stores per lane and tensor -- every wave instruction touches 64 different rows, and the stores
// queue up behind each other at the end of the kernel when every wave of every CU is in its epilogue at once.  Measured on the
// dKV kernel (C3): 19 of 143 us were the epilogue (12.5 us the plain stores, 6.4 us the fused l2norm backward with its two
// 8-byte-per-lane reads of the normalised row).  Here the wave transposes its [32 rows x D] f32 tile through a private LDS
// scratch and then works in "row chunk" form: a lane owns 8 consecutive features (one 16-byte chunk of the 16-bit output) of
// a row, LPR consecutive lanes own a row, so every global access of a wave covers whole rows / whole 16-byte chunks:
//     16-bit D = 64: a wave instruction writes 8 complete 128-byte rows (1 KiB contiguous when the rows are).
// The l2norm backward  dx = r (g - x^ <g, x^>_group)  needs the group dot product across the 2^lgm lanes that hold a group:
// lgm DPP butterfly steps.  The caller guarantees that no other wave still uses `scr` (a barrier after the last tile) and
// synchronises again before the scratch region is overwritten by staging.
// ---------------------------------------------------------------------------------------------
template <typename T, int D> struct RowEpilogue {
  typedef Traits<T> TR;
  typedef TileGeom<D, TR::ES> G;
  static constexpr int CH = D / 8;                                                 // 8-feature chunks per row
  static constexpr int LPR = CH <= 2 ? 2 : CH <= 4 ? 4 : CH <= 8 ? 8 : 16;         // lanes per row (power of two >= CH)
  static constexpr int RPP = 64 / LPR;                                             // rows per pass
  static constexpr int NP = 32 / RPP;                                              // passes over the 32 rows
  static constexpr int PITCH = D * 4 + 16;                                         // scratch row pitch: +16 keeps the b128 writes conflict free
  static constexpr int XPITCH = D * TR::ES + 16;                                   // second scratch area: the wave's 32 normalised rows (16-bit types)
  static constexpr int XBYTES = TR::ES == 2 ? 32 * XPITCH : 0;
  static constexpr int BYTES_NOX = 32 * PITCH;                                     // scratch per wave without / with the second area
  static constexpr int BYTES = 32 * PITCH + XBYTES;

  // sum over the aligned block of (1 << steps) lanes this lane belongs to (steps <= 4)
  static FCSA_DEV float lane_block_sum(float v, int steps) {
    if (steps >= 1) v += as_f32(__builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    if (steps >= 2) v += as_f32(__builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    if (steps >= 3) v += as_f32(__builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror (quads are uniform)
    if (steps >= 4) v += as_f32(__builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror (halves are uniform)
    return v;
  }

  // The epilogue in three steps, so that a kernel can put the next pass's memory requests between them:
  //   load_inv  inverse norms of the rows this lane will finish (the only global reads of a fused epilogue with `xf`)
  //   put       acc (C layout, lane = row (lane & 31), hi = lane >> 5) * mul -> scratch; frees the accumulators
  //   finish    scratch -> (l2norm backward) -> out rows.  rows_valid: rows >= rows_valid are not stored.
  // xn0 != nullptr: fused l2norm backward against the normalised rows xn0 + row * xn_pitch (element type T, scaled by xn_scale)
  // with inverse norms inv_norm0[row * NG + group]; groups are (8 << lgm) features wide.
  // xf != nullptr (16-bit types, plans with EpiLds::X): the kernel still holds the wave's normalised rows in registers, as the MFMA
  // operand fragments of its own positions (lane (row, hi) has the 16-byte chunks 2 * kk + hi of its row); they go through the
  // second scratch area instead of being re-read from global memory.
  static FCSA_DEV void load_inv(float (&r)[NP], const float* inv_norm0, int NG, int lgm, int lane, int rows_valid) {
    const int c = lane % LPR, rr = lane / LPR;
    const int cc = (CH == LPR || c < CH) ? c : 0;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      const int row = pp * RPP + rr;
      const int rowc = row < rows_valid ? row : 0;            // clamped: always a valid address (rows_valid >= 1 here)
      r[pp] = inv_norm0 != nullptr ? inv_norm0[(int64_t)rowc * NG + (cc >> lgm)] : 1.f;
    }
  }
  // (xs, xpitch): this wave's area for the normalised rows -- behind the f32 scratch (scr + 32 * PITCH, pitch XPITCH) or wherever
  // the kernel's LDS plan puts it (EpiLds)
  static FCSA_DEV void put(char* scr, const f32x16 (&acc)[G::DB], float mul, int lane, const u32x4* xf, char* xs, int xpitch) {
    const int x = lane & 31, hi = lane >> 5;
    if constexpr (TR::ES == 2) {
      if (xf != nullptr) {
#pragma unroll
        for (int kk = 0; kk < G::KS; ++kk) *reinterpret_cast<u32x4*>(xs + x * xpitch + (2 * kk + hi) * 16) = xf[kk];
      }
    }
#pragma unroll
    for (int db = 0; db < G::DB; ++db)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq)
        if (32 * db + 8 * rq < D) {
          const f32x4 v = {acc[db][4 * rq] * mul, acc[db][4 * rq + 1] * mul, acc[db][4 * rq + 2] * mul, acc[db][4 * rq + 3] * mul};
          *reinterpret_cast<f32x4*>(scr + x * PITCH + (32 * db + 8 * rq + 4 * hi) * 4) = v;
        }
  }
  // (LDS operations of one wave execute in order: the reads here see the writes of put)
  // XS (compile time): a fused epilogue finds the normalised rows in the scratch (put got `xf`), never in global memory
  template <bool XS>
  static FCSA_DEV void finish(const char* scr, const char* xs, int xpitch, int lane, char* out0, int64_t out_pitch, int rows_valid, bool out_f32,
                              const char* xn0, int64_t xn_pitch, float xn_scale, const float (&rinv)[NP], int lgm, float eps) {
    constexpr bool x_in_scratch = XS && TR::ES == 2;
    const int c = lane % LPR, rr = lane / LPR;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
      const int row = pp * RPP + rr;
      const bool live = (CH == LPR || c < CH) && row < rows_valid;
      const int cc = (CH == LPR || c < CH) ? c : 0;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(scr + row * PITCH + cc * 32);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(scr + row * PITCH + cc * 32 + 16);
      float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      if (xn0 != nullptr) {
        float xh[8];
        const int rowc = row < rows_valid ? row : 0;            // clamped: always a valid address (rows_valid >= 1 here)
        const char* xr = xn0 + (int64_t)rowc * xn_pitch + cc * 8 * TR::ES;
        if constexpr (TR::ES == 4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(xr), b2 = *reinterpret_cast<const f32x4*>(xr + 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) { xh[e] = a[e] * xn_scale; xh[4 + e] = b2[e] * xn_scale; }
        } else {
          const u32x4 a = x_in_scratch ? *reinterpret_cast<const u32x4*>(xs + row * xpitch + cc * 16) : *reinterpret_cast<const u32x4*>(xr);
#pragma unroll
          for (int e = 0; e < 4; ++e) { xh[2 * e] = TR::lo(a[e]) * xn_scale; xh[2 * e + 1] = TR::hi(a[e]) * xn_scale; }
        }
        float dot = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) dot += g[e] * xh[e];
        if (!(CH == LPR || c < CH)) dot = 0.f;                  // padding lanes of a row (D = 96) contribute nothing
        dot = lane_block_sum(dot, lgm);
        const float r = rinv[pp];
        const bool clamped = r >= 1.f / eps;
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = clamped ? g[e] * r : r * (g[e] - xh[e] * dot);
      }
      if (live) {
        if (TR::ES == 4 || out_f32) {
          char* o = out0 + (int64_t)row * out_pitch + c * 32;
          const f32x4 a = {g[0], g[1], g[2], g[3]}, b2 = {g[4], g[5], g[6], g[7]};
          *reinterpret_cast<f32x4*>(o) = a;
          *reinterpret_cast<f32x4*>(o + 16) = b2;
        } else if constexpr (TR::ES == 2) {
          u32x4 u;
#pragma unroll
          for (int e = 0; e < 4; ++e) u[e] = TR::pack2(g[2 * e], g[2 * e + 1]);
          *reinterpret_cast<u32x4*>(out0 + (int64_t)row * out_pitch + c * 16) = u;
        }
      }
    }
  }
  // all three steps in one call
  static FCSA_DEV void store(char* scr, const f32x16 (&acc)[G::DB], float mul, int lane, char* out0, int64_t out_pitch, int rows_valid,
                             bool out_f32, const char* xn0, int64_t xn_pitch, float xn_scale, const float* inv_norm0, int NG, int lgm,
                             float eps, const u32x4* xf = nullptr) {
    float rinv[NP];
    load_inv(rinv, xn0 != nullptr ? inv_norm0 : nullptr, NG, lgm, lane, rows_valid);
    char* xs = scr + 32 * PITCH;
    put(scr, acc, mul, lane, xf, xs, XPITCH);
    if (TR::ES == 2 && xf != nullptr) finish<true>(scr, xs, XPITCH, lane, out0, out_pitch, rows_valid, out_f32, xn0, xn_pitch, xn_scale, rinv, lgm, eps);
    else finish<false>(scr, xs, XPITCH, lane, out0, out_pitch, rows_valid, out_f32, xn0, xn_pitch, xn_scale, rinv, lgm, eps);
  }
};

// LDS plan of a kernel that stages tiles in two buffers (STAGE bytes together) and ends every pass with a RowEpilogue.
// SEP: the f32 epilogue scratch lies BEHIND the staging buffers instead of in them, so the next pass can start loading (LDS-DMA
// into staging buffer 0, row fragments into registers) before the epilogue of the current one has run, and no barrier separates
// the two.  X: there is an area for the normalised rows that come from registers (RowEpilogue::put's `xf`): with SEP it is the
// SECOND staging buffer (free until the next pass's first loop iteration, which every wave reaches after its epilogue), rows
// unpadded; without SEP it follows the f32 scratch.  CAP: LDS bytes one workgroup may take (160 KiB when one workgroup per CU is
// all the registers allow anyway, else half).
template <typename T, int D, int NW, int STAGE, bool AHEAD_OK, int CAP> struct EpiLds {
  typedef RowEpilogue<T, D> EP;
  static constexpr int XROW = D * Traits<T>::ES;
  static constexpr bool SEP = AHEAD_OK && STAGE + NW * EP::BYTES_NOX <= CAP;
  static constexpr bool X = Traits<T>::ES == 2 && (SEP ? NW * 32 * XROW <= STAGE / 2 : NW * EP::BYTES <= CAP);
  static constexpr int PER_WAVE = (X && !SEP) ? EP::BYTES : EP::BYTES_NOX;
  static constexpr int TOTAL = SEP ? STAGE + NW * PER_WAVE : (STAGE > NW * PER_WAVE ? STAGE : NW * PER_WAVE);
  static constexpr int XPITCH = SEP ? XROW : EP::XPITCH;
  static FCSA_DEV char* scratch(char* smem, int wave) { return smem + (SEP ? STAGE : 0) + wave * PER_WAVE; }
  static FCSA_DEV char* xarea(char* smem, int wave) { return SEP ? smem + STAGE / 2 + wave * 32 * XROW : scratch(smem, wave) + 32 * EP::PITCH; }
};

// ---- in-kernel phase timing (trace builds only: make EXTRA=-DFCSA_TRACE OUT=../libfcsa_hip_trace.so) ----
// s_memtime stamps are ISSUED at phase boundaries and only READ after an explicit lgkmcnt(0) at the end of
// the iteration, so they do not add waits inside the pipeline (SMEM returns out of order: a pending stamp only
// makes the compiler's lgkmcnt(n) waits marginally more conservative).  Product builds compile all of it away.
#ifdef FCSA_TRACE
struct Trace {
  static constexpr int N = 12;
  unsigned long long t[N];
  unsigned long long acc[N];
  unsigned long long iters;
  FCSA_DEV void reset() { for (int k = 0; k < N; ++k) { t[k] = 0; acc[k] = 0; } iters = 0; }
  FCSA_DEV void stamp(int k) {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0" : "=s"(t[k]));
    __builtin_amdgcn_sched_barrier(0);
  }
  // call once per iteration after the closing barrier; `last` = index of the last stamp taken
  FCSA_DEV void close(int last) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t[0]), "+s"(t[1]), "+s"(t[2]), "+s"(t[3]), "+s"(t[4]), "+s"(t[5]), "+s"(t[6]), "+s"(t[7]),
                 "+s"(t[8]), "+s"(t[9]), "+s"(t[10]), "+s"(t[11]));
    for (int k = 0; k < last; ++k) acc[k] += t[k + 1] - t[k];
    iters += 1;
  }
  FCSA_DEV void dump(unsigned long long* out, unsigned long long total) const {
    for (int k = 0; k < N; ++k) out[k] = acc[k];
    out[N] = iters;
    out[N + 1] = total;
  }
};
#define FCSA_STAMP(ts, k) (ts).stamp(k)
#else
struct Trace { FCSA_DEV void reset() {} FCSA_DEV void close(int) {} };
#define FCSA_STAMP(ts, k) ((void)0)
#endif

#if defined(FCSA_TRACE) && !defined(FCSA_TRACE_WG)
#define FCSA_TRACE_WG      // the phase-trace build also records every workgroup's start / end time (tools/trace_wg.py)
#endif
#ifdef FCSA_TRACE_WG
FCSA_DEV unsigned long long trace_now() { unsigned long long v; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v)); return v; }
#endif

// Issue slots of the slot-scheduled kernels (fwd2, dkv2): one MFMA + a fixed share of the VALU work + at most a couple
// of memory instructions per slot, fenced so that hipcc's scheduler keeps exactly this program order.
#define FCSA_FENCE() __builtin_amdgcn_sched_barrier(0)
// items [m*N/S, (m+1)*N/S) of N items spread over S slots
#define FCSA_SHARE(m, S, N, i) for (int i = (m) * (N) / (S); i < ((m) + 1) * (N) / (S); ++i)

FCSA_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }      // v_exp_f32 (2^x, quarter rate)

// out[r] = a[r] * b[r] over a 16-register block (plain v_mul_f32: the v_pk_mul_f32 form measured slower)
FCSA_DEV void mul16(f32x16& out, const f32x16& a, const f32x16& b) {
#pragma unroll
  for (int r = 0; r < 16; ++r) out[r] = a[r] * b[r];
}


// Ablation build (-DFCSA_ABL_NOMASK; TIMING ONLY, results are wrong by construction): the tiles that need masking run the unmasked tile
// body, with the tile loops, skips and barriers unchanged -- an upper bound on what ANY cheaper treatment of the causal diagonal / key
// masks could gain (profiles/r06_ablations.txt).  Product builds: the identity.
#ifdef FCSA_ABL_NOMASK
template <int MODE> constexpr int tile_mode() { return 0; }
#else
template <int MODE> constexpr int tile_mode() { return MODE; }
#endif

// bit mask (over accumulator-row positions 0..31) of positions <= thr
FCSA_DEV uint32_t le_mask(int thr) { return thr < 0 ? 0u : (thr >= 31 ? 0xffffffffu : ((2u << thr) - 1u)); }
// bit mask of positions >= thr
FCSA_DEV uint32_t ge_mask(int thr) { return thr <= 0 ? 0xffffffffu : (thr > 31 ? 0u : ~((1u << thr) - 1u)); }

// block id -> (batch*head index, tile index).  Blocks are dispatched round-robin over the 8 XCDs
// (block b -> XCD b % 8, guide §1); keep all tiles of one (batch, head) on one XCD so its K/V (or
// Q/dO) panel stays in that XCD's private 4 MiB L2.  Pure speed choice: any placement is correct.
FCSA_DEV void block_to_work(int id, int n_bh, int n_tiles, int& bh, int& tile) {
  if ((n_bh & 7) == 0) {
    const int xcd = id & 7, slot = id >> 3;
    bh = (slot / n_tiles) * 8 + xcd;
    tile = slot % n_tiles;
  } else {
    bh = id / n_tiles;
    tile = id % n_tiles;
  }
}

// ---------------------------------------------------------------------------------------------
// Query rows of the forward kernels (fcsa_fwd.hip, fcsa_fwd3.hip)
// ---------------------------------------------------------------------------------------------
// Q fragments of query row i (B operand of S^T = K Q^T): lane (i, hi) holds the 16-byte chunks 2*kk + hi, so the lane
// pair of a row holds the whole row.  Three input forms:
//   q_raw    : RAW q (fused l2norm, 16-bit types, group size 8 * 2^lgm).  The grouped l2norm (reference py:50-55,
//              F.normalize eps 1e-12) is done here in registers: per-chunk sums of squares, one lane^32 add per k-step,
//              r = 1 / max(||q_g||, eps), q^ * c1 rounded ONCE to the 16-bit type -- bit-identical to what l2norm_kernel
//              writes -- and published (qn_out, rq_out) for the backward kernels.  Saves the separate HBM pass over q.
//   q_scaled : c1 * q^ already (written by l2norm_kernel: f32, odd group sizes)
//   else     : q^ as given (the reference extension's contract); c1 is folded in here
template <typename T, int D>
FCSA_DEV void request_q_rows(const FwdParams& p, int b, int h, int i, int hi, u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS]) {
  typedef TileGeom<D, Traits<T>::ES> G;
  const char* qrow = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh + (int64_t)i * p.q.sn;
#pragma unroll
  for (int kk = 0; kk < G::KS; ++kk) {
    const u32x4 z = {0u, 0u, 0u, 0u};
    qf[kk] = z;
    if (i < p.N) qf[kk] = *reinterpret_cast<const u32x4*>(qrow + (2 * kk + hi) * 16);
  }
}
// raw row chunks (request_q_rows) -> B operands of the S chains: fused (grouped) l2norm with the c1 * q^ / inverse-norm outputs
// the backward reads, or the plain c1 scaling
template <typename T, int D, bool OPQ = false>
FCSA_DEV void finish_q_frags(const FwdParams& p, int b, int h, int i, const FragAddr<T, D>& fa,
                             u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS], bool publish = true) {
  typedef TileGeom<D, Traits<T>::ES> G;
  if constexpr (Traits<T>::ES == 2) {
    if (p.q_raw) {
      float pair[G::KS];
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) {
        const float ss = dot_frag<T>(qf[kk], qf[kk]);
        pair[kk] = p.lgm >= 1 ? xhalf_sum(ss) : ss;      // groups of >= 16 features contain both chunks of a k-step
      }
      const int sh = p.lgm >= 1 ? p.lgm - 1 : 0;         // k-steps per group = 1 << sh
      const int64_t row = ((int64_t)b * p.H + h) * p.N + i;
      // (the lane half comes from an opaque value: derived from fa.hi, the 2 * KS lane-constant address pairs of the conditional
      //  stores below are hoisted to kernel entry, live across the whole kernel and get spilled in the wider instantiations)
      //  (OPQ: the lean two-waves-per-SIMD forms and the bias + dynamic-shift kernel.  The others keep the hoisted form: with it the C3
      //   kernel measured 1.3 % faster, registers are not its limit)
      const int hi_ = OPQ ? opaque(fa.hi) : fa.hi;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) {
        float tot = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < G::KS; ++k2) tot += ((k2 >> sh) == (kk >> sh)) ? pair[k2] : 0.f;
        const float r = 1.f / fmaxf(sqrtf(tot), p.norm_eps);
        qf[kk] = scale_frag<T>(qf[kk], r * p.c1);
        if (i < p.N && publish) {
          const int c = 2 * kk + hi_;
          if (p.qn_out != nullptr) *reinterpret_cast<u32x4*>(p.qn_out + (row * D + 8 * c) * 2) = qf[kk];      // (inference: nothing is saved)
          if (p.rq_out != nullptr && (c & ((1 << p.lgm) - 1)) == 0) p.rq_out[row * p.G + (c >> p.lgm)] = r;
        }
      }
      return;
    }
  }
  if (!p.q_scaled) {
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) qf[kk] = scale_frag<T>(qf[kk], p.c1);
  }
}

template <typename T, int D>
FCSA_DEV void load_q_frags(const FwdParams& p, int b, int h, int i, const FragAddr<T, D>& fa,
                           u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS]) {
  request_q_rows<T, D>(p, b, h, i, fa.hi, qf);
  finish_q_frags<T, D>(p, b, h, i, fa, qf);
}

}  // namespace fcsa
