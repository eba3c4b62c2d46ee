// fcsa_common.cuh -- device building blocks shared by the gfx950 cosine-sim attention kernels.
//
// Everything here is written for CDNA4 (wave64, v_mfma_f32_32x32x16_{bf16,f16},
// ds_read_b64_tr_b16, 160 KiB LDS with 64 dword banks).  It replaces the reference's
// mem::shared_fragment (cu:89-258), mma::warp_tile (cu:604-1067), rowsum_accumulator
// (cu:262-316) and layout:: tables (cu:320-597) with a different decomposition:
//
//   * One wave owns 32 sequence positions ("row-per-lane"): every tile product is issued so
//     that the MFMA C operand has  column = lane & 31 = the wave's own sequence position  and
//     rows = the other index.  Per-row scalars (row sum, 1/l, delta, norms) are then per-lane
//     scalars and the row reductions are a single cross-half add (lane ^ 32).
//   * C layout of v_mfma_f32_32x32x16 (guide §3): value r of lane l is
//         row  R(r, l>>5) = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),   col = l & 31.
//     A/B operands: lane l holds 8 consecutive k for row/col (l & 31): k = 8 * (l >> 5) + e.
//     A 32x32 f32 result is turned into the 16-bit B operand of the NEXT product without any
//     cross-lane traffic: registers r = 8*ks + e (e = 0..7) of a lane are exactly k-slot
//     (l>>5, e) of k-step ks if the other operand enumerates the contraction index in the order
//         idx(ks, hi, e) = 16*ks + 8*(e >> 2) + 4*hi + (e & 3).
//     That other operand always comes from a row-major LDS tile through two
//     ds_read_b64_tr_b16 (rows idx(ks,hi,0..3) and idx(ks,hi,4..7)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fcsa {

typedef float    f32x16 __attribute__((ext_vector_type(16)));
typedef float    f32x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4  __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2  __attribute__((ext_vector_type(2)));
typedef short    s16x4  __attribute__((ext_vector_type(4)));
typedef __bf16   bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16   bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8  __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2  __attribute__((ext_vector_type(2)));

#define FCSA_DEV __device__ __forceinline__

constexpr float kLog2e = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------
// dtype traits
// ---------------------------------------------------------------------------------------------
struct BF16 {};
struct F16 {};

template <typename T> struct Traits;

template <> struct Traits<BF16> {
  typedef __bf16 elem;
  static FCSA_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
  static FCSA_DEV uint32_t pack2(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(uint32_t, v);
  }
  static FCSA_DEV float lo(uint32_t u) { return __builtin_bit_cast(float, u << 16); }
  static FCSA_DEV float hi(uint32_t u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
};

template <> struct Traits<F16> {
  typedef _Float16 elem;
  static FCSA_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
  static FCSA_DEV uint32_t pack2(float a, float b) {
    f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
  }
  static FCSA_DEV float lo(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[0]; }
  static FCSA_DEV float hi(uint32_t u) { return (float)__builtin_bit_cast(f16x2, u)[1]; }
};

// row index (0..31) of accumulator register r for lane half hi
FCSA_DEV constexpr int crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// ---------------------------------------------------------------------------------------------
// LDS tile geometry: row-major [rows][D] 16-bit tile, 16-byte chunks XOR-swizzled per row so that
//   (a) ds_read_b128 of one chunk column by 16 different rows      (MFMA A/B fragments) and
//   (b) ds_read_b64_tr_b16 of 4 consecutive rows x 64 contiguous B (transposed fragments)
// are both bank-conflict free (64 banks x 4 B; derivation in DESIGN.md §LDS).
// ---------------------------------------------------------------------------------------------
template <int D> struct TileGeom {
  static_assert(D == 16 || D == 32 || D == 64 || D == 96 || D == 128, "dim_head");
  static constexpr int KS = D / 16;                  // 16-wide contraction steps over the feature dim
  static constexpr int DB = (D + 31) / 32;           // 32-wide feature blocks of an output tile
  static constexpr int CPR = D / 8;                  // valid 16-byte chunks per row
  static constexpr int RSC = D <= 16 ? 2 : D <= 32 ? 4 : D <= 64 ? 8 : 16;   // chunks per LDS row (pow2)
  static constexpr int ROWB = RSC * 16;              // LDS row pitch in bytes

  static FCSA_DEV int swz(int row) {
    if constexpr (RSC == 2) return (row >> 3) & 1;
    else if constexpr (RSC == 4) return (row >> 2) & 3;
    else if constexpr (RSC == 8) { int t = (row >> 1) & 7; return ((t & 1) << 2) | (t >> 1); }
    else return ((row & 3) << 2) | ((row >> 2) & 3);
  }
  // byte offset of 16-byte chunk `chunk` of row `row`
  static FCSA_DEV int off(int row, int chunk) { return row * ROWB + ((chunk ^ swz(row)) << 4); }
};

// Per-lane offsets for reading fragments out of a TileGeom<D> tile.
//   row-fragment (ds_read_b128): lane (x = l&31, hi) reads row (rbase + x), features 16*kk + 8*hi .. +8
//   transposed fragment (2 x ds_read_b64_tr_b16): lane (x = l&31, hi) receives, for feature
//   column c = 32*db + x, the 8 rows rbase + idx(ks, hi, e), e = 0..7.
template <int D> struct FragAddr {
  typedef TileGeom<D> G;
  int row_off;        // (l&31) * ROWB
  int row_swz;        // swz(l&31)
  int tr_off[2];      // byte offset (excluding rbase/ks/db terms) for the two 4-row halves
  int tr_swz[2];      // swizzle value of the row this lane ADDRESSES in each half
  int tr_col;         // chunk index contribution of this lane: 2*((l>>4)&1) + ((l&3)>>1)   (+4*db at use)
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
      tr_swz[half] = G::swz(r);                          // independent of 16*ks and 32*jb (see TileGeom)
    }
  }
  // A/B fragment of row (rbase + x): rbase must be a multiple of 32
  FCSA_DEV u32x4 row_frag(const char* tile, int rbase, int kk) const {
    const int chunk = 2 * kk + hi;
    return *reinterpret_cast<const u32x4*>(tile + rbase * G::ROWB + row_off + ((chunk ^ row_swz) << 4));
  }
  // transposed fragment: rows rbase + idx(0, hi, e) (rbase multiple of 16), feature block db
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
};

// ---------------------------------------------------------------------------------------------
// global -> registers -> LDS staging of a [ROWS][D] tile by NT threads (split issue / write, so the
// HBM/L2 latency hides under the MFMA phase in between: guide T14)
// ---------------------------------------------------------------------------------------------
template <int D, int ROWS, int NT> struct Stager {
  typedef TileGeom<D> G;
  static constexpr int NCH = ROWS * G::CPR;
  static constexpr int PER = (NCH + NT - 1) / NT;
  u32x4 r[PER];

  // g: address of (row 0, feature 0) of the tile; pitch in bytes; rows >= rows_valid read as zero
  FCSA_DEV void load(const char* g, int64_t pitch, int rows_valid, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      const int row = c / G::CPR, ch = c % G::CPR;
      u32x4 z = {0u, 0u, 0u, 0u};
      r[i] = z;
      if ((NCH % NT == 0 || c < NCH) && row < rows_valid)
        r[i] = *reinterpret_cast<const u32x4*>(g + (int64_t)row * pitch + ch * 16);
    }
  }
  FCSA_DEV void store(char* tile, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid + i * NT;
      const int row = c / G::CPR, ch = c % G::CPR;
      if (NCH % NT == 0 || c < NCH) *reinterpret_cast<u32x4*>(tile + G::off(row, ch)) = r[i];
    }
  }
};

// pack registers 8*ks .. 8*ks+7 of a 32x32 f32 result into the 16-bit operand of k-step ks
template <typename T> FCSA_DEV u32x4 pack8(const f32x16& p, int ks) {
  u32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = Traits<T>::pack2(p[8 * ks + 2 * e], p[8 * ks + 2 * e + 1]);
  return o;
}

FCSA_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// sum of a per-lane value over the two half-waves (lane ^ 32)
FCSA_DEV float xhalf_sum(float x) { return x + __shfl_xor(x, 32, 64); }

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

}  // namespace fcsa
