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
//   * K and V tiles are staged global -> VGPR -> LDS, double buffered, one barrier per tile;
//   * S^T = K Q^T on v_mfma_f32_32x32x16 (A = K rows via ds_read_b128, B = Q registers), so a
//     lane owns ONE query (column) and 16 keys (rows) of each 32x32 block;
//   * exp2 + masking stay in registers; P~ is packed to 16 bit in place and becomes the
//     B operand of O^T = V^T P~^T, whose A operand V^T comes from the row-major LDS V tile through
//     ds_read_b64_tr_b16 (no P~ round trip through shared memory, cf. cu:1222);
//   * 16-bit types: the row sum comes from the MATRIX pipe -- one extra MFMA per 16-key step with an
//     all-ones A operand accumulates sum_j P~[j][i] of exactly the rounded P~ that feeds P~V, so O is
//     a true convex combination of V rows (the reference also sums the rounded tile, cu:1236) and the
//     VALU, the busier pipe here, loses 32 adds per tile; f32: per-lane adds plus one lane^32 add;
//   * float32 inputs run the same skeleton on v_mfma_f32_32x32x2_f32 (exact f32, 1/16 of the bf16
//     rate): P~ stays f32 and V^T comes from ds_read_b32 (fcsa_common.cuh), no transposed read;
//   * the key loop is split in two SEQUENTIAL loops, first the tiles that need no masking, then the
//     tiles that do (key mask / tail / causal diagonal).  Each loop has one straight-line body, so the
//     accumulators never cross an if/else join (which costs dozens of register copies per tile).
#include <cstdlib>
#include <type_traits>

#include "fcsa_common.cuh"
#include "fcsa_kernels.h"

namespace fcsa {

// exp2 / mask / pack of one 32x32 block of logits (in place): s -> P~ (f32), pb = packed operand, l / lacc updated
template <typename T, bool MASKED, bool BIAS>
FCSA_DEV void fwd_softmax_block(f32x16& s, SecondB<T>& pb, float& l, f32x16& lacc, const FwdParams& p, uint32_t w,
                                int jbase, const char* bias_row) {
  typedef Traits<T> TR;
  float bv[16];
  if constexpr (BIAS) {
    // unconditional loads from clamped (always valid) addresses; out-of-range positions are masked below
    // or never stored, so their value is irrelevant.  bias_row already points at a valid row.
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = min(jbase + crow(r, 0), p.M - 1);
      bv[r] = (float)reinterpret_cast<const typename TR::elem*>(bias_row)[j] * p.bias_c;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float x = s[r];                 // = c1 * qh.kh - c2 already: c1 rides on q, -c2 is the accumulator's initial value
    if constexpr (BIAS) x += bv[r];
    float e = fast_exp2(x);
    if constexpr (MASKED) e = ((w >> crow(r, 0)) & 1u) ? e : 0.f;
    if constexpr (TR::ES == 4) l += e;
    s[r] = e;
  }
  pb.prep(s);
  if constexpr (TR::ES == 2) {      // lacc[*][i] += sum over this block's 32 keys of the rounded P~
    const u32x4 ones = {TR::kOne2, TR::kOne2, TR::kOne2, TR::kOne2};
    lacc = TR::mfma32(ones, pb.v[0], lacc);
    lacc = TR::mfma32(ones, pb.v[1], lacc);
  }
}

// One 64-key tile for one wave.  Software-pipelined INSIDE the wave (in-order issue: the matrix pipe and the VALU
// only overlap if their instructions alternate in program order):
//     K requests | S0 chain | V requests | S1 chain  x exp(block 0) | PV(block 0) x exp(block 1) | PV(block 1)
// The ablation that motivated it: K row fragments requested just in time cost 27% of the kernel, and two
// co-resident workgroups ran only 1.24x faster than one, i.e. other waves do not hide a serial chain.
template <typename T, int D, bool MASKED, bool BIAS>
FCSA_DEV void fwd_tile(const char* kt, const char* vt, const FragAddr<T, D>& fa,
                       const u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS], f32x16 (&o)[TileGeom<D, Traits<T>::ES>::DB],
                       float& l, f32x16& lacc, const FwdParams& p, uint64_t word, uint32_t ncm, int i, int j0, int diff,
                       const char* bias_row) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  // validity bits of this lane's 16 keys per block.  Branch-free and BEFORE the MFMA chains on purpose: a runtime
  // branch between the last MFMA and the first read of its result gets too few wait states on the
  // taken path (hipcc 7.2 pads only the fall-through; seen with the 16-pass v_mfma_f32_32x32x2_f32).
  uint32_t w[2] = {0xffffffffu, 0xffffffffu};
  if constexpr (MASKED) {
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
      w[jb] = ((uint32_t)(word >> (32 * jb)) >> (4 * fa.hi)) & (le_mask(i + diff - (j0 + 32 * jb + 4 * fa.hi)) | ncm);
  }
  u32x4 kf[2][G::KS];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) kf[jb][kk] = fa.row_frag(kt, 32 * jb, kk);
  __builtin_amdgcn_sched_barrier(0);     // keep the K requests up here (the scheduler otherwise sinks them next to each MFMA)

  if constexpr (TR::ES == 2 && !BIAS) {
    constexpr int MFMA = 0x8, VALU = 0x2 | 0x400, DSR = 0x100;
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = -p.c2; s1[r] = -p.c2; }   // exponent shift as the accumulator's initial value
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s0 = TR::mfma32(kf[0][kk], qf[kk], s0);
    u32x4 vf0[G::DB][2], vf1[G::DB][2];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) { vf0[db][0] = fa.tr_frag(vt, 0, db); vf0[db][1] = fa.tr_frag(vt, 16, db); }
    __builtin_amdgcn_sched_barrier(0);
    // --- S1 chain interleaved with exp / pack of block 0
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s1 = TR::mfma32(kf[1][kk], qf[kk], s1);
    SecondB<T> pb0, pb1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float e = fast_exp2(s0[r]);
      if constexpr (MASKED) e = ((w[0] >> crow(r, 0)) & 1u) ? e : 0.f;
      s0[r] = e;
    }
    pb0.prep(s0);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) { vf1[db][0] = fa.tr_frag(vt, 32, db); vf1[db][1] = fa.tr_frag(vt, 48, db); }
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(VALU, (MASKED ? 48 : 26) / G::KS + 1, 0);
      __builtin_amdgcn_sched_group_barrier(DSR, (4 * G::DB + G::KS - 1) / G::KS, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // --- PV of block 0 (row sum + DB output blocks) interleaved with exp / pack of block 1
    const u32x4 ones = {TR::kOne2, TR::kOne2, TR::kOne2, TR::kOne2};
    lacc = TR::mfma32(ones, pb0.v[0], lacc);
    lacc = TR::mfma32(ones, pb0.v[1], lacc);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      o[db] = TR::mfma32(vf0[db][0], pb0.v[0], o[db]);
      o[db] = TR::mfma32(vf0[db][1], pb0.v[1], o[db]);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float e = fast_exp2(s1[r]);
      if constexpr (MASKED) e = ((w[1] >> crow(r, 0)) & 1u) ? e : 0.f;
      s1[r] = e;
    }
    pb1.prep(s1);
    constexpr int NPV = 2 + 2 * G::DB;
#pragma unroll
    for (int m = 0; m < NPV; ++m) {
      __builtin_amdgcn_sched_group_barrier(MFMA, 1, 0);
      __builtin_amdgcn_sched_group_barrier(VALU, (MASKED ? 48 : 26) / NPV + 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // --- PV of block 1
    lacc = TR::mfma32(ones, pb1.v[0], lacc);
    lacc = TR::mfma32(ones, pb1.v[1], lacc);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      o[db] = TR::mfma32(vf1[db][0], pb1.v[0], o[db]);
      o[db] = TR::mfma32(vf1[db][1], pb1.v[1], o[db]);
    }
  } else {
    // generic order (f32, bias): both S chains first, then per block: softmax, PV
    f32x16 s[2];
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[jb][r] = -p.c2;
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
        fwd_softmax_block<T, MASKED, BIAS>(s[jb], pb, l, lacc, p, w[jb], j0 + 32 * jb + 4 * fa.hi, bias_row);
#pragma unroll
        for (int db = 0; db < G::DB; ++db) {
          o[db] = TR::mfma32(vf[db][0], pb.v[0], o[db]);
          o[db] = TR::mfma32(vf[db][1], pb.v[1], o[db]);
        }
      } else {
        fwd_softmax_block<T, MASKED, BIAS>(s[jb], pb, l, lacc, p, w[jb], j0 + 32 * jb + 4 * fa.hi, bias_row);
#pragma unroll
        for (int db = 0; db < G::DB; ++db) o[db] = second_mma<T, D>(o[db], vt, 32 * jb, db, pb, fa);
      }
    }
  }
}

template <typename T, int D, int NW, bool BIAS>
__global__ void __launch_bounds__(NW * 64, (D * Traits<T>::ES <= 128 ? 2 : 1)) fwd_kernel(const FwdParams p) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr int BN = 64, BM = 32 * NW, NT = NW * 64;
  constexpr int TILE_B = BN * G::ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K tile | V tile]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FragAddr<T, D> fa;
  fa.init(lane);

  // Work of a row tile grows with its index under causal masking (2 .. 2*MT key tiles), and a workgroup runs
  // start to finish on one CU, so the makespan would be set by the heaviest tile.  Each workgroup therefore
  // takes a PAIR of row tiles (MT-1-pt, pt): constant work per workgroup.  Non-causal: one tile each.
  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (p.causal && (MT - 1 - pt) != pt) ? 2 : 1;
  const int diff = p.M - p.N;                     // cu:1097 seq_len_diff
  const uint32_t ncm = p.causal ? 0u : 0xffffffffu;   // OR-ed into the causal bit mask: all ones when not causal
  for (int pass = 0; pass < npass; ++pass) {
  const int mt = p.causal ? (pass == 0 ? MT - 1 - pt : pt) : pt;      // heavy tile first
  const int m0 = mt * BM;
  const int mw = m0 + wave * 32;                  // first query row of this wave
  const int i = mw + (lane & 31);                 // this lane's query row

  // key tiles this workgroup needs
  int last_key = p.M - 1;
  if (p.causal) last_key = min(last_key, m0 + BM - 1 + diff);
  const int nt = last_key < 0 ? 0 : last_key / BN + 1;

  // Q fragments (B operand of S^T = K Q^T): 16-byte chunk 2*kk + hi of row i
  u32x4 qf[G::KS];
  {
    const char* qrow = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh + (int64_t)i * p.q.sn;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      u32x4 z = {0u, 0u, 0u, 0u};
      qf[kk] = z;
      if (i < p.N) qf[kk] = *reinterpret_cast<const u32x4*>(qrow + (2 * kk + fa.hi) * 16);
      if (!p.q_scaled) qf[kk] = scale_frag<T>(qf[kk], p.c1);     // reference-contract path: q^ given, fold c1 here
    }
  }

  f32x16 o[G::DB];
#pragma unroll
  for (int db = 0; db < G::DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float l = 0.f;          // f32: per-lane partial row sum
  f32x16 lacc;            // 16 bit: row sums from the ones-MFMA (every register holds the full sum of column i)
#pragma unroll
  for (int r = 0; r < 16; ++r) lacc[r] = 0.f;

  const char* kbase = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh;
  const char* vbase = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh;
  const uint8_t* mrow = p.mask ? p.mask + (int64_t)b * p.M : nullptr;
  const char* bias_row = nullptr;                 // row min(i, N-1): always a valid address
  if constexpr (BIAS)
    bias_row = p.bias + ((int64_t)(p.bias_batch ? b : h) * p.N + min(i, p.N - 1)) * (int64_t)p.M * (int64_t)sizeof(typename TR::elem);

  Stager<T, D, BN, NT> sk, sv;
  sk.init(p.k.sn, tid);
  sv.init(p.v.sn, tid);
  uint8_t mb = 1;
  if (nt > 0) {
    sk.load(kbase, p.k.sn, p.M);
    sv.load(vbase, p.v.sn, p.M);
    if (mrow) mb = lane < p.M ? mrow[lane] : (uint8_t)0;
    sk.store(smem, tid);
    sv.store(smem + TILE_B, tid);
  }
  __syncthreads();
  // Every prologue load (Q / dO / K / V fragments, first tile) is complete on the real path; say so on ALL paths.
  // Otherwise hipcc's waitcnt model keeps them pending along the no-tile path, the loop-header merge never
  // clears that, and each iteration re-waits with vmcnt(0) at its first MFMA -- right after issuing the next
  // tile's prefetch, which serialises the prefetch with the compute meant to hide it.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched

  // tiles [0, t_split) need no masking for THIS wave, tiles [t_split, nt) do (wave-uniform split; both
  // loops execute one barrier per tile, so waves of one workgroup may sit in different loops)
  int t_split = 0;
  if (!BIAS && mrow == nullptr) {
    t_split = p.M / BN;                                                // tail tile (j0 + BN > M) is masked
    if (p.causal) t_split = min(t_split, max(0, mw + diff + 1) / BN);  // needs (t+1)*BN - 1 <= mw + diff
    t_split = min(t_split, nt);
  }

  auto run = [&](auto masked_tag, int t_begin, int t_end) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    for (int t = t_begin; t < t_end; ++t) {
      const int j0 = t * BN;
      const char* kcur = smem + (t & 1) * 2 * TILE_B;
      const char* vcur = kcur + TILE_B;
      char* knxt = smem + ((t + 1) & 1) * 2 * TILE_B;
      const bool more = t + 1 < nt;
      uint64_t word = 0;
      if constexpr (MASKED) {
        // consume the mask byte loaded one tile ago BEFORE issuing new loads (its wait then covers nothing else)
        word = __ballot((j0 + lane) < p.M && mb != 0);                  // valid keys of this tile
        if (mrow && more) {
          const int key = j0 + BN + lane;
          mb = key < p.M ? mrow[key] : (uint8_t)0;
        }
      }
      if (more) {   // issue next tile's global loads now; they land while this tile is computed
        sk.load(kbase + (int64_t)(j0 + BN) * p.k.sn, p.k.sn, p.M - (j0 + BN));
        sv.load(vbase + (int64_t)(j0 + BN) * p.v.sn, p.v.sn, p.M - (j0 + BN));
      }
      if constexpr (MASKED) {
        const bool skip = p.causal && (j0 > mw + 31 + diff);            // no valid pair for this wave
        if (!skip) fwd_tile<T, D, true, BIAS>(kcur, vcur, fa, qf, o, l, lacc, p, word, ncm, i, j0, diff, bias_row);
      } else {
        fwd_tile<T, D, false, BIAS>(kcur, vcur, fa, qf, o, l, lacc, p, 0, ncm, i, j0, diff, bias_row);
      }
      if (more) {
        sk.store(knxt, tid);
        sv.store(knxt + TILE_B, tid);
      }
      __syncthreads();
    }
  };
  run(std::false_type{}, 0, t_split);
  run(std::true_type{}, t_split, nt);

  // epilogue: normalise and store.  Lane (i, hi) holds O[i][32*db + 8*rq + 4*hi + 0..3].
  const float lt = (TR::ES == 2) ? lacc[0] : xhalf_sum(l);
  const float inv = 1.f / fmaxf(lt, p.l_eps);     // cu:1239 (constants::eps, cu:83), rescaled with the shift
  if (i < p.N) {
    if (p.inv_l != nullptr && fa.hi == 0) p.inv_l[((int64_t)b * p.H + h) * p.N + i] = inv;
    char* orow = p.o.p + (int64_t)b * p.o.sb + (int64_t)h * p.o.sh + (int64_t)i * p.o.sn;
    store_row_tile<T, D>(orow, o, inv, fa.hi, false);
  }
  }   // pass
}

template <typename T, int D, bool BIAS>
static hipError_t launch_fwd_b(const FwdParams& p, hipStream_t s) {
  constexpr int NW = 4;
  constexpr int BM = 32 * NW;
  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  const size_t lds = 4 * 64 * TileGeom<D, Traits<T>::ES>::ROWB;
  auto kern = fwd_kernel<T, D, NW, BIAS>;
  static bool attr_set = false;                  // per instantiation; the attribute is sticky
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.H * PT)), dim3(NW * 64), lds, s, p);
  return hipGetLastError();
}

template <typename T, int D>
static hipError_t launch_fwd_t(const FwdParams& p, hipStream_t s) {
  return p.bias != nullptr ? launch_fwd_b<T, D, true>(p, s) : launch_fwd_b<T, D, false>(p, s);
}

template <typename T>
static hipError_t launch_fwd_d(int D, const FwdParams& p, hipStream_t s) {
  switch (D) {
    case 16:  return launch_fwd_t<T, 16>(p, s);
    case 32:  return launch_fwd_t<T, 32>(p, s);
    case 64:  return launch_fwd_t<T, 64>(p, s);
    case 96:  return launch_fwd_t<T, 96>(p, s);
    case 128: return launch_fwd_t<T, 128>(p, s);
    default:  return hipErrorInvalidValue;
  }
}

hipError_t launch_forward(int dtype, int D, const FwdParams& p, hipStream_t s) {
  if (p.B * p.H == 0 || p.N == 0) return hipSuccess;
  if (dtype == 2) return launch_fwd_d<BF16>(D, p, s);
  if (dtype == 1) return launch_fwd_d<F16>(D, p, s);
  if (dtype == 0) return launch_fwd_d<F32>(D, p, s);
  return hipErrorInvalidValue;
}

}  // namespace fcsa
