// fcsa_bwd.hip -- backward kernels of fused cosine-similarity attention for gfx950 (f16 / bf16 / f32).
//
// Replaces backward_preprocess (reference cu:1256-1335) and backward_kernel (cu:1339-1626).
// Math, with Qh/Kh the normalised inputs:
//     delta = rowsum(dO * O);  P = exp(S - shift) * inv_l;  dV = P^T dO;  dP = dO V^T;
//     dS = P * (dP - delta)  (= d_bias);  dQh = scale * dS Kh;  dKh = scale * dS^T Qh
//
// The reference runs ONE key-tile-parallel kernel and pushes dQ through f32 global atomics
// (cu:1610) after a one-row-per-block delta kernel (cu:1842-1850).  Here the work is split so
// that nothing needs atomics and every accumulator lives in registers of the wave that owns it:
//
//   bwd_dq_kernel  : query-row parallel (same skeleton as the forward kernel).  Prologue computes
//                    delta for its own rows in registers (no separate kernel) and publishes it;
//                    loop: S^T = K Q^T, dP^T = V dO^T, dS^T, dQ^T += K^T dS^T.
//   bwd_dkv_kernel : key-row parallel.  K/V fragments of the wave's 32 keys stay in VGPRs; Q and dO
//                    tiles stream through LDS; S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS.
//                    Writes one [M, D] slab per (batch, q-head); with single-headed K/V the slabs are
//                    reduced over heads by the finalize kernel (fcsa_norm.hip) instead of the
//                    reference's f32 atomics (cu:1613-1619).
//
// 7 tile products instead of the reference's 5 (S and dP are recomputed in both kernels); results
// are deterministic.  d_bias (optional path) is accumulated with f32 atomics like cu:1574-1576.
// As in the forward kernel, each kernel runs its unmasked tiles and its masked tiles in two
// sequential loops with one straight-line body each (no accumulator copies at if/else joins).
#include <type_traits>

// tuning knobs (row bytes D*ES up to which a kernel asks for 2 waves/SIMD, i.e. <= 256 registers)
#ifndef FCSA_DKV_2W_BYTES
#define FCSA_DKV_2W_BYTES 128
#endif
#ifndef FCSA_DQ_2W_BYTES
#define FCSA_DQ_2W_BYTES 128
#endif

#include "fcsa_common.cuh"
#include "fcsa_kernels.h"

namespace fcsa {
#ifdef FCSA_TRACE
__device__ unsigned long long g_trace_dkv[128];
#endif

// =============================================================================================
// dQ kernel
// =============================================================================================
template <typename T, int D, bool MASKED, bool BIAS>
FCSA_DEV void dq_tile(const char* kt, const char* vt, const FragAddr<T, D>& fa,
                      const u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS], const u32x4 (&dof)[TileGeom<D, Traits<T>::ES>::KS],
                      f32x16 (&dq)[TileGeom<D, Traits<T>::ES>::DB], float lc, float delta, const BwdParams& p, uint64_t word,
                      uint32_t ncm, int i, int j0, int diff, const char* bias_row, float* dbias_row) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    // branch-free and before the MFMA chains on purpose (see fwd_tile)
    uint32_t w = 0xffffffffu;
    if constexpr (MASKED)
      w = ((uint32_t)(word >> (32 * jb)) >> (4 * fa.hi)) & (le_mask(i + diff - (j0 + 32 * jb + 4 * fa.hi)) | ncm);
    // request every row fragment of this 32-key block first (just-in-time ds_read_b128 in front of their
    // dependent MFMA were the most expensive item of the forward tile, see fwd_tile)
    // (requests are batched PF k-steps at a time so the live fragments stay within the register budget)
    constexpr int PF = G::KS <= 4 ? G::KS : 2;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = lc; dp[r] = -delta; }  // lc = log2(inv_l) - c2 and -delta ride in as initial values
#pragma unroll
    for (int k0 = 0; k0 < G::KS; k0 += PF) {
      u32x4 kfr[PF], vfr[PF];
#pragma unroll
      for (int kk = 0; kk < PF; ++kk) { kfr[kk] = fa.row_frag(kt, 32 * jb, k0 + kk); vfr[kk] = fa.row_frag(vt, 32 * jb, k0 + kk); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < PF; ++kk) s = TR::mfma32(kfr[kk], qf[k0 + kk], s);
#pragma unroll
      for (int kk = 0; kk < PF; ++kk) dp = TR::mfma32(vfr[kk], dof[k0 + kk], dp);
    }

    const int jbase = j0 + 32 * jb + 4 * fa.hi;
    float bv[16];
    if constexpr (BIAS) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = min(jbase + crow(r, 0), p.M - 1);
        bv[r] = (float)reinterpret_cast<const typename TR::elem*>(bias_row)[j] * p.bias_c;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = s[r];               // = c1 * qh.kh + lc already (c1 rides on q, lc is the accumulator's initial value)
      if constexpr (BIAS) x += bv[r];
      float e = fast_exp2(x);
      if constexpr (MASKED) e = ((w >> crow(r, 0)) & 1u) ? e : 0.f;
      const float ds = e * dp[r];        // dp already holds dP - delta
      if constexpr (BIAS) {
        const int j = jbase + crow(r, 0);
        if (dbias_row != nullptr && j < p.M && ds != 0.f) atomicAdd(dbias_row + j, ds);   // cu:1574-1576
      }
      s[r] = ds;
    }
    SecondB<T> pb;
    pb.prep(s);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) dq[db] = second_mma<T, D>(dq[db], kt, 32 * jb, db, pb, fa);
  }
}

template <typename T, int D, int NW, bool BIAS>
__global__ void __launch_bounds__(NW * 64, (D * Traits<T>::ES <= FCSA_DQ_2W_BYTES ? 2 : 1)) bwd_dq_kernel(const BwdParams p) {
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

  // causal: a workgroup takes the PAIR of row tiles (MT-1-pt, pt) -> constant work per workgroup (see fwd_kernel)
  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (p.causal && (MT - 1 - pt) != pt) ? 2 : 1;
  const int diff = p.M - p.N;
  const uint32_t ncm = p.causal ? 0u : 0xffffffffu;   // OR-ed into the causal bit mask: all ones when not causal
  for (int pass = 0; pass < npass; ++pass) {
  const int mt = p.causal ? (pass == 0 ? MT - 1 - pt : pt) : pt;      // heavy tile first
  const int m0 = mt * BM;
  const int mw = m0 + wave * 32;
  const int i = mw + (lane & 31);

  int last_key = p.M - 1;
  if (p.causal) last_key = min(last_key, m0 + BM - 1 + diff);
  const int nt = last_key < 0 ? 0 : last_key / BN + 1;

  // Q, dO fragments (B operands) and delta = <dO_i, O_i>  (replaces backward_preprocess, cu:1256-1335)
  u32x4 qf[G::KS], dof[G::KS];
  float delta = 0.f, lc = 0.f;
  {
    const char* qrow = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh + (int64_t)i * p.q.sn;
    const char* dorow = p.d_out.p + (int64_t)b * p.d_out.sb + (int64_t)h * p.d_out.sh + (int64_t)i * p.d_out.sn;
    const char* orow = p.o.p + (int64_t)b * p.o.sb + (int64_t)h * p.o.sh + (int64_t)i * p.o.sn;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      u32x4 z = {0u, 0u, 0u, 0u};
      qf[kk] = z;
      dof[kk] = z;
      if (i < p.N) {
        qf[kk] = *reinterpret_cast<const u32x4*>(qrow + (2 * kk + fa.hi) * 16);
        if (!p.q_scaled) qf[kk] = scale_frag<T>(qf[kk], p.c1);
        dof[kk] = *reinterpret_cast<const u32x4*>(dorow + (2 * kk + fa.hi) * 16);
        const u32x4 of = *reinterpret_cast<const u32x4*>(orow + (2 * kk + fa.hi) * 16);
        delta += dot_frag<T>(dof[kk], of);
      }
    }
    delta = xhalf_sum(delta);
    if (i < p.N) {
      const int64_t ridx = ((int64_t)b * p.H + h) * p.N + i;
      lc = __builtin_amdgcn_logf(p.inv_l[ridx]) - p.c2;     // v_log_f32 = log2
      if (fa.hi == 0) p.delta[ridx] = delta;
    }
  }

  f32x16 dq[G::DB];
#pragma unroll
  for (int db = 0; db < G::DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  const char* kbase = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh;
  const char* vbase = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh;
  const uint8_t* mrow = p.mask ? p.mask + (int64_t)b * p.M : nullptr;
  const char* bias_row = nullptr;                 // row min(i, N-1): always a valid address
  float* dbias_row = nullptr;                     // only for real rows
  if constexpr (BIAS) {
    const int64_t boff = ((int64_t)(p.bias_batch ? b : h) * p.N + min(i, p.N - 1)) * (int64_t)p.M;
    bias_row = p.bias + boff * (int64_t)sizeof(typename TR::elem);
    if (p.d_bias != nullptr && i < p.N) dbias_row = p.d_bias + boff;
  }

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

  int t_split = 0;                                 // see fwd_kernel
  if (!BIAS && mrow == nullptr) {
    t_split = p.M / BN;
    if (p.causal) t_split = min(t_split, max(0, mw + diff + 1) / BN);
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
      if constexpr (MASKED) {     // consume the mask byte BEFORE issuing new loads (see fwd_kernel)
        word = __ballot((j0 + lane) < p.M && mb != 0);
        if (mrow && more) {
          const int key = j0 + BN + lane;
          mb = key < p.M ? mrow[key] : (uint8_t)0;
        }
      }
      if (more) {
        sk.load(kbase + (int64_t)(j0 + BN) * p.k.sn, p.k.sn, p.M - (j0 + BN));
        sv.load(vbase + (int64_t)(j0 + BN) * p.v.sn, p.v.sn, p.M - (j0 + BN));
      }
      if constexpr (MASKED) {
        const bool skip = p.causal && (j0 > mw + 31 + diff);
        if (!skip) dq_tile<T, D, true, BIAS>(kcur, vcur, fa, qf, dof, dq, lc, delta, p, word, ncm, i, j0, diff, bias_row, dbias_row);
      } else {
        dq_tile<T, D, false, BIAS>(kcur, vcur, fa, qf, dof, dq, lc, delta, p, 0, ncm, i, j0, diff, bias_row, dbias_row);
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

  if (i < p.N) {
    char* row = p.dq.p + (int64_t)b * p.dq.sb + (int64_t)h * p.dq.sh + (int64_t)i * p.dq.sn;
    if (p.rq != nullptr) {      // dq = l2norm_backward(scale * dS K^): p.q holds c1 * q^ (or q^), contiguous rows
      const char* xrow = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh + (int64_t)i * p.q.sn;
      store_row_tile_l2norm_bwd<T, D>(row, dq, p.scale, fa.hi, xrow, p.q_scaled ? 1.f / p.c1 : 1.f,
                                      p.rq + (((int64_t)b * p.H + h) * p.N + i) * p.G, p.lgm, p.norm_eps);
    } else {
      store_row_tile<T, D>(row, dq, p.scale, fa.hi, p.dq_f32 != 0);     // cu:1580-1582: dS *= scale
    }
  }
  }   // pass
}

// =============================================================================================
// dK / dV kernel
// =============================================================================================
template <typename T, int D, int BMQ, bool MASKED, bool BIAS>
FCSA_DEV void dkv_tile(const char* qt, const char* dot, const float* lcs, const float* dls, const FragAddr<T, D>& fa,
                       const u32x4 (&kf)[TileGeom<D, Traits<T>::ES>::KS], const u32x4 (&vf)[TileGeom<D, Traits<T>::ES>::KS],
                       f32x16 (&dk)[TileGeom<D, Traits<T>::ES>::DB], f32x16 (&dv)[TileGeom<D, Traits<T>::ES>::DB],
                       const BwdParams& p, uint32_t kmask, uint32_t ncm, int j, int i0, int diff, const char* bias_col, Trace& ts) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
#pragma unroll
  for (int ib = 0; ib < BMQ / 32; ++ib) {
    // query i of register r: i0 + 32*ib + crow(r, hi); valid iff i + diff >= j.  Rows >= N carry
    // lc = -inf (P = 0) and zero Q / dO, so whole blocks beyond N contribute exactly nothing.
    // Branch-free and before the MFMA chains on purpose (see fwd_tile).
    uint32_t w = 0xffffffffu;
    if constexpr (MASKED) w = kmask & (ge_mask(j - diff - (i0 + 32 * ib + 4 * fa.hi)) | ncm);
    f32x16 s, dp;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {      // per-query log-normaliser and -delta (negated at staging) as the accumulators' initial values
      const f32x4 lc4 = *reinterpret_cast<const f32x4*>(lcs + 32 * ib + 8 * rq + 4 * fa.hi);
      const f32x4 nd4 = *reinterpret_cast<const f32x4*>(dls + 32 * ib + 8 * rq + 4 * fa.hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[4 * rq + e] = lc4[e]; dp[4 * rq + e] = nd4[e]; }
    }
    // (row fragments are requested next to their MFMA here: this kernel sits at its register budget, and
    //  batching the requests as in dq_tile spills)
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s = TR::mfma32(fa.row_frag(qt, 32 * ib, kk), kf[kk], s);
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) dp = TR::mfma32(fa.row_frag(dot, 32 * ib, kk), vf[kk], dp);
    FCSA_STAMP(ts, 2 + 3 * ib);

    f32x16 pr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = s[r];                 // = c1 * qh.kh + lc already
      if constexpr (BIAS) {   // clamped row: always a valid address; rows >= N have P = 0 through lc = -inf
        const int i = min(i0 + 32 * ib + crow(r, 0) + 4 * fa.hi, p.N - 1);
        const typename TR::elem bv = *reinterpret_cast<const typename TR::elem*>(
            bias_col + (int64_t)i * p.M * (int64_t)sizeof(typename TR::elem));
        x += (float)bv * p.bias_c;
      }
      float pe = fast_exp2(x);
      if constexpr (MASKED) pe = ((w >> crow(r, 0)) & 1u) ? pe : 0.f;
      pr[r] = pe;
      s[r] = pe * dp[r];             // dp already holds dP - delta
    }
    SecondB<T> pp, pd;
    pp.prep(pr);
    pd.prep(s);
    FCSA_STAMP(ts, 3 + 3 * ib);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      dv[db] = second_mma<T, D>(dv[db], dot, 32 * ib, db, pp, fa);
      dk[db] = second_mma<T, D>(dk[db], qt, 32 * ib, db, pd, fa);
    }
    FCSA_STAMP(ts, 4 + 3 * ib);
  }
}

template <typename T, int D, int NW, int BMQ, bool BIAS>
__global__ void __launch_bounds__(NW * 64, (D * Traits<T>::ES <= FCSA_DKV_2W_BYTES ? 2 : 1)) bwd_dkv_kernel(const BwdParams p) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr int BNK = 32 * NW, NT = NW * 64;
  constexpr int TILE_B = BMQ * G::ROWB;
  constexpr int BUF_B = 2 * TILE_B + 2 * BMQ * 4;                // Q tile | dO tile | lc[BMQ] | delta[BMQ]
  static_assert(BMQ % 32 == 0 && BMQ <= NT, "query tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][BUF_B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FragAddr<T, D> fa;
  fa.init(lane);

  // causal: the LOW key tiles are the heavy ones (they see every later query); pair (pt, KT-1-pt) per workgroup
  const int KT = (p.M + BNK - 1) / BNK;
  const int PT = p.causal ? (KT + 1) / 2 : KT;
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (p.causal && (KT - 1 - pt) != pt) ? 2 : 1;
  const int diff = p.M - p.N;
  Trace ts;
  ts.reset();
#ifdef FCSA_TRACE
  const unsigned long long trace_t0 = trace_now();
#endif
  for (int pass = 0; pass < npass; ++pass) {
  const int kt = p.causal ? (pass == 0 ? pt : KT - 1 - pt) : pt;      // heavy tile first
  const int n0 = kt * BNK;
  const int nw = n0 + wave * 32;                        // first key of this wave
  const int j = nw + (lane & 31);                       // this lane's key

  // query tiles this workgroup needs: causal keeps i >= j - diff
  const int QT = (p.N + BMQ - 1) / BMQ;
  int t0 = 0;
  if (p.causal) t0 = max(0, n0 - diff) / BMQ;

  // K, V fragments of this lane's key (B operands of S = Q K^T and dP = dO V^T), kept for the whole loop
  u32x4 kf[G::KS], vf[G::KS];
  {
    const char* krow = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)j * p.k.sn;
    const char* vrow = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh + (int64_t)j * p.v.sn;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      u32x4 z = {0u, 0u, 0u, 0u};
      kf[kk] = z;
      vf[kk] = z;
      if (j < p.M) {
        kf[kk] = *reinterpret_cast<const u32x4*>(krow + (2 * kk + fa.hi) * 16);
        if (!p.q_scaled) kf[kk] = scale_frag<T>(kf[kk], p.c1);     // S = Q (c1 K)^T when the Q tile is plain q^
        vf[kk] = *reinterpret_cast<const u32x4*>(vrow + (2 * kk + fa.hi) * 16);
      }
    }
  }
  bool key_ok = j < p.M;
  if (p.mask != nullptr && key_ok) key_ok = p.mask[(int64_t)b * p.M + j] != 0;
  const uint32_t kmask = key_ok ? 0xffffffffu : 0u;      // this lane's key: valid for every query or for none
  const uint32_t ncm = p.causal ? 0u : 0xffffffffu;

  f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
  for (int db = 0; db < G::DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  const char* qbase = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh;
  const char* dobase = p.d_out.p + (int64_t)b * p.d_out.sb + (int64_t)h * p.d_out.sh;
  const float* invl_row = p.inv_l + ((int64_t)b * p.H + h) * p.N;
  const float* delta_row = p.delta + ((int64_t)b * p.H + h) * p.N;
  const char* bias_col = nullptr;                 // column min(j, M-1): always a valid address
  if constexpr (BIAS)
    bias_col = p.bias + ((int64_t)(p.bias_batch ? b : h) * p.N * (int64_t)p.M + min(j, p.M - 1)) * (int64_t)sizeof(typename TR::elem);

  Stager<T, D, BMQ, NT> sq, sdo;
  sq.init(p.q.sn, tid);
  sdo.init(p.d_out.sn, tid);
  float lc_r = 0.f, dl_r = 0.f;
  bool row_ok = false;
  auto load_tile = [&](int t) {
    const int i0 = t * BMQ;
    sq.load(qbase + (int64_t)i0 * p.q.sn, p.q.sn, p.N - i0);
    sdo.load(dobase + (int64_t)i0 * p.d_out.sn, p.d_out.sn, p.N - i0);
    if (tid < BMQ) {       // raw loads only: any arithmetic on them here would force an immediate vmcnt wait
      const int i = min(i0 + tid, p.N - 1);
      lc_r = invl_row[i];
      dl_r = delta_row[i];
      row_ok = i0 + tid < p.N;
    }
  };
  auto store_tile = [&](char* buf) {
    sq.store(buf, tid);
    sdo.store(buf + TILE_B, tid);
    if (tid < BMQ) {
      // rows beyond N: lc = -inf makes P exactly 0 there
      reinterpret_cast<float*>(buf + 2 * TILE_B)[tid] = row_ok ? __builtin_amdgcn_logf(lc_r) - p.c2 : -INFINITY;
      reinterpret_cast<float*>(buf + 2 * TILE_B + BMQ * 4)[tid] = row_ok ? -dl_r : 0.f;      // -delta
    }
  };

  if (t0 < QT) {
    load_tile(t0);
    store_tile(smem);
  }
  __syncthreads();
  // Every prologue load (Q / dO / K / V fragments, first tile) is complete on the real path; say so on ALL paths.
  // Otherwise hipcc's waitcnt model keeps them pending along the no-tile path, the loop-header merge never
  // clears that, and each iteration re-waits with vmcnt(0) at its first MFMA -- right after issuing the next
  // tile's prefetch, which serialises the prefetch with the compute meant to hide it.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched

  // query tiles [t0, t_m) need masking for THIS wave (key mask / invalid keys: all of them; causal: the tiles
  // that touch the diagonal, i0 + diff < nw + 31), tiles [t_m, QT) do not.  Wave-uniform split.
  int t_m = QT;
  if (!BIAS && p.mask == nullptr && n0 + BNK <= p.M) {
    t_m = t0;
    if (p.causal) t_m = min(QT, max(t0, (nw + 31 - diff + BMQ - 1) / BMQ));
  }

  auto run = [&](auto masked_tag, int t_begin, int t_end) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    for (int t = t_begin; t < t_end; ++t) {
      const int i0 = t * BMQ;
      const int par = (t - t0) & 1;
      const char* cur = smem + par * BUF_B;
      char* nxt = smem + (par ^ 1) * BUF_B;
      const bool more = t + 1 < QT;
      FCSA_STAMP(ts, 0);
      if (more) load_tile(t + 1);
      FCSA_STAMP(ts, 1);
      const float* lcs = reinterpret_cast<const float*>(cur + 2 * TILE_B);
      const float* dls = lcs + BMQ;
      if constexpr (MASKED) {
        const bool skip = p.causal && (i0 + BMQ - 1 + diff < nw);           // no valid pair for this wave
        if (!skip) dkv_tile<T, D, BMQ, true, BIAS>(cur, cur + TILE_B, lcs, dls, fa, kf, vf, dk, dv, p, kmask, ncm, j, i0, diff, bias_col, ts);
      } else {
        dkv_tile<T, D, BMQ, false, BIAS>(cur, cur + TILE_B, lcs, dls, fa, kf, vf, dk, dv, p, kmask, ncm, j, i0, diff, bias_col, ts);
      }
      FCSA_STAMP(ts, 8);
      if (more) store_tile(nxt);
      FCSA_STAMP(ts, 9);
      __syncthreads();
      FCSA_STAMP(ts, 10);
      if constexpr (!MASKED) ts.close(10);
    }
  };
  run(std::true_type{}, t0, t_m);
  run(std::false_type{}, t_m, QT);

  if (j < p.M) {
    char* dkrow = p.dk.p + (int64_t)b * p.dk.sb + (int64_t)h * p.dk.sh + (int64_t)j * p.dk.sn;
    char* dvrow = p.dv.p + (int64_t)b * p.dv.sb + (int64_t)h * p.dv.sh + (int64_t)j * p.dv.sn;
    // dKh = scale * dS^T Qh; when the Q tile holds c1 * qh the factor becomes scale / c1 (= ln 2)
    const float kmul = p.q_scaled ? p.scale / p.c1 : p.scale;
    if (p.rk != nullptr) {      // dk = l2norm_backward(dKh): p.k holds k^ (K/V with heads only)
      const char* xrow = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)j * p.k.sn;
      store_row_tile_l2norm_bwd<T, D>(dkrow, dk, kmul, fa.hi, xrow, 1.f,
                                      p.rk + (((int64_t)b * p.H + h) * p.M + j) * p.G, p.lgm, p.norm_eps);
    } else {
      store_row_tile<T, D>(dkrow, dk, kmul, fa.hi, p.dk_f32 != 0);
    }
    store_row_tile<T, D>(dvrow, dv, 1.f, fa.hi, p.dv_f32 != 0);
  }
  }   // pass
#ifdef FCSA_TRACE
  if (blockIdx.x == gridDim.x / 2 + 3 && (tid & 63) == 0 && wave < 4) ts.dump(g_trace_dkv + 32 * wave, trace_now() - trace_t0);
#endif
}

#ifdef FCSA_TRACE
}  // namespace fcsa
extern "C" int fcsa_trace_read_dkv(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_dkv), sizeof(unsigned long long) * 128);
}
namespace fcsa {
#endif

// =============================================================================================
// Wide dK / dV kernel (16-bit types, no bias, D <= 64): every wave owns 64 keys = two 32-key blocks that share each
// Q / dO fragment (two MFMAs per LDS fragment read; the narrow kernel issues 32 LDS instructions per 16 MFMAs).
// Slot-scheduled like fwd2_kernel (one fenced issue slot per MFMA).  Per 32-row query block ib:
//
//     G1  dP(ib)    | exp / pack of key block 0 of ib | Q row + aux requests of ib+1
//     G2  S(ib+1)   | exp / pack of key block 1 of ib | dO^T requests of ib
//     G3  dV(ib)    | dS = P (dP - delta), pack       | Q^T requests of ib
//     G4  dK(ib)    |                                 | dO row requests of ib+1
//
// The per-row additive terms -- log2 of the normaliser for S, -delta for dP -- enter through ONE extra k-step of the S and
// dP chains: an "aux" fragment per row [lc_hi, lc_lo, nd_hi, nd_lo, 0...] (16-bit hi + lo split, |error| < 2^-16 |x|)
// against the constant B operands [1,1,0,0,..] / [0,0,1,1,..].  That replaces 8 ds_read_b128 per block (16 here) by one.
// Query tiles of 64 rows (two blocks) rotate through THREE LDS buffers, so one barrier per tile is enough although the
// S chain of the next tile's first block runs before the last transposed reads of the current tile.
// =============================================================================================
template <typename T, int D> struct Dkv2State {
  typedef TileGeom<D, 2> G;
  f32x16 dp[2];                                // dP - delta [key block]
  SecondB<T> pp[2], pd[2];                     // packed P and dS [key block]
  u32x4 qrow[G::KS], dorow[G::KS];             // row fragments in flight (S of the next block / dP of this block)
  u32x4 dot_tr[G::DB][2], q_tr[G::DB][2];      // transposed fragments in flight [feature block][16-row step]
};

template <typename T, int D, bool MASKED, typename Extra>
FCSA_DEV void dkv2_block(const char* qt_cur, const char* dot_cur, int rb_cur,
                         const char* qt_nxt, const char* dot_nxt, const char* aux_nxt_tile, int rb_nxt,
                         f32x16 (&s_cur)[2], f32x16 (&s_nxt)[2], const u32x4& aux_cur, u32x4& aux_nxt, Dkv2State<T, D>& st,
                         const u32x4 (&kf)[2][TileGeom<D, 2>::KS], const u32x4 (&vf)[2][TileGeom<D, 2>::KS],
                         const u32x4& b_s, const u32x4& b_dp,
                         f32x16 (&dk)[2][TileGeom<D, 2>::DB], f32x16 (&dv)[2][TileGeom<D, 2>::DB],
                         const FragAddr<T, D>& fa, const FragAddr<T, 16>& fax, const uint32_t (&w)[2], Trace& ts, int sb, Extra&& extra) {
  typedef TileGeom<D, 2> G;
  typedef Traits<T> TR;
  constexpr int NS = 2 * (G::KS + 1);    // MFMAs of the S / dP groups (aux k-step included)
  constexpr int NPV = 4 * G::DB;         // MFMAs of the dV / dK groups
  f32x16 zero;
#pragma unroll
  for (int e = 0; e < 16; ++e) zero[e] = 0.f;
  // two logits of key block kb: exp, (mask), keep P in f32 for dS, pack for dV
  auto exp2chunk = [&](int kb, int c) {
    float e0 = fast_exp2(s_cur[kb][2 * c]), e1 = fast_exp2(s_cur[kb][2 * c + 1]);
    if constexpr (MASKED) {
      e0 = ((w[kb] >> crow(2 * c, 0)) & 1u) ? e0 : 0.f;
      e1 = ((w[kb] >> crow(2 * c + 1, 0)) & 1u) ? e1 : 0.f;
    }
    s_cur[kb][2 * c] = e0;
    s_cur[kb][2 * c + 1] = e1;
    st.pp[kb].v[c >> 2][c & 3] = TR::pack2(e0, e1);
  };
  FCSA_FENCE();
  // ---- G1: dP(ib) | exp of key block 0 | Q row + aux requests of the next block
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    const int kk = m >> 1, kb = m & 1;
    st.dp[kb] = TR::mfma32(kk < G::KS ? st.dorow[kk < G::KS ? kk : 0] : aux_cur, kk < G::KS ? vf[kb][kk < G::KS ? kk : 0] : b_dp,
                           kk == 0 ? zero : st.dp[kb]);
    FCSA_SHARE(m, NS, 8, c) exp2chunk(0, c);
    FCSA_SHARE(m, NS, G::KS + 1, f) {
      if (f < G::KS) st.qrow[f < G::KS ? f : 0] = fa.row_frag(qt_nxt, rb_nxt, f);
      else aux_nxt = fax.row_frag(aux_nxt_tile, rb_nxt, 0);
    }
    extra(1, m);
    FCSA_FENCE();
  }
  FCSA_STAMP(ts, sb + 1);
  // ---- G2: S(ib+1) | exp of key block 1 | dO^T requests
#pragma unroll
  for (int m = 0; m < NS; ++m) {
    const int kk = m >> 1, kb = m & 1;
    s_nxt[kb] = TR::mfma32(kk < G::KS ? st.qrow[kk < G::KS ? kk : 0] : aux_nxt, kk < G::KS ? kf[kb][kk < G::KS ? kk : 0] : b_s,
                           kk == 0 ? zero : s_nxt[kb]);
    FCSA_SHARE(m, NS, 8, c) exp2chunk(1, c);
    FCSA_SHARE(m, NS, 2 * G::DB, f) st.dot_tr[f >> 1][f & 1] = fa.tr_frag(dot_cur, rb_cur + 16 * (f & 1), f >> 1);
    extra(2, m);
    FCSA_FENCE();
  }
  FCSA_STAMP(ts, sb + 2);
  // ---- G3: dV(ib) | dS = P (dP - delta), pack | Q^T requests
#pragma unroll
  for (int m = 0; m < NPV; ++m) {
    const int ks = m / (2 * G::DB), db = (m >> 1) % G::DB, kb = m & 1;
    TR::mfma32_agpr(st.dot_tr[db][ks], st.pp[kb].v[ks], dv[kb][db]);
    FCSA_SHARE(m, NPV, 16, c) {
      const int b2 = c >> 3, cc = c & 7;
      st.pd[b2].v[cc >> 2][cc & 3] = TR::pack2(s_cur[b2][2 * cc] * st.dp[b2][2 * cc], s_cur[b2][2 * cc + 1] * st.dp[b2][2 * cc + 1]);
    }
    FCSA_SHARE(m, NPV, 2 * G::DB, f) st.q_tr[f >> 1][f & 1] = fa.tr_frag(qt_cur, rb_cur + 16 * (f & 1), f >> 1);
    extra(3, m);
    FCSA_FENCE();
  }
  FCSA_STAMP(ts, sb + 3);
  // ---- G4: dK(ib) | dO row requests of the next block
#pragma unroll
  for (int m = 0; m < NPV; ++m) {
    const int ks = m / (2 * G::DB), db = (m >> 1) % G::DB, kb = m & 1;
    TR::mfma32_agpr(st.q_tr[db][ks], st.pd[kb].v[ks], dk[kb][db]);
    FCSA_SHARE(m, NPV, G::KS, f) st.dorow[f] = fa.row_frag(dot_nxt, rb_nxt, f);
    extra(4, m);
    FCSA_FENCE();
  }
}

template <typename T, int D, int NW>
__global__ void __launch_bounds__(NW * 64, 1) bwd_dkv2_kernel(const BwdParams p) {
  typedef TileGeom<D, 2> G;
  typedef TileGeom<16, 2> GX;                   // aux rows: 32 bytes (chunk 0 = the four aux values, chunk 1 = zeros)
  typedef Traits<T> TR;
  static_assert(TR::ES == 2, "16-bit types only");
  constexpr int KW = 64, BNK = KW * NW, NT = NW * 64, BMQ = 64;
  constexpr int TILE_B = BMQ * G::ROWB;
  constexpr int BUF_B = 2 * TILE_B + BMQ * GX::ROWB;             // Q tile | dO tile | aux rows
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][BUF_B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FragAddr<T, D> fa;
  fa.init(lane);
  FragAddr<T, 16> fax;
  fax.init(lane);

  const int KT = (p.M + BNK - 1) / BNK;
  const int PT = p.causal ? (KT + 1) / 2 : KT;
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (p.causal && (KT - 1 - pt) != pt) ? 2 : 1;
  const int diff = p.M - p.N;
  const uint32_t ncm = p.causal ? 0u : 0xffffffffu;
  const u32x4 z4 = {0u, 0u, 0u, 0u};
  u32x4 b_s = z4, b_dp = z4;                    // B operands of the aux k-step: k-slots 0,1 (S) / 2,3 (dP) of half 0
  if (fa.hi == 0) { b_s[0] = TR::kOne2; b_dp[1] = TR::kOne2; }
  const char* qbase = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh;
  const char* dobase = p.d_out.p + (int64_t)b * p.d_out.sb + (int64_t)h * p.d_out.sh;
  const float* invl_row = p.inv_l + ((int64_t)b * p.H + h) * p.N;
  const float* delta_row = p.delta + ((int64_t)b * p.H + h) * p.N;
  Stager<T, D, BMQ, NT> sq, sdo;
  sq.init(p.q.sn, tid);
  sdo.init(p.d_out.sn, tid);
  constexpr int PER = Stager<T, D, BMQ, NT>::PER;
  Trace ts;
  ts.reset();
#ifdef FCSA_TRACE
  const unsigned long long trace_t0 = trace_now();
#endif

  for (int pass = 0; pass < npass; ++pass) {
    const int kt = p.causal ? (pass == 0 ? pt : KT - 1 - pt) : pt;      // heavy tile first
    const int n0 = kt * BNK;
    const int nw = n0 + wave * KW;                        // first key of this wave
    const int QT = (p.N + BMQ - 1) / BMQ;
    int t0 = 0;
    if (p.causal) t0 = max(0, n0 - diff) / BMQ;

    // K, V fragments of this lane's two keys (B operands of S = Q K^T and dP = dO V^T), kept for the whole loop
    u32x4 kf[2][G::KS], vf[2][G::KS];
    uint32_t kmask[2];
    int jk[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int j = nw + 32 * kb + (lane & 31);
      jk[kb] = j;
      const char* krow = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)j * p.k.sn;
      const char* vrow = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh + (int64_t)j * p.v.sn;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) {
        kf[kb][kk] = z4;
        vf[kb][kk] = z4;
        if (j < p.M) {
          kf[kb][kk] = *reinterpret_cast<const u32x4*>(krow + (2 * kk + fa.hi) * 16);
          if (!p.q_scaled) kf[kb][kk] = scale_frag<T>(kf[kb][kk], p.c1);
          vf[kb][kk] = *reinterpret_cast<const u32x4*>(vrow + (2 * kk + fa.hi) * 16);
        }
      }
      bool key_ok = j < p.M;
      if (p.mask != nullptr && key_ok) key_ok = p.mask[(int64_t)b * p.M + j] != 0;
      kmask[kb] = key_ok ? 0xffffffffu : 0u;
    }
    f32x16 dk[2][G::DB], dv[2][G::DB];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[kb][db][r] = 0.f; dv[kb][db][r] = 0.f; }

    // ---- staging: Q / dO chunks through registers, the aux row by the first BMQ threads
    float lc_r = 0.f, dl_r = 0.f;
    bool row_ok = false;
    auto load_aux = [&](int t) {           // raw loads only
      if (tid < BMQ) {
        const int i = min(t * BMQ + tid, p.N - 1);
        lc_r = invl_row[i];
        dl_r = delta_row[i];
        row_ok = t * BMQ + tid < p.N;
      }
    };
    auto store_aux = [&](char* buf) {
      if (tid < BMQ) {
        // rows beyond N: lc = -1e4 makes P exactly 0 there (finite, so its hi / lo split is finite too)
        const float lc = row_ok ? __builtin_amdgcn_logf(lc_r) - p.c2 : -1e4f;
        const float nd = row_ok ? -dl_r : 0.f;
        const float lch = TR::lo(TR::pack2(lc, 0.f)), ndh = TR::lo(TR::pack2(nd, 0.f));
        u32x4 c0 = {TR::pack2(lch, lc - lch), TR::pack2(ndh, nd - ndh), 0u, 0u};
        char* ax = buf + 2 * TILE_B;
        *reinterpret_cast<u32x4*>(ax + GX::off(tid, 0)) = c0;
        *reinterpret_cast<u32x4*>(ax + GX::off(tid, 1)) = z4;
      }
    };
    auto rows_of = [&](int t) { return t < QT ? p.N - t * BMQ : 0; };      // 0 -> zero-record descriptor

    int bi = 0;                                   // LDS buffer of the current tile (0..2)
    if (t0 < QT) {
      sq.load(qbase + (int64_t)t0 * BMQ * p.q.sn, p.q.sn, rows_of(t0));
      sdo.load(dobase + (int64_t)t0 * BMQ * p.d_out.sn, p.d_out.sn, rows_of(t0));
      load_aux(t0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) on ALL paths (see fwd_kernel)
    f32x16 sA[2], sB[2];
    u32x4 auxA = z4, auxB = z4;
    Dkv2State<T, D> st;
    if (t0 < QT) {
      sq.store(smem, tid);
      sdo.store(smem + TILE_B, tid);
      store_aux(smem);
      sq.load(qbase + (int64_t)(t0 + 1) * BMQ * p.q.sn, p.q.sn, rows_of(t0 + 1));
      sdo.load(dobase + (int64_t)(t0 + 1) * BMQ * p.d_out.sn, p.d_out.sn, rows_of(t0 + 1));
      load_aux(t0 + 1);
    }
    __syncthreads();
    if (t0 < QT) {     // pipeline prologue: S of the first block, dO rows of the first block
      f32x16 zero;
#pragma unroll
      for (int e = 0; e < 16; ++e) zero[e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) st.qrow[kk] = fa.row_frag(smem, 0, kk);
      auxA = fax.row_frag(smem + 2 * TILE_B, 0, 0);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        sA[kb] = zero;
#pragma unroll
        for (int kk = 0; kk < G::KS; ++kk) sA[kb] = TR::mfma32(st.qrow[kk], kf[kb][kk], sA[kb]);
        sA[kb] = TR::mfma32(auxA, b_s, sA[kb]);
      }
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) st.dorow[kk] = fa.row_frag(smem + TILE_B, 0, kk);
    }

    // query tiles [t0, t_m) need masking for THIS wave, tiles [t_m, QT) do not (wave-uniform split, see bwd_dkv_kernel)
    int t_m = QT;
    if (p.mask == nullptr && n0 + BNK <= p.M) {
      t_m = t0;
      if (p.causal) t_m = min(QT, max(t0, (nw + KW - 1 - diff + BMQ - 1) / BMQ));
    }
    auto run = [&](auto masked_tag, int t_begin, int t_end) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      for (int t = t_begin; t < t_end; ++t) {
        const int i0 = t * BMQ;
        const char* cur = smem + bi * BUF_B;
        const int bn = bi == 2 ? 0 : bi + 1;
        char* nxt = smem + bn * BUF_B;
        uint32_t w0[2] = {0xffffffffu, 0xffffffffu}, w1[2] = {0xffffffffu, 0xffffffffu};
        if constexpr (MASKED) {
#pragma unroll
          for (int kb = 0; kb < 2; ++kb) {
            w0[kb] = kmask[kb] & (ge_mask(jk[kb] - diff - (i0 + 4 * fa.hi)) | ncm);
            w1[kb] = kmask[kb] & (ge_mask(jk[kb] - diff - (i0 + 32 + 4 * fa.hi)) | ncm);
          }
        }
        FCSA_STAMP(ts, 0);
        // block 0 (rows 0..31 of this tile); next block = rows 32..63 of the same tile.  Stage stores of tile t+1 ride in G3 / G4.
        dkv2_block<T, D, MASKED>(cur, cur + TILE_B, 0, cur, cur + TILE_B, cur + 2 * TILE_B, 32, sA, sB, auxA, auxB, st, kf, vf, b_s, b_dp,
                                 dk, dv, fa, fax, w0, ts, 0, [&](int g, int m) {
          if (g == 3) { FCSA_SHARE(m, 4 * G::DB, PER, x) sq.store_one(nxt, tid, x); }
          if (g == 4) { FCSA_SHARE(m, 4 * G::DB, PER, x) sdo.store_one(nxt + TILE_B, tid, x); }
        });
        FCSA_STAMP(ts, 4);
        store_aux(nxt);
        __syncthreads();                 // tile t+1 is visible; every read of tile t-1 (the buffer tile t+2 will take) has returned
        FCSA_STAMP(ts, 5);
        // block 1 (rows 32..63); next block = rows 0..31 of tile t+1.  Stage loads of tile t+2 ride in G3 / G4.
        const __amdgpu_buffer_rsrc_t qd = Stager<T, D, BMQ, NT>::descriptor(qbase + (int64_t)(t + 2) * BMQ * p.q.sn, p.q.sn, rows_of(t + 2));
        const __amdgpu_buffer_rsrc_t dd = Stager<T, D, BMQ, NT>::descriptor(dobase + (int64_t)(t + 2) * BMQ * p.d_out.sn, p.d_out.sn, rows_of(t + 2));
        dkv2_block<T, D, MASKED>(cur, cur + TILE_B, 32, nxt, nxt + TILE_B, nxt + 2 * TILE_B, 0, sB, sA, auxB, auxA, st, kf, vf, b_s, b_dp,
                                 dk, dv, fa, fax, w1, ts, 5, [&](int g, int m) {
          if (g == 3) { FCSA_SHARE(m, 4 * G::DB, PER, x) sq.load_one(qd, x); }
          if (g == 4) { FCSA_SHARE(m, 4 * G::DB, PER, x) sdo.load_one(dd, x); }
        });
        load_aux(t + 2);
        FCSA_STAMP(ts, 9);
        if constexpr (!MASKED) ts.close(9);
        bi = bn;
      }
    };
    run(std::true_type{}, t0, t_m);
    run(std::false_type{}, t_m, QT);
    __syncthreads();      // the next pass restages buffer 0 while slower waves may still read their last tile
    mfma_drain();         // dk / dv were accumulated by asm MFMAs the compiler does not know about

    const float kmul = p.q_scaled ? p.scale / p.c1 : p.scale;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int j = jk[kb];
      if (j < p.M) {
        char* dkrow = p.dk.p + (int64_t)b * p.dk.sb + (int64_t)h * p.dk.sh + (int64_t)j * p.dk.sn;
        char* dvrow = p.dv.p + (int64_t)b * p.dv.sb + (int64_t)h * p.dv.sh + (int64_t)j * p.dv.sn;
        if (p.rk != nullptr) {
          const char* xrow = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)j * p.k.sn;
          store_row_tile_l2norm_bwd<T, D>(dkrow, dk[kb], kmul, fa.hi, xrow, 1.f,
                                          p.rk + (((int64_t)b * p.H + h) * p.M + j) * p.G, p.lgm, p.norm_eps);
        } else {
          store_row_tile<T, D>(dkrow, dk[kb], kmul, fa.hi, p.dk_f32 != 0);
        }
        store_row_tile<T, D>(dvrow, dv[kb], 1.f, fa.hi, p.dv_f32 != 0);
      }
    }
  }   // pass
#ifdef FCSA_TRACE
  if (blockIdx.x == gridDim.x / 2 + 3 && (tid & 63) == 0 && wave < 4) ts.dump(g_trace_dkv + 32 * wave, trace_now() - trace_t0);
#endif
}

// ---------------------------------------------------------------------------------------------
template <typename K>
static hipError_t set_lds_once(K kern, size_t lds, bool& done) {
  if (done) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e == hipSuccess) done = true;
  return e;
}

// 8 waves per workgroup when the grid still gives every CU a workgroup (see row_tile_waves in fcsa_fwd.hip), else 4
static int tile_waves(int64_t batch_heads, int len, bool causal) {
  const int MT = (len + 255) / 256;
  return batch_heads * (causal ? (MT + 1) / 2 : MT) >= 224 ? 8 : 4;
}

template <typename T, int D, bool BIAS, int NW>
static hipError_t launch_dq_nw(const BwdParams& p, hipStream_t s) {
  constexpr int BM = 32 * NW;
  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  const size_t lds = 4 * 64 * TileGeom<D, Traits<T>::ES>::ROWB;
  auto kern = bwd_dq_kernel<T, D, NW, BIAS>;
  static bool attr_set = false;
  if (hipError_t e = set_lds_once(kern, lds, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.H * PT)), dim3(NW * 64), lds, s, p);
  return hipGetLastError();
}

template <typename T, int D, bool BIAS>
static hipError_t launch_dq_b(const BwdParams& p, hipStream_t s) {
  if constexpr (D * Traits<T>::ES <= FCSA_DQ_2W_BYTES) {
    if (tile_waves((int64_t)p.B * p.H, p.N, p.causal) == 8) return launch_dq_nw<T, D, BIAS, 8>(p, s);
  }
  return launch_dq_nw<T, D, BIAS, 4>(p, s);
}

template <typename T, int D, bool BIAS, int NW>
static hipError_t launch_dkv_nw(const BwdParams& p, hipStream_t s) {
  constexpr int BNK = 32 * NW;
  // staged query tile: 32 rows for wide feature rows (16-bit D >= 96, f32 D >= 64: VGPR budget of the staging registers),
  // else 64; 128 in the 8-wave form (one workgroup per CU: the LDS is there, and half the barriers per key tile: -4.5%)
  constexpr int BMQ = (D * Traits<T>::ES >= 192) ? 32 : (NW == 8 ? 128 : 64);
  const int KT = (p.M + BNK - 1) / BNK;
  const int PT = p.causal ? (KT + 1) / 2 : KT;
  const size_t lds = 2 * (2 * BMQ * TileGeom<D, Traits<T>::ES>::ROWB + 2 * BMQ * 4);
  auto kern = bwd_dkv_kernel<T, D, NW, BMQ, BIAS>;
  static bool attr_set = false;
  if (hipError_t e = set_lds_once(kern, lds, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.H * PT)), dim3(NW * 64), lds, s, p);
  return hipGetLastError();
}

template <typename T, int D>
static hipError_t launch_dkv2(const BwdParams& p, hipStream_t s) {
  constexpr int NW = 4, BNK = 64 * NW, BMQ = 64;
  const int KT = (p.M + BNK - 1) / BNK;
  const int PT = p.causal ? (KT + 1) / 2 : KT;
  const size_t lds = 3 * (2 * BMQ * TileGeom<D, 2>::ROWB + BMQ * TileGeom<16, 2>::ROWB);
  auto kern = bwd_dkv2_kernel<T, D, NW>;
  static bool attr_set = false;
  if (hipError_t e = set_lds_once(kern, lds, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.H * PT)), dim3(NW * 64), lds, s, p);
  return hipGetLastError();
}

template <typename T, int D, bool BIAS>
static hipError_t launch_dkv_b(const BwdParams& p, hipStream_t s) {
#ifndef FCSA_FORCE_NARROW_DKV      // (A/B measurement builds only)
  if constexpr (Traits<T>::ES == 2 && !BIAS && (D == 32 || D == 64)) {
    if (tile_waves((int64_t)p.B * p.H, p.M, p.causal) == 8) return launch_dkv2<T, D>(p, s);      // >= 224 workgroups of 256 keys
  }
#endif
  if constexpr (D * Traits<T>::ES <= FCSA_DKV_2W_BYTES) {
    if (tile_waves((int64_t)p.B * p.H, p.M, p.causal) == 8) return launch_dkv_nw<T, D, BIAS, 8>(p, s);
  }
  return launch_dkv_nw<T, D, BIAS, 4>(p, s);
}

template <typename T, int D> static hipError_t launch_dq_t(const BwdParams& p, hipStream_t s) {
  return p.bias != nullptr ? launch_dq_b<T, D, true>(p, s) : launch_dq_b<T, D, false>(p, s);
}
template <typename T, int D> static hipError_t launch_dkv_t(const BwdParams& p, hipStream_t s) {
  return p.bias != nullptr ? launch_dkv_b<T, D, true>(p, s) : launch_dkv_b<T, D, false>(p, s);
}

#define FCSA_DISPATCH_D(FN, T)                      \
  switch (D) {                                      \
    case 16:  return FN<T, 16>(p, s);               \
    case 32:  return FN<T, 32>(p, s);               \
    case 64:  return FN<T, 64>(p, s);               \
    case 96:  return FN<T, 96>(p, s);               \
    case 128: return FN<T, 128>(p, s);              \
    default:  return hipErrorInvalidValue;          \
  }

hipError_t launch_backward_dq(int dtype, int D, const BwdParams& p, hipStream_t s) {
  if (p.B * p.H == 0 || p.N == 0) return hipSuccess;
  if (dtype == 2) { FCSA_DISPATCH_D(launch_dq_t, BF16) }
  if (dtype == 1) { FCSA_DISPATCH_D(launch_dq_t, F16) }
  if (dtype == 0) { FCSA_DISPATCH_D(launch_dq_t, F32) }
  return hipErrorInvalidValue;
}

hipError_t launch_backward_dkv(int dtype, int D, const BwdParams& p, hipStream_t s) {
  if (p.B * p.H == 0 || p.M == 0) return hipSuccess;
  if (dtype == 2) { FCSA_DISPATCH_D(launch_dkv_t, BF16) }
  if (dtype == 1) { FCSA_DISPATCH_D(launch_dkv_t, F16) }
  if (dtype == 0) { FCSA_DISPATCH_D(launch_dkv_t, F32) }
  return hipErrorInvalidValue;
}

}  // namespace fcsa
