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
//                    Writes dk / dv per (batch, q-head); with single-headed K/V f32 slabs are
//                    reduced over heads by the finalize kernel (fcsa_norm.hip) instead of the
//                    reference's f32 atomics (cu:1613-1619).
// Both kernels finish the l2norm backward in their epilogue (RowEpilogue::finish, through the LDS) when the group size allows,
// pair causal tiles (constant work per workgroup) and run with 4 or 8 waves per workgroup (8: tiles staged once per CU;
// dkv then stages 128-row query tiles, half the barriers).
//
// 7 tile products instead of the reference's 5 (S and dP are recomputed in both kernels); results
// are deterministic, d_bias included (a workgroup owns its slice of d_bias and reduces the broadcast index itself: no atomics,
// cf. cu:1574-1576).
// As in the forward kernel, each kernel runs its unmasked tiles and its masked tiles in two
// sequential loops with one straight-line body each (no accumulator copies at if/else joins); the masked loop selects per
// logit in the causal instantiations and adds a key mask / ragged tail as a rank-1 MFMA per block in the non-causal ones (KM).
// Grids that cannot fill the chip: the dQ kernel splits the key range, the dK/dV kernel the query range (gridDim.y, f32 slabs,
// finalize kernel).
#include <type_traits>

#include "fcsa_common.cuh"
#include "fcsa_kernels.h"
#ifdef FCSA_VAR_SPLIT_ENV
#include "dev/fcsa_sweep_env.h"
#endif

namespace fcsa {
// Tuning constants (each settled by a same-box A/B on MI355X; the rejected alternatives are listed in DESIGN.md §8):
constexpr int kDq2WBytes = 128;      // row bytes D*ES up to which the dQ kernel runs its 8-wave form (two waves / SIMD, one workgroup per CU)
// Two waves per SIMD (<= 256 registers) for the dQ kernel (template parameter TWO): always for rows up to 128 bytes, and -- round 3 --
// for 16-bit rows up to 256 bytes (D = 96, 128) in the 4-wave form WHEN the grid puts two 128-row workgroups on every CU, whose waves
// then hide each other's LDS latency (launch_dq_b).  Those widths ran one wave per SIMD before (435 registers at D = 128), at ~40 %
// of what the same kernel reaches at D = 64; smaller grids still do (a lone wave is better off with the pipelined tile).
template <typename T, int D> constexpr bool dq_can_two_waves() { return D * Traits<T>::ES <= (Traits<T>::ES == 2 ? 256 : 128); }
constexpr int kDkv2WBytes = 128;     // same for the dKV kernel
constexpr int kDqSub8 = 4;           // 64-key tiles per LDS stage of the 8-wave dQ kernel (16 bit): one barrier per 256 keys
constexpr bool kDqSplitFirstStage = true;      // cold first stage of a dQ pass requested in two parts (bwd_dq_kernel, SPLIT0)
constexpr int kDkvBmq8 = 128;        // staged query rows of the 8-wave dKV kernel
constexpr int kDkvBmqWide = 64;      // staged query rows of the dKV kernel for 16-bit D >= 96 (LDS-DMA form)
constexpr bool kDkvRing = true;
      // three-buffer ring + tile pipeline across the tile barrier (dkv_tile_pipe)
#ifdef FCSA_TRACE
__device__ unsigned long long g_trace_dkv[128];
__device__ unsigned long long g_trace_dq[128];
#endif
#ifdef FCSA_TRACE_BAR      // per wave of one workgroup: ticks spent at the tile barrier / in the tile loops (two s_memtime per tile, nothing else)
__device__ unsigned long long g_trace_bar_dkv[64];
__device__ unsigned long long g_trace_bar_dq[64];
#define FCSA_BAR_BEGIN(v) do { asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(v)); } while (0)
#define FCSA_BAR_END(v, acc) do { unsigned long long e_; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(e_)); acc += e_ - v; } while (0)
#else
#define FCSA_BAR_BEGIN(v) ((void)0)
#define FCSA_BAR_END(v, acc) ((void)0)
#endif
#ifdef FCSA_TRACE_WG
__device__ unsigned long long g_trace_wg_dkv[2048];      // per workgroup: [2 * id] = start time, [2 * id + 1] = end time (wave 0)
__device__ unsigned long long g_trace_wg_dq[2048];
__device__ unsigned long long g_trace_pass_dq[2560];       // per workgroup (first 256): [pass][5] pass marks of wave 0
__device__ unsigned long long g_trace_pass_dkv[2560];
#endif

// =============================================================================================
// dQ kernel
// =============================================================================================
// MODE: 0 = every pair valid, 1 = causal diagonal tiles (select per logit), 2 = key mask / ragged tail of a non-causal launch
// (rank-1 MFMA per block, key_mask_rank1)
template <typename T, int D, int MODE, bool BIAS, bool TWO>
FCSA_DEV void dq_tile(const char* kt, const char* vt, const FragAddr<T, D>& fa,
                      const u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS], const u32x4 (&dof)[TileGeom<D, Traits<T>::ES>::KS],
                      f32x16 (&dq)[TileGeom<D, Traits<T>::ES>::DB], float lc, float delta, const BwdParams& p, uint64_t word,
                      uint32_t ncm, int i, int j0, int diff, const char* bias_row, int m_lim) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr bool MASKED = MODE == 1, KEYM = MODE == 2;
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    // branch-free and before the MFMA chains on purpose (see fwd_tile)
    uint32_t w = 0xffffffffu;
    if constexpr (MASKED)
      w = ((uint32_t)(word >> (32 * jb)) >> (4 * fa.hi)) & (le_mask(i + diff - (j0 + 32 * jb + 4 * fa.hi)) | ncm);
    // request every row fragment of this 32-key block first (just-in-time ds_read_b128 in front of their
    // dependent MFMA were the most expensive item of the forward tile, see fwd_tile)
    // (requests are batched PF k-steps at a time so the live fragments stay within the register budget)
    constexpr int PF = G::KS <= 4 ? G::KS : (G::KS >= 8 && TWO ? 1 : 2);      // (16-bit D = 128 at two waves per SIMD: 256 registers)
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = lc; dp[r] = -delta; }  // lc = log2(inv_l) - c2 and -delta ride in as initial values
    if constexpr (KEYM) s = key_mask_rank1<T>(s, (uint32_t)(word >> (32 * jb)), fa.row_off / G::ROWB, fa.hi);
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
    if constexpr (BIAS)
      load_bias_block<T>(bv, bias_row, jbase, m_lim, (p.M & 3) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0, p.bias_c,
                         (p.M & 7) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0, fa.hi);      // (key numbering of this workgroup's key range, like j0)
    // (exponentials first, products second: a v_exp_f32 directly followed by the multiply that consumes it costs a hazard nop
    //  plus the transcendental latency, and hipcc schedules the interleaved form that way under register pressure)
    f32x16 pe;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = s[r];               // = c1 * qh.kh + lc already (c1 rides on q, lc is the accumulator's initial value)
      if constexpr (BIAS) x += bv[r];
      pe[r] = fast_exp2(x);
    }
    if constexpr (MASKED) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pe[r] = ((w >> crow(r, 0)) & 1u) ? pe[r] : 0.f;
    }
    mul16(s, pe, dp);                    // dS; dp already holds dP - delta
    SecondB<T> pb;
    pb.prep(s);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) dq[db] = second_mma<T, D>(dq[db], kt, 32 * jb, db, pb, fa);
  }
}

// Software-pipelined form of dq_tile (16-bit types, no bias), same idea as dkv_tile_pipe: every LDS request is issued one
// stage before its consumer.  Per 32-key block:  M1 (S^T, dP^T chains from the K / V row fragments requested during the previous
// block's M2) | T (transposed K fragments for dQ, requested behind the M1 MFMAs) | X (exp, dS, pack) | R (next block's row
// fragments, interleaved with) M2 (dQ^T += K^T dS^T).  The state crosses tile boundaries inside an LDS stage.  Measured: no gain
// at two waves per SIMD (D = 64: the partner wave already hides the latency), -x% at one wave per SIMD (D >= 96), where it is used.
template <typename T, int D> struct DqPipe {
  typedef TileGeom<D, Traits<T>::ES> G;
  u32x4 kfr[G::KS], vfr[G::KS];
  int tile;                          // key tile whose block-0 fragments are in flight (-1: none)
  FCSA_DEV void request(const char* kt, const char* vt, const FragAddr<T, D>& fa, int jb) {
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) { kfr[kk] = fa.row_frag(kt, 32 * jb, kk); vfr[kk] = fa.row_frag(vt, 32 * jb, kk); }
  }
};

template <typename T, int D, int MODE>
FCSA_DEV void dq_tile_pipe(const char* kt, const char* vt, const char* knext, const char* vnext, int next_tile, int t,
                           const FragAddr<T, D>& fa, const u32x4 (&qf)[TileGeom<D, Traits<T>::ES>::KS],
                           const u32x4 (&dof)[TileGeom<D, Traits<T>::ES>::KS], f32x16 (&dq)[TileGeom<D, Traits<T>::ES>::DB],
                           float lc, float delta, uint64_t word, uint32_t ncm, int i, int j0, int diff, DqPipe<T, D>& pp_) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr bool MASKED = MODE == 1, KEYM = MODE == 2;
  if (pp_.tile != t) pp_.request(kt, vt, fa, 0);        // first tile of a stage (or after skipped tiles): exposed request
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    uint32_t w = 0xffffffffu;
    if constexpr (MASKED)
      w = ((uint32_t)(word >> (32 * jb)) >> (4 * fa.hi)) & (le_mask(i + diff - (j0 + 32 * jb + 4 * fa.hi)) | ncm);
    FCSA_FENCE();
    // ---- M1
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[r] = lc; dp[r] = -delta; }     // lc = log2(inv_l) - c2 and -delta ride in as initial values
    if constexpr (KEYM) s = key_mask_rank1<T>(s, (uint32_t)(word >> (32 * jb)), fa.row_off / G::ROWB, fa.hi);
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s = TR::mfma32(pp_.kfr[kk], qf[kk], s);
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) dp = TR::mfma32(pp_.vfr[kk], dof[kk], dp);
    // ---- T
    u32x4 ktr[G::DB][2];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) { ktr[db][0] = fa.tr_frag(kt, 32 * jb, db); ktr[db][1] = fa.tr_frag(kt, 32 * jb + 16, db); }
    FCSA_FENCE();
    // ---- X
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float e = fast_exp2(s[r]);
      if constexpr (MASKED) e = ((w >> crow(r, 0)) & 1u) ? e : 0.f;
      s[r] = e * dp[r];               // dp already holds dP - delta
    }
    SecondB<T> pb;
    pb.prep(s);
    FCSA_FENCE();
    // ---- R (branch-free: past the last tile of a stage the request reads a valid but unrelated LDS address and is re-issued
    //      at the top of the next tile) interleaved with M2
    if (jb == 0) pp_.request(kt, vt, fa, 1);
    else pp_.request(knext, vnext, fa, 0);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      dq[db] = TR::mfma32(ktr[db][0], pb.v[0], dq[db]);
      dq[db] = TR::mfma32(ktr[db][1], pb.v[1], dq[db]);
    }
    constexpr int NM2 = 2 * G::DB, NR = 2 * G::KS;
#pragma unroll
    for (int m = 0; m < NM2; ++m) {
      __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, (NR + NM2 - 1) / NM2, 0);
    }
  }
  pp_.tile = next_tile;
}

// LDS plan of the dQ kernel (EpiLds): the next iteration is requested ahead of the epilogue when the stages arrive by LDS-DMA and
// the scratch fits behind them.
// K / V stages of the dQ kernel arrive by LDS-DMA for the 16-bit types and for 512-byte rows (f32, D = 128: the staging registers of
// the register path -- 64 + 16 per lane -- are what pushed that instantiation into scratch)
template <typename T, int D> constexpr bool dq_dma(int sub) {
  return (Traits<T>::ES == 2 || D * Traits<T>::ES >= 512) && (64 * sub * TileGeom<D, Traits<T>::ES>::ROWB) % 1024 == 0;
}
template <typename T, int D, int NW, int SUB, bool TWO, bool WHOLE_CU = false> struct DqLds      // (NW: waves with an epilogue of their own)
    : EpiLds<T, D, NW, 4 * 64 * SUB * TileGeom<D, Traits<T>::ES>::ROWB, dq_dma<T, D>(SUB),
             ((NW == 8 || !TWO || WHOLE_CU) ? 160 : 80) * 1024> {};

// LDS plan of the dKV kernel: two staging buffers of (Q tile | dO tile | lc | -delta), epilogue scratch behind them when it fits.
// LEAN (16-bit rows of 129 .. 256 bytes, two waves per SIMD): the V rows of the workgroup's own keys live in the LDS behind the staging
// buffers instead of in registers (VOWN bytes); the epilogue scratch then shares those bytes (never "SEP").
// NBUF: staging buffers -- 2, or 3 in the ring form (dkv_ring: the tile after the current one is always complete in the LDS).
template <typename T, int D, int NW, int BMQ, bool BIAS, bool LEAN = false, int NBUF = 2> struct DkvLds
    : EpiLds<T, D, NW, NBUF * (2 * BMQ * TileGeom<D, Traits<T>::ES>::ROWB + 2 * BMQ * 4) + (LEAN ? 32 * NW * TileGeom<D, Traits<T>::ES>::ROWB : 0),
             NBUF == 2 && !LEAN && Traits<T>::ES == 2 && !BIAS && (BMQ * TileGeom<D, Traits<T>::ES>::ROWB) % 1024 == 0,
             ((NW == 8 || (D * Traits<T>::ES > kDkv2WBytes && !LEAN)) ? 160 : 80) * 1024> {
  static constexpr int VOWN = NBUF * (2 * BMQ * TileGeom<D, Traits<T>::ES>::ROWB + 2 * BMQ * 4);      // byte offset of the own-V tile (LEAN)
};

// SUB = 64-key tiles per LDS stage: 1, or 2 / 4 in the 8-wave form (one workgroup per CU has the LDS for 128- / 256-key stages).
// The phase trace showed the waves of this kernel waiting 25 % of their time at the per-tile barrier; one barrier per 128 keys
// halves that (the same change gave the dKV kernel 4.5 %), one per 256 keys (LDS-DMA staging: no staging registers) another 1.2 %.
// KM: the launch is not causal; tiles that need masking take the rank-1 form (see fwd_kernel)
// KSPLIT (8 waves, the two-wave tile with or without a bias, SUB = 2): the workgroup owns 128 query rows and its wave halves split the KEYS -- waves 0-3 take
// the even 64-key tile of a stage, waves 4-7 the odd one -- and add their dQ partials through the LDS at the end of the pass (see
// fwd_kernel, KSPLIT): for grids of at most one 128-row workgroup per CU, whose four waves would each have a SIMD to themselves.
template <typename T, int D, int NW, bool BIAS, int SUB, bool TWO, bool KM, bool KSPLIT = false>
__global__ void __launch_bounds__(NW * 64, (TWO ? 2 : 1)) bwd_dq_kernel(const BwdParams p) {
  static_assert(!KSPLIT || (NW == 8 && TWO && SUB == 2 && Traits<T>::ES == 2), "key-split form: 8 waves, two-wave tile, 16 bit, 2 tiles per stage");
  // (same type and value as p.causal: the causal instantiations compile to what they were.  The key-split form's !KM twin is only ever
  //  launched causal -- launch_dq_b -- and says so: at 256-byte rows the kernel sits at its 256 registers)
  const int causal = KM ? 0 : KSPLIT ? 1 : p.causal;
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr int RWAVES = KSPLIT ? NW / 2 : NW;      // waves that own distinct row slices
  constexpr int BN = 64, BM = 32 * RWAVES, NT = NW * 64, BNS = BN * SUB;
  constexpr int TILE_B = BN * G::ROWB;          // one 64-key tile of K or V
  constexpr int HALF_B = SUB * TILE_B;          // K (or V) part of a stage
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K stage | V stage]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rwave = KSPLIT ? (wave & (RWAVES - 1)) : wave;      // row slice of this wave
  const int half = KSPLIT ? wave / RWAVES : 0;                  // KSPLIT: which tile of a stage
  FragAddr<T, D> fa;
  fa.init(lane);

  // causal: a workgroup takes the PAIR of row tiles (MT-1-pt, pt) -> constant work per workgroup (see fwd_kernel)
  const int MT = (p.N + BM - 1) / BM;
  const int PT = causal ? (MT + 1) / 2 : MT;
  // (d_bias is not this kernel's business: bwd_dbias_kernel below recomputes the dS tiles of a bias slice and writes it once)
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (causal && (MT - 1 - pt) != pt) ? 2 : 1;
  // split-key launches (gridDim.y = p.dq_splits > 1): this workgroup sees the keys [k_lo, k_lo + Mk) only and
  // writes its partial dQ^ (f32) to slab blockIdx.y; the finalize kernel sums the slabs (and applies the l2norm backward).
  // Like the forward's split (fcsa_fwd.hip), for grids whose row tiles cannot fill the chip.
  // Causal launches (round 6) split the key range of EACH row tile -- the keys up to its diagonal -- so the window is set per pass
  // (geometry below): the pair (MT-1-pt, pt) keeps its constant work, 1 / dq_splits of it per workgroup.
  int k_lo = 0, Mk = p.M;
  if (p.dq_splits > 1 && !causal) {
    const int tps = ((p.M + BN - 1) / BN + p.dq_splits - 1) / p.dq_splits;      // 64-key tiles per split
    k_lo = (int)blockIdx.y * tps * BN;
    Mk = max(0, min(p.M, k_lo + tps * BN) - k_lo);
  }
  int diff = p.M - p.N - k_lo;
  const uint32_t ncm = causal ? 0u : 0xffffffffu;   // OR-ed into the causal bit mask: all ones when not causal
  Trace ts;
  ts.reset();
#ifdef FCSA_TRACE_WG
  const unsigned long long trace_t0 = trace_now();
#endif
#ifdef FCSA_TRACE
  unsigned long long pass_marks[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
#define FCSA_PASS_MARK(k) pass_marks[pass][k] = trace_now()
#elif defined(FCSA_TRACE_WG)
#define FCSA_PASS_MARK(k) do { if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 256) g_trace_pass_dq[blockIdx.x * 10 + pass * 5 + (k)] = trace_now(); } while (0)
#else
#define FCSA_PASS_MARK(k) ((void)0)
#endif
  // K / V stages: LDS-DMA for 16-bit types (no staging registers, no ds_write passes; see DmaStager), else through registers.
  constexpr bool DMA = dq_dma<T, D>(SUB);
  // SEP: the epilogue scratch has its own LDS bytes behind the staging buffers.  Then nothing of one (row tile) iteration has to
  // be finished before the next one starts loading: the first K / V stage and this lane's Q^ / dO / O row chunks of the NEXT
  // iteration are requested before the epilogue of the current one and land while it runs (the pass marks of the WG trace showed
  // 11 % of this kernel in prologues and 7 % in epilogues: exposed round trips, every workgroup of the chip in step).
  typedef DqLds<T, D, RWAVES, SUB, TWO, KSPLIT> LDS;
  constexpr bool SEP = LDS::SEP;
  Stager<T, D, BNS, NT> sk, sv;
  typedef DmaStager<T, D, DMA ? BNS : 1024, NW> DS;
  DS dk_, dv_;
  typename DS::Stream stk, stv;       // K / V walked stage by stage from key k_lo (DMA form): one descriptor per pass
  uint32_t k_step = 0, v_step = 0, lds0 = 0;
  bool far = false;
  if constexpr (DMA) {
    dk_.init(p.k.sn, wave, lane);
    dv_.init(p.v.sn, wave, lane);
    k_step = (uint32_t)(BNS * p.k.sn);
    v_step = (uint32_t)(BNS * p.v.sn);
    far = BNS * p.k.sn > (int64_t)DS::REBASE || BNS * p.v.sn > (int64_t)DS::REBASE;
    lds0 = DS::lds_addr(smem);
  } else {
    sk.init(p.k.sn, tid);
    sv.init(p.v.sn, tid);
  }
  // geometry of iteration (pass_): row tile, this lane's row, number of 64-key tiles
  auto geometry = [&](int pass_, int& m0_, int& nt_) {
    const int mt_ = causal ? (pass_ == 0 ? MT - 1 - pt : pt) : pt;      // heavy tile first
    m0_ = mt_ * BM;
    if constexpr (!KM && !KSPLIT && !BIAS && NW == 4) {      // (the form causal split launches take; the others compile to what they were)
      if (p.dq_splits > 1 && causal) {            // this row tile's visible keys [0, vis), split over gridDim.y workgroups
        const int vis = max(0, min(p.M, m0_ + BM + p.M - p.N));
        const int tps = max(1, ((vis + BN - 1) / BN + p.dq_splits - 1) / p.dq_splits);
        k_lo = min((int)blockIdx.y * tps * BN, p.M);
        Mk = max(0, min(p.M, k_lo + tps * BN) - k_lo);
        diff = p.M - p.N - k_lo;
      }
    }
    int last_key = Mk - 1;
    if (causal) last_key = min(last_key, m0_ + BM - 1 + diff);
    nt_ = last_key < 0 ? 0 : last_key / BN + 1;
  };
  // requests of an iteration that need nothing but free staging buffers: first K / V stage (DMA form) and the raw row chunks
  u32x4 rq_[G::KS], rdo_[G::KS], ro_[G::KS];
  float rinvl = 1.f;
  float rinv_n[RowEpilogue<T, D>::NP];
#pragma unroll
  for (int e = 0; e < RowEpilogue<T, D>::NP; ++e) rinv_n[e] = 1.f;
  bool have_pre = false;
  // SPLIT0: a COLD first stage (requested at the top of a pass, nothing to overlap it with: every workgroup of the chip asks for its
  // 64 KiB stage and its 96 KiB of row chunks at once, ~11 B/clk/CU) is requested in two parts -- the first 64-key tile ahead of the
  // row chunks, the other tiles of the stage at the top of the first tile, landing while it is computed; one extra barrier behind
  // that tile publishes them.  Needs piece i of every wave == tile i of the stage.  (A stage requested from inside the previous
  // pass's epilogue is not split: its flight is covered.)
  // (not with a bias: its loads inside the tile would be younger than the stage pieces the counted wait must leave in flight)
  constexpr bool SPLIT0 = !KSPLIT && DMA && !BIAS && SUB > 1 && DS::PER == SUB && (NW * 1024) / G::ROWB == BN && kDqSplitFirstStage;
  // The counted wait below (vmcnt(2 * PER)) publishes tiles 1.. of the cold stage only if (a) no VM load / store is issued between the
  // stage-1 DMA requests and the wait, and (b) tile 0 reads nothing but its own LDS tile.  Both hold for the two-wave tile without a
  // bias (dq_tile<.., TWO>: LDS reads and MFMAs only); the pipelined one-wave tile prefetches the NEXT tile's fragments and must not
  // be combined with it.
  static_assert(!SPLIT0 || (TWO && !BIAS), "SPLIT0's counted vmcnt needs a tile body without VM operations or next-tile LDS reads");
  bool split0 = false;
  auto request_ahead = [&](int b_, int h_, int pass_, bool cold) {
    int m0_, nt_;
    geometry(pass_, m0_, nt_);
    if constexpr (DMA) {
      stk = dk_.open(p.k.p + (int64_t)b_ * p.k.sb + (int64_t)h_ * p.k.sh + (int64_t)k_lo * p.k.sn, p.k.sn, Mk);
      stv = dv_.open(p.v.p + (int64_t)b_ * p.v.sb + (int64_t)h_ * p.v.sh + (int64_t)k_lo * p.v.sn, p.v.sn, Mk);
      split0 = SPLIT0 && cold && nt_ > 1;
      if (nt_ > 0) {
        if (SPLIT0 && split0) {
          dk_.issue_piece(stk, lds0, 0, wave);
          dv_.issue_piece(stv, lds0 + HALF_B, 0, wave);
        } else {
          dk_.issue(stk, lds0, wave);
          dv_.issue(stv, lds0 + HALF_B, wave);
        }
      }
    }
    if constexpr (!SEP) return;      // (without SEP the rows are loaded where they are used: fewer registers live at once)
    const int ln = opaque(lane), hi_ = ln >> 5;
    const int i_ = m0_ + rwave * 32 + (ln & 31);
    const char* qrow = p.q.p + (int64_t)b_ * p.q.sb + (int64_t)h_ * p.q.sh + (int64_t)i_ * p.q.sn;
    const char* dorow = p.d_out.p + (int64_t)b_ * p.d_out.sb + (int64_t)h_ * p.d_out.sh + (int64_t)i_ * p.d_out.sn;
    const char* orow = p.o.p + (int64_t)b_ * p.o.sb + (int64_t)h_ * p.o.sh + (int64_t)i_ * p.o.sn;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      rq_[kk] = z;
      rdo_[kk] = z;
      ro_[kk] = z;
      if (i_ < p.N) {
        rq_[kk] = *reinterpret_cast<const u32x4*>(qrow + (2 * kk + hi_) * 16);
        rdo_[kk] = *reinterpret_cast<const u32x4*>(dorow + (2 * kk + hi_) * 16);
        ro_[kk] = *reinterpret_cast<const u32x4*>(orow + (2 * kk + hi_) * 16);
      }
    }
    rinvl = i_ < p.N ? p.inv_l[((int64_t)b_ * p.H + h_) * p.N + i_] : 1.f;
    // inverse norms of the rows this lane FINISHES in the epilogue (fused l2norm backward): loaded here, carried through the tile
    // loops (NP registers), so that the epilogue has no global read to wait for
    const int rows_valid_ = p.N - (m0_ + rwave * 32);
    typedef RowEpilogue<T, D> EP;
    if (p.rq != nullptr && rows_valid_ > 0)
      EP::load_inv(rinv_n, p.rq + (((int64_t)b_ * p.H + h_) * p.N + m0_ + rwave * 32) * p.G, p.G, p.lgm, ln, rows_valid_);
  };

#ifdef FCSA_TRACE_BAR
  unsigned long long bar_wait = 0, bar_loop = 0;
#endif
  for (int pass = 0; pass < npass; ++pass) {
  FCSA_PASS_MARK(0);
  int m0, nt;
  geometry(pass, m0, nt);
  const int mw = m0 + rwave * 32;
  const int i = mw + (lane & 31);
  const char* kbase = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)k_lo * p.k.sn;
  const char* vbase = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh + (int64_t)k_lo * p.v.sn;
  // (the first stage is requested ahead of the row chunks and the delta reduction -- or, with SEP, before the previous epilogue)
  if (!have_pre) request_ahead(b, h, pass, true);

  // Q, dO fragments (B operands) and delta = <dO_i, O_i>  (replaces backward_preprocess, cu:1256-1335)
  u32x4 qf[G::KS], dof[G::KS];
  float delta = 0.f, lc = 0.f;
  float rinv[RowEpilogue<T, D>::NP];
  if constexpr (SEP) {
#pragma unroll
    for (int e = 0; e < RowEpilogue<T, D>::NP; ++e) rinv[e] = rinv_n[e];
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      qf[kk] = p.q_scaled ? rq_[kk] : scale_frag<T>(rq_[kk], p.c1);
      dof[kk] = rdo_[kk];
      delta += dot_frag<T>(rdo_[kk], ro_[kk]);
    }
  } else {
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
    rinvl = i < p.N ? p.inv_l[((int64_t)b * p.H + h) * p.N + i] : 1.f;
  }
  delta = xhalf_sum(delta);
  if (i < p.N) {
    const int64_t ridx = ((int64_t)b * p.H + h) * p.N + i;
    lc = (p.invl_log2 ? rinvl : __builtin_amdgcn_logf(rinvl)) - p.c2;     // v_log_f32 = log2
    if (fa.hi == 0 && blockIdx.y == 0 && half == 0) p.delta[ridx] = delta;
  }

  f32x16 dq[G::DB];
#pragma unroll
  for (int db = 0; db < G::DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;

  const uint8_t* mrow = p.mask ? p.mask + (int64_t)b * p.M + k_lo : nullptr;
  const char* bias_row = nullptr;                 // row min(i, N-1): always a valid address
  if constexpr (BIAS) {
    const int64_t boff = ((int64_t)(p.bias_batch ? b : h) * p.N + min(i, p.N - 1)) * (int64_t)p.M + k_lo;      // first key of this workgroup's range
    bias_row = p.bias + boff * (int64_t)sizeof(typename TR::elem);
  }

  // stages u = t / SUB of SUB 64-key tiles: loads of stage u+1 are issued at the first tile of stage u, stored after its last
  // tile, one barrier per stage (double-buffered LDS)
  uint8_t mb = 1;
  const int nst = (nt + SUB - 1) / SUB;
  if (nt > 0) {
    if constexpr (!DMA) {
      sk.load(kbase, p.k.sn, Mk);
      sv.load(vbase, p.v.sn, Mk);
    }
    if (mrow) mb = half * BN + lane < Mk ? mrow[half * BN + lane] : (uint8_t)0;
    if constexpr (DMA) {
      dma_wait();
    } else {
      sk.store(smem, tid);
      sv.store(smem + HALF_B, tid);
    }
  }
  __syncthreads();
  // Every prologue load (Q / dO / K / V fragments, first stage) is complete on the real path; say so on ALL paths.
  // Otherwise hipcc's waitcnt model keeps them pending along the no-tile path, the loop-header merge never
  // clears that, and each iteration re-waits with vmcnt(0) at its first MFMA -- right after issuing the next
  // stage's prefetch, which serialises the prefetch with the compute meant to hide it.
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), expcnt/lgkmcnt untouched

  int t_split = 0;                                 // see fwd_kernel
  if (!BIAS && mrow == nullptr) {
    t_split = Mk / BN;
    if (causal) t_split = min(t_split, max(0, mw + diff + 1) / BN);
    t_split = min(t_split, nt);
  }

#ifdef FCSA_TRACE_BAR
  unsigned long long bar_t = 0, loop_t = 0;
  FCSA_BAR_BEGIN(loop_t);
#endif
  DqPipe<T, D> pipe;
  pipe.tile = -1;
  auto run = [&](auto masked_tag, int t_begin, int t_end) {
    constexpr int MODE = decltype(masked_tag)::value;      // dq_tile: 0 all valid, 1 causal select, 2 key mask by rank-1 MFMA
    constexpr bool MASKED = MODE != 0;
    for (int t = t_begin; t < t_end; ++t) {
      const int j0 = t * BN;
      const int u = t / SUB, sub = t % SUB;
      const char* kcur = smem + (u & 1) * 2 * HALF_B + sub * TILE_B;
      const char* vcur = kcur + HALF_B;
      char* snxt = smem + ((u + 1) & 1) * 2 * HALF_B;
      const bool more = u + 1 < nst;                       // another stage follows
      const bool last_of_stage = sub == SUB - 1 || t + 1 >= nt;
      FCSA_STAMP(ts, 0);
      uint64_t word = 0;
      if constexpr (MASKED) {     // consume the mask byte BEFORE issuing new loads (see fwd_kernel)
        word = __ballot((j0 + lane) < Mk && mb != 0);
        if (mrow && t + 1 < nt) {
          const int key = j0 + BN + lane;
          mb = key < Mk ? mrow[key] : (uint8_t)0;
        }
      }
      if constexpr (SPLIT0) {
        if (split0 && t == 0) {      // the rest of the cold first stage (ahead of stage 1's requests: it is needed first)
#pragma unroll
          for (int i = 1; i < DS::PER; ++i) {
            dk_.issue_piece(stk, lds0, i, wave);
            dv_.issue_piece(stv, lds0 + HALF_B, i, wave);
          }
        }
      }
      if (sub == 0 && more) {       // the buffer of stage u + 1 was last read in stage u - 1, which ended with a barrier
        if constexpr (DMA) {
          stk.off += k_step;
          stv.off += v_step;
          if (far || (stk.off | stv.off) > DS::REBASE) {      // 32-bit offsets about to run out: re-open at this stage
            stk = dk_.open(kbase + (int64_t)(u + 1) * BNS * p.k.sn, p.k.sn, Mk - (u + 1) * BNS);
            stv = dv_.open(vbase + (int64_t)(u + 1) * BNS * p.v.sn, p.v.sn, Mk - (u + 1) * BNS);
          }
          const uint32_t lds_nxt = lds0 + ((u + 1) & 1) * 2 * HALF_B;
          dk_.issue(stk, lds_nxt, wave);
          dv_.issue(stv, lds_nxt + HALF_B, wave);
        } else {
          sk.load(kbase + (int64_t)(u + 1) * BNS * p.k.sn, p.k.sn, Mk - (u + 1) * BNS);
          sv.load(vbase + (int64_t)(u + 1) * BNS * p.v.sn, p.v.sn, Mk - (u + 1) * BNS);
        }
      }
      FCSA_STAMP(ts, 1);
      if constexpr (kPrioBwd == 1 && NW == 8 && SUB >= 2) {
        if (sub == 0) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
        if (sub == SUB / 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); }
      }
      if constexpr (TR::ES == 2 && !BIAS && !TWO) {      // pipelined tile: one wave per SIMD only
        bool skip = false;
        if constexpr (MASKED) skip = causal && (j0 > mw + 31 + diff);
        const bool next_here = !last_of_stage;              // the next key tile sits in this stage's buffer
        if (!skip) dq_tile_pipe<T, D, tile_mode<MODE>()>(kcur, vcur, next_here ? kcur + TILE_B : kcur, next_here ? vcur + TILE_B : vcur,
                                              next_here ? t + 1 : -1, t, fa, qf, dof, dq, lc, delta, word, ncm, i, j0, diff, pipe);
      } else if constexpr (MASKED) {
        const bool skip = causal && (j0 > mw + 31 + diff);
        if (!skip) dq_tile<T, D, tile_mode<MODE>(), BIAS, TWO>(kcur, vcur, fa, qf, dof, dq, lc, delta, p, word, ncm, i, j0, diff, bias_row, Mk);
      } else {
        dq_tile<T, D, 0, BIAS, TWO>(kcur, vcur, fa, qf, dof, dq, lc, delta, p, 0, ncm, i, j0, diff, bias_row, Mk);
      }
      FCSA_STAMP(ts, 2);
      if constexpr (SPLIT0) {
        if (split0 && t == 0 && !last_of_stage) {      // publish tiles 1 .. of the first stage (stage 1's pieces -- 2 * PER, younger -- stay in flight)
          if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * DS::PER) : "memory");
          else dma_wait();
          __syncthreads();
        }
      }
      if (last_of_stage) {                                 // workgroup-uniform
        // (a ragged last stage, or a causal pass that ends early, can leave the younger half at priority 1: drop it before the merge /
        //  epilogue / next prologue, where the older half does the stores)
        if constexpr (kPrioBwd == 1 && NW == 8 && SUB >= 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(0); }
        if (more) {
          if constexpr (DMA) {
            dma_wait();
          } else {
            sk.store(snxt, tid);
            sv.store(snxt + HALF_B, tid);
          }
        }
        FCSA_STAMP(ts, 3);
        FCSA_BAR_BEGIN(bar_t);
        __syncthreads();
        FCSA_BAR_END(bar_t, bar_wait);
      }
      FCSA_STAMP(ts, 4);
      if constexpr (!MASKED) ts.close(4);
    }
  };
  FCSA_PASS_MARK(1);
  if constexpr (!KSPLIT) {
    run(std::integral_constant<int, 0>{}, 0, t_split);
    FCSA_PASS_MARK(2);
    run(std::integral_constant<int, KM ? 2 : 1>{}, t_split, nt);
  } else {
    // stage u = tiles 2u, 2u + 1; this wave's tile is t = 2u + half; one barrier per stage for every wave (see fwd_kernel, KSPLIT)
    const int u_split = min(nst, max(0, (t_split - half + 1) / 2));
    auto stage = [&](auto masked_tag, int u) {
      constexpr int MODE = decltype(masked_tag)::value;
      const int t = 2 * u + half, j0 = t * BN;
      const char* kcur = smem + (u & 1) * 2 * HALF_B + half * TILE_B;
      const char* vcur = kcur + HALF_B;
      const bool more = u + 1 < nst;
      uint64_t word = 0;
      if constexpr (MODE != 0) {
        word = __ballot((j0 + lane) < Mk && mb != 0);
        if (mrow && t + 2 < nt) {
          const int key = j0 + 2 * BN + lane;
          mb = key < Mk ? mrow[key] : (uint8_t)0;
        }
      }
      if (more) {       // the buffer of stage u + 1 was last read in stage u - 1, which ended with a barrier
        stk.off += k_step;
        stv.off += v_step;
        if (far || (stk.off | stv.off) > DS::REBASE) {
          stk = dk_.open(kbase + (int64_t)(u + 1) * BNS * p.k.sn, p.k.sn, Mk - (u + 1) * BNS);
          stv = dv_.open(vbase + (int64_t)(u + 1) * BNS * p.v.sn, p.v.sn, Mk - (u + 1) * BNS);
        }
        const uint32_t lds_nxt = lds0 + ((u + 1) & 1) * 2 * HALF_B;
        dk_.issue(stk, lds_nxt, wave);
        dv_.issue(stv, lds_nxt + HALF_B, wave);
      }
      bool skip = t >= nt;
      if constexpr (MODE == 1) skip = skip || (causal && j0 > mw + 31 + diff);
      if (!skip) dq_tile<T, D, tile_mode<MODE>(), BIAS, TWO>(kcur, vcur, fa, qf, dof, dq, lc, delta, p, word, ncm, i, j0, diff, bias_row, Mk);
      if (more) dma_wait();
      FCSA_BAR_BEGIN(bar_t);
      __syncthreads();
      FCSA_BAR_END(bar_t, bar_wait);
    };
    for (int u = 0; u < u_split; ++u) stage(std::integral_constant<int, 0>{}, u);
    FCSA_PASS_MARK(2);
    for (int u = u_split; u < nst; ++u) stage(std::integral_constant<int, KM ? 2 : 1>{}, u);
  }
  FCSA_PASS_MARK(3);
#ifdef FCSA_TRACE_BAR
  FCSA_BAR_END(loop_t, bar_loop);
#endif
  if constexpr (KSPLIT) {
    // the odd-tile half hands its dQ partials to the even-tile half of the same rows (the staging buffers are free: every stage ended
    // with a barrier); done before anything of the epilogue or the next pass touches the LDS
    f32x4* ms = reinterpret_cast<f32x4*>(smem) + rwave * (G::DB * 4 * 64) + lane;
    if (half == 1) {
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {dq[db][4 * g], dq[db][4 * g + 1], dq[db][4 * g + 2], dq[db][4 * g + 3]};
          ms[(db * 4 + g) * 64] = v;
        }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = ms[(db * 4 + g) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) dq[db][4 * g + e] += v[e];
        }
    }
    __syncthreads();
  }

  // Epilogue through the LDS (RowEpilogue): every stage ended with a barrier, so no wave still reads the staging buffers.
  // With SEP, what the NEXT iteration needs from memory is requested between the epilogue's steps -- after the accumulators went to
  // the scratch (their registers are free) -- and lands while the epilogue finishes.  The epilogue itself reads nothing from
  // global memory (inverse norms: loaded with the row chunks; normalised rows: qf, still live), so it never waits for them.
  {
    typedef RowEpilogue<T, D> EP;
    char* scr = LDS::scratch(smem, rwave);
    char* xs = LDS::xarea(smem, rwave);
    const int rows_valid = half == 0 ? p.N - mw : 0;      // (KSPLIT: the odd half has handed its partials over and only takes part in the requests)
    const bool fused = p.rq != nullptr;      // dq = l2norm_backward(scale * dS K^): p.q holds c1 * q^ (or q^), contiguous rows
    const bool xreg = LDS::X && fused;      // (fused implies l2norm_qk, i.e. q_scaled: qf is exactly the stored row c1 * q^)
    const int le = opaque(lane);
    if constexpr (!SEP) {      // (with SEP the inverse norms came with the row chunks)
#pragma unroll
      for (int e = 0; e < EP::NP; ++e) rinv[e] = 1.f;
      if (fused && rows_valid > 0) EP::load_inv(rinv, p.rq + (((int64_t)b * p.H + h) * p.N + mw) * p.G, p.G, p.lgm, le, rows_valid);
    }
    if (rows_valid > 0) EP::put(scr, dq, p.scale, le, xreg ? qf : nullptr, xs, LDS::XPITCH);                  // cu:1580-1582: dS *= scale
    have_pre = false;
    if constexpr (SEP) {
      if (pass + 1 < npass) {
        request_ahead(b, h, pass + 1, false);
        have_pre = true;
      }
    }
    if (rows_valid > 0) {
      char* dq0 = p.dq.p + (int64_t)b * p.dq.sb + (int64_t)h * p.dq.sh + (int64_t)mw * p.dq.sn + (int64_t)blockIdx.y * p.dq_split_stride;
      const char* x0 = fused ? p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh + (int64_t)mw * p.q.sn : nullptr;
      EP::template finish<LDS::X>(scr, xs, LDS::XPITCH, le, dq0, p.dq.sn, rows_valid, fused ? false : p.dq_f32 != 0, x0, p.q.sn, p.q_scaled ? 1.f / p.c1 : 1.f,
                                  rinv, p.lgm, p.norm_eps);
    }
    if constexpr (!SEP) {
      if (pass + 1 < npass) __syncthreads();      // the scratch overlaps the staging buffers of the next pass
    }
  }
  FCSA_PASS_MARK(4);
  }   // pass
#undef FCSA_PASS_MARK
#ifdef FCSA_TRACE_BAR
  if (blockIdx.x == gridDim.x / 2 + 3 && blockIdx.y == 0 && lane == 0) { g_trace_bar_dq[2 * wave] = bar_wait; g_trace_bar_dq[2 * wave + 1] = bar_loop; }
#endif
#ifdef FCSA_TRACE_WG
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 1024) { g_trace_wg_dq[2 * blockIdx.x] = trace_t0; g_trace_wg_dq[2 * blockIdx.x + 1] = trace_now(); }
#endif
#ifdef FCSA_TRACE
  if (blockIdx.x == gridDim.x / 2 + 3 && (tid & 63) == 0 && (wave & 2) == 0) {      // waves 0, 1, 4, 5
    unsigned long long* out = g_trace_dq + 32 * ((wave & 1) + 2 * (wave >> 2));
    ts.dump(out, trace_now() - trace_t0);
    for (int ps = 0; ps < 2; ++ps)
      for (int k = 0; k < 4; ++k) out[14 + 4 * ps + k] = pass_marks[ps][k + 1] - pass_marks[ps][k];   // prologue | unmasked tiles | masked tiles | epilogue
  }
#endif
}

// =============================================================================================
// d_bias kernel.  d_bias[slice, i, j] = sum over the OTHER index (batch for a per-head bias, heads for a per-batch bias) of
// dS[b, h, i, j].  The reference pushes every dS element through an f32 atomic (cu:1574-1576).  Round 2 first made a dQ workgroup
// own (bias slice, row tile) and add its dS blocks to d_bias with read-modify-writes; that is deterministic but slow (a strided
// 16-byte read and write per lane and block, and a grid of only slices x row tiles workgroups).  This kernel recomputes S and dP
// once more for ONE [128 queries x 64 keys] tile of ONE bias slice, runs the reduced index in a loop, keeps the sum in registers
// and writes the tile once, transposed through the LDS into whole rows: no atomics, no read-modify-write, slices x row tiles x
// key tiles workgroups, and the dQ kernel runs its plain form.  Needs delta (published by the dQ kernel, launched first).
// =============================================================================================
// Registers: two row-fragment sets (current + requested-ahead), two staging sets, 64 bias / sum values -- two waves per SIMD (256
// registers) only fit rows of <= 128 bytes; wider rows run one wave per SIMD, and 512-byte rows (f32, D = 128) load their row
// fragments at the top of the iteration instead of one iteration ahead.  (Round 2 asked for 2 waves / SIMD up to 256-byte rows and
// spilled 60 - 145 registers to scratch there.)
template <typename T, int D>
__global__ void __launch_bounds__(256, (D * Traits<T>::ES <= 128 ? 2 : 1)) bwd_dbias_kernel(const BwdParams p) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr int BN = 64, BM = 128, NT = 256;
  constexpr int TILE_B = BN * G::ROWB;
  constexpr int OPITCH = 64 * 4 + 16;                                    // output scratch row: 64 f32 + pad
  extern __shared__ __attribute__((aligned(16))) char smem[];           // K tile | V tile | 4 x [32][OPITCH] output scratch

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FragAddr<T, D> fa;
  fa.init(lane);
  const int MT = (p.N + BM - 1) / BM;
  const int owner = blockIdx.x / MT, mt = blockIdx.x % MT;
  const int j0 = (int)blockIdx.y * BN;                                   // first key of this workgroup's tile
  const int n_red = p.bias_batch ? p.H : p.B;
  const int m0 = mt * BM, mw = m0 + wave * 32, i = mw + (lane & 31);
  const int diff = p.M - p.N;
  const uint32_t ncm = p.causal ? 0u : 0xffffffffu;
  typedef typename TR::elem E;                                           // d_bias has the dtype of the bias (cu:1912 casts; here it is written once)
  E* const dbias = reinterpret_cast<E*>(p.d_bias);
  if (p.causal && j0 > m0 + BM - 1 + diff) {                             // tile entirely above the diagonal: d_bias = 0 (workgroup-uniform)
    for (int e = tid; e < BM * (BN / 4); e += NT) {
      const int row = m0 + e / (BN / 4), col = j0 + 4 * (e % (BN / 4));
      if (row < p.N)
        for (int c = col; c < min(col + 4, p.M); ++c) dbias[((int64_t)owner * p.N + row) * p.M + c] = (E)0.f;
    }
    return;
  }

  // bias values of this lane's row for the tile: the same for every reduced index
  float bv[2][16];
  {
    const char* brow = p.bias + (((int64_t)owner * p.N + min(i, p.N - 1)) * (int64_t)p.M + j0) * (int64_t)sizeof(typename TR::elem);
    const bool aligned = (p.M & 3) == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
      load_bias_block<T>(bv[jb], brow, 32 * jb + 4 * fa.hi, p.M - j0, aligned, p.bias_c, aligned && (p.M & 7) == 0, fa.hi);
  }
  f32x16 acc[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[jb][r] = 0.f;

  Stager<T, D, BN, NT> sk, sv;
  sk.init(p.k.sn, tid);
  sv.init(p.v.sn, tid);
  // everything an iteration reads from global memory is requested one iteration ahead (its K / V tile into the staging registers,
  // this lane's row chunks and per-row terms into n*): the round trip overlaps the previous iteration's tile
  constexpr bool ROWS_AHEAD = G::KS < 16;
  u32x4 nq[ROWS_AHEAD ? G::KS : 1], ndo[ROWS_AHEAD ? G::KS : 1];
  float ninvl = 1.f, ndelta = 0.f;
  uint8_t nmask = 1;
  auto load_rows = [&](int red_, u32x4* rq, u32x4* rdo) {
    const int b = p.bias_batch ? owner : red_, h = p.bias_batch ? red_ : owner;
    const char* qrow = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh + (int64_t)i * p.q.sn;
    const char* dorow = p.d_out.p + (int64_t)b * p.d_out.sb + (int64_t)h * p.d_out.sh + (int64_t)i * p.d_out.sn;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      rq[kk] = z;
      rdo[kk] = z;
      if (i < p.N) {
        rq[kk] = *reinterpret_cast<const u32x4*>(qrow + (2 * kk + fa.hi) * 16);
        rdo[kk] = *reinterpret_cast<const u32x4*>(dorow + (2 * kk + fa.hi) * 16);
      }
    }
  };
  auto request = [&](int red_) {
    const int b = p.bias_batch ? owner : red_, h = p.bias_batch ? red_ : owner;
    sk.load(p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)j0 * p.k.sn, p.k.sn, p.M - j0);
    sv.load(p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh + (int64_t)j0 * p.v.sn, p.v.sn, p.M - j0);
    if constexpr (ROWS_AHEAD) load_rows(red_, nq, ndo);
    if (i < p.N) {
      const int64_t ridx = ((int64_t)b * p.H + h) * p.N + i;
      ninvl = p.inv_l[ridx];
      ndelta = p.delta[ridx];
    }
    nmask = (p.mask != nullptr) ? p.mask[(int64_t)b * p.M + min(j0 + lane, p.M - 1)] : (uint8_t)1;
  };
  request(0);
  for (int red = 0; red < n_red; ++red) {
    // this lane's row: Q^ and dO fragments (B operands), log2-normaliser and delta
    u32x4 qf[G::KS], dof[G::KS];
    if constexpr (ROWS_AHEAD) {
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) {
        qf[kk] = p.q_scaled ? nq[kk] : scale_frag<T>(nq[kk], p.c1);
        dof[kk] = ndo[kk];
      }
    } else {
      load_rows(red, qf, dof);
      if (!p.q_scaled) {
#pragma unroll
        for (int kk = 0; kk < G::KS; ++kk) qf[kk] = scale_frag<T>(qf[kk], p.c1);
      }
    }
    const float lc = i < p.N ? (p.invl_log2 ? ninvl : __builtin_amdgcn_logf(ninvl)) - p.c2 : 0.f, delta = i < p.N ? ndelta : 0.f;
    const uint64_t word = __ballot((j0 + lane) < p.M && nmask != 0);     // valid keys of this tile for this batch element
    __syncthreads();                                                      // the previous iteration's fragment reads are done
    sk.store(smem, tid);
    sv.store(smem + TILE_B, tid);
    __syncthreads();
    if (red + 1 < n_red) request(red + 1);                                // (the staging registers and n* are free again)
    if (p.causal && j0 > mw + 31 + diff) continue;                        // nothing visible for this wave's rows (wave-uniform; barriers already passed)
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {
      const uint32_t w = ((uint32_t)(word >> (32 * jb)) >> (4 * fa.hi)) & (le_mask(i + diff - (j0 + 32 * jb + 4 * fa.hi)) | ncm);
      f32x16 sacc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sacc[r] = lc; dp[r] = -delta; }
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) sacc = TR::mfma32(fa.row_frag(smem, 32 * jb, kk), qf[kk], sacc);
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) dp = TR::mfma32(fa.row_frag(smem + TILE_B, 32 * jb, kk), dof[kk], dp);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = ((w >> crow(r, 0)) & 1u) ? fast_exp2(sacc[r] + bv[jb][r]) : 0.f;
        acc[jb][r] += e * dp[r];                                          // dS (cu:1574: before the scale factor)
      }
    }
  }

  // [32 rows x 64 keys] of this wave -> scratch (C layout: lane (row, hi) holds keys 32 * jb + 8 * rq + 4 * hi + 0..3) -> rows
  __syncthreads();                                                        // (scratch is separate from the tiles, but keep the waves' LDS traffic apart)
  char* scr = smem + 2 * TILE_B + wave * 32 * OPITCH;
  {
    const int x = lane & 31;
#pragma unroll
    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
      for (int rq = 0; rq < 4; ++rq) {
        const f32x4 v = {acc[jb][4 * rq], acc[jb][4 * rq + 1], acc[jb][4 * rq + 2], acc[jb][4 * rq + 3]};
        *reinterpret_cast<f32x4*>(scr + x * OPITCH + (32 * jb + 8 * rq + 4 * fa.hi) * 4) = v;
      }
    const bool vec = (p.M & 3) == 0 && (reinterpret_cast<uintptr_t>(p.d_bias) & 15) == 0;
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      const int row = 4 * pp + (lane >> 4), col = 4 * (lane & 15);
      const f32x4 v = *reinterpret_cast<const f32x4*>(scr + row * OPITCH + col * 4);
      if (mw + row < p.N) {
        E* o = dbias + ((int64_t)owner * p.N + mw + row) * (int64_t)p.M + j0 + col;
        if (vec && j0 + col + 3 < p.M) {      // 16 lanes write one row's 64 keys: 256 (f32) / 128 (16 bit) contiguous bytes
          if constexpr (TR::ES == 4) {
            *reinterpret_cast<f32x4*>(o) = v;
          } else {
            const u32x2 u = {TR::pack2(v[0], v[1]), TR::pack2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(o) = u;
          }
        } else {
          for (int c = 0; c < 4; ++c)
            if (j0 + col + c < p.M) o[c] = (E)v[c];
        }
      }
    }
  }
}

template <typename T, int D>
static hipError_t launch_dbias_t(const BwdParams& p, hipStream_t s) {
  const int MT = (p.N + 127) / 128, KT = (p.M + 63) / 64;
  const int64_t owners = p.bias_batch ? p.B : p.H;
  const size_t lds = 2 * 64 * TileGeom<D, Traits<T>::ES>::ROWB + 4 * 32 * (64 * 4 + 16);
  auto kern = bwd_dbias_kernel<T, D>;
  static std::atomic<uint64_t> lds_ok{0};
  if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(owners * MT), (unsigned)KT), dim3(256), lds, s, p);
  return hipGetLastError();
}

// =============================================================================================
// dK / dV kernel
// =============================================================================================
// LEAN: the B operand of the dP chain (this lane's V row) is read from `vown`, the LDS copy of the workgroup's own V rows, next to
// each MFMA instead of being held in registers (`vf` is unused): with the dk / dv accumulators (128 registers at D = 128) and the K
// fragments this is what lets the kernel run two waves per SIMD at 16-bit D = 96 / 128.
// MODE: 0 / 1 as in dq_tile; 2 = key mask of a non-causal launch: the lanes ARE the keys here (S = Q K^T, columns = keys), so the
// rank-1 term is ones (A, query rows) x this lane's 0 / -inf (B): `kmb` = the B operand's k-slot-0 register pair, set up once per pass
template <typename T, int D, int BMQ, int MODE, bool BIAS, bool LEAN = false>
FCSA_DEV void dkv_tile(const char* qt, const char* dot, const float* lcs, const float* dls, const FragAddr<T, D>& fa,
                       const u32x4 (&kf)[TileGeom<D, Traits<T>::ES>::KS], const u32x4 (&vf)[TileGeom<D, Traits<T>::ES>::KS],
                       f32x16 (&dk)[TileGeom<D, Traits<T>::ES>::DB], f32x16 (&dv)[TileGeom<D, Traits<T>::ES>::DB],
                       const BwdParams& p, uint32_t kmask, uint32_t ncm, int j, int i0, int diff, const char* bias_col, Trace& ts,
                       BiasBlock<T>& bb, char* bscr, const char* bias_blk, bool bvec, int next_i0, int lane, const char* vown = nullptr,
                       int vrow0 = 0, bool young = false) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr bool MASKED = MODE == 1, KEYM = MODE == 2;
#pragma unroll
  for (int ib = 0; ib < BMQ / 32; ++ib) {
    if constexpr (LEAN) FCSA_FENCE();      // blocks stay apart: interleaved by the scheduler, two blocks' fragments do not fit 256 registers
    if constexpr (LEAN && kPrioLean == 1) {      // barrier interval = ONE or two blocks here: the younger half is favoured through the
      if (BMQ == 32 || ib == 0) { if (young) __builtin_amdgcn_s_setprio(1); }      // first half of it (see kPrioBwd)
    }
    if constexpr (BIAS) {
      if (bvec) {      // this block's bias values (requested one block ago) -> scratch; request the next block's (see BiasBlock)
        bb.stage(bscr, lane);
        const int rown = ib + 1 < BMQ / 32 ? i0 + 32 * (ib + 1) : next_i0;
        if (rown >= 0) bb.request(bias_blk, min(rown + (lane & 31), p.N - 1), (int64_t)p.M * TR::ES, fa.hi);
      }
    }
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
    if constexpr (KEYM) s = key_mask_rank1_cols<T>(s, kmask == 0u, fa.hi);
    // (row fragments are requested next to their MFMA here: this kernel sits at its register budget, and
    //  batching the requests as in dq_tile spills)
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s = TR::mfma32(fa.row_frag(qt, 32 * ib, kk), kf[kk], s);
    if constexpr (LEAN) {
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) dp = TR::mfma32(fa.row_frag(dot, 32 * ib, kk), fa.row_frag(vown, vrow0, kk), dp);
    } else {
#pragma unroll
      for (int kk = 0; kk < G::KS; ++kk) dp = TR::mfma32(fa.row_frag(dot, 32 * ib, kk), vf[kk], dp);
    }
    FCSA_STAMP(ts, 2 + 3 * ib);

    f32x16 pr;
    // (this lane's column of the transposed bias block; recomputed per block so that it does not live across the tile loops)
    const char* bcol = BIAS ? bscr + 4 * fa.hi * BiasBlock<T>::PITCH + (opaque(lane) & 31) * TR::ES : nullptr;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float x = s[r];                 // = c1 * qh.kh + lc already
      if constexpr (BIAS) {
        if (bvec) {
          x += (float)*reinterpret_cast<const typename TR::elem*>(bcol + crow(r, 0) * BiasBlock<T>::PITCH) * p.bias_c;
        } else {                // element loads; clamped row: always a valid address; rows >= N have P = 0 through lc = -inf
          const int i = min(i0 + 32 * ib + crow(r, 0) + 4 * fa.hi, p.N - 1);
          const typename TR::elem bv = *reinterpret_cast<const typename TR::elem*>(
              bias_col + ((int64_t)i * p.M + min(j, p.M - 1)) * (int64_t)sizeof(typename TR::elem));
          x += (float)bv * p.bias_c;
        }
      }
      float pe = fast_exp2(x);
      if constexpr (MASKED) pe = ((w >> crow(r, 0)) & 1u) ? pe : 0.f;
      pr[r] = pe;
      s[r] = pe * dp[r];             // dp already holds dP - delta
    }
    SecondB<T> pp, pd;
    pp.prep(pr);
    pd.prep(s);
    if constexpr (LEAN && kPrioLean == 1) {
      if (BMQ == 32) { if (young) __builtin_amdgcn_s_setprio(0); }
    }
    FCSA_STAMP(ts, 3 + 3 * ib);
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      dv[db] = second_mma<T, D>(dv[db], dot, 32 * ib, db, pp, fa);
      dk[db] = second_mma<T, D>(dk[db], qt, 32 * ib, db, pd, fa);
    }
    FCSA_STAMP(ts, 4 + 3 * ib);
    if constexpr (LEAN && kPrioLean == 1) {
      if (BMQ == 64 && ib == 0) { if (young) __builtin_amdgcn_s_setprio(0); }
    }
  }
}

// Software-pipelined form of dkv_tile (16-bit types, no bias).  Two findings shape it (MI355X, C3, `tools/ab_libs.py`):
//   * The plain form reads every A fragment right in front of the MFMA that consumes it; hipcc then funnels them through ONE
//     register quad (ds_read_b128 v[4:7] -> s_waitcnt lgkmcnt(0) -> v_mfma, 16 times per 32x32 block).  Here every LDS
//     request is issued one pipeline stage before its consumer:
//         M1(ib) : S and dP chains from the row fragments requested during M2(ib-1)
//         T(ib)  : the 4*DB transposed fragments of this block are requested right behind the M1 MFMAs ...
//         X(ib)  : ... and land during exp / pack
//         R(ib+1): the next block's row fragments are requested (their registers died in M1), interleaved with ...
//         M2(ib) : ... the dV / dK products
//     The per-query terms (log2-normaliser and -delta) stay accumulator seeds read from LDS (8 broadcast ds_read_b128 per
//     block).  Tried and dropped: feeding them through one extra MFMA k-step per chain (the staging thread splits each term into
//     three 16-bit pieces; -7 KiB of LDS reads per block, +2 MFMAs): 2 % slower at D = 64 and D = 128 once the tiles arrive by
//     LDS-DMA -- the matrix pipe, not the LDS, is the scarcer resource here.
template <typename T, int D, int BMQ>
struct DkvPipe {
  typedef TileGeom<D, Traits<T>::ES> G;
  u32x4 qa[G::KS], da[G::KS];      // A operands of the next S / dP chains
  f32x16 s, dp;                    // their accumulators, seeded with the log2-normaliser and -delta of the block's queries

  FCSA_DEV void request(const char* qt, const char* dot, const float* lcs, const float* dls, const FragAddr<T, D>& fa, int ib) {
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) {
      const f32x4 lc4 = *reinterpret_cast<const f32x4*>(lcs + 32 * ib + 8 * rq + 4 * fa.hi);
      const f32x4 nd4 = *reinterpret_cast<const f32x4*>(dls + 32 * ib + 8 * rq + 4 * fa.hi);
#pragma unroll
      for (int e = 0; e < 4; ++e) { s[4 * rq + e] = lc4[e]; dp[4 * rq + e] = nd4[e]; }
    }
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) { qa[kk] = fa.row_frag(qt, 32 * ib, kk); da[kk] = fa.row_frag(dot, 32 * ib, kk); }
  }
};

// RING (dkv_ring): the pipeline state crosses tile boundaries.  The staging buffers form a ring of three, so the NEXT tile (`nxt`: its
// Q tile, dO tile at nxt + TILE_B, per-query terms behind them) is complete in the LDS while this one is processed: its first block's
// fragments are requested during the last block's M2 products, i.e. BEFORE the tile barrier, and the first chains of a tile start
// right behind it.  (Two buffers: that request sits at the top of the tile, exposed -- block 0 of a tile took 1320 ticks against 880
// for the others, phase trace.)  `fresh`: nothing is in flight for this tile (first tile of a pass, or the previous one was skipped).
// QS (query-split form of the kernel): this wave works on BMQ rows of a staged tile of 2 * BMQ, starting `hq` rows (`hoff` bytes) into it;
// qt / dot / lcs / dls point at its share already, `nxt` at the next staged tile's start
template <typename T, int D, int BMQ, int MODE, bool RING = false, bool QS = false>
FCSA_DEV void dkv_tile_pipe(const char* qt, const char* dot, const float* lcs, const float* dls, const FragAddr<T, D>& fa,
                            const u32x4 (&kf)[TileGeom<D, Traits<T>::ES>::KS], const u32x4 (&vf)[TileGeom<D, Traits<T>::ES>::KS],
                            f32x16 (&dk)[TileGeom<D, Traits<T>::ES>::DB], f32x16 (&dv)[TileGeom<D, Traits<T>::ES>::DB],
                            uint32_t kmask, uint32_t ncm, int j, int i0, int diff, Trace& ts, DkvPipe<T, D, BMQ>& pp_, bool fresh = true,
                            const char* nxt = nullptr, bool young = false, int hoff = 0, int hq = 0) {
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr bool MASKED = MODE == 1, KEYM = MODE == 2;
  constexpr int NB = BMQ / 32;
  if (!RING || fresh) pp_.request(qt, dot, lcs, dls, fa, 0);
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
    uint32_t w = 0xffffffffu;
    if constexpr (MASKED) w = kmask & (ge_mask(j - diff - (i0 + 32 * ib + 4 * fa.hi)) | ncm);
    FCSA_FENCE();
    if constexpr (kPrioBwd == 1) {      // (`young` is wave-uniform and lives in an SGPR: a scalar branch around one s_setprio)
      if (ib == 0) { if (young) __builtin_amdgcn_s_setprio(1); }
      if (ib == NB / 2) { if (young) __builtin_amdgcn_s_setprio(0); }
    }
    if (ib == 1) FCSA_STAMP(ts, 2);
    // ---- M1: S = Q K^T + lc, dP = dO V^T - delta (the per-query terms are the accumulators' initial values)
    f32x16 s = pp_.s, dp = pp_.dp;
    if constexpr (KEYM) s = key_mask_rank1_cols<T>(s, kmask == 0u, fa.hi);
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) s = TR::mfma32(pp_.qa[kk], kf[kk], s);
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) dp = TR::mfma32(pp_.da[kk], vf[kk], dp);
    // ---- T: transposed fragments of this block (dO^T for dV, Q^T for dK)
    u32x4 td[G::DB][2], tq[G::DB][2];
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      td[db][0] = fa.tr_frag(dot, 32 * ib, db);
      td[db][1] = fa.tr_frag(dot, 32 * ib + 16, db);
    }
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      tq[db][0] = fa.tr_frag(qt, 32 * ib, db);
      tq[db][1] = fa.tr_frag(qt, 32 * ib + 16, db);
    }
    FCSA_FENCE();
    if (ib == 1) FCSA_STAMP(ts, 3);
    // ---- X: P = exp2(s), dS = P * (dP - delta), packed in place
    f32x16 pr;
#pragma unroll
    for (int r = 0; r < 16; ++r) pr[r] = fast_exp2(s[r]);
    if constexpr (MASKED) {
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] = ((w >> crow(r, 0)) & 1u) ? pr[r] : 0.f;
    }
    mul16(s, pr, dp);
    SecondB<T> pb, db_;
    pb.prep(pr);
    db_.prep(s);
    FCSA_FENCE();
    if (ib == 1) FCSA_STAMP(ts, 4);
    // ---- R: next block's row fragments (their registers are dead now); RING: behind the last block, block 0 of the next tile
    if (ib + 1 < NB) pp_.request(qt, dot, lcs, dls, fa, ib + 1);
    else if constexpr (RING) {
      constexpr int BMT = QS ? 2 * BMQ : BMQ, TB = BMT * G::ROWB;
      if constexpr (QS)
        pp_.request(nxt + hoff, nxt + TB + hoff, reinterpret_cast<const float*>(nxt + 2 * TB) + hq, reinterpret_cast<const float*>(nxt + 2 * TB) + BMT + hq, fa, 0);
      else
        pp_.request(nxt, nxt + TB, reinterpret_cast<const float*>(nxt + 2 * TB), reinterpret_cast<const float*>(nxt + 2 * TB) + BMQ, fa, 0);
    }
    // ---- M2: dV^T += dO^T P, dK^T += Q^T dS
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      dv[db] = TR::mfma32(td[db][0], pb.v[0], dv[db]);
      dv[db] = TR::mfma32(td[db][1], pb.v[1], dv[db]);
    }
#pragma unroll
    for (int db = 0; db < G::DB; ++db) {
      dk[db] = TR::mfma32(tq[db][0], db_.v[0], dk[db]);
      dk[db] = TR::mfma32(tq[db][1], db_.v[1], dk[db]);
    }
    if (ib + 1 < NB || RING) {      // spread the R requests behind the first M2 MFMAs (hipcc otherwise sinks them below the products)
      constexpr int NM2 = 4 * G::DB, NR = 8 + 2 * G::KS;
#pragma unroll
      for (int m = 0; m < NM2; ++m) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);                                   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, (NR + NM2 - 1) / NM2, 0);              // its share of the LDS reads
      }
    }
    if (ib == 1) { FCSA_FENCE(); FCSA_STAMP(ts, 5); }
  }
}

// RING: three staging buffers instead of two and the tile pipeline crosses the tile barrier (dkv_tile_pipe); pipelined LDS-DMA form only
// QSPLIT (8 waves; the pipelined ring tile, or the generic two-buffer tile where a bias rides along): the workgroup owns 128 keys and its wave halves split the QUERIES of every staged tile -- waves 0-3 take its
// first BMQ / 2 rows, waves 4-7 the others, for the same four 32-key slices -- and add their dK / dV partials through the LDS at the end of
// the pass: the mirror image of the key-split forward / dQ forms, for grids of at most one 128-key workgroup per CU.
template <typename T, int D, int NW, int BMQ, bool BIAS, bool LEAN, bool KM, bool RING = false, bool QSPLIT = false>      // KM: not causal, masked tiles in the rank-1 form (see fwd_kernel)
__global__ void __launch_bounds__(NW * 64, ((D * Traits<T>::ES <= kDkv2WBytes || LEAN) ? 2 : 1)) bwd_dkv_kernel(const BwdParams p) {
  static_assert(!QSPLIT || (NW == 8 && !LEAN && (RING || BIAS) && BMQ % 64 == 0), "query-split form: 8 waves; pipelined ring tile, or the generic tile with a bias");
  const int causal = KM ? 0 : p.causal;      // (same type and value as p.causal: the causal instantiations compile to what they were)
  typedef TileGeom<D, Traits<T>::ES> G;
  typedef Traits<T> TR;
  constexpr int RWAVES = QSPLIT ? NW / 2 : NW;      // waves that own distinct key slices
  constexpr int BMS = QSPLIT ? BMQ / 2 : BMQ;       // rows of a staged tile one wave works on
  constexpr int BNK = 32 * RWAVES, NT = NW * 64;
  constexpr int TILE_B = BMQ * G::ROWB;
  static_assert(!LEAN || (Traits<T>::ES == 2 && !BIAS), "lean form: 16-bit types without bias");
  constexpr bool PIPE = Traits<T>::ES == 2 && !BIAS && !LEAN;      // software-pipelined tile (dkv_tile_pipe)
  constexpr int BUF_B = 2 * TILE_B + 2 * BMQ * 4;                // Q tile | dO tile | lc[BMQ] | -delta[BMQ]
  constexpr int NBUF = RING ? 3 : 2;
  static_assert(BMQ % 32 == 0 && BMQ <= NT, "query tile");
  static_assert(!RING || PIPE, "ring form: pipelined LDS-DMA tile only");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][BUF_B]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rwave = QSPLIT ? (wave & (RWAVES - 1)) : wave;      // key slice of this wave
  const int half = QSPLIT ? wave / RWAVES : 0;
  const int hq = half * BMS, hoff = hq * G::ROWB;               // QSPLIT: this wave's rows inside a staged tile
  FragAddr<T, D> fa;
  fa.init(lane);

  // causal: the LOW key tiles are the heavy ones (they see every later query); pair (pt, KT-1-pt) per workgroup
  const int KT = (p.M + BNK - 1) / BNK;
  const int PT = causal ? (KT + 1) / 2 : KT;
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (causal && (KT - 1 - pt) != pt) ? 2 : 1;
  const int diff = p.M - p.N;
  Trace ts;
  ts.reset();
#ifdef FCSA_TRACE_WG
  const unsigned long long trace_t0 = trace_now();
#endif
#ifdef FCSA_TRACE
  unsigned long long pass_marks[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
#define FCSA_PASS_MARK(k) pass_marks[pass][k] = trace_now()
#elif defined(FCSA_TRACE_WG)
#define FCSA_PASS_MARK(k) do { if (tid == 0 && blockIdx.x < 256) g_trace_pass_dkv[blockIdx.x * 10 + pass * 5 + (k)] = trace_now(); } while (0)
#else
#define FCSA_PASS_MARK(k) ((void)0)
#endif
  // (batch, head) is fixed for the workgroup: everything that does not depend on the pass is set up once
  // split-query launches (gridDim.y = p.dkv_splits > 1): this workgroup sees the query tiles [t_lo, QT) of its key tile
  // only and writes its partial dK^ / dV (f32) to slab blockIdx.y; the finalize kernel sums the slabs (and applies the l2norm backward).
  // For key grids that cannot fill the chip: few keys, many queries (the mirror image of the split-key forward / dQ).
  // Causal launches (round 6) split the query range of EACH key tile -- the tiles from its diagonal down -- so the window [t0, QT) is set
  // per pass (geometry below): the pair (pt, KT-1-pt) keeps its constant work, 1 / dkv_splits of it per workgroup.
  const int QT_all = (p.N + BMQ - 1) / BMQ;
  int QT = QT_all, t_lo = 0;
  if (p.dkv_splits > 1 && !causal) {
    const int tps = (QT + p.dkv_splits - 1) / p.dkv_splits;      // query tiles per split
    t_lo = (int)blockIdx.y * tps;
    QT = min(QT, t_lo + tps);
  }
  const char* qbase = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh;
  const char* dobase = p.d_out.p + (int64_t)b * p.d_out.sb + (int64_t)h * p.d_out.sh;
  const float* invl_row = p.inv_l + ((int64_t)b * p.H + h) * p.N;
  const float* delta_row = p.delta + ((int64_t)b * p.H + h) * p.N;
  // Q / dO tiles: LDS-DMA in the pipelined form (no staging registers, no ds_write passes), else through registers
  constexpr bool DMA = (PIPE || LEAN) && (BMQ * G::ROWB) % 1024 == 0;
  typedef DkvLds<T, D, NW, BMQ, BIAS, LEAN, NBUF> LDS;
  constexpr bool SEP = LDS::SEP;      // see bwd_dq_kernel: the next pass is requested from inside the epilogue of the current one
  static_assert(!RING || (DMA && !SEP), "ring form");
  Stager<T, D, BMQ, NT> sq, sdo;
  typedef DmaStager<T, D, DMA ? BMQ : 1024, NW> DS;
  DS dq_, ddo_;
  typename DS::Stream stq, stdo;      // Q / dO walked tile by tile from the pass's first tile (DMA form)
  uint32_t q_step = 0, do_step = 0, lds0 = 0;
  bool far = false;
  if constexpr (DMA) {
    dq_.init(p.q.sn, wave, lane);
    ddo_.init(p.d_out.sn, wave, lane);
    q_step = (uint32_t)(BMQ * p.q.sn);
    do_step = (uint32_t)(BMQ * p.d_out.sn);
    far = BMQ * p.q.sn > (int64_t)DS::REBASE || BMQ * p.d_out.sn > (int64_t)DS::REBASE;
    lds0 = DS::lds_addr(smem);
  } else {
    sq.init(p.q.sn, tid);
    sdo.init(p.d_out.sn, tid);
  }
  float lc_r = 0.f, dl_r = 0.f;
  bool row_ok = false;
  float lc_r2 = 0.f, dl_r2 = 0.f;      // RING: the pass prologue stages two tiles
  bool row_ok2 = false;
  auto load_rows = [&](int i0) {      // per-query terms of a tile; raw loads only: any arithmetic on them here would force an immediate vmcnt wait
    if (tid < BMQ) {
      const int i = min(i0 + tid, p.N - 1);
      lc_r = invl_row[i];
      dl_r = delta_row[i];
      row_ok = i0 + tid < p.N;
    }
  };
  // Q / dO streams move on to tile t (DMA form): one scalar add each; re-opened when the 32-bit offset would run out
  auto advance = [&](int t) {
    stq.off += q_step;
    stdo.off += do_step;
    if (far || (stq.off | stdo.off) > DS::REBASE) {
      stq = dq_.open(qbase + (int64_t)t * BMQ * p.q.sn, p.q.sn, p.N - t * BMQ);
      stdo = ddo_.open(dobase + (int64_t)t * BMQ * p.d_out.sn, p.d_out.sn, p.N - t * BMQ);
    }
  };
  // loads of tile t through registers (non-DMA form); `buf` = the LDS buffer it is going to
  auto load_tile = [&](int t, char* buf) {
    const int i0 = t * BMQ;
    sq.load(qbase + (int64_t)i0 * p.q.sn, p.q.sn, p.N - i0);
    sdo.load(dobase + (int64_t)i0 * p.d_out.sn, p.d_out.sn, p.N - i0);
    load_rows(i0);
  };
  // geometry of a pass: first key of the workgroup's key tile, first query tile it needs (causal keeps i >= j - diff)
  auto geometry = [&](int pass_, int& n0_, int& t0_) {
    const int kt_ = causal ? (pass_ == 0 ? pt : KT - 1 - pt) : pt;      // heavy tile first
    n0_ = kt_ * BNK;
    t0_ = causal ? max(0, n0_ - diff) / BMQ : t_lo;
    if constexpr (!KM && NW == 4 && !LEAN && !QSPLIT && !BIAS) {      // (the form causal split launches take; the others compile to what they were)
      if (p.dkv_splits > 1 && causal) {
        const int tps = max(1, (QT_all - t0_ + p.dkv_splits - 1) / p.dkv_splits);
        t0_ = min(QT_all, t0_ + (int)blockIdx.y * tps);
        QT = min(QT_all, t0_ + tps);
      }
    }
  };
  // requests of a pass that need nothing but a free staging buffer 0: first Q / dO tile (DMA form) with its per-query terms, this
  // lane's K / V fragments, its key-mask byte and the inverse norms its epilogue will use
  u32x4 rk_[G::KS], rv_[G::KS];
  uint8_t rmask = 1;
  float rinv_n[RowEpilogue<T, D>::NP];
#pragma unroll
  for (int e = 0; e < RowEpilogue<T, D>::NP; ++e) rinv_n[e] = 1.f;
  bool have_pre = false;
  auto request_ahead = [&](int pass_) {
    int n0_, t0_;
    geometry(pass_, n0_, t0_);
    if constexpr (DMA) {
      stq = dq_.open(qbase + (int64_t)t0_ * BMQ * p.q.sn, p.q.sn, p.N - t0_ * BMQ);
      stdo = ddo_.open(dobase + (int64_t)t0_ * BMQ * p.d_out.sn, p.d_out.sn, p.N - t0_ * BMQ);
      if (t0_ < QT) {
        dq_.issue(stq, lds0, wave);
        ddo_.issue(stdo, lds0 + TILE_B, wave);
        load_rows(t0_ * BMQ);
      }
      if constexpr (RING) {      // second tile of the pass -> buffer 1 (its per-query terms travel in the second register set)
        if (t0_ + 1 < QT) {
          advance(t0_ + 1);
          dq_.issue(stq, lds0 + BUF_B, wave);
          ddo_.issue(stdo, lds0 + BUF_B + TILE_B, wave);
          if (tid < BMQ) {
            const int i = min((t0_ + 1) * BMQ + tid, p.N - 1);
            lc_r2 = invl_row[i];
            dl_r2 = delta_row[i];
            row_ok2 = (t0_ + 1) * BMQ + tid < p.N;
          }
        }
      }
    }
    if constexpr (LEAN) {      // the V rows of this workgroup's keys -> LDS (rows past M are zero-filled by the descriptor's range check)
      DmaStager<T, D, BNK, NW> dvown_;      // (set up here, once per pass, from an opaque lane id: nothing of it lives across the tile loops)
      dvown_.init(p.v.sn, wave, opaque(lane));
      dvown_.issue(p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh + (int64_t)n0_ * p.v.sn, p.v.sn, p.M - n0_, smem + LDS::VOWN, wave);
    }
    const int ln = opaque(lane), hi_ = ln >> 5;
    const int nw_ = n0_ + rwave * 32, j_ = nw_ + (ln & 31);
    const char* krow = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)j_ * p.k.sn;
    const char* vrow = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh + (int64_t)j_ * p.v.sn;
#pragma unroll
    for (int kk = 0; kk < G::KS; ++kk) {
      const u32x4 z = {0u, 0u, 0u, 0u};
      rk_[kk] = z;
      rv_[kk] = z;
      if (j_ < p.M) {
        rk_[kk] = *reinterpret_cast<const u32x4*>(krow + (2 * kk + hi_) * 16);
        if constexpr (!LEAN) rv_[kk] = *reinterpret_cast<const u32x4*>(vrow + (2 * kk + hi_) * 16);
      }
    }
    rmask = (p.mask != nullptr && j_ < p.M) ? p.mask[(int64_t)b * p.M + j_] : (uint8_t)1;
    typedef RowEpilogue<T, D> EP;
    if constexpr (SEP) {      // (without SEP the epilogue loads them itself: fewer registers live across the tile loops)
      if (p.rk != nullptr && p.M - nw_ > 0)
        EP::load_inv(rinv_n, p.rk + (((int64_t)b * p.H + h) * p.M + nw_) * p.G, p.G, p.lgm, ln, p.M - nw_);
    }
  };

#ifdef FCSA_TRACE_BAR
  unsigned long long bar_wait = 0, bar_loop = 0;
#endif
  for (int pass = 0; pass < npass; ++pass) {
  FCSA_PASS_MARK(0);
  int n0, t0;
  geometry(pass, n0, t0);
  const int nw = n0 + rwave * 32;                       // first key of this wave
  const int j = nw + (lane & 31);                       // this lane's key
  const char* bias_col = nullptr;                 // &bias[slice][0][0] (wave-uniform: the element-load fallback adds row and column itself)
  const char* bias_blk = nullptr;                 // &bias[slice][0][first key of this wave]
  BiasBlock<T> bb;
  char* bscr = smem + LDS::TOTAL + wave * BiasBlock<T>::BYTES;      // private scratch of the bias transposition (BIAS launches only)
  bool bvec = false;                              // whole key tile inside M, rows 16-byte aligned: block loads (BiasBlock)
  if constexpr (BIAS) {
    const char* slice = p.bias + (int64_t)(p.bias_batch ? b : h) * p.N * (int64_t)p.M * (int64_t)sizeof(typename TR::elem);
    bias_col = slice;
    bias_blk = slice + (int64_t)nw * (int64_t)sizeof(typename TR::elem);
    bvec = n0 + BNK <= p.M && ((int64_t)p.M * TR::ES) % 16 == 0 && (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
    if (bvec && t0 < QT) bb.request(bias_blk, min(t0 * BMQ + hq + (lane & 31), p.N - 1), (int64_t)p.M * TR::ES, fa.hi);
  }
  // (the first query tile is requested ahead of the K / V fragment loads -- or, with SEP, from the previous epilogue)
  if (!have_pre) request_ahead(pass);
  auto store_tile = [&](char* buf) {
    if constexpr (DMA) {
      dma_wait();
    } else {
      sq.store(buf, tid);
      sdo.store(buf + TILE_B, tid);
    }
    if (tid < BMQ) {
      {
        // rows beyond N: lc = -inf makes P exactly 0 there
        reinterpret_cast<float*>(buf + 2 * TILE_B)[tid] = row_ok ? (p.invl_log2 ? lc_r : __builtin_amdgcn_logf(lc_r)) - p.c2 : -INFINITY;
        reinterpret_cast<float*>(buf + 2 * TILE_B + BMQ * 4)[tid] = row_ok ? -dl_r : 0.f;      // -delta
      }
    }
  };

  if constexpr (!DMA) {
    if (t0 < QT) load_tile(t0, smem);
  }

  // K, V fragments of this lane's key (B operands of S = Q K^T and dP = dO V^T), kept for the whole loop
  u32x4 kf[G::KS], vf[G::KS];
#pragma unroll
  for (int kk = 0; kk < G::KS; ++kk) {
    kf[kk] = p.q_scaled ? rk_[kk] : scale_frag<T>(rk_[kk], p.c1);     // S = Q (c1 K)^T when the Q tile is plain q^
    if constexpr (!LEAN) vf[kk] = rv_[kk];
  }
  float rinv[RowEpilogue<T, D>::NP];
  if constexpr (SEP) {
#pragma unroll
    for (int e = 0; e < RowEpilogue<T, D>::NP; ++e) rinv[e] = rinv_n[e];
  }
  const bool key_ok = j < p.M && rmask != 0;
  const uint32_t kmask = key_ok ? 0xffffffffu : 0u;      // this lane's key: valid for every query or for none
  const uint32_t ncm = causal ? 0u : 0xffffffffu;

  f32x16 dk[G::DB], dv[G::DB];
#pragma unroll
  for (int db = 0; db < G::DB; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dk[db][r] = 0.f; dv[db][r] = 0.f; }

  if (t0 < QT) store_tile(smem);
  if constexpr (RING) {      // per-query terms of the pass's second tile (its DMA was covered by the wait above)
    if (t0 + 1 < QT && tid < BMQ) {
      char* buf1 = smem + BUF_B;
      reinterpret_cast<float*>(buf1 + 2 * TILE_B)[tid] = row_ok2 ? (p.invl_log2 ? lc_r2 : __builtin_amdgcn_logf(lc_r2)) - p.c2 : -INFINITY;
      reinterpret_cast<float*>(buf1 + 2 * TILE_B + BMQ * 4)[tid] = row_ok2 ? -dl_r2 : 0.f;
    }
  }
  if constexpr (LEAN) dma_wait();      // the own-V tile (also on the no-tile path: its bytes are the epilogue's scratch)
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
    if (causal) t_m = min(QT, max(t0, (nw + 31 - diff - hq + BMQ - 1) / BMQ));      // (QSPLIT: of this wave's rows of the tile)
  }

#ifdef FCSA_TRACE_BAR
  unsigned long long bar_t = 0, loop_t = 0;
  FCSA_BAR_BEGIN(loop_t);
#endif
  DkvPipe<T, D, BMS> pipe;      // RING: lives across the tiles of a pass
  int pipe_tile = -1;           // RING: the tile whose first block's fragments are in flight
  int ring = 0;                 // RING: staging buffer of the current tile (t - t0 mod 3, kept without a division)
  auto run = [&](auto masked_tag, int t_begin, int t_end) {
    constexpr int MODE = decltype(masked_tag)::value;      // dkv_tile: 0 all valid, 1 causal select, 2 key mask by rank-1 MFMA
    constexpr bool MASKED = MODE != 0;
    for (int t = t_begin; t < t_end; ++t) {
      const int i0 = t * BMQ;
      // two buffers: the next tile lands in the other one during this tile.  RING: tile t + 1 is complete already (requested at the
      // top of tile t - 1, or in the pass prologue); tile t + 2 is requested here into the buffer tile t - 1 has left
      const int par = RING ? ring : (t - t0) & 1;
      const int par_nxt = RING ? (ring == 2 ? 0 : ring + 1) : (par ^ 1);          // buffer of tile t + 1
      const int par_ld = RING ? (ring == 0 ? 2 : ring - 1) : (par ^ 1);           // buffer this tile's requests fill
      const int t_ld = RING ? t + 2 : t + 1;                                      // ... with this tile
      const char* cur = smem + par * BUF_B;
      char* nxt = smem + par_ld * BUF_B;
      const bool more = t_ld < QT;
      FCSA_STAMP(ts, 0);
      const float* lcs = reinterpret_cast<const float*>(cur + 2 * TILE_B);
      const float* dls = lcs + BMQ;
      if constexpr (DMA) {
        // the next tile arrives by LDS-DMA, all pieces requested at the top of this tile
        if (more) {
          advance(t_ld);
          load_rows(t_ld * BMQ);
        }
        const uint32_t lds_nxt = lds0 + par_ld * BUF_B;
        FCSA_STAMP(ts, 1);
        if (more) {
          dq_.issue(stq, lds_nxt, wave);
          ddo_.issue(stdo, lds_nxt + TILE_B, wave);
        }
      } else {
        if (more) load_tile(t + 1, nxt);
        FCSA_STAMP(ts, 1);
      }
      {
      bool skip = false;
      if constexpr (MASKED) skip = causal && (i0 + hq + BMS - 1 + diff < nw);  // no valid pair for this wave
      if constexpr (PIPE) {
        if constexpr (RING) {
          if (!skip) {
            const bool has_next = t + 1 < QT;      // (else: the request reads this tile's buffer again and is never used)
            if constexpr (QSPLIT)
              dkv_tile_pipe<T, D, BMS, tile_mode<MODE>(), true, true>(cur + hoff, cur + TILE_B + hoff, lcs + hq, dls + hq, fa, kf, vf, dk, dv, kmask, ncm, j, i0 + hq, diff, ts,
                                                          pipe, pipe_tile != t, has_next ? smem + par_nxt * BUF_B : cur, wave >= 4, hoff, hq);
            else
            dkv_tile_pipe<T, D, BMQ, tile_mode<MODE>(), true>(cur, cur + TILE_B, lcs, dls, fa, kf, vf, dk, dv, kmask, ncm, j, i0, diff, ts, pipe, pipe_tile != t,
                                                  has_next ? smem + par_nxt * BUF_B : cur, NW == 8 && wave >= 4);
            pipe_tile = has_next ? t + 1 : -1;
          }
          ring = par_nxt;
        } else {
          if (!skip) dkv_tile_pipe<T, D, BMQ, tile_mode<MODE>()>(cur, cur + TILE_B, lcs, dls, fa, kf, vf, dk, dv, kmask, ncm, j, i0, diff, ts, pipe, true, nullptr,
                                                    NW == 8 && wave >= 4);
        }
      } else if constexpr (LEAN) {
        if (!skip) dkv_tile<T, D, BMQ, tile_mode<MODE>(), false, true>(cur, cur + TILE_B, lcs, dls, fa, kf, vf, dk, dv, p, kmask, ncm, j, i0, diff, nullptr, ts,
                                                            bb, bscr, nullptr, false, -1, lane, smem + LDS::VOWN, wave * 32, NW == 8 && wave >= 4);
      } else {
        const int next_i0 = more ? i0 + hq + BMQ : -1;      // (QSPLIT: of this wave's rows of the next staged tile)
        if (!skip) {
          dkv_tile<T, D, BMS, tile_mode<MODE>(), BIAS>(cur + hoff, cur + TILE_B + hoff, lcs + hq, dls + hq, fa, kf, vf, dk, dv, p, kmask, ncm, j, i0 + hq, diff, bias_col, ts, bb, bscr,
                                            bias_blk, bvec, next_i0, lane);
        } else if constexpr (BIAS) {      // the block requested for this tile is not used: request the next tile's first block instead
          if (bvec && more) bb.request(bias_blk, min(next_i0 + (lane & 31), p.N - 1), (int64_t)p.M * TR::ES, fa.hi);
        }
      }
      }
      FCSA_STAMP(ts, 8);
      if (more) store_tile(nxt);
      FCSA_STAMP(ts, 9);
      FCSA_BAR_BEGIN(bar_t);
      __syncthreads();
      FCSA_BAR_END(bar_t, bar_wait);
      FCSA_STAMP(ts, 10);
      if constexpr (!MASKED) ts.close(10);
    }
  };
  FCSA_PASS_MARK(1);
  run(std::integral_constant<int, KM ? 2 : 1>{}, t0, t_m);
  FCSA_PASS_MARK(2);
  run(std::integral_constant<int, 0>{}, t_m, QT);
  FCSA_PASS_MARK(3);
#ifdef FCSA_TRACE_BAR
  FCSA_BAR_END(loop_t, bar_loop);
#endif

  if constexpr (QSPLIT) {
    // the second-half waves hand their dK / dV partials to the first-half waves of the same keys (the staging buffers are free: every
    // tile ended with a barrier); done before anything of the epilogue or the next pass touches the LDS
    f32x4* ms = reinterpret_cast<f32x4*>(smem) + rwave * (G::DB * 8 * 64) + lane;
    if (half == 1) {
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 a = {dk[db][4 * g], dk[db][4 * g + 1], dk[db][4 * g + 2], dk[db][4 * g + 3]};
          const f32x4 c = {dv[db][4 * g], dv[db][4 * g + 1], dv[db][4 * g + 2], dv[db][4 * g + 3]};
          ms[(db * 8 + g) * 64] = a;
          ms[(db * 8 + 4 + g) * 64] = c;
        }
    }
    __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int db = 0; db < G::DB; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 a = ms[(db * 8 + g) * 64], c = ms[(db * 8 + 4 + g) * 64];
#pragma unroll
          for (int e = 0; e < 4; ++e) { dk[db][4 * g + e] += a[e]; dv[db][4 * g + e] += c[e]; }
        }
    }
    __syncthreads();
  }
  // Epilogue through the LDS (RowEpilogue): every tile ended with a barrier, so no wave still reads the staging buffers.
  // With SEP the next pass is requested between its steps and the epilogue reads nothing from global memory (see bwd_dq_kernel).
  {
    typedef RowEpilogue<T, D> EP;
    char* scr = LDS::scratch(smem, rwave);
    char* xs = LDS::xarea(smem, rwave);
    const int rows_valid = half == 0 ? p.M - nw : 0;      // (QSPLIT: the second half has handed its partials over)
    const bool fused = p.rk != nullptr;      // dk = l2norm_backward(dKh): p.k holds k^ (K/V with heads only), kf = this wave's rows of it
    const int le = opaque(lane);
    // dKh = scale * dS^T Qh; when the Q tile holds c1 * qh the factor becomes scale / c1 (= ln 2)
    const float kmul = p.q_scaled ? p.scale / p.c1 : p.scale;
    if constexpr (!SEP) {
#pragma unroll
      for (int e = 0; e < EP::NP; ++e) rinv[e] = 1.f;
      if (fused && rows_valid > 0) EP::load_inv(rinv, p.rk + (((int64_t)b * p.H + h) * p.M + nw) * p.G, p.G, p.lgm, le, rows_valid);
    }
    if (rows_valid > 0) EP::put(scr, dk, kmul, le, (LDS::X && fused) ? kf : nullptr, xs, LDS::XPITCH);
    have_pre = false;
    if constexpr (SEP) {
      if (pass + 1 < npass) {
        request_ahead(pass + 1);
        have_pre = true;
      }
    }
    if (rows_valid > 0) {
      const int64_t split_off = p.dkv_splits > 1 ? (int64_t)blockIdx.y * p.dkv_split_stride : 0;
      char* dk0 = p.dk.p + (int64_t)b * p.dk.sb + (int64_t)h * p.dk.sh + (int64_t)nw * p.dk.sn + split_off;
      char* dv0 = p.dv.p + (int64_t)b * p.dv.sb + (int64_t)h * p.dv.sh + (int64_t)nw * p.dv.sn + split_off;
      const char* x0 = fused ? p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh + (int64_t)nw * p.k.sn : nullptr;
      EP::template finish<LDS::X>(scr, xs, LDS::XPITCH, le, dk0, p.dk.sn, rows_valid, fused ? false : p.dk_f32 != 0, x0, p.k.sn, 1.f, rinv, p.lgm,
                                  p.norm_eps);
      EP::put(scr, dv, 1.f, le, nullptr, xs, LDS::XPITCH);
      EP::template finish<false>(scr, xs, LDS::XPITCH, le, dv0, p.dv.sn, rows_valid, p.dv_f32 != 0, nullptr, 0, 1.f, rinv, 0, 1.f);
    }
    if constexpr (!SEP) {
      if (pass + 1 < npass) __syncthreads();      // the scratch overlaps the staging buffers of the next pass
    }
  }
  FCSA_PASS_MARK(4);
  }   // pass
#undef FCSA_PASS_MARK
#ifdef FCSA_TRACE_BAR
  if (blockIdx.x == gridDim.x / 2 + 3 && lane == 0) { g_trace_bar_dkv[2 * wave] = bar_wait; g_trace_bar_dkv[2 * wave + 1] = bar_loop; }
#endif
#ifdef FCSA_TRACE_WG
  if (tid == 0 && blockIdx.x < 1024) { g_trace_wg_dkv[2 * blockIdx.x] = trace_t0; g_trace_wg_dkv[2 * blockIdx.x + 1] = trace_now(); }
#endif
#ifdef FCSA_TRACE
  if (blockIdx.x == gridDim.x / 2 + 3 && (tid & 63) == 0 && (wave & 2) == 0) {      // waves 0, 1, 4, 5
    unsigned long long* out = g_trace_dkv + 32 * ((wave & 1) + 2 * (wave >> 2));
    ts.dump(out, trace_now() - trace_t0);
    for (int ps = 0; ps < 2; ++ps)
      for (int k = 0; k < 4; ++k) out[14 + 4 * ps + k] = pass_marks[ps][k + 1] - pass_marks[ps][k];   // prologue | masked tiles | unmasked tiles | epilogue
  }
#endif
}

#ifdef FCSA_TRACE_WG
}  // namespace fcsa
extern "C" int fcsa_trace_read_wg_dq(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_wg_dq), sizeof(unsigned long long) * 2048);
}
extern "C" int fcsa_trace_read_pass_dq(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_pass_dq), sizeof(unsigned long long) * 2560);
}
extern "C" int fcsa_trace_read_pass_dkv(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_pass_dkv), sizeof(unsigned long long) * 2560);
}
extern "C" int fcsa_trace_read_wg_dkv(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_wg_dkv), sizeof(unsigned long long) * 2048);
}
namespace fcsa {
#endif
#ifdef FCSA_TRACE_BAR
}  // namespace fcsa
extern "C" int fcsa_trace_read_bar_dkv(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_bar_dkv), sizeof(unsigned long long) * 64);
}
extern "C" int fcsa_trace_read_bar_dq(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_bar_dq), sizeof(unsigned long long) * 64);
}
namespace fcsa {
#endif
#ifdef FCSA_TRACE
}  // namespace fcsa
extern "C" int fcsa_trace_read_dq(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_dq), sizeof(unsigned long long) * 128);
}
extern "C" int fcsa_trace_read_dkv(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_dkv), sizeof(unsigned long long) * 128);
}
namespace fcsa {
#endif

// key-split forms of the backward kernels (bwd_dq_kernel<.., KSPLIT>): 16-bit, no bias, the head dims the forward has it for
template <typename T, int D, bool BIAS> constexpr bool bwd_ksplit() {
  return Traits<T>::ES == 2 && !BIAS && (D == 16 || D == 32 || D == 64 || D == 96 || D == 128);
}

// ---------------------------------------------------------------------------------------------
// 8 waves per workgroup when the grid still gives every CU a workgroup (see row_tile_waves in fcsa_fwd.hip), else 4
// tail: the last-round rule (below): 1 = dK/dV (and the forward, fcsa_fwd.hip row_tile_waves), 2 = dQ
static int tile_waves(int64_t batch_heads, int len, bool causal, bool bits16 = false, int tail = 0) {
  const int MT = (len + 255) / 256, cus = cu_count();
  const int64_t w256 = batch_heads * (causal ? (MT + 1) / 2 : MT);
  // More 256-position workgroups than CUs, 16-bit (round 6, profiles/r06_form_sweep_big*.txt): the LAST round decides.  A last round that
  // fills at most ~55 % of the CUs costs the 8-wave form a whole 256-position workgroup time; as 4-wave workgroups the same tail is
  // 128-position workgroups running alone on their CUs: dK/dV -5 ... -9 % at 264 ... 384, 544 ... 640, 800, 1088 workgroups on 256 CUs
  // (D = 128 lean form against the pipelined 4-wave form: -8 ... -12 % at 264 ... 352), full or nearly full last rounds keep the 8-wave
  // form.  dQ: the 4-wave form wins whenever the last round is not full (-3 ... -25 %); with whole rounds the 8-wave form is ahead (C3,
  // one round: 6 %; (8,8,4096,64) causal, two rounds: 4 %, profiles/r06_ab_forms_tail.txt).
  if (bits16 && tail != 0 && w256 > cus) {
    const int64_t rem = w256 % cus;
    if (tail == 2) return rem != 0 ? 4 : 8;
    return (rem != 0 && rem * 20 <= (int64_t)cus * 11) ? 4 : 8;
  }
  if (w256 >= cus * 7 / 8) return 8;
  // 16-bit types (round 6, tools/form_sweep.py): once the 128-position tiles outnumber the CUs -- where the split-halves 8-wave forms no
  // longer apply -- the 256-position 8-wave workgroup wins from 132 workgroups on 256 CUs up, not only from 7/8 of the CUs: rows <= 128
  // bytes dQ -3 ... -10 %, dK/dV -10 ... -20 % (profiles/r06_form_sweep_d64_b.txt); D = 96 / 128 lean dK/dV -15 ... -20 %
  // (profiles/r06_form_sweep_d128_b.txt)
  if (bits16) {
    const int MT4 = (len + 127) / 128;
    if (batch_heads * (causal ? (MT4 + 1) / 2 : MT4) > cus) return 8;
  }
  return 4;
}

template <typename T, int D, bool BIAS, int NW, bool TWO, bool KSPLIT = false>
static hipError_t launch_dq_nw(const BwdParams& p, hipStream_t s) {
  constexpr int RWAVES = KSPLIT ? NW / 2 : NW, BM = 32 * RWAVES;
  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  // 128-key stages (one barrier per 128 keys) in the 8-wave form and, with LDS-DMA staging (no staging registers), also for the
  // one-wave-per-SIMD configurations (16-bit D >= 96: one workgroup per CU, the LDS is there)
  // 8-wave form: 256-key stages where they arrive by LDS-DMA (no staging registers), 128-key stages through registers (f32)
  constexpr int SUB = KSPLIT ? 2 : NW == 8 ? (Traits<T>::ES == 2 ? kDqSub8 : 2) : 1;
  size_t lds = DqLds<T, D, RWAVES, SUB, TWO, KSPLIT>::TOTAL;      // 2 buffers x (K stage + V stage), epilogue scratch behind or inside them
  if (KSPLIT && lds < (size_t)RWAVES * 64 * 16 * TileGeom<D, Traits<T>::ES>::DB * 4) lds = (size_t)RWAVES * 64 * 16 * TileGeom<D, Traits<T>::ES>::DB * 4;
  const dim3 grid((unsigned)((int64_t)p.B * p.H * PT), (unsigned)(p.dq_splits > 1 ? p.dq_splits : 1));
  // (two instantiations, see launch_fwd_nw.  The two-wave form of 256-byte rows sits at its 256 registers: its non-causal
  //  instantiation came out with spill reloads inside the tile loops -- +5.6 % time -- so those launches keep the general kernel)
  constexpr bool GENERAL_ONLY = TWO && D * Traits<T>::ES >= 256;      // (its non-causal twin is not even instantiated)
  if (p.causal || GENERAL_ONLY) {
    auto kern = bwd_dq_kernel<T, D, NW, BIAS, SUB, TWO, false, KSPLIT>;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p);
  } else if constexpr (!GENERAL_ONLY) {
    auto kern = bwd_dq_kernel<T, D, NW, BIAS, SUB, TWO, true, KSPLIT>;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p);
  }
  return hipGetLastError();
}

template <typename T, int D, bool BIAS>
static hipError_t launch_dq_b(const BwdParams& p, hipStream_t s) {
  constexpr bool NARROW = D * Traits<T>::ES <= kDq2WBytes;      // rows <= 128 bytes: two waves per SIMD whatever the grid
  if (p.dq_splits > 1) return launch_dq_nw<T, D, BIAS, 4, NARROW>(p, s);       // split-key path: 128-row tiles x key ranges (the key-split form measured level there)
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only (tools/form_sweep.py): FCSA_DQ_FORM = 1 row tiles of 8 waves, 2 key-split 8 waves, 3 four waves
  if constexpr (NARROW && bwd_ksplit<T, D, BIAS>()) {
    const int f = fcsa_dev::env_int("FCSA_DQ_FORM");
    if (f == 1) return launch_dq_nw<T, D, BIAS, 8, true>(p, s);
    if (f == 2) return launch_dq_nw<T, D, BIAS, 8, true, true>(p, s);
    if (f == 3) return launch_dq_nw<T, D, BIAS, 4, NARROW>(p, s);
  } else if constexpr (!NARROW && dq_can_two_waves<T, D>() && !BIAS && bwd_ksplit<T, D, BIAS>()) {      // 16-bit D = 96 / 128: 1 = four waves, two-wave tile, 2 = key-split 8 waves (causal), 3 = four waves, one per SIMD
    const int f = fcsa_dev::env_int("FCSA_DQ_FORM");
    if (f == 1) return launch_dq_nw<T, D, BIAS, 4, true>(p, s);
    if (f == 2 && p.causal) return launch_dq_nw<T, D, BIAS, 8, true, true>(p, s);
    if (f == 3) return launch_dq_nw<T, D, BIAS, 4, NARROW>(p, s);
  }
#endif
  if constexpr (NARROW) {
    if (tile_waves((int64_t)p.B * p.H, p.N, p.causal, Traits<T>::ES == 2, 2) == 8) return launch_dq_nw<T, D, BIAS, 8, true>(p, s);
    if constexpr (bwd_ksplit<T, D, false>()) {      // at most one 128-row workgroup per CU: its wave halves split the keys
      const int MT4 = (p.N + 127) / 128;
      if ((int64_t)p.B * p.H * (p.causal ? (MT4 + 1) / 2 : MT4) <= cu_count()) return launch_dq_nw<T, D, BIAS, 8, true, true>(p, s);
    }
  } else if constexpr (dq_can_two_waves<T, D>() && !BIAS) {
    // two waves per SIMD need two 128-row workgroups on every CU; smaller grids keep the one-wave (pipelined) form
    // (bias launches keep the one-wave form too: their two-wave instantiation spills 17 registers and was never measured ahead)
    const int MT4 = (p.N + 127) / 128;
    // (round 6: from MORE 128-row workgroups than CUs on -- rounds 3 - 5 asked for 7/4 of the CUs; at 264 ... 416 workgroups on 256 CUs the two-wave
    //  tile is 19 - 28 % faster than the one-wave and key-split forms: profiles/r06_form_sweep_d128_b.txt)
    if ((int64_t)p.B * p.H * (p.causal ? (MT4 + 1) / 2 : MT4) > cu_count()) return launch_dq_nw<T, D, BIAS, 4, true>(p, s);
    // fewer: the same tile, 8 waves on 128 rows.  Causal launches only: 256-byte rows have no non-causal instantiation of the two-wave tile
    // (GENERAL_ONLY in launch_dq_nw), and the general one measured +6 % there against the one-wave pipelined form
    if constexpr (bwd_ksplit<T, D, BIAS>()) {
      if (p.causal) return launch_dq_nw<T, D, BIAS, 8, true, true>(p, s);
    }
  }
  return launch_dq_nw<T, D, BIAS, 4, NARROW>(p, s);
}

template <typename T, int D, bool BIAS, int NW, bool LEAN = false, bool QSPLIT = false>
static hipError_t launch_dkv_nw(const BwdParams& p, hipStream_t s) {
  constexpr int BNK = 32 * (QSPLIT ? NW / 2 : NW);
  // staged query tile: 32 rows for wide feature rows (16-bit D >= 96, f32 D >= 64: VGPR budget of the staging registers),
  // else 64; 128 in the 8-wave form (one workgroup per CU: the LDS is there, and half the barriers per key tile: -4.5%)
  // (the pipelined LDS-DMA form has no staging registers: wide rows can take deeper tiles too -> fragment prefetch across
  //  blocks, fewer barriers)
  constexpr bool DMA_FORM = Traits<T>::ES == 2 && !BIAS;
  // (lean form: 64-row query tiles; 32 at 256-byte rows, where two blocks' fragments do not fit the 256 registers)
  // (64-row tiles at D = 128 measured +8 ... +15 % time: two blocks' fragments do not fit, the reloads sit in the tile loop)
  constexpr int LEAN_BMQ = D * Traits<T>::ES < 256 ? 64 : 32;
  constexpr int BMQ = LEAN ? LEAN_BMQ : (D * Traits<T>::ES >= 192) ? (DMA_FORM ? kDkvBmqWide : 32) : (NW == 8 ? kDkvBmq8 : 64);
  const int KT = (p.M + BNK - 1) / BNK;
  const int PT = p.causal ? (KT + 1) / 2 : KT;
  // ring form: the pipelined LDS-DMA tile (16 bit, no bias, not lean) where three buffers fit the workgroup's LDS share
  constexpr bool RING = kDkvRing && DMA_FORM && !LEAN && (BMQ * TileGeom<D, Traits<T>::ES>::ROWB) % 1024 == 0 &&
                        DkvLds<T, D, NW, BMQ, BIAS, LEAN, 3>::TOTAL <= ((NW == 8 || D * Traits<T>::ES > kDkv2WBytes) ? 160 : 80) * 1024;
  // NBUF x [Q tile | dO tile | lc | -delta], epilogue scratch behind or inside them; bias launches: + the waves' transposition scratch
  static_assert(!QSPLIT || RING || BIAS, "query-split form: ring tile, or the generic tile with a bias");
  size_t lds = DkvLds<T, D, NW, BMQ, BIAS, LEAN, RING ? 3 : 2>::TOTAL + (BIAS ? (size_t)NW * BiasBlock<T>::BYTES : 0);
  if (QSPLIT && lds < (size_t)(NW / 2) * 64 * 16 * TileGeom<D, Traits<T>::ES>::DB * 8) lds = (size_t)(NW / 2) * 64 * 16 * TileGeom<D, Traits<T>::ES>::DB * 8;
  const dim3 grid((unsigned)(p.B * p.H * PT), (unsigned)(p.dkv_splits > 1 ? p.dkv_splits : 1));
  if (p.causal) {        // (two instantiations, see launch_fwd_nw)
    auto kern = bwd_dkv_kernel<T, D, NW, BMQ, BIAS, LEAN, false, RING, QSPLIT>;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p);
  } else {
    auto kern = bwd_dkv_kernel<T, D, NW, BMQ, BIAS, LEAN, true, RING, QSPLIT>;
    static std::atomic<uint64_t> lds_ok{0};
    if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(NW * 64), lds, s, p);
  }
  return hipGetLastError();
}

template <typename T, int D, bool BIAS>
static hipError_t launch_dkv_b(const BwdParams& p, hipStream_t s) {
  if (p.dkv_splits > 1) return launch_dkv_nw<T, D, BIAS, 4>(p, s);       // split-query path: 128-key tiles x query ranges
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only: FCSA_DKV_FORM = 1 key tiles of 8 waves, 2 query-split 8 waves, 3 four waves
  if constexpr (D * Traits<T>::ES <= kDkv2WBytes && bwd_ksplit<T, D, BIAS>() && (D == 64 || D == 32 || D == 16)) {
    const int f = fcsa_dev::env_int("FCSA_DKV_FORM");
    if (f == 1) return launch_dkv_nw<T, D, BIAS, 8>(p, s);
    if (f == 2) return launch_dkv_nw<T, D, BIAS, 8, false, true>(p, s);
    if (f == 3) return launch_dkv_nw<T, D, BIAS, 4>(p, s);
  } else if constexpr (Traits<T>::ES == 2 && !BIAS && D * Traits<T>::ES > kDkv2WBytes && D * Traits<T>::ES <= 256) {      // 16-bit D = 96 / 128: 1 = lean 8 waves, 3 = four waves (pipelined, one per SIMD)
    const int f = fcsa_dev::env_int("FCSA_DKV_FORM");
    if (f == 1) return launch_dkv_nw<T, D, BIAS, 8, true>(p, s);
    if (f == 3) return launch_dkv_nw<T, D, BIAS, 4>(p, s);
  }
#endif
  if constexpr (D * Traits<T>::ES <= kDkv2WBytes) {
    if (tile_waves((int64_t)p.B * p.H, p.M, p.causal, Traits<T>::ES == 2, 1) == 8) return launch_dkv_nw<T, D, BIAS, 8>(p, s);
    if constexpr (bwd_ksplit<T, D, false>() && (D == 64 || D == 32 || D == 16)) {      // at most one 128-key workgroup per CU: its wave halves split the queries
      const int KT4 = (p.M + 127) / 128;
      // (from 512 queries: below, the four or fewer 128-row tiles of a pass do not pay for the hand-over -- 23.5 vs 24.7 us at N = 333 / 777)
      if (p.N >= 512 && (int64_t)p.B * p.H * (p.causal ? (KT4 + 1) / 2 : KT4) <= cu_count()) return launch_dkv_nw<T, D, BIAS, 8, false, true>(p, s);
    }
  } else if constexpr (Traits<T>::ES == 2 && !BIAS && D * Traits<T>::ES <= 256) {
    // lean form (two waves per SIMD, V fragments from the LDS) where an 8-wave workgroup per CU still covers the chip; smaller grids
    // keep the one-wave pipelined form.  (Two 4-wave workgroups per CU would do as well, but a grid with >= 448 of those always has
    // >= 224 of the 8-wave ones.)
    if (tile_waves((int64_t)p.B * p.H, p.M, p.causal, true, 1) == 8) return launch_dkv_nw<T, D, BIAS, 8, true>(p, s);
  }
  return launch_dkv_nw<T, D, BIAS, 4>(p, s);
}

template <typename T, int D> static hipError_t launch_dq_t(const BwdParams& p, hipStream_t s) {
  return p.bias != nullptr ? launch_dq_b<T, D, true>(p, s) : launch_dq_b<T, D, false>(p, s);
}
template <typename T, int D> static hipError_t launch_dkv_t(const BwdParams& p, hipStream_t s) {
  return p.bias != nullptr ? launch_dkv_b<T, D, true>(p, s) : launch_dkv_b<T, D, false>(p, s);
}

#ifdef FCSA_DEV_ONLY      // development builds: one instantiation (bf16, D = 64) for quick compiles / ISA inspection
#ifndef FCSA_DEV_D
#define FCSA_DEV_D 64
#endif
#define FCSA_DISPATCH_D(FN, T) if (D == FCSA_DEV_D) return FN<BF16, FCSA_DEV_D>(p, s); return hipErrorInvalidValue;
#else
#define FCSA_DISPATCH_D(FN, T)                      \
  switch (D) {                                      \
    case 16:  return FN<T, 16>(p, s);               \
    case 32:  return FN<T, 32>(p, s);               \
    case 64:  return FN<T, 64>(p, s);               \
    case 96:  return FN<T, 96>(p, s);               \
    case 128: return FN<T, 128>(p, s);              \
    default:  return hipErrorInvalidValue;          \
  }
#endif

hipError_t launch_backward_dbias(int dtype, int D, const BwdParams& p, hipStream_t s) {
  if (p.B * p.H == 0 || p.N == 0 || p.M == 0 || p.d_bias == nullptr || p.bias == nullptr) return hipSuccess;
  if (dtype == 2) { FCSA_DISPATCH_D(launch_dbias_t, BF16) }
  if (dtype == 1) { FCSA_DISPATCH_D(launch_dbias_t, F16) }
  if (dtype == 0) { FCSA_DISPATCH_D(launch_dbias_t, F32) }
  return hipErrorInvalidValue;
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
