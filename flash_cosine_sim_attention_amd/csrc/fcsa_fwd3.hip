// fcsa_fwd3.hip -- forward kernel for 16-bit D = 128 with the WHOLE 512-entry register file per wave: 4 waves per workgroup, one per
// SIMD, 64 query rows per wave, operands placed in the accumulator half of the file by register-class constraints.
//
// Replaces forward_kernel (reference cu:1072-1247) at the head dim real models use, on grids that cover the chip.  Same mathematics
// and the same fragment conventions as the other forward forms (fcsa_fwd.hip header, fcsa_common.cuh): S^T = K Q^T on
// v_mfma_f32_32x32x16 with the exponent shift as the accumulator's initial value, P~ = exp2(S) packed in place into the B operand of
// O^T = V^T P~^T, V^T through ds_read_b64_tr_b16, K / V tiles by LDS-DMA into XOR-swizzled row-major tiles, no running max.
//
// Why another form.  At D = 128 a 64-row wave needs O (128 registers), c1 * q^ (64), a block's K fragments (32), two key blocks of
// logits (64), their packed P~ (32), a block's V^T fragments (32) plus addresses: ~400 live registers.  hipcc allocates MFMA operands
// from ONE register class per function (-amdgpu-mfma-vgpr-form: the 256 architectural VGPRs; without it every accumulator sits in
// AGPRs and each logit is copied out before its exp): the slot-scheduled wide kernel (fwd2_kernel) spilled 76 ... 250 registers at
// this width in round 4, and the forms that fit (32 rows per wave, two waves per SIMD: `LEAN`) move 1 KiB through the LDS per MFMA
// and stop at ~0.37 of the matrix peak.  Here every instruction of the tile loop is a one-instruction `asm volatile` statement whose
// operands carry their register class ("a": accumulator half -- O, c1 * q^, the K fragments, which only MFMAs and LDS loads touch;
// "v": logits, P~, V^T fragments, everything the VALU touches), so the ALLOCATION is still hipcc's (no literal register names: nothing
// to audit but `.vgpr_spill_count 0`) while the ORDER is exactly the program order below -- volatile statements are not reordered --
// and every wait is counted by hand (hipcc does not see loads inside asm statements: guide section 5.7).
//
// Rotating pipeline per 64-key tile t (two 32-key blocks kb = 0, 1; two 32-row blocks qb = 0, 1 sharing every K / V fragment), four
// phases of 16 MFMAs; the exp / row-sum / pack work of a block (80 VALU instructions) is spread over the 32 MFMAs that follow its S
// chain, the LDS reads of the next phase's fragments and the LDS-DMA of later tiles ride in the same gaps:
//
//     P1  S0(t)    | softmax of (t-1, kb 1, rows 32..63) | V^T reads of (t-1, kb 1)
//     P2  PV1(t-1) | softmax of (t, kb 0, rows 0..31)    | K reads of (t, kb 1)     | LDS-DMA K(t+R-1) -> slot of K(t-1)
//     --  s_waitcnt lgkmcnt(0) vmcnt(8 (R - 3) + 4); s_barrier  (every wave is done with K(t) and V(t-1); K(t+1) and V(t) are visible)
//     P3  S1(t)    | softmax of (t, kb 0, rows 32..63)   | V^T reads of (t, kb 0)
//     P4  PV0(t)   | softmax of (t, kb 1, rows 0..31)    | K reads of (t+1, kb 0)   | LDS-DMA V(t+R-2) -> slot of V(t-2)
//
// i.e. <= 5 fillers per MFMA gap (one exp, one add, a cvt_pk every other gap, <= 2 LDS reads or an address step / DMA piece), one barrier
// per tile, rings of R tiles for K and for V.  Row sums are plain f32 adds of the un-rounded P~ (two partial sums per row block: v_dot2c
// on the packed values costs more than two adds beside MFMAs, MI355X guide "price of one filler"; measured +5 ... 7 %).  Tiles that
// need masking for a wave (the causal diagonal, a ragged last tile) run a second instantiation of the same body with a compare + select
// in front of each exp.
//
// Measured (MI355X, bf16, forward call = k-l2norm launch + this kernel; same-process A/B against the lean 32-row form, profiles/r05_fwd3_*):
// (4,8,4096,128) causal 149 -> 138 us, non-causal 256 -> 231, (16,8,2048,128) causal 191 -> 181, (2,8,8192,128) causal 258 -> 235.
// Phase trace: 2480 ticks per 64-MFMA tile (matrix floor 2048): S phases 615 - 635, PV phases 578, barrier 60 - 100; per pass
// ~12 k ticks of prologue (rows + fused l2norm under the first tiles' flight) and ~7 k of drain + epilogue.  Version history of the
// schedule with its traces: profiles/NOTES.md (round 5).
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "fcsa_common.cuh"
#include "fcsa_kernels.h"

namespace fcsa {

template <typename F, int... I> FCSA_DEV void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F> FCSA_DEV void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// One-instruction statements.  "a" = accumulator-file register (AGPR), "v" = architectural VGPR.
template <typename T> struct Ins;
#define FCSA_INS(TYPE, MF, CVT, DOT)                                                                                                  \
  template <> struct Ins<TYPE> {                                                                                                   \
    /* S chain, first k-step: D (fresh logits, VGPR) = A (K fragment, AGPR) x B (q^ fragment, AGPR) + C (the -shift tuple, VGPR) */ \
    static FCSA_DEV void s_first(f32x16& d, const u32x4& a, const u32x4& b, const f32x16& c) {                                      \
      asm volatile(MF " %0, %1, %2, %3" : "=&v"(d) : "a"(a), "a"(b), "v"(c));                                                        \
    }                                                                                                                              \
    static FCSA_DEV void s_next(f32x16& d, const u32x4& a, const u32x4& b) { asm volatile(MF " %0, %1, %2, %0" : "+v"(d) : "a"(a), "a"(b)); } \
    /* O^T += V^T fragment (VGPR) x packed P~ (VGPR), O in the accumulator file */                                                  \
    static FCSA_DEV void pv(f32x16& d, const u32x4& a, const u32x4& b) { asm volatile(MF " %0, %1, %2, %0" : "+a"(d) : "v"(a), "v"(b)); }    \
    static FCSA_DEV void cvt_pk(uint32_t& d, float a, float b) { asm volatile(CVT " %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }              \
    /* acc += lo(u) + hi(u): exact f32 sum of the two ROUNDED values (packed ones in `one2`) */                                        \
    static FCSA_DEV void add_pair(float& acc, uint32_t u, uint32_t one2) { asm volatile(DOT " %0, %1, %2" : "+v"(acc) : "v"(u), "v"(one2)); }  \
  }
FCSA_INS(BF16, "v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32", "v_dot2c_f32_bf16_e32");
FCSA_INS(F16, "v_mfma_f32_32x32x16_f16", "v_cvt_pk_f16_f32", "v_dot2c_f32_f16_e32");
#undef FCSA_INS

template <int OFF> FCSA_DEV void lds_read_k(u32x4& d, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "i"(OFF)); }
// one 4-row half of a transposed fragment (registers 2 HALF, 2 HALF + 1 of the MFMA operand)
template <int OFF, int HALF> FCSA_DEV void lds_read_vt_half(u32x2 (&d)[2], uint32_t a) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d[HALF]) : "v"(a), "i"(OFF));
}
FCSA_DEV u32x4 vt_frag(const u32x2 (&h)[2]) { u32x4 d; d[0] = h[0][0]; d[1] = h[0][1]; d[2] = h[1][0]; d[3] = h[1][1]; return d; }
template <int N> FCSA_DEV void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }
template <int N> FCSA_DEV void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
FCSA_DEV void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

// Registers of the pipeline (all indices compile-time: scalar-replaced, never addressed)
template <typename T> struct F3State {
  f32x16 o[2][4];        // "a"  O^T accumulators [row block][feature block]
  u32x4 q[2][8];         // "a"  c1 * q^ fragments [row block][k-step]
  u32x4 kf[8];           // "a"  K fragments of ONE 32-key block [k-step] (both row blocks use each one)
  f32x16 s[2][2];        // "v"  logits -> P~ [key block][row block]
  u32x4 pk[2][2][2];     // "v"  packed P~ [key block][row block][16-key step]
  u32x2 vfh[8][2];       // "v"  V^T fragments of one 32-key block [16-key step * 4 + feature block][4-row half]
  f32x16 cinit;          // "v"  -shift in all 16 registers: the C operand of every S chain's first k-step
  float l[2][2];         // "v"  row-sum partials [row block][even / odd register]
  float ninf;            // "v"  -inf (select operand of the masked tiles)
  uint32_t one2;         // "v"  two packed 1.0 (RSUM)
  uint32_t ka[8];        //      LDS byte address of this lane's K row-fragment chunk per k-step (slot 0, key block 0)
  uint32_t va[8];        //      LDS byte address of this lane's transposed-read rows per (feature block, half) (slot 0, key step 0)
};

// Softmax of one [32 keys x 32 rows] block half (= one row block's 16 accumulator registers), spread over the 16 MFMA gaps of a phase
// with ONE transcendental per gap (v_exp_f32 issues at quarter rate: two in one gap were 32 of its 32 matrix cycles):
//     gap G:  (select,) exp of register G  |  row-sum add of register G - 1  |  even G >= 2: pack of registers G - 2, G - 1
// so every consumer sits one MFMA behind its producer, and the tail -- the add of register 15 and the pack of registers 14, 15 -- is the
// head of the NEXT phase's gap 0 (`sm_tail`), whatever block half that phase works on.
template <bool MASKED, int R_, int KOFF> FCSA_DEV void sm_exp1(f32x16& s, int thr, float ninf) {
  if constexpr (MASKED) {      // key row KOFF + crow(r, 0) (relative to this lane's threshold, which carries j0 and 4 * hi) is visible iff <= thr
    asm volatile("v_cmp_le_i32_e32 vcc, %2, %1\n\tv_cndmask_b32_e32 %0, %3, %0, vcc" : "+v"(s[R_]) : "v"(thr), "n"(KOFF + crow(R_, 0)), "v"(ninf) : "vcc");
  }
  asm volatile("v_exp_f32_e32 %0, %0" : "+v"(s[R_]));
}
// RSUM: the row sum takes the ROUNDED pair (v_dot2c against packed ones, like the other 16-bit forward forms: O is then a true convex
// combination of V rows); else the un-rounded values (plain adds, two partial sums per row block)
template <bool RSUM, int R_> FCSA_DEV void sm_add1(const f32x16& s, float (&l)[2]) {
  if constexpr (!RSUM) asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(l[R_ & 1]) : "v"(s[R_]));
}
template <typename T, bool RSUM, int E> FCSA_DEV void sm_pack1(const f32x16& s, u32x4 (&pk)[2], float (&l)[2], uint32_t one2) {
  uint32_t u;
  Ins<T>::cvt_pk(u, s[2 * E], s[2 * E + 1]);
  if constexpr (RSUM) Ins<T>::add_pair(l[E & 1], u, one2);
  pk[E >> 2][E & 3] = u;
}
template <typename T, bool RSUM, bool MASKED, int HALF, int G, int KOFF>
FCSA_DEV void sm_gap(F3State<T>& st, f32x16 (&s)[2], u32x4 (&pk)[2][2], const int (&thr)[2]) {
  if constexpr (G >= 1) sm_add1<RSUM, G - 1>(s[HALF], st.l[HALF]);
  sm_exp1<MASKED, G, KOFF>(s[HALF], thr[HALF], st.ninf);
  if constexpr (G >= 2 && (G & 1) == 0) sm_pack1<T, RSUM, (G - 2) / 2>(s[HALF], pk[HALF], st.l[HALF], st.one2);
}
// the tail of the block half the PREVIOUS phase worked on
template <typename T, bool RSUM, int HALF>
FCSA_DEV void sm_tail(F3State<T>& st, f32x16 (&s)[2], u32x4 (&pk)[2][2]) {
  sm_add1<RSUM, 15>(s[HALF], st.l[HALF]);
  sm_pack1<T, RSUM, 7>(s[HALF], pk[HALF], st.l[HALF], st.one2);
}

template <int A, int B> constexpr int cmin() { return A < B ? A : B; }
// One LDS-DMA piece inside the tile loop: M0 <- LDS byte address of the piece (wave-uniform), 64 x 16 bytes from `rs` at per-lane
// byte offset voff + wave-uniform soff.  M0 is clobbered, not saved (guide 5.7): three instructions instead of DmaStager::issue_piece's
// six plus its per-call descriptor shuffle -- the pieces are the most expensive fillers of the loop (45 ... 90 ticks each, trace).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (the M0 clobber is deliberate: see above)
FCSA_DEV void dma_piece(uint32_t lds_dst, uint32_t voff, const u32x4& rs, uint32_t soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" :: "s"(lds_dst), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop
FCSA_DEV void addr_step(uint32_t& a, uint32_t delta) { asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(a) : "s"(delta)); }

// The ring slot of a tile is a RUN-TIME quantity carried in the 16 LDS address registers (ka: slot of K(t), stepped to K(t+1) during P3;
// va: slot of V(t-1), stepped to V(t) during P2; one v_add per gap, delta = +16 KiB or the wrap back): a body per slot made every
// pipeline register a phi over R x 2 copies of the tile and hipcc answered with ~1200 accumulator moves and 286 spilled registers.
//
// S phase of key block KB: 16 MFMAs (k-step outer, row block inner: the two chains alternate), the K fragments of this block were
// requested during the previous phase.  `filler(g)` = this gap's softmax share; V^T fragments at va + VOFF (key block, 16-key step) are
// requested two reads per gap in gaps 0..7; `extra(g)` = address steps / DMA pieces of this phase.
template <typename T, int VOFF, typename Filler, typename Extra>
FCSA_DEV void s_phase(F3State<T>& st, f32x16 (&sblk)[2], Filler&& filler, Extra&& extra) {
  static_for<16>([&](auto gc) {
    constexpr int g = decltype(gc)::value, ks = g >> 1, qb = g & 1;
    // K fragment ks has landed: behind it are 7 - ks K reads and the V^T reads this phase has issued so far (LDS reads return in order)
    if constexpr (qb == 0) wait_lgkm<cmin<15, (7 - ks) + 2 * cmin<g, 8>()>()>();
    if constexpr (ks == 0) Ins<T>::s_first(sblk[qb], st.kf[0], st.q[qb][0], st.cinit);
    else Ins<T>::s_next(sblk[qb], st.kf[ks], st.q[qb][ks]);
    filler(gc);
    // V^T fragment f = g in the first eight gaps, both halves: (16-key step f / 4, feature block f % 4) -- consumed in that order by the PV
    // phase.  (One read per gap over all sixteen gaps measured +90 ticks per S phase: profiles/r05_fwd3_trace_v4.txt.)
    if constexpr (g < 8) {
      lds_read_vt_half<VOFF + 16 * (g >> 2) * 256, 0>(st.vfh[g], st.va[2 * (g & 3)]);
      lds_read_vt_half<VOFF + 16 * (g >> 2) * 256, 1>(st.vfh[g], st.va[2 * (g & 3) + 1]);
    }
    extra(gc);
  });
}
// PV phase of a key block: 16 MFMAs (16-key step outer, feature block, row block inner: a fragment feeds two MFMAs, an accumulator
// returns after 8).  K fragments at ka + KOFFB are requested one per gap in gaps 0..7.
template <typename T, int KOFFB, typename Filler, typename Extra>
FCSA_DEV void pv_phase(F3State<T>& st, u32x4 (&pblk)[2][2], Filler&& filler, Extra&& extra) {
  static_for<16>([&](auto gc) {
    constexpr int g = decltype(gc)::value, f = g >> 1, qb = g & 1, ks2 = f >> 2, db = f & 3;
    // V^T fragment f (reads 2f, 2f + 1 of 16) has landed: behind it are 14 - 2f V^T reads and the K reads issued so far
    if constexpr (qb == 0) wait_lgkm<cmin<15, (14 - 2 * f) + cmin<g, 8>()>()>();
    Ins<T>::pv(st.o[qb][db], vt_frag(st.vfh[f]), pblk[qb][ks2]);
    filler(gc);
    if constexpr (g < 8) lds_read_k<KOFFB>(st.kf[g], st.ka[g]);
    extra(gc);
  });
}

// One tile.  dk_next / dv_next: LDS byte distance from the slot of K(t) to that of K(t+1) (= from V(t-1) to V(t) one tile earlier).
// thr_prev / thr_cur: per-lane visibility thresholds of the tile whose block 1 is still in flight and of this tile (MASKED only).
// The LDS-DMA pieces ride in the PV phases (a piece costs ~45 ticks there, ~90 beside the 16 transposed reads of an S phase): K(t+R-1)
// into the slot of K(t-1) during P2, V(t+R-2) into the slot of V(t-2) during P4 -- both slots were released by the PREVIOUS tile's barrier.
template <typename T, int R, bool RSUM, bool MASKED, typename DmaK, typename DmaV>
FCSA_DEV void fwd3_tile(F3State<T>& st, Trace& ts, uint32_t dk_next, uint32_t dv_next, const int (&thr_prev)[2], const int (&thr_cur)[2], DmaK&& dma_k, DmaV&& dma_v) {
  // P1: S0(t) | softmax of (t-1, kb 1, rows 32..63) [tail of (t-1, kb 1, rows 0..31)] | V^T reads of (t-1, kb 1) | va: V(t-1) -> V(t)
  FCSA_STAMP(ts, 0);
  s_phase<T, 8192>(st, st.s[0],
                        [&](auto gc) {
                          constexpr int g = decltype(gc)::value;
                          if constexpr (g == 0) sm_tail<T, RSUM, 0>(st, st.s[1], st.pk[1]);
                          sm_gap<T, RSUM, MASKED, 1, g, 32>(st, st.s[1], st.pk[1], thr_prev);
                        },
                        [&](auto gc) { constexpr int g = decltype(gc)::value; if constexpr (g >= 8) addr_step(st.va[g - 8], dv_next); });
  FCSA_STAMP(ts, 1);
  // P2: PV1(t-1) | softmax of (t, kb 0, rows 0..31) [tail of (t-1, kb 1, rows 32..63)] | K reads of (t, kb 1) | DMA K(t+R-1)
  pv_phase<T, 8192>(st, st.pk[1],
                         [&](auto gc) {
                           constexpr int g = decltype(gc)::value;
                           if constexpr (g == 0) sm_tail<T, RSUM, 1>(st, st.s[1], st.pk[1]);
                           sm_gap<T, RSUM, MASKED, 0, g, 0>(st, st.s[0], st.pk[0], thr_cur);
                         },
                         [&](auto gc) { constexpr int g = decltype(gc)::value; if constexpr (g >= 8 && (g & 1) == 0) dma_k((g - 8) >> 1); });
  FCSA_STAMP(ts, 2);
  // every wave is done with K(t) and V(t-1) -- lgkmcnt(0): its reads of them have RETURNED (the K reads went out >= 8 gaps ago: free),
  // so a DMA behind the barrier cannot overtake a read; the pieces of K(t+1) and V(t) that THIS wave requested have landed (vmcnt: the
  // 8 pieces of every younger period and the 4 of this P2 may stay in flight), the barrier publishes all
  wait_lgkm<0>();
  wait_vm<8 * (R - 3) + 4>();
  wg_barrier();
  FCSA_STAMP(ts, 3);
  // P3: S1(t) | softmax of (t, kb 0, rows 32..63) [tail of (t, kb 0, rows 0..31)] | V^T reads of (t, kb 0) | ka: K(t) -> K(t+1)
  s_phase<T, 0>(st, st.s[1],
                     [&](auto gc) {
                       constexpr int g = decltype(gc)::value;
                       if constexpr (g == 0) sm_tail<T, RSUM, 0>(st, st.s[0], st.pk[0]);
                       sm_gap<T, RSUM, MASKED, 1, g, 0>(st, st.s[0], st.pk[0], thr_cur);
                     },
                     [&](auto gc) { constexpr int g = decltype(gc)::value; if constexpr (g >= 8) addr_step(st.ka[g - 8], dk_next); });
  FCSA_STAMP(ts, 4);
  // P4: PV0(t) | softmax of (t, kb 1, rows 0..31) [tail of (t, kb 0, rows 32..63)] | K reads of (t+1, kb 0) | DMA V(t+R-2)
  pv_phase<T, 0>(st, st.pk[0],
                      [&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if constexpr (g == 0) sm_tail<T, RSUM, 1>(st, st.s[0], st.pk[0]);
                        sm_gap<T, RSUM, MASKED, 0, g, 32>(st, st.s[1], st.pk[1], thr_cur);
                      },
                      [&](auto gc) { constexpr int g = decltype(gc)::value; if constexpr (g >= 8 && (g & 1) == 0) dma_v((g - 8) >> 1); });
  FCSA_STAMP(ts, 5);
}

// After the last tile: the rest of its block 1 (tail of rows 0..31, rows 32..63) and PV1 (not overlapped: once per pass).  va points at
// the last tile's slot.
template <typename T, bool RSUM>
FCSA_DEV void fwd3_drain(F3State<T>& st, const int (&thr_last)[2]) {
  static_for<8>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    lds_read_vt_half<8192 + 16 * (g >> 2) * 256, 0>(st.vfh[g], st.va[2 * (g & 3)]);
    lds_read_vt_half<8192 + 16 * (g >> 2) * 256, 1>(st.vfh[g], st.va[2 * (g & 3) + 1]);
  });
  sm_tail<T, RSUM, 0>(st, st.s[1], st.pk[1]);
  static_for<16>([&](auto gc) {
    sm_gap<T, RSUM, true, 1, decltype(gc)::value, 32>(st, st.s[1], st.pk[1], thr_last);
    asm volatile("s_nop 1");      // (no MFMA between a v_exp and its consumer here: the transcendental's forwarding gap)
  });
  asm volatile("s_nop 1");
  sm_tail<T, RSUM, 1>(st, st.s[1], st.pk[1]);
  wait_lgkm<0>();
  asm volatile("s_nop 4");      // (the last v_cvt_pk -> MFMA operand)
  static_for<16>([&](auto gc) {
    constexpr int g = decltype(gc)::value, f = g >> 1, qb = g & 1;
    Ins<T>::pv(st.o[qb][f & 3], vt_frag(st.vfh[f]), st.pk[1][qb][f >> 2]);
  });
  asm volatile("s_nop 15\n\ts_nop 15");      // the last MFMA's result, before hipcc's own accumulator reads (it does not see the MFMA)
}

// ---- query rows ------------------------------------------------------------------------------------------------------------------
// The 64 query rows of a wave as B operands: lane (x, hi) holds the 16-byte chunks 2 kk + hi of rows i0 + x and i0 + 32 + x.  The rows
// are requested FIRST, by buffer loads inside asm statements (rows >= N read as zero through the descriptor's range check), the ring's
// first tiles behind them, and ONE counted wait -- vmcnt(number of DMA pieces) -- hands the rows over while the tiles are still in
// flight: the l2norm arithmetic below then runs under their latency.  (With compiler-visible loads hipcc's own wait for them counts only
// ITS loads, sees none outstanding behind them, and drains the younger DMA pieces as well: the phase trace showed 17.4 k ticks between
// "tiles requested" and "c1 * q^ ready", per pass, of which the arithmetic is a quarter.)
template <int OFF, bool FIRST> FCSA_DEV void q_load(u32x4& d, uint32_t voff, const u32x4& rs) {
  // (FIRST: the descriptor's SGPRs may be fresh from v_readfirstlane -- five wait states before a buffer instruction reads them, inside
  //  the statement: hipcc does not pad what it cannot see)
  if constexpr (FIRST) asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(d) : "v"(voff), "s"(rs), "i"(OFF));
  else asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(d) : "v"(voff), "s"(rs), "i"(OFF));
}
template <int N> FCSA_DEV void q_wait(u32x4 (&q)[2][8]) {      // every destination is named, so no use of it moves above the wait
  asm volatile("s_waitcnt vmcnt(%8)" : "+v"(q[0][0]), "+v"(q[0][1]), "+v"(q[0][2]), "+v"(q[0][3]), "+v"(q[0][4]), "+v"(q[0][5]), "+v"(q[0][6]), "+v"(q[0][7]) : "n"(N));
  asm volatile("" : "+v"(q[1][0]), "+v"(q[1][1]), "+v"(q[1][2]), "+v"(q[1][3]), "+v"(q[1][4]), "+v"(q[1][5]), "+v"(q[1][6]), "+v"(q[1][7]));
}
FCSA_DEV float xhalf_sum_swap(float x) {      // x(lane) + x(lane ^ 32) on the VALU (v_permlane32_swap): no LDS round trip in the prologue
  const uint32_t bits = __builtin_bit_cast(uint32_t, x);
  const auto sw = __builtin_amdgcn_permlane32_swap(bits, bits, false, false);
  return __builtin_bit_cast(float, (uint32_t)sw[0]) + __builtin_bit_cast(float, (uint32_t)sw[1]);
}
// raw rows -> c1 * q^ (grouped l2norm fused, the saved state of the backward published), or the plain c1 scaling: the arithmetic of
// finish_q_frags (fcsa_common.cuh) with the same summation order -- group sums over increasing k-steps -- but one reciprocal norm per
// GROUP instead of per k-step (a wave-uniform case split on the group size) and the lane-pair sum on the VALU.
template <typename T>
FCSA_DEV void fwd3_finish_q(const FwdParams& p, int b, int h, int i, int hi_, u32x4 (&qf)[8]) {
  constexpr int D = 128;
  if (p.q_raw) {
    float pr[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float ss = dot_frag<T>(qf[kk], qf[kk]);
      pr[kk] = p.lgm >= 1 ? xhalf_sum_swap(ss) : ss;      // groups of >= 16 features contain both chunks of a k-step
    }
    const int sh = p.lgm >= 1 ? p.lgm - 1 : 0;            // k-steps per group = 1 << sh
    auto rn = [&](float tot) { return 1.f / fmaxf(sqrtf(tot), p.norm_eps); };
    float r[8];
    if (sh >= 3) {
      const float v = rn(((((((pr[0] + pr[1]) + pr[2]) + pr[3]) + pr[4]) + pr[5]) + pr[6]) + pr[7]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) r[kk] = v;
    } else if (sh == 2) {
      const float v0 = rn(((pr[0] + pr[1]) + pr[2]) + pr[3]), v1 = rn(((pr[4] + pr[5]) + pr[6]) + pr[7]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) r[kk] = kk < 4 ? v0 : v1;
    } else if (sh == 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) { const float v = rn(pr[2 * g] + pr[2 * g + 1]); r[2 * g] = v; r[2 * g + 1] = v; }
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) r[kk] = rn(pr[kk]);
    }
    const int64_t row = ((int64_t)b * p.H + h) * p.N + i;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      qf[kk] = scale_frag<T>(qf[kk], r[kk] * p.c1);
      if (i < p.N) {
        const int c = 2 * kk + hi_;
        if (p.qn_out != nullptr) *reinterpret_cast<u32x4*>(p.qn_out + (row * D + 8 * c) * 2) = qf[kk];      // (inference: nothing is saved)
        if (p.rq_out != nullptr && (c & ((1 << p.lgm) - 1)) == 0) p.rq_out[row * p.G + (c >> p.lgm)] = r[kk];
      }
    }
    return;
  }
  if (!p.q_scaled) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) qf[kk] = scale_frag<T>(qf[kk], p.c1);
  }
}

#ifdef FCSA_TRACE
__device__ unsigned long long g_trace_fwd3[128];
#endif

template <typename T, int R, bool RSUM>
__global__ void __launch_bounds__(256, 1) fwd3_kernel(const FwdParams p) {
  constexpr int D = 128;
  typedef TileGeom<D, 2> G;
  typedef DmaStager<T, D, 64, 4> DS;
  constexpr int BN = 64, RW = 64, NW = 4, BM = RW * NW;
  constexpr int TILE_B = BN * G::ROWB;             // 16 KiB
  static_assert(TILE_B == 16384 && DS::PER == 4 && DS::UNIFORM && R >= 3, "fwd3 geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];      // [R] K tiles | [R] V tiles; the epilogue scratch reuses the bytes

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  FragAddr<T, D> fa;
  fa.init(lane);
  const uint32_t lds0 = DS::lds_addr(smem);

  const int MT = (p.N + BM - 1) / BM;
  const int PT = p.causal ? (MT + 1) / 2 : MT;                 // causal: pairs of row tiles (MT-1-pt, pt), constant work
  int bh, pt;
  block_to_work(blockIdx.x, p.B * p.H, PT, bh, pt);
  const int b = bh / p.H, h = bh % p.H;
  const int npass = (p.causal && (MT - 1 - pt) != pt) ? 2 : 1;
  const int diff = p.M - p.N;
  const char* kbase = p.k.p + (int64_t)b * p.k.sb + (int64_t)h * p.k.sh;
  const char* vbase = p.v.p + (int64_t)b * p.v.sb + (int64_t)h * p.v.sh;
  DS dk_, dv_;
  dk_.init(p.k.sn, wave, lane);
  dv_.init(p.v.sn, wave, lane);
  const uint32_t k_step = (uint32_t)(BN * p.k.sn), v_step = (uint32_t)(BN * p.v.sn);      // (launcher: M * pitch < 2 GiB)

  Trace ts;
  ts.reset();
#ifdef FCSA_TRACE
  const unsigned long long trace_t0 = trace_now();
  unsigned long long pm[2][8];
  for (int a_ = 0; a_ < 2; ++a_) for (int b_ = 0; b_ < 8; ++b_) pm[a_][b_] = 0;
#define FCSA_PASS_MARK(k) pm[pass][k] = trace_now()
#else
#define FCSA_PASS_MARK(k) ((void)0)
#endif
  F3State<T> st;
  {
    // (through an opaque VGPR: as a uniform value hipcc keeps the 16-register tuple in SGPRs and re-materialises it with 8 v_mov_b64 per tile)
    float c0 = -p.c2;
    asm volatile("" : "+v"(c0));
#pragma unroll
    for (int e = 0; e < 16; ++e) st.cinit[e] = c0;
    float ni = -INFINITY;
    asm volatile("" : "+v"(ni));
    st.ninf = ni;
    uint32_t o2 = Traits<T>::kOne2;
    asm volatile("" : "+v"(o2));
    st.one2 = o2;
  }
  for (int pass = 0; pass < npass; ++pass) {
    const int mt = p.causal ? (pass == 0 ? MT - 1 - pt : pt) : pt;      // heavy tile first
    const int m0 = mt * BM;
    const int mw = m0 + wave * RW;                  // first query row of this wave
    const int i0 = mw + (lane & 31);                // this lane's row in block 0; block 1 is i0 + 32
    int last_key = p.M - 1;
    if (p.causal) last_key = min(last_key, m0 + BM - 1 + diff);
    const int nt = last_key < 0 ? 0 : last_key / BN + 1;

    FCSA_PASS_MARK(0);
    // ---- prologue: the query rows, then the ring's first tiles by LDS-DMA (K(0..R-2); V(-1) = zeros in slot R-1; V(0..R-3)) ----
    u32x4 qf[2][8];
    {
      const char* qbase = p.q.p + (int64_t)b * p.q.sb + (int64_t)h * p.q.sh;
      int64_t qbytes = (int64_t)(p.N - 1) * p.q.sn + D * 2;
      if (qbytes > 0x7fffffff) qbytes = 0x7fffffff;
      const uint64_t qa = reinterpret_cast<uint64_t>(qbase);
      u32x4 qrs;
      qrs[0] = __builtin_amdgcn_readfirstlane((uint32_t)qa);
      qrs[1] = __builtin_amdgcn_readfirstlane((uint32_t)(qa >> 32) & 0xffffu);
      qrs[2] = __builtin_amdgcn_readfirstlane((uint32_t)qbytes);
      qrs[3] = 0x00020000u;
      const int hio = opaque(fa.hi);
      const uint32_t qv0 = (uint32_t)i0 * (uint32_t)p.q.sn + (uint32_t)hio * 16u, qv1 = qv0 + 32u * (uint32_t)p.q.sn;      // (launcher: N * pitch < 2 GiB)
      static_for<8>([&](auto kc) { constexpr int kk = decltype(kc)::value; q_load<32 * kk, kk == 0>(qf[0][kk], qv0, qrs); });
      static_for<8>([&](auto kc) { constexpr int kk = decltype(kc)::value; q_load<32 * kk, false>(qf[1][kk], qv1, qrs); });
    }
    // The 4 (2R - 2) DMA pieces of the ring's first tiles are expensive to ISSUE back to back at a cold start (the trace: 6.7 k ticks for
    // the 24 of R = 4, the vector-memory queue is full), so only the tiles the first period needs -- V(-1) = zeros, K(0), V(0) -- go out
    // ahead of the counted wait; the others are issued between the two rows' l2norm arithmetic, which their flight then covers.
    typename DS::Stream stk = dk_.open(kbase, p.k.sn, p.M), stv = dv_.open(vbase, p.v.sn, p.M);
    {
      const typename DS::Stream stz = dv_.open(vbase, p.v.sn, 0);      // zero records: the DMA writes zeros
      dv_.issue(stz, lds0 + (2 * R - 1) * TILE_B, wave);
      dk_.issue(stk, lds0, wave);
      stk.off += k_step;
      dv_.issue(stv, lds0 + R * TILE_B, wave);
      stv.off += v_step;
    }
    FCSA_PASS_MARK(1);
    u32x4 krs, vrs;      // the streams' descriptors, provably in SGPRs for the loop's DMA statements
#pragma unroll
    for (int e = 0; e < 4; ++e) { krs[e] = __builtin_amdgcn_readfirstlane(stk.rs[e]); vrs[e] = __builtin_amdgcn_readfirstlane(stv.rs[e]); }
    q_wait<12>(qf);      // the rows have landed; the 12 DMA pieces behind them stay in flight
    {
      const int hio = opaque(fa.hi);
      fwd3_finish_q<T>(p, b, h, i0, hio, qf[0]);
#pragma unroll
      for (int s = 1; s < R - 1; ++s) {      // K(1..R-2), V(1..R-3)
        dk_.issue(stk, lds0 + s * TILE_B, wave);
        stk.off += k_step;
        if (s < R - 2) {
          dv_.issue(stv, lds0 + (R + s) * TILE_B, wave);
          stv.off += v_step;
        }
      }
      fwd3_finish_q<T>(p, b, h, i0 + 32, hio, qf[1]);
    }
    FCSA_PASS_MARK(2);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) asm volatile("" : "=a"(st.q[r][kk]) : "0"(qf[r][kk]));      // into the accumulator half of the file
    {
      f32x16 z;
#pragma unroll
      for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int db = 0; db < 4; ++db) asm volatile("" : "=a"(st.o[r][db]) : "0"(z));
    }
    // pipeline state "nothing in flight": the logits of tile -1 exponentiate to 0, its V tile is the zero slot
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
      for (int e = 0; e < 16; ++e) st.s[1][r][e] = r == 0 ? 0.f : -1e30f;      // rows 0..31: "already exponentiated" (only its tail is pending)
      const u32x4 z4 = {0u, 0u, 0u, 0u};
      st.pk[1][r][0] = z4;
      st.pk[1][r][1] = z4;
      st.l[r][0] = 0.f;
      st.l[r][1] = 0.f;
    }
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) st.ka[kk] = lds0 + fa.row_off + (((2 * kk + fa.hi) ^ fa.row_swz) << 4);                 // slot 0
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int half = 0; half < 2; ++half)                                                                                   // V slot R - 1
        st.va[2 * db + half] = lds0 + (2 * R - 1) * TILE_B + fa.tr_off[half] + (((4 * db + fa.tr_col) ^ fa.tr_swz[half]) << 4);
    FCSA_PASS_MARK(3);
    wait_vm<0>();
    wg_barrier();
    FCSA_PASS_MARK(4);
    static_for<8>([&](auto kc) { lds_read_k<0>(st.kf[decltype(kc)::value], st.ka[decltype(kc)::value]); });      // K(0), key block 0

    // thresholds: key j of the tile at j0 is visible to this lane's row i iff  j <= min(i + diff, M - 1); the lane compares the
    // tile-relative key row  KOFF + crow(r, 0)  against  thr - j0 - 4 * hi
    int thr0[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) thr0[r] = (p.causal ? min(i0 + 32 * r + diff, p.M - 1) : p.M - 1) - 4 * fa.hi;
    int t_split = p.M / BN;                       // tiles [0, t_split): no masking for this wave
    if (p.causal) t_split = min(t_split, max(0, mw + diff + 1) / BN);
    t_split = min(t_split, nt);

    // the ring walk: K(t) and V(t) live in slot t % R; ka starts at slot 0, va at slot R - 1 (the zero tile "V(-1)")
    int slot = 0;
    auto run = [&](auto masked_tag, int t_begin, int t_end) {
      constexpr bool MASKED = decltype(masked_tag)::value;
      for (int t = t_begin; t < t_end; ++t) {
        const int sm1 = slot == 0 ? R - 1 : slot - 1, sm2 = sm1 == 0 ? R - 1 : sm1 - 1;
        const uint32_t lds_k = lds0 + (uint32_t)sm1 * TILE_B + (uint32_t)wave * 1024u;               // K(t+R-1) -> slot of K(t-1)
        const uint32_t lds_v = lds0 + (uint32_t)(R + sm2) * TILE_B + (uint32_t)wave * 1024u;         // V(t+R-2) -> slot of V(t-2)
        auto dma_k = [&](int i) { dma_piece(lds_k + (uint32_t)i * 4096u, (uint32_t)dk_.piece_offset(i), krs, stk.off); };
        auto dma_v = [&](int i) { dma_piece(lds_v + (uint32_t)i * 4096u, (uint32_t)dv_.piece_offset(i), vrs, stv.off); };
        const uint32_t dk_next = slot + 1 == R ? (uint32_t)(-(R - 1) * TILE_B) : (uint32_t)TILE_B;      // slot(t) -> slot(t+1)
        const uint32_t dv_next = slot == 0 ? (uint32_t)(-(R - 1) * TILE_B) : (uint32_t)TILE_B;          // slot(t-1) -> slot(t)
        const int j0 = t * BN;
        const int thr_cur[2] = {thr0[0] - j0, thr0[1] - j0};
        const int thr_prev[2] = {thr_cur[0] + BN, thr_cur[1] + BN};
        fwd3_tile<T, R, RSUM, MASKED>(st, ts, dk_next, dv_next, thr_prev, thr_cur, dma_k, dma_v);
        if constexpr (!MASKED) ts.close(5);
        stk.off += k_step;
        stv.off += v_step;
        slot = slot + 1 == R ? 0 : slot + 1;
      }
    };
    run(std::false_type{}, 0, t_split);
    FCSA_PASS_MARK(5);
    run(std::true_type{}, t_split, nt);
    FCSA_PASS_MARK(6);
    if (nt > 0) {
      const int jl = (nt - 1) * BN;
      const int thr_last[2] = {thr0[0] - jl, thr0[1] - jl};
      fwd3_drain<T, RSUM>(st, thr_last);
    }
    wait_lgkm<0>();
    wait_vm<0>();                // (the ring's look-ahead requests: they must not land in the epilogue scratch)
    wg_barrier();                // every wave has left the ring: its bytes become the epilogue scratch

    // ---- epilogue: normalise, transpose through the LDS, whole-row stores (RowEpilogue) ----
    char* scr = smem + wave * RowEpilogue<T, D>::BYTES_NOX;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int i = i0 + 32 * r;
      const float inv = 1.f / fmaxf(xhalf_sum(st.l[r][0] + st.l[r][1]), p.l_eps);
      if (i < p.N && p.inv_l != nullptr && fa.hi == 0) p.inv_l[((int64_t)b * p.H + h) * p.N + i] = inv;
      const int rows_valid = min(32, p.N - (mw + 32 * r));
      if (rows_valid > 0) {      // (wave-uniform)
        char* out0 = p.o.p + (int64_t)b * p.o.sb + (int64_t)h * p.o.sh + (int64_t)(mw + 32 * r) * p.o.sn;
        RowEpilogue<T, D>::store(scr, st.o[r], inv, lane, out0, p.o.sn, rows_valid, false, nullptr, 0, 1.f, nullptr, 1, 0, 1.f);
      }
    }
    FCSA_PASS_MARK(7);
    if (pass + 1 < npass) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      wg_barrier();              // the scratch is free again before the next pass's DMA overwrites it
    }
  }   // pass
#ifdef FCSA_TRACE
  if (blockIdx.x == gridDim.x / 2 + 3 && lane == 0) {
    ts.dump(g_trace_fwd3 + 32 * wave, trace_now() - trace_t0);
    for (int a_ = 0; a_ < 2; ++a_) for (int b_ = 0; b_ < 8; ++b_) g_trace_fwd3[32 * wave + 14 + 8 * a_ + b_] = pm[a_][b_] ? pm[a_][b_] - trace_t0 : 0;
  }
#endif
}
#ifdef FCSA_TRACE
}  // namespace fcsa
extern "C" int fcsa_trace_read_fwd3(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fcsa::g_trace_fwd3), sizeof(unsigned long long) * 128);
}
namespace fcsa {
#endif

constexpr int kFwd3Ring = 4;             // K and V ring depth (tiles): 128 KiB of the CU's 160; K is requested 3 tiles ahead, V 2
constexpr bool kFwd3RoundedSums = false; // row sums of the un-rounded P~ (plain adds); true = of the rounded pair (v_dot2c: bit-identical to the
                                         // lean form's output, +5 ... 7 % time: profiles/r05_fwd3_ab_v2.txt)

template <typename T, int R, bool RSUM>
static hipError_t launch_fwd3_t(const FwdParams& p, hipStream_t s) {
  const int MT = (p.N + 255) / 256;
  const int PT = p.causal ? (MT + 1) / 2 : MT;
  size_t lds = (size_t)2 * R * 16384;
  if (lds < (size_t)4 * RowEpilogue<T, 128>::BYTES_NOX) lds = (size_t)4 * RowEpilogue<T, 128>::BYTES_NOX;
  auto kern = fwd3_kernel<T, R, RSUM>;
  static std::atomic<uint64_t> lds_ok{0};
  if (hipError_t e = ensure_dynamic_lds(kern, lds, lds_ok); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(p.B * p.H * PT)), dim3(256), lds, s, p);
  return hipGetLastError();
}

// Escape hatch (same-process A/B, triage): 0 = never take this form.  The environment is read ONCE, when the library is loaded
// (FCSA_FWD_WIDE128=0); afterwards only fcsa_debug_forward_form (include/fcsa.h) changes it -- no getenv on the launch path.
static int wide128_env_default() {
  const char* e = std::getenv("FCSA_FWD_WIDE128");
  return (e != nullptr && e[0] == '0') ? 0 : 1;
}
static std::atomic<int> g_wide128{wide128_env_default()};
int forward_wide128_mode(int set) {      // set < 0: query only; returns the previous value
  return set < 0 ? g_wide128.load(std::memory_order_relaxed) : g_wide128.exchange(set != 0 ? 1 : 0, std::memory_order_relaxed);
}

// Does this problem take the form above?  16-bit D = 128, static exponent shift, no bias, no key mask, no key split, a grid of 256-row
// (causal: paired) workgroups that covers the chip -- or, from 2048 keys, more than half of it -- K / V slices addressable with 32-bit offsets.
bool use_forward_wide128(int dtype, int D, const FwdParams& p) {
  if (D != 128 || (dtype != 1 && dtype != 2)) return false;
  if (p.bias != nullptr || p.mask != nullptr || p.dyn || p.splits > 1) return false;
  if (g_wide128.load(std::memory_order_relaxed) == 0) return false;
  const int MT = (p.N + 255) / 256;
  const int64_t wgs = (int64_t)p.B * p.H * (p.causal ? (MT + 1) / 2 : MT);
  if (wgs < cu_count() * 7 / 8) {
    // round 6 (tools/form_sweep.py, profiles/r06_form_sweep_d128_b.txt): also where the 128-row tiles outnumber the CUs (132 ... 223 of
    // these workgroups on 256 CUs) and the pass is long enough for its prologue: 31 - 36 % faster than the key-split lean form there, and
    // ahead of the 256-row lean form from 2048 keys (level at 1024 keys up to ~176 workgroups, behind beyond)
    const int MT4 = (p.N + 127) / 128;
    if ((int64_t)p.B * p.H * (p.causal ? (MT4 + 1) / 2 : MT4) <= cu_count() || p.M < 2048) return false;
  }
  if ((int64_t)(p.M + 64 * 6) * p.k.sn >= 0x7fffffffLL || (int64_t)(p.M + 64 * 6) * p.v.sn >= 0x7fffffffLL) return false;
  if ((int64_t)(p.N + 256) * p.q.sn >= 0x7fffffffLL) return false;
  return true;
}

hipError_t launch_forward_wide128(int dtype, const FwdParams& p, hipStream_t s) {
  return dtype == 2 ? launch_fwd3_t<BF16, kFwd3Ring, kFwd3RoundedSums>(p, s) : launch_fwd3_t<F16, kFwd3Ring, kFwd3RoundedSums>(p, s);
}

}  // namespace fcsa
