// fcsa_capi.hip -- the C ABI of libfcsa_hip.so (include/fcsa.h): validation, workspace carving and
// launch sequencing.  Replaces the reference's host launchers + pybind module (cu:1630-1933) and
// dispatch.h (dh:38-73): unsupported dtypes / head dims are REJECTED with an error instead of the
// reference's silent no-op default branch (dh:50-52), nothing synchronises the device (cf. cu:1745,
// cu:1889) and every launch goes to the caller's stream (cf. the default-stream launches cu:1720).
#include "../../include/fcsa.h"
#include "fcsa_kernels.h"
#ifdef FCSA_VAR_SPLIT_ENV
#include "dev/fcsa_sweep_env.h"
#endif

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
constexpr float kLog2e = 1.4426950408889634f;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int elem_size(int dtype) { return dtype == FCSA_F32 ? 4 : 2; }

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

bool dim_ok(int d) { return d == 16 || d == 32 || d == 64 || d == 96 || d == 128; }   // cu:84 allowed_dim_heads

int check_problem(const fcsa_problem& p) {
  if (p.dtype != FCSA_F16 && p.dtype != FCSA_BF16 && p.dtype != FCSA_F32)
    return fail(FCSA_ERR_UNSUPPORTED, "unsupported dtype %d (expected f32=0, f16=1, bf16=2)", p.dtype);
  if (!dim_ok(p.dim_head))
    return fail(FCSA_ERR_UNSUPPORTED, "dim_head %d not in {16, 32, 64, 96, 128}", p.dim_head);
  if (p.batch < 0 || p.heads < 0 || p.q_len < 0 || p.k_len < 0)
    return fail(FCSA_ERR_INVALID_ARG, "negative size (B=%d H=%d N=%d M=%d)", p.batch, p.heads, p.q_len, p.k_len);
  if (p.kv_heads != p.heads && p.kv_heads != 1)
    return fail(FCSA_ERR_INVALID_ARG, "kv_heads must be heads (%d) or 1, got %d", p.heads, p.kv_heads);
  if (p.l2norm_qk) {
    if (p.groups < 1 || p.dim_head % p.groups != 0)
      return fail(FCSA_ERR_INVALID_ARG, "groups (%d) must divide dim_head (%d)", p.groups, p.dim_head);
    if (p.dtype == FCSA_F16 && fabsf(p.scale) * kLog2e > 60000.f)
      return fail(FCSA_ERR_UNSUPPORTED, "float16: scale %g puts scale * log2(e) * q^ outside the type", (double)p.scale);
  } else if (p.groups != 1) {
    return fail(FCSA_ERR_INVALID_ARG, "groups must be 1 when l2norm_qk is 0");
  }
  if (!(p.scale == p.scale) || std::isinf(p.scale)) return fail(FCSA_ERR_INVALID_ARG, "scale is NaN or infinite");
  return FCSA_OK;
}

int check_tensor(const char* name, const fcsa_tensor& t, int es, bool required) {
  if (t.ptr == nullptr) return required ? fail(FCSA_ERR_INVALID_ARG, "%s: null pointer", name) : FCSA_OK;
  if ((reinterpret_cast<uintptr_t>(t.ptr) & 15) != 0) return fail(FCSA_ERR_INVALID_ARG, "%s: base not 16-byte aligned", name);
  const int64_t m = 16 / es;
  if (t.stride0 % m || t.stride1 % m || t.stride2 % m)
    return fail(FCSA_ERR_INVALID_ARG, "%s: strides (%lld, %lld, %lld) must keep rows 16-byte aligned", name,
                (long long)t.stride0, (long long)t.stride1, (long long)t.stride2);
  // the kernels address the rows of a 256-row tile with 32-bit byte offsets (buffer loads): keep a tile below 1 GiB
  if (t.stride2 < 0 || t.stride2 * es > (int64_t)(0x3fffffff / 256))
    return fail(FCSA_ERR_UNSUPPORTED, "%s: row stride %lld elements is negative or above 4 MiB", name, (long long)t.stride2);
  return FCSA_OK;
}

fcsa::View view(const fcsa_tensor& t, int es, bool zero_head_stride = false) {
  fcsa::View v;
  v.p = static_cast<char*>(t.ptr);
  v.sb = t.stride0 * es;
  v.sh = zero_head_stride ? 0 : t.stride1 * es;
  v.sn = t.stride2 * es;
  return v;
}

fcsa::View contiguous_view(void* p, int64_t heads, int64_t len, int64_t d, int es, bool zero_head_stride = false) {
  fcsa::View v;
  v.p = static_cast<char*>(p);
  v.sb = heads * len * d * es;
  v.sh = zero_head_stride ? 0 : len * d * es;
  v.sn = d * es;
  return v;
}

// Exponent shift (natural-log units): P~ = exp(s - shift).  Any shift gives the same O; only the saved
// inv_l carries it, and forward / backward derive it identically from the problem description.
//   * l2norm_qk == 0 (the reference extension's contract, q,k pre-normalised by the caller): shift = scale,
//     exactly cu:1216, so inv_l has the reference's values.
//   * fused l2norm: the logit lies in [-bound, +bound], bound = |scale| * groups.
//       f16 : shift = bound - 10  ->  P~ <= e^10 = 22026 < 65504, and typical P~ (logit ~ 0) stays in
//             f16's NORMAL range even for large scale (with the reference's shift, scale = 16 puts exp(-16)
//             = 1e-7 into f16 subnormals and the output error grows 10x).
//       bf16 / f32 (8-bit exponent): every P~ must stay a normal f32 and a row sum (up to ~1e5 keys of it) below f32's top:
//             exp(s - shift) in [e^-85, e^65]  <=>  shift in [bound - 65, 85 - bound], non-empty for bound <= 75.  The shift is
//             the reference's (= scale) wherever that lies in the interval, else the nearest end.  (Round 2 used
//             max(scale, bound - 40): the same values for bound <= 42, but it sent 60 < bound <= 75 -- C5 at the default scale 8 --
//             through the two-pass dynamic form, and it kept the reference's underflow for groups = 1, scale > 42.)
// A static shift only works while the whole logit range fits the exponent range of the type P~ is rounded to.  Real rows peak
// far below the theoretical bound once groups > 1 (found by the fuzz test: f16, groups >= 4, scale >= 8 underflowed every P~ of a
// row to 0; the reference's shift = scale overflows there instead).
// Beyond the safe range the forward kernel finds each row's max logit first and shifts by that ("dynamic"); what it saves
// for the backward is then log2(1 / sum_j exp(S_ij)) -- the value the backward kernels seed their S accumulators with anyway --
// so nothing ever holds exp() of the full logit range and there is no limit on scale * groups.
constexpr float kStaticTop = 65.f, kStaticBottom = 85.f;      // exp(s - shift) stays within [e^-85, e^65] (bf16 / f32)
// An additive bias makes the exponent unbounded.  bf16 / f32 absorb it inside the static window (e^-85 ... e^65 leaves +-20 around any
// shift; a strongly negative bias underflows to an exact 0 weight).  f16 does not: its static shift puts the largest logit at e^10 of a
// 65504 range, so a bias of +1.1 on such a logit overflowed P~ to inf (found by an exploratory fuzz seed in round 3: f16, scale 1, bias
// ~ N(0, 0.5)).  f16 problems WITH a bias therefore always take the per-row-reference form (whose online max includes the bias).
// bf16 / f32 WITH a bias: the static window only has room for the bias while the logits themselves leave it: with shift = scale the
// largest exponent is (bound - scale) + bias, and e^88 is f32's top (inf -> NaN rows).  Up to bound = scale * groups = 40 that leaves
// a positive bias at least +45 of headroom on top of ln(M) for the row sum (the documented limit of the static form, include/fcsa.h);
// beyond it -- round 3 drew the line at 60, where the headroom was down to +23 ... +36 (round 4 advice) -- problems with a bias take
// the per-row form, whose online reference includes the bias and has no limit at all.
constexpr float kStaticBoundBias = 40.f;
bool dynamic_shift(const fcsa_problem& p, bool has_bias) {
  if (!p.l2norm_qk) return false;
  const float bound = fabsf(p.scale) * (float)p.groups;
  if (p.dtype == FCSA_F16) return has_bias || bound > 11.f;
  return 2.f * bound > kStaticTop + kStaticBottom || (has_bias && bound > kStaticBoundBias);
}

float exponent_shift(const fcsa_problem& p, bool has_bias) {
  if (!p.l2norm_qk) return p.scale;
  if (dynamic_shift(p, has_bias)) return 0.f;
  const float bound = fabsf(p.scale) * (float)p.groups;
  if (p.dtype == FCSA_F16) return bound - 10.f;
  const float lo = bound - kStaticTop, hi = kStaticBottom - bound;
  return p.scale < lo ? lo : (p.scale > hi ? hi : p.scale);
}

// Row-sum clamp: the reference clamps l at 1e-10 (cu:83, cu:1239) with shift = scale; with another shift the
// same clamp in the reference's units is 1e-10 * exp(scale - shift) (kept inside f32's normal range).
float rowsum_eps(const fcsa_problem& p, bool has_bias) {
  if (dynamic_shift(p, has_bias)) return 1e-30f;          // row sums are >= 1 there (the max element contributes exp(0))
  float e = 1e-10f * expf(p.scale - exponent_shift(p, has_bias));
  if (!(e > 1e-37f)) e = 1e-37f;
  if (e > 1e30f) e = 1e30f;
  return e;
}

struct BwdLayout {
  size_t delta, dq_slab, dk_slab, dv_slab, total;
  bool need_dq_slab, need_dk_slab, need_dv_slab, fuse_norm;
  int dq_splits;          // > 1: split-key dQ kernel, dq_slab holds dq_splits partial slabs
  int dkv_splits;         // > 1: split-query dK/dV kernel, dk_slab / dv_slab hold dkv_splits partial slabs each
};

// ---- split counts (forward keys / dQ keys / dK-dV queries) -------------------------------------------------------------------------
// Where a problem's 128-position tiles (causal, since round 6: PAIRS of tiles) cannot fill the chip, several workgroups share one tile's loop range and write f32
// partials that a second pass (fwd_combine_kernel / finalize) sums.  How many: rounds 2 - 5 took "enough workgroups for two per CU";
// since round 6 the count is the argmin of a small cost model over s = 1 .. 16 (16-bit types; float32 keeps the old rule).  The model
// prices, in microseconds on MI355X, what a launch with s splits costs:
//   * the form the launchers run for that count: form A = the 8-wave one-workgroup-per-CU forms (forward: wave halves split the keys,
//     fcsa_fwd.hip use_ksplit_fwd -- rows <= 128 bytes while tiles * s <= CUs, wider rows always; backward: whatever runs un-split),
//     else 4-wave workgroups, one per CU (form B) or -- rows <= 128 bytes and more workgroups than CUs -- two per CU (form C);
//   * a workgroup's time t0 + c * positions, t0 and c growing with D (the exponentials do not shrink with it: floor);
//   * rounds of workgroups over the slots, a partly filled last round at alpha + (1 - alpha) * fill;
//   * the second pass: a launch + s partial slabs read once.
// Constants: least squares in log space over tools/split_sweep.py tables of 27 shapes x 7 counts per kernel, D = 16 .. 128
// (profiles/r06_split_sweep_*.txt; tools/split_model_fit.py prints them and the table below).  Mean / worst regret of the model's choice
// against the measured best: forward 0.9 / 10.9 %, dQ 0.3 / 4.5 %, dK/dV 0.2 / 3.6 %; the rule it replaces: 11.2 / 44 %, 8.9 / 42 %,
// 11.8 / 52 % (it split 160 .. 224 tiles of a 2048-position problem three or four ways where the un-split 8-wave form is 30 - 50 %
// faster, and took counts that leave a quarter-full last round).  The ONE definition both the workspace sizes and the launches use.
struct SplitModel { double tA, cA, tB, cB, tC, cC, a0, b0, k0, k1, alpha; int slabs, extra; };
static const SplitModel kSplitFwd = {5.57, 8.68, 3.61, 11.8, 5.77, 16.1, 0.0, 0.536, 7.77, 0.177, 0.585, 1, 1};
static const SplitModel kSplitDq  = {11.3, 9.94, 2.35, 12.5, 1.0, 20.9, 0.369, 0.244, 12.2, 0.455, 0.526, 1, 0};
static const SplitModel kSplitDkv = {11.5, 12.0, 2.25, 15.5, 1.0, 26.6, 0.11, 0.281, 13.5, 0.34, 0.528, 2, 0};
// tiles: 128-position tiles the un-split grid has; rows: rows of ONE partial slab; len: positions the split loop runs over
static double split_cost(const SplitModel& m, int D, int64_t tiles, int64_t rows, int len, int s, int cus, bool form_a) {
  const bool wide = D * 2 > 128;
  const double ft = m.a0 + (1.0 - m.a0) * D / 64.0, fc = m.b0 + (1.0 - m.b0) * std::max(0.44, D / 64.0);
  const int64_t tot = tiles * s;
  double t0 = m.tA, c = m.cA;
  int64_t slots = cus;
  if (!form_a) {
    if (wide || tot <= cus) { t0 = m.tB; c = m.cB; }
    else { t0 = m.tC; c = m.cC; slots = 2 * (int64_t)cus; }
  }
  const double per = t0 * ft + c * fc * ((double)len / s) / 1024.0;
  const int64_t full = tot / slots, rem = tot % slots;
  const double rounds = full == 0 ? 1.0 : (double)full + (rem == 0 ? 0.0 : m.alpha + (1.0 - m.alpha) * (double)rem / (double)slots);
  const double second = s == 1 ? 0.0 : m.k0 + m.k1 * m.slabs * (double)s * (double)rows * (D + m.extra) * 4.0 / 1e6;
  return rounds * per + second;
}
// fwd: the forward's form rule (see above); else the backward's (un-split = form A, split = 4-wave workgroups)
static int best_split(const SplitModel& m, bool fwd, int D, int64_t tiles, int64_t rows, int len) {
  const int cus = fcsa::cu_count();
  if (tiles <= 0 || tiles >= cus) return 1;
  int best = 1;
  double best_cost = split_cost(m, D, tiles, rows, len, 1, cus, true);
  for (int s = 2; s <= 16 && len / s >= 512; ++s) {
    const bool form_a = fwd && (D * 2 > 128 || tiles * s <= cus);
    const double cost = split_cost(m, D, tiles, rows, len, s, cus, form_a);
    if (cost < best_cost) { best_cost = cost; best = s; }
  }
  return best;
}
// the rule of rounds 2 - 5, still used for float32: two 4-wave workgroups per CU where those run two waves per SIMD (rows <= 128 bytes)
int split_target(const fcsa_problem& p) { return (elem_size(p.dtype) * p.dim_head <= 128 ? 2 : 1) * fcsa::cu_count(); }
static int split_by_target(const fcsa_problem& p, int64_t wgs, int len) {
  const int target = split_target(p);
  if (wgs <= 0 || wgs >= target / 2) return 1;
  int64_t s = (target + wgs - 1) / wgs;
  if (s > 16) s = 16;
  if (s > len / 512) s = len / 512;
  return s >= 2 ? (int)s : 1;
}

// Split-key dQ: every split keeps >= 512 keys.  C4 (1 x 8 heads x 1024 queries, 8192 keys): 64 row tiles, 8 splits.
int backward_dq_splits(const fcsa_problem& p) {
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only (tools/split_sweep.py, dev/fcsa_sweep_env.h): the count from the environment, per call
  if (const int v = fcsa_dev::env_int("FCSA_DQ_SPLITS"); v >= 1 && (!p.causal || (elem_size(p.dtype) == 2 && p.q_len >= 256)))
    return std::min(std::min(v, 16), std::max(1, p.k_len / 64));
#endif
  if (p.causal) {
    // Causal (round 6), like the forward: the workgroups are PAIRS of 128-row tiles; where the pairs cannot fill the chip each row tile's
    // key range (up to its diagonal) is split.  16-bit only.
    if (elem_size(p.dtype) != 2 || p.q_len < 256) return 1;
    const int mt = (p.q_len + 127) / 128;
    return best_split(kSplitDq, false, p.dim_head, (int64_t)p.batch * p.heads * ((mt + 1) / 2), (int64_t)p.batch * p.heads * p.q_len, p.k_len);
  }
  const int64_t wgs = (int64_t)p.batch * p.heads * ((p.q_len + 127) / 128);
  if (elem_size(p.dtype) == 2) return best_split(kSplitDq, false, p.dim_head, wgs, (int64_t)p.batch * p.heads * p.q_len, p.k_len);
  return split_by_target(p, wgs, p.k_len);
}

// Split-query dK/dV: the mirror image -- few keys, many queries (B * H * ceil(M / 128) key tiles cannot fill the chip),
// K/V with heads (the single-headed form already reduces over slabs), every split keeps >= 512 queries.  Partial dK^ / dV go to f32
// slabs [batch * heads][split][M][D] and the finalize kernel sums them (and applies the l2norm backward to dK^).
int backward_dkv_splits(const fcsa_problem& p) {
  // (single-headed K/V, round 6: split like any other problem -- its per-head slabs simply become heads x splits slabs for the same finalize launch)
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only
  if (const int v = fcsa_dev::env_int("FCSA_DKV_SPLITS"); v >= 1 && (!p.causal || (elem_size(p.dtype) == 2 && p.k_len >= 256)))
    return std::min(std::min(v, 16), std::max(1, p.q_len / 64));
#endif
  if (p.causal) {
    // Causal (round 6): the workgroups are PAIRS of 128-key tiles; where the pairs cannot fill the chip each key tile's query range (from
    // its diagonal down) is split.  16-bit only.
    if (elem_size(p.dtype) != 2 || p.k_len < 256) return 1;
    const int kt = (p.k_len + 127) / 128;
    return best_split(kSplitDkv, false, p.dim_head, (int64_t)p.batch * p.heads * ((kt + 1) / 2), (int64_t)p.batch * p.heads * p.k_len,
                      std::min(p.q_len, p.k_len));      // (a key tile sees the queries from its diagonal down: at most k_len of them)
  }
  const int64_t wgs = (int64_t)p.batch * p.heads * ((p.k_len + 127) / 128);
  if (elem_size(p.dtype) == 2) return best_split(kSplitDkv, false, p.dim_head, wgs, (int64_t)p.batch * p.heads * p.k_len, p.q_len);
  return split_by_target(p, wgs, p.q_len);
}

int log2_blocks_per_group(const fcsa_problem& p) {     // log2(group size / 8), or -1 if not a power of two of 8-blocks
  const int dg = p.dim_head / (p.groups > 0 ? p.groups : 1);
  if (dg % 8 != 0) return -1;
  int m = dg / 8, lg = 0;
  while ((1 << lg) < m) ++lg;
  // ONE group over the whole head (groups = 1) may be any number of 8-blocks: the fused forms sum over the power-of-two lane / k-step
  // block that contains the row, and the padding positions of that block contribute nothing (RowEpilogue: lanes c >= D / 8 hold 0;
  // finish_q_frags: k-steps >= KS do not exist).  D = 96: 12 blocks -> 4.  (Round 6; until then D = 96 took the slab + finalize path.)
  if (p.groups <= 1) return lg;
  return (1 << lg) == m ? lg : -1;
}
bool fusable_groups(const fcsa_problem& p) { return log2_blocks_per_group(p) >= 0; }

BwdLayout bwd_layout(const fcsa_problem& p) {
  BwdLayout L;
  const bool single = p.kv_heads == 1 && p.heads > 1;
  const size_t qn = (size_t)p.batch * p.heads * p.q_len;
  const size_t kn = (size_t)p.batch * p.heads * p.k_len;      // slabs are per q-head
  // the l2norm backward is fused into the dQ / dKV epilogues when every group is 8 * 2^k features wide;
  // otherwise (odd group sizes) and for the head reduction of single-headed K/V the kernels write f32
  // slabs that the finalize kernel reduces / differentiates.
  L.fuse_norm = p.l2norm_qk != 0 && fusable_groups(p);
  L.dq_splits = backward_dq_splits(p);
  L.need_dq_slab = (p.l2norm_qk != 0 && !L.fuse_norm) || L.dq_splits > 1;
  L.dkv_splits = backward_dkv_splits(p);
  L.need_dk_slab = single || (p.l2norm_qk != 0 && !L.fuse_norm) || L.dkv_splits > 1;
  L.need_dv_slab = single || L.dkv_splits > 1;
  size_t off = 0;
  L.delta = off;   off = align_up(off + qn * 4, 256);
  L.dq_slab = off; off = align_up(off + (L.need_dq_slab ? qn * p.dim_head * 4 * (size_t)L.dq_splits : 0), 256);
  L.dk_slab = off; off = align_up(off + (L.need_dk_slab ? kn * p.dim_head * 4 * (size_t)L.dkv_splits : 0), 256);
  L.dv_slab = off; off = align_up(off + (L.need_dv_slab ? kn * p.dim_head * 4 * (size_t)L.dkv_splits : 0), 256);
  L.total = off;
  return L;
}

int launch_check(hipError_t e, const char* what) {
  if (e != hipSuccess) return fail(FCSA_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return FCSA_OK;
}

// ---- optional per-kernel timing (fcsa_profile_*) -------------------------------------------------
struct TimedLaunch { const char* name; hipEvent_t start, stop; };
std::mutex g_prof_mu;
std::atomic<bool> g_prof_on{false};       // read lock-free on every launch; the mutex only guards the record list
std::vector<TimedLaunch> g_prof;

// run one kernel launch, bracketed by events on ITS stream when profiling is enabled
template <typename F> int timed(const char* name, const char* what, hipStream_t s, F&& launch) {
  if (!g_prof_on.load(std::memory_order_relaxed)) return launch_check(launch(), what);
  TimedLaunch t{name, nullptr, nullptr};
  if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess)
    return fail(FCSA_ERR_LAUNCH, "%s: hipEventCreate failed", what);
  (void)hipEventRecord(t.start, s);
  const hipError_t e = launch();
  (void)hipEventRecord(t.stop, s);
  { std::lock_guard<std::mutex> g(g_prof_mu); g_prof.push_back(t); }
  return launch_check(e, what);
}

}  // namespace

extern "C" {

const char* fcsa_last_error(void) { return g_err.c_str(); }

int fcsa_profile_enable(int32_t enable) {
  g_prof_on.store(enable != 0, std::memory_order_relaxed);
  return FCSA_OK;
}

int fcsa_profile_collect(fcsa_kernel_stat* stats, int32_t capacity) {
  std::vector<TimedLaunch> rec;
  { std::lock_guard<std::mutex> g(g_prof_mu); rec.swap(g_prof); }
  std::vector<fcsa_kernel_stat> agg;
  for (TimedLaunch& t : rec) {
    float ms = 0.f;
    const bool ok = hipEventSynchronize(t.stop) == hipSuccess && hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess;
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
    if (!ok) continue;
    fcsa_kernel_stat* st = nullptr;
    for (auto& a : agg) if (strcmp(a.name, t.name) == 0) st = &a;
    if (st == nullptr) {
      fcsa_kernel_stat n;
      memset(&n, 0, sizeof(n));
      strncpy(n.name, t.name, sizeof(n.name) - 1);
      n.min_ms = ms; n.max_ms = ms;
      agg.push_back(n);
      st = &agg.back();
    }
    st->calls += 1;
    st->total_ms += ms;
    if (ms < st->min_ms) st->min_ms = ms;
    if (ms > st->max_ms) st->max_ms = ms;
  }
  if (stats != nullptr)
    for (int i = 0; i < (int)agg.size() && i < capacity; ++i) stats[i] = agg[i];
  return (int)agg.size();
}

int fcsa_debug(char* buf, size_t buf_bytes) {
  if (buf != nullptr && buf_bytes > 0) {
    snprintf(buf, buf_bytes,
             "libfcsa_hip abi=%d arch=gfx950 dtypes=f32,f16,bf16 dim_head=16,32,64,96,128 "
             "kernels=l2norm,l2norm_pair,fwd(32 rows/wave; lean two-wave form at D=96/128),fwd2(64 rows/wave),fwd3(D=128: 64 rows/wave, 1 wave/SIMD),fwd_ksplit(128 rows, wave halves split the keys),fwd_split+combine,"
             "fwd_dyn(per-row shift),bwd_dq(+split-key; key-split form on 8 waves),bwd_dkv(+lean; query-split form on 8 waves),bwd_dbias,finalize",
             FCSA_ABI_VERSION);
  }
  return FCSA_ABI_VERSION;
}

int fcsa_debug_forward_form(int32_t form) { return fcsa::forward_wide128_mode(form); }

int fcsa_l2norm(int32_t dtype, int32_t batch, int32_t heads, int32_t len, int32_t dim_head, int32_t groups,
                const fcsa_tensor* x, void* xn, float* inv_norm, void* stream) {
  if (dtype != FCSA_F16 && dtype != FCSA_BF16 && dtype != FCSA_F32) return fail(FCSA_ERR_UNSUPPORTED, "fcsa_l2norm: dtype %d not supported", dtype);
  if (batch < 0 || heads < 0 || len < 0) return fail(FCSA_ERR_INVALID_ARG, "fcsa_l2norm: negative size (B=%d H=%d L=%d)", batch, heads, len);
  if (batch == 0 || heads == 0 || len == 0) return FCSA_OK;      // no row: nothing to launch (empty tensors carry NULL pointers)
  if (x == nullptr || xn == nullptr) return fail(FCSA_ERR_INVALID_ARG, "fcsa_l2norm: null argument");
  if (dim_head <= 0 || dim_head % 8 != 0) return fail(FCSA_ERR_UNSUPPORTED, "fcsa_l2norm: dim_head %d must be a multiple of 8", dim_head);
  if (dim_head > 512) return fail(FCSA_ERR_UNSUPPORTED, "fcsa_l2norm: dim_head %d > 512", dim_head);
  if (groups < 1 || dim_head % groups != 0) return fail(FCSA_ERR_INVALID_ARG, "fcsa_l2norm: groups (%d) must divide dim_head (%d)", groups, dim_head);
  const int es = elem_size(dtype);
  if (int rc = check_tensor("x", *x, es, true)) return rc;
  fcsa::NormParams np;
  np.x = view(*x, es);
  np.xn = static_cast<char*>(xn);
  np.inv_norm = inv_norm;
  np.B = batch; np.H = heads; np.L = len; np.D = dim_head; np.G = groups;
  np.eps = 1e-12f;                                        // F.normalize default (flash_cosine_sim_attention.py:46)
  np.out_scale = 1.f;
  hipStream_t s = static_cast<hipStream_t>(stream);
  return timed("l2norm", "l2norm", s, [&] { return fcsa::launch_l2norm(dtype, np, s); });
}

// Split-key forward (the count: best_split above): only where the 128-row tiles (causal: pairs of them) cannot fill the chip, the static
// exponent shift applies (partials with a common shift add up exactly) and every split keeps >= 512 keys.
static int forward_splits(const fcsa_problem& p) {
  if (dynamic_shift(p, false)) return 1;      // (never called with a bias: fcsa_forward only splits bias-free problems)
  if (p.causal) {
    // Causal (round 6): the workgroups are PAIRS of 128-row tiles (constant work: about k_len + 128 keys each); where the pairs cannot fill
    // the chip -- one sequence of 4096 with 8 heads is 128 pairs on 256 CUs, and takes as long as two sequences -- each row tile's key range
    // (up to its diagonal) is split.  16-bit only (the form rule of the model); the count from the same model with the pair as the tile.
    if (elem_size(p.dtype) != 2 || p.q_len < 256) return 1;
    const int mt = (p.q_len + 127) / 128;
    const int64_t pairs = (int64_t)p.batch * p.heads * ((mt + 1) / 2);
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only
    if (const int v = fcsa_dev::env_int("FCSA_SPLITS"); v >= 1) return std::min(std::min(v, 16), std::max(1, p.k_len / 64));
#endif
    return best_split(kSplitFwd, true, p.dim_head, pairs, (int64_t)p.batch * p.heads * p.q_len, p.k_len);
  }
  const int64_t wgs = (int64_t)p.batch * p.heads * ((p.q_len + 127) / 128);
  if (wgs <= 0) return 1;
#ifdef FCSA_VAR_SPLIT_ENV      // sweep builds only
  if (const int v = fcsa_dev::env_int("FCSA_SPLITS"); v >= 1) return std::min(std::min(v, 16), std::max(1, p.k_len / 64));
#endif
  if (elem_size(p.dtype) == 2) return best_split(kSplitFwd, true, p.dim_head, wgs, (int64_t)p.batch * p.heads * p.q_len, p.k_len);
  return split_by_target(p, wgs, p.k_len);
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

size_t fcsa_forward_workspace_bytes(const fcsa_problem* p) {
  if (p == nullptr || p->batch <= 0 || p->heads <= 0 || p->q_len <= 0 || p->dim_head <= 0) return 0;
  const int s = forward_splits(*p);
  if (s <= 1) return 0;
  const size_t rows = (size_t)s * p->batch * p->heads * p->q_len;
  return align256(rows * p->dim_head * 4) + align256(rows * 4);
}

// Zero-size problems (empty tensors: what torch hands over has NULL data pointers then).  batch, heads or q_len == 0: the forward has no
// output element; k_len == 0: every query row is a row without a valid key, for which the kernels' semantics are o = 0 (cu:1239) and
// finite (zero) gradients.  Nothing is launched; the outputs that DO have elements are zero-filled on the caller's stream, row by row
// (any strides).  The reference launches a zero-sized grid there (a CUDA error it prints and ignores, cu:17-28).
static int zero_rows(const char* name, const fcsa_tensor& t, int es, int B, int H, int L, int D, hipStream_t s) {
  if (B <= 0 || H <= 0 || L <= 0) return FCSA_OK;
  if (int rc = check_tensor(name, t, es, true)) return rc;
  for (int b = 0; b < B; ++b)
    for (int h = 0; h < H; ++h) {
      char* base = static_cast<char*>(t.ptr) + ((int64_t)b * t.stride0 + (int64_t)h * t.stride1) * es;
      const hipError_t e = t.stride2 == D ? hipMemsetAsync(base, 0, (size_t)L * D * es, s)
                                          : hipMemset2DAsync(base, (size_t)t.stride2 * es, 0, (size_t)D * es, (size_t)L, s);
      if (e != hipSuccess) return fail(FCSA_ERR_LAUNCH, "%s: zero fill failed: %s", name, hipGetErrorString(e));
    }
  return FCSA_OK;
}

int fcsa_forward(const fcsa_forward_args* a) {
  if (a == nullptr) return fail(FCSA_ERR_INVALID_ARG, "null args");
  const fcsa_problem& p = a->p;
  if (int rc = check_problem(p)) return rc;
  if (p.causal && a->mask != nullptr) return fail(FCSA_ERR_INVALID_ARG, "mask should not be given if causal (cu:1675)");
  const int es = elem_size(p.dtype);
  if (p.batch == 0 || p.heads == 0 || p.q_len == 0) return FCSA_OK;      // no output element (k-side saved state is not needed by a backward either)
  if (p.k_len == 0) {                                                     // rows without a key: o = 0; inv_l = 1 (never read: the backward below has no key to visit)
    hipStream_t s0 = static_cast<hipStream_t>(a->stream);
    if (int rc = zero_rows("o", a->o, es, p.batch, p.heads, p.q_len, p.dim_head, s0)) return rc;
    if (a->inv_l != nullptr) {
      const float one = 1.f;
      uint32_t bits; memcpy(&bits, &one, 4);
      if (hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(a->inv_l), (int)bits, (size_t)p.batch * p.heads * p.q_len, s0) != hipSuccess)
        return fail(FCSA_ERR_LAUNCH, "inv_l: fill failed");
    }
    return FCSA_OK;
  }
  if (int rc = check_tensor("q", a->q, es, true)) return rc;
  if (int rc = check_tensor("k", a->k, es, true)) return rc;
  if (int rc = check_tensor("v", a->v, es, true)) return rc;
  if (int rc = check_tensor("o", a->o, es, true)) return rc;
  hipStream_t s = static_cast<hipStream_t>(a->stream);
  const bool single = p.kv_heads == 1 && p.heads > 1;

  fcsa::FwdParams fp;
  bool fuse_q = false;
  fp.q = view(a->q, es);
  fp.k = view(a->k, es, single);
  fp.v = view(a->v, es, single);
  fp.o = view(a->o, es);
  if (p.l2norm_qk) {
    const fcsa_norm_state& n = a->norm;
    if (n.kn == nullptr) return fail(FCSA_ERR_INVALID_ARG, "l2norm_qk needs the norm.kn buffer");
    if (n.qn == nullptr && fcsa_forward_needs_qn(&p, a->inv_l != nullptr || n.rq != nullptr))
      return fail(FCSA_ERR_INVALID_ARG, "l2norm_qk needs the norm.qn buffer for this problem (fcsa_forward_needs_qn)");
    fcsa::NormParams nq, nk;
    nq.eps = nk.eps = 1e-12f;
    nq.D = nk.D = p.dim_head; nq.G = nk.G = p.groups;
    nq.x = fp.q; nq.xn = static_cast<char*>(n.qn); nq.inv_norm = n.rq;
    nq.B = p.batch; nq.H = p.heads; nq.L = p.q_len;
    nq.out_scale = p.scale * kLog2e;       // qn = c1 * q^ (one rounding): the kernels then need no per-logit multiply
    nk.x = view(a->k, es); nk.xn = static_cast<char*>(n.kn); nk.inv_norm = n.rk;
    nk.B = p.batch; nk.H = p.kv_heads; nk.L = p.k_len;
    nk.out_scale = 1.f;
    // 16-bit types with group sizes of 8 * 2^k: q is normalised in the forward kernel's prologue (load_q_frags), which also
    // writes qn / rq for the backward; only k takes the HBM pass.  Otherwise both go through the row kernel.
    fuse_q = p.dtype != FCSA_F32 && fusable_groups(p);
    if (fuse_q) {
      if (int rc = timed("l2norm", "l2norm(k)", s, [&] { return fcsa::launch_l2norm(p.dtype, nk, s); })) return rc;
    } else {
      if (int rc = timed("l2norm", "l2norm(q,k)", s, [&] { return fcsa::launch_l2norm_pair(p.dtype, nq, nk, s); })) return rc;
    }
    if (!fuse_q) fp.q = contiguous_view(n.qn, p.heads, p.q_len, p.dim_head, es);
    fp.k = contiguous_view(n.kn, p.kv_heads, p.k_len, p.dim_head, es, single);
  }
  fp.inv_l = a->inv_l;
  fp.mask = a->mask;
  fp.bias = static_cast<const char*>(a->attn_bias);
  fp.B = p.batch; fp.H = p.heads; fp.N = p.q_len; fp.M = p.k_len;
  fp.causal = p.causal; fp.bias_batch = p.bias_batch_dim;
  fp.c1 = p.scale * kLog2e;
  const bool has_bias = a->attn_bias != nullptr;
  fp.c2 = exponent_shift(p, has_bias) * kLog2e;
  fp.bias_c = kLog2e;
  fp.l_eps = rowsum_eps(p, has_bias);
  fp.q_scaled = p.l2norm_qk ? 1 : 0;
  fp.q_raw = fuse_q ? 1 : 0;
  fp.qn_out = fuse_q ? static_cast<char*>(a->norm.qn) : nullptr;
  fp.rq_out = fuse_q ? a->norm.rq : nullptr;
  fp.G = p.groups; fp.lgm = fuse_q ? log2_blocks_per_group(p) : 0; fp.norm_eps = 1e-12f;
  fp.dyn = dynamic_shift(p, has_bias) ? 1 : 0;      // then inv_l holds log2 of the normaliser
  fp.splits = 1; fp.ws_o = nullptr; fp.ws_l = nullptr;
  if (a->workspace != nullptr && a->attn_bias == nullptr) {
    const int sp = forward_splits(p);
    const size_t need = fcsa_forward_workspace_bytes(&p);
    if (sp > 1 && a->workspace_bytes >= need && (reinterpret_cast<uintptr_t>(a->workspace) & 255) == 0) {
      const size_t rows = (size_t)sp * p.batch * p.heads * p.q_len;
      fp.splits = sp;
      fp.ws_o = static_cast<float*>(a->workspace);
      fp.ws_l = reinterpret_cast<float*>(static_cast<char*>(a->workspace) + align256(rows * p.dim_head * 4));
    }
  }
  return timed("fwd", "forward", s, [&] { return fcsa::launch_forward(p.dtype, p.dim_head, fp, s); });
}

int fcsa_forward_needs_qn(const fcsa_problem* p, int32_t need_backward) {
  if (p == nullptr || !p->l2norm_qk) return 0;
  if (need_backward) return 1;
  return (p->dtype != FCSA_F32 && fusable_groups(*p)) ? 0 : 1;
}

size_t fcsa_backward_workspace_bytes(const fcsa_problem* p) {
  if (p == nullptr) return 0;
  return bwd_layout(*p).total;
}

int fcsa_backward(const fcsa_backward_args* a) {
  if (a == nullptr) return fail(FCSA_ERR_INVALID_ARG, "null args");
  const fcsa_problem& p = a->p;
  if (int rc = check_problem(p)) return rc;
  if (p.causal && a->mask != nullptr) return fail(FCSA_ERR_INVALID_ARG, "mask should not be given if causal (cu:1675)");
  const int es = elem_size(p.dtype);
  if (p.batch == 0 || p.heads == 0) return FCSA_OK;                        // every gradient is empty
  if (p.q_len == 0 || p.k_len == 0) {                                      // no (query, key) pair: the gradients that have elements are zero (d_bias has none)
    hipStream_t s0 = static_cast<hipStream_t>(a->stream);
    if (int rc = zero_rows("dq", a->dq, es, p.batch, p.heads, p.q_len, p.dim_head, s0)) return rc;
    if (int rc = zero_rows("dk", a->dk, es, p.batch, p.kv_heads, p.k_len, p.dim_head, s0)) return rc;
    return zero_rows("dv", a->dv, es, p.batch, p.kv_heads, p.k_len, p.dim_head, s0);
  }
  if (int rc = check_tensor("d_out", a->d_out, es, true)) return rc;
  if (int rc = check_tensor("o", a->o, es, true)) return rc;
  if (int rc = check_tensor("v", a->v, es, true)) return rc;
  if (int rc = check_tensor("dq", a->dq, es, true)) return rc;
  if (int rc = check_tensor("dk", a->dk, es, true)) return rc;
  if (int rc = check_tensor("dv", a->dv, es, true)) return rc;
  if (!p.l2norm_qk) {
    if (int rc = check_tensor("q", a->q, es, true)) return rc;
    if (int rc = check_tensor("k", a->k, es, true)) return rc;
  } else if (a->norm.qn == nullptr || a->norm.kn == nullptr || a->norm.rq == nullptr || a->norm.rk == nullptr) {
    return fail(FCSA_ERR_INVALID_ARG, "l2norm_qk backward needs norm.qn, norm.kn, norm.rq, norm.rk from forward");
  }
  if (a->inv_l == nullptr) return fail(FCSA_ERR_INVALID_ARG, "inv_l: null pointer");
  if (a->attn_bias == nullptr && a->d_bias != nullptr) return fail(FCSA_ERR_INVALID_ARG, "d_bias without attn_bias");
  const BwdLayout L = bwd_layout(p);
  if (a->workspace == nullptr || a->workspace_bytes < L.total)
    return fail(FCSA_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", a->workspace_bytes, L.total);
  if ((reinterpret_cast<uintptr_t>(a->workspace) & 255) != 0) return fail(FCSA_ERR_WORKSPACE, "workspace not 256-byte aligned");

  hipStream_t s = static_cast<hipStream_t>(a->stream);
  const bool single = p.kv_heads == 1 && p.heads > 1;
  char* ws = static_cast<char*>(a->workspace);

  fcsa::BwdParams bp;
  if (p.l2norm_qk) {
    bp.q = contiguous_view(a->norm.qn, p.heads, p.q_len, p.dim_head, es);
    bp.k = contiguous_view(a->norm.kn, p.kv_heads, p.k_len, p.dim_head, es, single);
  } else {
    bp.q = view(a->q, es);
    bp.k = view(a->k, es, single);
  }
  bp.v = view(a->v, es, single);
  bp.o = view(a->o, es);
  bp.d_out = view(a->d_out, es);
  // split-key dQ needs (batch, head) to be one flat index of the dq output (the finalize kernel sums the partial slabs per
  // (batch * head) row block); otherwise the unsplit kernel runs
  const bool dq_flat = a->dq.stride0 == (int64_t)p.heads * a->dq.stride1;
  const bool want_dbias = a->attn_bias != nullptr && a->d_bias != nullptr;
  const int dq_splits = (L.dq_splits > 1 && dq_flat && !(p.causal && a->attn_bias != nullptr)) ? L.dq_splits : 1;      // (causal splits: bias-free kernels only)
  const bool dq_slab = dq_splits > 1 || (p.l2norm_qk != 0 && !L.fuse_norm);
  bp.dq_splits = dq_splits;
  bp.dq_split_stride = (int64_t)p.q_len * p.dim_head * 4;
  // split-query dK/dV: same condition on dk / dv (flat (batch, head) index for the finalize kernel); the bias form keeps the unsplit kernel
  const bool dkv_flat = a->dk.stride0 == (int64_t)p.heads * a->dk.stride1 && a->dv.stride0 == (int64_t)p.heads * a->dv.stride1;
  const int dkv_splits = (L.dkv_splits > 1 && (dkv_flat || single) && a->attn_bias == nullptr) ? L.dkv_splits : 1;
  const bool dk_slab = dkv_splits > 1 || single || (p.l2norm_qk != 0 && !L.fuse_norm);
  const bool dv_slab = dkv_splits > 1 || single;
  bp.dkv_splits = dkv_splits;
  bp.dkv_split_stride = (int64_t)p.k_len * p.dim_head * 4;
  bp.dq_f32 = dq_slab;
  bp.dk_f32 = dk_slab;
  bp.dv_f32 = dv_slab;
  if (dq_splits > 1) {          // slab layout [batch * heads][split][N][D]: (b, h) block stride = splits * N * D floats
    bp.dq.p = ws + L.dq_slab;
    bp.dq.sn = (int64_t)p.dim_head * 4;
    bp.dq.sh = (int64_t)dq_splits * p.q_len * p.dim_head * 4;
    bp.dq.sb = (int64_t)p.heads * bp.dq.sh;
  } else {
    bp.dq = dq_slab ? contiguous_view(ws + L.dq_slab, p.heads, p.q_len, p.dim_head, 4) : view(a->dq, es);
  }
  if (dkv_splits > 1) {         // slab layout [batch * heads][split][M][D], like the split dq slabs
    bp.dk.p = ws + L.dk_slab;
    bp.dk.sn = (int64_t)p.dim_head * 4;
    bp.dk.sh = (int64_t)dkv_splits * p.k_len * p.dim_head * 4;
    bp.dk.sb = (int64_t)p.heads * bp.dk.sh;
    bp.dv = bp.dk;
    bp.dv.p = ws + L.dv_slab;
  } else {
    bp.dk = dk_slab ? contiguous_view(ws + L.dk_slab, p.heads, p.k_len, p.dim_head, 4) : view(a->dk, es);
    bp.dv = dv_slab ? contiguous_view(ws + L.dv_slab, p.heads, p.k_len, p.dim_head, 4) : view(a->dv, es);
  }
  bp.inv_l = a->inv_l;
  bp.delta = reinterpret_cast<float*>(ws + L.delta);
  bp.mask = a->mask;
  bp.bias = static_cast<const char*>(a->attn_bias);
  bp.d_bias = a->d_bias;
  bp.B = p.batch; bp.H = p.heads; bp.N = p.q_len; bp.M = p.k_len;
  bp.causal = p.causal; bp.bias_batch = p.bias_batch_dim;
  bp.c1 = p.scale * kLog2e;
  bp.c2 = exponent_shift(p, a->attn_bias != nullptr) * kLog2e;
  bp.invl_log2 = dynamic_shift(p, a->attn_bias != nullptr) ? 1 : 0;
  bp.bias_c = kLog2e;
  bp.scale = p.scale;
  bp.q_scaled = p.l2norm_qk ? 1 : 0;
  bp.G = p.groups; bp.lgm = L.fuse_norm ? log2_blocks_per_group(p) : 0; bp.norm_eps = 1e-12f;
  bp.rq = (L.fuse_norm && dq_splits <= 1) ? a->norm.rq : nullptr;    // fused: dq kernel writes the final dq
  bp.rk = (L.fuse_norm && !single && dkv_splits <= 1) ? a->norm.rk : nullptr;          // fused: dkv kernel writes the final dk

  // 1. dQ (also publishes delta), 2. dK/dV, 3. head reduction + l2norm backward where needed
  if (int rc = timed("bwd_dq", "backward dq", s, [&] { return fcsa::launch_backward_dq(p.dtype, p.dim_head, bp, s); })) return rc;
  if (int rc = timed("bwd_dkv", "backward dkv", s, [&] { return fcsa::launch_backward_dkv(p.dtype, p.dim_head, bp, s); })) return rc;
  if (want_dbias) {      // d_bias from recomputed dS tiles; needs delta, which the dQ kernel published
    if (int rc = timed("bwd_dbias", "backward d_bias", s, [&] { return fcsa::launch_backward_dbias(p.dtype, p.dim_head, bp, s); })) return rc;
  }

  fcsa::NormBwdParams nb;
  nb.eps = 1e-12f;
  nb.D = p.dim_head;
  nb.B = p.batch;
  nb.xn_scale = 1.f;
  if (dq_splits > 1) {          // sum the split slabs ("heads" of a flat (batch * head) batch), then the l2norm backward if any
    nb.B = p.batch * p.heads;
    nb.slab = ws + L.dq_slab; nb.slab_f32 = 1; nb.HS = dq_splits; nb.HO = 1; nb.L = p.q_len;
    if (p.l2norm_qk) { nb.xn = static_cast<const char*>(a->norm.qn); nb.inv_norm = a->norm.rq; nb.G = p.groups; nb.xn_scale = 1.f / (p.scale * kLog2e); }
    else             { nb.xn = nullptr; nb.inv_norm = nullptr; nb.G = 1; }
    nb.dx = view(a->dq, es);
    nb.dx.sb = nb.dx.sh;          // flat (batch * head) index: stride0 == heads * stride1 (checked above)
    nb.dx.sh = 0;
  } else if (dq_slab) {
    nb.slab = ws + L.dq_slab; nb.slab_f32 = 1; nb.HS = p.heads; nb.HO = p.heads; nb.L = p.q_len;
    nb.xn = static_cast<const char*>(a->norm.qn); nb.inv_norm = a->norm.rq; nb.G = p.groups;
    nb.xn_scale = 1.f / (p.scale * kLog2e);           // qn holds c1 * q^
    nb.dx = view(a->dq, es);
  }
  // (the dq pass is launched below, together with the dk / dv passes where there are any: one grid instead of two or three)
  const fcsa::NormBwdParams nq = nb;
  nb.B = p.batch;
  fcsa::NormBwdParams nk = nb, nv = nb;
  if (dk_slab) {
    nk.slab = ws + L.dk_slab; nk.slab_f32 = 1; nk.HS = p.heads; nk.HO = p.kv_heads; nk.L = p.k_len;
    nk.xn_scale = 1.f;
    if (p.l2norm_qk) { nk.xn = static_cast<const char*>(a->norm.kn); nk.inv_norm = a->norm.rk; nk.G = p.groups; }
    else             { nk.xn = nullptr; nk.inv_norm = nullptr; nk.G = 1; }
    nk.dx = view(a->dk, es);
  }
  if (dv_slab) {
    nv.slab = ws + L.dv_slab; nv.slab_f32 = 1; nv.HS = p.heads; nv.HO = p.kv_heads; nv.L = p.k_len;
    nv.xn = nullptr; nv.inv_norm = nullptr; nv.G = 1; nv.xn_scale = 1.f;
    nv.dx = view(a->dv, es);
  }
  if (dkv_splits > 1) {         // the splits are the "heads" of a flat (batch * head) batch, summed down to one
    for (fcsa::NormBwdParams* n : {&nk, &nv}) {
      if (single) {               // slabs [batch][heads x splits][M][D] summed down to the one K/V head
        n->HS = p.heads * dkv_splits; n->HO = 1;
        continue;
      }
      n->B = p.batch * p.heads; n->HS = dkv_splits; n->HO = 1;
      n->dx.sb = n->dx.sh;        // flat (batch * head) index: stride0 == heads * stride1 (checked above)
      n->dx.sh = 0;
    }
  }
  if (dq_slab && dk_slab && dv_slab) {      // causal problems on small grids (dQ and dK/dV both split), split dQ with single-headed K/V: one launch
    if (int rc = timed("finalize", "finalize dq+dk+dv", s, [&] { return fcsa::launch_l2norm_bwd_triple(p.dtype, nq, nk, nv, s); })) return rc;
  } else if (dq_slab && dk_slab) {          // l2norm groups that are not 8 * 2^k features wide
    if (int rc = timed("finalize", "finalize dq+dk", s, [&] { return fcsa::launch_l2norm_bwd_pair(p.dtype, nq, nk, s); })) return rc;
  } else {
    if (dq_slab) {
      if (int rc = timed("finalize", "finalize dq", s, [&] { return fcsa::launch_l2norm_bwd(p.dtype, nq, s); })) return rc;
    }
    if (dk_slab && dv_slab) {      // single-headed K/V, split-query dK/dV: both reductions in one launch
      if (int rc = timed("finalize", "finalize dk+dv", s, [&] { return fcsa::launch_l2norm_bwd_pair(p.dtype, nk, nv, s); })) return rc;
    } else if (dk_slab) {
      if (int rc = timed("finalize", "finalize dk", s, [&] { return fcsa::launch_l2norm_bwd(p.dtype, nk, s); })) return rc;
    } else if (dv_slab) {
      if (int rc = timed("finalize", "finalize dv", s, [&] { return fcsa::launch_l2norm_bwd(p.dtype, nv, s); })) return rc;
    }
  }
  // Two degenerate problems whose dq and dk are EXACTLY zero, and for which the kernels' arithmetic is not meaningful:
  //   scale == 0: the logits do not depend on q, k.  The kernels carry c1 = scale * log2(e) folded into the saved q^ and undo it with
  //     1 / c1 in the l2norm backward and the dK^ epilogue: 0 * inf = NaN.
  //   l2norm groups of ONE feature (groups == dim_head): x^ = sign(x), whose derivative is zero; the tangent-space projection
  //     r (g - x^ <g, x^>) is then a pure cancellation that the 16-bit rounding of c1 * q^ leaves at 2^-11 |g| / |x| -- unbounded for
  //     elements near zero (measured 0.7 against an exact 0).
  // The reference's autograd through F.normalize / scale * sim gives zeros in both; dv (and d_bias) are what the kernels wrote.
  if (p.scale == 0.f || (p.l2norm_qk && p.groups == p.dim_head)) {
    if (int rc = zero_rows("dq", a->dq, es, p.batch, p.heads, p.q_len, p.dim_head, s)) return rc;
    if (int rc = zero_rows("dk", a->dk, es, p.batch, p.kv_heads, p.k_len, p.dim_head, s)) return rc;
  }
  return FCSA_OK;
}

}  // extern "C"
