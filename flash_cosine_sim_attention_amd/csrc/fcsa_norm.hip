// fcsa_norm.hip -- HBM-bound row kernels around the attention kernels (gfx950):
//
//   l2norm_kernel      x -> x / max(||x_group||, eps), inv_norm      (reference: l2norm_tensors /
//                      grouped_l2norm / F.normalize, flash_cosine_sim_attention.py:44-65; in the
//                      reference these are separate eager PyTorch passes outside the extension)
//   l2norm_bwd_kernel  dx = l2norm_backward(sum_over_heads(slab))    (reference: torch.autograd through
//                      F.normalize, plus the f32->dtype casts at cu:1904-1914 and, for single-headed
//                      K/V, the head reduction the reference does with f32 atomics, cu:1613-1619)
//
// Both stream rows with 16-byte loads/stores: D/8 lanes per row, so a wave covers 64/(D/8) rows and
// one load instruction moves 1 KiB per wave (guide G13).  The group reductions are lane shuffles.
// A per-(row, group) scalar path handles group sizes that are not a multiple of 8.
#include "fcsa_common.cuh"
#include "fcsa_kernels.h"

namespace fcsa {

// 8 consecutive elements of type T <-> 8 floats (16 bytes for f16/bf16, 32 bytes for f32)
template <typename T> FCSA_DEV void load8(const char* p, float (&f)[8]) {
  if constexpr (Traits<T>::ES == 4) {
    const f32x4 a = reinterpret_cast<const f32x4*>(p)[0], b = reinterpret_cast<const f32x4*>(p)[1];
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[e] = a[e]; f[4 + e] = b[e]; }
  } else {
    const u32x4 u = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = Traits<T>::lo(u[e]); f[2 * e + 1] = Traits<T>::hi(u[e]); }
  }
}
template <typename T> FCSA_DEV void store8(char* p, const float (&f)[8]) {
  if constexpr (Traits<T>::ES == 4) {
    f32x4 a = {f[0], f[1], f[2], f[3]}, b = {f[4], f[5], f[6], f[7]};
    reinterpret_cast<f32x4*>(p)[0] = a;
    reinterpret_cast<f32x4*>(p)[1] = b;
  } else {
    u32x4 u;
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = Traits<T>::pack2(f[2 * e], f[2 * e + 1]);
    *reinterpret_cast<u32x4*>(p) = u;
  }
}

// sum `v` over the LPG consecutive lanes [base, base + LPG) this lane belongs to
// (lpg = 2^a * odd: groups start at multiples of 2^a lanes -- a row takes G * lpg lanes -- so the butterfly runs over the 2^a part and
//  only the odd factor is gathered lane by lane: D = 96, one group: 2 + 3 shuffles instead of 12)
FCSA_DEV float group_sum(float v, int lpg, int pos_in_group, int lane) {
  const int p2 = lpg & -lpg;
  for (int o = 1; o < p2; o <<= 1) v += __shfl_xor(v, o, 64);
  if (p2 == lpg) return v;
  const int base = lane - pos_in_group + (pos_in_group & (p2 - 1));
  float s = 0.f;
  for (int t = 0; t < lpg; t += p2) s += __shfl(v, base + t, 64);
  return s;
}

// row index -> (batch, head, position).  64-bit integer division costs a few hundred VALU instructions on
// gfx950, which dominated these short kernels; row counts below 2^31 (every realistic call) take the 32-bit path.
FCSA_DEV void split_row(int64_t row, int64_t nrows, int L, int H, int& b, int& h, int& l) {
  if (nrows <= 0x7fffffff) {
    const uint32_t r = (uint32_t)row, bh = r / (uint32_t)L;
    l = (int)(r - bh * (uint32_t)L);
    b = (int)(bh / (uint32_t)H);
    h = (int)(bh - (uint32_t)b * (uint32_t)H);
  } else {
    const int64_t bh = row / L;
    l = (int)(row - bh * L);
    b = (int)(bh / H);
    h = (int)(bh - (int64_t)b * H);
  }
}

struct RowMap {
  int tpr, rpw, lane, c;
  int64_t row;
  bool active;
  FCSA_DEV void init(int D, int64_t nrows, int block = blockIdx.x) {
    tpr = D >> 3;
    rpw = 64 / tpr;
    lane = threadIdx.x & 63;
    const int r = lane / tpr;
    c = lane - r * tpr;
    const int64_t wave = (int64_t)block * (blockDim.x >> 6) + (threadIdx.x >> 6);
    row = wave * rpw + r;
    active = r < rpw && row < nrows;
  }
};

// ---------------------------------------------------------------------------------------------
// forward, group size multiple of 8
// ---------------------------------------------------------------------------------------------
// Row chunks (16 bytes) per thread.  Rounds 1 - 5 took four (all loads of a thread issued before the first use) and, since the first half
// of round 6, one on grids below two blocks per CU; re-measured over sizes in round 6: ONE is never slower and 7 - 18 % faster on the large
// tensors too (C3's k: 11.7 -> 10.9 us, D = 128: 21.2 -> 18.0, (8,8,4096,64): 21.2 -> 17.4; two: level with one; three, eight: slower --
// profiles/r06_ab_norm_unroll.txt): four times the blocks in flight hide the HBM latency better than four loads per thread.
constexpr int kNormUnroll = 1;
template <typename T, int UNROLL = kNormUnroll>
FCSA_DEV void l2norm_rows(const NormParams& p, int block) {
  const int64_t nrows = (int64_t)p.B * p.H * p.L;
  const int dg = p.D / p.G, lpg = dg >> 3;
  RowMap m[UNROLL];
  float f[UNROLL][8];
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    m[u].init(p.D, nrows, block * UNROLL + u);
#pragma unroll
    for (int e = 0; e < 8; ++e) f[u][e] = 0.f;
    if (m[u].active) {
      int b, h, l;
      split_row(m[u].row, nrows, p.L, p.H, b, h, l);
      load8<T>(p.x.p + (int64_t)b * p.x.sb + (int64_t)h * p.x.sh + (int64_t)l * p.x.sn + m[u].c * 8 * Traits<T>::ES, f[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < UNROLL; ++u) {
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss += f[u][e] * f[u][e];
    ss = group_sum(ss, lpg, m[u].c % lpg, m[u].lane);
    const float inv = 1.f / fmaxf(sqrtf(ss), p.eps);          // F.normalize: x / max(||x||, eps)
    if (m[u].active) {
#pragma unroll
      for (int e = 0; e < 8; ++e) f[u][e] *= inv * p.out_scale;
      store8<T>(p.xn + (m[u].row * p.D + m[u].c * 8) * Traits<T>::ES, f[u]);
      if (p.inv_norm != nullptr && (m[u].c % lpg) == 0) p.inv_norm[m[u].row * p.G + m[u].c / lpg] = inv;
    }
  }
}

template <typename T, int UNROLL = kNormUnroll>
__global__ void __launch_bounds__(256) l2norm_kernel(const NormParams p) { l2norm_rows<T, UNROLL>(p, blockIdx.x); }

// q and k of one attention call in ONE grid (both are short HBM-bound passes; a second launch costs
// about as much as the pass itself): blocks [0, blocks_a) take `a`, the rest take `b`.
template <typename T>
__global__ void __launch_bounds__(256) l2norm_pair_kernel(const NormParams a, const NormParams b, const int blocks_a) {
  if ((int)blockIdx.x < blocks_a) l2norm_rows<T>(a, blockIdx.x);
  else l2norm_rows<T>(b, blockIdx.x - blocks_a);
}

// forward, any group size: one thread per (row, group)
template <typename T>
__global__ void __launch_bounds__(256) l2norm_generic_kernel(const NormParams p) {
  typedef typename Traits<T>::elem E;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)p.B * p.H * p.L * p.G;
  if (idx >= total) return;
  const int64_t row = idx / p.G;
  const int g = (int)(idx - row * p.G);
  const int64_t bh = row / p.L;
  const int l = (int)(row - bh * p.L);
  const int b = (int)(bh / p.H), h = (int)(bh - (int64_t)b * p.H);
  const int dg = p.D / p.G;
  const E* x = reinterpret_cast<const E*>(p.x.p + (int64_t)b * p.x.sb + (int64_t)h * p.x.sh + (int64_t)l * p.x.sn) + g * dg;
  E* xn = reinterpret_cast<E*>(p.xn) + row * p.D + g * dg;
  float ss = 0.f;
  for (int e = 0; e < dg; ++e) { const float v = (float)x[e]; ss += v * v; }
  const float inv = 1.f / fmaxf(sqrtf(ss), p.eps);
  for (int e = 0; e < dg; ++e) xn[e] = (E)((float)x[e] * inv * p.out_scale);
  if (p.inv_norm != nullptr) p.inv_norm[idx] = inv;
}

// ---------------------------------------------------------------------------------------------
// finalize: head reduction (+ l2norm backward), group size multiple of 8 (or no norm)
// ---------------------------------------------------------------------------------------------
template <typename T>
FCSA_DEV void l2norm_bwd_rows(const NormBwdParams& p, int block) {
  RowMap m;
  const int64_t nrows = (int64_t)p.B * p.HO * p.L;
  m.init(p.D, nrows, block);
  const bool norm = p.xn != nullptr;
  const int dg = p.D / p.G, lpg = norm ? (dg >> 3) : 1;
  float g[8], xh[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { g[e] = 0.f; xh[e] = 0.f; }
  int b = 0, h = 0, l = 0;
  if (m.active) {
    split_row(m.row, nrows, p.L, p.HO, b, h, l);
    const int nsum = (p.HS == p.HO) ? 1 : p.HS;
    if (p.slab_f32) {
      // f32 slabs: the loads of FOUR slabs (8 x 16 bytes per lane) are issued before the first add -- the head loop is the whole kernel
      // (single-headed K/V at C5: 8 slabs of 33.5 MB each way), and one slab per iteration left a single dependent load pair in flight
      const int64_t hstride = (int64_t)p.L * p.D * 4;
      const char* s0 = p.slab + ((((int64_t)b * p.HS + (p.HS == p.HO ? h : 0)) * p.L + l) * p.D + m.c * 8) * 4;
      int hs = 0;
      for (; hs + 4 <= nsum; hs += 4) {
        f32x4 a[4], c[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const f32x4* sp = reinterpret_cast<const f32x4*>(s0 + (hs + u) * hstride);
          a[u] = sp[0];      // (plain loads: the slabs were written by the previous kernel and are still on chip; nontemporal loads
          c[u] = sp[1];      //  measured 15.4 vs 14.7 us at C5, profiles/r06_ab_norm.txt)
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) { g[e] += a[u][e]; g[4 + e] += c[u][e]; }
      }
      for (; hs < nsum; ++hs) {
        const f32x4* sp = reinterpret_cast<const f32x4*>(s0 + hs * hstride);
        const f32x4 a = sp[0], c = sp[1];
#pragma unroll
        for (int e = 0; e < 4; ++e) { g[e] += a[e]; g[4 + e] += c[e]; }
      }
    } else {
      for (int hs = 0; hs < nsum; ++hs) {
        const int64_t srow = ((int64_t)b * p.HS + (p.HS == p.HO ? h : hs)) * p.L + l;
        float t[8];
        load8<T>(p.slab + (srow * p.D + m.c * 8) * Traits<T>::ES, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] += t[e];
      }
    }
    if (norm) {
      load8<T>(p.xn + (m.row * p.D + m.c * 8) * Traits<T>::ES, xh);
#pragma unroll
      for (int e = 0; e < 8; ++e) xh[e] *= p.xn_scale;
    }
  }
  if (norm) {
    float dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dot += g[e] * xh[e];
    dot = group_sum(dot, lpg, m.c % lpg, m.lane);
    if (m.active) {
      const float r = p.inv_norm[m.row * p.G + m.c / lpg];
      const bool clamped = r >= 1.f / p.eps;                  // ||x|| <= eps: the clamp has zero slope
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = clamped ? g[e] * r : r * (g[e] - xh[e] * dot);
    }
  }
  if (m.active)
    store8<T>(p.dx.p + (int64_t)b * p.dx.sb + (int64_t)h * p.dx.sh + (int64_t)l * p.dx.sn + m.c * 8 * Traits<T>::ES, g);
}

template <typename T>
__global__ void __launch_bounds__(256) l2norm_bwd_kernel(const NormBwdParams p) { l2norm_bwd_rows<T>(p, blockIdx.x); }

// two finalize passes of one backward call in ONE grid (dk and dv of single-headed K/V: each is a short HBM-bound pass, a
// second launch costs about as much as the pass itself): blocks [0, blocks_a) take `a`, the rest take `b`
template <typename T>
__global__ void __launch_bounds__(256) l2norm_bwd_pair_kernel(const NormBwdParams a, const NormBwdParams b, const int blocks_a) {
  if ((int)blockIdx.x < blocks_a) l2norm_bwd_rows<T>(a, blockIdx.x);
  else l2norm_bwd_rows<T>(b, blockIdx.x - blocks_a);
}

// three passes (dq, dk, dv slabs of a backward whose dQ AND dK/dV launches were split -- causal problems on small grids) in one grid
template <typename T>
__global__ void __launch_bounds__(256) l2norm_bwd_triple_kernel(const NormBwdParams a, const NormBwdParams b, const NormBwdParams c, const int blocks_a,
                                                                const int blocks_ab) {
  if ((int)blockIdx.x < blocks_a) l2norm_bwd_rows<T>(a, blockIdx.x);
  else if ((int)blockIdx.x < blocks_ab) l2norm_bwd_rows<T>(b, blockIdx.x - blocks_a);
  else l2norm_bwd_rows<T>(c, blockIdx.x - blocks_ab);
}

// finalize, any group size: one thread per (row, group)
template <typename T>
__global__ void __launch_bounds__(256) l2norm_bwd_generic_kernel(const NormBwdParams p) {
  typedef typename Traits<T>::elem E;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)p.B * p.HO * p.L * p.G;
  if (idx >= total) return;
  const int64_t row = idx / p.G;
  const int gi = (int)(idx - row * p.G);
  const int64_t bh = row / p.L;
  const int l = (int)(row - bh * p.L);
  const int b = (int)(bh / p.HO), h = (int)(bh - (int64_t)b * p.HO);
  const int dg = p.D / p.G;
  const int nsum = (p.HS == p.HO) ? 1 : p.HS;
  const E* xn = p.xn ? reinterpret_cast<const E*>(p.xn) + row * p.D + gi * dg : nullptr;
  E* dx = reinterpret_cast<E*>(p.dx.p + (int64_t)b * p.dx.sb + (int64_t)h * p.dx.sh + (int64_t)l * p.dx.sn) + gi * dg;
  auto grad = [&](int e) {
    float s = 0.f;
    for (int hs = 0; hs < nsum; ++hs) {
      const int64_t srow = ((int64_t)b * p.HS + (p.HS == p.HO ? h : hs)) * p.L + l;
      const int64_t o = srow * p.D + gi * dg + e;
      s += p.slab_f32 ? reinterpret_cast<const float*>(p.slab)[o] : (float)reinterpret_cast<const E*>(p.slab)[o];
    }
    return s;
  };
  if (xn == nullptr) {
    for (int e = 0; e < dg; ++e) dx[e] = (E)grad(e);
    return;
  }
  float dot = 0.f;
  for (int e = 0; e < dg; ++e) dot += grad(e) * (float)xn[e] * p.xn_scale;
  const float r = p.inv_norm[idx];
  const bool clamped = r >= 1.f / p.eps;
  for (int e = 0; e < dg; ++e) {
    const float ge = grad(e);
    dx[e] = (E)(clamped ? ge * r : r * (ge - (float)xn[e] * p.xn_scale * dot));
  }
}

// ---------------------------------------------------------------------------------------------
template <typename T>
static hipError_t launch_l2norm_t(const NormParams& p, hipStream_t s) {
  const int64_t nrows = (int64_t)p.B * p.H * p.L;
  if (nrows == 0) return hipSuccess;
  const int dg = p.D / p.G;
  if (dg % 8 == 0) {
    const int rows_per_block = kNormUnroll * 4 * (64 / (p.D / 8));
    const int64_t blocks = (nrows + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(l2norm_kernel<T>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  } else {
    const int64_t total = nrows * p.G;
    hipLaunchKernelGGL(l2norm_generic_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

template <typename T>
static hipError_t launch_l2norm_bwd_t(const NormBwdParams& p, hipStream_t s) {
  const int64_t nrows = (int64_t)p.B * p.HO * p.L;
  if (nrows == 0) return hipSuccess;
  const int dg = p.D / p.G;
  if (p.xn == nullptr || dg % 8 == 0) {
    const int rows_per_block = 4 * (64 / (p.D / 8));
    hipLaunchKernelGGL(l2norm_bwd_kernel<T>, dim3((unsigned)((nrows + rows_per_block - 1) / rows_per_block)), dim3(256), 0, s, p);
  } else {
    const int64_t total = nrows * p.G;
    hipLaunchKernelGGL(l2norm_bwd_generic_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p);
  }
  return hipGetLastError();
}

// both tensors with the 16-byte row kernel in one launch; requires group size % 8 == 0 and equal D, G
template <typename T>
static hipError_t launch_l2norm_pair_t(const NormParams& a, const NormParams& b, hipStream_t s) {
  const int rows_per_block = kNormUnroll * 4 * (64 / (a.D / 8));
  const int64_t ra = (int64_t)a.B * a.H * a.L, rb = (int64_t)b.B * b.H * b.L;
  const int64_t ba = (ra + rows_per_block - 1) / rows_per_block, bb = (rb + rows_per_block - 1) / rows_per_block;
  if (ba + bb == 0) return hipSuccess;
  hipLaunchKernelGGL(l2norm_pair_kernel<T>, dim3((unsigned)(ba + bb)), dim3(256), 0, s, a, b, (int)ba);
  return hipGetLastError();
}

hipError_t launch_l2norm_pair(int dtype, const NormParams& a, const NormParams& b, hipStream_t s) {
  if ((a.D / a.G) % 8 != 0 || a.D != b.D || a.G != b.G) {          // generic group sizes: two launches
    const hipError_t e = launch_l2norm(dtype, a, s);
    return e != hipSuccess ? e : launch_l2norm(dtype, b, s);
  }
  if (dtype == 2) return launch_l2norm_pair_t<BF16>(a, b, s);
  if (dtype == 1) return launch_l2norm_pair_t<F16>(a, b, s);
  if (dtype == 0) return launch_l2norm_pair_t<F32>(a, b, s);
  return hipErrorInvalidValue;
}

hipError_t launch_l2norm(int dtype, const NormParams& p, hipStream_t s) {
  if (dtype == 2) return launch_l2norm_t<BF16>(p, s);
  if (dtype == 1) return launch_l2norm_t<F16>(p, s);
  if (dtype == 0) return launch_l2norm_t<F32>(p, s);
  return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_l2norm_bwd_pair_t(const NormBwdParams& a, const NormBwdParams& b, hipStream_t s) {
  const int rows_per_block = 4 * (64 / (a.D / 8));
  const int64_t ra = (int64_t)a.B * a.HO * a.L, rb = (int64_t)b.B * b.HO * b.L;
  const int64_t ba = (ra + rows_per_block - 1) / rows_per_block, bb = (rb + rows_per_block - 1) / rows_per_block;
  if (ba + bb == 0) return hipSuccess;
  hipLaunchKernelGGL(l2norm_bwd_pair_kernel<T>, dim3((unsigned)(ba + bb)), dim3(256), 0, s, a, b, (int)ba);
  return hipGetLastError();
}

// both passes in one launch when both can use the 16-byte row kernel (equal D; group size % 8 == 0 or no norm), else two launches
hipError_t launch_l2norm_bwd_pair(int dtype, const NormBwdParams& a, const NormBwdParams& b, hipStream_t s) {
  auto rows_ok = [](const NormBwdParams& p) { return p.xn == nullptr || (p.D / p.G) % 8 == 0; };
  if (!rows_ok(a) || !rows_ok(b) || a.D != b.D) {
    const hipError_t e = launch_l2norm_bwd(dtype, a, s);
    return e != hipSuccess ? e : launch_l2norm_bwd(dtype, b, s);
  }
  if (dtype == 2) return launch_l2norm_bwd_pair_t<BF16>(a, b, s);
  if (dtype == 1) return launch_l2norm_bwd_pair_t<F16>(a, b, s);
  if (dtype == 0) return launch_l2norm_bwd_pair_t<F32>(a, b, s);
  return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_l2norm_bwd_triple_t(const NormBwdParams& a, const NormBwdParams& b, const NormBwdParams& c, hipStream_t s) {
  const int rows_per_block = 4 * (64 / (a.D / 8));
  auto blocks = [&](const NormBwdParams& p) { return ((int64_t)p.B * p.HO * p.L + rows_per_block - 1) / rows_per_block; };
  const int64_t ba = blocks(a), bb = blocks(b), bc = blocks(c);
  if (ba + bb + bc == 0) return hipSuccess;
  hipLaunchKernelGGL(l2norm_bwd_triple_kernel<T>, dim3((unsigned)(ba + bb + bc)), dim3(256), 0, s, a, b, c, (int)ba, (int)(ba + bb));
  return hipGetLastError();
}

// three passes in one launch under the pair's conditions, else one launch + a pair
hipError_t launch_l2norm_bwd_triple(int dtype, const NormBwdParams& a, const NormBwdParams& b, const NormBwdParams& c, hipStream_t s) {
  auto rows_ok = [](const NormBwdParams& p) { return p.xn == nullptr || (p.D / p.G) % 8 == 0; };
  if (!rows_ok(a) || !rows_ok(b) || !rows_ok(c) || a.D != b.D || a.D != c.D) {
    const hipError_t e = launch_l2norm_bwd(dtype, a, s);
    return e != hipSuccess ? e : launch_l2norm_bwd_pair(dtype, b, c, s);
  }
  if (dtype == 2) return launch_l2norm_bwd_triple_t<BF16>(a, b, c, s);
  if (dtype == 1) return launch_l2norm_bwd_triple_t<F16>(a, b, c, s);
  if (dtype == 0) return launch_l2norm_bwd_triple_t<F32>(a, b, c, s);
  return hipErrorInvalidValue;
}

hipError_t launch_l2norm_bwd(int dtype, const NormBwdParams& p, hipStream_t s) {
  if (dtype == 2) return launch_l2norm_bwd_t<BF16>(p, s);
  if (dtype == 1) return launch_l2norm_bwd_t<F16>(p, s);
  if (dtype == 0) return launch_l2norm_bwd_t<F32>(p, s);
  return hipErrorInvalidValue;
}

}  // namespace fcsa
