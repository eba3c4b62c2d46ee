"""CPU oracle for fused cosine-similarity attention (numpy, float64 by default).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
`flash_cosine_sim_attention_amd`) may import this module; only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do, and only
as the checker.

This is a restatement, in plain numpy, of the algorithm the reference
(lucidrains/flash-cosine-sim-attention v0.1.40) implements for the hot path.
Each function cites the reference lines it follows (paths relative to the
reference checkout, `fcsa.py` = flash_cosine_sim_attention/flash_cosine_sim_attention.py,
`cu` = flash_cosine_sim_attention/flash_cosine_sim_attention_cuda.cu).

Parity pinning: `tests/golden/make_golden.py` imports the reference's own
pure-PyTorch `plain_cosine_sim_attention` / `flash_cosine_sim_attention_cpu`
(and torch.autograd through them) in the build container and stores their
outputs as fixtures under `tests/golden/*.npz`; `tests/test_oracle_golden.py`
checks every function in this file against those fixtures.
"""
from __future__ import annotations

import math
import numpy as np

__all__ = [
    "l2norm", "l2norm_backward", "plain_attention", "tiled_attention",
    "attention_forward_stats", "attention_backward", "causal_valid_count",
    "algorithmic_flops", "round_to", "rounded_operands", "attention_backward_emulated",
]

LOG2E = 1.4426950408889634


# ----------------------------------------------------------------------------
# 16-bit operand rounding ("operand-faithful" evaluation)
# ----------------------------------------------------------------------------

def round_to(x: np.ndarray, dtype_name: str | None) -> np.ndarray:
    """x rounded to `dtype_name` ("bf16" | "f16" | "f32" | None) with round-to-nearest-even, returned as float64."""
    x = np.asarray(x, dtype=np.float64)
    if dtype_name in (None, "f64"):
        return x
    if dtype_name == "f32":
        return x.astype(np.float32).astype(np.float64)
    if dtype_name == "f16":
        return x.astype(np.float16).astype(np.float64)
    if dtype_name == "bf16":
        f = np.ascontiguousarray(x.astype(np.float32))
        u = f.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)      # RNE on the upper 16 bits (finite values)
        out = r.view(np.float32).astype(np.float64)
        return np.where(np.isfinite(f), out, f.astype(np.float64))
    raise ValueError(dtype_name)


def rounded_operands(qh, kh, scale, operand_dtype):
    """The operands every 16-bit implementation of this op feeds its S product (the reference's too: it normalises in the input
    dtype, fcsa.py:57-65): kh rounded to the dtype, and qh rounded AFTER folding c1 = scale * log2(e) in, as the gfx950 kernels
    do (they compute exp2(c1 qh . kh - c2)); returned in qh's units.  Exact arithmetic on THESE operands is the "operand-faithful"
    reference: it leaves out the one error every implementation shares and that grows with the logit range scale * groups."""
    if operand_dtype in (None, "f64", "f32"):
        return qh, kh
    c1 = abs(float(scale)) * LOG2E
    c1 = c1 if c1 > 0 else 1.0
    return round_to(qh * c1, operand_dtype) / c1, round_to(kh, operand_dtype)


# ----------------------------------------------------------------------------
# l2norm  (fcsa.py:38-65)
# ----------------------------------------------------------------------------

def l2norm(x: np.ndarray, groups: int = 1, eps: float = 1e-12, return_inv_norm: bool = False):
    """Grouped L2 normalisation over the last dim.

    fcsa.py:50-55 reshapes the last dim to (groups, d/groups) and normalises each
    group; fcsa.py:44-46 (GPU path) is F.normalize: x / max(||x||, eps), eps=1e-12.
    (fcsa.py:38-42, the CPU path, uses where(norm > eps, norm, eps) which is the
    same function of the norm.)
    """
    shape = x.shape
    d = shape[-1]
    assert d % groups == 0, "groups must divide the feature dimension"
    xg = x.reshape(*shape[:-1], groups, d // groups)
    norm = np.sqrt((xg.astype(np.float64) ** 2).sum(-1, keepdims=True))
    inv = 1.0 / np.maximum(norm, eps)
    out = (xg * inv).reshape(shape).astype(x.dtype)
    if return_inv_norm:
        return out, inv[..., 0]
    return out


def l2norm_backward(dxhat: np.ndarray, x: np.ndarray, groups: int = 1, eps: float = 1e-12, xh_used: np.ndarray | None = None) -> np.ndarray:
    """Gradient of `l2norm` w.r.t. x given the gradient w.r.t. its output.

    The reference leaves this to torch.autograd through F.normalize
    (fcsa.py:320-321 sits outside the autograd.Function); restated analytically:
    per group, with n = max(||x||, eps) and xh = x / n,
        dx = (dxh - xh <dxh, xh>) / n      if ||x|| > eps
        dx = dxh / eps                     otherwise (clamp has zero slope)
    xh_used: the normalised rows an implementation actually holds (e.g. rounded to 16 bit, `rounded_operands`) and uses in the
    projection instead of the exact x / n.
    """
    shape = x.shape
    d = shape[-1]
    xg = x.reshape(*shape[:-1], groups, d // groups).astype(np.float64)
    dg = dxhat.reshape(*shape[:-1], groups, d // groups).astype(np.float64)
    norm = np.sqrt((xg ** 2).sum(-1, keepdims=True))
    n = np.maximum(norm, eps)
    xh = xg / n
    if xh_used is not None:
        xh = np.asarray(xh_used, dtype=np.float64).reshape(xg.shape)
    dot = (dg * xh).sum(-1, keepdims=True)
    dx = np.where(norm > eps, (dg - xh * dot) / n, dg / eps)
    return dx.reshape(shape)


# ----------------------------------------------------------------------------
# shape canonicalisation  (fcsa.py:90-99, cu:1647-1660)
# ----------------------------------------------------------------------------

def _canon(q, k, v, attn_bias, attn_bias_batch_dim):
    merged = q.ndim == 3
    if merged:
        assert k.ndim == 3 and v.ndim == 3, \
            "if batch and heads are merged for queries, keys and values must also have 3 dims"
        attn_bias_batch_dim = True
        q = q[:, None]
    if k.ndim == 3:
        k = k[:, None]
    if v.ndim == 3:
        v = v[:, None]
    if attn_bias is not None:
        # fcsa.py:105-107: bias is [heads, i, j] (broadcast over batch) or, with
        # attn_bias_batch_dim, [batch, i, j] (broadcast over heads)
        attn_bias = attn_bias[:, None] if attn_bias_batch_dim else attn_bias[None]
    return q, k, v, attn_bias, merged


def _valid_mask(n, m, causal, mask):
    """valid(i, j) of SURVEY §0.1 (cu:1208-1211, fcsa.py:112-118): [B or 1, 1, n, m] bool."""
    valid = np.ones((1, 1, n, m), dtype=bool)
    if causal:
        i = np.arange(n)[:, None]
        j = np.arange(m)[None, :]
        # fcsa.py:114: triu(j - i + 1) is masked out  <=>  keep  j - (m - n) <= i
        valid = valid & ((j - (m - n)) <= i)[None, None]
    if mask is not None:
        valid = valid & mask.astype(bool)[:, None, None, :]
    return valid


# ----------------------------------------------------------------------------
# plain O(N*M) attention  (fcsa.py:75-126)  -- THE parity oracle
# ----------------------------------------------------------------------------

def plain_attention(q, k, v, mask=None, attn_bias=None, scale=8, groups=1, causal=False,
                    l2norm_qk=True, attn_bias_batch_dim=False, dtype=np.float64):
    """softmax(scale * l2norm(q) l2norm(k)^T + bias, masked with -finfo.max) @ v.

    Follows fcsa.py:75-126 line by line, evaluated in `dtype` (float64 default).
    Rows with no valid key come out as the uniform average of v (softmax over an
    all -max row), exactly as in the reference.
    """
    assert not (causal and mask is not None), "mask should not be supplied if causality is needed"
    q, k, v = (np.asarray(t, dtype=dtype) for t in (q, k, v))
    if attn_bias is not None:
        attn_bias = np.asarray(attn_bias, dtype=dtype)
    q, k, v, attn_bias, merged = _canon(q, k, v, attn_bias, attn_bias_batch_dim)
    if l2norm_qk:
        q, k = l2norm(q, groups), l2norm(k, groups)
    sim = np.einsum("bhid,bhjd->bhij", q, np.broadcast_to(k, (k.shape[0], q.shape[1]) + k.shape[2:])) * scale
    if attn_bias is not None:
        sim = sim + attn_bias
    n, m = sim.shape[-2:]
    mask_value = -np.finfo(dtype).max
    if causal:
        sim = np.where(_valid_mask(n, m, True, None), sim, mask_value)
    if mask is not None:
        sim = np.where(np.asarray(mask, dtype=bool)[:, None, None, :], sim, mask_value)
    sim = sim - sim.max(-1, keepdims=True)
    e = np.exp(sim)
    attn = e / e.sum(-1, keepdims=True)
    out = np.einsum("bhij,bhjd->bhid", attn, np.broadcast_to(v, (v.shape[0], q.shape[1]) + v.shape[2:]))
    return out[:, 0] if merged else out


# ----------------------------------------------------------------------------
# blockwise constant-shift form  (fcsa.py:130-241; kernel cu:1072-1247)
# ----------------------------------------------------------------------------

def attention_forward_stats(q, k, v, mask=None, attn_bias=None, scale=8, groups=1, causal=False,
                            l2norm_qk=True, attn_bias_batch_dim=False, dtype=np.float64, eps=1e-10, operand_dtype=None):
    """Un-tiled statement of the kernel math: returns (o, inv_l) with
    P~ = valid ? exp(S - scale) : 0 (cu:1216), l = rowsum(P~), inv_l = 1/max(l, eps)
    (cu:1236-1242, eps = 1e-10 cu:83), o = inv_l * P~ V (cu:1244).
    Rows with no valid key give o = 0 (kernel behaviour), unlike `plain_attention`.
    operand_dtype ("bf16" | "f16"): exact arithmetic on the 16-bit operands of the S product (`rounded_operands`).
    """
    assert not (causal and mask is not None)
    q, k, v = (np.asarray(t, dtype=dtype) for t in (q, k, v))
    if attn_bias is not None:
        attn_bias = np.asarray(attn_bias, dtype=dtype)
    q, k, v, attn_bias, merged = _canon(q, k, v, attn_bias, attn_bias_batch_dim)
    if l2norm_qk:
        q, k = l2norm(q, groups), l2norm(k, groups)
    q, k = rounded_operands(q, k, scale, operand_dtype)
    h = q.shape[1]
    kb = np.broadcast_to(k, (k.shape[0], h) + k.shape[2:])
    vb = np.broadcast_to(v, (v.shape[0], h) + v.shape[2:])
    s = np.einsum("bhid,bhjd->bhij", q, kb) * scale
    if attn_bias is not None:
        s = s + attn_bias
    n, m = s.shape[-2:]
    valid = _valid_mask(n, m, causal, mask)
    p = np.where(valid, np.exp(s - scale), 0.0)
    l = p.sum(-1)
    inv_l = 1.0 / np.maximum(l, eps)
    o = np.einsum("bhij,bhjd->bhid", p, vb) * inv_l[..., None]
    if merged:
        return o[:, 0], inv_l[:, 0]
    return o, inv_l


def tiled_attention(q, k, v, mask=None, attn_bias=None, scale=8, causal=False,
                    attn_bias_batch_dim=False, row_tile=512, col_tile=512, dtype=np.float64):
    """The reference's tiled CPU forward, fcsa.py:130-241, restated tile for tile.

    Expects ALREADY normalised q, k (the reference normalises in the wrapper,
    fcsa.py:320-321, before dispatching here).  Accumulates un-normalised
    (o, l) over column tiles with the constant shift exp(w - scale) (fcsa.py:230)
    and a causal tile skip (fcsa.py:215-216); o /= clamp(l, 1e-12) (fcsa.py:240).
    """
    assert not (causal and mask is not None)
    q, k, v = (np.asarray(t, dtype=dtype) for t in (q, k, v))
    if attn_bias is not None:
        attn_bias = np.asarray(attn_bias, dtype=dtype)
    shape = q.shape
    q, k, v, attn_bias, merged = _canon(q, k, v, attn_bias, attn_bias_batch_dim)
    b, h, n, d = q.shape
    m = k.shape[2]
    diff = m - n                                    # fcsa.py:153
    kb = np.broadcast_to(k, (b, h, m, d))
    vb = np.broadcast_to(v, (b, h, m, d))
    o = np.zeros_like(q)
    l = np.zeros(q.shape[:-1] + (1,), dtype=dtype)
    for r0 in range(0, n, row_tile):
        r1 = min(r0 + row_tile, n)
        q_start = r0 + diff                          # fcsa.py:197
        for c0 in range(0, m, col_tile):
            c1 = min(c0 + col_tile, m)
            # KNOWN REFERENCE DEFECT (not restated): fcsa.py:215 reads
            #   `if causal and q_start_index >= (k_start_index + col_tile_size - 1): continue`
            # which skips tiles that lie entirely BELOW the diagonal (all pairs valid) and
            # never skips tiles above it.  With one 512x512 tile (every reference test,
            # N <= 127) the branch is never taken; for causal N > 512 the reference's tiled
            # CPU path disagrees with its own plain_cosine_sim_attention by O(1) (measured
            # in the build container: N=513 -> max|diff| 2.0).  The CUDA kernel has the
            # intended skip (cu:1178-1179).  This oracle implements the intended skip, so it
            # equals plain_cosine_sim_attention for every N.
            if causal and (r1 - 1 + diff) < c0:              # tile entirely above the diagonal
                continue
            w = np.einsum("bhid,bhjd->bhij", q[:, :, r0:r1], kb[:, :, c0:c1]) * scale
            if attn_bias is not None:
                w = w + attn_bias[:, :, r0:r1, c0:c1]
            keep = np.ones((1, 1, r1 - r0, c1 - c0), dtype=bool)
            if mask is not None:
                keep = keep & np.asarray(mask, dtype=bool)[:, None, None, c0:c1]
            if causal:
                i = np.arange(r0, r1)[:, None] + diff
                j = np.arange(c0, c1)[None, :]
                keep = keep & (j <= i)[None, None]
            e = np.where(keep, np.exp(w - scale), 0.0)       # fcsa.py:230-233
            o[:, :, r0:r1] += np.einsum("bhij,bhjd->bhid", e, vb[:, :, c0:c1])
            l[:, :, r0:r1] += e.sum(-1, keepdims=True)
    o = o / np.maximum(l, 1e-12)
    return o.reshape(shape)


# ----------------------------------------------------------------------------
# backward  (cu:1256-1626)
# ----------------------------------------------------------------------------

def attention_backward(do, q, k, v, mask=None, attn_bias=None, scale=8, groups=1, causal=False,
                       l2norm_qk=True, attn_bias_batch_dim=False, dtype=np.float64, eps=1e-10, operand_dtype=None, o_saved=None):
    """Analytic gradients (dq, dk, dv, d_bias) w.r.t. the RAW q, k, v, bias.

    Kernel math (w.r.t. the normalised qh, kh):
        delta = rowsum(dO * O)                     cu:1256-1335
        P  = P~ * inv_l                            cu:1525
        dV = P^T dO                                cu:1534-1540 (summed over heads for single-head kv, cu:1613)
        dP = dO V^T                                cu:1544-1553
        dS = P * (dP - delta)                      cu:1564-1570  (= d_bias, summed over b or h: cu:1474, 1574)
        dQh = scale * dS Kh ; dKh = scale * dS^T Qh   cu:1580-1610
    then chained through `l2norm_backward` when l2norm_qk (autograd does this in
    the reference because l2norm sits outside the Function, fcsa.py:320-321).
    operand_dtype ("bf16" | "f16"): the same formulas in exact arithmetic on the 16-bit normalised operands (`rounded_operands`),
    which also stand in for x / n in the l2norm backward -- the gradient a 16-bit implementation computes, minus its own roundings.
    """
    assert not (causal and mask is not None)
    q0, k0, v0 = (np.asarray(t, dtype=dtype) for t in (q, k, v))
    do = np.asarray(do, dtype=dtype)
    bias0 = None if attn_bias is None else np.asarray(attn_bias, dtype=dtype)
    qc, kc, vc, bias, merged = _canon(q0, k0, v0, bias0, attn_bias_batch_dim)
    if merged:
        do = do[:, None]
        attn_bias_batch_dim = True
    qh, kh = (l2norm(qc, groups), l2norm(kc, groups)) if l2norm_qk else (qc, kc)
    qh, kh = rounded_operands(qh, kh, scale, operand_dtype)
    faithful = operand_dtype in ("bf16", "f16")
    b, h, n, d = qh.shape
    m = kh.shape[2]
    hk = kh.shape[1]
    kb = np.broadcast_to(kh, (b, h, m, d))
    vb = np.broadcast_to(vc, (b, h, m, d))
    s = np.einsum("bhid,bhjd->bhij", qh, kb) * scale
    if bias is not None:
        s = s + bias
    valid = _valid_mask(n, m, causal, mask)
    pt = np.where(valid, np.exp(s - scale), 0.0)
    l = pt.sum(-1)
    inv_l = 1.0 / np.maximum(l, eps)
    p = pt * inv_l[..., None]
    o = np.einsum("bhij,bhjd->bhid", p, vb)
    # o_saved: `o` is an INPUT of the backward (reference backward(..., o, do, ...): delta is taken from the stored output,
    # backward_preprocess cu:1256-1335).  Given, delta uses it instead of the exact output: the gradient a backward pass computes from
    # THAT saved tensor.  It matters where a row's weight sits on one or two keys: dP - delta cancels, and the rounding of a 16-bit `o`
    # is what is left -- 0.5 ... 8 % of dq / dk / d_bias on a single-row problem (exploratory fuzz seed 5, case X11 of tests/test_gpu_fuzz.py),
    # the same in every implementation that keeps `o` in 16 bits.
    o_used = o if o_saved is None else np.asarray(o_saved, dtype=dtype).reshape(o.shape)
    delta = (do * o_used).sum(-1)
    dv = np.einsum("bhij,bhid->bhjd", p, do)
    dp = np.einsum("bhid,bhjd->bhij", do, vb)
    ds = p * (dp - delta[..., None])
    dqh = scale * np.einsum("bhij,bhjd->bhid", ds, kb)
    dkh = scale * np.einsum("bhij,bhid->bhjd", ds, qh)
    if hk == 1 and h > 1:
        dv = dv.sum(1, keepdims=True)
        dkh = dkh.sum(1, keepdims=True)
    dbias = None
    if bias is not None:
        dbias = ds.sum(1) if attn_bias_batch_dim else ds.sum(0)
    dq = l2norm_backward(dqh, qc, groups, xh_used=qh if faithful else None) if l2norm_qk else dqh
    dk = l2norm_backward(dkh, kc, groups, xh_used=kh if faithful else None) if l2norm_qk else dkh
    dq = dq.reshape(q0.shape)
    dk = dk.reshape(k0.shape)
    dv = dv.reshape(v0.shape)
    return dq, dk, dv, dbias


# ----------------------------------------------------------------------------
# working-precision model of the gfx950 kernels (what a CORRECT float32-accumulating implementation with their rounding points returns)
# ----------------------------------------------------------------------------

def _f32(x):
    return np.asarray(x, dtype=np.float32)


def _seq_matmul_f32(a, b, chunk=16):
    """a[..., i, c] @ b[..., c, j] in float32 with the contraction index consumed SEQUENTIALLY in chunks of `chunk` (one MFMA k-step:
    products of a chunk summed, then added to the float32 accumulator) -- no pairwise / blocked re-association of the long sum."""
    a, b = _f32(a), _f32(b)
    n = a.shape[-1]
    acc = np.zeros(np.broadcast_shapes(a.shape[:-2], b.shape[:-2]) + (a.shape[-2], b.shape[-1]), dtype=np.float32)
    for c0 in range(0, n, chunk):
        acc = (acc + np.matmul(a[..., :, c0:c0 + chunk], b[..., c0:c0 + chunk, :])).astype(np.float32)
    return acc


def attention_backward_emulated(do, q, k, v, storage, mask=None, attn_bias=None, scale=8, groups=1, causal=False,
                                l2norm_qk=True, attn_bias_batch_dim=False, o_saved=None):
    """(o, dq, dk, dv, d_bias) as a float32-ACCUMULATING implementation with the rounding points of the gfx950 kernels computes them
    (DESIGN.md sections 2, 4.1): inputs, the saved c1 * q^ / k^ and every output in `storage` ("f32" | "f16" | "bf16"); S, P~, row
    sums, delta, dP, dS, every accumulator and the l2norm backward in float32, long sums taken sequentially (`_seq_matmul_f32`); P and dS
    rounded to `storage` in front of the second products (16-bit types).  This is NOT a checker of the kernels' values -- their
    summation order differs -- but of the SIZE of the error such arithmetic leaves on a given problem:
        err_model = || emulated - float64 || / || float64 ||
    is what `tests/test_gpu_fuzz.py` derives the allowance of its ill-conditioned classes from (few query rows: dq / dk / d_bias are
    one row's cancellation residue of dP - delta; a handful of keys under many rows: dk sums every row's rounding of dP - delta), instead
    of hand-set factors (round-5 review: "verify the allowances instead of explaining them").  Reference formulas: cu:1256-1626.
    o_saved: the stored output the backward is GIVEN (`o` is an input of the reference's backward, cu:1752-1764; delta = rowsum(dO * o) is
    taken from it, cu:1256-1335).  Where dP - delta cancels, WHICH way the 16-bit elements of `o` were rounded decides the residue: a
    one-row problem measured 3 - 5e-2 between two correct implementations whose stored outputs differ by one bf16 ulp in a few elements
    (profiles/r06_fuzz_model_probe.txt).  Given, the model's backward uses it, like `attention_backward` does; the returned o is the model's own.
    """
    assert not (causal and mask is not None)
    st = storage
    rnd = (lambda x: _f32(x)) if st == "f32" else (lambda x: round_to(x, st).astype(np.float32))
    q0, k0, v0, do0 = (_f32(round_to(t, st)) for t in (q, k, v, do))
    bias0 = None if attn_bias is None else _f32(round_to(attn_bias, st))
    qc, kc, vc, bias, merged = _canon(q0, k0, v0, bias0, attn_bias_batch_dim)
    if merged:
        do0 = do0[:, None]
        attn_bias_batch_dim = True
    b, h, n, d = qc.shape
    m, hk = kc.shape[2], kc.shape[1]
    c1 = np.float32(scale * LOG2E)

    def norm32(x):      # grouped l2norm in float32: x / max(||x_g||, 1e-12), inverse norms
        xg = x.reshape(*x.shape[:-1], groups, d // groups)
        ss = np.zeros(xg.shape[:-1], dtype=np.float32)
        for e in range(xg.shape[-1]):
            ss = (ss + xg[..., e] * xg[..., e]).astype(np.float32)
        r = (np.float32(1.0) / np.maximum(np.sqrt(ss), np.float32(1e-12))).astype(np.float32)
        return (xg * r[..., None]).reshape(x.shape).astype(np.float32), r

    if l2norm_qk:
        qh, rq = norm32(qc)
        kh, rk = norm32(kc)
    else:
        qh, kh, rq, rk = qc, kc, None, None
    qn = rnd(qh * c1)                         # the saved c1 * q^ (one rounding), B operand of the S chains
    kn = rnd(kh)
    kb = np.broadcast_to(kn, (b, h, m, d))
    vb = np.broadcast_to(vc, (b, h, m, d))
    s2 = _seq_matmul_f32(qn, np.swapaxes(kb, -1, -2))                       # log2 units
    if bias is not None:
        s2 = (s2 + bias * np.float32(LOG2E)).astype(np.float32)
    valid = np.broadcast_to(_valid_mask(n, m, causal, mask), s2.shape)
    # forward: per-row reference = the row max (any shift gives the same o; the kernels use a constant or an online one)
    ref = np.where(valid, s2, -np.inf).max(-1, keepdims=True)
    ref = np.where(np.isfinite(ref), ref, 0).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        pt = np.where(valid, np.exp2((s2 - ref).astype(np.float32)), 0).astype(np.float32)
    pt16 = rnd(pt)
    l = pt16.sum(-1, dtype=np.float64).astype(np.float32)                   # exact f32 sum of the ROUNDED values (v_dot2c)
    inv_l = (np.float32(1.0) / np.maximum(l, np.float32(1e-30))).astype(np.float32)
    o = rnd(_seq_matmul_f32(pt16, vb) * inv_l[..., None])
    lc = (np.log2(inv_l) - ref[..., 0]).astype(np.float32)                  # seed of the backward's S accumulators
    # backward
    o_used = o if o_saved is None else _f32(round_to(o_saved, st)).reshape(o.shape)
    delta = np.zeros((b, h, n), dtype=np.float32)
    for e in range(d):
        delta = (delta + do0[..., e] * o_used[..., e]).astype(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        p = np.where(valid, np.exp2((s2 + lc[..., None]).astype(np.float32)), 0).astype(np.float32)
    dp = (_seq_matmul_f32(do0, np.swapaxes(vb, -1, -2)) - delta[..., None]).astype(np.float32)
    ds = (p * dp).astype(np.float32)
    p16, ds16 = rnd(p), rnd(ds)
    dv = _seq_matmul_f32(np.swapaxes(p16, -1, -2), do0)
    dqh = (np.float32(scale) * _seq_matmul_f32(ds16, kb)).astype(np.float32)
    dkh = (np.float32(scale / float(c1)) * _seq_matmul_f32(np.swapaxes(ds16, -1, -2), qn)).astype(np.float32)
    if hk == 1 and h > 1:
        dv = dv.sum(1, keepdims=True, dtype=np.float32)
        dkh = dkh.sum(1, keepdims=True, dtype=np.float32)
    dbias = None
    if bias is not None:
        dbias = rnd(ds.sum(1 if attn_bias_batch_dim else 0, dtype=np.float32))

    def norm_bwd32(g, xhat, r):          # dx = r (g - x^ <g, x^>) per group, float32; x^ = the stored normalised rows
        gg = g.reshape(*g.shape[:-1], groups, d // groups)
        xg = xhat.reshape(gg.shape)
        dot = np.zeros(gg.shape[:-1], dtype=np.float32)
        for e in range(gg.shape[-1]):
            dot = (dot + gg[..., e] * xg[..., e]).astype(np.float32)
        out = (r[..., None] * (gg - xg * dot[..., None])).astype(np.float32)
        out = np.where((r >= np.float32(1e12))[..., None], gg * r[..., None], out)
        return out.reshape(g.shape)

    if l2norm_qk:
        dq = norm_bwd32(dqh, (qn / c1).astype(np.float32), rq)
        dk = norm_bwd32(dkh, kn, rk)
    else:
        dq, dk = dqh, dkh
    to64 = lambda x, ref_shape: rnd(x).astype(np.float64).reshape(ref_shape)
    return (to64(o, np.asarray(q).shape), to64(dq, np.asarray(q).shape), to64(dk, np.asarray(k).shape), to64(dv, np.asarray(v).shape),
            None if dbias is None else dbias.astype(np.float64))


# ----------------------------------------------------------------------------
# FLOP convention (SURVEY §8d / BASELINE.md §2)
# ----------------------------------------------------------------------------

def causal_valid_count(n: int, m: int) -> int:
    """n_valid = sum_i min(m, max(0, i + (m - n) + 1))."""
    diff = m - n
    return int(sum(min(m, max(0, i + diff + 1)) for i in range(n)))


def algorithmic_flops(b, h, n, m, d, causal=False, fwd=True, bwd=True) -> float:
    """GEMM FLOPs only, 2 per MAC: fwd 4*BHNMD, bwd 10*BHNMD; causal scaled by n_valid/(N*M)."""
    per = (4 if fwd else 0) + (10 if bwd else 0)
    f = float(per) * b * h * n * m * d
    if causal:
        f *= causal_valid_count(n, m) / float(n * m)
    return f
