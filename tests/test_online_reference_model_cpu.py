"""CPU model of the forward kernel's online per-row exponent reference (fcsa_fwd.hip online_recentre), in numpy with P~ rounded to
float16 / bfloat16 the way the kernel rounds it.  It pins the ALGORITHM's invariants independently of the GPU:

  * whatever the logits do along the key axis (ramps of hundreds of log2 units, a -inf prefix, a first tile far below the rest),
    no rounded P~ overflows and the result equals softmax(S) V to the rounding of P~;
  * a row ends with its max inside [0, tau] of its reference (tau = 10 for float16, 64 otherwise);
  * testing a whole 64-key tile at once (prefetching 16-bit form: block 0 is exponentiated against the reference the tile started
    with and recomputed when the test moves it) gives the same result as testing every 32-key block (lean / bias / f32 forms).

The GPU tests (test_gpu_online_reference.py, test_gpu_fuzz.py) check the kernels themselves against the float64 oracle; this file
runs on the CPU and uses nothing from oracle/.
"""
import numpy as np
import pytest
import torch


def _round(x, kind):
    if kind == "f32":
        return x.astype(np.float32)
    dt = torch.float16 if kind == "f16" else torch.bfloat16
    return torch.from_numpy(x.astype(np.float32)).to(dt).float().numpy()


def online_rows(s2, v, kind, block):
    """s2: [N, M] logits in log2 units (-inf = masked), v: [M, D].  Returns (o, reference m, running max relative to m, max |P~|)."""
    tau = 10.0 if kind == "f16" else 64.0
    n, m_keys = s2.shape
    o = np.zeros((n, v.shape[1]), np.float64)
    l = np.zeros(n, np.float64)
    ref = np.zeros(n, np.float64)
    rmax = np.full(n, -np.inf)
    pmax = 0.0
    for j0 in range(0, m_keys, block):
        x = s2[:, j0:j0 + block] - ref[:, None]                  # the S accumulators start from -reference
        bm = x.max(axis=1)
        first = np.isinf(rmax) & (rmax < 0) & np.isfinite(bm)
        move = first | (bm > tau)
        d = np.where(move, bm, 0.0)
        f = np.where(first, 1.0, np.exp2(-d))                     # nothing accumulated before the first valid key
        x = x - d[:, None]
        o *= f[:, None]
        l *= f
        ref += d
        rmax = np.where(first, 0.0, rmax - d)
        rmax = np.maximum(rmax, bm - d)
        with np.errstate(over="ignore"):
            p = _round(np.exp2(x.astype(np.float32)), kind).astype(np.float64)
        assert np.isfinite(p).all(), "a rounded P~ overflowed"
        pmax = max(pmax, float(p.max()))
        o += p @ v[j0:j0 + block]
        l += p.sum(axis=1)                                         # the row sum is taken from the ROUNDED P~ (ones-MFMA)
    valid = l > 0
    out = np.zeros_like(o)
    out[valid] = o[valid] / l[valid, None]
    return out, ref, rmax, pmax


def exact(s2, v):
    out = np.zeros((s2.shape[0], v.shape[1]))
    for i in range(s2.shape[0]):
        row = s2[i]
        ok = np.isfinite(row)
        if ok.any():
            w = np.exp2(row[ok] - row[ok].max())
            out[i] = (w / w.sum()) @ v[ok]
    return out


def _logits(kind_of_input, n, m, rng):
    if kind_of_input == "ramp_up":
        return np.linspace(-180.0, 180.0, m)[None, :] + rng.normal(0, 2.0, (n, m))
    if kind_of_input == "ramp_down":
        return np.linspace(180.0, -180.0, m)[None, :] + rng.normal(0, 2.0, (n, m))
    if kind_of_input == "low_first_tile":
        s = rng.normal(0, 3.0, (n, m))
        s[:, :64] -= 300.0                                         # far below anything a 16-bit P~ could hold against the later keys
        return s
    if kind_of_input == "masked_prefix":
        s = rng.normal(0, 6.0, (n, m))
        s[:, :100] = -np.inf
        s[0, :] = -np.inf                                          # a row without any valid key
        return s
    if kind_of_input == "steps":
        return np.repeat(rng.uniform(-150, 150, (n, m // 32 + 1)), 32, axis=1)[:, :m] + rng.normal(0, 1.0, (n, m))
    raise ValueError(kind_of_input)


@pytest.mark.parametrize("kind", ["f16", "bf16", "f32"])
@pytest.mark.parametrize("inp", ["ramp_up", "ramp_down", "low_first_tile", "masked_prefix", "steps"])
def test_online_reference_model(kind, inp):
    rng = np.random.RandomState(7)
    n, m, d = 24, 700, 16
    s2 = _logits(inp, n, m, rng)
    v = rng.normal(0, 1.0, (m, d))
    ref_o = exact(s2, v)
    tau = 10.0 if kind == "f16" else 64.0
    outs = []
    for block in (32, 64):                                         # per-block test (lean / bias / f32 forms), per-tile test (prefetching form)
        o, mref, rmax, pmax = online_rows(s2, v, kind, block)
        rows = np.isfinite(s2).any(axis=1)
        # rows end with their max inside [0, tau] of the reference; P~ never above 2^tau
        assert (rmax[rows] >= 0).all() and (rmax[rows] <= tau).all()
        assert pmax <= 2.0 ** tau
        assert np.allclose(mref[rows] + rmax[rows], np.where(np.isfinite(s2), s2, -np.inf).max(axis=1)[rows])
        eps = {"f16": 2.0 ** -11, "bf16": 2.0 ** -8, "f32": 2.0 ** -23}[kind]
        assert np.abs(o - ref_o).max() <= 4 * eps * np.abs(v).max(), (kind, inp, block, np.abs(o - ref_o).max())
        assert (o[~rows] == 0).all()                               # rows without a valid key: O = 0 (kernel semantics)
        outs.append(o)
    # both test granularities agree to the rounding of P~
    assert np.abs(outs[0] - outs[1]).max() <= 4 * {"f16": 2.0 ** -11, "bf16": 2.0 ** -8, "f32": 2.0 ** -23}[kind] * np.abs(v).max()
