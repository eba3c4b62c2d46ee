#!/usr/bin/env python3
"""One-shot GPU diagnostics: structured inputs that localise layout / indexing errors in the kernels.
Writes a text report to gpurun_out/diag.txt (and stdout).  Test infrastructure (uses the oracle)."""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F          # noqa: E402
from oracle import cosine_sim_oracle as O           # noqa: E402

OUT = os.path.join(ROOT, "gpurun_out")
os.makedirs(OUT, exist_ok=True)
lines = []


def P(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    lines.append(s)


def npf(t):
    return t.detach().cpu().double().numpy()


def errmap(name, got, ref, axis_names=("row", "col")):
    d = np.abs(got - ref)
    P(f"  {name}: max-abs {d.max():.3e}  rel-L2 {np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30):.3e}  finite={np.isfinite(got).all()}")
    if d.max() > 5e-2:
        flat = d.reshape(-1, d.shape[-2], d.shape[-1]).max(0)
        bad_r = np.where(flat.max(1) > 5e-2)[0]
        bad_c = np.where(flat.max(0) > 5e-2)[0]
        P(f"    bad {axis_names[0]}s ({len(bad_r)}/{flat.shape[0]}): {bad_r[:40].tolist()}")
        P(f"    bad {axis_names[1]}s ({len(bad_c)}/{flat.shape[1]}): {bad_c[:40].tolist()}")


def run(dtype, b, h, n, m, d, causal=False, grads=True, v_mode="rand", seed=0, l2norm=True):
    torch.manual_seed(seed)
    q = torch.randn(b, h, n, d, device="cuda", dtype=dtype)
    k = torch.randn(b, h, m, d, device="cuda", dtype=dtype)
    if v_mode == "ones":
        v = torch.ones(b, h, m, d, device="cuda", dtype=dtype)
    elif v_mode == "onehot":       # V[j][dd] = (j % d == dd): O[i][dd] = sum of P over keys j == dd (mod d)
        v = torch.zeros(b, h, m, d, device="cuda", dtype=dtype)
        idx = torch.arange(m, device="cuda")
        v[:, :, idx, idx % d] = 1
    else:
        v = torch.randn(b, h, m, d, device="cuda", dtype=dtype)
    q.requires_grad_(grads); k.requires_grad_(grads); v.requires_grad_(grads)
    P(f"case dtype={dtype} B{b} H{h} N{n} M{m} D{d} causal={causal} v={v_mode} l2norm={l2norm}")
    try:
        o = F.flash_cosine_sim_attention(q, k, v, causal=causal, l2norm_qk=l2norm, scale=8 if l2norm else 0.125)
        torch.cuda.synchronize()
        kw = dict(causal=causal, l2norm_qk=l2norm, scale=8 if l2norm else 0.125)
        ro, _ = O.attention_forward_stats(npf(q), npf(k), npf(v), **kw)
        errmap("o ", npf(o), ro, ("query", "feature"))
        if grads:
            do = torch.randn_like(o)
            o.backward(do)
            torch.cuda.synchronize()
            rdq, rdk, rdv, _ = O.attention_backward(npf(do), npf(q), npf(k), npf(v), **kw)
            errmap("dq", npf(q.grad), rdq, ("query", "feature"))
            errmap("dk", npf(k.grad), rdk, ("key", "feature"))
            errmap("dv", npf(v.grad), rdv, ("key", "feature"))
    except Exception:
        P("  EXCEPTION:\n" + traceback.format_exc())


def norm_check(dtype, d, groups=1):
    from flash_cosine_sim_attention_amd import ext
    torch.manual_seed(5)
    x = torch.randn(2, 3, 37, d, device="cuda", dtype=dtype)
    try:
        got = ext.l2norm_device(x, groups)
        ref = O.l2norm(npf(x), groups)
        P(f"l2norm {dtype} D{d} G{groups}: max-abs {np.abs(npf(got) - ref).max():.3e}")
    except Exception:
        P("  l2norm EXCEPTION:\n" + traceback.format_exc())


def main():
    P("device:", torch.cuda.get_device_name(0), "| torch", torch.__version__)
    P(F.debug())
    bf, hf = torch.bfloat16, torch.float16
    if os.environ.get("FCSA_DIAG", "") == "f32":
        f32 = torch.float32
        for d in (16, 32, 64, 128):
            norm_check(f32, d)
        norm_check(bf, 64)
        run(f32, 1, 1, 32, 32, 64, grads=False, v_mode="ones", l2norm=False)
        run(f32, 1, 1, 32, 32, 64, grads=False, v_mode="onehot", l2norm=False)
        run(f32, 1, 1, 32, 32, 64, grads=False, l2norm=False)
        run(f32, 1, 1, 32, 32, 64, grads=False)
        run(f32, 1, 1, 32, 32, 32, grads=False, l2norm=False)
        run(f32, 1, 1, 32, 32, 16, grads=False, l2norm=False)
        run(f32, 1, 1, 128, 256, 64, grads=False)
        run(f32, 1, 1, 32, 32, 64, l2norm=False)
        run(f32, 1, 2, 128, 256, 64)
        for d in (16, 32, 96, 128):
            run(f32, 1, 2, 100, 130, d, causal=True)
        open(os.path.join(OUT, "diag_f32.txt"), "w").write("\n".join(lines) + "\n")
        return
    # single 32x32 block, then one tile, then multi-tile; forward only first
    run(bf, 1, 1, 32, 32, 64, grads=False, v_mode="ones")
    run(bf, 1, 1, 32, 32, 64, grads=False, v_mode="onehot")
    run(bf, 1, 1, 32, 64, 64, grads=False, v_mode="onehot")
    run(bf, 1, 1, 32, 32, 64, grads=False)
    run(bf, 1, 1, 128, 64, 64, grads=False)
    run(bf, 1, 1, 128, 256, 64, grads=False)
    run(bf, 1, 1, 128, 256, 64, grads=False, l2norm=False)
    for d in (16, 32, 96, 128):
        run(bf, 1, 1, 64, 128, d, grads=False)
    run(hf, 1, 1, 128, 256, 64, grads=False)
    # backward
    run(bf, 1, 1, 32, 32, 64, l2norm=False)
    run(bf, 1, 1, 32, 32, 64)
    run(bf, 1, 2, 128, 256, 64)
    run(bf, 2, 2, 200, 200, 64, causal=True)
    for d in (16, 32, 96, 128):
        run(hf, 1, 2, 100, 130, d, causal=False)
    open(os.path.join(OUT, "diag.txt"), "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
