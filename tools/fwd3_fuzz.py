"""Differential fuzz of the D = 128 wide forward (csrc/fcsa_fwd3.hip) against the lean 32-row form it replaces: random shapes that the
dispatch sends to it (16-bit, D = 128, no bias / mask, grid >= 7/8 of the CUs), random causal / N != M / ragged sizes / single-headed K/V /
groups / scale / l2norm_qk / strided views / inference vs training, both forms run in ONE process (fcsa_debug_forward_form, include/fcsa.h, switches between launches).
The outputs may differ by the rounding of the 16-bit output (the forms sum rows and P~ in different orders): any element further apart
than one output ulp, any non-finite value and any inv_l further apart than 1e-5 relative is reported with the configuration.
usage: python tools/fwd3_fuzz.py [--seed S] [--n 200]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def draw(rs):
    causal = bool(rs.randint(0, 2))
    n = int(rs.choice([1, 63, 64, 65, 127, 128, 200, 255, 256, 257, 300, 511, 512, 513, 700, 1000, 1025]))
    m = int(rs.choice([1, 2, 33, 63, 64, 65, 127, 128, 129, 191, 192, 255, 256, 300, 448, 460, 512, 777, 1030]))
    if rs.rand() < 0.4:
        m = n
    mt = (n + 255) // 256
    pt = (mt + 1) // 2 if causal else mt
    need = (224 + pt - 1) // pt                     # batch * heads so that the grid covers the chip
    h = int(rs.choice([1, 2, 3, 4, 8, 16, 28]))
    b = (need + h - 1) // h + int(rs.randint(0, 2))
    groups = int(rs.choice([1, 1, 1, 2, 4, 8, 16]))
    scale = float(rs.choice([8.0, 8.0, 1.0, 4.0, -8.0, 2.5]))
    if abs(scale) * groups > 40:
        scale = 40.0 / groups * (1 if scale > 0 else -1)
    return dict(B=b, H=h, N=n, M=m, causal=causal, groups=groups, scale=scale, dtype=str(rs.choice(["bf16", "f16"])),
                single_kv=bool(rs.rand() < 0.25), l2norm=bool(rs.rand() < 0.85), strided=bool(rs.rand() < 0.3), grad=bool(rs.rand() < 0.5))


def run(cfg, seed):
    import flash_cosine_sim_attention_amd as F
    from flash_cosine_sim_attention_amd import _lib
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[cfg["dtype"]]
    g = torch.Generator(device="cuda").manual_seed(seed)
    B, H, N, M, D = cfg["B"], cfg["H"], cfg["N"], cfg["M"], 128
    if cfg["strided"] and not cfg["single_kv"] and N == M:
        qkv = torch.randn((B, N, 3, H, D), device="cuda", dtype=dt, generator=g)
        q, k, v = (t.transpose(1, 2) for t in qkv.unbind(2))
    else:
        q = torch.randn((B, H, N, D), device="cuda", dtype=dt, generator=g)
        ks = (B, M, D) if cfg["single_kv"] else (B, H, M, D)
        k = torch.randn(ks, device="cuda", dtype=dt, generator=g)
        v = torch.randn(ks, device="cuda", dtype=dt, generator=g)
    groups, scale = cfg["groups"], cfg["scale"]
    if not cfg["l2norm"]:      # the reference extension's contract: q, k as given, exponent shift = scale (a negative scale overflows f16 there, by contract)
        q = torch.nn.functional.normalize(q.float(), dim=-1).to(dt)
        k = torch.nn.functional.normalize(k.float(), dim=-1).to(dt)
        groups, scale = 1, abs(scale)
    kw = dict(scale=scale, groups=groups, causal=cfg["causal"], l2norm_qk=cfg["l2norm"])
    outs = []
    for form in ("on", "0"):
        _lib.forward_form(0 if form == "0" else 1)
        if cfg["grad"]:
            qq, kk, vv = (t.detach().clone().requires_grad_() for t in (q, k, v))
            o = F.flash_cosine_sim_attention(qq, kk, vv, **kw)
            o.backward(torch.ones_like(o))
            outs.append((o.detach(), qq.grad, kk.grad, vv.grad))
        else:
            with torch.no_grad():
                outs.append((F.flash_cosine_sim_attention(q, k, v, **kw),))
    _lib.forward_form(1)
    torch.cuda.synchronize()
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    new, old = outs
    bad = []
    if not all(torch.isfinite(t).all() for t in new):
        bad.append("non-finite")
    d = (new[0].float() - old[0].float()).abs()
    # one ulp of the VALUE = ulp * |old|: between one and two spacings of the 16-bit grid (two at the top of a binade -- round 6, seed 7: one
    # element of 7.4 M exactly two spacings apart at |old| = 0.957, 1.001 of the relative bar), so the bar is two spacings of |old|'s binade
    spacing = torch.exp2(torch.floor(torch.log2(old[0].float().abs().clamp_min(2.0 ** -14)))) * ulp
    lim = torch.maximum(1.01 * ulp * old[0].float().abs(), 2.0 * spacing) + 4 * ulp * 2.0 ** -7
    if not (d <= lim).all():
        w = (d / lim).argmax()      # the worst VIOLATOR (largest |new - old| relative to its own bar), not the largest difference
        bad.append("o: |new - old| %.3e at |old| %.3e (%.2f of the bar there; %d of %d elements over; largest difference %.3e)"
                   % (d.flatten()[w].item(), old[0].float().abs().flatten()[w].item(), (d / lim).flatten()[w].item(), int((d > lim).sum().item()), d.numel(), d.max().item()))
    # Every row sees at most two keys (M <= 2): P is 1 or nearly so, the exact dS is (nearly) 0 and what a kernel returns for dq / dk is the
    # cancellation residue of dP - delta.  The lean form sums the ROUNDED P~, so with one key o == v bit for bit and its residue is f32
    # noise; the wide form sums the un-rounded P~ (like the standard flash-attention forward), o = v (1 +- 2^-9) before rounding, and its
    # residue is a 2^-9 fraction of dP: tiny against any real gradient, large against zero.  Not comparable form against form; the parity
    # suite covers such rows against the oracle with the suite's bars (e.g. causal_n_1, full_m_33 in tests/test_gpu_wide128.py).
    few_keys = cfg["M"] <= 2
    for name, a, b2 in zip(("dq", "dk", "dv"), new[1:] if not few_keys else (), old[1:] if not few_keys else ()):
        # the backward consumes o and inv_l of the forward form, so the two runs differ by what a one-ulp change of o does to delta: both
        # are within the suite's gradient bar of exact math, i.e. within twice that bar of each other; the floor is the suite's (where
        # P == 1 -- one visible key -- dS is exactly 0 and both forms return cancellation noise)
        floor = 1e-3 * (b2.numel() ** 0.5)
        rel = ((a.float() - b2.float()).norm() / b2.float().norm().clamp_min(floor)).item()
        bar = 2 * (8.5e-3 if dt == torch.bfloat16 else 1.4e-3)
        if not rel <= bar:
            bad.append("%s rel-L2 between the forms %.3e > %.1e" % (name, rel, bar))
    return bad


def main():
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--n", type=int, default=200)
    ap.add_argument("--only", type=int, default=-1, help="run only configuration number ONLY of the seed's sequence")
    args = ap.parse_args()
    rs = np.random.RandomState(args.seed)
    fails = 0
    for i in range(args.n):
        cfg = draw(rs)
        if args.only >= 0 and i != args.only:
            continue
        bad = run(cfg, 1000 * args.seed + i)
        if bad:
            fails += 1
            print("FAIL #%d" % i, cfg, bad)
    print(f"seed {args.seed}: {args.n} configurations, {fails} failures")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
