#!/usr/bin/env python3
"""Fits the split-count cost model of fcsa_capi.hip (split_cost) to tools/split_sweep.py tables and reports, per shape, what the fitted
model picks against the measured best (the regret) and what the round-5 rule ("enough workgroups for two per CU") picked.
usage: split_model_fit.py fwd|dq|dkv table.txt [table.txt ...]        (prints the constants in the order split_cost takes them)"""
import re, sys, math
import numpy as np
from scipy.optimize import least_squares
CUS = 256

def parse(kind, files):
    """-> list of shapes: dict(B,H,N,D,M, tiles, L, rows, pts = {(s, form): us})   form: 'A' = the 8-wave one-per-CU form, 'W' = 4-wave workgroups"""
    out = []
    for fn in files:
        cur = None
        for line in open(fn):
            m = re.match(r"shape (\d+),(\d+),(\d+),(\d+),(\d+) (\w+): (.*)", line)
            if m:
                B, H, N, D, M = (int(x) for x in m.groups()[:5])
                rest = m.group(7)
                k = "fwd" if "row tiles" in rest else ("dq" if rest.startswith("dq") else "dkv")
                cur = None
                if k != kind: continue
                loop, own = (M, N) if k != "dkv" else (N, M)
                cur = dict(B=B, H=H, N=N, D=D, M=M, tiles=B * H * ((own + 127) // 128), L=loop, rows=B * H * own, pts={})
                out.append(cur)
                continue
            if cur is None: continue
            if kind == "fwd":
                m = re.match(r"(4-wave|ksplit\(8w\))\s+(.*)", line)
                if m:
                    form = "A" if m.group(1).startswith("ksplit") else "W"
                    for sp, us in re.findall(r"s(\d+):\s*([\d.]+)", m.group(2)): cur["pts"][(int(sp), form)] = float(us)
            else:
                vals = re.findall(r"s(\d+):\s*([\d.]+) \(\s*([\d.]+)\)", line)
                if vals:
                    fin1 = float(vals[0][2]) - float(vals[0][1])          # finalize launches of the step that do not belong to this kernel
                    for sp, kus, tot in vals:
                        s = int(sp)
                        cur["pts"][(s, "A" if s == 1 else "W")] = float(kus) + (float(tot) - float(kus) - fin1 if s > 1 else 0.0)
    return [x for x in out if x["pts"]]

def cost(theta, kind, sh, s, form):
    tA, cA, tB, cB, tC, cC, a0, b0, k0, k1, alpha = theta
    D = sh["D"]
    wide = D * 2 > 128
    ft = a0 + (1 - a0) * D / 64.0
    fc = b0 + (1 - b0) * max(0.44, D / 64.0)      # a key's cost: the exponentials do not shrink with D (floor at D = 28)
    tot = sh["tiles"] * s
    if form == "A": slots, t0, c = CUS, tA, cA
    elif wide or tot <= CUS: slots, t0, c = CUS, tB, cB
    else: slots, t0, c = 2 * CUS, tC, cC
    per = t0 * ft + c * fc * (sh["L"] / s) / 1024.0
    full, rem = divmod(tot, slots)
    rounds = 1.0 if full == 0 else full + (0.0 if rem == 0 else alpha + (1 - alpha) * rem / slots)
    slabs = 2 if kind == "dkv" else 1
    comb = 0.0 if s == 1 else k0 + k1 * slabs * s * sh["rows"] * (D + (1 if kind == "fwd" else 0)) * 4 / 1e6
    return rounds * per + comb

def product_form(kind, sh, s):
    """the form the launchers run for this count (fwd: use_ksplit_fwd; backward: 8-wave forms un-split, 4-wave workgroups when split)"""
    if kind == "fwd": return "A" if (sh["D"] * 2 > 128 or sh["tiles"] * s <= CUS) else "W"
    return "A" if s == 1 else "W"

def old_rule(kind, sh):
    target = (2 if sh["D"] * 2 <= 128 else 1) * CUS
    if sh["tiles"] >= target // 2: return 1
    s = min(16, -(-target // sh["tiles"]), sh["L"] // 512)
    return s if s >= 2 else 1

def main():
    kind, files = sys.argv[1], sys.argv[2:]
    shapes = parse(kind, files)
    pts = [(sh, s, f, us) for sh in shapes for (s, f), us in sh["pts"].items() if s <= 8 or sh["tiles"] * s <= 2 * CUS]
    x0 = np.array([9.0, 8.0, 9.0, 10.5, 14.0, 16.0, 0.6, 0.5, 4.0, 0.3, 0.55])
    lo = np.array([1, 1, 1, 1, 1, 1, 0.0, 0.0, 0.5, 0.0, 0.2]); hi = np.array([40, 40, 40, 40, 60, 60, 1.0, 1.0, 15, 3.0, 1.0])
    res = least_squares(lambda th: [math.log(cost(th, kind, sh, s, f) / us) for sh, s, f, us in pts], x0, bounds=(lo, hi))
    th = res.x
    err = np.array(res.fun)
    print(f"{kind}: {len(pts)} points of {len(shapes)} shapes; rms log error {math.sqrt((err ** 2).mean()):.3f}, worst {np.abs(err).max():.3f}")
    print("constants {tA, cA, tB, cB, tC, cC, a0, b0, k0, k1, alpha} = {" + ", ".join(f"{v:.3g}" for v in th) + "}")
    worst, tot_new, tot_old = 0.0, 0.0, 0.0
    for sh in shapes:
        cands = [s for s in range(1, 17) if (s, product_form(kind, sh, s)) in sh["pts"] and (s == 1 or sh["L"] // s >= 512)]
        meas = {s: sh["pts"][(s, product_form(kind, sh, s))] for s in cands}
        pick = min(cands, key=lambda s: cost(th, kind, sh, s, product_form(kind, sh, s)))
        best = min(meas, key=meas.get)
        old = old_rule(kind, sh)
        old_us = meas.get(old)
        if old_us is None:      # a count the sweep did not run (5, 7, ...): the nearest measured one below
            old_us = meas[max(s for s in meas if s <= old)]
        reg, reg_old = meas[pick] / meas[best] - 1, old_us / meas[best] - 1
        worst = max(worst, reg); tot_new += reg; tot_old += reg_old
        print(f"  {sh['B']},{sh['H']},{sh['N']},{sh['D']},{sh['M']}: tiles {sh['tiles']:4d}  model s{pick} = {meas[pick]:6.1f} us   best s{best} = {meas[best]:6.1f}   regret {100 * reg:5.1f} %   (round-5 rule s{old}: {100 * reg_old:5.1f} %)")
    print(f"  mean regret {100 * tot_new / len(shapes):.1f} % (worst {100 * worst:.1f} %); round-5 rule: mean {100 * tot_old / len(shapes):.1f} %")
main()
