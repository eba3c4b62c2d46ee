#!/usr/bin/env python3
"""A/B kernel timing across library builds (same ABI, different -D knobs).  Run on the GPU box:
     python tools/ab_bench.py name=path.so [name=path.so ...] [--rounds R]
Each variant runs in its own process (FCSA_LIB), interleaved over R rounds; prints per-kernel avg us."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variants = [a.split("=", 1) for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
rounds = 2
for a in sys.argv[1:]:
    if a.startswith("--rounds="):
        rounds = int(a.split("=")[1])
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for n, path in variants:
        env = dict(os.environ, FCSA_LIB=os.path.join(ROOT, path))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "20", "--warmup", "5"],
                             env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", out.stderr[-400:]); continue
        j = json.loads(line[-1])
        res[n].append(j)
for n, _ in variants:
    for j in res[n]:
        k = j.get("kernels") or {}
        print(f"{n:14s} {j['value']:8.1f} TF  {j['ms_per_step']*1e3:8.1f} us/step | " +
              "  ".join(f"{kk} {vv['per_step_us']:.1f}" for kk, vv in k.items()) + f" | maxdelta {j.get('max_abs_delta_vs_pytorch_f32'):.2e}")
