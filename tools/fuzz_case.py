#!/usr/bin/env python3
"""Re-run ONE configuration of tests/test_gpu_fuzz.py in every dtype and print measured / limit for the forward and each gradient.
A rounding effect shrinks 8x from bf16 to f16 and vanishes in f32; a defect does not.  Measurement tool (GPU box).
usage: fuzz_case.py "<python dict as printed by the failing assert>"  [more dicts ...]"""
import ast, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))      # `cases` (tests/conftest.py does this under pytest)
sys.path.insert(0, os.path.join(ROOT, "tests"))                # `tolerances`
from tests.test_gpu_fuzz import evaluate

for text in sys.argv[1:]:
    cfg = ast.literal_eval(text)
    print(cfg)
    for dtype in ("bf16", "f16", "f32"):
        c = dict(cfg, dtype=dtype)
        worst = {}
        for what, got, lim in evaluate(c):
            key = what.split(": ")[-1]
            if key not in worst or got / max(lim, 1e-30) > worst[key][0] / max(worst[key][1], 1e-30):
                worst[key] = (got, lim)
        print(f"  {dtype:5s}", "   ".join(f"{k} {g:.2e}/{l:.2e}" for k, (g, l) in worst.items()))
