"""The stated tolerances of the GPU parity tests, in ONE place, and a measurement log.

Bars (round 5: tightened to <= 1.5x what the kernels measure over the whole GPU suite -- profiles/r05_tolerance_margins.txt lists the
worst measured value of every class next to its bar).  rel-L2 = ||got - ref|| / ||ref||.

    forward o, elementwise  |got - ref| <= atol (* max|v| where stated) + rtol * |ref|,   and   rel-L2 of the whole output
    gradients (dq, dk, dv; d_bias x 1.5)  rel-L2

bf16 keeps 8 significant bits: rounding the OUTPUT alone is 1.1e-3 ... 1.6e-3 rel-L2, so north_star's "within 1e-3 rel." is below what any
bf16 result can reach (README, first screen); f16 and f32 meet it.

`check(label, dtype, measured, bar)` is what every comparison goes through: it returns measured <= bar and, when FCSA_TOL_LOG names a
file, appends one JSON line per comparison (tools/tolerance_margins.py reduces the log to the table in profiles/).
"""
import json
import os

#            atol,  rtol (one output ulp), rel-L2 of the forward output
FWD_TOL = {"f16": (5e-3, 2.0 ** -10, 1e-3), "bf16": (2e-2, 2.0 ** -7, 5e-3), "f32": (2e-5, 2e-5, 1e-5)}
GRAD_TOL = {"f16": 3e-3, "bf16": 1.2e-2, "f32": 2e-5}

_LOG = os.environ.get("FCSA_TOL_LOG")


def check(label, dtype, measured, bar, case=None):
    measured, bar = float(measured), float(bar)
    if _LOG:
        with open(_LOG, "a") as f:
            f.write(json.dumps({"label": label, "dtype": str(dtype), "measured": measured, "bar": bar, "case": case}) + "\n")
    return measured <= bar
