#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE implementation (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Imports the reference's pure-PyTorch functions by file path
(/root/reference/flash_cosine_sim_attention/flash_cosine_sim_attention.py; the
package __init__ cannot be imported without its CUDA extension) and records,
for each case in `cases.py`:

  o_plain   reference plain_cosine_sim_attention, evaluated in float64      (fcsa.py:75-126)
  dq,dk,dv,(db)  torch.autograd through it with a seeded random dO, float64  (the grad oracle of tests/test.py:73-125)
  o_tiled   reference flash_cosine_sim_attention on CPU tensors, i.e. l2norm_tensors + the tiled
            CPU forward (fcsa.py:130-241, 308-334), float32 internally       (only where the reference is correct, see cases.py)
  qn,kn     reference l2norm_tensors(q, k, groups), a few cases only         (fcsa.py:57-65)

Inputs are NOT stored: they are regenerated from the case's seed by
`cases.make_inputs` (numpy legacy RandomState -> float32 -> rounded to the case
dtype with torch), identically here and in the tests.  Outputs are stored as
float32 arrays in tests/golden/<case>.npz.  Nothing from /root/reference is
copied; the fixtures are data.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import cases as C  # noqa: E402

REF = "/root/reference/flash_cosine_sim_attention/flash_cosine_sim_attention.py"


def load_reference():
    spec = importlib.util.spec_from_file_location("fcsa_ref", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)      # prints a "CUDA extension ..." notice; harmless
    return ref


def main():
    ref = load_reference()
    torch.set_num_threads(4)
    only = set(sys.argv[1:])                     # optional: case-name prefixes (e.g. g28 g29) -- the other files stay untouched
    for case in C.CASES:
        if only and not any(case["name"].startswith(p) for p in only):
            continue
        inp = C.make_inputs(case)               # dict of torch tensors in the case dtype
        kw = C.op_kwargs(case)
        f64 = {n: (t.double() if t is not None and t.is_floating_point() else t) for n, t in inp.items()}
        q = f64["q"].clone().requires_grad_()
        k = f64["k"].clone().requires_grad_()
        v = f64["v"].clone().requires_grad_()
        bias = f64["attn_bias"].clone().requires_grad_() if f64["attn_bias"] is not None else None
        o = ref.plain_cosine_sim_attention(q, k, v, mask=inp["mask"], attn_bias=bias, **kw)
        (o * f64["do"]).sum().backward()
        out = {
            "o_plain": o.detach().numpy().astype(np.float32),
            "dq": q.grad.numpy().astype(np.float32),
            "dk": k.grad.numpy().astype(np.float32),
            "dv": v.grad.numpy().astype(np.float32),
        }
        if bias is not None:
            out["db"] = bias.grad.numpy().astype(np.float32)
        if case.get("tiled_ok", True):
            with torch.no_grad():
                f32 = {n: (t.float() if t is not None and t.is_floating_point() else t) for n, t in inp.items()}
                ot = ref.flash_cosine_sim_attention(f32["q"], f32["k"], f32["v"], mask=inp["mask"],
                                                    attn_bias=f32["attn_bias"], **kw)
            out["o_tiled"] = ot.numpy().astype(np.float32)
        if kw.get("l2norm_qk", True) and case["name"][:3] in ("g01", "g12", "g16", "g17"):
            qn, kn = ref.l2norm_tensors(f64["q"], f64["k"], groups=kw.get("groups", 1))
            out["qn"] = qn.numpy().astype(np.float32)
            out["kn"] = kn.numpy().astype(np.float32)
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **out)
        print(f"{case['name']:40s} {os.path.getsize(path) / 1024:8.1f} KiB")


if __name__ == "__main__":
    main()
