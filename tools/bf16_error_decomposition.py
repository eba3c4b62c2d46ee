#!/usr/bin/env python3
"""Where does the bf16 forward error of the fused op come from?  

CPU emulation of the forward kernel's arithmetic on one (batch, head) slice of the C3 workload, f64 everywhere except at the
roundings under test, against exact f64 softmax attention on the SAME bf16 inputs.  The kernel rounds in three places:
    A  the normalised operands of the S chain:  c1 * q^ and k^ -> 16 bit          (fcsa_fwd.hip finish_q_frags, fcsa_norm.hip)
    B  the un-normalised probabilities P~ = exp2(S - c2) -> 16 bit before P~ V    (SecondB::prep); the row sum uses the rounded P~
    C  the output O = P~ V / l -> bf16                                              (RowEpilogue)
Each is switched on alone, then all together; A is also evaluated with float16 operands (|q^|, |k^| <= 1: 11 significant bits,
same MFMA rate) -- the candidate fix.  No GPU needed:  python tools/bf16_error_decomposition.py [N] [scale]
"""
import sys
import math
import torch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
D = 64
torch.manual_seed(0)
q, k, v = (torch.randn(N, D).to(torch.bfloat16).double() for _ in range(3))
c1 = scale * math.log2(math.e)
causal = torch.ones(N, N, dtype=torch.bool).tril()


def rnd(x, dt):
    return x if dt is None else x.to(dt).double()


def run(a_dt=None, b_dt=None, c_dt=None):
    qh = rnd(c1 * torch.nn.functional.normalize(q, dim=-1), a_dt)
    kh = rnd(torch.nn.functional.normalize(k, dim=-1), a_dt)
    s = qh @ kh.t() - c1                                  # log2 units, shift = scale (groups = 1)
    p = torch.where(causal, torch.exp2(s), torch.zeros_like(s))
    p = rnd(p, b_dt)
    o = (p @ v) / p.sum(-1, keepdim=True)
    return rnd(o, c_dt)


exact = run()
bf, hf = torch.bfloat16, torch.float16
rows = [("A  q^, k^ -> bf16", dict(a_dt=bf)), ("A' q^, k^ -> f16", dict(a_dt=hf)), ("B  P~ -> bf16", dict(b_dt=bf)),
        ("C  O -> bf16", dict(c_dt=bf)), ("A+B+C (the kernel today)", dict(a_dt=bf, b_dt=bf, c_dt=bf)),
        ("A'+B+C (f16 S operands)", dict(a_dt=hf, b_dt=bf, c_dt=bf)), ("B+C (exact S)", dict(b_dt=bf, c_dt=bf))]
print(f"one (b, h) slice of C3: N = M = {N}, D = {D}, bf16 inputs, causal, scale {scale}; error vs exact f64 math on the same inputs")
print(f"{'roundings switched on':32s} {'rel-L2':>10s} {'max-abs':>10s}")
for name, kw in rows:
    o = run(**kw)
    print(f"{name:32s} {((o - exact).norm() / exact.norm()).item():10.2e} {(o - exact).abs().max().item():10.2e}")
