#!/usr/bin/env python3
"""For ONE configuration of tests/test_gpu_fuzz.py: the kernels' gradients, the working-precision model's
(oracle.attention_backward_emulated) and the two float64 references (raw inputs / 16-bit operands), pairwise rel-L2 distances per
gradient.  Tells a conditioning effect (kernel ~ model, both far from float64) from a kernel defect (kernel far from model AND from
float64).  usage: fuzz_model_probe.py "<python dict>" [...]      Measurement tool (GPU box)."""
import ast, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import flash_cosine_sim_attention_amd as F
from oracle import cosine_sim_oracle as O
from tests.test_gpu_fuzz import DT, _npf


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-3 * np.sqrt(b.size))


for text in sys.argv[1:]:
    cfg = ast.literal_eval(text)
    for dtype in ([cfg["dtype"]] if os.environ.get("ONE_DTYPE") else ["bf16", "f16", "f32"]):
        c = dict(cfg, dtype=dtype)
        dt = DT[dtype]
        B, H, N, M, D = c["B"], c["H"], c["N"], c["M"], c["D"]
        g = torch.Generator(device="cuda").manual_seed(c["seed"])
        q = torch.randn((B, H, N, D), device="cuda", dtype=dt, generator=g)
        kv = (B, M, D) if c["single_kv"] else (B, H, M, D)
        k = torch.randn(kv, device="cuda", dtype=dt, generator=g)
        v = torch.randn(kv, device="cuda", dtype=dt, generator=g)
        mask = None
        if c["mask"]:
            mask = torch.rand((B, M), device="cuda", generator=g) > 0.3
            mask[:, 0] = True
        bias = None
        if c["bias"]:
            bias = (0.5 * torch.randn((B if c["bias_batch"] else H, N, M), device="cuda", generator=g)).to(dt).requires_grad_()
        q.requires_grad_(); k.requires_grad_(); v.requires_grad_()
        kw = dict(mask=mask, attn_bias=bias, scale=c["scale"], groups=c["groups"], causal=c["causal"], l2norm_qk=c["l2norm"],
                  attn_bias_batch_dim=c["bias_batch"] if bias is not None else False)
        o = F.flash_cosine_sim_attention(q, k, v, **kw)
        do = torch.randn(o.shape, device="cuda", dtype=dt, generator=g)
        o.backward(do)
        torch.cuda.synchronize()
        okw = dict(mask=None if mask is None else _npf(mask).astype(bool), attn_bias=None if bias is None else _npf(bias), scale=c["scale"], groups=c["groups"],
                   causal=c["causal"], l2norm_qk=c["l2norm"], attn_bias_batch_dim=kw["attn_bias_batch_dim"])
        args = (_npf(do), _npf(q), _npf(k), _npf(v))
        raw = O.attention_backward(*args, **okw)[:3]
        fai = O.attention_backward(*args, operand_dtype=dtype, o_saved=_npf(o), **okw)[:3] if dtype != "f32" else raw
        em = O.attention_backward_emulated(*args, dtype, o_saved=_npf(o) if not os.environ.get("MODEL_OWN_O") else None, **okw)
        ker = (_npf(q.grad), _npf(k.grad), _npf(v.grad))
        print(f"{c['id']} {dtype}: |o_kernel - o_model| rel {rel(_npf(o), em[0]):.2e}")
        for name, kk, mm, rr, ff in zip(("dq", "dk", "dv"), ker, em[1:4], raw, fai):
            print(f"   {name}: kernel-raw {rel(kk, rr):.2e}  model-raw {rel(mm, rr):.2e}  kernel-model {rel(kk, mm):.2e}  kernel-faithful {rel(kk, ff):.2e}  "
                  f"model-faithful {rel(mm, ff):.2e}  |raw| {np.linalg.norm(rr):.2e}")
