#!/usr/bin/env python3
"""Interleaved A/B of several builds of libfcsa_hip.so in ONE process (guide rule 24): per-kernel HIP-event times of the
C3 step (or --shape B,H,N,D,causal[,M[,bias[,mask]]]).  usage: ab_libs.py [--rounds R] [--shape ...] tag1 tag2 ...   ('main' = libfcsa_hip.so)"""
import os, sys, argparse, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--shape", default="4,8,4096,64,1", help="B,H,N,D,causal[,M[,bias[,mask]]]; several shapes separated by ':'")
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--scale", type=float, default=8.0)
ap.add_argument("--groups", type=int, default=1)
ap.add_argument("--single-kv", action="store_true", help="k, v of shape (B, M, D): single-headed K/V (C5)")
ap.add_argument("tags", nargs="+")
a = ap.parse_args()
import ctypes
from flash_cosine_sim_attention_amd import _torch_ops
_torch_ops.load()
binding = ctypes.CDLL(_torch_ops.BINDING_PATH)          # same loaded object as the dispatcher ops use
libs, paths = {}, {}
pkg = os.path.join(ROOT, "flash_cosine_sim_attention_amd")
def open_lib(path):
    """ctypes handle with the profile hooks only (the binding itself refuses a library of another ABI version: fcsa_torch_use_library -> -3)"""
    lib = ctypes.CDLL(path)
    lib.fcsa_profile_enable.argtypes = [ctypes.c_int32]
    lib.fcsa_profile_collect.argtypes = [ctypes.POINTER(_lib.KernelStat), ctypes.c_int32]
    lib.fcsa_last_error.restype = ctypes.c_char_p
    return lib
for t in a.tags:
    paths[t] = os.path.join(pkg, "libfcsa_hip.so" if t == "main" else f"libfcsa_hip_{t}.so")
    libs[t] = open_lib(paths[t])
def run_shape(shape):
    B, H, N, D, causal, *rest = (int(x) for x in shape.split(","))
    M = rest[0] if rest else N                     # optional sixth field: key length
    with_bias = len(rest) > 1 and rest[1] != 0     # optional seventh field: 1 = learned bias [H, N, M] (and its gradient)
    with_mask = len(rest) > 2 and rest[2] != 0     # optional eighth field: 1 = random key mask, 25 % masked (the README protocol)
    mask = (torch.rand(B, M, device="cuda") > 0.25) if with_mask else None
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
    q = torch.randn(B, H, N, D, device="cuda", dtype=dt, requires_grad=True)
    kshape = (B, M, D) if a.single_kv else (B, H, M, D)
    k, v = (torch.randn(kshape, device="cuda", dtype=dt, requires_grad=True) for _ in range(2))
    do = torch.randn(B, H, N, D, device="cuda", dtype=dt)
    bias = (0.5 * torch.randn(H, N, M, device="cuda")).to(dt).requires_grad_() if with_bias else None
    if os.environ.get("FCSA_AB_FILL"):      # constant inputs: no operand bits toggle -- how much of the time is the power limit?
        with torch.no_grad():
            for x_ in (q, k, v, do): x_.fill_(float(os.environ["FCSA_AB_FILL"]))
    def step():
        q.grad = k.grad = v.grad = None
        if bias is not None: bias.grad = None
        F.flash_cosine_sim_attention(q, k, v, mask=mask, attn_bias=bias, causal=bool(causal), scale=a.scale, groups=a.groups).backward(do)
    res = {t: {} for t in a.tags}
    for r in range(a.rounds + 1):
        for t in a.tags:
            _lib._lib = libs[t]
            assert binding.fcsa_torch_use_library(paths[t].encode()) == 0, t      # route torch.ops.fcsa.* to this build
            for _ in range(3): step()
            torch.cuda.synchronize()
            _lib.profile_enable(True)
            for _ in range(a.steps): step()
            torch.cuda.synchronize()
            st = _lib.profile_collect()
            _lib.profile_enable(False)
            if r == 0: continue          # first round = warm-up
            for s in st: res[t].setdefault(s["name"], []).append(s["total_ms"] / s["calls"] * 1e3)
    # same inputs through every build: max |difference| of (o, dq, dk, dv) against the first tag (identical math => expect 0)
    outs = {}
    for t in a.tags:
        _lib._lib = libs[t]
        assert binding.fcsa_torch_use_library(paths[t].encode()) == 0, t
        q.grad = k.grad = v.grad = None
        if bias is not None: bias.grad = None
        o = F.flash_cosine_sim_attention(q, k, v, mask=mask, attn_bias=bias, causal=bool(causal), scale=a.scale, groups=a.groups)
        o.backward(do)
        outs[t] = [x.detach().float().clone() for x in (o, q.grad, k.grad, v.grad) + ((bias.grad,) if bias is not None else ())]
    for t in a.tags[1:]:
        print(f"{t} vs {a.tags[0]}: max|diff| o/dq/dk/dv[/dbias] = " + " ".join(f"{(x - y).abs().max().item():.3g}" for x, y in zip(outs[t], outs[a.tags[0]])))
    names = ["fwd", "bwd_dq", "bwd_dkv", "bwd_dbias", "finalize", "l2norm"]
    print(f"shape {shape} {a.dtype}; median (min) us over {a.rounds} interleaved rounds")
    for t in a.tags:
        print(f"{t:12s} " + "  ".join(f"{n} {statistics.median(res[t][n]):7.1f} ({min(res[t][n]):7.1f})" for n in names if n in res[t]))

for _shape in a.shape.split(":"):
    run_shape(_shape)
