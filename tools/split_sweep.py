#!/usr/bin/env python3
"""Sweep of the split-key forward's launch parameters on a `-DFCSA_VAR_SPLIT_ENV` build (libfcsa_hip_sweep.so: fcsa_capi.hip and
fcsa_fwd.hip compiled with the macro read FCSA_SPLITS / FCSA_KSPLIT from the environment per call; the product build has no such hook).
Per shape: forward time (k-l2norm + forward kernel + combine kernel, HIP-event pairs inside the library) for split counts x forms in ONE
process, interleaved over rounds.  usage: split_sweep.py [--dtype f16] --shape B,H,N,D,M[:...]"""
import os, sys, argparse, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib, _torch_ops
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--dtype", default="f16")
ap.add_argument("--shape", default="1,8,1024,64,8192")
ap.add_argument("--splits", default="1,2,3,4,6,8,16")
ap.add_argument("--causal", action="store_true", help="causal problems (each tile's loop range up to / from its diagonal is split)")
ap.add_argument("--bwd", choices=["dq", "dkv"], help="sweep FCSA_DQ_SPLITS / FCSA_DKV_SPLITS of the backward instead (kernel + its share of finalize)")
a = ap.parse_args()
_torch_ops.load()
binding = ctypes.CDLL(_torch_ops.BINDING_PATH)
path = os.path.join(ROOT, "flash_cosine_sim_attention_amd", "libfcsa_hip_sweep.so")
lib = ctypes.CDLL(path)
lib.fcsa_profile_enable.argtypes = [ctypes.c_int32]
lib.fcsa_profile_collect.argtypes = [ctypes.POINTER(_lib.KernelStat), ctypes.c_int32]
lib.fcsa_last_error.restype = ctypes.c_char_p
_lib._lib = lib
assert binding.fcsa_torch_use_library(path.encode()) == 0
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
for shape in a.shape.split(":"):
    B, H, N, D, M = (int(x) for x in shape.split(","))
    q = torch.randn(B, H, N, D, device="cuda", dtype=dt)
    k, v = (torch.randn(B, H, M, D, device="cuda", dtype=dt) for _ in range(2))
    if a.bwd:
        for t in (q, k, v): t.requires_grad_()
        do = torch.randn(B, H, N, D, device="cuda", dtype=dt)
        var, other = ("FCSA_DQ_SPLITS", "FCSA_DKV_SPLITS") if a.bwd == "dq" else ("FCSA_DKV_SPLITS", "FCSA_DQ_SPLITS")
        os.environ.pop(other, None); os.environ.pop("FCSA_SPLITS", None); os.environ.pop("FCSA_KSPLIT", None)
        if a.causal: os.environ[other] = "1"      # (the other kernel un-split: its finalize launch stays out of the second column)
        length = M if a.bwd == "dq" else N
        counts = [c for c in (int(x) for x in a.splits.split(",")) if c <= max(1, length // 64)]
        kern = "bwd_dq" if a.bwd == "dq" else "bwd_dkv"
        res, g0 = {c: [] for c in counts}, None
        def step():
            q.grad = k.grad = v.grad = None
            F.flash_cosine_sim_attention(q, k, v, causal=a.causal).backward(do)
        for r in range(a.rounds + 1):
            for c in counts:
                os.environ[var] = str(c)
                for _ in range(3): step()
                torch.cuda.synchronize()
                if r == 0:
                    g = torch.cat([x.grad.float().flatten() for x in (q, k, v)])
                    if g0 is None: g0 = g.clone()
                    else: assert (g - g0).abs().max().item() < 2e-2 * g0.abs().max().item(), (c, (g - g0).abs().max().item())
                    continue
                # the finalize share of THIS kernel: finalize time with this count minus finalize time at count 1 is not separable from the
                # other kernel's finalize, so both are printed: kernel alone, and kernel + all finalize launches of the step
                _lib.profile_enable(True)
                for _ in range(a.steps): step()
                torch.cuda.synchronize()
                st = {s_["name"]: s_["total_ms"] / a.steps * 1e3 for s_ in _lib.profile_collect()}
                _lib.profile_enable(False)
                res[c].append((st.get(kern, 0.0), st.get(kern, 0.0) + st.get("finalize", 0.0)))
        os.environ.pop(var, None); os.environ.pop(other, None)
        tiles = B * H * (((N if a.bwd == "dq" else M) + 127) // 128)
        if a.causal: tiles = B * H * ((((N if a.bwd == "dq" else M) + 127) // 128 + 1) // 2)
        print(f"shape {shape} {a.dtype}{' causal' if a.causal else ''}: {a.bwd}, {tiles} tiles of 128{' (pairs)' if a.causal else ''}; us per step, median over {a.rounds} rounds: kernel (kernel + every finalize launch of the step)")
        print("   " + "  ".join(f"s{c}:{statistics.median(x[0] for x in res[c]):6.1f} ({statistics.median(x[1] for x in res[c]):6.1f})" for c in counts))
    cfgs = [] if a.bwd else [(s, ks) for s in (int(x) for x in a.splits.split(",")) for ks in ("0", "1") if s <= max(1, M // 64)]
    res = {c: [] for c in cfgs}
    ref = None
    for r in range(a.rounds + 1):
        for c in cfgs:
            os.environ["FCSA_SPLITS"], os.environ["FCSA_KSPLIT"] = str(c[0]), c[1]
            with torch.no_grad():
                for _ in range(3): o = F.flash_cosine_sim_attention(q, k, v, causal=a.causal)
                torch.cuda.synchronize()
                if r == 0:
                    if ref is None: ref = o.float().clone()
                    else: assert (o.float() - ref).abs().max().item() < 2e-3, (c, (o.float() - ref).abs().max().item())
                    continue
                _lib.profile_enable(True)
                for _ in range(a.steps): F.flash_cosine_sim_attention(q, k, v, causal=a.causal)
                torch.cuda.synchronize()
                st = _lib.profile_collect()
                _lib.profile_enable(False)
            res[c].append(sum(s["total_ms"] / s["calls"] * 1e3 for s in st if s["name"] in ("fwd",)))
    if a.bwd:
        continue
    wgs = B * H * ((N + 127) // 128) if not a.causal else B * H * (((N + 127) // 128 + 1) // 2)
    print(f"shape {shape} {a.dtype}{' causal' if a.causal else ''}: {wgs} row tiles of 128{' (pairs)' if a.causal else ''}; forward (kernel + combine) us, median over {a.rounds} rounds; columns = splits, rows = form")
    for ks in ("0", "1"):
        print(("4-wave     " if ks == "0" else "ksplit(8w) ") + "  ".join(f"s{c[0]}:{statistics.median(res[c]):6.1f}" for c in cfgs if c[1] == ks))
