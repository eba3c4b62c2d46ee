#!/usr/bin/env python3
"""Which FORM of each kernel is fastest where -- on a `-DFCSA_VAR_SPLIT_ENV` build (libfcsa_hip_sweep.so; csrc/dev/fcsa_sweep_env.h: the
sweep build reads FCSA_FWD_FORM / FCSA_DQ_FORM / FCSA_DKV_FORM per call, the product build has no such hook).  Forms (rows <= 128 bytes):
0 = the product's dispatch, 1 = 256-position tiles of 8 waves, 2 = 128-position tiles of 8 waves whose halves split the loop range,
3 = 128-position tiles of 4 waves (two workgroups per CU), forward only: 4 = the 64-rows-per-wave kernel (D <= 64), 5 = its D = 128
counterpart (fcsa_fwd3.hip) whatever the grid.  16-bit D = 96 / 128: 1 / 2 are the lean (two waves per SIMD) forms, 3 one wave per SIMD.  Per shape and kernel: HIP-event
time of that kernel in a forward + backward step, forms interleaved over rounds in ONE process.
usage: form_sweep.py [--dtype bf16] --shape B,H,N,D,causal[,M][:...]"""
import os, sys, argparse, statistics, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib, _torch_ops
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=15)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--shape", default="4,8,4096,64,1")
ap.add_argument("--single-kv", action="store_true", help="k, v of shape (B, M, D)")
ap.add_argument("--groups", type=int, default=1)
ap.add_argument("--scale", type=float, default=8.0)
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
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
KERNELS = (("fwd", "FCSA_FWD_FORM", (0, 1, 2, 3, 4, 5)), ("bwd_dq", "FCSA_DQ_FORM", (0, 1, 2, 3)), ("bwd_dkv", "FCSA_DKV_FORM", (0, 1, 2, 3)))
for shape in a.shape.split(":"):
    B, H, N, D, causal, *rest = (int(x) for x in shape.split(","))
    M = rest[0] if rest else N
    q = torch.randn(B, H, N, D, device="cuda", dtype=dt, requires_grad=True)
    k, v = (torch.randn((B, M, D) if a.single_kv else (B, H, M, D), device="cuda", dtype=dt, requires_grad=True) for _ in range(2))
    do = torch.randn(B, H, N, D, device="cuda", dtype=dt)
    def step():
        q.grad = k.grad = v.grad = None
        F.flash_cosine_sim_attention(q, k, v, causal=bool(causal), groups=a.groups, scale=a.scale).backward(do)
    for var in ("FCSA_FWD_FORM", "FCSA_DQ_FORM", "FCSA_DKV_FORM"): os.environ.pop(var, None)
    step(); torch.cuda.synchronize()
    ref = [x.grad.float().clone() for x in (q, k, v)]
    print(f"shape {shape} {a.dtype}: us per launch, median over {a.rounds} rounds (form 0 = product dispatch)")
    for kern, var, forms in KERNELS:
        res = {f: [] for f in forms}
        bad = set()
        for r in range(a.rounds + 1):
            for f in forms:
                if f in bad: continue
                os.environ[var] = str(f)
                try:
                    for _ in range(2): step()
                    torch.cuda.synchronize()
                except Exception as e:      # a form the shape cannot take (e.g. the wide forward with a key mask)
                    bad.add(f); continue
                if r == 0:
                    err = max((x.grad.float() - y).abs().max().item() / max(y.abs().max().item(), 1e-6) for x, y in zip((q, k, v), ref))
                    assert err < 2e-2, (kern, f, err)
                    continue
                _lib.profile_enable(True)
                for _ in range(a.steps): step()
                torch.cuda.synchronize()
                st = {s_["name"]: s_["total_ms"] / s_["calls"] * 1e3 for s_ in _lib.profile_collect()}
                _lib.profile_enable(False)
                res[f].append(st.get(kern, float("nan")))
        os.environ.pop(var, None)
        print(f"  {kern:8s} " + "  ".join(f"f{f}:{statistics.median(res[f]):7.1f}" for f in forms if f not in bad and res[f]))
