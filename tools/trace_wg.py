#!/usr/bin/env python3
"""Start / end time of every workgroup of the C3 kernels from the trace build (libfcsa_hip_trace.so, -DFCSA_TRACE):
how evenly the one-workgroup-per-CU grids finish.  usage: FCSA_LIB=.../libfcsa_hip_trace.so python tools/trace_wg.py"""
import ctypes as C, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
# SHAPE="B,H,N,D[,single_kv[,groups[,scale]]]" (default: C3); the trace arrays hold the first 256 workgroups of a launch
_sh = [float(x) for x in os.environ.get("SHAPE", "4,8,4096,64").split(",")]
B, H, N, D = (int(x) for x in _sh[:4])
single, groups, scale = (len(_sh) > 4 and _sh[4] != 0), (int(_sh[5]) if len(_sh) > 5 else 1), (_sh[6] if len(_sh) > 6 else 8.0)
q = torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
k, v = (torch.randn((B, N, D) if single else (B, H, N, D), device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(2))
do = torch.randn_like(q)
print("shape", (B, H, N, D), "single_kv", single, "groups", groups, "scale", scale)
for _ in range(int(os.environ.get("ITERS", "20"))):
    q.grad = k.grad = v.grad = None
    F.flash_cosine_sim_attention(q, k, v, causal=True, groups=groups, scale=scale).backward(do)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 2048)()
for which in ("fwd", "dq", "dkv"):
    fn = getattr(lib, "fcsa_trace_read_wg_" + which, None)
    if fn is None:
        print(f"== {which}: no trace symbol in this build"); continue
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    n = int(os.environ.get("NWG", "256"))
    st = [buf[2 * i] for i in range(n)]
    en = [buf[2 * i + 1] for i in range(n)]
    t0 = min(st)
    dur = [e - s for s, e in zip(st, en)]
    print(f"== {which}: kernel span {max(en) - t0} ticks; workgroup start spread {max(st) - t0}; end: first {min(en) - t0}, median {statistics.median(en) - t0:.0f}, last {max(en) - t0}")
    print(f"   duration min / median / max = {min(dur)} / {statistics.median(dur):.0f} / {max(dur)}")
    for x in range(8):
        d = [dur[i] for i in range(n) if i % 8 == x]
        e = [en[i] - t0 for i in range(n) if i % 8 == x]
        print(f"   XCD {x}: duration median {statistics.median(d):.0f} max {max(d)}; last end {max(e)}")
    # by tile pair index (slot % PT): pairs differ in where their pass boundary falls
    pt = {}
    for i in range(n):
        pt.setdefault((i >> 3) % 8, []).append(dur[i])
    print("   by pair index: " + "  ".join(f"{k_}:{statistics.median(v_):.0f}" for k_, v_ in sorted(pt.items())))
    pb = (C.c_ulonglong * 2560)()
    fn = getattr(lib, "fcsa_trace_read_pass_" + which, None)
    if fn is not None:
        fn.argtypes = [C.POINTER(C.c_ulonglong)]
        assert fn(pb) == 0
        # marks: 0 pass start | 1 prologue done | 2 first loop done | 3 second loop done | 4 epilogue done
        seg = {"prologue": [], "loops": [], "epilogue": [], "between": []}
        for i in range(n):
            m = [[pb[i * 10 + ps * 5 + k_] for k_ in range(5)] for ps in range(2)]
            pro = sum(m[ps][1] - m[ps][0] for ps in range(2)); lo = sum(m[ps][3] - m[ps][1] for ps in range(2)); ep = sum(m[ps][4] - m[ps][3] for ps in range(2))
            seg["prologue"].append(pro); seg["loops"].append(lo); seg["epilogue"].append(ep); seg["between"].append(dur[i] - pro - lo - ep)
        tot = statistics.median(dur)
        print("   median per workgroup (both passes): " + "  ".join(f"{k_} {statistics.median(v_):.0f} ({100 * statistics.median(v_) / tot:.1f}%)" for k_, v_ in seg.items()))
        first = [pb[i * 10 + 1] - pb[i * 10 + 0] for i in range(n)]; second = [pb[i * 10 + 6] - pb[i * 10 + 5] for i in range(n)]
        e1 = [pb[i * 10 + 4] - pb[i * 10 + 3] for i in range(n)]; e2 = [pb[i * 10 + 9] - pb[i * 10 + 8] for i in range(n)]
        print(f"   prologue pass 0 / pass 1 median {statistics.median(first):.0f} / {statistics.median(second):.0f}; epilogue pass 0 / pass 1 median {statistics.median(e1):.0f} / {statistics.median(e2):.0f}")
    if which == "dkv":
        order = sorted(range(n), key=lambda i: -dur[i])
        print("   slowest 48 (block: duration start-offset-in-XCD): " + "  ".join(f"{i}:{dur[i]}:{st[i] - min(st[j] for j in range(n) if j % 8 == i % 8)}" for i in order[:48]))
        print("   fastest 16: " + "  ".join(f"{i}:{dur[i]}" for i in order[-16:]))
