#!/usr/bin/env python3
"""Phase timing of the forward kernel's inner loop from the trace build (libfcsa_hip_trace.so, -DFCSA_TRACE).
usage: FCSA_LIB=.../libfcsa_hip_trace.so python tools/trace_fwd.py [fwd|dkv|dq]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
B, H, N, D = 4, 8, 4096, 64
q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn_like(q)
for _ in range(int(os.environ.get("ITERS", "20"))):
    q.grad = k.grad = v.grad = None
    F.flash_cosine_sim_attention(q, k, v, causal=True).backward(do)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 128)()
fn = getattr(lib, "fcsa_trace_read_" + ("fwd" if which == "fwd2" else which))
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(buf) == 0
names = {"fwd2": ["P1 S0(t) | E1b(t-1) | V0 req, stage store", "P2 PV1(t-1) | E0a | stage loads", "P3 S1(t) | E0b | V1 req", "barrier", "P4 PV0(t) | E1a | K req"],
         "fwd": ["mask word", "DMA issue (tile t+1)", "-> S0 chain + V requests", "S1 chain x exp(block 0)", "-> barrier entry", "dma wait + barrier", "K requests (tile t+1)", "PV0 x exp(block 1)", "PV1", "loop end"], "dkv": ["issue loads (tile t+1)", "block 0 (incl. exposed first requests)", "ib1 M1: S + dP chains + T requests", "ib1 X: exp / dS / pack", "ib1 M2: dV + dK + R requests", "blocks 2, 3", "-", "-", "stage store", "barrier"], "dq": ["mask word + issue loads (tile t+1)", "tile compute (S, dP, exp, dQ)", "stage store", "barrier"]}.get(which, [f"seg{i}" for i in range(7)])
for w in range(4):
    a = list(buf[32 * w:32 * w + 32])
    if which in ("dkv", "fwd", "dq"): print("(slot", w, "= wave", (w & 1) + 4 * (w >> 1), ")")
    it, total = a[12], a[13]
    if it == 0: print("wave", w, "no iterations"); continue
    seg = a[:len(names)]
    print(f"wave {w}: unmasked tiles {it}, counted {sum(seg)} of kernel total {total} ticks ({100.0*sum(seg)/max(total,1):.1f}%), per tile {sum(seg)/it:.0f} ticks")
    for n, s_ in zip(names, seg):
        print(f"    {n:<36} {s_/it:8.1f} ticks/tile  {100.0*s_/max(sum(seg),1):5.1f}%")
    if which == "fwd":
        for ps in range(2):
            print(f"    pass {ps}: Q fragments (+ fused l2norm) {a[14 + 4 * ps]:7d} | first tile visible {a[15 + 4 * ps]:7d} | key loop {a[16 + 4 * ps]:8d} | epilogue {a[17 + 4 * ps]:7d}   ticks;  first iteration of the unmasked loop {a[23 + 2 * ps]:7d}, of the masked loop {a[24 + 2 * ps]:7d}")
    if which in ("dkv", "dq"):
        order = "prologue | masked tiles | unmasked tiles | epilogue" if which == "dkv" else "prologue | unmasked tiles | masked tiles | epilogue"
        for ps in range(2):
            print(f"    pass {ps}: {order} = " + " | ".join(f"{a[14 + 4 * ps + k_]:7d}" for k_ in range(4)) + "   ticks")
