#!/usr/bin/env python3
"""Phase timing of the D = 128 wide forward's tile loop (csrc/fcsa_fwd3.hip) from the trace build (libfcsa_hip_trace.so, -DFCSA_TRACE):
s_memtime stamps at the phase boundaries of the UNMASKED tiles of one workgroup (all four waves).
usage: FCSA_LIB=.../libfcsa_hip_trace.so python tools/trace_fwd3.py [causal=1] [N=4096]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
causal = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
B, H, D = 4, 8, 128
q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
with torch.no_grad():
    for _ in range(20):
        F.flash_cosine_sim_attention(q, k, v, causal=causal)
torch.cuda.synchronize()
lib = _lib.load()
buf = (C.c_ulonglong * 128)()
fn = lib.fcsa_trace_read_fwd3
fn.argtypes = [C.POINTER(C.c_ulonglong)]
assert fn(buf) == 0
names = ["P1 S0(t)    | sm 2nd half (t-1, kb1) | V^T reads", "P2 PV1(t-1) | sm 1st half (t, kb0)   | K reads, va step", "lgkm(0) + vmcnt + barrier",
         "P3 S1(t)    | sm 2nd half (t, kb0)   | V^T reads, ka step, DMA K", "P4 PV0(t)   | sm 1st half (t, kb1)   | K reads, DMA V"]
print(f"(B,H,N,D)=({B},{H},{N},{D}) causal={causal}: 64 MFMAs per tile and wave = 2048 matrix-pipe cycles")
for w in range(4):
    a = list(buf[32 * w:32 * w + 32])
    it, total = a[12], a[13]
    if it == 0:
        print("wave", w, "no unmasked tiles"); continue
    seg = a[:5]
    print(f"wave {w}: unmasked tiles {it}, counted {sum(seg)} of kernel total {total} ticks ({100.0 * sum(seg) / max(total, 1):.1f}%), per tile {sum(seg) / it:.0f} ticks")
    for n, s_ in zip(names, seg):
        print(f"    {n:<62} {s_ / it:8.1f} ticks/tile  {100.0 * s_ / max(sum(seg), 1):5.1f}%")
    marks = ["pass start", "DMA of the first tiles issued", "q rows loaded + l2norm + published", "q -> accumulator file, state init", "vmcnt(0) + barrier: first tiles visible",
             "unmasked tiles done", "masked tiles done", "drain + epilogue done"]
    for ps in range(2):
        m = a[14 + 8 * ps:22 + 8 * ps]
        if m[7] == 0: continue
        print(f"    pass {ps}: " + " | ".join(f"{marks[i]} +{m[i] - (m[i - 1] if i else (a[14 + 8 * (ps - 1) + 7] if ps else 0)):d}" for i in range(8)))
