#!/usr/bin/env python3
"""Ticks each wave of one workgroup spends waiting at the tile barrier vs in the tile loops (build: -DFCSA_TRACE_BAR, two s_memtime
per tile and nothing else).  usage: FCSA_LIB=.../libfcsa_hip_bar.so python tools/trace_bar.py"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _lib
B, H, N, D = 4, 8, 4096, 64
q, k, v = (torch.randn(B, H, N, D, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
do = torch.randn_like(q)
for _ in range(10):
    q.grad = k.grad = v.grad = None
    F.flash_cosine_sim_attention(q, k, v, causal=True).backward(do)
torch.cuda.synchronize()
lib = _lib.load()
for which in ("fwd", "dq", "dkv"):
    buf = (C.c_ulonglong * 64)()
    fn = getattr(lib, "fcsa_trace_read_bar_" + which)
    fn.argtypes = [C.POINTER(C.c_ulonglong)]
    assert fn(buf) == 0
    print(which, "(both passes of workgroup gridDim.x / 2 + 3): wave: barrier wait / tile loops ticks")
    for w in range(8):
        wait, loop = buf[2 * w], buf[2 * w + 1]
        if loop: print(f"   wave {w}: {wait:8d} / {loop:8d} = {100.0 * wait / loop:5.1f} %")
