#!/usr/bin/env python3
"""Where does the per-call host time of a tiny forward+backward go?  (B4 H8 N128 D64 bf16 causal: the GPU work is ~40 us, so
every number here is host-bound.)  Prints us per call for nested slices of the call path, and the same for SDPA."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import _torch_ops
fc = _torch_ops.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
q, k, v = (torch.randn(4, 8, N, 64, device="cuda", dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
qd, kd, vd = q.detach(), k.detach(), v.detach()
do = torch.randn_like(qd)
def bench(name, fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    print(f"{name:58s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us")
bench("torch.ops.fcsa.forward (no backward state)", lambda: fc.forward(qd, kd, vd, None, None, False, 8.0, True, True, 1, False))
bench("torch.ops.fcsa.forward (with backward state)", lambda: fc.forward(qd, kd, vd, None, None, False, 8.0, True, True, 1, True))
st = fc.forward(qd, kd, vd, None, None, False, 8.0, True, True, 1, True)
bench("torch.ops.fcsa.backward", lambda: fc.backward(do, st[0], st[1], qd, kd, vd, None, None, st[2], st[3], st[4], st[5], False, 8.0, True, True, 1, False))
bench("flash_cosine_sim_attention, no grad", lambda: F.flash_cosine_sim_attention(qd, kd, vd, causal=True))
bench("flash_cosine_sim_attention, grad (forward only)", lambda: F.flash_cosine_sim_attention(q, k, v, causal=True))
def fb():
    q.grad = k.grad = v.grad = None
    F.flash_cosine_sim_attention(q, k, v, causal=True).backward(do)
import ctypes
_b = ctypes.CDLL(_torch_ops.BINDING_PATH)
_cnt = (ctypes.c_uint64 * 8)()
_b.fcsa_torch_host_ns(_cnt)                    # reset the binding's host-time counters
bench("forward + backward(dO)", fb)
_b.fcsa_torch_host_ns(_cnt)
nf, nb = max(_cnt[6], 1), max(_cnt[7], 1)
print("   inside the binding, us per call: forward checks %.1f | allocations %.1f | fcsa_forward %.1f || backward checks %.1f | allocations %.1f | fcsa_backward %.1f"
      % (_cnt[0] / nf / 1e3, _cnt[1] / nf / 1e3, _cnt[2] / nf / 1e3, _cnt[3] / nb / 1e3, _cnt[4] / nb / 1e3, _cnt[5] / nb / 1e3))
def fbs():
    q.grad = k.grad = v.grad = None
    F.flash_cosine_sim_attention(q, k, v, causal=True).sum().backward()
bench("forward + sum().backward()   [benchmark.py protocol]", fbs)
def sd():
    q.grad = k.grad = v.grad = None
    torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True).sum().backward()
bench("SDPA forward + sum().backward()", sd)
bench("SDPA forward only (grad)", lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True))
bench("torch.empty x 6", lambda: [torch.empty(4, 8, N, 64, device="cuda", dtype=torch.bfloat16) for _ in range(6)])
