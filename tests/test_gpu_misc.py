"""GPU checks of the smaller boundary pieces: the standalone fcsa_l2norm entry (include/fcsa.h), two devices in one process
(per-device kernel attributes), the benchmark CLI, and autograd housekeeping (save_for_backward)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("d,groups", [(64, 1), (64, 8), (128, 2), (96, 3), (96, 4), (32, 1), (16, 2), (48, 3)])
def test_fcsa_l2norm_entry_vs_oracle(dtype, d, groups):
    """fcsa_l2norm (the public l2norm_tensors as a C entry point): xn and the saved inverse norms, incl. group sizes that are
    not 8 * 2^k (96 / 4 = 24, 96 / 3 = 32, 48 / 3 = 16) and a strided input view."""
    import ctypes as C
    from flash_cosine_sim_attention_amd import _core, _lib
    from oracle import cosine_sim_oracle as O
    lib = _lib.load()
    torch.manual_seed(d + groups)
    base = torch.randn(2, 37, 3, d, device="cuda", dtype=dtype)
    x = base.transpose(1, 2)                                   # [2, 3, 37, d] view of [2, 37, 3, d] memory
    out = torch.empty((2, 3, 37, d), device="cuda", dtype=dtype)
    inv = torch.empty((2, 3, 37, groups), device="cuda", dtype=torch.float32)
    t = _lib.Tensor(x.data_ptr(), x.stride(0), x.stride(1), x.stride(2))
    rc = lib.fcsa_l2norm(_core._DTYPES[dtype], 2, 3, 37, d, groups, C.byref(t), out.data_ptr(), inv.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.fcsa_last_error()
    xd = x.double().cpu().numpy()
    ref = O.l2norm(xd, groups)
    tol = {torch.float32: 2e-6, torch.float16: 1e-3, torch.bfloat16: 8e-3}[dtype]
    assert np.abs(out.double().cpu().numpy() - ref).max() <= tol
    norms = np.linalg.norm(xd.reshape(2, 3, 37, groups, d // groups), axis=-1)
    assert np.abs(inv.double().cpu().numpy() * norms - 1).max() <= 1e-5
    # the Python helper over the same entry point
    assert torch.equal(_core.l2norm_device(x.contiguous(), groups), out)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_two_devices_in_one_process():
    """hipFuncAttributeMaxDynamicSharedMemorySize is per device: kernels that need > 64 KiB of LDS must launch on the second
    device too, and results must agree across devices."""
    import flash_cosine_sim_attention_amd as F
    outs = []
    for dev in ("cuda:0", "cuda:1"):
        g = torch.Generator(device=dev).manual_seed(5)
        q, k, v = (torch.randn((2, 4, 1000, 128), device=dev, dtype=torch.bfloat16, generator=g).requires_grad_() for _ in range(3))
        o = F.flash_cosine_sim_attention(q, k, v, causal=True)
        o.backward(torch.ones_like(o))
        torch.cuda.synchronize(dev)
        outs.append((o.detach().cpu(), q.grad.cpu(), k.grad.cpu(), v.grad.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_benchmark_cli_smoke():
    """benchmark.py (reference flags, benchmark.py:21-42) runs end to end for one iteration."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "benchmark.py"), "--num-times", "1", "--dtypes", "bfloat16",
                          "--seq-lens", "128", "256", "--causal"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "128" in res.stdout and "256" in res.stdout


def test_saved_tensors_are_tracked_by_autograd():
    """save_for_backward semantics (reference py:270): modifying a saved input in place between forward and backward is an
    error instead of silently wrong gradients, and nothing is kept alive through a ctx reference cycle."""
    import gc
    import weakref
    import flash_cosine_sim_attention_amd as F
    q, k, v = (torch.randn(1, 2, 64, 32, device="cuda", dtype=torch.float16).requires_grad_() for _ in range(3))
    kk = k * 1.0
    o = F.flash_cosine_sim_attention(q, kk, v)
    kk.mul_(2)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        o.sum().backward()
    gc.disable()
    try:
        o = F.flash_cosine_sim_attention(q, k, v)
        ref = weakref.ref(o.grad_fn)
        o.sum().backward()
        del o
        assert ref() is None            # freed by reference counting alone, no cyclic GC needed
    finally:
        gc.enable()
