"""CPU-side unit test of the per-device dynamic-LDS bookkeeping (csrc/fcsa_kernels.h `ensure_dynamic_lds`: one hipFuncSetAttribute per
(kernel instantiation, device), a 64-bit device mask) and of `cu_count`, against stubbed HIP runtime entry points
(tests/native/dynamic_lds_stub.cpp, host-only g++ build).  On a 1-GPU box the device != 0 branches never execute for real."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include"), reason="needs g++ and the HIP headers")
def test_ensure_dynamic_lds_device_mask_and_cu_count(tmp_path):
    exe = str(tmp_path / "dynamic_lds_stub")
    cmd = ["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "flash_cosine_sim_attention_amd", "csrc"), os.path.join(ROOT, "tests", "native", "dynamic_lds_stub.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60, env=dict(os.environ, FCSA_STUB_CUS="304"))
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
