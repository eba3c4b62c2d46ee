"""ctypes binding of libfcsa_hip.so -- the C ABI declared in include/fcsa.h.

The library is built in-tree by `flash_cosine_sim_attention_amd/csrc/Makefile`
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C .../csrc`).
There is NO fallback: if the shared object is missing or a call fails, an
exception is raised (the reference prints a notice and carries on with an
undefined `forward`, flash_cosine_sim_attention.py:15-23).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# FCSA_LIB selects an alternative build of the same ABI (kernel-tuning A/B runs, tools/ab_bench.py)
LIB_PATH = os.environ.get("FCSA_LIB") or os.path.join(_HERE, "libfcsa_hip.so")
CSRC = os.path.join(_HERE, "csrc")

FCSA_F32, FCSA_F16, FCSA_BF16 = 0, 1, 2
ABI_VERSION = 2


class Tensor(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("stride0", C.c_int64), ("stride1", C.c_int64), ("stride2", C.c_int64)]


class Problem(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("batch", C.c_int32), ("heads", C.c_int32), ("kv_heads", C.c_int32),
                ("q_len", C.c_int32), ("k_len", C.c_int32), ("dim_head", C.c_int32), ("causal", C.c_int32),
                ("bias_batch_dim", C.c_int32), ("l2norm_qk", C.c_int32), ("groups", C.c_int32), ("scale", C.c_float)]


class NormState(C.Structure):
    _fields_ = [("qn", C.c_void_p), ("kn", C.c_void_p), ("rq", C.c_void_p), ("rk", C.c_void_p)]


class ForwardArgs(C.Structure):
    _fields_ = [("p", Problem), ("q", Tensor), ("k", Tensor), ("v", Tensor), ("o", Tensor),
                ("inv_l", C.c_void_p), ("mask", C.c_void_p), ("attn_bias", C.c_void_p),
                ("norm", NormState), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p)]


class BackwardArgs(C.Structure):
    _fields_ = [("p", Problem), ("d_out", Tensor), ("o", Tensor), ("inv_l", C.c_void_p),
                ("q", Tensor), ("k", Tensor), ("v", Tensor), ("mask", C.c_void_p), ("attn_bias", C.c_void_p),
                ("norm", NormState), ("dq", Tensor), ("dk", Tensor), ("dv", Tensor), ("d_bias", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t), ("stream", C.c_void_p)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("calls", C.c_int32), ("total_ms", C.c_float), ("min_ms", C.c_float),
                ("max_ms", C.c_float)]


EXPORTS = ("fcsa_forward", "fcsa_backward", "fcsa_backward_workspace_bytes", "fcsa_forward_workspace_bytes", "fcsa_l2norm", "fcsa_debug",
           "fcsa_last_error", "fcsa_profile_enable", "fcsa_profile_collect")

_lib = None


def build(verbose: bool = False) -> str:
    """Compile libfcsa_hip.so for gfx950 with hipcc (works without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libfcsa_hip.so failed (see output above)")
    return LIB_PATH


def load():
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run `make -C {CSRC}` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'` at the repo root). "
            "There is no non-HIP fallback for GPU tensors.")
    lib = C.CDLL(LIB_PATH)
    lib.fcsa_forward.argtypes = [C.POINTER(ForwardArgs)]
    lib.fcsa_forward.restype = C.c_int
    lib.fcsa_backward.argtypes = [C.POINTER(BackwardArgs)]
    lib.fcsa_backward.restype = C.c_int
    lib.fcsa_backward_workspace_bytes.argtypes = [C.POINTER(Problem)]
    lib.fcsa_backward_workspace_bytes.restype = C.c_size_t
    lib.fcsa_forward_workspace_bytes.argtypes = [C.POINTER(Problem)]
    lib.fcsa_forward_workspace_bytes.restype = C.c_size_t
    lib.fcsa_l2norm.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.POINTER(Tensor), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fcsa_l2norm.restype = C.c_int
    lib.fcsa_debug.argtypes = [C.c_char_p, C.c_size_t]
    lib.fcsa_debug.restype = C.c_int
    lib.fcsa_profile_enable.argtypes = [C.c_int32]
    lib.fcsa_profile_enable.restype = C.c_int
    lib.fcsa_profile_collect.argtypes = [C.POINTER(KernelStat), C.c_int32]
    lib.fcsa_profile_collect.restype = C.c_int
    lib.fcsa_last_error.argtypes = []
    lib.fcsa_last_error.restype = C.c_char_p
    ver = lib.fcsa_debug(None, 0)
    if ver != ABI_VERSION:
        raise ImportError(f"libfcsa_hip.so ABI version {ver} != expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().fcsa_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (status {rc}): {msg}")


def profile_enable(on: bool):
    check(load().fcsa_profile_enable(1 if on else 0), "fcsa_profile_enable")


def profile_collect():
    """[{name, calls, total_ms, min_ms, max_ms}] for the kernels launched since profiling was enabled."""
    arr = (KernelStat * 16)()
    n = load().fcsa_profile_collect(arr, 16)
    if n < 0:
        raise RuntimeError("fcsa_profile_collect failed")
    return [dict(name=arr[i].name.decode(), calls=arr[i].calls, total_ms=arr[i].total_ms, min_ms=arr[i].min_ms,
                 max_ms=arr[i].max_ms) for i in range(min(n, 16))]
