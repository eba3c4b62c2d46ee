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
# The C-ABI header the struct layouts below are generated from.  It ships INSIDE the package (`include/fcsa.h` next to this file:
# in the source tree a link to the repository's include/fcsa.h, in a built / installed copy the file itself -- setup.py lists it as
# package data), so an installed or vendored copy of the package directory imports without the repository around it; the source
# tree's own header is the fallback.
_HEADER_CANDIDATES = (os.path.join(_HERE, "include", "fcsa.h"), os.path.join(os.path.dirname(_HERE), "include", "fcsa.h"))
HEADER = next((h for h in _HEADER_CANDIDATES if os.path.isfile(h)), _HEADER_CANDIDATES[0])


# ---- ctypes mirrors of the structs, GENERATED from include/fcsa.h (one source of truth for the layout; the compiled binding
# csrc/fcsa_torch.cpp includes the same header) ----------------------------------------------------------------------------

_SCALARS = {"int32_t": C.c_int32, "int64_t": C.c_int64, "uint64_t": C.c_uint64, "float": C.c_float, "size_t": C.c_size_t,
            "char": C.c_char, "int": C.c_int}


def _parse_header(path):
    """{struct name: [(field, ctype)]} and {macro: int} from a C header that sticks to `typedef struct name { ... } name;`
    with one declaration per line (comma-separated declarators allowed), pointer / scalar / nested-struct / char[N] fields."""
    import re
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(FCSA_\w+)\s+(-?\d+)\s*$", text, flags=re.M)}
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            head, _, names = decl.rpartition(" ")
            while "," in head:                                   # `fcsa_tensor q, k, v` -> type = first token(s), names = the rest
                head, _, more = head.rpartition(" ")
                names = more + " " + names
            base = head.replace("const", "").strip()
            for nm in names.replace(",", " ").split():
                stars = base.count("*") + nm.count("*")
                nm = nm.strip("*")
                tname = base.replace("*", "").strip()
                arr = re.match(r"(\w+)\[(\d+)\]$", nm)
                if stars:
                    ct = C.c_void_p                              # every pointer is an opaque device / host address here
                elif tname in structs:
                    ct = structs[tname][1]
                else:
                    ct = _SCALARS[tname]
                if arr:
                    nm, ct = arr.group(1), ct * int(arr.group(2))
                fields.append((nm, ct))
        cls = type(m.group(3), (C.Structure,), {"_fields_": fields})
        structs[m.group(3)] = (fields, cls)
    return {k: v[1] for k, v in structs.items()}, macros


_STRUCTS, _MACROS = _parse_header(HEADER)
ABI_VERSION = _MACROS["FCSA_ABI_VERSION"]
Tensor = _STRUCTS["fcsa_tensor"]
Problem = _STRUCTS["fcsa_problem"]
NormState = _STRUCTS["fcsa_norm_state"]
ForwardArgs = _STRUCTS["fcsa_forward_args"]
BackwardArgs = _STRUCTS["fcsa_backward_args"]
KernelStat = _STRUCTS["fcsa_kernel_stat"]


EXPORTS = ("fcsa_forward", "fcsa_backward", "fcsa_backward_workspace_bytes", "fcsa_forward_workspace_bytes", "fcsa_forward_needs_qn",
           "fcsa_l2norm", "fcsa_debug", "fcsa_debug_forward_form", "fcsa_last_error", "fcsa_profile_enable", "fcsa_profile_collect")

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
    lib.fcsa_forward_needs_qn.argtypes = [C.POINTER(Problem), C.c_int32]
    lib.fcsa_forward_needs_qn.restype = C.c_int
    lib.fcsa_l2norm.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                C.POINTER(Tensor), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fcsa_l2norm.restype = C.c_int
    lib.fcsa_debug.argtypes = [C.c_char_p, C.c_size_t]
    lib.fcsa_debug.restype = C.c_int
    lib.fcsa_debug_forward_form.argtypes = [C.c_int32]
    lib.fcsa_debug_forward_form.restype = C.c_int
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


def forward_form(form: int) -> int:
    """Debug knob of the C ABI (include/fcsa.h, fcsa_debug_forward_form): 1 = automatic, 0 = never the 64-rows-per-wave D = 128
    forward, < 0 = query.  Returns the previous setting.  (Applies to the library THIS module loaded -- the one the compiled binding
    links, unless FCSA_LIB / fcsa_torch_use_library routed the ops elsewhere.)"""
    return int(load().fcsa_debug_forward_form(int(form)))


def source_sha256() -> str:
    """sha256 over the kernel sources and the build recipe (csrc/*.hip, *.cuh, *.h, Makefile, include/fcsa.h; names and bytes, sorted):
    what measurement files in profiles/ are keyed on.  The library BINARY is not reproducible bit for bit (two clean builds of one
    tree differ in a few bytes), so a hash of the .so would orphan every committed measurement at the first rebuild."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cuh", ".h", ".cpp")) or f == "Makefile"]
    files.append(HEADER)
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


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


# ---- helpers for code that drives the C ABI directly over torch tensors (tests, tools, ext.l2norm_device) ------------------

def dtype_code(torch_dtype) -> int:
    import torch
    return {torch.float32: FCSA_F32, torch.float16: FCSA_F16, torch.bfloat16: FCSA_BF16}[torch_dtype]


def tensor4(t) -> "Tensor":
    """fcsa_tensor over a 4-D torch tensor (element strides of the three leading dims)."""
    assert t.dim() == 4 and t.stride(3) == 1
    return Tensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))


def problem(torch_dtype, dims, causal=False, bias_batch=False, l2norm_qk=True, groups=1, scale=8.0) -> "Problem":
    """fcsa_problem; dims = (B, H, Hk, N, M, D)."""
    B, H, Hk, N, M, D = dims
    return Problem(dtype_code(torch_dtype), B, H, Hk, N, M, D, int(bool(causal)), int(bool(bias_batch)),
                   int(bool(l2norm_qk)), int(groups if l2norm_qk else 1), float(scale))


def forward_needs_qn(prob: "Problem", need_backward: bool) -> bool:
    return bool(load().fcsa_forward_needs_qn(C.byref(prob), 1 if need_backward else 0))
