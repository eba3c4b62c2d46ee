"""CPU-only checks of the drop-in boundary: libfcsa_hip.so loads, exports every symbol include/fcsa.h
declares, its structs have the layout the ctypes binding assumes, and argument validation returns the
documented error codes WITHOUT touching a GPU (no kernel is launched by any call here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fcsa.h")


@pytest.fixture(scope="module")
def lib():
    from flash_cosine_sim_attention_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fcsa_[a-z0-9_]+)\s*\(", src)))


def test_exports_every_declared_symbol(lib):
    names = _declared_functions()
    assert {"fcsa_forward", "fcsa_backward", "fcsa_backward_workspace_bytes", "fcsa_forward_workspace_bytes", "fcsa_l2norm",
            "fcsa_debug", "fcsa_last_error"} <= set(names)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fcsa.h but not exported"
    from flash_cosine_sim_attention_amd import _lib
    assert set(_lib.EXPORTS) == set(names)


def test_struct_layout_matches_c_compiler(tmp_path):
    """sizeof/offsetof as gcc sees include/fcsa.h == the ctypes mirror in _lib.py."""
    from flash_cosine_sim_attention_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "fcsa.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(fcsa_tensor), sizeof(fcsa_problem), sizeof(fcsa_norm_state),
         sizeof(fcsa_forward_args), sizeof(fcsa_backward_args));
  printf("%zu %zu %zu %zu\n", offsetof(fcsa_forward_args, o), offsetof(fcsa_forward_args, inv_l),
         offsetof(fcsa_forward_args, norm), offsetof(fcsa_forward_args, stream));
  printf("%zu %zu %zu %zu %zu\n", offsetof(fcsa_backward_args, inv_l), offsetof(fcsa_backward_args, q),
         offsetof(fcsa_backward_args, dq), offsetof(fcsa_backward_args, workspace), offsetof(fcsa_backward_args, stream));
  printf("%zu %zu\n", offsetof(fcsa_problem, scale), offsetof(fcsa_problem, groups));
  return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(prog), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    sizes = list(map(int, out[0].split()))
    assert sizes == [C.sizeof(_lib.Tensor), C.sizeof(_lib.Problem), C.sizeof(_lib.NormState),
                     C.sizeof(_lib.ForwardArgs), C.sizeof(_lib.BackwardArgs)]
    F, B, P = _lib.ForwardArgs, _lib.BackwardArgs, _lib.Problem
    assert list(map(int, out[1].split())) == [F.o.offset, F.inv_l.offset, F.norm.offset, F.stream.offset]
    assert list(map(int, out[2].split())) == [B.inv_l.offset, B.q.offset, B.dq.offset, B.workspace.offset, B.stream.offset]
    assert list(map(int, out[3].split())) == [P.scale.offset, P.groups.offset]


def _problem(**kw):
    from flash_cosine_sim_attention_amd import _lib
    d = dict(dtype=_lib.FCSA_BF16, batch=1, heads=2, kv_heads=2, q_len=8, k_len=8, dim_head=64, causal=0,
             bias_batch_dim=0, l2norm_qk=0, groups=1, scale=8.0)
    d.update(kw)
    return _lib.Problem(*[d[f[0]] for f in _lib.Problem._fields_])


def _fwd_args(prob, mask=None):
    from flash_cosine_sim_attention_amd import _lib
    t = _lib.Tensor(0x1000, 1024, 512, 64)      # fake, never dereferenced: validation fails first
    return _lib.ForwardArgs(prob, t, t, t, t, None, mask, None, _lib.NormState(None, None, None, None), None, 0, None)


def test_validation_errors_without_gpu(lib):
    from flash_cosine_sim_attention_amd import _lib
    UNSUPPORTED, INVALID = -2, -1
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(dim_head=48))))
    assert rc == UNSUPPORTED and b"dim_head 48" in lib.fcsa_last_error()
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(dtype=7))))
    assert rc == UNSUPPORTED and b"dtype" in lib.fcsa_last_error()
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(kv_heads=3))))
    assert rc == INVALID and b"kv_heads" in lib.fcsa_last_error()
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(causal=1), mask=0x2000)))
    assert rc == INVALID and b"causal" in lib.fcsa_last_error()
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(l2norm_qk=1, groups=5))))
    assert rc == INVALID and b"groups" in lib.fcsa_last_error()
    bad = _fwd_args(_problem())
    bad.q = _lib.Tensor(0x1004, 1024, 512, 64)
    rc = lib.fcsa_forward(C.byref(bad))
    assert rc == INVALID and b"aligned" in lib.fcsa_last_error()
    bad = _fwd_args(_problem())
    bad.k = _lib.Tensor(None, 0, 0, 0)
    rc = lib.fcsa_forward(C.byref(bad))
    assert rc == INVALID and b"k: null" in lib.fcsa_last_error()
    assert lib.fcsa_forward(None) == INVALID
    with pytest.raises(RuntimeError, match="status -2"):
        _lib.check(lib.fcsa_forward(C.byref(_fwd_args(_problem(dim_head=48)))), "fcsa_forward")


def test_backward_workspace_formula(lib):
    # delta [B,H,N] f32 always.  f32 slabs only where the epilogues cannot finish the job:
    #   dq, dk when l2norm groups are not 8 * 2^k features wide (finalize kernel does the l2norm backward),
    #   dk, dv for single-headed K/V (finalize kernel reduces over heads).
    al = lambda x: (x + 255) // 256 * 256
    B, H, N, M, D = 2, 4, 100, 120, 64
    p = _problem(batch=B, heads=H, kv_heads=H, q_len=N, k_len=M, dim_head=D)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(B * H * N * 4)
    p = _problem(batch=B, heads=H, kv_heads=H, q_len=N, k_len=M, dim_head=D, l2norm_qk=1, groups=2)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(B * H * N * 4)                      # fused epilogues
    p = _problem(batch=B, heads=H, kv_heads=H, q_len=N, k_len=M, dim_head=96, l2norm_qk=1, groups=1)   # ONE group of 12 blocks: fused (round 6)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(B * H * N * 4)
    p = _problem(batch=B, heads=H, kv_heads=H, q_len=N, k_len=M, dim_head=96, l2norm_qk=1, groups=2)   # groups of 6 blocks: slabs
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(B * H * N * 4) + al(B * H * N * 96 * 4) + al(B * H * M * 96 * 4)
    p = _problem(batch=B, heads=H, kv_heads=H, q_len=N, k_len=M, dim_head=96, l2norm_qk=1, groups=8)   # group size 12
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(B * H * N * 4) + al(B * H * N * 96 * 4) + al(B * H * M * 96 * 4)
    p = _problem(batch=B, heads=H, kv_heads=1, q_len=N, k_len=M, dim_head=D)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(B * H * N * 4) + 2 * al(B * H * M * D * 4)
    p = _problem(batch=B, heads=H, kv_heads=1, q_len=N, k_len=M, dim_head=D, l2norm_qk=1)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(B * H * N * 4) + 2 * al(B * H * M * D * 4)


def test_backward_workspace_single_head_non_causal_stays_small(lib):
    """the dQ slab is sized by the SAME split rule the launch uses (batch * heads row tiles), so a heads == 1 problem
    whose grid already fills the chip needs only the delta workspace (it used to reserve dq_splits x f32 dQ: 1 GiB here)."""
    al = lambda x: (x + 255) // 256 * 256
    p = _problem(batch=64, heads=1, kv_heads=1, q_len=8192, k_len=8192, dim_head=128, dtype=2, l2norm_qk=1)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(64 * 8192 * 4)
    p = _problem(batch=64, heads=1, kv_heads=1, q_len=8192, k_len=8192, dim_head=128, dtype=2, l2norm_qk=1, bias_batch_dim=1)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(64 * 8192 * 4)
    # a grid that cannot fill the chip still gets its split slabs: C4 = 8 heads x 8 row tiles of 128 -> 8 splits (512 workgroups:
    # two per CU for the two-waves-per-SIMD kernels of 128-byte rows)
    p = _problem(batch=1, heads=8, kv_heads=8, q_len=1024, k_len=8192, dim_head=64, dtype=1, l2norm_qk=1)
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(8 * 1024 * 4) + al(8 * 8 * 1024 * 64 * 4)


def test_backward_workspace_split_query_dkv(lib):
    """Split-query dK/dV (few keys, many queries, K/V with heads): dkv_splits f32 slabs for dk and for dv, sized by the
    rule the launch uses (fcsa_capi.hip backward_dkv_splits -> best_split: the argmin of the cost model over 1 .. 16)."""
    al = lambda x: (x + 255) // 256 * 256
    # 1 x 8 heads x 1024 keys = 64 key tiles -> 4 splits of 2048 queries: 256 workgroups, one round (measured 53.3 us against 53.6 with
    # the 8 splits of rounds 2 - 5, profiles/r06_split_sweep_bwd_d64.txt; its 512 dQ row tiles need no split)
    p = _problem(batch=1, heads=8, kv_heads=8, q_len=8192, k_len=1024, dim_head=64, dtype=2, l2norm_qk=1)
    slab = 4 * 8 * 1024 * 64 * 4                                # splits x heads x M x D floats
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(8 * 8192 * 4) + al(slab) + al(slab)
    # single-headed K/V (round 6): split the same way -- heads x splits slabs per gradient instead of heads
    q = _problem(batch=1, heads=8, kv_heads=1, q_len=8192, k_len=1024, dim_head=64, dtype=2, l2norm_qk=1)
    assert lib.fcsa_backward_workspace_bytes(C.byref(q)) == al(8 * 8192 * 4) + 2 * al(slab)
    # causal with more rows than keys (a key tile sees at most k_len = 1024 queries) and key grids that fill the chip keep the unsplit kernel
    q = _problem(batch=1, heads=8, kv_heads=8, q_len=8192, k_len=1024, dim_head=64, dtype=2, l2norm_qk=1, causal=1)
    assert lib.fcsa_backward_workspace_bytes(C.byref(q)) == al(8 * 8192 * 4)
    p = _problem(batch=4, heads=8, kv_heads=8, q_len=8192, k_len=1024, dim_head=64, dtype=2, l2norm_qk=1)      # 256 key tiles
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(4 * 8 * 8192 * 4)
    # 2 heads x 4 key tiles of 256-byte rows = 8 workgroups: as many splits as keep 512 queries each (4096 / 512 = 8)
    p = _problem(batch=1, heads=2, kv_heads=2, q_len=4096, k_len=512, dim_head=128, dtype=2, l2norm_qk=1)
    slab = 8 * 2 * 512 * 128 * 4
    assert lib.fcsa_backward_workspace_bytes(C.byref(p)) == al(2 * 4096 * 4) + al(slab) + al(slab)


def _split_counts(lib, **kw):
    """(forward, dQ, dK/dV) split counts of a 16-bit problem with fusable l2norm groups, read back from the workspace sizes"""
    al = lambda x: (x + 255) // 256 * 256
    p = _problem(l2norm_qk=1, **kw)
    B, H, N, M, D = p.batch, p.heads, p.q_len, p.k_len, p.dim_head
    fws, bws = lib.fcsa_forward_workspace_bytes(C.byref(p)), lib.fcsa_backward_workspace_bytes(C.byref(p))
    f = [s for s in range(1, 17) if (0 if s == 1 else al(s * B * H * N * D * 4) + al(s * B * H * N * 4)) == fws]
    b = [(sq, sk) for sq in range(1, 17) for sk in range(1, 17)
         if al(B * H * N * 4) + (al(sq * B * H * N * D * 4) if sq > 1 else 0) + (2 * al(sk * B * H * M * D * 4) if sk > 1 else 0) == bws]
    assert len(f) == 1 and len(b) == 1, (f, b)
    return f[0], b[0][0], b[0][1]


def test_split_counts_follow_the_measured_tables(lib):
    """The split counts are the argmin of a cost model fitted to tools/split_sweep.py tables (fcsa_capi.hip best_split, round 6).  Pinned
    here: its choice on shapes of those tables where the winner is clear (profiles/r06_split_sweep_*.txt; the CPU answer uses 256 CUs,
    the MI355X count), and the invariants every count keeps."""
    f16, bf16 = 1, 2
    H8 = dict(batch=1, heads=8, kv_heads=8)
    assert _split_counts(lib, dtype=f16, q_len=1024, k_len=8192, dim_head=64, **H8) == (4, 8, 1)          # C4: forward 33.3 us (8 splits: 34.8)
    assert _split_counts(lib, dtype=f16, q_len=8192, k_len=1024, dim_head=64, **H8) == (1, 1, 4)          # its mirror image
    # 192 row tiles of a 2048-key problem: the un-split 8-wave forms (25.7 / 30.8 us) -- rounds 2 - 5 split them three ways (37.0 / 43.6 us)
    assert _split_counts(lib, dtype=f16, batch=3, heads=8, kv_heads=8, q_len=1024, k_len=2048, dim_head=64) == (1, 1, 1)
    assert _split_counts(lib, dtype=f16, batch=3, heads=8, kv_heads=8, q_len=2048, k_len=1024, dim_head=64) == (1, 1, 1)
    # 160 row tiles x 8192 keys: three splits = 480 workgroups in ONE round of two per CU (64.0 us; four splits leave a quarter-full
    # second round: 78.5 us)
    assert _split_counts(lib, dtype=f16, batch=1, heads=20, kv_heads=20, q_len=1024, k_len=8192, dim_head=64) == (3, 3, 1)
    assert _split_counts(lib, dtype=bf16, q_len=512, k_len=4096, dim_head=128, **H8) == (8, 8, 1)         # 256-byte rows, 32 row tiles
    assert _split_counts(lib, dtype=bf16, batch=2, heads=8, kv_heads=8, q_len=1024, k_len=2048, dim_head=128) == (2, 2, 1)
    assert _split_counts(lib, dtype=f16, batch=1, heads=4, kv_heads=4, q_len=300, k_len=16384, dim_head=64) == (16, 16, 1)
    # causal (round 6): the tiles are PAIRS of 128-position tiles and each tile's loop range up to / from its diagonal is split (k_len = q_len - 1
    # below only keeps the workspace sizes invertible).  (1,4,4096,64): 64 pairs -> 4 splits each (forward 29.7 us against 45.1 un-split, dQ
    # 37.4 / 55.9, dK/dV 43.6 / 61.7); (1,16,2048,64): 128 pairs of a 2048-position problem stay un-split (forward 30.5 us, two splits 33.1; dQ 37.6 /
    # 38.8; dK/dV 41.3 / 44.7) -- profiles/r06_split_sweep_causal*.txt
    assert _split_counts(lib, dtype=bf16, batch=1, heads=4, kv_heads=4, q_len=4096, k_len=4095, dim_head=64, causal=1) == (4, 4, 4)
    assert _split_counts(lib, dtype=bf16, batch=1, heads=16, kv_heads=16, q_len=2048, k_len=2047, dim_head=64, causal=1) == (1, 1, 1)
    assert _split_counts(lib, dtype=bf16, batch=1, heads=4, kv_heads=4, q_len=8192, k_len=8191, dim_head=128, causal=1) == (2, 2, 2)
    assert _split_counts(lib, dtype=bf16, batch=1, heads=1, kv_heads=1, q_len=16384, k_len=16383, dim_head=64, causal=1) == (4, 8, 4)
    assert _split_counts(lib, dtype=f16, q_len=1024, k_len=8192, dim_head=64, causal=1, **H8) == (8, 8, 1)      # more keys than rows: every row sees >= 7169 keys
    assert _split_counts(lib, dtype=0, batch=1, heads=4, kv_heads=4, q_len=4096, k_len=4095, dim_head=64, causal=1) == (1, 1, 1)      # float32: causal problems are not split
    # invariants: never when the tiles fill the chip, every split keeps >= 512 positions of its loop
    assert _split_counts(lib, dtype=f16, batch=4, heads=8, kv_heads=8, q_len=1024, k_len=8192, dim_head=64) == (1, 1, 1)
    for M in (511, 1023, 1024, 1500, 2047, 4096, 5000):
        for tiles in (1, 7, 40, 100, 200):
            f, dq, dkv = _split_counts(lib, dtype=bf16, batch=1, heads=tiles, kv_heads=tiles, q_len=128, k_len=M, dim_head=64)
            assert 1 <= f <= 16 and 1 <= dq <= 16 and (f == 1 or M // f >= 512) and (dq == 1 or M // dq >= 512) and dkv == 1, (M, tiles, f, dq, dkv)


def test_forward_needs_qn(lib):
    """Inference calls of the 16-bit kernels save no normalised q (the reference's need_store_rowsum == false path, cu:1086)."""
    f = lambda need, **kw: lib.fcsa_forward_needs_qn(C.byref(_problem(l2norm_qk=1, **kw)), need)
    assert f(1, dtype=2) == 1 and f(0, dtype=2) == 0 and f(0, dtype=1, groups=8) == 0
    assert f(0, dtype=0) == 1                              # float32: q goes through the row kernel
    assert f(0, dtype=2, dim_head=96, groups=8) == 1       # group size 12: row kernel
    assert lib.fcsa_forward_needs_qn(C.byref(_problem(l2norm_qk=0)), 1) == 0
    assert lib.fcsa_forward_needs_qn(None, 1) == 0


def test_scale_range(lib):
    """No scale * groups limit any more (the public signature has none, py:308-319); only f16 refuses a scale whose folded
    c1 = scale * log2(e) leaves the type, and non-finite scales are invalid."""
    # (validation order: problem first, then buffers -- the NULL norm.kn of _fwd_args stops the call before any launch)
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(l2norm_qk=1, groups=8, scale=16.0, dtype=2))))
    assert rc == -1 and b"norm.kn" in lib.fcsa_last_error()              # scale * groups = 128 passed the problem check
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(l2norm_qk=1, scale=50000.0, dtype=1))))
    assert rc == -2 and b"float16" in lib.fcsa_last_error()
    rc = lib.fcsa_forward(C.byref(_fwd_args(_problem(l2norm_qk=1, scale=float("inf")))))
    assert rc == -1 and b"scale" in lib.fcsa_last_error()


def test_forward_workspace_formula(lib):
    # split-key forward: only problems whose 128-row tiles (causal: pairs) cannot fill the chip, with >= 1024 keys
    al = lambda x: (x + 255) // 256 * 256
    p = _problem(batch=1, heads=8, kv_heads=8, q_len=1024, k_len=8192, dim_head=64)      # C4: 64 row tiles -> 4 splits (256 8-wave workgroups)
    assert lib.fcsa_forward_workspace_bytes(C.byref(p)) == al(4 * 8 * 1024 * 64 * 4) + al(4 * 8 * 1024 * 4)
    p = _problem(batch=4, heads=8, kv_heads=8, q_len=1024, k_len=1024, dim_head=64)      # C2: 256 row tiles
    assert lib.fcsa_forward_workspace_bytes(C.byref(p)) == 0
    p = _problem(batch=1, heads=8, kv_heads=8, q_len=1024, k_len=1024, dim_head=64, causal=1)      # causal: 32 pairs, but 1024 keys are one split's worth
    assert lib.fcsa_forward_workspace_bytes(C.byref(p)) == 0
    p = _problem(batch=1, heads=8, kv_heads=8, q_len=4096, k_len=4096, dim_head=64, causal=1)      # round 6: 128 PAIRS of row tiles on 256 CUs -> 2 splits
    assert lib.fcsa_forward_workspace_bytes(C.byref(p)) == al(2 * 8 * 4096 * 64 * 4) + al(2 * 8 * 4096 * 4)
    p = _problem(batch=2, heads=8, kv_heads=8, q_len=4096, k_len=4096, dim_head=64, causal=1)      # 256 pairs: the chip is full
    assert lib.fcsa_forward_workspace_bytes(C.byref(p)) == 0
    p = _problem(batch=1, heads=8, kv_heads=8, q_len=4096, k_len=4096, dim_head=64, causal=1, dtype=0)     # float32: not split
    assert lib.fcsa_forward_workspace_bytes(C.byref(p)) == 0
    p = _problem(batch=1, heads=2, kv_heads=2, q_len=8, k_len=4096, dim_head=128)        # 2 row tiles -> 8 splits of 512 keys
    assert lib.fcsa_forward_workspace_bytes(C.byref(p)) == al(8 * 2 * 8 * 128 * 4) + al(8 * 2 * 8 * 4)
    assert lib.fcsa_forward_workspace_bytes(None) == 0


def test_debug_string(lib):
    buf = C.create_string_buffer(512)
    from flash_cosine_sim_attention_amd import _lib
    assert lib.fcsa_debug(buf, 512) == _lib.ABI_VERSION == 4          # FCSA_ABI_VERSION (include/fcsa.h)
    assert b"gfx950" in buf.value and b"bf16" in buf.value


def test_debug_forward_form_knob(lib):
    """fcsa_debug_forward_form (include/fcsa.h): set / query, returns the previous value; no device needed.  The library reads
    FCSA_FWD_WIDE128 once at load and never again (no getenv on the launch path: round-5 review)."""
    prev = lib.fcsa_debug_forward_form(-1)
    assert prev in (0, 1)
    assert lib.fcsa_debug_forward_form(0) == prev
    assert lib.fcsa_debug_forward_form(-1) == 0
    assert lib.fcsa_debug_forward_form(7) == 0          # any non-zero value means "automatic"
    assert lib.fcsa_debug_forward_form(-1) == 1
    lib.fcsa_debug_forward_form(prev)
    import os, subprocess, sys
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "flash_cosine_sim_attention_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".cuh", ".h")):
            text = open(os.path.join(src, f)).read()
            assert text.count("getenv(") <= (1 if f == "fcsa_fwd3.hip" else 0), f"{f}: getenv outside the load-time initialiser"
    # the environment is honoured at LOAD time: a fresh process with FCSA_FWD_WIDE128=0 starts at form 0
    code = "from flash_cosine_sim_attention_amd import _lib; print(_lib.forward_form(-1))"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, env=dict(os.environ, FCSA_FWD_WIDE128="0"), timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr


def test_gpu_entry_points_reject_cpu_tensors():
    """The HIP path never falls back: the C-ABI glue refuses host tensors (CPU tensors are served by the operator's own
    forward-only host path instead, see tests/test_cpu_path.py)."""
    import torch
    from flash_cosine_sim_attention_amd import ext, ops
    q = torch.randn(1, 2, 8, 64)
    with pytest.raises(RuntimeError, match="GPU"):
        ext.forward(q, q, q, None, None, False, 8.0, False)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.flash_cosine_sim_attention_hip(q, q, q, None, None, 8, 1, False, True, False)


def test_public_api_names_and_signature():
    """Same exported names / keyword defaults as the reference package (__init__.py:1, fcsa.py:308-319)."""
    import inspect
    import flash_cosine_sim_attention_amd as F
    for n in ("flash_cosine_sim_attention", "plain_cosine_sim_attention", "l2norm_tensors", "debug"):
        assert hasattr(F, n)
    sig = inspect.signature(F.flash_cosine_sim_attention)
    assert list(sig.parameters) == ["q", "k", "v", "mask", "attn_bias", "scale", "groups", "causal", "l2norm_qk",
                                    "attn_bias_batch_dim"]
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["mask"], d["attn_bias"], d["scale"], d["groups"], d["causal"], d["l2norm_qk"], d["attn_bias_batch_dim"]) == \
           (None, None, 8, 1, False, True, False)
    assert list(inspect.signature(F.plain_cosine_sim_attention).parameters) == list(sig.parameters)


def test_plain_torch_restatement_matches_oracle_on_cpu():
    """ops.plain_cosine_sim_attention (public export) against the numpy oracle and a reference fixture."""
    import numpy as np
    import cases as Cs
    import flash_cosine_sim_attention_amd as F
    from oracle import cosine_sim_oracle as O
    for name in ("g03_mask_d32_n63_f32", "g04_causal_biasH_d32_n63_f32", "g07_causal_singlekv_d128_n63_f32", "g18_merged_bh_d32_f32"):
        case = Cs.BY_NAME[name]
        inp = Cs.make_inputs(case)
        o = F.plain_cosine_sim_attention(inp["q"], inp["k"], inp["v"], mask=inp["mask"], attn_bias=inp["attn_bias"],
                                         **Cs.op_kwargs(case))
        gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))["o_plain"]
        assert np.abs(o.numpy() - gold).max() <= 2e-5


def test_package_carries_its_own_header_and_imports_without_the_repository(tmp_path):
    """round 3 advisor: `_lib.py` generates its ctypes structs from the C-ABI header at import time, so the header has to travel with
    the package.  A copy of the package directory alone (links resolved, as a wheel build or a vendored copy makes it) -- no
    repository-level include/ next to it -- must import, find its header inside itself and report the same ABI."""
    import shutil, subprocess, sys, filecmp
    pkg = os.path.join(ROOT, "flash_cosine_sim_attention_amd")
    assert filecmp.cmp(os.path.join(pkg, "include", "fcsa.h"), os.path.join(ROOT, "include", "fcsa.h"), shallow=False)
    dst = tmp_path / "site" / "flash_cosine_sim_attention_amd"
    shutil.copytree(pkg, dst, symlinks=False, ignore=shutil.ignore_patterns("csrc", "__pycache__", "libfcsa_hip_*.so"))
    assert not (tmp_path / "site" / "include").exists()
    code = ("import sys; sys.path.insert(0, %r); from flash_cosine_sim_attention_amd import _lib; "
            "assert _lib.HEADER.startswith(%r), _lib.HEADER; print(_lib.ABI_VERSION)") % (str(tmp_path / "site"), str(dst))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0, out.stderr[-2000:]
    from flash_cosine_sim_attention_amd import _lib
    assert int(out.stdout.strip()) == _lib.ABI_VERSION


def test_zero_size_problems_return_ok_without_touching_anything(lib):
    """batch, heads or q_len == 0: no output element exists -- FCSA_OK before any pointer is looked at or anything is launched (this
    runs without a GPU); torch hands NULL data pointers over for empty tensors."""
    from flash_cosine_sim_attention_amd import _lib
    for kw in (dict(batch=0), dict(heads=0, kv_heads=0), dict(q_len=0)):
        a = _fwd_args(_problem(**kw))
        a.q = a.k = a.v = a.o = _lib.Tensor(None, 0, 0, 0)
        assert lib.fcsa_forward(C.byref(a)) == 0, lib.fcsa_last_error()
    assert lib.fcsa_forward_workspace_bytes(C.byref(_problem(q_len=0))) == 0
