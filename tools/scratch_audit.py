#!/usr/bin/env python3
"""Where does a kernel touch scratch memory?  Reads the device assembly of a translation unit
(hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc -mllvm -amdgpu-mfma-vgpr-form -S --cuda-device-only fcsa_bwd.hip -o bwd.s)
and prints, per kernel that has any, its scratch instructions in total and inside blocks of loop depth >= 2 -- the tile loops of these
kernels sit inside the pass loop, so depth >= 2 means "every tile" (a spill reload there is a memory round trip that counts in vmcnt and
forces a wait behind the loads meant to stay in flight; pass-level spill costs nothing that shows).
usage: scratch_audit.py file.s [substring of the mangled kernel name]        (measurement / build hygiene tool)"""
import re, subprocess, sys

text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else "kernel"
for m in re.finditer(r"^(_ZN4fcsa\w+):.*\n", text, re.M):
    name = m.group(1)
    if pat not in name:
        continue
    body = text[m.end():text.index("s_endpgm", m.end())]
    depth, scratch, cur = {}, {}, None
    for ln in body.split("\n"):
        if ln.startswith(".LBB"):
            cur = ln.split(":")[0]
            d = re.search(r"Depth=(\d+)", ln)
            depth[cur], scratch[cur] = (int(d.group(1)) if d else 0), 0
        elif cur and "scratch_" in ln:
            scratch[cur] += 1
    total = sum(scratch.values())
    if total == 0:
        continue
    inner = sum(v for k, v in scratch.items() if depth[k] >= 2)
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    print(f"{name[:110]:110s} scratch instructions {total:4d}   inside tile loops {inner:4d}")
