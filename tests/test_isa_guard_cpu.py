"""Build-hygiene guard for the asm-statement kernel (csrc/fcsa_fwd3.hip): tools/fwd3_isa_check.py on freshly generated device assembly
(hipcc cross-compiles gfx950 without a GPU; ~5 s), plus a negative control -- the same assembly with ONE compiler-side read of an MFMA
destination injected right behind the MFMA must be flagged.  (Round-5 advisor: "nothing in the build or tests guards this".)"""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "fwd3_isa_check.py")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_fwd3_isa_is_clean_and_the_guard_detects_a_violation(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fwd3_isa_check as G
    asm = str(tmp_path / "fcsa_fwd3.s")
    r = subprocess.run([HIPCC] + G.FLAGS + [os.path.join(G.CSRC, "fcsa_fwd3.hip"), "-o", asm], capture_output=True, text=True, cwd=G.CSRC, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    ok = subprocess.run([sys.executable, TOOL, asm], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stdout + ok.stderr
    assert ok.stdout.count("clean") == 2 and "144 MFMA" in ok.stdout
    # negative control: a compiler-side v_mov that reads the first accumulator register of an MFMA, one slot behind it
    text = open(asm).read()
    m = re.search(r"(\tv_mfma_f32_32x32x16_bf16 (v\[(\d+):\d+\])[^\n]*\n\t;;#ASMEND\n)", text)
    assert m, "no asm-statement MFMA with a VGPR destination found"
    bad = text[:m.end()] + "\tv_mov_b32_e32 v1, v%s\n" % m.group(3) + text[m.end():]
    bad_path = str(tmp_path / "bad.s")
    open(bad_path, "w").write(bad)
    flagged = subprocess.run([sys.executable, TOOL, bad_path], capture_output=True, text=True, timeout=120)
    assert flagged.returncode == 1 and "touches its destination" in flagged.stdout, flagged.stdout
    # ... and a spilled register is reported through the metadata
    spilled = text.replace(".vgpr_spill_count: 0", ".vgpr_spill_count: 3", 1)
    sp_path = str(tmp_path / "spill.s")
    open(sp_path, "w").write(spilled)
    sp = subprocess.run([sys.executable, TOOL, sp_path], capture_output=True, text=True, timeout=120)
    assert sp.returncode == 1 and "vgpr_spill_count = 3" in sp.stdout, sp.stdout
