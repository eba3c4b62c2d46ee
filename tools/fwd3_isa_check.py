#!/usr/bin/env python3
"""ISA guard for fwd3_kernel (csrc/fcsa_fwd3.hip; round-5 advisor finding).  Every MFMA, exponential, LDS read and LDS-DMA of that kernel
sits in a one-instruction `asm volatile` statement, so hipcc's hazard recognizer and waitcnt inserter do not see them: correctness depends
on the COMPILER never placing an instruction of its own that reads (or overwrites) an MFMA destination inside the MFMA's result latency,
and on it not spilling or shuffling the pipeline registers (an earlier variant of the kernel came out with ~1200 accumulator moves).
This script compiles the translation unit to device assembly and fails unless, for every fwd3_kernel instantiation:
  * the kernel has exactly MFMAS_EXPECTED (144 = 64 unmasked tile + 64 masked tile + 16 drain) v_mfma instructions, all inside asm statements;
  * .vgpr_spill_count == 0, .sgpr_spill_count == 0 and .private_segment_fixed_size == 0 (no scratch);
  * no compiler-generated (outside ;;#ASMSTART / ;;#ASMEND) v_accvgpr_* instruction exists inside the tile loops (loop depth >= 2), and no
    compiler-generated instruction anywhere touches a register of an MFMA destination within WINDOW issue slots behind that MFMA
    (an 8-pass MFMA needs ~18 wait states before a VALU may read its result; s_nop N counts N + 1 slots).
It prints the compiler version the check ran with.  usage: fwd3_isa_check.py [file.s]   (no argument: compiles csrc/fcsa_fwd3.hip itself)
Run by tests/test_isa_guard_cpu.py; exit status 0 = clean."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flash_cosine_sim_attention_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
MFMAS_EXPECTED = 144
WINDOW = 20
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-mllvm", "-amdgpu-mfma-vgpr-form", "-S", "--cuda-device-only"]

REG = re.compile(r"\b([vas])(?:\[(\d+):(\d+)\]|(\d+)\b)")


def regs(text):
    """set of (file, index) named in an operand string"""
    out = set()
    for m in REG.finditer(text):
        f = m.group(1)
        if m.group(4) is not None:
            out.add((f, int(m.group(4))))
        else:
            out.update((f, i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_kernel(name, body, meta):
    problems = []
    in_asm, depth = False, 0
    mfma_total, mfma_outside = 0, 0
    recent = []          # (slot index, set of dest registers) of MFMAs
    slot = 0
    for ln in body:
        s = ln.strip()
        if s.startswith(".LBB") or s.startswith("; %bb."):
            d = re.search(r"Depth=(\d+)", ln)
            depth = int(d.group(1)) if d else 0
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".")) or s.endswith(":"):
            continue
        op = s.split()[0]
        operands = s[len(op):].split(";")[0]
        slot += 1
        if op == "s_nop":
            slot += int(operands.strip() or 0)
        if op.startswith("v_mfma"):
            mfma_total += 1
            if not in_asm:
                mfma_outside += 1
            dest = regs(operands.split(",")[0])
            recent.append((slot, dest))
            recent = [r for r in recent if slot - r[0] <= WINDOW]
            continue
        if in_asm:
            continue
        # compiler-generated instruction
        if op.startswith("v_accvgpr") and depth >= 2:
            problems.append(f"compiler {op} inside a tile loop: {s}")
        if op.startswith("scratch_"):
            problems.append(f"scratch access: {s}")
        touched = regs(operands)
        for (at, dest) in recent:
            if slot - at <= WINDOW and touched & dest:
                problems.append(f"compiler instruction {slot - at} slots behind an MFMA touches its destination: {s}")
                break
    if mfma_total != MFMAS_EXPECTED:
        problems.append(f"{mfma_total} v_mfma instructions, expected {MFMAS_EXPECTED}")
    if mfma_outside:
        problems.append(f"{mfma_outside} v_mfma outside asm statements")
    for key in ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"):
        if meta.get(key, 0) != 0:
            problems.append(f".{key} = {meta[key]}")
    return problems, mfma_total


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.mkdtemp(prefix="fwd3_isa_"), "fcsa_fwd3.s")
        r = subprocess.run([HIPCC] + FLAGS + [os.path.join(CSRC, "fcsa_fwd3.hip"), "-o", path], capture_output=True, text=True, cwd=CSRC)
        if r.returncode != 0:
            print(r.stderr[-3000:])
            return 2
    ver = subprocess.run([HIPCC, "--version"], capture_output=True, text=True).stdout.split("\n")
    print("compiler:", " | ".join(v.strip() for v in ver[:2]))
    text = open(path).read()
    metas = {}
    for ent in re.split(r"\n  - (?=\.agpr_count:)", text)[1:]:
        nm = re.search(r"^    \.name:\s*(\S+)", ent, flags=re.M)
        if nm:
            metas[nm.group(1)] = {k: int(v) for k, v in re.findall(r"^\s*\.(\w+):\s*(\d+)\s*$", ent, flags=re.M)}
    bad, seen = 0, 0
    for m in re.finditer(r"^(_ZN4fcsa11fwd3_kernel\w+):.*\n", text, re.M):
        name = m.group(1)
        end = text.index(".Lfunc_end", m.end())
        problems, n = check_kernel(name, text[m.end():end].split("\n"), metas.get(name, {}))
        seen += 1
        meta = metas.get(name, {})
        print(f"{name}: {n} MFMA, vgpr {meta.get('vgpr_count')}, agpr {meta.get('agpr_count')}, spill {meta.get('vgpr_spill_count')}/{meta.get('sgpr_spill_count')}, "
              f"scratch {meta.get('private_segment_fixed_size')} -> {'clean' if not problems else 'PROBLEMS'}")
        for p in problems[:20]:
            print("   ", p)
        bad += len(problems)
    if seen < 2:
        print(f"only {seen} fwd3_kernel instantiation(s) found (expected bf16 and f16)")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
