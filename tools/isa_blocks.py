#!/usr/bin/env python3
"""Basic-block histogram of one kernel in a hipcc -S listing: finds the hot loop bodies and shows, per block, how many
MFMA / VALU / LDS / VMEM / s_waitcnt / s_barrier instructions it has.  usage: isa_blocks.py file.s <kernel substring> [--dump LABEL]"""
import re, sys
path, key = sys.argv[1], sys.argv[2]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(":")[0].endswith(l.split(":")[0]) and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", []]
for l in lines[start + 1:end]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append(cur); cur = [m.group(1), []]
        continue
    cur[1].append(s)
blocks.append(cur)
def cls(i):
    op = i.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "dsr"
    if op.startswith("ds_"): return "dsw"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op == "s_waitcnt": return "wait"
    if op == "s_barrier": return "bar"
    if op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rcp") or op.startswith("v_sqrt"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_"): return "salu"
    return "other"
print(f"{'block':14s} {'n':>5s} mfma valu trans dsr dsw vmem wait bar nop salu")
for name, ins in blocks:
    if dump and name == dump:
        print("\n".join(ins)); continue
    if len(ins) < 30: continue
    h = {}
    for i in ins: h[cls(i)] = h.get(cls(i), 0) + 1
    print(f"{name:14s} {len(ins):5d} " + " ".join(f"{h.get(k,0):4d}" for k in ("mfma","valu","trans","dsr","dsw","vmem","wait","bar","nop","salu")))
