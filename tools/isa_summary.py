#!/usr/bin/env python3
"""Per-kernel summary of a device assembly file (hipcc -S --cuda-device-only): instruction totals by class, registers, scratch.
usage: isa_summary.py file.s [substring filter]     -- two files: prints rows side by side for kernels whose numbers differ.
Used to check that a source clean-up left the hot kernels' code unchanged (compare before / after)."""
import re, sys, subprocess

def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))

def summarize(path):
    res, cur, body = {}, None, []
    meta = {}
    for line in open(path):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and line.startswith(".Lfunc_end"):
            ins = [l.split()[0] for l in body if l.startswith("\t") and not l.lstrip().startswith((".", ";"))]
            c = lambda pre: sum(1 for i in ins if i.startswith(pre))
            res[cur] = dict(n=len(ins), mfma=c("v_mfma"), ds=c("ds_"), exp=c("v_exp"), wait=c("s_waitcnt"), bar=c("s_barrier"),
                            vmem=c("buffer_") + c("global_"), scratch=c("scratch_"), lane=c("v_readlane") + c("v_writelane"), salu=c("s_") )
            cur = None
            continue
        if cur is not None:
            body.append(line)
    # amdhsa.kernels metadata: one entry per kernel, starting at "  - .agpr_count:"; the kernel's own keys are indented 4 spaces
    text = open(path).read()
    for ent in re.split(r"\n  - (?=\.agpr_count:)", text)[1:]:
        nm = re.search(r"^    \.name:\s*(\S+)", ent, flags=re.M)
        if not nm: continue
        d = {}
        for key in ("vgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "agpr_count"):
            mm = re.search(r"^(?:    |)\." + key + r":\s*(\d+)", ent, flags=re.M)
            if mm: d[key] = int(mm.group(1))
        meta[nm.group(1)] = d
    for k, v in meta.items():
        if k in res: res[k].update(v)
    return res

if __name__ == "__main__":
    files = [a for a in sys.argv[1:] if a.endswith(".s")]
    filt = [a for a in sys.argv[1:] if not a.endswith(".s")]
    sums = [summarize(f) for f in files]
    names = sorted(set().union(*[set(s) for s in sums]))
    dm = demangle(names)
    keys = ["n", "mfma", "ds", "exp", "wait", "bar", "vmem", "scratch", "lane", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size"]
    for n in names:
        d = dm[n].replace("fcsa::", "").replace("void ", "")
        if "kernel" not in d or any(f not in d for f in filt): continue
        rows = [s.get(n) for s in sums]
        if len(rows) == 2 and rows[0] == rows[1]:
            continue
        print(d[:110])
        for f, r in zip(files, rows):
            print("   %-28s %s" % (f.split("/")[-1], "absent" if r is None else " ".join(f"{k}={r.get(k, '-')}" for k in keys)))
