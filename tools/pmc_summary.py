#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir> [<dir> ...]   (prints a table; test/profiling infrastructure only)"""
import csv
import glob
import os
import sys
from collections import defaultdict

KEEP = ("fwd_kernel", "bwd_dq_kernel", "bwd_dkv_kernel", "l2norm_pair_kernel", "l2norm_kernel", "l2norm_bwd_kernel")


def short(name):
    for k in KEEP:
        if k in name:
            return k
    return None


def main():
    acc = defaultdict(lambda: defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    k = short(row.get("Kernel_Name", ""))
                    if k is None:
                        continue
                    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k in KEEP:
        if k not in acc:
            continue
        print(f"== {k}")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:34s} mean/dispatch {sum(v) / len(v):18.1f}   (n={len(v)})")
    # HBM traffic per launch, corrected as MI355X_MICROARCH.md "HBM" prescribes: FETCH_SIZE / WRITE_SIZE are KiB and on
    # gfx950 FETCH_SIZE counts half of a wide (16 B/lane) coalesced read stream -> x2.  WRITE_SIZE taken as reported.
    out = os.environ.get("PMC_TRAFFIC_JSON")
    if out:
        import json
        mean = lambda v: sum(v) / len(v)
        tr = {}
        for k in acc:
            if "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
                rd, wr = mean(acc[k]["FETCH_SIZE"]) * 1024 * 2, mean(acc[k]["WRITE_SIZE"]) * 1024
                tr[k] = dict(read_bytes=rd, write_bytes=wr, total_bytes=rd + wr)
        import hashlib
        here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        lib = os.path.join(here, "flash_cosine_sim_attention_amd", "libfcsa_hip.so")
        sha = hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None
        json.dump(dict(lib_sha256=sha, note="HBM bytes per launch on the bench workload (C3); FETCH_SIZE KiB x1024 x2 (gfx950 half-count "
                            "correction), WRITE_SIZE KiB x1024; separate --pmc passes (tools/gpu_pmc.sh)", kernels=tr),
                  open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
