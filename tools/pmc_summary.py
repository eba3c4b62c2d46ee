#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir> [<dir> ...]   (prints a table; test/profiling infrastructure only)"""
import csv
import glob
import os
import sys
from collections import defaultdict

KEEP = ("fwd_kernel", "bwd_dq_kernel", "bwd_dkv_kernel", "l2norm_kernel", "l2norm_bwd_kernel")


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


if __name__ == "__main__":
    main()
