#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV output (counter_collection.csv) per kernel: mean counter value per dispatch, plus -- per kernel --
the matrix-pipe occupancy in real clocks and the effective shader clock (round-5 review item 8):
    cycles per XCD     = GRBM_GUI_ACTIVE / 8                  (the counter is summed over the 8 XCDs)
    mfma_busy          = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x cycles per XCD)       (32 busy cycles per v_mfma_f32_32x32x16)
    effective clock    = cycles per XCD / kernel duration     (duration: rocprofv3 --kernel-trace --stats of the same command,
                                                               KERNEL_STATS_CSV; the PMC passes themselves run ~3 % slower)
usage: pmc_summary.py <dir> [<dir> ...]   (prints a table; test/profiling infrastructure only)
env: PMC_TRAFFIC_JSON=<out.json>  KERNEL_STATS_CSV=<kernel_stats.csv of the kernel-trace run>  FCSA_CUS=<compute units, default 256>"""
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
    # kernel durations of the un-profiled kernel-trace run (ns), by short kernel name
    dur_ns = {}
    stats_csv = os.environ.get("KERNEL_STATS_CSV")
    if stats_csv and os.path.exists(stats_csv):
        with open(stats_csv) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get("Name", ""))
                if k is not None and k not in dur_ns:      # (sorted by total time: the first match is the bench workload's instantiation)
                    dur_ns[k] = float(row["AverageNs"])
    cus, xcds = int(os.environ.get("FCSA_CUS", "256")), 8
    derived = {}
    print("== derived (per launch): matrix-pipe occupancy in real clocks, effective clock")
    for k in KEEP:
        if k not in acc or "GRBM_GUI_ACTIVE" not in acc[k]:
            continue
        cyc = sum(acc[k]["GRBM_GUI_ACTIVE"]) / len(acc[k]["GRBM_GUI_ACTIVE"]) / xcds
        d = {"cycles_per_xcd": round(cyc, 1)}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in acc[k]:
            busy = sum(acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc[k]["SQ_VALU_MFMA_BUSY_CYCLES"])
            d["mfma_busy"] = round(busy / (4.0 * cus * cyc), 4)
        if k in dur_ns:
            d["kernel_us_kernel_trace"] = round(dur_ns[k] / 1e3, 2)
            d["effective_clock_ghz"] = round(cyc / dur_ns[k], 3)
        derived[k] = d
        print(f"   {k:20s} " + "  ".join(f"{a} {b}" for a, b in d.items()))
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
        sys.path.insert(0, here)
        from flash_cosine_sim_attention_amd import _lib
        json.dump(dict(lib_sha256=sha, src_sha256=_lib.source_sha256(), derived=derived, note="HBM bytes per launch on the bench workload (C3); FETCH_SIZE KiB x1024 x2 (gfx950 half-count "
                            "correction), WRITE_SIZE KiB x1024; separate --pmc passes (tools/gpu_pmc.sh)", kernels=tr),
                  open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
