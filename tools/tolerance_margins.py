"""Reduce a tolerance log (tests/tolerances.py, FCSA_TOL_LOG) to one line per (label, dtype): comparisons, worst measured / bar ratio and
the case that produced it.  The bars of tests/tolerances.py are set to <= 1.5 x the worst measured value of their class (round 4 review).
usage: python tools/tolerance_margins.py gpurun_out/tol_log.jsonl [more logs ...] > profiles/r05_tolerance_margins.txt"""
import collections
import json
import sys


def main():
    worst = collections.defaultdict(lambda: [0, 0.0, None, 0.0, 0.0])      # n, ratio, case, measured, bar
    for fn in sys.argv[1:]:
        for line in open(fn):
            r = json.loads(line)
            key = (r["label"], r["dtype"])
            w = worst[key]
            w[0] += 1
            ratio = r["measured"] / r["bar"] if r["bar"] > 0 else (0.0 if r["measured"] <= 0 else float("inf"))
            if ratio >= w[1]:
                w[1], w[2], w[3], w[4] = ratio, r.get("case"), r["measured"], r["bar"]
    print("%-48s %-5s %7s %9s %11s %11s  %s" % ("comparison class", "dtype", "n", "max m/bar", "measured", "bar", "worst case"))
    for (label, dt), (n, ratio, case, m, b) in sorted(worst.items()):
        print("%-48s %-5s %7d %9.3f %11.3e %11.3e  %s" % (label, dt, n, ratio, m, b, case))


if __name__ == "__main__":
    main()
