#!/bin/bash
# Copies the outputs of tools/gpu_profile_round6.sh (merged into gpurun_out/ by gpurun) to profiles/<tag>_*, runs the clock cross-check and
# regenerates DESIGN.md's measurement table.   usage: tools/collect_evidence.sh [tag]
set -eu
TAG="${1:-r06}"; O=gpurun_out; P=profiles
cp $O/bench_line.json $P/${TAG}_bench_line.json
cp $O/kernel_stats.csv $P/${TAG}_rocprofv3_kernel_stats.csv
cp $O/pmc_summary.txt $P/${TAG}_pmc_summary.txt
cp $O/pmc_traffic.json $P/${TAG}_pmc_traffic.json
{ echo "rocprofv3 --kernel-trace --stats of 'python tools/run_config.py NAME 60' (20 warm-up + 60 un-instrumented steps per configuration: short runs, the chip is still ramping -- read min next to avg; shapes: tools/kernel_breakdown.py SHAPES)"; cat $O/kstats_all.txt; } > $P/${TAG}_kstats.txt
cp $O/dims.txt $P/${TAG}_dims.txt
cp $O/benchmark_causal.txt $P/${TAG}_benchmark_causal.txt
cp $O/benchmark_full.txt $P/${TAG}_benchmark_full.txt
cp $O/breakdown.txt $P/${TAG}_breakdown.txt
cp $O/fwd_dyn_ab.txt $P/${TAG}_fwd_dyn_ab.txt
cp $O/host_128.txt $P/${TAG}_host_overhead_n128.txt
cp $O/host_512.txt $P/${TAG}_host_overhead_n512.txt
[ -f $O/trace_wg.txt ] && cp $O/trace_wg.txt $P/${TAG}_trace_wg.txt
python tools/pmc_clock_crosscheck.py $TAG > $P/${TAG}_pmc_clock_crosscheck.txt
python tools/design_table.py $TAG > /dev/null
tail -n 5 $P/${TAG}_pmc_clock_crosscheck.txt
