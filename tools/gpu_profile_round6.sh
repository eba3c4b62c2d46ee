#!/bin/bash
# The round-6 evidence session (one gpurun call; outputs under gpurun_out/, copied to profiles/r06_* afterwards):
#   rocprofv3 kernel-trace stats of the bench command -> five PMC passes (own runs, --pmc only) + the per-kernel derived numbers
#   (matrix-pipe occupancy in real clocks, effective clock: tools/pmc_summary.py) + the traffic JSON keyed on the SOURCE hash ->
#   bench line (reads that JSON for roofline.traffic / mfma_busy) -> per-config kernel-trace durations -> dims sweep ->
#   benchmark.py tables -> event-pair breakdown (regime stated in the file) -> host overhead -> per-workgroup pass trace.
# usage: tools/gpu_profile_round6.sh [tag]
set -u
TAG="${1:-r06}"
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
echo "== rocprofv3 kernel stats (bench command)"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_stats" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra-configs > "$O/rocprof_stats.log" 2>&1 )
f=$(find "$O/prof_stats" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" "$O/kernel_stats.csv"; head -n 6 "$f" | cut -c1-200; }
rm -rf "$O/prof_stats"
echo "== pmc"
KERNEL_STATS_CSV="$O/kernel_stats.csv" bash tools/gpu_pmc.sh > "$O/pmc.log" 2>&1; grep -A6 "^== derived" "$O/pmc_summary.txt"
cp "$O/pmc_traffic.json" "$R/profiles/${TAG}_pmc_traffic.json"
echo "== bench"; timeout 900 python bench.py > "$O/bench.log" 2>&1; tail -n 1 "$O/bench.log" > "$O/bench_line.json"; cut -c1-700 "$O/bench_line.json"
echo "== per-config kernel-trace durations"
bash tools/gpu_round6.sh kstats > "$O/kstats.log" 2>&1; cat "$O"/kstats_*.txt > "$O/kstats_all.txt" 2>/dev/null; grep "us per step" "$O/kstats.log"
echo "== dims sweep"; bash tools/dims_sweep.sh > "$O/dims.log" 2>&1; tail -n 44 "$O/dims.txt" | cut -c1-160
echo "== benchmark.py --causal"; timeout 600 python benchmark.py --causal --dtypes bfloat16,float16 > "$O/benchmark_causal.txt" 2>&1; tail -n 18 "$O/benchmark_causal.txt"
echo "== benchmark.py (non-causal, forward+backward)"; timeout 600 python benchmark.py --dtypes bfloat16,float16,float32 > "$O/benchmark_full.txt" 2>&1; tail -n 5 "$O/benchmark_full.txt"
echo "== breakdown"; timeout 300 python tools/kernel_breakdown.py d64 d128 d96 C5 C5s8 C4 C2 C2bias d64f32 f16s16 f16s16d128 C5s16 > "$O/breakdown.txt" 2>&1; grep -v amdgpu.ids "$O/breakdown.txt" | tail -n 70
echo "== forward: constant exponent shift vs online per-row reference"; timeout 200 python tools/fwd_dyn_ab.py > "$O/fwd_dyn_ab.txt" 2>&1; grep -v amdgpu.ids "$O/fwd_dyn_ab.txt" | tail -n 12
echo "== host overhead"; for n in 128 512; do timeout 100 python tools/host_overhead.py $n > "$O/host_$n.txt" 2>&1; tail -n 13 "$O/host_$n.txt"; done
echo "== per-workgroup pass timing (FCSA_TRACE_WG build)"
[ -f flash_cosine_sim_attention_amd/libfcsa_hip_wg.so ] && FCSA_LIB="$R/flash_cosine_sim_attention_amd/libfcsa_hip_wg.so" ITERS=3000 timeout 120 python tools/trace_wg.py > "$O/trace_wg.txt" 2>&1 && grep -v "XCD\|slowest\|fastest\|amdgpu.ids" "$O/trace_wg.txt" | tail -n 16
