#!/bin/bash
# One GPU-box session that produces the evidence files of a round (copied to profiles/ afterwards):
#   bench line, rocprofv3 kernel stats, five PMC passes (+ traffic JSON tied to the library hash), configs vs SDPA, benchmark.py table
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
# PMC passes first: their traffic JSON (tied to the library's sha256) is what the bench line's roofline.traffic is read from
echo "== pmc"; bash tools/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -n 3 gpurun_out/pmc.log
cp gpurun_out/pmc_traffic.json profiles/r02_pmc_traffic.json
echo "== bench"; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log > gpurun_out/bench_line.json; cut -c1-400 gpurun_out/bench_line.json
echo "== rocprofv3 kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > "$R/gpurun_out/rocprof_stats.log" 2>&1 )
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" gpurun_out/kernel_stats.csv; head -n 8 "$f" | cut -c1-220; }
find gpurun_out/prof_stats -name "*kernel_trace.csv" -size +20M -delete
echo "== configs"; timeout 600 python tools/bench_configs.py > gpurun_out/configs.log 2>&1; tail -n 8 gpurun_out/configs.log | cut -c1-250
echo "== benchmark.py --causal"; timeout 600 python benchmark.py --causal --dtypes bfloat16,float16 > gpurun_out/benchmark_causal.txt 2>&1; tail -n 9 gpurun_out/benchmark_causal.txt
echo "== probes"; timeout 120 tools/probes/atomic_probe > gpurun_out/atomic_probe.txt 2>&1; timeout 120 tools/probes/coissue_probe > gpurun_out/coissue_probe.txt 2>&1; tail -n 3 gpurun_out/coissue_probe.txt
echo "== per-workgroup pass timing (needs the FCSA_TRACE_WG build next to the library)"
[ -f flash_cosine_sim_attention_amd/libfcsa_hip_wg.so ] && FCSA_LIB="$R/flash_cosine_sim_attention_amd/libfcsa_hip_wg.so" ITERS=3000 timeout 120 python tools/trace_wg.py > gpurun_out/trace_wg.txt 2>&1; grep -v "XCD\|slowest\|fastest" gpurun_out/trace_wg.txt | tail -n 16
echo "== bias configurations, per kernel"; timeout 100 python tools/kernel_breakdown.py C2bias T5bias > gpurun_out/bias_breakdown.txt 2>&1; tail -n 14 gpurun_out/bias_breakdown.txt
