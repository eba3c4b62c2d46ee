#!/bin/bash
# One GPU-box session that produces the evidence files of a round (copied to profiles/<tag>_* afterwards):
#   five PMC passes (+ traffic JSON tied to the library hash) -> bench line -> rocprofv3 kernel stats -> dims sweep ->
#   benchmark.py tables -> breakdown of the other configs -> host-time accounting.      usage: tools/gpu_profile_round.sh [tag]
set -u
TAG="${1:-r03}"
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
# PMC passes first: their traffic JSON (tied to the library's sha256) is what the bench line's roofline.traffic is read from
echo "== pmc"; bash tools/gpu_pmc.sh > gpurun_out/pmc.log 2>&1; tail -n 3 gpurun_out/pmc.log
cp gpurun_out/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
echo "== bench"; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log > gpurun_out/bench_line.json; cut -c1-600 gpurun_out/bench_line.json
echo "== rocprofv3 kernel stats"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_stats" -o bench -- python "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra-configs > "$R/gpurun_out/rocprof_stats.log" 2>&1 )
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" gpurun_out/kernel_stats.csv; head -n 8 "$f" | cut -c1-220; }
find gpurun_out/prof_stats -name "*kernel_trace.csv" -size +20M -delete
echo "== dims sweep"; bash tools/dims_sweep.sh > gpurun_out/dims.log 2>&1; tail -n 44 gpurun_out/dims.txt | cut -c1-160
echo "== benchmark.py --causal"; timeout 600 python benchmark.py --causal --dtypes bfloat16,float16 > gpurun_out/benchmark_causal.txt 2>&1; tail -n 18 gpurun_out/benchmark_causal.txt
echo "== benchmark.py (non-causal, forward+backward)"; timeout 600 python benchmark.py --dtypes bfloat16,float16,float32 > gpurun_out/benchmark_full.txt 2>&1; tail -n 5 gpurun_out/benchmark_full.txt
echo "== breakdown"; timeout 200 python tools/kernel_breakdown.py d64 d128 d96 C5 C5s8 C4 C2 C2bias d64f32 f16s16 f16s16d128 C5s16 > gpurun_out/breakdown.txt 2>&1; tail -n 50 gpurun_out/breakdown.txt
echo "== forward: constant exponent shift vs online per-row reference"; timeout 200 python tools/fwd_dyn_ab.py > gpurun_out/fwd_dyn_ab.txt 2>&1; grep -v amdgpu.ids gpurun_out/fwd_dyn_ab.txt | tail -n 12
echo "== host overhead"; for n in 128 512; do timeout 100 python tools/host_overhead.py $n > gpurun_out/host_$n.txt 2>&1; tail -n 13 gpurun_out/host_$n.txt; done
echo "== per-workgroup pass timing (needs the FCSA_TRACE_WG build next to the library)"
[ -f flash_cosine_sim_attention_amd/libfcsa_hip_wg.so ] && FCSA_LIB="$R/flash_cosine_sim_attention_amd/libfcsa_hip_wg.so" ITERS=3000 timeout 120 python tools/trace_wg.py > gpurun_out/trace_wg.txt 2>&1 && grep -v "XCD\|slowest\|fastest" gpurun_out/trace_wg.txt | tail -n 16
