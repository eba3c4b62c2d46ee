#!/bin/bash
# One GPU-box session: diagnostics -> parity tests -> bench -> rocprofv3 kernel stats.  Everything lands in gpurun_out/.
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo =="; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6
echo "== diag =="; timeout 600 python tools/gpu_diag.py > gpurun_out/diag.log 2>&1; tail -n 70 gpurun_out/diag.log
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; tail -n 40 gpurun_out/pytest_gpu.log
echo "== bench =="; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -n 3 gpurun_out/bench.log
echo "== rocprofv3 kernel stats =="
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_stats" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_stats.log" 2>&1 )
tail -n 5 gpurun_out/rocprof_stats.log
find gpurun_out/prof_stats -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" gpurun_out/kernel_stats.csv; head -n 12 "$f" | cut -c1-250; }
# keep the merged-back payload small
find gpurun_out/prof_stats -name "*kernel_trace.csv" -size +20M -delete
