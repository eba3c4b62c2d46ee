#!/bin/bash
# One GPU-box session of round 3: parity tests -> bench line -> interleaved A/B of library variants -> per-kernel breakdown of
# the other configs -> host-time accounting.  Everything lands in gpurun_out/.   usage: tools/gpu_session.sh [ab tags...]
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
TAGS="${*:-main}"
echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; tail -n 12 gpurun_out/pytest_gpu.log | cut -c1-300
echo "== bench =="; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log > gpurun_out/bench_line.json; cut -c1-3000 gpurun_out/bench_line.json
echo "== A/B $TAGS =="; timeout 300 python tools/ab_libs.py --rounds 6 $TAGS > gpurun_out/ab.txt 2>&1; tail -n 8 gpurun_out/ab.txt
echo "== breakdown =="; timeout 200 python tools/kernel_breakdown.py C5 C4 d128 C2bias > gpurun_out/breakdown.txt 2>&1; tail -n 30 gpurun_out/breakdown.txt
echo "== host overhead =="; for n in 128 512; do timeout 100 python tools/host_overhead.py $n > gpurun_out/host_$n.txt 2>&1; tail -n 14 gpurun_out/host_$n.txt; done
