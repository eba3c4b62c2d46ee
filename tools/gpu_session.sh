#!/bin/bash
# One GPU-box session of round 3: parity tests -> bench line -> interleaved A/B of library variants over a few shapes.
# Everything lands in gpurun_out/.   usage: tools/gpu_session.sh "<shape[:shape...]>" [ab tags...]   (SKIP_TESTS=1 / SKIP_BENCH=1)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
SHAPES="${1:-4,8,4096,64,1}"; shift || true
TAGS="${*:-main}"
if [ -z "${SKIP_TESTS:-}" ]; then echo "== pytest gpu =="; timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; tail -n 12 gpurun_out/pytest_gpu.log | cut -c1-300; fi
if [ -z "${SKIP_BENCH:-}" ]; then echo "== bench =="; timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log > gpurun_out/bench_line.json; cut -c1-3500 gpurun_out/bench_line.json; fi
echo "== A/B $TAGS =="; timeout 600 python tools/ab_libs.py --rounds 5 --shape "$SHAPES" $TAGS > gpurun_out/ab.txt 2>&1; tail -n 24 gpurun_out/ab.txt
