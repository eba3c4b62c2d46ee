#!/bin/bash
# Head-dim x dtype sweep at N = 4096: benchmark.py's fused-op time and TFLOP/s, forward and forward+backward,
# causal, B4 H8.  Writes gpurun_out/dims.txt.
set -u
mkdir -p gpurun_out; : > gpurun_out/dims.txt
for d in 16 32 64 96 128; do
  for mode in "--only-forwards" ""; do
    echo "== dim_head $d ${mode:-forward+backward}" >> gpurun_out/dims.txt
    timeout 300 python benchmark.py --causal --dim-head $d --seq-lens 4096 --num-times 20 --no-baseline $mode 2>&1 | grep "seq_len\|^float\|^bfloat" >> gpurun_out/dims.txt
  done
done
cat gpurun_out/dims.txt
