#!/bin/bash
# Round 6, third session: causal split forms of the backward.  usage: gpu_round6b.sh <stages...>   (outputs under gpurun_out/)
#   splittests  tests/test_gpu_split_forward.py
#   ab          step-level same-process A/B main vs base (libfcsa_hip_base.so = the library before the causal split forms) on causal small grids
#   sweep_dq / sweep_dkv   tools/split_sweep.py --causal --bwd ... on the sweep build
#   suite       whole GPU suite
mkdir -p gpurun_out
S="1,8,4096,64,4096:1,4,4096,64,4096:1,2,8192,64,8192:1,8,2048,64,2048:1,16,2048,64,2048:1,12,4096,64,4096:1,1,16384,64,16384:1,8,4096,128,4096:1,4,8192,128,8192:1,8,2048,128,2048:1,12,4096,128,4096:1,6,4096,32,4096"
for st in "$@"; do
  case $st in
    splittests) python -m pytest tests/test_gpu_split_forward.py -x -q -m gpu 2>&1 | tail -5 ;;
    ab) python tools/ab_libs.py --rounds 3 --shape 1,4,4096,64,1:1,8,4096,64,1:1,8,2048,64,1:1,8,4096,128,1:1,4,8192,128,1:1,2,8192,64,1:1,8,4000,64,1,4500:2,8,4096,64,1:4,8,4096,64,1 base main > gpurun_out/ab_causal_split_bwd.txt 2>&1; tail -40 gpurun_out/ab_causal_split_bwd.txt ;;
    sweep_dq) python tools/split_sweep.py --dtype bf16 --causal --bwd dq --splits 1,2,3,4,5,6,8,12,16 --shape $S > gpurun_out/split_sweep_causal_dq.txt 2>&1 ;;
    sweep_dkv) python tools/split_sweep.py --dtype bf16 --causal --bwd dkv --splits 1,2,3,4,5,6,8,12,16 --shape $S > gpurun_out/split_sweep_causal_dkv.txt 2>&1 ;;
    suite) python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/suite_r06c.txt; cat gpurun_out/suite_r06c.txt ;;
    fuzz) for s in 61 62; do FCSA_FUZZ_SEED=$s python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3 | sed "s/^/FCSA_FUZZ_SEED=$s: /" >> gpurun_out/fuzz_r06c.txt; done; cat gpurun_out/fuzz_r06c.txt ;;
  esac
done
