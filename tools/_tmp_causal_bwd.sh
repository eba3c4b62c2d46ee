set -x
python -m pytest tests/test_gpu_split_forward.py -x -q -m gpu 2>&1 | tail -5
S="1,8,4096,64,4096:1,4,4096,64,4096:1,2,8192,64,8192:1,8,2048,64,2048:1,16,2048,64,2048:1,12,4096,64,4096:1,1,16384,64,16384:1,8,4096,128,4096:1,4,8192,128,8192:1,8,2048,128,2048:1,12,4096,128,4096:1,6,4096,32,4096"
python tools/split_sweep.py --dtype bf16 --causal --bwd dq --splits 1,2,3,4,5,6,8,12,16 --shape $S > gpurun_out/split_sweep_causal_dq.txt 2>&1
python tools/split_sweep.py --dtype bf16 --causal --bwd dkv --splits 1,2,3,4,5,6,8,12,16 --shape $S > gpurun_out/split_sweep_causal_dkv.txt 2>&1
tail -30 gpurun_out/split_sweep_causal_dq.txt gpurun_out/split_sweep_causal_dkv.txt
