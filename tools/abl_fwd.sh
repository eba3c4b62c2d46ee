# forward-kernel ablation timings (us per launch on C3), timing only
for v in 0 1 2 4 8 16 32 64 3 12 96 127; do
  lib=flash_cosine_sim_attention_amd/libfcsa_hip_abl$v.so; [ $v = 0 ] && lib=flash_cosine_sim_attention_amd/libfcsa_hip.so
  for kb in 32 96; do
    r=$(FCSA_LIB=$PWD/$lib FCSA_EXPERIMENT_LDS_KB=$kb python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['kernels']['fwd']['per_step_us'])")
    echo "ABL=$v LDS_KB=$kb fwd_us=$r"
  done
done
