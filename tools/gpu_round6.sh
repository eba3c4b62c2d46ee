#!/bin/bash
# Round-6 GPU sessions, by stage (one gpurun call runs a few of them).  usage: bash tools/gpu_round6.sh stage [stage ...]
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"; mkdir -p "$O"
kstats() {   # kstats NAME ITERS: rocprofv3 kernel-trace stats of tools/run_config.py NAME -> $O/kstats_NAME.csv (our kernels only)
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/ks_$1" -o k -- python "$R/tools/run_config.py" "$1" "$2" > "$O/ks_$1.log" 2>&1 )
  f=$(find "$O/ks_$1" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python - "$f" > "$O/kstats_$1.txt" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "fcsa::" in r["Name"]:
        n = r["Name"].replace("void fcsa::", "").split("(")[0]
        print(f"{n[:70]:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f}  max {float(r['MaxNs'])/1e3:8.2f}")
PY
  rm -rf "$O/ks_$1"; tail -n 1 "$O/ks_$1.log"; cat "$O/kstats_$1.txt" 2>/dev/null
}
for stage in "$@"; do
  echo "=== stage $stage"
  case "$stage" in
    quick)    # the tests that exercise this round's host-side changes
      timeout 900 python -m pytest tests/test_gpu_wide128.py tests/test_gpu_cabi_direct.py tests/test_gpu_misc.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 5 ;;
    kstats)   # true (un-instrumented) kernel durations of the other configs
      for n in C5 C2 C4 d128 d64; do kstats $n 60; done ;;
    bench)
      timeout 600 python bench.py > "$O/bench.log" 2>&1; tail -n 1 "$O/bench.log" > "$O/bench_line.json"; cut -c1-900 "$O/bench_line.json" ;;
    breakdown)
      timeout 300 python tools/kernel_breakdown.py d64 d128 C5 C4 C2 > "$O/breakdown.txt" 2>&1; grep -v amdgpu.ids "$O/breakdown.txt" ;;
    suite)
      python -c "import os, torch; n = torch.cuda.device_count(); print('torch.cuda.device_count() =', n, [torch.cuda.get_device_name(i) for i in range(n)], 'HIP_VISIBLE_DEVICES =', os.environ.get('HIP_VISIBLE_DEVICES'))" > "$O/pytest_gpu.log" 2>&1
      FCSA_TOL_LOG="$O/tol_log.jsonl" timeout 2400 python -m pytest tests -m gpu -q --maxfail=50 -p no:cacheprovider >> "$O/pytest_gpu.log" 2>&1; tail -n 8 "$O/pytest_gpu.log"
      python tools/tolerance_margins.py "$O/tol_log.jsonl" > "$O/tolerance_margins.txt" 2>&1; rm -f "$O/tol_log.jsonl" ;;
    fuzz_explore)   # exploratory seeds of round 5 (31, 32, 33: 408 configurations beyond the committed 136) under the model-derived allowances
      for sd in 31 32 33; do
        FCSA_FUZZ_SEED=$sd timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_config --maxfail=30 -p no:cacheprovider > "$O/fuzz_seed$sd.log" 2>&1
        echo "seed $sd: $(tail -n 1 "$O/fuzz_seed$sd.log")"; grep "^FAILED\|AssertionError: {" "$O/fuzz_seed$sd.log" | cut -c1-600 | head -n 12
      done ;;
    fuzz)
      timeout 1500 python -m pytest tests/test_gpu_fuzz.py -m gpu -q --maxfail=30 -p no:cacheprovider > "$O/fuzz.log" 2>&1; tail -n 3 "$O/fuzz.log"; grep "AssertionError: {" "$O/fuzz.log" | cut -c1-600 | head ;;
    probe_L02)
      timeout 300 python tools/fuzz_model_probe.py "{'id': 'L02', 'dtype': 'bf16', 'B': 1, 'H': 3, 'N': 1, 'M': 3283, 'D': 16, 'causal': False, 'mask': True, 'bias': False, 'bias_batch': True, 'single_kv': False, 'groups': 4, 'l2norm': True, 'scale': 16.0, 'seed': 758000559}" "{'id': 'L21', 'dtype': 'f16', 'B': 1, 'H': 3, 'N': 2961, 'M': 3, 'D': 32, 'causal': False, 'mask': True, 'bias': False, 'bias_batch': False, 'single_kv': False, 'groups': 2, 'l2norm': True, 'scale': 8.0, 'seed': 254545501}" 2>&1 | grep -v amdgpu.ids | tee "$O/fuzz_model_probe.txt" ;;
    ab_norm)   # finalize / l2norm variants: main = unrolled head loop + single-chunk l2norm on small grids; finold = round-5 code; finnt = nontemporal slab loads
      ( timeout 300 python tools/ab_libs.py --rounds 4 --shape 4,8,2048,128,1 --single-kv --groups 8 --scale 1 main finold finnt
        timeout 300 python tools/ab_libs.py --rounds 4 --dtype f16 --shape 1,8,1024,64,0,8192,0,1:4,8,1024,64,0 main finold finnt
        timeout 300 python tools/ab_libs.py --rounds 4 --shape 4,8,4096,64,1 main finold ) 2>&1 | grep -v amdgpu.ids | tee "$O/ab_norm.txt" ;;
    ablate)   # timing-only ablation: the masked (diagonal) tiles run the unmasked body (-DFCSA_ABL_NOMASK), DEV_ONLY builds of both arms
      timeout 400 python tools/ab_libs.py --rounds 5 --shape 4,8,4096,64,1:4,8,2048,64,1 dev0 devnomask 2>&1 | grep -v amdgpu.ids | tee "$O/ablate_nomask.txt" ;;
    ab_ringsep)   # dK/dV ring form: 16-row epilogue scratch behind the ring + next pass requested from the epilogue (main) vs round 4's plan (nosep)
      ( timeout 400 python tools/ab_libs.py --rounds 5 --shape 4,8,4096,64,1:4,8,2048,64,1:4,8,4096,64,0 main nosep
        timeout 300 python tools/ab_libs.py --rounds 4 --shape 4,8,2048,128,1 --single-kv --groups 8 --scale 1 main nosep
        timeout 300 python tools/ab_libs.py --rounds 4 --shape 4,8,4096,128,1:2,8,2048,96,1:2,8,2048,32,1 main nosep
        timeout 300 python tools/ab_libs.py --rounds 4 --dtype f16 --shape 4,8,4096,64,1 main nosep ) 2>&1 | grep -v amdgpu.ids | tee "$O/ab_ring_sep.txt" ;;
    parity_bwd)   # the backward-heavy parity files on the new library
      timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_misc.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -n 4 ;;
    fuzz_more)   # six more exploratory seeds (816 configurations) under the model-derived bars + the fwd3-vs-lean differential fuzz through the debug knob
      for sd in 41 42 43 44 45 46; do
        FCSA_FUZZ_SEED=$sd timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -k random_config --maxfail=30 -p no:cacheprovider > "$O/fuzz_seed$sd.log" 2>&1
        echo "FCSA_FUZZ_SEED=$sd: $(tail -n 1 "$O/fuzz_seed$sd.log")"; grep "AssertionError: {" "$O/fuzz_seed$sd.log" | cut -c1-500 | head -n 8
      done | tee "$O/fuzz_more.txt"
      timeout 900 python tools/fwd3_fuzz.py --n 300 --seed 7 2>&1 | grep -v amdgpu.ids | tail -n 6 | tee "$O/fwd3_fuzz.txt" ;;
    probe_more)   # the three marginal exceedances of exploratory seeds 41 / 42 / 46 (tiny bf16 D = 16 problems): kernel vs model
      ONE_DTYPE=1 timeout 300 python tools/fuzz_model_probe.py "{'id': 'c42', 'dtype': 'bf16', 'B': 2, 'H': 1, 'N': 5, 'M': 5, 'D': 16, 'causal': True, 'mask': False, 'bias': True, 'bias_batch': False, 'single_kv': False, 'groups': 2, 'l2norm': True, 'scale': 10.0, 'seed': 840320648}" "{'id': 'c46', 'dtype': 'bf16', 'B': 1, 'H': 1, 'N': 5, 'M': 178, 'D': 16, 'causal': True, 'mask': False, 'bias': False, 'bias_batch': False, 'single_kv': True, 'groups': 2, 'l2norm': True, 'scale': 8.0, 'seed': 158148712}" "{'id': 'L10', 'dtype': 'bf16', 'B': 1, 'H': 1, 'N': 130, 'M': 1449, 'D': 16, 'causal': True, 'mask': False, 'bias': False, 'bias_batch': True, 'single_kv': False, 'groups': 1, 'l2norm': True, 'scale': 16.0, 'seed': 592557245}" 2>&1 | grep -v amdgpu.ids | tee "$O/fuzz_model_probe_more.txt"
      timeout 600 python tools/fwd3_fuzz.py --n 300 --seed 7 --only 0 > /dev/null 2>&1; timeout 900 python tools/fwd3_fuzz.py --n 300 --seed 7 2>&1 | grep -v amdgpu.ids | tail -n 4 | tee "$O/fwd3_fuzz.txt" ;;
    ab_rowsum)   # forward row sums: v_dot2c on the rounded pairs (product) vs plain adds of the un-rounded values vs none (timing only)
      timeout 400 python tools/ab_libs.py --rounds 5 --shape 4,8,4096,64,1:4,8,4096,64,0 dev0 devrowadd devnorowsum 2>&1 | grep -v amdgpu.ids | tee "$O/ab_rowsum.txt" ;;
    ablate_x)   # timing only: dK/dV without the exponentials and products of its X phase -- how much of the one-wave-per-SIMD forms (C5, small D = 128 grids) is the X phase that no partner wave covers?  (the d*nox variants were a local edit of dkv_tile_pipe that was not kept; result: profiles/r06_ablate_x.txt)
      ( timeout 300 python tools/ab_libs.py --rounds 4 --shape 4,8,2048,128,1 --single-kv --groups 8 --scale 1 d128base d128nox
        timeout 300 python tools/ab_libs.py --rounds 4 --shape 4,8,4096,128,1:2,8,2048,128,1 d128base d128nox
        timeout 300 python tools/ab_libs.py --rounds 4 --shape 4,8,4096,64,1 d64base d64nox ) 2>&1 | grep -v amdgpu.ids | tee "$O/ablate_x.txt" ;;
    *) echo "unknown stage $stage" ;;
  esac
done
