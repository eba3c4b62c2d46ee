#!/bin/bash
# One GPU-box session of round 5.  Everything lands in gpurun_out/.   usage: tools/gpu_round5.sh <steps...>   steps: suite0 wide ab suite bench prof
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for step in "$@"; do
  case "$step" in
    suite0)   # the whole GPU suite on the PREVIOUS forward forms (FCSA_FWD_WIDE128=0), with the tolerance log
      rm -f gpurun_out/tol_log0.jsonl
      FCSA_FWD_WIDE128=0 FCSA_TOL_LOG=$PWD/gpurun_out/tol_log0.jsonl timeout 1500 python -m pytest tests -m gpu -q --maxfail=200 -p no:cacheprovider --deselect tests/test_gpu_wide128.py > gpurun_out/pytest_suite0.log 2>&1
      tail -n 15 gpurun_out/pytest_suite0.log | cut -c1-300 ;;
    canary)   # one small launch of the new kernel under a short timeout: a hang here switches the new form off for the rest of the session
      timeout 240 python -m pytest tests/test_gpu_wide128.py -m gpu -q -x -k "causal_square or full_one_tile" -p no:cacheprovider > gpurun_out/pytest_canary.log 2>&1
      rc=$?; tail -n 25 gpurun_out/pytest_canary.log | cut -c1-400
      if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "CANARY TIMED OUT: disabling the wide form"; export FCSA_FWD_WIDE128=0; export FCSA_CANARY_FAILED=1; fi ;;
    wide)     # the new kernel's own tests
      if [ -n "${FCSA_CANARY_FAILED:-}" ]; then echo "skipped (canary)"; continue; fi
      rm -f gpurun_out/tol_log_wide.jsonl
      FCSA_TOL_LOG=$PWD/gpurun_out/tol_log_wide.jsonl timeout 900 python -m pytest tests/test_gpu_wide128.py tests/test_gpu_wide_forward.py -m gpu -q --maxfail=100 -p no:cacheprovider > gpurun_out/pytest_wide.log 2>&1
      tail -n 40 gpurun_out/pytest_wide.log | cut -c1-400 ;;
    ab)
      if [ -n "${FCSA_CANARY_FAILED:-}" ]; then echo "skipped (canary)"; continue; fi
      timeout 600 python tools/fwd3_ab.py --variants ${AB_VARIANTS:-0 on} > gpurun_out/fwd3_ab.txt 2>&1; cat gpurun_out/fwd3_ab.txt
      ;;
    abl)      # ablations of the wide form's tile loop (development build of the library: wrong results, timing only) + phase trace
      timeout 600 python tools/fwd3_ab.py --shapes 4,8,4096,4096,0 4,8,4096,4096,1 --variants ${ABL_VARIANTS:-0 r4 rx ry rz rw rv ru} --rounds 3 > gpurun_out/fwd3_abl.txt 2>&1; cat gpurun_out/fwd3_abl.txt
      FCSA_LIB=$PWD/flash_cosine_sim_attention_amd/libfcsa_hip_trace.so timeout 300 python tools/trace_fwd3.py 0 > gpurun_out/trace_fwd3.txt 2>&1
      FCSA_LIB=$PWD/flash_cosine_sim_attention_amd/libfcsa_hip_trace.so timeout 300 python tools/trace_fwd3.py 1 >> gpurun_out/trace_fwd3.txt 2>&1; cat gpurun_out/trace_fwd3.txt ;;
    trace)
      FCSA_LIB=$PWD/flash_cosine_sim_attention_amd/libfcsa_hip_trace.so timeout 300 python tools/trace_fwd3.py 0 > gpurun_out/trace_fwd3.txt 2>&1
      FCSA_LIB=$PWD/flash_cosine_sim_attention_amd/libfcsa_hip_trace.so timeout 300 python tools/trace_fwd3.py 1 >> gpurun_out/trace_fwd3.txt 2>&1; grep -v amdgpu.ids gpurun_out/trace_fwd3.txt | grep 'pass\|wave' ;;
    suite)    # the whole GPU suite as shipped
      rm -f gpurun_out/tol_log.jsonl
      python -c "import os, torch; n = torch.cuda.device_count(); print('torch.cuda.device_count() =', n, [torch.cuda.get_device_name(i) for i in range(n)], 'HIP_VISIBLE_DEVICES =', os.environ.get('HIP_VISIBLE_DEVICES'), '(the two-device test of tests/test_gpu_misc.py runs only with >= 2)')" > gpurun_out/pytest_gpu.log 2>&1
      FCSA_TOL_LOG=$PWD/gpurun_out/tol_log.jsonl timeout 1800 python -m pytest tests -m gpu -q --maxfail=200 -p no:cacheprovider >> gpurun_out/pytest_gpu.log 2>&1
      tail -n 15 gpurun_out/pytest_gpu.log | cut -c1-300 ;;
    bench)
      timeout 600 python bench.py > gpurun_out/bench.log 2>&1; tail -n 1 gpurun_out/bench.log > gpurun_out/bench_line.json; cut -c1-6000 gpurun_out/bench_line.json ;;
    prof)
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_stats" -o bench -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_stats.log" 2>&1 )
      f=$(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp "$f" gpurun_out/kernel_stats.csv; head -n 14 "$f" | cut -c1-250; }
      find gpurun_out/prof_stats -name "*kernel_trace.csv" -size +20M -delete ;;
  esac
done
