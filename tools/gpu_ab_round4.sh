#!/bin/bash
# HISTORICAL (round 4): the -DFCSA_* form switches this script builds variants with were removed from the product sources in round 5 (they
# are named constants in the .hip files now).  It documents how profiles/r04_ab_*.txt were produced -- check out the round-4 tree
# (commit 8243d98) to re-run it.
# Round 4 evidence: the same-process A/Bs behind the kernel decisions of the round.  Two steps:
#   tools/gpu_ab_round4.sh build     (build container: development variants, bf16 D = 64 / 128 only, seconds each)
#   tools/gpu_ab_round4.sh run       (GPU box: writes gpurun_out/r04_*.txt; copy them to profiles/)
set -u
cd "$(dirname "$0")/.."
R03="-DFCSA_DKV_RING=0 -DFCSA_PRIO_BWD=0 -DFCSA_DQ_SPLIT0=0"
if [ "${1:-}" = build ]; then
  ( tools/build_dev.sh r03eq "$R03" & tools/build_dev.sh ring "-DFCSA_DKV_RING=1 -DFCSA_PRIO_BWD=0 -DFCSA_DQ_SPLIT0=0" &
    tools/build_dev.sh prio "-DFCSA_DKV_RING=0 -DFCSA_PRIO_BWD=1 -DFCSA_DQ_SPLIT0=0" & tools/build_dev.sh ringprio "-DFCSA_DQ_SPLIT0=0" & wait
    tools/build_dev.sh final "" & tools/build_dev.sh fwdp1 "-DFCSA_PRIO_FWD=1" & tools/build_dev.sh fwdp2 "-DFCSA_PRIO_FWD=2" &
    tools/build_dev.sh fwds4p "-DFCSA_FWD_SUB=4 -DFCSA_PRIO_FWD=3" & wait
    tools/build_dev.sh d128 "-DFCSA_DEV_D=128" & tools/build_dev.sh d128leanprio "-DFCSA_DEV_D=128 -DFCSA_PRIO_LEAN=1" &
    tools/build_dev.sh bar_r03eq "$R03 -DFCSA_TRACE_BAR" & tools/build_dev.sh bar_final "-DFCSA_TRACE_BAR" & wait
    tools/build_dev.sh wg_r03eq "$R03 -DFCSA_TRACE_WG" & tools/build_dev.sh wg_final "-DFCSA_TRACE_WG" & wait ) 2>&1 | grep -v hip-link
  exit 0
fi
O=gpurun_out; mkdir -p $O; P=flash_cosine_sim_attention_amd
f() { grep -v amdgpu.ids; }
echo "== dK/dV ring, backward priority, split first dQ stage (C3, bf16)"
python tools/ab_libs.py --rounds 7 --shape 4,8,4096,64,1 r03eq ring prio ringprio final 2>&1 | f > $O/r04_ab_ring_prio.txt; tail -n 7 $O/r04_ab_ring_prio.txt
echo "== the same on other shapes (non-causal, short, many heads)"
python tools/ab_libs.py --rounds 5 --shape 4,8,4096,64,0:8,8,2048,64,1:16,8,1024,64,0,1024,0,1 r03eq final 2>&1 | f > $O/r04_ab_ring_prio_shapes.txt; grep -v "^shape\|max" $O/r04_ab_ring_prio_shapes.txt
echo "== forward priority forms (all rejected)"
python tools/ab_libs.py --rounds 5 --shape 4,8,4096,64,1 final fwdp1 fwdp2 fwds4p 2>&1 | f > $O/r04_ab_fwd_priority.txt; tail -n 5 $O/r04_ab_fwd_priority.txt
echo "== lean dK/dV (D = 128) with an in-block priority split (rejected)"
python tools/ab_libs.py --rounds 4 --shape 4,8,4096,128,1:8,8,2048,128,1 d128 d128leanprio 2>&1 | f > $O/r04_ab_lean_priority.txt; grep -v "^shape\|max" $O/r04_ab_lean_priority.txt
echo "== constant inputs (no operand bit toggles): how much of the step is the power limit"
( echo "# random inputs"; python tools/ab_libs.py --rounds 3 --shape 4,8,4096,64,1 r03eq final 2>&1 | f | tail -n 3
  echo "# all inputs = 1.0 (FCSA_AB_FILL=1)"; FCSA_AB_FILL=1 python tools/ab_libs.py --rounds 3 --shape 4,8,4096,64,1 r03eq final 2>&1 | f | tail -n 3 ) > $O/r04_ab_constant_inputs.txt; cat $O/r04_ab_constant_inputs.txt
echo "== barrier waits per wave"
( for t in bar_r03eq bar_final; do echo "== $t"; FCSA_LIB=$PWD/$P/libfcsa_hip_$t.so python tools/trace_bar.py 2>&1 | f; done ) > $O/r04_trace_bar.txt; cat $O/r04_trace_bar.txt
echo "== per-workgroup pass timing"
( for t in wg_r03eq wg_final; do echo "== $t"; FCSA_LIB=$PWD/$P/libfcsa_hip_$t.so ITERS=300 python tools/trace_wg.py 2>&1 | f | grep -v "XCD\|slowest\|fastest"; done ) > $O/r04_trace_wg_ab.txt; cat $O/r04_trace_wg_ab.txt
echo "== gap probe"; ./tools/probes/gap_probe > $O/r04_gap_probe.txt 2>&1; tail -n 24 $O/r04_gap_probe.txt
