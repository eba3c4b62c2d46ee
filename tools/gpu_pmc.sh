#!/bin/bash
# PMC passes (separate runs, --pmc only; never combined with tracing flags) over a short bench run.
set -u
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out/pmc"; mkdir -p "$O"
export KERNEL_STATS_CSV="${KERNEL_STATS_CSV:-}"
CMD="python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra-configs"
run() { n=$1; shift; ( cd /tmp && timeout 300 rocprofv3 --pmc "$@" -d "$O/$n" -o p --output-format csv -- $CMD > "$O/$n.log" 2>&1 ); tail -n 2 "$O/$n.log" | cut -c1-200; }
( cd /tmp && rocprofv3 -L > "$O/counters_avail.txt" 2>&1 ); grep -c . "$O/counters_avail.txt"
run p1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run p2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_SALU
run p3 FETCH_SIZE GRBM_GUI_ACTIVE
run p4 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run p5 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVE32_INSTS SQ_INSTS_VALU_TRANS
PMC_TRAFFIC_JSON="$R/gpurun_out/pmc_traffic.json" python "$R/tools/pmc_summary.py" "$O/p1" "$O/p2" "$O/p3" "$O/p4" "$O/p5" > "$R/gpurun_out/pmc_summary.txt" 2>&1
cat "$R/gpurun_out/pmc_summary.txt"
find "$O" -name "*.db" -delete; find "$O" -name "*.csv" -size +8M -delete
