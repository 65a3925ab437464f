#!/bin/bash
# Memory-pipeline counter passes (vector L1 / texture addresser) for the traversal kernel.
set -u
OUT=${1:-gpurun_out/pmc2}; shift || true
ARGS=${@:---steps 1 --warmup 0 --spp 2 --no-cpu-baseline --no-kernel-timing}
export TMPDIR=/tmp
mkdir -p "$OUT"
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
pass m1 SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_WAVES GRBM_GUI_ACTIVE GRBM_TA_BUSY &&
pass m2 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum &&
pass m3 TD_TD_BUSY_sum TD_TC_STALL_sum &&
pass m4 TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum &&
pass m5 TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum
