#!/bin/bash
# Short counter passes to compare traversal variants.
set -u
OUT=${1:-gpurun_out/pmc3}; shift || true
ARGS=${@:---steps 1 --warmup 0 --spp 2 --no-cpu-baseline --no-kernel-timing}
export TMPDIR=/tmp
mkdir -p "$OUT"
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
pass a1 SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS &&
pass a2 TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum &&
pass a3 TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum &&
pass a4 TD_TD_BUSY_sum TD_TC_STALL_sum
