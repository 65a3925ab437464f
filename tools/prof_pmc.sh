#!/bin/bash
# Hardware-counter passes over a short bench run (one rocprofv3 --pmc run per pass,
# never combined with tracing).  Usage: tools/prof_pmc.sh <out_dir> [bench args...]
set -u
OUT=${1:-gpurun_out/pmc}; shift || true
ARGS=${@:---steps 1 --warmup 0 --spp 2 --no-cpu-baseline --no-kernel-timing}
export TMPDIR=/tmp
mkdir -p "$OUT"
pass() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py $ARGS > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY &&
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM &&
pass tcc1 FETCH_SIZE TCC_HIT_sum &&
pass tcc2 WRITE_SIZE TCC_MISS_sum TCC_REQ_sum &&
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum
