#!/bin/bash
# Profile set of the 10 M-triangle configuration: one kernel-trace run of `bench.py --workload soup10m` and separate
# hardware-counter passes (never combined with tracing).
# Usage: tools/prof_soup.sh <out_dir>      (run from the repository root on the GPU box)
set -u
OUT=${1:-gpurun_out/prof_soup}
BENCH="--workload soup10m --steps 2 --warmup 1 --no-cpu-baseline"
PMCB="--workload soup10m --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-timing"
export TMPDIR=/tmp
mkdir -p "$OUT"
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python3 bench.py $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
grep -h "^{\"metric\"" "$OUT/trace.log" > "$OUT/bench_line.json"
DB=$(find "$OUT/trace" -name '*.db' | head -1)
python3 tools/kernel_stats.py "$DB" 60 > "$OUT/kernel_stats.txt" || exit 1
rm -rf "$OUT/trace"
echo "trace ok"
pass() {
  name=$1; shift
  timeout -s KILL 400 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 bench.py $PMCB > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
pass fetch FETCH_SIZE TCC_HIT_sum &&
pass write WRITE_SIZE TCC_MISS_sum TCC_REQ_sum &&
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY &&
pass sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM &&
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum &&
pass tatd TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_TC_STALL_sum GRBM_GUI_ACTIVE GRBM_COUNT
python3 tools/pmc_summary.py "$OUT" > "$OUT/pmc_summary.json"
python3 tools/pmc_traffic.py "$OUT/pmc_summary.json" "$OUT/bench_line.json" > "$OUT/pmc_traffic_soup.json"
for p in fetch write sq1 sq2 tcp tatd; do rm -rf "$OUT/$p"; done
echo done
