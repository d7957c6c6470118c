#!/bin/bash
# Collects rocprofv3 hardware counters for the bench, one pass per run.
# Usage (on the GPU box): tools/pmc_passes.sh <out-dir> [bench args]
set -u
OUT=${1:?out dir}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY &&
pass sq2 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE &&
pass tcc_rd FETCH_SIZE &&
pass tcc_wr WRITE_SIZE &&
pass tcc_hit TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
echo done
