#!/bin/bash
# Two SQ counter passes over one short bench run, summarised per kernel: tools/pmc_quick.sh <tag>
set -u
TAG=${1:?tag}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmcq_$TAG
rm -rf $OUT; mkdir -p $OUT
pass() {
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exhaustive --frames-in-flight 1 > "$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -3 "$OUT/$name.log"; return 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY &&
pass sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM GRBM_GUI_ACTIVE &&
python3 tools/pmc_summary.py $OUT
