#!/bin/bash
# Counter passes of one configuration and what bench.py reads of them: tools/pmc_round.sh <tag> <config>
#   -> gpurun_out/<tag>_pmc_<config>/..., gpurun_out/<tag>_pmc_summary_<config>.txt, gpurun_out/kernel_counters.json
set -u
TAG=${1:?tag}; CFG=${2:?config}
cd "$GRAFT_REPO_ROOT"
D=gpurun_out/${TAG}_pmc_$CFG
export BENCH_ARGS="--config $CFG --no-exhaustive"
bash tools/pmc_passes.sh $D > $D.log 2>&1 || { tail -n 5 $D.log; exit 1; }
python3 tools/pmc_summary.py $D > gpurun_out/${TAG}_pmc_summary_$CFG.txt || exit 1
SPECS="$D"
if [ "$CFG" = c3 ]; then
  export BENCH_ARGS="--config c3 --no-other-arithmetic"
  bash tools/pmc_passes.sh ${D}_exhaustive > ${D}_exhaustive.log 2>&1 || { tail -n 5 ${D}_exhaustive.log; exit 1; }
  { echo; echo "== the same passes over a run with exhaustive-scan frames (k_mesh)"; python3 tools/pmc_summary.py ${D}_exhaustive | grep -E "k_mesh|^=="; } >> gpurun_out/${TAG}_pmc_summary_$CFG.txt
  SPECS="$D ${D}_exhaustive:k_mesh"
fi
python3 tools/kernel_counters.py $CFG ${TAG}_pmc_summary_$CFG.txt $SPECS || exit 1
cp profiles/kernel_counters.json gpurun_out/kernel_counters.json
