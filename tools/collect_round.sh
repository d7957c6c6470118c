#!/bin/bash
# Everything profiles/ holds for one state of the code, in one go (on the GPU box):
#   tools/collect_round.sh <tag>      -> gpurun_out/<tag>/...
set -u
TAG=${1:?tag}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
bash tools/prof_round.sh $TAG > gpurun_out/$TAG/prof_round.log 2>&1 || { echo "prof_round failed"; tail -5 gpurun_out/$TAG/prof_round.log; exit 1; }
bash tools/pmc_passes.sh gpurun_out/$TAG/pmc > gpurun_out/$TAG/pmc.log 2>&1 || { echo "pmc failed"; tail -5 gpurun_out/$TAG/pmc.log; exit 1; }
python3 tools/pmc_summary.py gpurun_out/$TAG/pmc > gpurun_out/$TAG/pmc_summary_culled.txt 2>&1 || { echo "pmc summary failed"; exit 1; }
timeout -k 10 400 python3 tools/bench_configs.py --out gpurun_out/$TAG/configs.json > gpurun_out/$TAG/configs.log 2>&1 || { echo "configs failed"; tail -5 gpurun_out/$TAG/configs.log; exit 1; }
timeout -k 10 300 python3 tools/bench_large_mesh.py > gpurun_out/$TAG/large_mesh.log 2>&1 || { echo "large mesh failed"; tail -5 gpurun_out/$TAG/large_mesh.log; exit 1; }
timeout -k 10 400 python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { echo "bench failed"; tail -5 gpurun_out/$TAG/bench.err; exit 1; }
tail -1 gpurun_out/$TAG/bench.json
