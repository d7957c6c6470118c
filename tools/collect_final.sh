#!/bin/bash
# The measurements a change of the fast arithmetic's triangle stage moves, in one go (on the GPU box):
#   tools/collect_final.sh <tag>   -> gpurun_out/<tag>/{bench.json,frame_fast.txt,configs_fast.json,scaling_fast.txt}, gpurun_out/<tag>_pmc_summary_c3.txt,
#                                     profiles/kernel_counters.json (on the box; copied to gpurun_out/)
set -u
TAG=${1:?tag}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/$TAG
bash tools/pmc_round.sh $TAG c3 > gpurun_out/$TAG/pmc_round.log 2>&1 || { echo "pmc failed"; tail -5 gpurun_out/$TAG/pmc_round.log; exit 1; }
timeout -k 10 400 python3 tools/bench_configs.py --arithmetic 1 --out gpurun_out/$TAG/configs_fast.json > gpurun_out/$TAG/configs_fast.log 2>&1 || { echo "configs failed"; exit 1; }
timeout -k 10 300 python3 tools/scaling_probe.py 1 2 4 8 three serial fast > gpurun_out/$TAG/scaling_fast.txt 2>&1 || { echo "scaling failed"; exit 1; }
bash tools/prof_frame.sh ${TAG}_fast > gpurun_out/$TAG/frame_fast.txt 2>&1 || { echo "frame failed"; exit 1; }
timeout -k 10 400 python3 bench.py > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err || { echo "bench failed"; tail -5 gpurun_out/$TAG/bench.err; exit 1; }
tail -1 gpurun_out/$TAG/bench.json | cut -c1-260
grep "^share" gpurun_out/$TAG/scaling_fast.txt
cat gpurun_out/$TAG/frame_fast.txt
grep -E "^\| " gpurun_out/$TAG/configs_fast.log
