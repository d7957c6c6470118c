#!/bin/bash
# Kernel trace of a rank's share of the headline frame (tools/scaling_probe.py N serial), one frame launch by launch:
#   tools/prof_share.sh N
set -u
N=${1:?ranks}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/share_$N
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 tools/scaling_probe.py $N serial > $OUT.log 2>&1 || { echo failed; tail -5 $OUT.log; exit 1; }
grep "^share" $OUT.log
python3 tools/frame_trace.py $OUT 20
