#!/bin/bash
# Kernel times of one headline frame (frames one at a time), one line per kernel: tools/prof_frame.sh <tag> [bench args]
# RT_AMD_LIB selects another build of the library (tools/build_variant.sh).
set -u
TAG=${1:?tag}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-exhaustive --no-other-arithmetic --frames-in-flight 1 --steps 8 --warmup 2 "$@" > $OUT.log 2>&1 || { echo failed; tail -5 $OUT.log; exit 1; }
echo "== $TAG"
python3 tools/frame_trace.py $OUT 2 sum
