set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r2_prof_$1
rm -rf $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline --no-exhaustive --frames-in-flight 1 --steps 8 --warmup 2 > $OUT.log 2>&1 || { echo failed; tail -5 $OUT.log; exit 1; }
python3 tools/frame_trace.py $OUT 2 | awk '{n[$1]+=$2; c[$1]++} END {for (k in n) printf "%-24s %8.1f us  x%d\n", k, n[k], c[k]}' | sort
python3 tools/frame_trace.py $OUT 2 | tail -1
