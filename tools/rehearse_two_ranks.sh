#!/bin/bash
# Two bench.py ranks on ONE GPU over gloo (RCCL refuses two ranks on one device): rehearses the
# multi-rank code path of bench.py.  Logs go to gpurun_out/rank*.log; a rank that is still running
# after 120 s is aborted and leaves a Python traceback there.
#   tools/rehearse_two_ranks.sh [bench arguments, e.g. --config c4_glass --steps 3]
mkdir -p gpurun_out
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 WORLD_SIZE=2 RT_BENCH_BACKEND=gloo PYTHONFAULTHANDLER=1
pids=()
for r in 0 1; do
  RANK=$r LOCAL_RANK=$r timeout -s ABRT -k 10 120 python -u -X faulthandler bench.py --gpus 2 --steps 10 --warmup 2 --no-exhaustive --no-cpu-baseline "$@" > gpurun_out/rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=$?; done
tail -n 25 gpurun_out/rank0.log
echo "---- rank 1"
tail -n 12 gpurun_out/rank1.log
exit $rc
