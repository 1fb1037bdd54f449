#!/bin/bash
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2; do
  for v in 0 48 64 96 128 192; do
    DRN_FORK_PREP_THROTTLE=$v $B 2>/dev/null | get "prep_throttle=$v"
  done
done
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_forked_thr
mkdir -p $OUT
DRN_FORK_PREP_THROTTLE=128 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --steps 30 > $OUT/bench_under_trace.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python scripts/rocprof_forked.py $OUT/trace/t_results.db > $OUT/forked_timeline.txt 2> $OUT/err.txt
rm -rf $OUT/trace
