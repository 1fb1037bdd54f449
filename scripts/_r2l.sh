#!/bin/bash
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r2m
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2m/prof -o q -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_step.py gpurun_out/r2m/prof/q_results.db > gpurun_out/r2m/seq.txt 2>&1; tail -1 gpurun_out/r2m/seq.txt
rm -rf gpurun_out/r2m/prof
