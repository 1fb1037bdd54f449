#!/bin/bash
# Quick GPU check of a work-in-progress tree (via gpurun): the GPU parity tests, a default-shape bench line without the CPU /
# f32 legs, and the launch sequence of one replayed step from a clean rocprofv3 kernel trace.
# usage: gpurun -- bash scripts/gpu_check.sh <tag> [pytest -k expression]
set -u
TAG=${1:-chk}
KEXPR=${2:-}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ -n "$KEXPR" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu -k "$KEXPR" 2>&1 | tail -25 > $OUT/tests.txt
else
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $OUT/tests.txt
fi
tail -25 $OUT/tests.txt
timeout 900 python bench.py --cpu-steps 0 --no-f32 --dump-gemms $OUT/gemms.json > $OUT/bench.json 2> $OUT/bench.err
head -c 600 $OUT/bench.json; echo; python -c "import json,sys; d=json.load(open(sys.argv[1])); print(json.dumps({k: d.get(k) for k in ('eager_ms_per_step', 'trainer')}))" $OUT/bench.json; tail -3 $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 30 > $OUT/bench_under_trace.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python scripts/rocprof_step.py $OUT/trace/t_results.db > $OUT/step_kernel_sequence.txt 2> $OUT/step.err
python scripts/rocprof_summary.py $OUT/trace/t_results.db 40 > $OUT/kernel_stats.txt 2>> $OUT/step.err
python scripts/gemm_table.py $OUT/trace/t_results.db $OUT/gemms.json > $OUT/gemm_table.txt 2>> $OUT/step.err
rm -rf $OUT/trace
tail -2 $OUT/step_kernel_sequence.txt
