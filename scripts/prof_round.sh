#!/bin/bash
# One round's rocprofv3 evidence for bench.py on the GPU box: clean kernel trace (+ the step's launch sequence), then the two
# PMC passes for the fabric-traffic table -- each PMC pass on its own, only --kernel-trace beside it.
# usage (via gpurun): bash scripts/prof_round.sh r02_a
set -u
TAG=${1:-r02_a}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 30"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/bench_under_trace.json 2> /dev/null
PMCCMD="python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 4 --warmup 2"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o f -- $PMCCMD > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o w -- $PMCCMD > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --dump-gemms $OUT/gemms.json > $OUT/bench_default.json 2> $OUT/bench_default.err
python scripts/rocprof_summary.py $OUT/trace/t_results.db 35 > $OUT/kernel_stats.txt
python scripts/rocprof_step.py $OUT/trace/t_results.db > $OUT/step_kernel_sequence.txt
python scripts/gemm_table.py $OUT/trace/t_results.db $OUT/gemms.json 2500 $OUT/gemm_in_graph.json > $OUT/gemm_table.txt 2> $OUT/gemm_table.err
python scripts/pmc_hbm_table.py $OUT/trace/t_results.db $OUT/fetch/f_results.db $OUT/write/w_results.db $OUT/pmc_traffic.json > $OUT/hbm_kernels.txt 2> $OUT/hbm_kernels.err
rm -rf $OUT/trace $OUT/fetch $OUT/write
tail -3 $OUT/step_kernel_sequence.txt; head -30 $OUT/hbm_kernels.txt; cat $OUT/hbm_kernels.err | tail -5; cat $OUT/bench_default.json | head -c 400
