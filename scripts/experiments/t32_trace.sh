cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_t32
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --T 32 --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 30 > $OUT/bench_under_trace.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python scripts/rocprof_step.py $OUT/trace/t_results.db > $OUT/step_kernel_sequence.txt 2> $OUT/err.txt
rm -rf $OUT/trace
tail -3 $OUT/step_kernel_sequence.txt; head -c 300 $OUT/bench_under_trace.json
