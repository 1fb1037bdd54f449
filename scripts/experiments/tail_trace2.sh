cd /tmp && export TMPDIR=/tmp
for tag in 0 1; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_tail$tag
  mkdir -p $OUT
  DRN_EXT_SUMSQ=$tag timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 30 > $OUT/bench.json 2> /dev/null
  python $GRAFT_REPO_ROOT/scripts/rocprof_step.py $OUT/trace/t_results.db > $OUT/seq.txt 2>/dev/null
  rm -rf $OUT/trace
done
