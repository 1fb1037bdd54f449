cd /tmp && export TMPDIR=/tmp
for cfgv in "DRN_EXT_SUMSQ=0 DRN_WGRAD_DEFER=1" "DRN_EXT_SUMSQ=1 DRN_WGRAD_DEFER=1" "DRN_EXT_SUMSQ=1 DRN_WGRAD_DEFER=0"; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_tail
  mkdir -p $OUT
  env $cfgv timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 30 > $OUT/bench.json 2> /dev/null
  python $GRAFT_REPO_ROOT/scripts/rocprof_step.py $OUT/trace/t_results.db > $OUT/seq.txt 2>/dev/null
  rm -rf $OUT/trace
  echo "== $cfgv"; grep -E "wgrad_reduce|sumsq|adam" $OUT/seq.txt | cut -c1-90; tail -1 $OUT/seq.txt
done
