# conv0's forward inside the replayed step, 128x128 split-2 kernel vs gemm_nt_w4h_kernel split-4: two kernel traces on ONE box
cd /tmp && export TMPDIR=/tmp
for v in 0 1 0 1; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_conv0_$v
  mkdir -p $OUT
  DRN_KSPLIT_W4H=$v timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 30 > $OUT/bench.json 2> /dev/null
  python $GRAFT_REPO_ROOT/scripts/rocprof_step.py $OUT/trace/t_results.db > $OUT/seq.txt 2>/dev/null
  rm -rf $OUT/trace
  echo "ksplit_w4h=$v: $(sed -n 21,23p $OUT/seq.txt | cut -c1-60 | tr '\n' '|') $(tail -1 $OUT/seq.txt)"
done
