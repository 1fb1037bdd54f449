# usage: tail_trace3.sh "<ENV=..>" tag  -> gpurun_out/r05_seq_<tag>.txt (kernel sequence of one replayed linear-graph step)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_seqtmp_$2
mkdir -p $OUT
env $1 timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --no-forked --steps 30 ${3:-} > $OUT/bench.json 2> /dev/null
python $GRAFT_REPO_ROOT/scripts/rocprof_step.py $OUT/trace/t_results.db > $GRAFT_REPO_ROOT/gpurun_out/r05_seq_$2.txt 2>/dev/null
rm -rf $OUT
