# forked-step timeline under a given environment: bash forked_trace_env.sh <tag> VAR=VAL ...
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r05_forked_$TAG
mkdir -p $OUT
env "$@" timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --no-trainer --no-other-configs --steps 30 > $OUT/bench_under_trace.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python scripts/rocprof_forked.py $OUT/trace/t_results.db > $OUT/forked_timeline.txt 2> $OUT/err.txt
rm -rf $OUT/trace
tail -1 $OUT/forked_timeline.txt
