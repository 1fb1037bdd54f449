#!/bin/bash
# timeline of the Trainer's two-branch step vs bench.py's (rocprofv3 kernel trace of both, scripts/rocprof_forked.py)
OUT=$GRAFT_REPO_ROOT/gpurun_out/tgap
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PRE_DEV=${PRE_DEV:-1099511627776} ONLY_FORKED=1 rocprofv3 --kernel-trace -d $OUT/tr -o t -- python $GRAFT_REPO_ROOT/scripts/experiments/trainer_gap.py > $OUT/trainer.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_forked.py $OUT/tr/t_results.db > $OUT/trainer_forked_timeline.txt 2>&1
rm -rf $OUT/tr $OUT/be
tail -n 2 $OUT/trainer_forked_timeline.txt; cat $OUT/trainer.log | grep round
