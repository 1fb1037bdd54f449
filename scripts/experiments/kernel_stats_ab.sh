#!/bin/bash
# per-kernel average durations of the step under several builds of the library (rocprofv3 --kernel-trace --stats over graph replays):
# bash scripts/experiments/kernel_stats_ab.sh <name> [<name> ...] -> gpurun_out/kstats_<name>.txt   ("shipped" = the tree's library)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
mkdir -p $R/gpurun_out
for n in "$@"; do
  if [ "$n" = shipped ]; then unset DRN_LIB_PATH; else export DRN_LIB_PATH=$R/scripts/experiments/libdrn_hip_$n.so; fi
  rm -rf /tmp/ks_$n
  (cd $R && rocprofv3 --kernel-trace --stats -d /tmp/ks_$n -o t -- python bench.py --steps 40 --warmup 10 --cpu-steps 0 --no-f32 --no-other-configs --no-trainer --no-kernel-timing --no-forked > /dev/null 2>&1)
  (cd $R && python scripts/rocprof_summary.py /tmp/ks_$n/t_results.db 45 > $R/gpurun_out/kstats_$n.txt)
done
