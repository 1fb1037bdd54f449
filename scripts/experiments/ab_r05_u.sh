#!/bin/bash
# in-box A/B (again, after the non-temporal optimizer accesses): optimizer-first order of the two-branch step
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2 3; do
  $B 2>/dev/null | get "classic"
  DRN_BENCH_FORK_ROTATE=1 $B 2>/dev/null | get "optimizer-first"
done
