#!/bin/bash
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2; do
  for v in 0 128 256 512 1024; do
    DRN_FORK_PREP_THROTTLE=$v $B 2>/dev/null | get "prep_throttle=$v"
  done
done
