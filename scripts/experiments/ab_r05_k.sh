#!/bin/bash
# in-box A/B: BatchNorm backward in one launch (DRN_BN_BWD_ONE)
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2 3; do
  DRN_BN_BWD_ONE=0 $B 2>/dev/null | get "T256 one=0"
  DRN_BN_BWD_ONE=1 $B 2>/dev/null | get "T256 one=1"
done
for rep in 1 2; do
  DRN_BN_BWD_ONE=0 $B --T 32 2>/dev/null | get "T32 one=0"
  DRN_BN_BWD_ONE=1 $B --T 32 2>/dev/null | get "T32 one=1"
done
