#!/bin/bash
# in-box A/B: position embedding launched with the input preparation (beside the query encoder) instead of after the prop_fc GEMM
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2 3; do
  DRN_POS_EARLY=0 $B 2>/dev/null | get "T256 pos_early=0"
  DRN_POS_EARLY=1 $B 2>/dev/null | get "T256 pos_early=1"
done
for rep in 1 2; do
  DRN_POS_EARLY=0 $B --T 32 2>/dev/null | get "T32 pos_early=0"
  DRN_POS_EARLY=1 $B --T 32 2>/dev/null | get "T32 pos_early=1"
done
