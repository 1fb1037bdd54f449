#!/bin/bash
# in-box A/B: workgroup budget of the one-launch BatchNorm backward
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2; do
  DRN_BN_BWD_ONE=0 $B 2>/dev/null | get "T256 one=0"
  DRN_BN_BWD_ONE=1 $B 2>/dev/null | get "T256 one=1 maxwg=512"
  DRN_BN_BWD_ONE=1 DRN_BN1_MAXWG=256 $B 2>/dev/null | get "T256 one=1 maxwg=256"
  DRN_BN_BWD_ONE=1 DRN_BN1_MAXWG=448 $B 2>/dev/null | get "T256 one=1 maxwg=448"
done
