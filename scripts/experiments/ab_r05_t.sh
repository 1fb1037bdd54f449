#!/bin/bash
# in-box A/B at T = 32: the pre-touch of prop_fc's bf16 weight copy (DRN_TOUCH_W) now that Adam leaves it in the Infinity Cache
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60 --T 32"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2 3; do
  DRN_TOUCH_W=1 $B 2>/dev/null | get "T32 touch=1"
  DRN_TOUCH_W=0 $B 2>/dev/null | get "T32 touch=0"
done
