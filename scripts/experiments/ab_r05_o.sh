#!/bin/bash
# in-box A/B: conv0's forward split 4 ways on 256x128 tiles with the taps interleaved (DRN_KSPLIT_W4H=1) against the shipped unsplit path
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2 3; do
  DRN_KSPLIT_W4H=0 $B 2>/dev/null | get "T256 ksplit_w4h=0"
  DRN_KSPLIT_W4H=1 $B 2>/dev/null | get "T256 ksplit_w4h=1 (taps interleaved)"
done
for rep in 1 2; do
  DRN_KSPLIT_W4H=0 $B --T 32 2>/dev/null | get "T32 ksplit_w4h=0"
  DRN_KSPLIT_W4H=1 $B --T 32 2>/dev/null | get "T32 ksplit_w4h=1"
done
