#!/bin/bash
# round-5: conv0's forward on gemm_nt_w4h_kernel with the in-launch split (DRN_KSPLIT_W4H) and the w4h tile threshold at T = 32
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for rep in 1 2 3; do
  DRN_KSPLIT_W4H=0 $B 2>/dev/null | get "T256 ksplit_w4h=0"
  DRN_KSPLIT_W4H=1 $B 2>/dev/null | get "T256 ksplit_w4h=1"
done
for rep in 1 2; do
  $B --T 32 --tune nt_w4h=0 2>/dev/null | get "T32 w4h=0"
  $B --T 32 --tune nt_w4h=128 2>/dev/null | get "T32 w4h=128"
  $B --T 32 --tune nt_w4h=160 2>/dev/null | get "T32 w4h=160"
done
