#!/bin/bash
# round-5 A/Bs with existing kernels (one process each, interleaved): conv->BN fusion at T = 32; the 4-slot ring for short-K launches
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for rep in 1 2; do
  DRN_BN_FUSE=0 $B --T 32 2>/dev/null | get "T32 bn_fuse=0"
  DRN_BN_FUSE=1 $B --T 32 2>/dev/null | get "T32 bn_fuse=1"
  $B 2>/dev/null | get "T256 base"
  $B --tune nt_deep2=512 --tune nt_deep_ks=16 2>/dev/null | get "T256 deep2=512 ks=16"
  $B --tune nt_deep2=512 --tune nt_deep_ks=8 2>/dev/null | get "T256 deep2=512 ks=8"
  $B --tune nt_deep2=512 --tune nt_deep_ks=24 2>/dev/null | get "T256 deep2=512 ks=24"
  $B --T 32 --tune nt_deep2=512 --tune nt_deep_ks=16 2>/dev/null | get "T32 deep2=512 ks=16"
done
