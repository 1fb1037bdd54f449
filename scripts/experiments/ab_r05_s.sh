#!/bin/bash
# in-box A/B: non-temporal accesses to the Adam moments (two library builds: drn_amd/libdrn_hip.so = OPT_NT 1, libdrn_hip_nt0.so = 0)
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2 3; do
  DRN_LIB_PATH=$PWD/drn_amd/libdrn_hip_nt0.so $B 2>/dev/null | get "T256 nt=0"
  $B 2>/dev/null | get "T256 nt=1"
done
for rep in 1 2; do
  DRN_LIB_PATH=$PWD/drn_amd/libdrn_hip_nt0.so $B --T 32 2>/dev/null | get "T32 nt=0"
  $B --T 32 2>/dev/null | get "T32 nt=1"
done
