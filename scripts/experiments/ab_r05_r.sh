#!/bin/bash
# in-box A/B: the query-gate backward of backbone levels 0 / 1 inside the BatchNorm backward launch (DRN_GATE_BN_FUSE)
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2 3; do
  DRN_GATE_BN_FUSE=0 $B 2>/dev/null | get "T256 gate_bn_fuse=0"
  DRN_GATE_BN_FUSE=1 $B 2>/dev/null | get "T256 gate_bn_fuse=1"
done
