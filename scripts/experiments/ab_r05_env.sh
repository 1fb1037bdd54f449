#!/bin/bash
# round-5: ROCm runtime switches that could lower the per-kernel floor inside a replayed hipGraph
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2; do
  $B 2>/dev/null | get "base"
  HIP_FORCE_DEV_KERNARG=1 $B 2>/dev/null | get "HIP_FORCE_DEV_KERNARG=1"
  HIP_FORCE_DEV_KERNARG=0 $B 2>/dev/null | get "HIP_FORCE_DEV_KERNARG=0"
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $B 2>/dev/null | get "GRAPH_PACKET_CAPTURE=1"
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 $B 2>/dev/null | get "GRAPH_PACKET_CAPTURE=0"
  GPU_MAX_HW_QUEUES=8 $B 2>/dev/null | get "GPU_MAX_HW_QUEUES=8"
  GPU_MAX_HW_QUEUES=2 $B 2>/dev/null | get "GPU_MAX_HW_QUEUES=2"
  HSA_ENABLE_SDMA=0 $B 2>/dev/null | get "HSA_ENABLE_SDMA=0"
  $B --T 32 2>/dev/null | get "T32 base"
  HIP_FORCE_DEV_KERNARG=1 $B --T 32 2>/dev/null | get "T32 HIP_FORCE_DEV_KERNARG=1"
  DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 $B --T 32 2>/dev/null | get "T32 GRAPH_PACKET_CAPTURE=1"
done
env | grep -i -E "^HIP_|^HSA_|^GPU_|^ROC|CLR" | head
