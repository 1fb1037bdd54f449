#!/bin/bash
# round-5: issue order / phase order of the two-branch step (ForkedStep), one process each, interleaved
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config']['launch_ab'])"; }
for rep in 1 2; do
  DRN_FORK_ROTATE=0 $B 2>/dev/null | get "rot0 wf0 mf0"
  DRN_FORK_ROTATE=0 DRN_FORK_WGRADS_FIRST=1 $B 2>/dev/null | get "rot0 wf1 mf0"
  DRN_FORK_ROTATE=0 DRN_FORK_MAIN_FIRST=1 $B 2>/dev/null | get "rot0 wf0 mf1"
  DRN_FORK_ROTATE=0 DRN_FORK_MAIN_FIRST=1 DRN_FORK_WGRADS_FIRST=1 $B 2>/dev/null | get "rot0 wf1 mf1"
  DRN_FORK_ROTATE=1 $B 2>/dev/null | get "rot1 wf0 mf0"
  DRN_FORK_ROTATE=1 DRN_FORK_MAIN_FIRST=1 $B 2>/dev/null | get "rot1 wf0 mf1"
  DRN_FORK_ROTATE=1 DRN_FORK_MAIN_FIRST=1 DRN_FORK_WGRADS_FIRST=1 $B 2>/dev/null | get "rot1 wf1 mf1"
  DRN_FORK_ROTATE=1 DRN_FORK_WGRADS_FIRST=1 $B 2>/dev/null | get "rot1 wf1 mf0"
done
