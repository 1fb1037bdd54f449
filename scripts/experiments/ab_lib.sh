#!/bin/bash
# in-box A/B of two builds of the library: interleaved bench.py runs (linear graph, the step only), ms/step each
# usage: bash scripts/experiments/ab_lib.sh <other.so> [rounds]
OTHER=$1; R=${2:-3}
ARGS="--steps 300 --warmup 30 --cpu-steps 0 --no-f32 --no-other-configs --no-trainer --no-kernel-timing --no-forked"
for r in $(seq 1 $R); do
  python bench.py $ARGS 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('shipped', d['ms_per_step'])"
  DRN_LIB_PATH=$OTHER python bench.py $ARGS 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('other  ', d['ms_per_step'])"
done
