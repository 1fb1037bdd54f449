#!/bin/bash
# in-box comparison of several builds of the library: interleaved bench.py runs (linear graph, the step only), ms/step each
# usage: bash scripts/experiments/ab_multi.sh <rounds> <name> [<name> ...]   (name -> scripts/experiments/libdrn_hip_<name>.so; "shipped" = the tree's)
R=$1; shift
ARGS="--steps 300 --warmup 30 --cpu-steps 0 --no-f32 --no-other-configs --no-trainer --no-kernel-timing ${AB_EXTRA:---no-forked}"
for r in $(seq 1 $R); do
  for n in "$@"; do
    if [ "$n" = shipped ]; then unset DRN_LIB_PATH; else export DRN_LIB_PATH=$PWD/scripts/experiments/libdrn_hip_$n.so; fi
    python bench.py $ARGS 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('$n', d['ms_per_step'])"
  done
done
