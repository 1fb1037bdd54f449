#!/bin/bash
# A/B of a drn_tune switch inside ONE gpurun call (boxes differ by +-3 %): bash scripts/experiments/ab_tune.sh exp0 0 1 [...]
KEY=$1; shift
REPS=${REPS:-2}
for rep in $(seq $REPS); do for v in "$@"; do
python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-kernel-timing --steps 40 --tune $KEY=$v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$KEY=$v', d['ms_per_step'])"
done; done
