#!/bin/bash
# A/B of an environment switch on the Trainer lines of bench.py inside ONE gpurun call: bash scripts/experiments/ab_trainer.sh DRN_TRAINER_PREFETCH 1 0
KEY=$1; shift
for rep in 1 2; do for v in "$@"; do
env $KEY=$v python bench.py --cpu-steps 0 --no-f32 --no-kernel-timing --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); t=d['trainer']; print('$KEY=$v', {k: v.get('ms_per_step') for k, v in t.items() if isinstance(v, dict) and 'graph' in k})"
done; done
