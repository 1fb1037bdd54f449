#!/bin/bash
# A/B of an environment switch inside ONE gpurun call (boxes differ by +-3 %): bash scripts/experiments/ab_env.sh DRN_KSPLIT_WGS 512 256
# prints the bench step (T = 256) and the Trainer's T = 32 / T = 256 graph steps
KEY=$1; shift
for rep in 1 2; do for v in "$@"; do
env $KEY=$v python bench.py --cpu-steps 0 --no-f32 --no-kernel-timing --steps 40 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); t=d.get('trainer') or {}; print('$KEY=$v', d['ms_per_step'], 'T32', (t.get('T32_graph') or {}).get('ms_per_step'), 'T256', (t.get('T256_graph') or {}).get('ms_per_step'))"
done; done
