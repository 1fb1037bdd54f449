#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_optim_gpu.py -x -q 2>&1 | tail -2 | cut -c1-300
for i in 1 2; do
timeout 300 python bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --steps 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['loss_cls'])"
done
