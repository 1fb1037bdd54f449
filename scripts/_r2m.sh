#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --tb=line -rf 2>&1 | grep -E "passed|failed|^FAILED|Error" | cut -c1-300
timeout 300 python bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --steps 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['loss_cls'])"
