#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_graph_gpu.py -x -q 2>&1 | tail -15
for f in "" "--single-stream"; do
timeout 300 python bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --steps 40 $f 2>&1 | python -c "
import sys,json
t=sys.stdin.read()
try:
  d=json.loads(t.strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['config']['loss_cls'], d['config']['launch'][:60])
except Exception as e: print('ERR', t[-2000:])"
done
