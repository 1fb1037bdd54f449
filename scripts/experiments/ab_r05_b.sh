#!/bin/bash
# round-5: gemm_nt_w4h_kernel (256 x 128 tiles) on / off, one process each, interleaved
B="python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --no-kernel-timing --steps 60"
get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['config'].get('loss_cls'))"; }
for rep in 1 2 3; do
  $B --tune nt_w4h=0 2>/dev/null | get "T256 w4h=0"
  $B --tune nt_w4h=160 2>/dev/null | get "T256 w4h=128"
done
python bench.py --cpu-steps 0 --no-f32 --no-trainer --no-other-configs --steps 20 --verbose 2>&1 >/dev/null | grep -E "gemm_nt|gemm_wgrad" | head -30
