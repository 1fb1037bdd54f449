get() { python -c "import json,sys; d=json.loads(sys.stdin.readline()); t=d['trainer']; print('$1', {k: v.get('ms_per_step') for k, v in t.items() if isinstance(v, dict) and 'graph' in k})"; }
B="python bench.py --cpu-steps 0 --no-f32 --no-kernel-timing --no-other-configs --steps 20"
DRN_TRAINER_FORKED=0 $B 2>/dev/null | get "forked=0 q=4"
DRN_TRAINER_FORKED=1 $B 2>/dev/null | get "forked=1 q=4"
GPU_MAX_HW_QUEUES=8 DRN_TRAINER_FORKED=1 $B 2>/dev/null | get "forked=1 q=8"
GPU_MAX_HW_QUEUES=8 DRN_TRAINER_FORKED=0 $B 2>/dev/null | get "forked=0 q=8"
GPU_MAX_HW_QUEUES=16 DRN_TRAINER_FORKED=1 $B 2>/dev/null | get "forked=1 q=16"
