# step time with the GEMM exchanges' confirmation forced: 0 (off) / 2 (sc1 read-back) / 1 (returning atomics) / auto, one box
mkdir -p gpurun_out
LOG=gpurun_out/ab_r05_confirm_modes.log
: > $LOG
ARGS="--steps 300 --warmup 30 --cpu-steps 0 --no-f32 --no-other-configs --no-trainer --no-kernel-timing"
for T in 256 32; do for r in 1 2; do for c in 0 2 1 auto; do
  DRN_XCHG_CONFIRM=$c python bench.py $ARGS --T $T 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('T=$T confirm=$c ms_per_step', d['ms_per_step'])" | tee -a $LOG
done; done; done
