mkdir -p gpurun_out
SER=$(rocm-smi --showserial 2>/dev/null | grep -i "serial number:" | head -1 | awk '{print $NF}')
LOG=gpurun_out/race_hunt10_$SER.log
echo "box $SER" | tee $LOG
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2 | tee -a $LOG
echo "== TINY forked, shipped library, 2 x 60 k replays bf16 + 30 k f32" | tee -a $LOG
for rep in 1 2; do timeout 600 python scripts/experiments/forked_race_hunt.py 60000 bf16 forked 2>&1 | grep "EVENT\|replays" | cut -c1-260 | tee -a $LOG; done
timeout 600 python scripts/experiments/forked_race_hunt.py 30000 f32 forked 2>&1 | grep "EVENT\|replays" | cut -c1-260 | tee -a $LOG
HUNT_SHAPE=32,256,4096 timeout 900 python scripts/experiments/forked_race_hunt.py 12000 bf16 forked 2>&1 | grep "EVENT\|replays" | cut -c1-260 | tee -a $LOG
echo "== step time: base library (exp3=8) vs shipped, auto policy; shipped with DRN_XCHG_CONFIRM=1" | tee -a $LOG
B="python bench.py --steps 300 --warmup 30 --cpu-steps 0 --no-f32 --no-other-configs --no-trainer --no-kernel-timing"
for T in 256 32; do for r in 1 2; do
  DRN_LIB_PATH=$PWD/drn_amd/libdrn_hip_base.so $B --T $T --tune exp3=8 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('T=$T base    ms_per_step', d['ms_per_step'])" | tee -a $LOG
  $B --T $T 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('T=$T shipped ms_per_step', d['ms_per_step'])" | tee -a $LOG
  DRN_XCHG_CONFIRM=1 $B --T $T 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('T=$T always  ms_per_step', d['ms_per_step'])" | tee -a $LOG
done; done
