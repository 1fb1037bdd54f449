# the exchange fix against the library of the commit before it (drn_amd/libdrn_hip_base.so, built from 5942b34), same box
mkdir -p gpurun_out
LOG=gpurun_out/ab_r05_race.log
: > $LOG
cat > /tmp/bench_base.py <<'PY'
import os, sys, runpy
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import drn_amd.ops as o
# (round-5 script: compared against a library of that round; the round-6 ABI (version 9) no longer loads such a library)
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.environ["GRAFT_REPO_ROOT"], "bench.py"), run_name="__main__")
PY
ARGS="--steps 300 --warmup 30 --cpu-steps 0 --no-f32 --no-other-configs --no-trainer --no-kernel-timing"
for T in 256 32; do for r in 1 2 3; do
  DRN_LIB_PATH=$PWD/drn_amd/libdrn_hip_base.so python /tmp/bench_base.py $ARGS --T $T 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('T=$T before ms_per_step', d['ms_per_step'])" | tee -a $LOG
  python bench.py $ARGS --T $T 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('T=$T after  ms_per_step', d['ms_per_step'])" | tee -a $LOG
done; done
