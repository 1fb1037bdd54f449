mkdir -p gpurun_out/r2b
python -m pytest tests/test_functional_gpu.py tests/test_graph_gpu.py tests/test_model_gpu.py tests/test_optim_gpu.py tests/test_trainer_gpu.py -x -q -m gpu > gpurun_out/r2b/tests.log 2>&1; tail -5 gpurun_out/r2b/tests.log
for cfg in "DRN_SIDE_LANE=0" "DRN_SIDE_LANE=1" "DRN_SIDE_LANE=1 DRN_DEFER_WGRAD=1"; do
  echo "== $cfg"
  env $cfg python bench.py --cpu-steps 0 --no-kernel-timing --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['loss_cls'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2b/prof -o lane -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/r2b/prof/* | head
python scripts/rocprof_step.py $(ls gpurun_out/r2b/prof/*/*.db | head -1) > gpurun_out/r2b/seq.txt 2>&1; tail -3 gpurun_out/r2b/seq.txt
