python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config']['loss_cls'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2g/prof -o q -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-kernel-timing --no-f32 --steps 30 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python scripts/rocprof_step.py gpurun_out/r2g/prof/q_results.db > gpurun_out/r2g/seq.txt 2>&1; tail -1 gpurun_out/r2g/seq.txt
rm -rf gpurun_out/r2g/prof
