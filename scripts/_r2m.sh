#!/bin/bash
cd $GRAFT_REPO_ROOT
export DRN_DIST_BACKEND=gloo DRN_FORCE_DEVICE=0
for i in 1 2; do
S=$SECONDS
timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2951$i bench.py --gpus 2 --steps 6 --warmup 2 --cpu-steps 0 --verbose > gpurun_out/n2_$i.log 2>&1
echo "run $i rc $? wall $((SECONDS-S)) s"; grep -E "bench rank|Error|Signal" gpurun_out/n2_$i.log | cut -c1-160 | tail -6
done
