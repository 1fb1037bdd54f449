#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_parity_grad_gpu.py -q --durations=4 -x 2>&1 | tail -9 | cut -c1-250
