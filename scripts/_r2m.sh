#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_lstm_gpu.py tests/test_qenc_gpu.py -x -q 2>&1 | tail -2 | cut -c1-300
bash scripts/_r2l.sh
