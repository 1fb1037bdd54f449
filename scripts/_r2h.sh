#!/bin/bash
# full GPU check: parity tests, then a profile round
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r2h_tests.txt
cat gpurun_out/r2h_tests.txt
bash scripts/prof_round.sh r02_e
