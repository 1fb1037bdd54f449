#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_functional_gpu.py -x -q -k "variants or top_blocks" 2>&1 | tail -15 | cut -c1-250
