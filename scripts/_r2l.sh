#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 1500 python -m pytest tests -q -m gpu --tb=line -rf 2>&1 | grep -E "passed|failed|^FAILED|Error" | cut -c1-300; done
