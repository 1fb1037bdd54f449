#!/bin/bash
cd $GRAFT_REPO_ROOT
python scripts/dual_probe.py -1 1 0 2>&1 | tail -3
python scripts/dual_probe.py -1 1 1 2>&1 | tail -3
