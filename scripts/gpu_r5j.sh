#!/bin/bash
set -u
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2 3; do GROUP_COUNTS=1,2,3,4 timeout 300 python scripts/group_bench.py f64 1500 2>&1 | grep "replica(s)" | sed 's/device ms.*//'; done
