#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6n; mkdir -p $O; cd $R
timeout 900 python tools/module_breakdown.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^initializing" > $O/module_breakdown.txt; cat $O/module_breakdown.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sharded" > $O/pytest_sharded.log 2>&1; tail -2 $O/pytest_sharded.log
timeout 1500 python tools/scaling_model.py r06 --steps 2000 --out $O/scaling_model.json > $O/scaling_model.log 2>&1; tail -12 $O/scaling_model.log | cut -c1-200
