#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6m; mkdir -p $O; cd $R
timeout 900 python tools/module_breakdown.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^initializing" > $O/module_breakdown.txt; cat $O/module_breakdown.txt
