#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6o; mkdir -p $O; cd $R
timeout 900 python tools/module_breakdown.py --spheres 512 --steps 600 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^initializing" > $O/module_breakdown_512.txt; cat $O/module_breakdown_512.txt
