#!/bin/bash
# round 6, call z: the product kernel's phases as wave 0 / wave 11 of six mid-launch workgroups see them (`stamps`), three scenes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6z; mkdir -p $O; cd $R
export TSSPLAT_AMD_LIB=$R/tssplat_amd/libtssplat_amd_stamps.so
timeout 600 python tools/run_eval.py --spheres 512 --evals 4 > $O/stamps_kuhn19.log 2>&1; grep -c "^blk" $O/stamps_kuhn19.log
timeout 600 python tools/run_eval.py --scene aveg --spheres 952 --evals 4 > $O/stamps_aveg.log 2>&1; grep -c "^blk" $O/stamps_aveg.log
timeout 600 python tools/run_eval.py --scene delaunay6000 --spheres 540 --evals 4 > $O/stamps_delaunay.log 2>&1; grep -c "^blk" $O/stamps_delaunay.log
timeout 600 python tools/run_eval.py --scene kuhn8 --spheres 64 --evals 4 > $O/stamps_kuhn8x64.log 2>&1; tail -3 $O/stamps_kuhn8x64.log
