#!/bin/bash
# round 6, call aa: inside the per-vertex sums of waves 0 / 1 / 3 (`stamps2`): row loop, wait for the destination ids, stores -- lattice against a.veg
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6aa; mkdir -p $O; cd $R
export TSSPLAT_AMD_LIB=$R/tssplat_amd/libtssplat_amd_stamps2.so
timeout 600 python tools/run_eval.py --spheres 512 --evals 3 > $O/stamps2_kuhn19.log 2>&1; grep "^blk" $O/stamps2_kuhn19.log | sort | tail -18
timeout 600 python tools/run_eval.py --scene aveg --spheres 952 --evals 3 > $O/stamps2_aveg.log 2>&1; grep "^blk" $O/stamps2_aveg.log | sort | tail -18
timeout 600 python tools/run_eval.py --scene delaunay6000 --spheres 540 --evals 3 > $O/stamps2_delaunay.log 2>&1; grep "^blk" $O/stamps2_delaunay.log | sort | tail -18
