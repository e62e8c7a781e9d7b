#!/bin/bash
# round 6, call ac: the per-vertex sums software-pipelined (`sums_pipe`): A/B on three scenes + the small batches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ac; mkdir -p $O; cd $R
timeout 900 python tools/ab_variants.py base sums_pipe rows8 --spheres 512 --passes 2 --rounds 2 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base sums_pipe --scene aveg --spheres 952 --passes 2 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
timeout 600 python tools/ab_variants.py base sums_pipe --scene delaunay6000 --spheres 540 --passes 2 --rounds 2 > $O/ab_delaunay.log 2>&1; cat $O/ab_delaunay.log
timeout 600 python tools/ab_variants.py base sums_pipe --scene kuhn8 --spheres 64 --passes 2 --rounds 2 > $O/ab_kuhn8x64.log 2>&1; cat $O/ab_kuhn8x64.log
