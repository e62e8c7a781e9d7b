#!/bin/bash
# round 6, call g: does a staggered second workgroup help SHORT launches (the 8-way share of the headline scene: 4.75 rounds of workgroups)?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6g; mkdir -p $O; cd $R
timeout 600 python tools/ab_variants.py base stagger stagger2 staggerL staggerL2 staggerJ --spheres 64 --evals 600 --warm 600 --passes 2 --rounds 2 > $O/ab_kuhn19x64.log 2>&1; cat $O/ab_kuhn19x64.log
timeout 600 python tools/ab_variants.py base stagger2 staggerL2 --spheres 128 --evals 400 --warm 400 --passes 1 --rounds 2 > $O/ab_kuhn19x128.log 2>&1; cat $O/ab_kuhn19x128.log
timeout 300 python tools/ab_variants.py base stagger2 staggerL2 --scene kuhn8 --spheres 256 --evals 600 --warm 600 --passes 1 --rounds 2 > $O/ab_kuhn8x256.log 2>&1; cat $O/ab_kuhn8x256.log
