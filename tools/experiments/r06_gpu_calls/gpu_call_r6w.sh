#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6w; mkdir -p $O; cd $R
timeout 600 python tools/ab_variants.py base dma2 dma2_chk --spheres 8 --evals 20 --warm 20 --passes 1 --rounds 1 > $O/ab.log 2>&1; cat $O/ab.log
