#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6k; mkdir -p $O; cd $R
timeout 600 python tools/ab_variants.py base ntstore_excl --spheres 512 --passes 2 --rounds 2 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base ntstore_excl --scene aveg --spheres 952 --passes 1 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
