#!/bin/bash
# round 6, call ab: does halving the longest walk of the per-vertex sums (rows8) shorten the workgroup?  stamps on rows8 against stamps, a.veg and Delaunay; A/B of rows8 itself
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ab; mkdir -p $O; cd $R
for v in stamps stamps_rows8; do
  for sc in "aveg 952" "delaunay6000 540"; do set -- $sc
    TSSPLAT_AMD_LIB=$R/tssplat_amd/libtssplat_amd_$v.so timeout 600 python tools/run_eval.py --scene $1 --spheres $2 --evals 4 > $O/${v}_$1.log 2>&1; grep -c "^blk" $O/${v}_$1.log
  done
done
timeout 600 python tools/ab_variants.py base rows8 --scene aveg --spheres 952 --passes 1 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
