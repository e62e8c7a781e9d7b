#!/bin/bash
# round 6, call r: three / four workgroups per CU (smaller tiles) re-measured on this round's kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6r; mkdir -p $O; cd $R
timeout 600 python tools/ab_variants.py base --spheres 512 --passes 1 --rounds 2 > $O/a.log 2>&1; cat $O/a.log
timeout 600 python tools/ab_variants.py base --spheres 512 --passes 1 --rounds 2 --opts max_threads=512 lds_budget_bytes=54400 > $O/b.log 2>&1; cat $O/b.log
timeout 600 python tools/ab_variants.py base --spheres 512 --passes 1 --rounds 2 --opts max_threads=384 lds_budget_bytes=40960 > $O/c.log 2>&1; cat $O/c.log
timeout 600 python tools/ab_variants.py base --spheres 512 --passes 1 --rounds 2 --opts max_threads=640 lds_budget_bytes=65536 > $O/d.log 2>&1; cat $O/d.log
