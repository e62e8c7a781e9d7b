#!/bin/bash
# round 6, call e: product = paired + nt everything + adaptive row trips + C++ exchange; A/B against each piece; small-batch retile cases; module
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6e; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python tools/ab_variants.py base rows4 rows8 tids tfinish tplanes --spheres 512 --passes 2 --rounds 2 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base rows4 rows8 rows8_24 tids tfinish --scene aveg --spheres 952 --passes 2 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
timeout 600 python tools/ab_variants.py base rows4 rows8 rows8_24 tids tfinish --scene delaunay6000 --spheres 540 --passes 2 --rounds 2 > $O/ab_delaunay.log 2>&1; cat $O/ab_delaunay.log
for cfg in "aveg 8" "aveg 16" "kuhn19 8" "kuhn19 12" "delaunay3000 20" "kuhn8 20" "kuhn8 64"; do set -- $cfg
  for extra in "" "--lds-budget 81920" "--target-owned 768"; do
    echo "== $1 x $2 $extra" >> $O/small.log
    timeout 300 python bench.py --scene $1 --spheres $2 --steps 400 --warmup 50 --no-cpu-baseline --launch graph $extra 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('ms_per_step %.5f tile %.5f finish %.5f tiles %d slots/tet %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['finish_kernel_ms'], r['config']['tiles_rank0'], r['config']['slots_per_tet']))" >> $O/small.log 2>&1
  done
done
cat $O/small.log
timeout 600 python tools/host_overhead.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^initializing" > $O/host_overhead.txt; tail -12 $O/host_overhead.txt
timeout 1200 python tools/scaling_model.py r06 --steps 2000 --out $O/scaling_model.json > $O/scaling_model.log 2>&1; tail -32 $O/scaling_model.log | cut -c1-400
