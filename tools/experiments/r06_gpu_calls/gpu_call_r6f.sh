#!/bin/bash
# round 6, call f: final product kernels; module (every 1 / 16) and the scaling model; auto tiling on small batches
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6f; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python tools/ab_variants.py base unpaired tplanes tids --spheres 512 --passes 2 --rounds 2 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base unpaired tplanes tids --scene aveg --spheres 952 --passes 2 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
for cfg in "aveg 8" "aveg 16" "aveg 48" "kuhn19 8" "kuhn19 12" "delaunay3000 20" "kuhn8 64" "kuhn8 256"; do set -- $cfg
  echo "== $1 x $2 (automatic tiling)" >> $O/small.log
  timeout 300 python bench.py --scene $1 --spheres $2 --steps 400 --warmup 50 --no-cpu-baseline --launch graph 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('ms_per_step %.5f tile %.5f finish %.5f tiles %d slots/tet %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['finish_kernel_ms'], r['config']['tiles_rank0'], r['config']['slots_per_tet']))" >> $O/small.log 2>&1
done
cat $O/small.log
timeout 600 python tools/host_overhead.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^initializing" > $O/host_overhead.txt; grep "Sharded\|piece" $O/host_overhead.txt
timeout 1500 python tools/scaling_model.py r06 --steps 2000 --out $O/scaling_model.json > $O/scaling_model.log 2>&1; tail -36 $O/scaling_model.log | cut -c1-200
