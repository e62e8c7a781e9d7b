#!/bin/bash
# round 6, call d: nt planes as product (vs tplanes), ntids, rows8, lone-workgroup pricing; module host breakdown; aveg mid-size
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6d; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "multi_tile or lane_layouts or config2 or config3 or aveg or delaunay or sharded or operator" > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 900 python tools/ab_variants.py base tplanes ntids rows8 onewg onewg_exit_p1 exit_p1 --spheres 512 --passes 2 --rounds 2 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base tplanes ntids rows8 --scene aveg --spheres 952 --passes 2 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
timeout 600 python tools/ab_variants.py base ntids rows8 --scene delaunay6000 --spheres 540 --passes 1 --rounds 2 > $O/ab_delaunay.log 2>&1; cat $O/ab_delaunay.log
for extra in "" "--rebuild-dminv 0" "--rebuild-dminv 1" "--rebuild-dminv 0 --lds-budget 81920" "--rebuild-dminv 1 --lds-budget 81920"; do
  echo "== aveg x 48 $extra" >> $O/mid.log
  timeout 300 python bench.py --scene aveg --spheres 48 --steps 300 --warmup 40 --no-cpu-baseline --launch graph $extra 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('ms_per_step %.5f tile %.5f finish %.5f tiles %d slots/tet %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['finish_kernel_ms'], r['config']['tiles_rank0'], r['config']['slots_per_tet']))" >> $O/mid.log 2>&1
done
cat $O/mid.log
timeout 600 python tools/host_overhead.py 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|^initializing" > $O/host_overhead.txt; tail -22 $O/host_overhead.txt
for S in 512 64; do
  timeout 300 python bench.py --spheres $S --steps 3000 --warmup 300 --no-cpu-baseline --force-collective --launch module > $O/module_$S.json 2> $O/module_$S.log
  python -c "
import json; r=json.loads([l for l in open('$O/module_$S.json').read().splitlines() if l.startswith('{')][-1]); print('module', $S, r['ms_per_step'])"
done
