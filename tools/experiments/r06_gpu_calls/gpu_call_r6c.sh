#!/bin/bash
# round 6, call c: merged vertex arrays A/B, nt planes again, L2-fed pricing builds, sharded-module step, auto-rebuild thresholds
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6c; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 900 python tools/ab_variants.py base unmerged ntplanes nostream32 onewg_nostream32 --spheres 512 --passes 2 --rounds 2 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base unmerged ntplanes nostream32 --scene aveg --spheres 952 --passes 2 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
timeout 600 python tools/ab_variants.py base unmerged ntplanes --scene delaunay6000 --spheres 540 --passes 1 --rounds 2 > $O/ab_delaunay.log 2>&1; cat $O/ab_delaunay.log
for cfg in "kuhn8 512" "kuhn8 384" "kuhn19 32" "kuhn19 48" "aveg 48" "kuhn8 128"; do set -- $cfg
  for rb in 0 1; do
    echo "== $1 x $2 rebuild $rb" >> $O/mid.log
    timeout 300 python bench.py --scene $1 --spheres $2 --steps 300 --warmup 40 --no-cpu-baseline --launch graph --rebuild-dminv $rb 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.5f tile %.5f finish %.5f tiles %d slots/tet %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['finish_kernel_ms'], r['config']['tiles_rank0'], r['config']['slots_per_tet']))" >> $O/mid.log 2>&1
  done
done
cat $O/mid.log
for S in 512 64; do
  timeout 300 python bench.py --spheres $S --steps 200 --warmup 20 --no-cpu-baseline --force-collective --launch module > $O/module_$S.json 2> $O/module_$S.log
  python -c "
import json; r=json.loads(open('$O/module_$S.json').read().strip().splitlines()[-1]); print('module', $S, r['ms_per_step'], r['config']['launch'], r['config']['energy_exchange'][:80])"
done
timeout 300 python bench.py --spheres 64 --steps 200 --warmup 20 --no-cpu-baseline --force-collective > $O/private_64.json 2> $O/private_64.log
python -c "
import json; r=json.loads(open('$O/private_64.json').read().strip().splitlines()[-1]); print('private loop 64', r['ms_per_step'], r.get('eager_autograd_ms_per_step'), r.get('graph_autograd_ms_per_step'))"
