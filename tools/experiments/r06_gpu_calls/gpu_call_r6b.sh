#!/bin/bash
# round 6, call b: GPU suite on the paired-plane product; pricing builds for the stream-hiding designs; nt plane loads;
# small-batch tilings; the sharded module's step and the scaling model
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6b; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
timeout 900 python tools/ab_variants.py base unpaired ntplanes nostream onewg onewg_nostream --spheres 512 --passes 1 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base ntplanes nostream --scene aveg --spheres 952 --passes 1 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
for S in 64 256; do
  for extra in "" "--target-owned 384" "--target-owned 512" "--rebuild-dminv 1" "--target-owned 384 --rebuild-dminv 1" "--target-owned 512 --rebuild-dminv 1"; do
    echo "== kuhn8 x $S $extra" >> $O/small.log
    timeout 200 python bench.py --scene kuhn8 --spheres $S --steps 400 --warmup 50 --no-cpu-baseline --launch graph $extra 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.5f tile %.5f finish %.5f tiles %d slots/tet %.3f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['finish_kernel_ms'], r['config']['tiles_rank0'], r['config']['slots_per_tet']))" >> $O/small.log 2>&1
  done
done
cat $O/small.log
timeout 900 python tools/scaling_model.py r06 --out $O/scaling_model.json > $O/scaling_model.log 2>&1; tail -30 $O/scaling_model.log
