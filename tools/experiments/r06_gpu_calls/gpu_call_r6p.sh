#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6p2; mkdir -p $O; cd $R
for mode in "--launch graph" "--launch module --module-every 16" "--launch graph" "--launch module --module-every 16"; do
  timeout 600 python bench.py --spheres 512 --steps 1000 --warmup 50 --no-cpu-baseline --force-collective $mode 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$mode', 'ms_per_step %.5f tile %.5f finish %.5f' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['finish_kernel_ms']))"
done
timeout 900 python tools/module_breakdown.py --spheres 512 --steps 1000 2>&1 | grep "total" 
