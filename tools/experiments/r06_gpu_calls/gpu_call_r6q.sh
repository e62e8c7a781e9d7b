#!/bin/bash
# round 6, call q: capacity rows of BASELINE.md with this round's kernel (2 048 / 8 192 spheres), the cone stress, 256 x kuhn19
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6q; mkdir -p $O; cd $R
for cfg in "kuhn19 2048 20" "kuhn19 8192 8" "cone 2000 200"; do set -- $cfg
  timeout 1500 python bench.py --scene $1 --spheres $2 --steps $3 --warmup 3 --no-cpu-baseline > $O/bench_$1x$2.json 2> $O/bench_$1x$2.log
  python -c "
import json; r=json.loads([l for l in open('$O/bench_$1x$2.json').read().splitlines() if l.startswith('{')][-1]); print('$1 x $2', 'ms %.4f G %.2f' % (r['ms_per_step'], r['value']/1e9), {k:round(r['roofline'][k],4) for k in ('frac','frac_step','kernel_ms','finish_kernel_ms')}, r['config']['tets_rank0'], 'plan s', round(r['plan_build_s'],1))"
done
