#!/bin/bash
# round 6, call j: pricing of a finish folded into the tile kernel for small plans (no finish launch; a release fence + atomic per workgroup)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6j; mkdir -p $O; cd $R
for cfg in "kuhn8 64" "kuhn8 256" "kuhn19 8" "aveg 16" "kuhn19 64"; do set -- $cfg
  for lib in "" "_fold_price"; do
    echo "== $1 x $2 lib${lib}" >> $O/fold.log
    TSSPLAT_AMD_LIB=$R/tssplat_amd/libtssplat_amd${lib}.so timeout 300 python bench.py --scene $1 --spheres $2 --steps 600 --warmup 60 --no-cpu-baseline --launch graph 2>/dev/null | python -c "
import json,sys
r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('ms_per_step %.5f tile %.5f finish %.5f tiles %d' % (r['ms_per_step'], r['roofline']['kernel_ms'], r['roofline']['finish_kernel_ms'], r['config']['tiles_rank0']))" >> $O/fold.log 2>&1
  done
done
cat $O/fold.log
