#!/bin/bash
# Condense the raw output of tools/profile_round.sh (gpurun_out/<round>p/, scratch) into the tracked files under profiles/.
#   bash tools/collect_round.sh r04
set -u
ROUND=${1:-r04}
O=gpurun_out/${ROUND}p
for w in kuhn19x512 kuhn19x256 kuhn8x256 kuhn8x64; do
  extra=""
  [ $w = kuhn19x512 ] && extra="$O/pmc_sq1_kuhn19x512 $O/pmc_sq2_kuhn19x512"
  python tools/summarize_prof.py --round ${ROUND}_$w --label "$w" --workload $w --stats $O/stats_$w --pmc $O/pmc_fetch_$w $O/pmc_write_$w $extra 2>&1 | tail -2
done
for f in bench_kuhn19x512 bench_kuhn19x512_driver bench_kuhn19x256 bench_kuhn8x256 bench_kuhn8x64; do cp $O/$f.json profiles/${ROUND}_$f.json; done
cp $O/bench_2rank_dev0.json profiles/${ROUND}_bench_2rank_gloo_one_device.json
cp $O/parity.txt profiles/${ROUND}_parity.txt
cp $O/scaling_model.json profiles/${ROUND}_scaling_model.json
for f in bench_raster pipeline_512 pipeline_1x120 train_loop_kuhn8x64 train_loop_kuhn8x256; do tail -1 $O/$f.json > profiles/${ROUND}_$f.json; done
cp $O/bench_operator.txt profiles/${ROUND}_bench_operator.txt
f=$(find $O/stats_raster -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" profiles/${ROUND}_raster_kernel_stats.csv
(echo "## $(date +%F) tools/profile_round.sh $ROUND"; grep -v "libdrm\|initializing" $O/host_overhead.txt) >> profiles/${ROUND}_host_overhead.txt
grep -v "^{" $O/stamps_512.log | grep -v "libdrm\|initializing\|^[0-9]*/[0-9]\|resident workgroups\|launch span" > profiles/${ROUND}_stamps_512.txt
python - "$O" "$ROUND" <<'PY'
import json, sys
o, r = sys.argv[1], sys.argv[2]
s = open(f"{o}/train_mario.json").read()
d = json.loads(s[s.index("{"):])
json.dump(d, open(f"profiles/{r}_train_mario.json", "w"), indent=1)
print({k: d[k] for k in ("ms_per_iteration", "silhouette_iou", "inverted_tets", "stages_ms")})
b = json.loads(open(f"profiles/{r}_bench_kuhn19x512.json").read().strip().splitlines()[-1])
print(b["value"], b["ms_per_step"], b["roofline"], b["cpu_baseline"]["value"])
PY
tail -2 $O/pytest_gpu.log
