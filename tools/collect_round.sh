#!/bin/bash
# Condense the raw output of tools/profile_round.sh (gpurun_out/<round>p/, scratch) into the tracked files under profiles/.
#   bash tools/collect_round.sh r05
set -u
ROUND=${1:-r06}
O=gpurun_out/${ROUND}p
for w in kuhn19x512 avegx952 delaunay6000x540 kuhn8x256 kuhn8x64; do
  extra=""
  [ -d $O/pmc_sq1_$w ] && extra="$O/pmc_sq1_$w $O/pmc_sq2_$w"
  python tools/summarize_prof.py --round ${ROUND}_$w --label "$w" --workload $w --stats $O/stats_$w --pmc $O/pmc_fetch_$w $O/pmc_write_$w $extra 2>&1 | tail -2
done
for f in bench_kuhn19x512 bench_kuhn19x512_driver bench_avegx952 bench_delaunay6000x540 bench_kuhn19x256 bench_kuhn8x256 bench_kuhn8x64; do tail -1 $O/$f.json > profiles/${ROUND}_$f.json; done
tail -1 $O/bench_2rank_dev0.json > profiles/${ROUND}_bench_2rank_gloo_one_device.json
cp $O/parity.txt profiles/${ROUND}_parity.txt
cp $O/scaling_model.json profiles/${ROUND}_scaling_model.json
cp $O/bench_operator.txt profiles/${ROUND}_bench_operator.txt
(echo "## $(date +%F) tools/profile_round.sh $ROUND"; grep -v "libdrm\|initializing" $O/host_overhead.txt) >> profiles/${ROUND}_host_overhead.txt
python - "$O" "$ROUND" <<'PY'
import json, sys
o, r = sys.argv[1], sys.argv[2]
s = open(f"{o}/train_mario.json").read()
d = json.loads(s[s.index("{"):])
json.dump(d, open(f"profiles/{r}_train_mario.json", "w"), indent=1)
print({k: d[k] for k in ("ms_per_iteration", "silhouette_iou", "inverted_tets", "stages_ms")})
for w in ("kuhn19x512", "avegx952", "delaunay6000x540", "kuhn8x256", "kuhn8x64"):
    b = json.loads(open(f"profiles/{r}_bench_{w}.json").read().strip().splitlines()[-1])
    print(w, b["value"], b["ms_per_step"], {k: b["roofline"][k] for k in ("kernel_ms", "finish_kernel_ms", "frac", "frac_step", "traffic")})
PY
tail -2 $O/pytest_gpu.log
