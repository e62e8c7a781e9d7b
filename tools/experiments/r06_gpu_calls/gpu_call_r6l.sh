#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6l; mkdir -p $O; cd $R
timeout 1500 python tools/scaling_model.py r06 --steps 2000 --out $O/scaling_model.json > $O/scaling_model.log 2>&1; tail -34 $O/scaling_model.log | cut -c1-200
timeout 600 python bench.py --gpus 2 --dist-backend gloo --all-ranks-on-device0 --steps 20 --warmup 5 > $O/bench_2rank_dev0.json 2> $O/bench_2rank_dev0.log; tail -c 600 $O/bench_2rank_dev0.json
