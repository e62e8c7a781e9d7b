#!/bin/bash
# round 6, call t: does RCCL run two ranks on ONE device?  (a multi-rank RCCL execution has never happened: 1-GPU boxes only)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6t; mkdir -p $O; cd $R
timeout 300 python bench.py --gpus 2 --dist-backend nccl --all-ranks-on-device0 --spheres 16 --scene kuhn8 --steps 10 --warmup 2 --no-cpu-baseline > $O/two_ranks_nccl.json 2> $O/two_ranks_nccl.log; echo rc=$?; tail -5 $O/two_ranks_nccl.log | cut -c1-300; tail -c 300 $O/two_ranks_nccl.json
