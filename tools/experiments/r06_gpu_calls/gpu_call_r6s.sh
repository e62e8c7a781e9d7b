#!/bin/bash
# round 6, call s: a one-off soak of the random tiling-option test (400 trials incl. lane layouts 3 and 4) on the final kernels
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6s; mkdir -p $O; cd $R
TSSPLAT_AMD_SOAK=400 timeout 2400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_tiling_options" -s > $O/soak.log 2>&1; tail -3 $O/soak.log; grep -c "^\[random#" $O/soak.log
