#!/bin/bash
# round 6, call y: where the time of a `dma2p` pair goes (100 MHz timestamps of two mid-launch workgroups, a few evaluations behind a pre-heat)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6y; mkdir -p $O; cd $R
TSSPLAT_AMD_LIB=$R/tssplat_amd/libtssplat_amd_dma2p_t.so timeout 600 python tools/run_eval.py --spheres 512 --evals 6 > $O/stamps_kuhn19.log 2>&1; tail -40 $O/stamps_kuhn19.log
