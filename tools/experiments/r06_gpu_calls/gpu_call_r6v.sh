#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6v; mkdir -p $O; cd $R
timeout 120 tools/_bin/ubench_dma 2>&1 | head -12 > $O/dma_where.txt; cat $O/dma_where.txt
