#!/bin/bash
# round 6, call x: `dma2p` = dma2 + the head of the DMA-fed tile (neighbour planes, row table, descriptor) prefetched by the consumers: parity and time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6x; mkdir -p $O; cd $R
TSSPLAT_AMD_LIB=$R/tssplat_amd/libtssplat_amd_dma2p.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "config2 or real_mesh_aveg or unstructured_delaunay or cone_hub or config4_full_size or aveg_full_size" > $O/pytest_dma2p.log 2>&1; tail -4 $O/pytest_dma2p.log
timeout 900 python tools/ab_variants.py base onewg dma2 dma2p --spheres 512 --passes 2 --rounds 2 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base onewg dma2 dma2p --scene aveg --spheres 952 --passes 1 --rounds 2 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
