#!/bin/bash
# round 6, call a: GPU suite with the new tests, pair16 (16-byte paired plane loads) parity + A/B, LDS-DMA ubench, baseline bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6a; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
TSSPLAT_AMD_LIB=$R/tssplat_amd/libtssplat_amd_pair16.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "multi_tile_hires or explicit_operator or lane_layouts or config2 or aveg or unfactored" > $O/pytest_pair16.log 2>&1; tail -3 $O/pytest_pair16.log
timeout 600 python tools/ab_variants.py base pair16 --spheres 512 > $O/ab_kuhn19.log 2>&1; cat $O/ab_kuhn19.log
timeout 600 python tools/ab_variants.py base pair16 --scene aveg --spheres 952 > $O/ab_aveg.log 2>&1; cat $O/ab_aveg.log
timeout 300 python tools/ab_variants.py base pair16 --scene kuhn8 --spheres 256 --evals 400 > $O/ab_kuhn8.log 2>&1; cat $O/ab_kuhn8.log
timeout 120 tools/_bin/ubench_dma > $O/ubench_dma.txt 2>&1; tail -40 $O/ubench_dma.txt
timeout 120 tools/_bin/ubench_ingest > $O/ubench_ingest.txt 2>&1; tail -14 $O/ubench_ingest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.log; cat $O/bench.json
