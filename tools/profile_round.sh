#!/bin/bash
# Round profile run (GPU box): rocprofv3 kernel stats + PMC passes for the BASELINE scenes and the unstructured ones, parity
# report, bench lines.   gpurun -- 'bash tools/profile_round.sh r05'
# Raw output lands in gpurun_out/<round>p/; tools/collect_round.sh condenses it into profiles/.
set -u
ROUND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${ROUND}p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# headline: the bench command itself under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_kuhn19x512 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats_kuhn19x512.log 2>&1
for cfg in "kuhn19 512" "aveg 952" "delaunay6000 540" "kuhn8 256" "kuhn8 64"; do set -- $cfg
  W=$1x$2
  # (>= 500 launches behind a pre-heat, like the headline run: 40 cold launches showed the clock ramp)
  [ "$W" != "kuhn19x512" ] && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --preheat-ms 150 --evals 600 > $OUT/stats_$W.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --evals 6 > $OUT/pmc_fetch_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --evals 6 > $OUT/pmc_write_$W.log 2>&1
done
for cfg in "kuhn19 512" "aveg 952" "delaunay6000 540"; do set -- $cfg
  W=$1x$2
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq1_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --evals 6 > $OUT/pmc_sq1_$W.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --evals 6 > $OUT/pmc_sq2_$W.log 2>&1
done
cd $R
rm -f $OUT/parity.txt
TSSPLAT_AMD_PARITY_REPORT=$OUT/parity.txt python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_kuhn19x512.json 2> $OUT/bench_kuhn19x512.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_kuhn19x512_driver.json 2>> $OUT/bench_kuhn19x512.log
python bench.py --scene aveg --spheres 952 --no-cpu-baseline > $OUT/bench_avegx952.json 2> $OUT/bench_avegx952.log
python bench.py --scene delaunay6000 --spheres 540 --no-cpu-baseline > $OUT/bench_delaunay6000x540.json 2> $OUT/bench_delaunay6000x540.log
for cfg in "kuhn19 256" "kuhn8 256" "kuhn8 64"; do set -- $cfg
  python bench.py --scene $1 --spheres $2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_$1x$2.json 2> $OUT/bench_$1x$2.log
done
python bench.py --gpus 2 --dist-backend gloo --all-ranks-on-device0 --steps 20 --warmup 5 > $OUT/bench_2rank_dev0.json 2> $OUT/bench_2rank_dev0.log
python tools/scaling_model.py $ROUND --out $OUT/scaling_model.json > $OUT/scaling_model.log 2>&1
python tools/host_overhead.py > $OUT/host_overhead.txt 2>&1
python tools/bench_operator.py > $OUT/bench_operator.txt 2>&1
python tools/train_object.py > $OUT/train_mario.json 2> $OUT/train_mario.log
# drop the bulky per-dispatch traces, keep stats + counters
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls $OUT | head -60
