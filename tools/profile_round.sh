#!/bin/bash
# Round profile run (GPU box): rocprofv3 kernel stats + PMC passes for the BASELINE scenes, parity report, bench lines.
#   gpurun -- 'bash tools/profile_round.sh r02'
# Raw output lands in gpurun_out/<round>p/; tools/summarize_prof.py condenses it into profiles/.
set -u
ROUND=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${ROUND}p
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# headline: the bench command itself under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_kuhn19x512 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/stats_kuhn19x512.log 2>&1
for cfg in "kuhn19 512" "kuhn19 256" "kuhn8 256" "kuhn8 64"; do set -- $cfg
  W=$1x$2
  [ "$W" != "kuhn19x512" ] && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --evals 40 > $OUT/stats_$W.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --evals 6 > $OUT/pmc_fetch_$W.log 2>&1
  rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc_write_$W -- python $R/tools/run_eval.py --scene $1 --spheres $2 --evals 6 > $OUT/pmc_write_$W.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/pmc_sq1_kuhn19x512 -- python $R/tools/run_eval.py --evals 6 > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq2_kuhn19x512 -- python $R/tools/run_eval.py --evals 6 > $OUT/pmc_sq2.log 2>&1
cd $R
rm -f $OUT/parity.txt
TSSPLAT_AMD_PARITY_REPORT=$OUT/parity.txt python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
python bench.py > $OUT/bench_kuhn19x512.json 2> $OUT/bench_kuhn19x512.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_kuhn19x512_driver.json 2>> $OUT/bench_kuhn19x512.log
for cfg in "kuhn19 256" "kuhn8 256" "kuhn8 64"; do set -- $cfg
  python bench.py --scene $1 --spheres $2 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_$1x$2.json 2> $OUT/bench_$1x$2.log
done
python bench.py --gpus 2 --dist-backend gloo --all-ranks-on-device0 --steps 20 --warmup 5 > $OUT/bench_2rank_dev0.json 2> $OUT/bench_2rank_dev0.log
# renderer slice: kernel stats of the raster bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_raster -- python $R/tools/bench_raster.py --reps 5 > $OUT/stats_raster.log 2>&1)
python tools/bench_raster.py > $OUT/bench_raster.json 2> $OUT/bench_raster.log
python tools/bench_raster.py --spheres 64 > $OUT/bench_raster64.json 2>> $OUT/bench_raster.log
# the reference's inner loop (geometry + renderer + optimiser) on the headline scene and on a single object
python tools/bench_pipeline.py > $OUT/pipeline_512.json 2> $OUT/pipeline.log
python tools/bench_pipeline.py --spheres 1 > $OUT/pipeline_1.json 2>> $OUT/pipeline.log
python tools/bench_pipeline.py --spheres 1 --views 120 --iters 10 > $OUT/pipeline_1x120.json 2>> $OUT/pipeline.log
python tools/train_synthetic.py > $OUT/train_synthetic.json 2>> $OUT/pipeline.log
for cfg in "kuhn8 64" "kuhn8 256"; do set -- $cfg
  python tools/bench_train_loop.py --scene $1 --spheres $2 > $OUT/train_loop_$1x$2.json 2>> $OUT/pipeline.log
done
# per-phase shader-clock stamps and stage ablations of the tile kernel (ablation build), two and one workgroups per CU
python tools/ablate.py --spheres 512 --reps 10 > $OUT/ablate_512.log 2>&1
python tools/ablate.py --spheres 512 --reps 5 --masks 0 --lds-request 100000 > $OUT/ablate_512_1wg.log 2>&1
python tools/scaling_model.py $ROUND --out $OUT/scaling_model.json > $OUT/scaling_model.log 2>&1
# round 4: production kernel + stamps, host cost of the operator surface (C++ autograd nodes), explicit operator, config 5 on the reference's object
python tools/ablate.py --stamps-only --spheres 512 --reps 50 > $OUT/stamps_512.log 2>&1
python tools/host_overhead.py > $OUT/host_overhead.txt 2>&1
python tools/bench_operator.py > $OUT/bench_operator.txt 2>&1
python tools/train_object.py > $OUT/train_mario.json 2> $OUT/train_mario.log
# drop the bulky per-dispatch traces, keep stats + counters
find $OUT -name "*kernel_trace.csv" -size +2M -delete
ls $OUT | head -50
