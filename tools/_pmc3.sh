cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2m
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VSKIPPED SQ_INSTS"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r2m/pmc_$i -- python $R/tools/run_eval.py --evals 4 > $R/gpurun_out/r2m/pmc_$i.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/r2m/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "tile_energy_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
slots=27055430/64
for k,v in sorted(agg.items()): print(k, round(sum(v)/len(v)/1e6,2), 'per slot-row %.1f'%(sum(v)/len(v)/slots))
PY
