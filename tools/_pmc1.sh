cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in 4 0; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/r2k/pmc_d$d -- python $R/tools/run_eval.py --evals 4 --debug-shuffle $d > $R/gpurun_out/r2k/pmc_d$d.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob,collections
for d in (4,0):
    agg=collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r2k/pmc_d{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "tile_energy_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("debug_shuffle",d,{k:round(sum(v)/len(v)/1e6,2) for k,v in agg.items()})
PY
