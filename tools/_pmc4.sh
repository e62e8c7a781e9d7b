cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r2n
for rb in 0 1; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/r2n/pmc_rb$rb -- python $R/tools/run_eval.py --evals 30 --rebuild-dminv $rb > $R/gpurun_out/r2n/pmc_rb$rb.log 2>&1
done
cd $R
python - <<'PY'
import csv,glob,collections
for rb in (0,1):
    agg=collections.defaultdict(list); dur=[]
    for f in glob.glob(f"gpurun_out/r2n/pmc_rb{rb}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "tile_energy_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"gpurun_out/r2n/pmc_rb{rb}/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            if "tile_energy_kernel" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
    dur=dur[5:]
    d=sum(dur)/len(dur)
    g=agg["GRBM_GUI_ACTIVE"][5:]; g=sum(g)/len(g)
    print("rebuild",rb,"kernel us %.1f"%d,"GRBM_GUI_ACTIVE %.0f"%g,"-> clock GHz %.3f"%(g/d/1e3), {k:round(sum(v)/len(v)/1e6,2) for k,v in agg.items() if k!='GRBM_GUI_ACTIVE'})
PY
