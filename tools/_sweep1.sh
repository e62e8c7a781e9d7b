mkdir -p gpurun_out/r2b
i=0
for cfg in "" "--max-threads 1024 --lds-budget 163840 --spt 2" "--max-threads 1024 --lds-budget 163840 --spt 4" "--max-threads 512 --lds-budget 54400 --spt 2" "--max-threads 640 --lds-budget 81920 --spt 2" "--max-threads 512 --lds-budget 81920 --spt 4" "--max-threads 384 --lds-budget 40960 --spt 2"; do
  i=$((i+1))
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg > gpurun_out/r2b/sweep_$i.json 2> gpurun_out/r2b/sweep_$i.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r2b/sweep_$i.json"))
print("$cfg", "| tile_ms %.4f fin %.4f slots/tet %.3f block %d lds %d step %.4f"%(d["roofline"]["kernel_ms"],d["roofline"]["finish_kernel_ms"],d["config"]["slots_per_tet"],d["config"]["block_threads"],d["config"]["lds_bytes"],d["ms_per_step"]))
PY
done
