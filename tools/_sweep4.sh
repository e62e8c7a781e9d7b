mkdir -p gpurun_out/r2e
for d in 0 512 1024; do
  TSSPLAT_AMD_DBG=$d TSSPLAT_AMD_WALK=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2e/dbg_$d.json 2> gpurun_out/r2e/dbg_$d.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r2e/dbg_$d.json"))
print("dbg=$d", "| tile_ms %.4f fin %.4f step %.4f E %.6f"%(d["roofline"]["kernel_ms"],d["roofline"]["finish_kernel_ms"],d["ms_per_step"],d["energy"]))
PY
done
