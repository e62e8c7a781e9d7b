mkdir -p gpurun_out/r2c
i=0
for cfg in "--spt 1 --max-threads 1024" "--spt 1 --max-threads 1024 --lds-budget 54400" "--spt 1 --max-threads 768 --lds-budget 54400" "--spt 1 --max-threads 512 --lds-budget 40960"; do
  i=$((i+1))
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline $cfg > gpurun_out/r2c/sweep_$i.json 2> gpurun_out/r2c/sweep_$i.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r2c/sweep_$i.json"))
print("$cfg", "| tile_ms %.4f fin %.4f slots/tet %.3f block %d lds %d step %.4f E %.6f"%(d["roofline"]["kernel_ms"],d["roofline"]["finish_kernel_ms"],d["config"]["slots_per_tet"],d["config"]["block_threads"],d["config"]["lds_bytes"],d["ms_per_step"],d["energy"]))
PY
done
python -m pytest tests/test_gpu_parity.py -x -q -k "multi_tile or config2" 2>&1 | tail -3
