mkdir -p gpurun_out/r2d
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for w in 1 0; do
  TSSPLAT_AMD_WALK=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2d/walk_$w.json 2> gpurun_out/r2d/walk_$w.log
  python - <<PY
import json
d=json.load(open("gpurun_out/r2d/walk_$w.json"))
print("walk=$w", "| tile_ms %.4f fin %.4f slots/tet %.3f step %.4f E %.6f"%(d["roofline"]["kernel_ms"],d["roofline"]["finish_kernel_ms"],d["config"]["slots_per_tet"],d["ms_per_step"],d["energy"]))
PY
done
