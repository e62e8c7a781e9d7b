mkdir -p gpurun_out/r2i
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2i/bench_auto.json 2> gpurun_out/r2i/bench_auto.log
python bench.py --gpus 2 --dist-backend gloo --all-ranks-on-device0 --spheres 64 --steps 20 --warmup 5 > gpurun_out/r2i/bench_2rank.json 2> gpurun_out/r2i/bench_2rank.log
python bench.py --scene kuhn8 --spheres 64 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2i/bench_k8_64.json 2> gpurun_out/r2i/bench_k8_64.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2i/*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f,"BAD",e, open(f).read()[:300]); continue
    print(f.split('/')[-1], "n_gpus",d["n_gpus"],"ms/step %.4f"%d["ms_per_step"],"Gtet/s %.2f"%(d["value"]/1e9),"tile %.4f fin %.4f"%(d["roofline"]["kernel_ms"],d["roofline"]["finish_kernel_ms"]),"other",d.get("eager_autograd_ms_per_step",d.get("graph_replay_ms_per_step")),d["config"]["launch"][:12])
PY
tail -3 gpurun_out/r2i/bench_auto.log
