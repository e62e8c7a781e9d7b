#!/bin/bash
# round 6, call ad: the driver's end-of-round sequence on the final tree, plus bench.py under torch.distributed.run (one rank, RCCL group forced)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6ad; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log; python -c "
import json; r=json.loads([l for l in open('$O/bench_default.json').read().splitlines() if l.startswith('{')][-1]); print(r['ms_per_step'], r['value'], {k:r['roofline'][k] for k in ('frac','frac_step','frac_kernels','traffic','kernel_ms','finish_kernel_ms')}, r['cpu_baseline']['value'])"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --force-collective > $O/bench_torchrun.json 2> $O/bench_torchrun.log; python -c "
import json; r=json.loads([l for l in open('$O/bench_torchrun.json').read().splitlines() if l.startswith('{')][-1]); print(r['ms_per_step'], r['value'], r['n_gpus'], r['config'])"
