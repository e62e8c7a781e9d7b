python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for pf in 0 512 384 768; do
  export TSSPLAT_AMD_PREFETCH_AHEAD=$pf
  python bench.py --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$pf', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['finish_kernel_ms'])"
done
