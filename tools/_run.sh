python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for v in "" nont "" nont; do
  if [ -n "$v" ]; then export TSSPLAT_AMD_LIB=$PWD/tssplat_amd/libtssplat_amd_$v.so; else unset TSSPLAT_AMD_LIB; fi
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['finish_kernel_ms'])"
done
