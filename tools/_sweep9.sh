python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch eager 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f tile %.4f fin %.4f'%(d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['finish_kernel_ms']))"
