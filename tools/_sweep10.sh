python -m pytest tests/test_gpu_parity.py -x -q -k "multi_tile" 2>&1 | tail -3
for r in 0 1; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch eager --rebuild-dminv $r 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rebuild=$r ms/step %.4f tile %.4f fin %.4f slots/tet %.3f E %.6f'%(d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['finish_kernel_ms'],d['config']['slots_per_tet'],d['energy']))"
done
