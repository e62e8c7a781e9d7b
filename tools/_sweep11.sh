for cfg in "" "--max-threads 512 --lds-budget 54400" "--max-threads 640 --lds-budget 68000" "--max-threads 704 --lds-budget 74000"; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch eager $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg | ms/step %.4f tile %.4f fin %.4f slots/tet %.3f'%(d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['finish_kernel_ms'],d['config']['slots_per_tet']))"
done
