TSSPLAT_AMD_WALK=16 python tools/_walk_test.py 2>&1 | grep -v amdgpu | tail -7
for w in 0 512; do
TSSPLAT_AMD_WALK=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --launch eager --rebuild-dminv 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('walk=$w rebuild=1 ms/step %.4f tile %.4f fin %.4f slots/tet %.3f E %.6f'%(d['ms_per_step'],d['roofline']['kernel_ms'],d['roofline']['finish_kernel_ms'],d['config']['slots_per_tet'],d['energy']))"
done
