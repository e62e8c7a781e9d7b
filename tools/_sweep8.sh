mkdir -p gpurun_out/r2j
python -m pytest tests/test_gpu_parity.py -x -q -k "multi_tile or config2 or random_tiling or delaunay" 2>&1 | tail -2
for d in 4 0 $((40<<8)); do
  python - <<PY
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from tssplat_amd import scenes, tet_spheres_ext as T, _capi
lib=_capi.load()
sc=scenes.make_scene("kuhn19",512)
t0=time.time()
ts=T.TetSpheres(sc.rest.reshape(-1),sc.tets.reshape(-1),debug_shuffle=$d)
tp=time.time()-t0
x=torch.from_numpy(scenes.deform(sc,0.02)).cuda(); g=torch.empty_like(x); e=torch.empty((),device="cuda"); st=torch.cuda.current_stream().cuda_stream
for _ in range(60): _capi.check(lib.tsamd_forward_backward(ts._handle(),x.data_ptr(),None,1e-4,2e-4,2,st,e.data_ptr(),g.data_ptr()))
ts.set_timing(True)
for _ in range(40): _capi.check(lib.tsamd_forward_backward(ts._handle(),x.data_ptr(),None,1e-4,2e-4,2,st,e.data_ptr(),g.data_ptr()))
a,b,n=ts.get_timing()
print("debug_shuffle=$d plan %.1f s tile %.4f ms finish %.4f ms"%(tp,a/n,b/n))
PY
done
