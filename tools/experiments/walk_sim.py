import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tssplat_amd import geometry, scenes
def counts(pos, tri, H, W):
    # pos [V,4] clip; returns candidate bbox counts per triangle (0 if invalid)
    w = pos[:,3]; ok = w > 0
    X = (pos[:,0]/w*0.5+0.5)*W*256; Y = (pos[:,1]/w*0.5+0.5)*H*256
    X = np.rint(X).astype(np.int64); Y = np.rint(Y).astype(np.int64)
    t = tri
    okt = ok[t].all(1)
    x = X[t]; y = Y[t]
    area = (x[:,1]-x[:,0])*(y[:,2]-y[:,0]) - (y[:,1]-y[:,0])*(x[:,2]-x[:,0])
    minx, maxx = x.min(1), x.max(1); miny, maxy = y.min(1), y.max(1)
    px0 = np.maximum(0, (minx-128+255)>>8); px1 = np.minimum(W-1, (maxx-128)>>8)
    py0 = np.maximum(0, (miny-128+255)>>8); py1 = np.minimum(H-1, (maxy-128)>>8)
    valid = okt & (area != 0) & (px0<=px1) & (py0<=py1)
    w_ = np.where(valid, px1-px0+1, 0); h_ = np.where(valid, py1-py0+1, 0)
    return w_, h_, valid
def report(name, ws, hs, valid):
    c = ws*hs
    n = len(c); pad = (-n) % 64
    cc = np.concatenate([c, np.zeros(pad, np.int64)]).reshape(-1,64)
    print(f"== {name}: {n} (view,tri) lanes, valid {valid.mean():.3f}, mean cand over all {c.mean():.2f}, over valid {c[valid].mean():.2f}, max {c.max()}")
    ideal = cc.sum(1)/64
    cur = cc.max(1)
    print(f"   current iterations/wave: mean {cur.mean():.1f}; ideal (sum/64) {ideal.mean():.2f}; utilisation {ideal.sum()/cur.sum():.3f}")
    vv = np.concatenate([valid, np.zeros(pad,bool)]).reshape(-1,64)
    print(f"   waves with no valid lane {np.mean(~vv.any(1)):.3f}; mean valid lanes/wave {vv.sum(1).mean():.1f}")
    W_ = np.concatenate([ws, np.zeros(pad, np.int64)]).reshape(-1,64); H_ = np.concatenate([hs, np.zeros(pad, np.int64)]).reshape(-1,64)
    for T in (4,8,16,32):
        # chunk: R rows with R = pow2 floor(T/w) (>=1); chunk cands = R*w (w<=T) else split columns
        w = np.maximum(W_,1); 
        ncol = (w + T-1)//T; wseg = (w + ncol-1)//ncol
        R = np.maximum(1, T//wseg); R = 2**np.floor(np.log2(R)).astype(np.int64)
        nrow = (H_ + R-1)//R
        nch = np.where(W_>0, ncol*nrow, 0)
        chunk_cost = R*wseg   # upper bound per chunk
        rounds = (nch.sum(1)+63)//64
        # iteration estimate: per round max chunk cost (~ T) 
        it = rounds * np.minimum(T, np.where(nch.sum(1)>0, (chunk_cost*(nch>0)).max(1), 0))
        for ov in (1.5, 3.0):
            tot = it + rounds*ov
            print(f"   T={T:2d}: chunks/wave {nch.sum(1).mean():6.1f} rounds {rounds.mean():5.2f} iter {it.mean():6.1f} (+{ov}/round: {tot.mean():6.1f})  vs current {cur.mean():.1f}")
    # block-level (256) compaction of valid lanes then current walk
    nb = (len(c)+255)//256; pad2 = nb*256-len(c)
    c4 = np.concatenate([c, np.zeros(pad2,np.int64)]).reshape(nb,256)
    srt = -np.sort(-c4,axis=1)  # sorted descending within block: best case dealing
    print(f"   block(256) sort-by-size then waves of 64: iterations/wave {srt.reshape(nb,4,64).max(2).mean():.1f}")
which = sys.argv[1]
if which == 'dense':
    sc = scenes.make_scene('kuhn19', 512)
    vid, faces = geometry.get_surface_vf(sc.tets)
    v = scenes.deform(sc, 0.02)[np.asarray(vid)]
    mvp = scenes.orbit_mvps(8)
    pos = scenes.transform_pos(mvp, v)
    ws=[];hs=[];vs=[]
    for b in range(2):
        a,b_,c_ = counts(pos[b].astype(np.float64), np.asarray(faces), 512, 512); ws.append(a);hs.append(b_);vs.append(c_)
    report('dense 512xkuhn19, 2 of 8 views', np.concatenate(ws), np.concatenate(hs), np.concatenate(vs))
else:
    sc = scenes.make_scene('kuhn8', 20)
    vid, faces = geometry.get_surface_vf(sc.tets)
    v = sc.rest[np.asarray(vid)]
    # spheres ~ radius 0.1-0.24: emulate train_object by scaling each sphere? use scene as is but scale to typical size
    mvp = scenes.dataset_mvps(120) if hasattr(scenes,'dataset_mvps') else scenes.orbit_mvps(120)
    pos = scenes.transform_pos(mvp[:6], v)
    ws=[];hs=[];vs=[]
    for b in range(6):
        a,b_,c_ = counts(pos[b].astype(np.float64), np.asarray(faces), 512, 512); ws.append(a);hs.append(b_);vs.append(c_)
    report('20 x kuhn8 (config-5 like), 6 views', np.concatenate(ws), np.concatenate(hs), np.concatenate(vs))

def classes(c):
    k = np.where(c < 32, c, 32 + 2*(np.floor(np.log2(np.maximum(c,32))).astype(np.int64)-5) + ((c >> np.maximum(np.floor(np.log2(np.maximum(c,32))).astype(np.int64)-1,0)) & 1))
    return k
def sim_sort(name, c):
    for N in (256, 512, 1024, 2048):
        nb = (len(c)+N-1)//N; pad = nb*N-len(c)
        cb = np.concatenate([c, np.zeros(pad,np.int64)]).reshape(nb,N)
        k = classes(cb)
        order = np.argsort(-k, axis=1, kind='stable')
        srt = np.take_along_axis(cb, order, axis=1)
        it = srt.reshape(nb, N//64, 64).max(2)
        print(f"   {name}: class-sort in blocks of {N}: iterations per 64 items {it.mean():.2f} (ideal {c.mean():.2f}); utilisation {c.sum()/64/it.sum():.3f}")
sim_sort(which, np.concatenate(ws)*np.concatenate(hs))


def sim_coop(name, c, overhead=3.0):
    """Cooperative walk of big boxes: a wave walks its boxes of <= T candidates lane-per-triangle, then every bigger box with all 64 lanes."""
    pad = (-len(c)) % 64
    cc = np.concatenate([c, np.zeros(pad, np.int64)]).reshape(-1, 64)
    cur = cc.max(1)
    for T in (16, 32, 64, 128, 256):
        small = np.where(cc <= T, cc, 0).max(1)
        big = np.where(cc > T, (cc + 63) // 64 + overhead, 0).sum(1)
        it = small + big
        print(f"   {name}: cooperative walk above T={T:3d}: iterations/wave {it.mean():6.1f} (p99 {np.percentile(it, 99):6.1f}, max {it.max():6.1f})   now {cur.mean():.1f} (p99 {np.percentile(cur, 99):.1f}, max {cur.max()})")


sim_coop(which, np.concatenate(ws) * np.concatenate(hs))
