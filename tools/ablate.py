#!/usr/bin/env python3
"""Price the stages of the tile kernel by switching them off (results are wrong while a switch is on).

    python tools/ablate.py [--scene kuhn19 --spheres 64] [--max-threads 768] ...

Prints ms per evaluation (HIP events around the tile kernel) for each ablation mask.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

MASKS = [
    (0, "full kernel"),
    (2, "pass-3 gathers read own slot (no bank conflicts)"),
    (4, "pass-2 gathers read own slot"),
    (6, "both gather passes local"),
    (128, "skip the per-vertex gather loop (stores zeros)"),
    (256, "skip the gradient/stage stores"),
    (384, "skip vertex gather and stores"),
    (8, "skip pass 3"),
    (24, "skip passes 2 and 3"),
    (32, "exit after pass 1"),
    (64, "exit after loads (stream only)"),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="kuhn19")
    ap.add_argument("--spheres", type=int, default=64)
    ap.add_argument("--sigma", type=float, default=0.02)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--max-threads", type=int, default=0)
    ap.add_argument("--lds-budget", type=int, default=0)
    ap.add_argument("--target-owned", type=int, default=0)
    ap.add_argument("--no-balance", action="store_true")
    ap.add_argument("--masks", default="")
    ap.add_argument("--shuffle", action="store_true")
    ap.add_argument("--spt", type=int, default=0)
    ap.add_argument("--no-conflict-aware", action="store_true")
    ap.add_argument("--rebuild-dminv", action="store_true")
    ap.add_argument("--lds-request", type=int, default=0, help="request this much dynamic LDS per workgroup (limits workgroups per CU)")
    ap.add_argument("--stamps-only", action="store_true", help="production kernel + stamps (-DTSAMD_STAMPS): no switches, no mask sweep")
    args = ap.parse_args()
    if args.lds_request:
        os.environ["TSAMD_LDS_REQUEST"] = str(args.lds_request)
    # the production library has the switches compiled out; build / select the ablation variant
    from tssplat_amd import _build
    vname, vflag = ("stamps", "-DTSAMD_STAMPS") if args.stamps_only else ("ablation", "-DTSAMD_ABLATION")
    variant = os.path.join(os.path.dirname(_build.LIB), f"libtssplat_amd_{vname}.so")
    if not os.path.exists(variant):
        variant = _build.build_variant(vname, [vflag])
    os.environ.setdefault("TSSPLAT_AMD_LIB", variant)
    import torch
    from tssplat_amd import _capi, scenes, tet_spheres_ext as T
    lib = _capi.load()
    sc = scenes.make_scene(args.scene, args.spheres)
    ts = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), max_threads=args.max_threads,
                      lds_budget_bytes=args.lds_budget, target_owned=args.target_owned,
                      balance_slots=not args.no_balance, debug_shuffle=int(args.shuffle) | (2 if args.no_conflict_aware else 0), slots_per_thread=args.spt,
                      rebuild_dminv=args.rebuild_dminv)
    info = ts.plan_info()
    x = torch.from_numpy(scenes.deform(sc, args.sigma)).cuda()
    g = torch.empty_like(x)
    e = torch.empty((), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    masks = MASKS if not args.masks else [(int(m), "custom") for m in args.masks.split(",")]
    if args.stamps_only:
        masks = [(0, "production kernel + stamps (unarmed)")]
    out = {"scene": f"{args.spheres} x {args.scene}", "m": sc.n_tets, "n": sc.n_vertices, "plan": info, "rows": []}
    for mask, what in masks:
        _capi.check(lib.tsamd_debug_set_ablation(ts._handle(), mask))
        for _ in range(3):
            _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 1e-4, 2e-4, 2, stream,
                                                   e.data_ptr(), g.data_ptr()))
        ts.set_timing(True)
        for _ in range(args.reps):
            _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 1e-4, 2e-4, 2, stream,
                                                   e.data_ptr(), g.data_ptr()))
        tile_ms, fin_ms, n = ts.get_timing()
        ts.set_timing(False)
        row = {"mask": mask, "what": what, "tile_ms": tile_ms / n, "finish_ms": fin_ms / n,
               "gtets_per_s": sc.n_tets / (tile_ms / n * 1e-3) / 1e9}
        out["rows"].append(row)
        print(f"mask {mask:3d} {what:50s} tile {row['tile_ms']:.4f} ms  finish {row['finish_ms']:.4f} ms  "
              f"{row['gtets_per_s']:.2f} Gtet/s", flush=True)
    _capi.check(lib.tsamd_debug_set_ablation(ts._handle(), 0))
    # per-phase shader-clock stamps of lane 0 of every wave of every tile
    import numpy as np
    nw = (info["block_threads"] + 63) // 64
    clk = np.zeros(256 * info["n_tiles"], dtype=np.int64)
    _capi.check(lib.tsamd_debug_read_clocks(ts._handle(), clk.ctypes.data, clk.size))      # arm
    _capi.check(lib.tsamd_forward_backward(ts._handle(), x.data_ptr(), None, 1e-4, 2e-4, 2, stream,
                                           e.data_ptr(), g.data_ptr()))
    _capi.check(lib.tsamd_debug_read_clocks(ts._handle(), clk.ctypes.data, clk.size))      # read
    raw = clk.reshape(-1, 16, 16)[:, :nw, :].astype(np.float64)
    clk = raw[:, :, :10]
    names = ["load+stage", "pass1", "pass2", "H write+reload issue", "pass3 (own wave)", "wait others",
             "force write", "vertex gather+stores", "energy reduce"]
    d = np.diff(clk, axis=2)                       # [tile, wave, phase]
    t0 = clk[:, :, 0].min(axis=1, keepdims=True)
    out["phase_cycles_mean_wave0"] = {n: float(d[:, 0, i].mean()) for i, n in enumerate(names)}
    print("phase cycles, wave 0 (mean over tiles): " + ", ".join(f"{n} {d[:, 0, i].mean():.0f}" for i, n in enumerate(names)) +
          f", total {(clk[:, 0, 9] - clk[:, 0, 0]).mean():.0f}")
    print("per wave: mean cycles spent in each phase (rows = waves)")
    print("wave " + " ".join(f"{n[:10]:>10s}" for n in names))
    for w in range(nw):
        print(f"{w:4d} " + " ".join(f"{d[:, w, i].mean():10.0f}" for i in range(len(names))))
    out["phase_cycles_by_wave"] = d.mean(axis=0).tolist()
    # arrival spread at every stamp: last wave minus first wave (what a barrier right after it costs the early ones)
    spread = clk.max(axis=1) - clk.min(axis=1)
    print("arrival spread (last wave - first wave) at each stamp: " + ", ".join(f"{spread[:, k].mean():.0f}" for k in range(10)))
    if args.stamps_only:
        # sub-stamps (cycles since the wave's first instruction, mean over tiles; waves 0, 5, 10):
        #   10 descriptor + vertex id arrived, position loads issued | 11 positions arrived | 1 barrier: positions staged |
        #   12 planes arrived + first F stored | 13 this wave's pass 1 done | 2 barrier | ... 4 H written | 14 first slot of pass 3 done | 5 pass 3 done
        order = [10, 11, 1, 12, 13, 2, 3, 4, 14, 5, 6, 7, 8, 9]
        rel = raw - raw[:, :, 0:1]
        for w in sorted({0, min(5, nw - 1), max(nw - 2, 0)}):
            print(f"wave {w:2d} cumulative: " + "  ".join(f"s{k}={rel[:, w, k].mean():.0f}" for k in order))
        out["substamps_cumulative_wave0"] = {str(k): float(rel[:, 0, k].mean()) for k in order}
        # are the workgroups of a launch in step with each other?  start time of every tile relative to the launch, and the
        # distribution of the memory phase (stamp 2 - stamp 0) over the launch
        start = raw[:, 0, 0] - raw[:, 0, 0].min()
        mem = raw[:, 0, 2] - raw[:, 0, 0]
        life = raw[:, 0, 9] - raw[:, 0, 0]
        qs = [0, 10, 25, 50, 75, 90, 100]
        print("memory phase (entry -> pass 1 done), percentiles over tiles: " + ", ".join(f"p{q}={np.percentile(mem, q):.0f}" for q in qs))
        print("workgroup lifetime, percentiles over tiles: " + ", ".join(f"p{q}={np.percentile(life, q):.0f}" for q in qs))
        print(f"launch span {start.max() + life[np.argmax(start)]:.0f} cycles")
        # tiles in their memory phase over time (64 bins across the launch): a convoy shows as a wave pattern
        t_end = (raw[:, 0, 9]).max() - raw[:, 0, 0].min()
        bins = np.linspace(0, t_end, 65)
        s0 = raw[:, 0, 0] - raw[:, 0, 0].min()
        s2 = raw[:, 0, 2] - raw[:, 0, 0].min()
        s9 = raw[:, 0, 9] - raw[:, 0, 0].min()
        mids = 0.5 * (bins[1:] + bins[:-1])
        in_mem = [(int(((s0 <= t) & (t < s2)).sum()), int(((s2 <= t) & (t < s9)).sum())) for t in mids]
        print("resident workgroups in (memory phase, compute phases) at 64 instants across the launch:")
        print(" ".join(f"{a}/{b}" for a, b in in_mem))
        out["residency_over_time"] = in_mem
    print(json.dumps(out))


if __name__ == "__main__":
    main()
