#!/usr/bin/env python3
"""Strong-scaling MODEL of the 512-sphere headline scene from ONE GPU.

At N-way strong scaling a rank owns 512 / N spheres and runs exactly what a 1-GPU job of 512 / N spheres runs, plus the
energy exchange.  Spheres share nothing, so the only cross-rank cost is that exchange and the max-over-ranks of equal
work.  This tool times the per-rank share on one GPU -- `bench.py --spheres 512/N --force-collective`, i.e. with a
single-rank RCCL group so that the per-step host cost of the exchange (device-slot copy + one all-reduce per window) is
inside the timed loop -- and reports

    predicted_speedup(N) = t_step(512 spheres) / t_step(512 / N spheres)

for every launch mode.  What it cannot see: xGMI latency of the 8-byte-per-step collective across real ranks (off the
gradient's path, asynchronous) and rank-to-rank clock differences.  Writes profiles/<round>_scaling_model.json.

    python tools/scaling_model.py r03 [--steps 200] [--window 16]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(spheres, steps, window, extra=()):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--spheres", str(spheres), "--steps", str(steps), "--warmup", "20",
           "--no-cpu-baseline", "--force-collective", "--energy-window", str(window), *extra]
    p = subprocess.run(cmd, capture_output=True, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if not lines:
        raise RuntimeError(f"bench failed for {spheres} spheres:\n{p.stderr[-3000:]}")
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("round", nargs="?", default="r03")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--window", type=int, default=16)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rows = {}
    for n in (1, 2, 4, 8):
        s = 512 // n
        rec = run(s, args.steps, args.window)
        modes = {rec["config"]["launch"].split(":")[0].split(" ")[0]: rec["ms_per_step"]}
        t = {"main": rec["ms_per_step"], "main_launch": rec["config"]["launch"],
             "eager_autograd": rec.get("eager_autograd_ms_per_step"), "graph_replay": rec.get("graph_replay_ms_per_step"),
             "graph_autograd": rec.get("graph_autograd_ms_per_step"), "tile_kernel_ms": rec["roofline"]["kernel_ms"],
             "finish_kernel_ms": rec["roofline"]["finish_kernel_ms"], "roofline_frac": rec["roofline"]["frac"],
             "energy_exchange": rec["config"]["energy_exchange"]}
        # fill the main mode's own slot
        for key, tag in (("eager_autograd", "eager"), ("graph_replay", "HIP-graph replay of"), ("graph_autograd", "SmoothnessBarrierEnergy(graph=True)")):
            if t[key] is None and rec["config"]["launch"].startswith(tag):
                t[key] = rec["ms_per_step"]
        # the module a trainer would hold, with its own energy exchange (one all-reduce per step off the training thread):
        # ShardedSmoothnessBarrierEnergy(graph=True, exchange="overlap").forward + backward()  
        mod = run(s, args.steps, args.window, ("--launch", "module"))
        t["sharded_module"] = mod["ms_per_step"]
        t["sharded_module_exchange"] = mod["config"]["energy_exchange"]
        # ... and with one collective per 16 steps (every=16): what the 8-way share of the headline scene needs -- its step is ~70 us
        mod16 = run(s, args.steps, args.window, ("--launch", "module", "--module-every", "16"))
        t["sharded_module_every16"] = mod16["ms_per_step"]
        rows[n] = {"spheres_per_rank": s, **t}
        print(n, json.dumps(rows[n]), flush=True)
    one = rows[1]
    model = {}
    for n in (2, 4, 8):
        r = rows[n]
        model[str(n)] = {k: (one[k] / r[k] if one.get(k) and r.get(k) else None)
                         for k in ("main", "eager_autograd", "graph_replay", "graph_autograd", "sharded_module", "sharded_module_every16", "tile_kernel_ms")}
        best1 = min(v for v in (one["eager_autograd"], one["graph_replay"], one["graph_autograd"]) if v)
        bestn = min(v for v in (r["eager_autograd"], r["graph_replay"], r["graph_autograd"]) if v)
        model[str(n)]["best_mode_each"] = best1 / bestn
    # one more point: the exchange issued every step (window 1) at the 8-way share
    w1 = run(64, args.steps, 1)
    out = {"what": "predicted strong-scaling speedup of the 512 x kuhn19 scene = t_step(512) / t_step(512 / N) measured on ONE MI355X, "
                   "per-rank share incl. the energy exchange through a single-rank RCCL group (bench.py --force-collective)",
           "energy_window": args.window, "per_rank_share_ms": rows, "predicted_speedup": model,
           "window_1_at_64_spheres_ms": {"main": w1["ms_per_step"], "launch": w1["config"]["launch"],
                                         "eager_autograd": w1.get("eager_autograd_ms_per_step"),
                                         "graph_replay": w1.get("graph_replay_ms_per_step"),
                                         "graph_autograd": w1.get("graph_autograd_ms_per_step")},
           "target": "north_star: >= 6x at 8 GPUs"}
    path = args.out or os.path.join(ROOT, "gpurun_out", f"{args.round}_scaling_model.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(model, indent=1))


if __name__ == "__main__":
    main()
