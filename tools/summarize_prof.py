#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (gpurun_out/...) into the small summaries kept under profiles/.

    python tools/summarize_prof.py --round r01 --label "512 x kuhn19" --workload kuhn19x512 \
        --stats gpurun_out/prof_stats --pmc gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq1 ...

Writes profiles/<round>_kernel_stats.csv (the rocprofv3 --stats table), profiles/<round>_pmc.json
(per-kernel counter averages per launch) and updates profiles/traffic.json, which bench.py reads
for ``roofline.traffic``.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in
KiB and come from separate --pmc passes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for
wide coalesced reads, so the read side is doubled (cross-checked here against TCC_MISS x 128 B
when that counter was collected).
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name: str) -> str:
    for key in ("tile_energy_kernel", "finish_kernel", "scale_kernel", "absmax_kernel", "clamp_kernel"):
        if key in name:
            return key + ("<true>" if "<true" in name else "<false>" if "<false" in name else "")
    return name[:60]


def _traffic_digest():
    sys.path.insert(0, ROOT)
    from tssplat_amd import _build
    return _build.traffic_digest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("--label", default="")
    ap.add_argument("--workload", default="")
    ap.add_argument("--stats", default="")
    ap.add_argument("--pmc", nargs="*", default=[])
    args = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    if args.stats:
        # (gpurun MERGES a call's output into gpurun_out/: an earlier round-trip may have left an older file next to the new one)
        src = sorted(glob.glob(os.path.join(args.stats, "*", "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
        if src:
            shutil.copy(src[0], os.path.join(out_dir, f"{args.round}_kernel_stats.csv"))
            print("wrote", f"profiles/{args.round}_kernel_stats.csv")

    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for d in args.pmc:
        for f in sorted(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime, reverse=True)[:1]:
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                meta[k] = {"vgpr": int(r["VGPR_Count"]), "accum_vgpr": int(r["Accum_VGPR_Count"]),
                           "sgpr": int(r["SGPR_Count"]), "lds_bytes": int(r["LDS_Block_Size"]),
                           "scratch_bytes": int(r["Scratch_Size"]), "workgroup": int(r["Workgroup_Size"]),
                           "grid": int(r["Grid_Size"])}
    summary = {"label": args.label, "kernels": {}}
    for k, counters in agg.items():
        if not any(t in k for t in ("tile_energy", "finish", "scale")):
            continue
        row = {c: sum(v) / len(v) for c, v in counters.items()}
        row["_launches_sampled"] = {c: len(v) for c, v in counters.items()}
        row["_dispatch"] = meta[k]
        if "FETCH_SIZE" in row:
            row["hbm_read_bytes_raw"] = row["FETCH_SIZE"] * 1024
            row["hbm_read_bytes_gfx950_corrected"] = 2 * row["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in row:
            row["hbm_write_bytes"] = row["WRITE_SIZE"] * 1024
        if "TCC_MISS_sum" in row:
            row["l2_miss_bytes_at_128B"] = row["TCC_MISS_sum"] * 128
            row["l2_hit_rate"] = row["TCC_HIT_sum"] / (row["TCC_HIT_sum"] + row["TCC_MISS_sum"])
        summary["kernels"][k] = row
    if summary["kernels"]:
        path = os.path.join(out_dir, f"{args.round}_pmc.json")
        json.dump(summary, open(path, "w"), indent=1, sort_keys=True)
        print("wrote", os.path.relpath(path, ROOT))
        tile = next((v for k, v in summary["kernels"].items() if "tile_energy_kernel<true>" in k), None)
        if tile and args.workload and "hbm_read_bytes_gfx950_corrected" in tile and "hbm_write_bytes" in tile:
            tpath = os.path.join(out_dir, "traffic.json")
            rec = json.load(open(tpath)) if os.path.exists(tpath) else {}
            rec[args.workload] = {
                "round": args.round,
                "kernel": "tile_energy_kernel<true>",
                "hbm_bytes_per_launch": tile["hbm_read_bytes_gfx950_corrected"] + tile["hbm_write_bytes"],
                "read_bytes": tile["hbm_read_bytes_gfx950_corrected"],
                "write_bytes": tile["hbm_write_bytes"],
                "note": "FETCH_SIZE x 1024 x 2 (gfx950 wide-read correction) + WRITE_SIZE x 1024, separate --pmc passes",
                # fingerprint of kernels.hip + plan.cpp + plan.h at the time of the measurement: bench.py drops the figure when
                # the sources it runs have moved on (a static traffic.json goes stale silently)
                "sources": _traffic_digest(),
            }
            json.dump(rec, open(tpath, "w"), indent=1, sort_keys=True)
            print("updated profiles/traffic.json:", rec[args.workload])


if __name__ == "__main__":
    main()
