#!/usr/bin/env python3
"""Static instruction mix of a tile kernel per barrier-separated phase, from hipcc's -save-temps assembly.

    hipcc -c tssplat_amd/csrc/kernels.hip <DEVICE_FLAGS> -save-temps -o /tmp/k.o   (in a scratch directory)
    python tools/isa_phases.py /tmp/isa/kernels-hip-amdgcn-amd-amdhsa-gfx950.s [mangled-name-substring]

Counts are static (one pass through straight-line code; loops counted once), which is what the tile kernel's
passes are: fully unrolled over the slots of a lane.  Used for profiles/r03_issue_model.md.
"""
import re
import sys
from collections import Counter


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "tile_energy_kernelILb1ELi768ELi6ELb0ELb0ELi2E"
    lines = open(path).read().split("\n")
    start = next(i for i, ln in enumerate(lines) if ln.startswith("_ZN") and want in ln and ln.rstrip().split(":")[0].endswith("E") and ":" in ln)
    ph = [Counter()]
    for ln in lines[start + 1:]:
        t = ln.strip()
        if t.startswith(".Lfunc_end"):
            break
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_barrier":
            ph.append(Counter())
            continue
        cls = ("valu" if op.startswith("v_") else "lds" if op.startswith("ds_") else
               "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if op.startswith("s_") else "other")
        ph[-1][cls] += 1
        ph[-1][op] += 1
    tot = Counter()
    for i, c in enumerate(ph):
        tot.update(c)
        lds = {k: v for k, v in c.items() if k.startswith("ds_")}
        print(f"phase {i}: valu {c['valu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d} salu {c['salu']:3d}   {lds}")
    print("total:", {k: tot[k] for k in ("valu", "lds", "vmem", "salu")})
    print("top VALU:", [(k, v) for k, v in tot.most_common(60) if k.startswith("v_")][:30])


if __name__ == "__main__":
    main()
