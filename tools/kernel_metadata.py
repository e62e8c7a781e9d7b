#!/usr/bin/env python3
"""Kernel descriptors (VGPRs, spills, static LDS) of the gfx950 code object inside libtssplat_amd.so.

    python tools/kernel_metadata.py [path/to/lib.so]

Used by tests/test_build_metadata.py: the tile kernels address LDS by absolute byte address (kernels.hip: lds_at), which
is only right while they have NO static LDS object (group_segment_fixed_size == 0), and their occupancy depends on the
register count staying within the launch bounds without spills.  Needs the LLVM tools that ship with ROCm.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_metadata(lib: str) -> dict[str, dict[str, int]]:
    """Every kernel of every translation unit: .hip_fatbin holds one offload bundle per .hip file, back to back."""
    out: dict[str, dict[str, int]] = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(d, "copy.so")])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for k, a in enumerate(starts):
            part, co = os.path.join(d, f"bundle{k}.bin"), os.path.join(d, f"co{k}.elf")
            with open(part, "wb") as fh:
                fh.write(blob[a:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", co], text=True)
            for block in notes.split("  - .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                rec = {}
                for key in ("group_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_count", "sgpr_spill_count",
                            "private_segment_fixed_size", "max_flat_workgroup_size"):
                    m = re.search(r"\." + key + r":\s+(\d+)", block)
                    if m:
                        rec[key] = int(m.group(1))
                out[name] = rec
    return out


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "tssplat_amd", "libtssplat_amd.so")
    for name, rec in sorted(kernel_metadata(lib).items()):
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        print(f"{short[:110]:110s} vgpr {rec.get('vgpr_count'):3d} spill {rec.get('vgpr_spill_count')} scratch {rec.get('private_segment_fixed_size')} "
              f"static-LDS {rec.get('group_segment_fixed_size')}")


if __name__ == "__main__":
    main()
