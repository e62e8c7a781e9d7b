"""Kernel descriptors of the gfx950 code object the library was built with (no GPU needed).

The tile kernels address LDS by absolute byte address (kernels.hip: lds_at): correct only while they have no static
LDS object, which the library re-checks at run time (configure_kernels) and this test checks at build time.  Their
occupancy (two 768-thread workgroups per CU) needs <= 80 VGPRs without spills."""
import os
import sys

import pytest

from tssplat_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="needs the ROCm LLVM tools")
def test_tile_kernels_have_no_static_lds_and_do_not_spill():
    import kernel_metadata
    _capi.load()
    meta = kernel_metadata.kernel_metadata(_capi.lib_path())
    tiles = {k: v for k, v in meta.items() if "tile_energy_kernel" in k}
    # 2 slots x 768 threads: built-in, explicit operator, rebuild_dminv; the three fat-wave layouts (3 x 512, 4 x 768, 3 x 1024):
    # built-in only -- each with / without gradient
    assert len(tiles) == 12, sorted(tiles)
    for name, rec in tiles.items():
        assert rec["group_segment_fixed_size"] == 0, (name, rec)      # dynamic LDS array at LDS address 0
        assert rec["vgpr_spill_count"] == 0 and rec["private_segment_fixed_size"] == 0, (name, rec)
    default = [v for k, v in tiles.items() if "ILb1ELi768ELi6ELb0ELb0ELi2E" in k]
    assert len(default) == 1 and default[0]["vgpr_count"] <= 80, default      # 6 waves per SIMD: two workgroups per CU
    for k, v in tiles.items():                                                # the budgets the launch bounds stand for
        cap = {"Li768ELi6E": 80, "Li512ELi4E": 128, "Li1024ELi4E": 128, "Li768ELi3E": 168}
        assert v["vgpr_count"] <= next(c for key, c in cap.items() if key in k), (k, v)
    for name, rec in meta.items():                                          # every kernel of every translation unit
        if "antialias" in name and "masked" not in name:
            # the table-free antialias kernels are held to 64 VGPRs on purpose: the float64 silhouette analysis spills, the lanes
            # that only compare two ids run at eight waves per SIMD; the masked kernels (no lane per pixel) spill nothing
            assert rec["vgpr_count"] <= 64 and rec["private_segment_fixed_size"] <= 256, (name, rec)
            continue
        assert rec["vgpr_spill_count"] == 0 and rec["private_segment_fixed_size"] == 0, (name, rec)
    # the rasteriser's pixel walk is sized for eight waves per SIMD (drain(): three reads in flight, not four)
    bins = [v for k, v in meta.items() if "rasterize_bin_kernel" in k]
    assert len(bins) == 1 and bins[0]["vgpr_count"] <= 64, bins
    assert len(meta) >= 28                                                  # all bundles of .hip_fatbin were read, not just the first


def test_renderer_kernels_are_built_without_fp_contraction():
    """Bit-exact triangle ids / identical silhouette decisions need separately rounded multiplies and adds: the two renderer
    translation units must be compiled with -ffp-contract=off AFTER the global -ffp-contract=fast (the last flag wins)."""
    from tssplat_amd import _build
    for src in ("raster_kernels.hip", "aa_kernels.hip"):
        assert src in _build.SOURCES
        assert _build.SOURCE_FLAGS.get(src) == ["-ffp-contract=off"]
    assert "-ffp-contract=fast" in _build.DEVICE_FLAGS                      # the energy kernels want the fused forms
