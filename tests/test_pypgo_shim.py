"""the tet-mesh container / Vega I/O subset of `pypgo` the reference's energy path
touches (geometry/tetrahedron_mesh.py:14-24,70-91), and -- where /root/reference is present -- the
reference's own energies/smooth_barrier.py imported UNMODIFIED on top of our shims."""
import importlib.util
import os
import sys

import numpy as np
import pytest

import pypgo
from tssplat_amd import scenes


def test_tetmesh_container_and_veg_roundtrip(tmp_path, aveg):
    rest, tets = aveg
    tm = pypgo.create_tetmesh(rest.flatten().astype(np.float32), tets.flatten().astype(np.int32), 100000, 0.45, 1000)
    v, e = pypgo.get_tetmesh_vertex_positions(tm), pypgo.get_tetmesh_element_indices(tm)
    assert v.shape == rest.shape and e.shape == tets.shape and e.dtype == np.int32
    assert np.array_equal(e, tets) and np.allclose(v, rest)
    moved = pypgo.update_tetmesh_vertices(tm, v + 0.5)
    assert moved is not tm and np.allclose(pypgo.get_tetmesh_vertex_positions(moved), v + 0.5)
    assert np.allclose(pypgo.get_tetmesh_vertex_positions(tm), v)            # the old mesh is untouched
    path = tmp_path / "out.veg"
    pypgo.save_tetmesh_to_file(moved, str(path))
    text = path.read_text()
    assert "*VERTICES" in text and "*ELEMENTS\nTET" in text and "*MATERIAL defaultMaterial" in text
    assert "ENU, 1000, 100000, 0.45" in text and text.rstrip().endswith("allElements, defaultMaterial")
    back = pypgo.create_tetmesh_from_file(str(path))
    assert np.array_equal(back.elements, tets) and np.allclose(back.vertices, v + 0.5)
    assert (back.E, back.nu, back.density) == (100000.0, 0.45, 1000.0)
    # and the C library's own reader (TetSpheres(filename), tet_spheres.cpp:108-117) agrees
    from tssplat_amd import tet_spheres_ext
    ts = tet_spheres_ext.TetSpheres(str(path), host_only=True)
    assert ts.n == rest.shape[0] and ts.nele == tets.shape[0]
    v2, t2 = scenes.read_veg(path)
    assert np.array_equal(t2, tets)


def test_unprovided_calls_are_loud():
    with pytest.raises(NotImplementedError, match="tetmesh_geometry.py:291"):
        pypgo.mesh_isotropic_remeshing(None, 0.1, 5, 180.0)
    with pytest.raises(AttributeError):
        pypgo.does_not_exist


@pytest.mark.skipif(not os.path.exists("/root/reference/energies/smooth_barrier.py"),
                    reason="the reference checkout only exists in the authoring container")
def test_reference_smooth_barrier_imports_unmodified():
    """The reference file does `import pypgo` and `from tet_spheres import tet_spheres_ext`
    (energies/smooth_barrier.py:1,6): both resolve to this repo, nothing in the file is edited."""
    spec = importlib.util.spec_from_file_location("ref_smooth_barrier", "/root/reference/energies/smooth_barrier.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from tssplat_amd import tet_spheres_ext
    assert mod.tet_spheres_ext is tet_spheres_ext
    assert {"SmoothnessBarrierFunc", "SmoothnessBarrierEnergy"} <= set(dir(mod))
    # schedule of the reference class == schedule of our mirror
    from tssplat_amd.energies import SmoothnessBarrierEnergy

    class F:
        smooth_eng_coeff, barrier_coeff, increase_order_iter = 2e-4, 2e-4, 1000

    fake = type("M", (), {"FLAGS": F})()
    for it in (0, 7, 600, 1200, 3000):
        assert mod.SmoothnessBarrierEnergy.coeff_scheduler(fake, it) == SmoothnessBarrierEnergy.coeff_scheduler(fake, it)
