"""Streaming-tile plan (csrc/stream_plan.h), CPU: the plan is replayed step by step in float64 with tagged rings
(tests/stream_emulator.py: every ring read asserts that the band it expects is the band that is there) and compared with the
oracle; plan statistics; error paths.  GPU parity of the kernel that consumes the plan: tests/test_stream_gpu.py."""
import numpy as np
import pytest

from oracle import tet_energy_oracle as O
from tssplat_amd import scenes
from tssplat_amd.stream import StreamTetSpheres

import stream_emulator as SE


@pytest.mark.parametrize("kind,S,sigma,order", [("kuhn4", 3, 0.3, 2), ("kuhn8", 2, 0.02, 4), ("kuhn12", 1, 0.3, 2),
                                               ("cone", 1, 0.3, 4), ("delaunay700", 2, 0.3, 2), ("kuhn3", 5, 0.3, 2)])
def test_stream_plan_replays_to_oracle(kind, S, sigma, order):
    sc = scenes.make_scene(kind, S)
    st = StreamTetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True)
    info = st.plan_info()
    assert info["n_tets"] == sc.n_tets and info["total_slots"] >= sc.n_tets and info["band_slots"] == 256
    x = scenes.deform(sc, sigma)
    c1, c2 = 5e-5, 2e-4
    cache = O.prepare(sc.rest, sc.tets)
    E, Es, Eb, g = O.energy_and_grad(x, cache, c1, c2, order, grad_output=0.5)
    E2, Es2, Eb2, g2 = SE.emulate(st, x, c1, c2, order, grad_output=0.5)
    assert abs(Es - Es2) <= 1e-11 * max(Es, 1e-300) and abs(Eb - Eb2) <= 1e-11 * max(Eb, 1e-300)
    assert np.abs(g - g2).max() <= 1e-10 * np.abs(g).max()


def test_stream_plan_needs_fewer_slots_than_blobs():
    """The point of the exercise: kuhn_ball(19) in 2 x 2 tubes, ~1.08 slots per tet against the blob tiling's ~1.28."""
    from tssplat_amd import tet_spheres_ext as T
    sc = scenes.make_scene("kuhn19", 1)
    st = StreamTetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True)
    info = st.plan_info()
    blob = T.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True).plan_info()
    assert info["n_tubes"] == 4 and info["total_slots"] / sc.n_tets < 1.10 < blob["total_slots"] / sc.n_tets
    assert info["max_vertex_slots"] <= 1024 and info["lds_bytes"] <= 160 * 1024


def test_stream_plan_rejects_bad_meshes():
    sc = scenes.make_scene("kuhn2", 1)
    bad = sc.tets.copy().reshape(-1)
    bad[5] = sc.n_vertices + 3
    with pytest.raises(RuntimeError, match="out of range"):
        StreamTetSpheres(sc.rest.reshape(-1), bad, host_only=True)
    flat = sc.rest.copy()
    flat[sc.tets[0]] = flat[sc.tets[0, 0]]
    with pytest.raises(RuntimeError, match="singular"):
        StreamTetSpheres(flat.reshape(-1), sc.tets.reshape(-1), host_only=True)
