"""CPU tests of the host side of the product: the tiling plan built by libtssplat_amd.so
(``host_only=True`` never touches HIP), replayed in float64 by tests/tile_emulator.py against
the oracle, plus the C ABI's error behaviour.  No compute kernel runs here."""
import ctypes as C
import re

import numpy as np
import pytest

from oracle import tet_energy_oracle as O
from tssplat_amd import _capi, scenes
import tile_emulator as TE


@pytest.fixture(scope="module")
def ext():
    from tssplat_amd import tet_spheres_ext
    return tet_spheres_ext


def _check(ext, sc, kw, cases=((0.3, 4, 0.5),)):
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True, **kw)
    # rebuild_dminv plans carry rest positions, not the fp32-rounded Dm^-1: their exact replay is the unrounded operator
    cache = O.prepare(sc.rest, sc.tets, round_fp32=not kw.get("rebuild_dminv", False))
    assert np.array_equal(TE.adjacency(ts), cache.nbr)
    for sigma, order, go in cases:
        x = scenes.deform(sc, sigma)
        E, Es, Eb, g = O.energy_and_grad(x, cache, 5e-5, 2e-4, order, grad_output=go)
        E2, Es2, Eb2, g2 = TE.emulate(ts, x, 5e-5, 2e-4, order, grad_output=go)
        assert abs(E - E2) <= 1e-12 * abs(E)
        assert abs(Es - Es2) <= 1e-12 * Es and abs(Eb - Eb2) <= 1e-12 * max(Eb, 1e-300)
        assert np.abs(g - g2).max() <= 1e-11 * np.abs(g).max()
    return ts


@pytest.mark.parametrize("kind,S,kw", [
    ("kuhn8", 3, {}),                                  # default: two workgroups per CU, 2 tets per lane
    ("kuhn8", 3, dict(max_threads=768, lds_budget_bytes=163840)),   # one workgroup per CU
    ("kuhn8", 2, dict(lds_budget_bytes=40000)),        # forced multi-tile
    ("kuhn3", 40, {}),                                 # many tiny spheres packed into shared tiles
    ("kuhn12", 1, {}),                                 # ~10k tets: bisected
    ("kuhn12", 1, dict(max_threads=256, lds_budget_bytes=40960)),
    ("kuhn12", 1, dict(debug_flags=2)),
    ("cone", 2, {}),                                   # hub vertex of valence 1280
    ("cone", 1, dict(lds_budget_bytes=30000)),
    ("delaunay700", 2, {}),                            # unstructured: irregular valence, holes, no index locality
    ("delaunay2500", 1, dict(lds_budget_bytes=50000, max_threads=512)),
    ("kuhn12", 1, dict(rebuild_dminv=True)),           # rest positions instead of Dm^-1 planes
    ("kuhn3", 40, dict(rebuild_dminv=True)),
    ("cone", 2, dict(rebuild_dminv=True)),
    ("delaunay700", 2, dict(rebuild_dminv=True, lds_budget_bytes=50000, max_threads=512)),
])
def test_plan_replays_to_oracle(ext, kind, S, kw):
    sc = scenes.make_scene(kind, S)
    ts = _check(ext, sc, kw, cases=((0.02, 2, 1.0), (0.3, 4, 0.5)))
    info = ts.plan_info()
    assert info["n_tets"] == sc.n_tets and info["n_vertices"] == sc.n_vertices
    # sliver removal can split a Delaunay ball into several face-connected pieces
    assert info["n_components"] == S or (kind.startswith("delaunay") and info["n_components"] >= S)
    assert info["total_slots"] >= sc.n_tets
    assert info["block_threads"] % 64 == 0 and 64 <= info["block_threads"] <= 768
    assert info["lds_bytes"] <= (kw.get("lds_budget_bytes") or 81920)
    assert info["slots_per_thread"] == 2
    assert info["slots_per_thread"] * info["block_threads"] >= info["max_slots"]
    assert info["n_planes"] == (4 if kw.get("rebuild_dminv") else 13)


def test_real_mesh_plan(ext, aveg):
    rest, tets = aveg
    sc = scenes.replicate_spheres(rest.astype(np.float64), tets, 2, seed=3)
    ts = _check(ext, sc, {}, cases=((0.05, 2, 1.0),))
    info = ts.plan_info()
    assert info["n_tiles"] >= 2 * 7 and info["shared_vertex_copies"] > 0
    # halo overhead of the partitioner on a TetWild-quality mesh stays moderate
    assert info["total_slots"] / info["n_tets"] < 1.6


def test_small_components_are_packed(ext):
    sc = scenes.make_scene("kuhn3", 40)            # 162 tets each
    ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True)
    info = ts.plan_info()
    assert info["n_tiles"] < 10 and info["total_slots"] == sc.n_tets and info["shared_vertex_copies"] == 0


def test_isolated_vertices_get_zero_gradient(ext):
    v, t = scenes.kuhn_ball(2)
    rest = np.concatenate([v, [[5.0, 5.0, 5.0], [6.0, 6.0, 6.0]]]).astype(np.float32)   # two unreferenced vertices
    ts = ext.TetSpheres(rest.reshape(-1), t.reshape(-1), host_only=True)
    x = rest + 0.05 * np.random.default_rng(0).standard_normal(rest.shape).astype(np.float32)
    _, _, _, g = TE.emulate(ts, x, 1.0, 1.0, 2)
    assert np.all(g[-2:] == 0.0)


def test_veg_roundtrip_and_file_constructor(ext, tmp_path, aveg):
    rest, tets = aveg
    path = tmp_path / "mesh.veg"
    scenes.write_veg(path, rest[:200], np.array([[0, 1, 2, 3]], dtype=np.int32))   # minimal valid file
    v, t = scenes.read_veg(path)
    assert np.array_equal(t, [[0, 1, 2, 3]]) and np.allclose(v, rest[:200])
    scenes.write_veg(path, rest, tets)
    ts = ext.TetSpheres(str(path), host_only=True)
    assert ts.n == rest.shape[0] and ts.nele == tets.shape[0]
    ref = ext.TetSpheres(rest.reshape(-1), tets.reshape(-1), host_only=True)
    assert np.array_equal(TE.adjacency(ts), TE.adjacency(ref))
    with pytest.raises(RuntimeError):
        ext.TetSpheres(str(tmp_path / "missing.veg"), host_only=True)


def test_error_behaviour(ext, capsys):
    v, t = scenes.kuhn_ball(2)
    v32 = v.astype(np.float32)
    # index out of range
    bad = t.copy()
    bad[3, 2] = 10_000
    with pytest.raises(RuntimeError, match="out of range"):
        ext.TetSpheres(v32.reshape(-1), bad.reshape(-1), host_only=True)
    # non-manifold: three tets on one face
    nm_v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, -1], [1, 1, 1]], dtype=np.float32)
    nm_t = np.array([[0, 1, 2, 3], [0, 2, 1, 4], [0, 1, 2, 5]], dtype=np.int32)
    with pytest.raises(RuntimeError, match="non-manifold"):
        ext.TetSpheres(nm_v.reshape(-1), nm_t.reshape(-1), host_only=True)
    # singular rest tet
    flat = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0]], dtype=np.float32)
    with pytest.raises(RuntimeError, match="singular"):
        ext.TetSpheres(flat.reshape(-1), np.array([0, 1, 2, 3], dtype=np.int32), host_only=True)
    # 2-D arrays: the reference prints and returns an unusable object (tet_spheres.cpp:238-250)
    empty = ext.TetSpheres(v32, t.reshape(-1), host_only=True)
    assert "Wrong vertex type" in capsys.readouterr().err
    with pytest.raises(RuntimeError, match="empty"):
        empty.plan_info()
    # a host_only handle refuses every device entry point, loudly
    ts = ext.TetSpheres(v32.reshape(-1), t.reshape(-1), host_only=True)
    lib = _capi.load()
    rc = lib.tsamd_forward(ts._handle(), 1, 1.0, 1.0, 2, None, 1)
    assert rc == 7 and b"host_only" in lib.tsamd_last_error()
    # forcecast semantics: float64 / int64 inputs are converted like py::array::forcecast does
    ts64 = ext.TetSpheres(v.reshape(-1), t.astype(np.int64).reshape(-1), host_only=True)
    assert ts64.nele == t.shape[0]
    # options struct versioning
    o = _capi.make_options(host_only=1)
    o.struct_size = 4
    h = C.c_void_p()
    assert lib.tsamd_create(v32.ctypes.data, v32.shape[0], t.ctypes.data, t.shape[0], C.byref(o), C.byref(h)) == 1
    # ... and a caller compiled against another ABI version is rejected even when the struct size happens to agree
    o = _capi.make_options(host_only=1)
    o.abi_version = _capi.ABI_VERSION - 1
    assert lib.tsamd_create(v32.ctypes.data, v32.shape[0], t.ctypes.data, t.shape[0], C.byref(o), C.byref(h)) == 1
    assert b"abi_version" in lib.tsamd_last_error()


def test_train_loop_abi_checks_arguments():
    lib = _capi.load()
    assert lib.tsamd_train_loop_workspace_bytes(0) == -1 and lib.tsamd_train_loop_workspace_bytes(32) == 256
    out = C.c_void_p()
    assert lib.tsamd_train_loop_create(None, None, None, None, None, None, None, 4, C.byref(out)) == 1 and not out
    assert lib.tsamd_train_loop_launch(None, None, None, None, 0.1, 0.9, 0.999, 1, None, None) == 1
    assert b"null" in lib.tsamd_last_error()
    lib.tsamd_train_loop_destroy(None)                            # a no-op, like free(NULL)


def test_library_exports_every_declared_symbol():
    """include/tssplat_amd.h is the contract: every function it declares is exported, and the library exports no other
    `tsamd_` symbol (round 5: the diagnostic switches and the parked streaming path are gone from the product)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "tssplat_amd.h")).read()
    assert "tsamd_stream_" not in header and "tsamd_debug_" not in header
    declared = set(re.findall(r"\b(tsamd_[a-z0-9_]+)\s*\(", header))
    declared -= {"tsamd_options", "tsamd_plan_info", "tsamd_tile_view", "tsamd_status", "tsamd_handle"}
    lib = C.CDLL(_capi.lib_path())
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert declared == set(_capi.SIGNATURES), (declared ^ set(_capi.SIGNATURES))
    nm = subprocess.run(["nm", "-D", "--defined-only", _capi.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln and ln.split()[-1].startswith("tsamd_")}
    assert exported == declared, f"exported but not declared in the header: {sorted(exported - declared)}"
    assert b"gfx950" in _capi.load().tsamd_version()
    assert _capi.load().tsamd_abi_version() == _capi.ABI_VERSION == int(re.search(r"#define TSAMD_ABI_VERSION (\d+)", header).group(1))


def test_import_path_shim_and_operator_surface(capsys):
    """`from tet_spheres import tet_spheres_ext` (energies/smooth_barrier.py:6) must resolve here."""
    from tet_spheres import tet_spheres_ext as shim
    from tssplat_amd import tet_spheres_ext as real
    assert shim is real
    for name in ("TetSpheres", "forward", "backward", "random_x", "grad_limit"):
        assert hasattr(shim, name)
    from tssplat_amd.energies import SmoothnessBarrierEnergy, SmoothnessBarrierFunc
    import inspect
    # same apply(x_cur, tet_sp, c1, c2, order) call as smooth_barrier.py:9-31 (ctx-first spelling: see the class docstring)
    assert list(inspect.signature(SmoothnessBarrierFunc.forward).parameters) == ["ctx", "x_cur", "tet_sp", "c1", "c2", "order"]
    assert list(inspect.signature(SmoothnessBarrierEnergy.forward).parameters) == ["self", "x", "it", "c1", "c2"]
    rx = real.random_x(type("T", (), {"n": 7})())
    assert tuple(rx.shape) == (7, 3)


def test_coeff_schedule_matches_reference_formula():
    """energies/smooth_barrier.py:47-58: x1 at it=0, x16 from it=1200; order switches after 1000."""
    import math
    from tssplat_amd.energies import SmoothnessBarrierEnergy

    class F:
        smooth_eng_coeff, barrier_coeff, increase_order_iter = 2e-4, 3e-4, 1000

    fake = type("M", (), {"FLAGS": F})()
    for it in (0, 1, 300, 600, 1199, 1200, 5000):
        c1, c2 = SmoothnessBarrierEnergy.coeff_scheduler(fake, it)
        mult = math.pow(2, abs(math.sin(min(it / 300.0 / 4 * 0.5 * math.pi, 0.5 * math.pi))) * 4)
        assert c1 == F.smooth_eng_coeff * mult and c2 == F.barrier_coeff * mult
    assert SmoothnessBarrierEnergy.coeff_scheduler(fake, 0) == (2e-4, 3e-4)
    assert abs(SmoothnessBarrierEnergy.coeff_scheduler(fake, 1200)[0] / 2e-4 - 16) < 1e-12


# ---- randomised plans: any mesh, any tiling options -> the plan replays to the oracle or the build fails loudly ----
from hypothesis import HealthCheck, given, settings, strategies as st  # noqa: E402


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(kind=st.sampled_from(["kuhn2", "kuhn4", "kuhn6", "delaunay150", "delaunay500", "cone"]),
       spheres=st.integers(1, 3),
       lds=st.sampled_from([0, 12000, 24000, 40960, 65536, 81920, 120000, 163840]),
       threads=st.sampled_from([0, 64, 128, 256, 512, 640, 768]),
       rebuild=st.booleans(),
       spt=st.sampled_from([0, 0, 2, 3, 4]),
       target=st.sampled_from([0, 50, 300, 1000]),
       debug=st.integers(0, 3),
       lanes=st.sampled_from([0, 0, -1, 1, 4]),
       seed=st.integers(0, 3))
def test_random_plans_replay_to_oracle(kind, spheres, lds, threads, rebuild, spt, target, debug, lanes, seed):
    from tssplat_amd import tet_spheres_ext as ext
    sc = scenes.make_scene(kind, spheres, seed=seed)
    if spt in (3, 4):
        rebuild = False                                   # (the fat-wave kernels stream Dm^-1)
    if spt == 4 and threads == 0:
        pass
    kw = dict(lds_budget_bytes=lds, max_threads=threads, rebuild_dminv=rebuild,
              target_owned=target, debug_flags=debug, slots_per_thread=spt, lane_search_sweeps=lanes)
    try:
        ts = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True, **kw)
    except RuntimeError as e:      # an infeasible budget must say so, not produce a broken plan
        assert re.search(r"LDS budget|cannot be tiled|exceeds|too (long|many)", str(e)), str(e)
        return
    info = ts.plan_info()
    assert info["lds_bytes"] <= (lds or 81920) and info["block_threads"] <= (threads or 768)
    assert info["slots_per_thread"] == (spt or 2) and info["slots_per_thread"] * info["block_threads"] >= info["max_slots"]
    cache = O.prepare(sc.rest, sc.tets, round_fp32=not rebuild)   # (rebuild plans replay the unrounded operator)
    x = scenes.deform(sc, 0.2, seed=seed + 7)
    E, Es, Eb, g = O.energy_and_grad(x, cache, 3e-5, 2e-4, 4, grad_output=0.7)
    E2, Es2, Eb2, g2 = TE.emulate(ts, x, 3e-5, 2e-4, 4, grad_output=0.7)
    assert abs(E - E2) <= 1e-11 * abs(E) and np.abs(g - g2).max() <= 1e-10 * np.abs(g).max()


def test_cpp_autograd_extension_builds_and_binds():
    """csrc/torch_autograd.cpp: the in-tree host extension compiles against this torch, imports without a GPU and takes the
    entry-point addresses of the loaded library; with the nodes unset it refuses to evaluate instead of crashing."""
    import os
    import torch
    from tssplat_amd import _build
    path = _build.build_torch_ext()
    assert os.path.basename(path) == "_tsamd_autograd.so" and os.path.exists(path)
    ext = _capi.autograd_ext()
    assert ext is not None and {"energy_eval", "energy_replay", "set_entry_points"} <= set(dir(ext))
    x = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match="float32 GPU tensor"):
        ext.energy_eval(x, 0, 1.0, 1.0, 2)                    # CPU tensor: rejected before the library is called


def test_small_batches_get_smaller_tiles():
    """Default options, at most 1 024 tiles: the plan is re-tiled at 768 owned tets per tile (a small batch pays for a tile's latency,
    not for its halo): 3 k-tet spheres in four tiles instead of three.  Any explicit tiling option, or a large batch, keeps the
    fullest tiles that fit; both tilings replay to the oracle."""
    from tssplat_amd import tet_spheres_ext as ext
    sc = scenes.make_scene("kuhn8", 4)
    small = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True)
    full = ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True, max_threads=768)
    assert small.plan_info()["n_tiles"] == 16 and full.plan_info()["n_tiles"] == 12
    assert small.plan_info()["max_slots"] < full.plan_info()["max_slots"]
    cache = O.prepare(sc.rest, sc.tets)
    x = scenes.deform(sc, 0.2, seed=3)
    E, Es, Eb, g = O.energy_and_grad(x, cache, 3e-5, 2e-4, 2)
    for ts in (small, full):
        E2, _, _, g2 = TE.emulate(ts, x, 3e-5, 2e-4, 2)
        assert abs(E - E2) <= 1e-11 * abs(E) and np.abs(g - g2).max() <= 1e-10 * np.abs(g).max()
    big = scenes.make_scene("kuhn8", 400)                      # 1 200 tiles by default: left alone
    assert ext.TetSpheres(big.rest.reshape(-1), big.tets.reshape(-1), host_only=True).plan_info()["n_tiles"] == 1200


def _column_overflow(T, spt):
    """sum((d - 4)^2 over the columns a 16-lane read group meets d > 4 times) of one tile, pass 2 + pass 3 -- what
    conflict_opt.cpp's lane search minimises, recomputed from the planes alone."""
    sp, nq, rb = T["s_pad"], T["s_pad"] // spt, T["rec_base"]
    pl = T["planes"]
    slot = np.arange(sp)
    item = (slot % spt) * nq + slot // spt
    nb = np.empty((sp, 4), np.int64)
    nb[item] = (np.stack([pl[2] & 0xffff, pl[2] >> 16, pl[3] & 0xffff, pl[3] >> 16], axis=1).astype(np.int64) - rb // 4) // 12
    lane_group = np.zeros(64, np.int64)
    for g, lanes in enumerate(([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31])):
        lane_group[lanes] = g
        lane_group[[l + 32 for l in lanes]] = g + 2
    items = np.arange(T["n_slots"])
    grp = ((items // nq) * (-(-nq // 64)) + (items % nq) // 64) * 4 + lane_group[(items % nq) % 64]
    total = 0
    for lim in (T["n_owned"], T["n_slots"]):
        h = np.zeros((grp.max() + 1, 16), np.int64)
        np.add.at(h, (np.repeat(grp[:lim], 4), (nb[:lim] % 16).ravel()), 1)
        total += int((np.clip(h - 4, 0, None) ** 2).sum())
    return total


def test_lane_search_reseats_tets_within_their_tiles_and_lowers_column_overflow():
    """`lane_search_sweeps`: every tile keeps its owned and its halo tets (only their lanes change), both plans replay to the oracle, and
    the read-column overflow the search minimises drops by more than a third on a lattice and on an unstructured mesh."""
    from tssplat_amd import tet_spheres_ext as ext
    for kind, n in (("kuhn8", 3), ("delaunay1500", 2)):
        sc = scenes.make_scene(kind, n)
        plans = [ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True, lane_search_sweeps=k) for k in (-1, 0)]
        spt = plans[0].plan_info()["slots_per_thread"]
        over = [0, 0]
        for Ta, Tb in zip(TE.plan_tiles(plans[0]), TE.plan_tiles(plans[1])):
            assert (Ta["n_slots"], Ta["n_owned"], Ta["s_pad"], Ta["n_verts"]) == (Tb["n_slots"], Tb["n_owned"], Tb["s_pad"], Tb["n_verts"])
            nq = Ta["s_pad"] // spt
            slot = np.arange(Ta["s_pad"])
            item = (slot % spt) * nq + slot // spt
            for lo, hi in ((0, Ta["n_owned"]), (Ta["n_owned"], Ta["n_slots"])):
                sel = (item >= lo) & (item < hi)
                assert np.array_equal(np.sort(Ta["slot_tet"][sel]), np.sort(Tb["slot_tet"][sel]))
            assert np.array_equal(Ta["gvid"], Tb["gvid"])
            over[0] += _column_overflow(Ta, spt)
            over[1] += _column_overflow(Tb, spt)
        assert over[1] < 0.66 * over[0], (kind, over)
        cache = O.prepare(sc.rest, sc.tets)
        x = scenes.deform(sc, 0.2, seed=5)
        E, _, _, g = O.energy_and_grad(x, cache, 3e-5, 2e-4, 2)
        for ts in plans:
            E2, _, _, g2 = TE.emulate(ts, x, 3e-5, 2e-4, 2)
            assert abs(E - E2) <= 1e-11 * abs(E) and np.abs(g - g2).max() <= 1e-10 * np.abs(g).max()


def test_plan_does_not_depend_on_the_number_of_host_threads():
    """The planner works tile by tile with deterministic searches: one host thread and eight build the same bytes."""
    from tssplat_amd import tet_spheres_ext as ext
    sc = scenes.make_scene("delaunay1500", 4)
    a, b = (ext.TetSpheres(sc.rest.reshape(-1), sc.tets.reshape(-1), host_only=True, num_threads=k) for k in (1, 8))
    n = 0
    for Ta, Tb in zip(TE.plan_tiles(a), TE.plan_tiles(b)):
        for k in ("planes", "gvid", "vdst", "slot_tet", "row_start"):
            assert np.array_equal(Ta[k], Tb[k]), k
        n += 1
    assert n == a.plan_info()["n_tiles"] == b.plan_info()["n_tiles"] > 8
