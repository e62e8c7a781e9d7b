"""ctypes binding of include/tssplat_amd.h (libtssplat_amd.so).

The library is the product; this file is plumbing.  There is deliberately no
Python/CPU fallback: if the shared library cannot be loaded, importing the
operator surface raises.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

_lib = None


class TsamdError(RuntimeError):
    """A C-ABI call failed (the reference surfaces native failures as RuntimeError too,
    /root/reference/tssplat_ext/tet_spheres/cudaUtils.h:10-44)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"tssplat_amd error {code}: {message}")
        self.code = code


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32),
        ("device", C.c_int32),
        ("lds_budget_bytes", C.c_int32),
        ("max_threads", C.c_int32),
        ("target_owned", C.c_int32),
        ("lane_search_sweeps", C.c_int32),
        ("host_only", C.c_int32),
        ("num_threads", C.c_int32),
        ("debug_flags", C.c_int32),
        ("slots_per_thread", C.c_int32),
        ("rebuild_dminv", C.c_int32),
        ("abi_version", C.c_int32),
    ]


class PlanInfo(C.Structure):
    _fields_ = [
        ("n_vertices", C.c_int64), ("n_tets", C.c_int64), ("n_tiles", C.c_int64),
        ("n_components", C.c_int64), ("total_slots", C.c_int64), ("total_tile_vertices", C.c_int64),
        ("shared_vertex_copies", C.c_int64), ("finish_vertices", C.c_int64), ("device_bytes", C.c_int64),
        ("max_slots", C.c_int32), ("max_tile_vertices", C.c_int32), ("block_threads", C.c_int32),
        ("lds_bytes", C.c_int32), ("slots_per_thread", C.c_int32), ("n_planes", C.c_int32),
    ]

    def as_dict(self) -> dict:
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class TileView(C.Structure):
    _fields_ = [
        ("n_slots", C.c_int32), ("n_owned", C.c_int32), ("s_pad", C.c_int32), ("n_verts", C.c_int32),
        ("n_excl", C.c_int32), ("stage_off", C.c_int64), ("n_rows", C.c_int32), ("rec_base", C.c_int32),
        ("planes", C.POINTER(C.c_uint32)), ("row_start", C.POINTER(C.c_uint16)),
        ("gvid", C.POINTER(C.c_int32)), ("vdst", C.POINTER(C.c_int32)), ("slot_tet", C.POINTER(C.c_int32)), ("rest", C.POINTER(C.c_float)),
    ]


ABI_VERSION = 3          # include/tssplat_amd.h: TSAMD_ABI_VERSION

# every symbol include/tssplat_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "tsamd_last_error": (C.c_char_p, []),
    "tsamd_version": (C.c_char_p, []),
    "tsamd_abi_version": (C.c_int32, []),
    "tsamd_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "tsamd_create_with_operator": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "tsamd_create_from_veg": (C.c_int, [C.c_char_p, C.POINTER(Options), C.POINTER(C.c_void_p)]),
    "tsamd_destroy": (None, [C.c_void_p]),
    "tsamd_num_vertices": (C.c_int64, [C.c_void_p]),
    "tsamd_num_tets": (C.c_int64, [C.c_void_p]),
    "tsamd_get_plan_info": (C.c_int, [C.c_void_p, C.POINTER(PlanInfo)]),
    "tsamd_get_tile": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(TileView)]),
    "tsamd_get_finish_lists": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.POINTER(C.c_int32)),
                                         C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]),
    "tsamd_get_adjacency": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_int32))]),
    "tsamd_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "tsamd_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                 C.c_void_p]),
    "tsamd_forward_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsamd_evaluate_dev_coef": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "tsamd_graph_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "tsamd_graph_launch": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    "tsamd_graph_launch_to": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsamd_graph_destroy": (None, [C.c_void_p]),
    "tsamd_read_energy_terms": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]),
    "tsamd_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "tsamd_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "tsamd_scale": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "tsamd_adam_uniform_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float,
                                        C.c_float, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "tsamd_grad_limit_workspace_bytes": (C.c_int64, []),
    "tsamd_grad_limit": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "tsamd_train_loop_workspace_bytes": (C.c_int64, [C.c_int32]),
    "tsamd_train_loop_create": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.POINTER(C.c_void_p)]),
    "tsamd_train_loop_launch": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_float, C.c_float,
                                          C.c_float, C.c_int64, C.POINTER(C.c_float), C.c_void_p]),
    "tsamd_train_loop_destroy": (None, [C.c_void_p]),
    # renderer slice
    "tsamd_rasterize_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tsamd_pair_masks_bytes": (C.c_int64, [C.c_int64, C.c_int32, C.c_int32]),
    "tsamd_rasterize": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "tsamd_interpolate": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                    C.c_void_p, C.c_void_p]),
    "tsamd_interpolate_backward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32,
                                             C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsamd_rasterize_backward": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p]),
    "tsamd_antialias_topology_workspace_bytes": (C.c_int64, [C.c_int64]),
    "tsamd_antialias_topology": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsamd_antialias_prepared_bytes": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tsamd_antialias_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                          C.c_int32, C.c_void_p, C.c_void_p]),
    "tsamd_antialias": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "tsamd_antialias_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                                           C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    # surface glue
    "tsamd_extract_surface": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p,
                                      C.POINTER(C.c_int64)]),
    "tsamd_surface_create": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    "tsamd_surface_destroy": (None, [C.c_void_p]),
    "tsamd_surface_positions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsamd_surface_positions_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsamd_vertex_normals": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "tsamd_vertex_normals_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
}


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True):
    """Load libtssplat_amd.so (building it in-tree first if hipcc is around and it is stale/missing)."""
    global _lib
    if _lib is not None:
        return _lib
    # The host framework's HIP runtime must be in the process first: torch bundles its own
    # libamdhip64.so.7, and a second copy (from /opt/rocm) loaded ahead of it leaves one of the
    # two runtimes without devices ("no HIP device is visible").
    import torch  # noqa: F401
    path = _build.LIB
    override = os.environ.get("TSSPLAT_AMD_LIB")       # experiment builds (tssplat_amd._build.build_variant)
    if override:
        path, build_if_missing = override, False
    if build_if_missing:
        try:
            path = _build.build()
        except Exception:
            if not os.path.exists(path):
                raise
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python -m tssplat_amd._build` (needs hipcc, --offload-arch=gfx950). "
            "tssplat_amd has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.tsamd_abi_version() != ABI_VERSION:
        raise ImportError(f"{path}: ABI version {lib.tsamd_abi_version()}, these bindings are written for {ABI_VERSION} "
                          "(include/tssplat_amd.h: TSAMD_ABI_VERSION) -- rebuild the library")
    _lib = lib
    return lib


_autograd_ext = None


def autograd_ext(build_if_missing: bool = True):
    """The C++ autograd nodes (csrc/torch_autograd.cpp, tssplat_amd/_tsamd_autograd.so), wired to the entry points of the
    library loaded above -- or ``None`` when the extension cannot be built or imported (the Python autograd Functions of
    tssplat_amd.energies are then used: same results, 2-3x the host time per step).  TSSPLAT_AMD_PY_AUTOGRAD=1 forces that
    fallback."""
    global _autograd_ext
    if _autograd_ext is not None:
        return _autograd_ext or None
    if os.environ.get("TSSPLAT_AMD_PY_AUTOGRAD") == "1":
        _autograd_ext = False
        return None
    lib = load()
    try:
        if build_if_missing:
            try:
                _build.build_torch_ext()
            except Exception:
                if not os.path.exists(_build.TORCH_EXT):
                    raise
        import importlib
        ext = importlib.import_module("tssplat_amd._tsamd_autograd")
        addr = lambda fn: C.cast(fn, C.c_void_p).value
        ext.set_entry_points(addr(lib.tsamd_graph_launch), addr(lib.tsamd_forward_backward), addr(lib.tsamd_last_error))
        _autograd_ext = ext
    except Exception as exc:                           # noqa: BLE001
        import warnings
        warnings.warn(f"tssplat_amd: C++ autograd extension unavailable ({exc}); using the Python autograd Functions")
        _autograd_ext = False
    return _autograd_ext or None


def check(rc: int) -> None:
    if rc != 0:
        raise TsamdError(rc, load().tsamd_last_error().decode("utf-8", "replace"))


def make_options(**kw) -> Options:
    o = Options()
    o.struct_size = C.sizeof(Options)
    o.abi_version = ABI_VERSION
    o.device = -1
    for k, v in kw.items():
        if not hasattr(o, k):
            raise TypeError(f"unknown tsamd option {k!r}")
        setattr(o, k, int(v))
    return o
