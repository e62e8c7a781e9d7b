"""In-tree build of libtssplat_amd.so with hipcc for gfx950.

The shared library lands next to this file (``tssplat_amd/libtssplat_amd.so``),
is git-ignored, and travels to the GPU box with the gpurun snapshot.  There is
no JIT cache and no pip install: the driver records which in-tree .so files
the GPU tests load.
"""
from __future__ import annotations

import contextlib
import hashlib
import os
import shutil
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libtssplat_amd.so")
_OBJ = os.path.join(_HERE, "_obj")
ARCH = "gfx950"

SOURCES = ["plan.cpp", "conflict_opt.cpp", "capi.cpp", "kernels.hip", "surface.cpp", "surface_capi.cpp", "surface_kernels.hip",
           "raster_capi.cpp", "raster_kernels.hip", "aa_kernels.hip"]
HEADERS = ["plan.h", "conflict_opt.h", "kernels.h", "surface.h", "raster.h", "capi_common.h", os.path.join("..", "..", "include", "tssplat_amd.h")]

HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-pthread"]
# -fno-slp-vectorize: SLP packs the 3x3 algebra into v_pk_*_f32, which runs at the scalar-fp32 rate on
# gfx950 but costs ~400 v_mov_b32 of operand shuffling and 9 spilled VGPRs in the tile kernel.
# -amdgpu-sched-strategy=max-ilp: the launch bounds already pin the occupancy, so the scheduler may as well chase
# latency: tile kernel -0.5 % (0.4842 vs 0.4865 ms, same 79 VGPRs, no scratch; max-memory-clause and
# -amdgpu-schedule-metric-bias=0 measured +0.2 %).
DEVICE_FLAGS = [f"--offload-arch={ARCH}", "-ffp-contract=fast", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]
# The renderer kernels repeat the oracle's float32 / float64 operations one by one (bit-exact triangle ids, identical
# silhouette decisions): a multiply and an add must round separately.  The __fmul_rn / __dmul_rn family are plain operators
# in this toolchain's headers, so the only reliable switch is the flag (appended last: it overrides the one above).
SOURCE_FLAGS = {"raster_kernels.hip": ["-ffp-contract=off"], "aa_kernels.hip": ["-ffp-contract=off"]}


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the tssplat_amd extension cannot be built (no CPU fallback exists)")


def _digest() -> str:
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HOST_FLAGS + DEVICE_FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(SOURCE_FLAGS.items())]).encode())
    return h.hexdigest()


def traffic_digest() -> str:
    """Fingerprint of the sources that decide how many bytes one evaluation moves (the tile kernel and the plan layout).
    tools/summarize_prof.py stamps profiles/traffic.json with it when it condenses the rocprofv3 PMC passes; bench.py reports
    ``roofline.traffic`` only while the stamp matches the sources it runs."""
    h = hashlib.sha256()
    for name in ("kernels.hip", "plan.cpp", "plan.h"):
        with open(os.path.join(CSRC, name), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def build_variant(name: str, extra_device_flags: list[str]) -> str:
    """Experiment helper: build libtssplat_amd_<name>.so with extra compiler flags (the sources carry no experiment switches:
    source-level variants are patches of a copy, tools/lab_variants.py).  Select it at run time with TSSPLAT_AMD_LIB=<path>."""
    hipcc = _hipcc()
    out = os.path.join(_HERE, f"libtssplat_amd_{name}.so")
    os.makedirs(_OBJ, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(_OBJ, f"{os.path.splitext(src)[0]}_{name}.o")
        if src.endswith(".hip"):
            cmd = [hipcc, "-c", os.path.join(CSRC, src), "-o", obj] + HOST_FLAGS + DEVICE_FLAGS + SOURCE_FLAGS.get(src, []) + extra_device_flags
        else:
            cmd = [hipcc] + HOST_FLAGS + ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c",
                                           os.path.join(CSRC, src), "-o", obj] + \
                  [f for f in extra_device_flags if f.startswith("-D")]
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call([hipcc, "-shared", "-o", out] + objs + [f"--offload-arch={ARCH}", "-pthread"])
    return out


def _write_atomic(path: str, text: str) -> None:
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per tree (fcntl lock on a side file): the others wait and then find the stamp up to date."""
    import fcntl
    with open(os.path.join(_HERE, ".build.lock"), "w") as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile (if stale) and return the path of libtssplat_amd.so."""
    stamp = LIB + ".digest"
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == _digest():
        return LIB
    with _build_lock():
        return _build_locked(force, verbose)


def _build_locked(force: bool, verbose: bool) -> str:
    # the stamp sits NEXT TO the library (not under _obj/, which does not travel to the GPU box): a snapshot that carries an
    # up-to-date library is used as it is there instead of being rebuilt by the first import on every fresh box
    stamp = LIB + ".digest"
    digest = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == digest:
        return LIB
    hipcc = _hipcc()
    os.makedirs(_OBJ, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(_OBJ, os.path.splitext(src)[0] + ".o")
        if src.endswith(".hip"):
            cmd = [hipcc, "-c", os.path.join(CSRC, src), "-o", obj] + HOST_FLAGS + DEVICE_FLAGS + SOURCE_FLAGS.get(src, [])
        else:   # host-only translation units
            cmd = [hipcc] + HOST_FLAGS + ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-c",
                                           os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        objs.append(obj)
    # link under a private name and rename: ranks that start together on a fresh box must never map a half-written library
    tmp = f"{LIB}.{os.getpid()}.tmp"
    link = [hipcc, "-shared", "-o", tmp] + objs + [f"--offload-arch={ARCH}", "-pthread"]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link)
    os.replace(tmp, LIB)
    _write_atomic(stamp, digest)
    return LIB


TORCH_EXT = os.path.join(_HERE, "_tsamd_autograd.so")


def build_torch_ext(force: bool = False, verbose: bool = False) -> str:
    """Compile (if stale) tssplat_amd/_tsamd_autograd.so: the C++ autograd nodes of csrc/torch_autograd.cpp -- a plain host
    extension (g++, torch headers, no device code; it reaches libtssplat_amd.so through the entry-point addresses the ctypes
    loader hands it), in-tree like the library so that it travels to the GPU box with the snapshot."""
    import sysconfig
    src = os.path.join(CSRC, "torch_autograd.cpp")
    stamp = TORCH_EXT + ".digest"
    import torch
    from torch.utils import cpp_extension as ce
    h = hashlib.sha256(open(src, "rb").read())
    h.update(open(os.path.join(CSRC, "torch_exchange.cpp"), "rb").read())
    h.update(open(os.path.join(CSRC, "..", "..", "include", "tssplat_amd.h"), "rb").read())   # the entry-point types it hard-codes
    h.update(torch.__version__.encode())
    digest = h.hexdigest()
    if not force and os.path.exists(TORCH_EXT) and os.path.exists(stamp) and open(stamp).read() == digest:
        return TORCH_EXT
    with _build_lock():
        if not force and os.path.exists(TORCH_EXT) and os.path.exists(stamp) and open(stamp).read() == digest:
            return TORCH_EXT
        return _build_torch_ext_locked(src, stamp, digest, verbose)


def _build_torch_ext_locked(src: str, stamp: str, digest: str, verbose: bool) -> str:
    import sysconfig
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(_OBJ, exist_ok=True)
    cxx = os.environ.get("CXX") or shutil.which("g++") or "g++"
    libdir = ce.library_paths()[0]
    cmd = [cxx, "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_tsamd_autograd",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + ["-I/opt/rocm/include", f"-I{sysconfig.get_paths()['include']}"]
    tmp = f"{TORCH_EXT}.{os.getpid()}.tmp"
    # (torch_exchange.cpp: the energy exchange's helper thread -- c10d and HIP events through torch's own wrappers)
    cmd += [src, os.path.join(CSRC, "torch_exchange.cpp"), "-o", tmp, "-pthread", f"-L{libdir}", "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10",
            "-lc10_hip", "-ltorch_python", f"-Wl,-rpath,{libdir}"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, TORCH_EXT)
    _write_atomic(stamp, digest)
    return TORCH_EXT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch_ext(force="--force" in sys.argv, verbose=True))
