"""The drop-in boundary is a C ABI: compile a plain C99 program against include/tssplat_amd.h with gcc,
link it to libtssplat_amd.so and run its host-only calls (no GPU needed)."""
import os
import shutil
import subprocess

import pytest

from tssplat_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_plain_c_program_links_and_runs(tmp_path):
    _capi.load()                                      # builds the library if it is missing
    lib = _capi.lib_path()
    exe = tmp_path / "abi_smoke"
    libdir = os.path.dirname(lib)
    hip = "/opt/rocm/lib"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", str(exe), "-L", libdir, "-ltssplat_amd",
           f"-Wl,-rpath,{libdir}", f"-Wl,-rpath,{hip}", f"-Wl,-rpath-link,{hip}"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + hip + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, f"exit {out.returncode}: {out.stdout} {out.stderr}"
    assert "abi smoke ok" in out.stdout
