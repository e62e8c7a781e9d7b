"""Import-path shim: the reference does ``from tet_spheres import tet_spheres_ext``
(/root/reference/energies/smooth_barrier.py:6, tssplat_ext/test_ext.py:3; package layout
tssplat_ext/CMakeLists.txt:26-43).  Here that name resolves to the MI355X implementation."""
from tssplat_amd import tet_spheres_ext  # noqa: F401
