"""tssplat_amd -- MI355X-native geometry energy of TetSphere Splatting.

Only the hot path: the per-iteration tet-sphere energy (bi-harmonic smoothness
of the deformation gradient + polynomial inversion penalty) and its gradient,
as hand-written gfx950 kernels behind the reference's operator surface.

    from tssplat_amd import tet_spheres_ext            # the native module's twin
    from tssplat_amd.energies import SmoothnessBarrierEnergy

Importing ``tet_spheres_ext`` loads (and if needed builds) libtssplat_amd.so and
raises if that is impossible; nothing here falls back to the CPU.
"""
__version__ = "0.1.0"
