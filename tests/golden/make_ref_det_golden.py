"""Golden vectors from the reference's OWN det / ddetA_dA / penalty kernels.  Run in the authoring container only:

    python tests/golden/make_ref_det_golden.py

oracle/ref_recipe/Makefile compiles lines 9-102 of /root/reference/tssplat_ext/tet_spheres/tet_spheres_cuda.cu for the
host (oracle/_ref/libref_det.so); this script runs them on a fixed set of 3x3 matrices and commits the outputs as
ref_det_golden.npz, so that the pin survives where neither the reference nor the built library exists.
"""
import os
import sys

sys.dont_write_bytecode = True        # (nothing is written into /root/reference, not even a bytecode cache)

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import ref_det  # noqa: E402


def main():
    ref_det.build(force=True)
    rng = np.random.default_rng(2024)
    # deformation gradients around the identity (few inversions), wild ones (many), exactly singular and exact identity
    F = np.concatenate([
        np.eye(3).reshape(1, 9) + 0.45 * rng.standard_normal((200, 9)),
        1.5 * rng.standard_normal((200, 9)),
        np.eye(3).reshape(1, 9) * np.array([[1.0], [-1.0], [0.0]]),
        np.zeros((1, 9)),
    ]).astype(np.float32)
    out = {"F": F,
           "det_f32": ref_det.det(F), "det_f64": ref_det.det(F.astype(np.float64)),
           "cof_f32": ref_det.ddetA_dA(F), "cof_f64": ref_det.ddetA_dA(F.astype(np.float64))}
    for order in (2, 3, 4):
        out[f"fwd{order}"] = ref_det.forward_det(F, order)
        out[f"bwd{order}"] = ref_det.backward_det(F, order)
    np.savez_compressed(os.path.join(HERE, "ref_det_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()}, "inverted:", int((out["det_f64"] < 0).sum()))


if __name__ == "__main__":
    main()
