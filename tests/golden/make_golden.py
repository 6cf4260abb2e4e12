"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/lsfc_oracle.py).

The reference is Julia and cannot run in the build image, and it ships no fixtures
(parity unpinned): these vectors pin the ORACLE against regressions and
give the GPU tests inputs/outputs without recomputing the literal 4n pipeline.
Inputs are seeded (tests/cases.py); files hold outputs only.   python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import lsfc_oracle as o   # noqa: E402
import cases                          # noqa: E402


def main():
    for name in ["trap21", "gv33", "gv32", "gv128"]:
        c = cases.case_2d(name)
        M, b = c["M"], c["b"]
        X, _ = o.grid2d(c["x"], c["x"])
        pw = cases.plane_wave(c["k"], X)
        np.savez(os.path.join(HERE, f"2d_{name}.npz"),
                 apply_random=o.fastconvolution(M, b), conv_random=o.fft_convolution(M, b),
                 apply_planewave=o.fastconvolution(M, pw), k=c["k"], h=c["h"], n=c["n"])
    for name in ["gv16", "gv16k10", "gv32", "gv32k10"]:
        c = cases.case_3d(name)
        M, b = c["M"], c["b"]
        out = dict(apply_random=o.mul(M, b), k=c["k"], h=c["h"], n=c["n"])
        if c["n"] <= 16:
            out["conv_random"] = o.fft_convolution(M, b)
            out["apply_planewave"] = o.mul(M, cases.plane_wave(c["k"], c["X"]))
        np.savez(os.path.join(HERE, f"3d_{name}.npz"), **out)
    # GMRES residual history, 3D n=32, k=10, restart 10, 20 iterations, rhs from a plane wave (examples/example3D.jl:71-72)
    c = cases.case_3d("gv32k10")
    M = c["M"]
    G2 = o.reduce_symbol(M.GFFT, (32, 32, 32))
    A = lambda v: o.apply_reduced(G2, M.nu, M.omega, v, (32, 32, 32))
    u_inc = cases.plane_wave(c["k"], c["X"])
    rhs = -(A(u_inc) - u_inc)
    u = np.zeros(32 ** 3, dtype=np.complex128)
    u, hist = o.gmres(u, A, rhs, restart=10, maxiter=20, reltol=1e-12)
    np.savez(os.path.join(HERE, "3d_gv32k10_gmres.npz"), resnorm=np.array(hist.resnorm), u=u, mvps=hist.mvps)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
