#!/usr/bin/env python3
"""examples/example3D.jl of the reference, on the MI355X operator: 3D Lippmann-Schwinger scattering of a plane
wave by a Gaussian bump, solved with restarted GMRES (no preconditioner: the sparsifying preconditioner of the
reference is host-side sparse-direct code outside this build's scope; pass any in-place callable as Pl= to use one).

    python examples/example3D.py [n]          (reference: n = 48, h = 1/48, k = 1/h -- examples/example3D.jl:20-31)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as ls


def nu(x, y, z):                                    # examples/example3D.jl:43
    return 0.3 * np.exp(-40 * (x**2 + y**2 + z**2)) * (np.abs(x) < 0.48) * (np.abs(y) < 0.48) * (np.abs(z) < 0.48)


def main(n=48):
    h = 1.0 / n
    k = 1.0 / h
    a = 1.0
    x = -a / 2 + h * np.arange(n)                   # collect(-a/2:h:a/2-h)
    y, z = x.copy(), x.copy()
    Xg, Yg, Zg = np.meshgrid(x, y, z, indexing="ij")
    X, Y, Z = (A.reshape(-1, order="F") for A in (Xg, Yg, Zg))                       # :33-39, i fastest
    t0 = time.time()
    fastconv = ls.buildFastConvolution3D(x, y, z, X, Y, Z, h, k, nu, quadRule="Greengard_Vico")      # :54
    print(f"operator built in {time.time() - t0:.2f} s  (pipeline {fastconv.pipeline}, padded grid {fastconv.padded_dims})")
    u_inc = np.exp(1j * k * X)                      # :71
    rhs = -(fastconv * u_inc - u_inc)               # :72
    u = np.zeros(n**3, dtype=np.complex128)         # :75
    t0 = time.time()
    u, info = ls.gmres_(u, fastconv, rhs, log=True)                                  # :78
    print(f"gmres: {info.iters} iterations, converged={info.isconverged}, {time.time() - t0:.3f} s")
    print(info["resnorm"])                          # :79
    res = np.linalg.norm(fastconv * u - rhs) / np.linalg.norm(rhs)
    print(f"true relative residual {res:.3e}")
    U = (u + u_inc).reshape((n, n, n), order="F")   # :85, total field
    return U, info


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 48)
