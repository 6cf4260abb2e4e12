#!/usr/bin/env python3
"""examples/example.jl of the reference, on the MI355X operator: 2D Lippmann-Schwinger scattering, Greengard-Vico
quadrature (or the Duan-Rokhlin trapezoidal rule), GMRES without and with a left preconditioner callback.

    python examples/example.py [h]            (reference: h = 0.005, k = 1/h, n = 201 -- examples/example.jl:30-40)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as ls


def nu(x, y):                                       # examples/example.jl:48
    return 0.3 * np.exp(-40 * (x**2 + y**2)) * (np.abs(x) < 0.48) * (np.abs(y) < 0.48)


def main(h=0.005, quadRule="Greengard_Vico"):
    k = 1.0 / h
    a = 1.0
    n = int(round(a / h)) + 1
    x = -a / 2 + h * np.arange(n)                   # collect(-a/2:h:a/2)
    y = x.copy()
    m = n
    X = np.repeat(x[:, None], m, axis=1).reshape(-1, order="F")                      # :39
    Y = np.repeat(y[None, :], n, axis=0).reshape(-1, order="F")                      # :40
    fastconv = ls.buildFastConvolution(x, y, h, k, nu, quadRule=quadRule)            # :54
    print(f"n = {n}, N = {n * m}, pipeline {fastconv.pipeline}, padded grid {fastconv.padded_dims[:2]}")
    # wrapper for linear maps (:56-61): apply_conv!(x) = fastconvolution(fastconv, x); LinearMap(apply_conv!, N)
    from scipy.sparse.linalg import LinearOperator
    convolution_map = LinearOperator((n * m, n * m), matvec=lambda v: ls.fastconvolution(fastconv, v), dtype=np.complex128)
    u_inc = np.exp(1j * k * X)                      # :76
    rhs = -k**2 * ls.FFTconvolution(fastconv, nu(X, Y) * u_inc)                      # :77
    u = np.zeros(n * m, dtype=np.complex128)
    t0 = time.time()
    u, info = ls.gmres_(u, fastconv, rhs, log=True)                                  # :91
    print(f"gmres: {info.iters} iterations, converged={info.isconverged}, {time.time() - t0:.3f} s")
    print(info["resnorm"])
    print(info["resnorm"].shape)
    # a (trivial) left preconditioner through the same in-place ldiv! boundary the reference uses (:85, Pl=precond)
    d = 1.0 + k**2 * 1e-4 * nu(X, Y)

    def precond(v):
        v /= d
    u2 = np.zeros(n * m, dtype=np.complex128)
    u2, info2 = ls.gmres_(u2, fastconv, rhs, Pl=precond, log=True)
    print(f"gmres with Pl: {info2.iters} iterations; |u - u2|/|u| = {np.linalg.norm(u - u2) / np.linalg.norm(u):.2e}")
    # precond = SparsifyingPreconditioner(Msp, As); gmres!(u, fastconv, rhs, Pl=precond) (:80-86): the same object with its
    # apply on the device.  The reference assembles (Msp, As) in src/SparsifyingMatrix2D.jl, which is outside this
    # package: a stand-in pair of the same structure is used here -- As = -(Lap_h + (1 + 0.2i) k^2) sparsifies
    # I + k^2 G nu into Msp = As + k^2 diag(nu), because (Lap + k^2) G = -delta.
    import scipy.sparse as sp
    e = np.ones(n)
    T = sp.diags([e[:-1], -2 * e, e[:-1]], [-1, 0, 1]) / h**2
    lap = sp.kron(sp.identity(n), T) + sp.kron(T, sp.identity(n))
    As = -(lap + (1 + 0.2j) * k**2 * sp.identity(n * m))          # complex-shifted: a well-conditioned stand-in
    Msp = As + k**2 * sp.diags(nu(X, Y))
    precond = ls.SparsifyingPreconditioner(Msp, As)
    u3 = np.zeros(n * m, dtype=np.complex128)
    u3, info3 = ls.gmres_(u3, fastconv, rhs, Pl=precond, maxiter=60, log=True)
    print(f"gmres with the sparsifying stand-in applied on the device (it has the structure of the reference's pair, not its\n"
          f"quality: a 5-point Laplacian on a Dirichlet box): {info3.iters} iterations, {precond.stats()}")
    # two incident directions (tests/plasma_example.jl:160-176 solves them one after the other): here in lock step, one
    # batched operator application per Arnoldi step
    U_inc = np.stack([u_inc, np.exp(1j * k * Y)])
    RHS = -k**2 * ls.apply_batch(fastconv, nu(X, Y) * U_inc, 1)
    UU = np.zeros_like(RHS)
    t0 = time.time()
    UU, infos = ls.gmres_batch_(UU, fastconv, RHS, log=True)
    print(f"two incident directions in one batch: {[i.iters for i in infos]} iterations, {time.time() - t0:.3f} s; "
          f"|u_batch - u|/|u| = {np.linalg.norm(UU[0] - u) / np.linalg.norm(u):.2e}; "
          f"LinearMap check |L*u - M*u| = {np.linalg.norm(convolution_map @ u - fastconv * u):.1e}")
    return (u + u_inc).reshape((n, m), order="F"), info     # :98, total field


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 0.005)
