"""CPU tests of the oracle: the identities that pin it (the reference has no fixtures,
SURVEY.md section 4) and the committed golden vectors."""
import os

import numpy as np
import pytest

from oracle import lsfc_oracle as o
import cases
from conftest import rel_err

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_dense_identity_2d_trapezoidal():
    # fastconvolution(M,b) == b + k^2 * buildConvMatrix * (nu .* b)   (FastConvolution.jl:58-82 vs :497-513)
    c = cases.case_2d("trap21")
    X, Y = o.grid2d(c["x"], c["x"])
    _, D = o.reference_vals_trap_rule()
    G = o.build_conv_matrix(c["k"], X, Y, D[0], c["h"])
    ref = c["b"] + c["k"] ** 2 * G @ (c["M"].nu * c["b"])
    assert rel_err(o.fastconvolution(c["M"], c["b"]), ref) < 1e-13


def test_analytic_gaussian_3d():
    # FFTconvolution(FastM3D, unit-mass Gaussian) == -solRefHelmholtz (Functions.jl:32-36), spectral accuracy at n=64
    n, k, sig = 64, 10.0, 0.05
    x, h = cases.grid(n, False)
    X, Y, Z = o.grid3d(x, x, x)
    M = o.build_fast_convolution3d(x, x, x, X, Y, Z, h, k, o.gaussian_bump)
    f = np.exp(-(X**2 + Y**2 + Z**2) / (2 * sig**2)) / ((2 * np.pi) ** 1.5 * sig**3)
    u = o.fft_convolution(M, f.astype(np.complex128))
    with np.errstate(all="ignore"):
        ref = -o.sol_ref_helmholtz(X, Y, Z, sig, k)
    ok = np.isfinite(ref)                  # the closed form is 0/0 at r = 0 (one grid point)
    assert ok.sum() == n**3 - 1
    assert rel_err(u[ok], ref[ok]) < 1e-12


def test_analytic_gaussian_2d_by_quadrature():
    # second independent pin, for the 2D Greengard-Vico branch: FFTconvolution of a unit-mass Gaussian against the
    # continuous integral of (i/4) H0(k r) f, evaluated by adaptive quadrature at three grid points (centre, off-centre,
    # far corner region).  Spectral accuracy of the truncated-kernel quadrature: ~1e-15.
    n, k, sig = 128, 20.0, 0.06
    x, h = cases.grid(n, True)
    M = o.build_fast_convolution(x, x, h, k, o.gaussian_bump, quadRule="Greengard_Vico")
    X, Y = o.grid2d(x, x)
    f = np.exp(-(X ** 2 + Y ** 2) / (2 * sig ** 2)) / (2 * np.pi * sig ** 2)
    u = o.fft_convolution(M, f.astype(complex)).reshape((n, n), order="F")
    for i, j, exact in cases.gaussian_2d_quadrature_points(x, k, sig, [(64, 64), (74, 57), (20, 90)]):
        assert abs(u[i, j] - exact) / abs(exact) < 1e-11, (i, j)


@pytest.mark.parametrize("name", ["gv33", "gv32"])
def test_padding_identities_2d(name):
    c = cases.case_2d(name)
    M, b, n = c["M"], c["b"], c["n"]
    lit = o.fastconvolution(M, b)
    # shifts folded into the symbol
    B = np.zeros((M.ne, M.me), complex); B[:n, :n] = (M.nu * b).reshape((n, n), order="F")
    pre = b + M.omega**2 * np.fft.ifft2(np.fft.fft2(B) * np.fft.ifftshift(M.GFFT))[:n, :n].reshape(-1, order="F")
    assert rel_err(pre, lit) < 1e-14
    G2 = o.reduce_symbol(M.GFFT, (n, n))
    assert rel_err(o.apply_reduced(G2, M.nu, M.omega, b, (n, n)), lit) < 1e-14


def test_padding_identity_3d_and_slabwise_symbol():
    c = cases.case_3d("gv16k10")
    M, b, n = c["M"], c["b"], c["n"]
    G2 = o.reduce_symbol(M.GFFT, (n, n, n))
    assert rel_err(o.apply_reduced(G2, M.nu, M.omega, b, (n, n, n)), o.mul(M, b)) < 1e-14
    G2s = o.reduced_symbol_gv3d(n, n, n, 1.0, c["k"], patch_singular=False, chunk=8)
    assert rel_err(G2s, G2) < 1e-13
    # the reduced symbol is even in every axis
    assert rel_err(G2[(-np.arange(2 * n)) % (2 * n)], G2) < 1e-12


def test_delta_response_is_shifted_kernel():
    # what sampleG3D relies on (FastConvolution3D.jl:136-160)
    c = cases.case_3d("gv16")
    M, n = c["M"], c["n"]
    T = np.fft.ifftn(np.fft.ifftshift(M.GFFT))
    src = (3, 5, 7)
    e = np.zeros((n, n, n), complex); e[src] = 1
    out = o.fft_convolution(M, e.reshape(-1, order="F")).reshape((n, n, n), order="F")
    i, j, k = np.meshgrid(*(np.arange(n),) * 3, indexing="ij")
    ref = T[(i - src[0]) % (4 * n), (j - src[1]) % (4 * n), (k - src[2]) % (4 * n)]
    assert rel_err(out, ref) < 1e-13
    assert rel_err(o.sample_g_conv([src[0] + n * (src[1] + n * src[2])], M)[0], ref.reshape(-1, order="F")) < 1e-13


def test_singular_lattice_points_and_patch():
    # omega = 64 pi on the half-open unit box hits s == k exactly: Inf unpatched, analytic limit patched
    L, k = 1.8, 64 * np.pi
    s = np.array([(np.pi / 2) * np.sqrt(16384.0), 3.0])
    assert s[0] == k
    with np.errstate(all="ignore"):
        assert not np.isfinite(o.gtruncated3d(L, k, s)[0])
    g = o.gtruncated3d(L, k, s, patch_singular=True)
    assert abs(g[0] - (4.27314550822e-6 + 4.47035129424e-3j)) < 1e-13
    near = o.gtruncated3d(L, k, np.array([k * (1 + 1e-9)]))
    assert abs(near[0] - g[0]) / abs(g[0]) < 1e-5
    a2 = o.gtruncated2d(1.5, 10 * np.pi, np.array([10 * np.pi]), patch_singular=True)
    assert abs(a2[0] - (-1.3427385473e-6 + 2.3621314013e-2j)) < 1e-10


def test_traits_and_errors():
    c = cases.case_2d("trap21")
    M = c["M"]
    assert o.size(M, 1) == 441 and o.size(M) == ((441,), (441,)) and o.eltype(M) == np.complex128
    M.quadRule = "nonsense"
    with pytest.raises(NameError):
        o.fastconvolution(M, c["b"])
    with pytest.raises(ValueError):
        x, h = cases.grid(20, True)
        o.build_fast_convolution(x, x, h, 1 / h, o.gaussian_bump)     # even n trapezoidal: reference errors too


def test_gmres_oracle_solves_and_history_monotone():
    c = cases.case_3d("gv16k10")
    M, n = c["M"], c["n"]
    G2 = o.reduce_symbol(M.GFFT, (n, n, n))
    A = lambda v: o.apply_reduced(G2, M.nu, M.omega, v, (n, n, n))
    u_inc = cases.plane_wave(c["k"], c["X"])
    rhs = -(A(u_inc) - u_inc)
    for orth in ["ModifiedGramSchmidt", "ClassicalGramSchmidt", "DGKS"]:
        u = np.zeros(n**3, complex)
        u, hist = o.gmres(u, A, rhs, restart=5, reltol=1e-10, orth_meth=orth)
        assert hist.isconverged
        r = np.array(hist.resnorm)
        assert np.all(np.diff(r) <= 1e-12 * r[0])
        assert np.linalg.norm(A(u) - rhs) / np.linalg.norm(rhs) < 2e-10
        # one extra mat-vec per restart (init!) on top of one per iteration
        assert hist.mvps == hist.iters + (hist.iters - 1) // 5 or hist.mvps == hist.iters + hist.iters // 5
    # diagonal left preconditioner: same solution
    d = 1.0 + M.omega**2 * 0.01 * M.nu
    u2 = np.zeros(n**3, complex)
    u2, h2 = o.gmres(u2, A, rhs, Pl=lambda v: v / d, restart=5, reltol=1e-10)
    assert h2.isconverged and rel_err(u2, u) < 1e-7


@pytest.mark.parametrize("precond", [False, True])
def test_gmres_oracle_against_an_independent_gmres(precond):
    # IterativeSolvers.jl is neither vendored nor pinned by the reference, so the restatement of gmres! is cross-checked
    # against an independent implementation of the same method: restarted GMRES iterates are unique in exact arithmetic,
    # so scipy.sparse.linalg.gmres (restart 6, left preconditioner M, one residual norm per inner step) must produce the
    # same (preconditioned) residual history -- including across restarts -- and the same solution.
    import scipy.sparse.linalg as spla
    c = cases.case_3d("gv16k10")
    Mo, n = c["M"], c["n"]
    G2 = o.reduce_symbol(Mo.GFFT, (n, n, n))
    A = lambda v: o.apply_reduced(G2, Mo.nu, Mo.omega, v, (n, n, n))
    u_inc = cases.plane_wave(c["k"], c["X"])
    rhs = -(A(u_inc) - u_inc)
    d = 1.0 + Mo.omega ** 2 * 0.01 * Mo.nu
    Pl = (lambda v: v / d) if precond else None
    u = np.zeros(n ** 3, complex)
    u, h = o.gmres(u, A, rhs, Pl=Pl, restart=6, reltol=1e-10)
    hist = []
    N = n ** 3
    L = spla.LinearOperator((N, N), matvec=A, dtype=complex)
    Mi = spla.LinearOperator((N, N), matvec=lambda v: v / d, dtype=complex) if precond else None
    us, info = spla.gmres(L, rhs, M=Mi, restart=6, rtol=1e-10, atol=0, maxiter=20, callback=lambda r: hist.append(r), callback_type="pr_norm")
    assert info == 0 and h.isconverged and len(hist) == h.iters > 6          # more than one restart cycle
    # scipy reports the preconditioned residual norm relative to a fixed norm of the right-hand side (|b| without M):
    # the two histories must be proportional, entry by entry, through every restart
    ratio = np.array(h.resnorm) / np.array(hist)
    assert np.max(np.abs(ratio / ratio[0] - 1.0)) < 1e-8
    if not precond:
        assert abs(ratio[0] / np.linalg.norm(rhs) - 1.0) < 1e-10
    assert rel_err(us, u) < 1e-12


@pytest.mark.parametrize("name", ["trap21", "gv33", "gv32", "gv128"])
def test_golden_2d(name):
    c = cases.case_2d(name)
    g = np.load(os.path.join(GOLD, f"2d_{name}.npz"))
    assert rel_err(o.fastconvolution(c["M"], c["b"]), g["apply_random"]) < 1e-13
    assert rel_err(o.fft_convolution(c["M"], c["b"]), g["conv_random"]) < 1e-13


@pytest.mark.parametrize("name", ["gv16", "gv16k10", "gv32k10"])
def test_golden_3d(name):
    c = cases.case_3d(name)
    g = np.load(os.path.join(GOLD, f"3d_{name}.npz"))
    assert rel_err(o.mul(c["M"], c["b"]), g["apply_random"]) < 1e-13


def test_sparsifying_preconditioner_restatement_is_the_dense_formula():
    # src/preconditioner.jl:132-145: P \ b == lu(Msp) \ (As * b); the oracle's sparse-LU restatement against dense algebra
    n = 9
    x, h = cases.grid(n, True)
    X, Y = o.grid2d(x, x)
    Msp, As = cases.sparsifying_pair_2d(n, h, 0.5 / h, o.gaussian_bump(X, Y))
    P = o.SparsifyingPreconditioner(Msp, As)
    b = o.random_vector(n * n)
    ref = np.linalg.solve(Msp.toarray(), As.toarray() @ b)
    assert rel_err(P.solve(b), ref) < 1e-12
    v = b.copy()
    P.ldiv_(v)                                    # two-argument in-place form, :147-170
    assert rel_err(v, ref) < 1e-12
