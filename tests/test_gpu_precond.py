"""GPU parity of the device-resident SparsifyingPreconditioner apply (csrc/precond.hip, through the C ABI) against
the oracle's restatement of src/preconditioner.jl:132-170 (host sparse LU), and its use as Pl of the device GMRES.
Tolerance 1e-10 relative l2 on the apply (observed ~1e-14: two sparse triangular solves of a well-conditioned pair)."""
import numpy as np
import pytest

from oracle import lsfc_oracle as o
import cases
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _pair(n, k=None):
    x, h = cases.grid(n, True)
    k = (0.5 / h) if k is None else k
    X, Y = o.grid2d(x, x)
    nu = o.gaussian_bump(X, Y)
    return cases.sparsifying_pair_2d(n, h, k, nu), (x, h, k, nu)


@pytest.mark.parametrize("n", [9, 33, 101])
def test_apply_matches_host_lu(lsfc, n):
    (Msp, As), _ = _pair(n)
    P = lsfc.SparsifyingPreconditioner(Msp, As)
    Po = o.SparsifyingPreconditioner(Msp, As)
    b = o.random_vector(n * n)
    ref = Po.solve(b)
    v = b.copy()
    P.ldiv_(v)                                            # host vector: staged over PCIe
    assert rel_err(v, ref) < 1e-10
    import torch
    t = torch.from_numpy(b).cuda()
    P.ldiv_(t)                                            # device vector: stays on the device
    torch.cuda.synchronize()
    assert rel_err(t.cpu().numpy(), ref) < 1e-10
    assert np.array_equal(t.cpu().numpy(), v)             # same graph, same arithmetic: bitwise equal
    t2 = torch.from_numpy(ref).cuda()                     # replay on another vector
    P.ldiv_(t2)
    torch.cuda.synchronize()
    assert rel_err(t2.cpu().numpy(), Po.solve(ref)) < 1e-10
    st = P.stats()
    assert st["levels_L"] >= 1 and st["levels_U"] >= 1 and st["launches"] <= st["levels_L"] + st["levels_U"] + 2
    assert rel_err(P.solve(b), ref) < 1e-10 and np.array_equal(b, o.random_vector(n * n))     # out of place leaves b alone


def test_general_lu_with_permutations_and_wide_levels(lsfc):
    # a NON-symmetric matrix made of 500 random 12 x 12 blocks with a few couplings between neighbouring blocks:
    # exercises the row and column permutations of the host LU and dependency levels that are hundreds of rows wide
    # (the multi-workgroup level kernel) next to narrow ones (the single-workgroup chain kernel)
    import scipy.sparse as sp
    rng = np.random.default_rng(5)
    nb, bs = 500, 12
    N = nb * bs
    blocks = [rng.standard_normal((bs, bs)) + 1j * rng.standard_normal((bs, bs)) + 6.0 * np.eye(bs) * np.exp(1j * rng.uniform(0, 6.28))
              for _ in range(nb)]
    Msp = sp.block_diag(blocks, format="lil")
    for b_ in range(0, nb - 1, 7):                                 # sparse couplings
        Msp[b_ * bs + 3, (b_ + 1) * bs + 5] = 0.3 - 0.2j
        Msp[(b_ + 1) * bs + 1, b_ * bs + 2] = -0.1 + 0.4j
    Msp = Msp.tocsc()
    As = (sp.random(N, N, density=5.0 / N, random_state=9, format="csr") * (1 + 2j) + sp.identity(N)).tocsr()
    P = lsfc.SparsifyingPreconditioner(Msp, As)
    b = o.random_vector(N)
    assert rel_err(P.solve(b), o.SparsifyingPreconditioner(Msp, As).solve(b)) < 1e-10
    st = P.stats()
    assert st["levels_L"] >= bs - 1 and st["launches"] >= 4


def test_row_scaling_through_the_c_abi(lsfc):
    # Julia's lu(Msp) (UMFPACK) factors a row-scaled matrix, (Rs .* Msp)[p, q] = L U; the C ABI takes Rs as row_scale
    import ctypes as C
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla
    from fast_solver_lippmann_schwinger_amd import _lib as L
    from fast_solver_lippmann_schwinger_amd.preconditioner import _csr_arrays
    (Msp, As), _ = _pair(21)
    N = Msp.shape[0]
    Rs = np.random.default_rng(3).uniform(0.5, 2.0, N)
    lu = spla.splu((sp.diags(Rs) @ Msp).tocsc())
    rg = np.empty(N, np.int64); rg[lu.perm_r] = np.arange(N)
    cs = np.empty(N, np.int64); cs[lu.perm_c] = np.arange(N)
    arrs = [*_csr_arrays(As), *_csr_arrays(lu.L), *_csr_arrays(lu.U), rg, cs, Rs]
    pc = C.c_void_p()
    L.check(L.load().lsfc_precond_create(C.byref(pc), N, *[a.ctypes.data_as(C.c_void_p) for a in arrs], 0))
    b = o.random_vector(N)
    v = b.copy()
    L.check(L.load().lsfc_precond_apply(pc, v.ctypes.data_as(C.c_void_p), 0))
    assert rel_err(v, o.SparsifyingPreconditioner(Msp, As).solve(b)) < 1e-10
    # malformed factors are refused, not solved: an upper entry in L, a permutation with a repeated index
    pc2 = C.c_void_p()
    args = [a.ctypes.data_as(C.c_void_p) for a in arrs]
    args_bad = list(args); args_bad[9] = np.zeros(N, np.int64).ctypes.data_as(C.c_void_p)      # row_gather all zero
    assert L.load().lsfc_precond_create(C.byref(pc2), N, *args_bad, 0) == -1
    Lt = _csr_arrays(lu.U)                                                                      # U passed as "L"
    args_bad = list(args); args_bad[3:6] = [a.ctypes.data_as(C.c_void_p) for a in Lt]
    assert L.load().lsfc_precond_create(C.byref(pc2), N, *args_bad, 0) == -1
    L.load().lsfc_precond_destroy(pc)


def test_argument_errors(lsfc):
    import scipy.sparse as sp
    (Msp, As), _ = _pair(9)
    with pytest.raises(ValueError):
        lsfc.SparsifyingPreconditioner(Msp, sp.identity(80, format="csr"))
    with pytest.raises(NameError):
        lsfc.SparsifyingPreconditioner(Msp, As, solverType="nope")
    P = lsfc.SparsifyingPreconditioner(Msp, As)
    with pytest.raises(ValueError):
        P.ldiv_(np.zeros(5, complex))


def test_gmres_with_device_preconditioner(lsfc):
    # examples/example.jl:85-91 in miniature: gmres!(u, fastconv, rhs, Pl=precond) with the preconditioner applied on
    # the device (no PCIe, no Python in the loop) against the oracle GMRES with the host LU preconditioner
    n = 33
    (Msp, As), (x, h, k, nu) = _pair(n)
    M = lsfc.buildFastConvolution(x, x, h, k, o.gaussian_bump, quadRule="Greengard_Vico")
    Mo = o.build_fast_convolution(x, x, h, k, o.gaussian_bump, quadRule="Greengard_Vico")
    X, Y = o.grid2d(x, x)
    u_inc = np.exp(1j * k * X)
    rhs = -k**2 * o.fft_convolution(Mo, nu * u_inc)
    P = lsfc.SparsifyingPreconditioner(Msp, As)
    Po = o.SparsifyingPreconditioner(Msp, As)
    uo = np.zeros(n * n, complex)
    uo, ho = o.gmres(uo, lambda v: o.fastconvolution(Mo, v), rhs, Pl=Po.solve, restart=20, reltol=1e-8, maxiter=60)
    u = np.zeros(n * n, complex)
    u, hist = lsfc.gmres_(u, M, rhs, Pl=P, restart=20, reltol=1e-8, maxiter=60, log=True)
    assert hist.isconverged == ho.isconverged and abs(hist.iters - ho.iters) <= 1
    m = min(hist.iters, ho.iters)
    ro = np.asarray(ho.resnorm[:m])
    big = ro > 1e3 * 1e-8 * ro[0]
    assert np.max(np.abs(hist["resnorm"][:m] - ro)[big] / ro[big]) < 1e-6
    assert rel_err(u, uo) < 1e-6
    # the same solve with device-resident vectors
    import torch
    ud = torch.zeros(n * n, dtype=torch.complex128, device="cuda")
    ud, hd = lsfc.gmres_(ud, M, torch.from_numpy(rhs).cuda(), Pl=P, restart=20, reltol=1e-8, maxiter=60, log=True)
    assert hd.iters == hist.iters and np.array_equal(ud.cpu().numpy(), u)
    # and it is a useful preconditioner: fewer iterations than without
    u0 = np.zeros(n * n, complex)
    u0, h0 = lsfc.gmres_(u0, M, rhs, restart=20, reltol=1e-8, maxiter=60, log=True)
    print("iterations with / without the sparsifying stand-in:", hist.iters, h0.iters)
