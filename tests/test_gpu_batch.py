"""GPU tests of the multi-right-hand-side path: lsfc_apply_batch as ONE pass of the pipeline per
group of up to 8 vectors (symbol tile loaded once per group in the fused pass) and lsfc_gmres_batch (several incident
directions solved in lock step, tests/plasma_example.jl:160-176).  Every column is compared with the CPU ORACLE on that
column, not with single applies of the same library."""
import numpy as np
import pytest

from oracle import lsfc_oracle as o
import cases
from cases import TOL
from conftest import rel_err

pytestmark = pytest.mark.gpu


def _columns(N, nrhs, seed=3):
    rng = np.random.default_rng(seed)
    return rng.standard_normal((nrhs, N)) + 1j * rng.standard_normal((nrhs, N))


@pytest.mark.parametrize("name,nrhs", [("gv16k10", 1), ("gv16k10", 2), ("gv32k10", 3), ("gv32k10", 8), ("gv32", 11)])
def test_apply_batch_3d_vs_oracle_per_column(lsfc, name, nrhs):
    c = cases.case_3d(name)
    Mo, n = c["M"], c["n"]
    M = lsfc.buildFastConvolution3D(c["x"], c["x"], c["x"], c["X"], c["Y"], c["Z"], c["h"], c["k"], c["nu"])
    B = _columns(n ** 3, nrhs)
    Y = lsfc.apply_batch(M, B, 0)
    Cv = lsfc.apply_batch(M, B, 1)
    Cn = lsfc.apply_batch(M, B, 2)
    for j in range(nrhs):
        assert rel_err(Y[j], o.mul(Mo, B[j])) < TOL, j
        assert rel_err(Cv[j], o.fft_convolution(Mo, B[j])) < TOL, j
        assert rel_err(Cn[j], o.fft_convolution(Mo, Mo.nu * B[j])) < TOL, j
    M.close()


@pytest.mark.parametrize("name", ["gv33", "gv32", "trap21"])
def test_apply_batch_2d_vs_oracle_per_column(lsfc, name):
    c = cases.case_2d(name)
    Mo = c["M"]
    M = lsfc.buildFastConvolution(c["x"], c["x"], c["h"], c["k"], c["nu"], quadRule=c["quadRule"])
    B = _columns(c["n"] ** 2, 5)
    Y = lsfc.apply_batch(M, B, 0)
    for j in range(5):
        assert rel_err(Y[j], o.fastconvolution(Mo, B[j])) < TOL, j
    M.close()


@pytest.mark.parametrize("dims", [(24, 20, 18), (48, 16, 30), (9, 9, 9)])
def test_apply_batch_mixed_radix_and_rocfft_fallback(lsfc, dims):
    # mixed-radix working grids (48 x 48 x 48, 96 x 32 x 64) and the rocFFT pipeline (9^3, no fused batch: member by member)
    n, m, l = dims
    rng = np.random.default_rng(sum(dims))
    G = rng.standard_normal((4 * n, 4 * m, 4 * l)) + 1j * rng.standard_normal((4 * n, 4 * m, 4 * l))
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    G2 = o.reduce_symbol(G, dims)
    M = lsfc.FastM3D(G, nu, 4 * n, 4 * m, 4 * l, n, m, l, 5.0)
    B = _columns(n * m * l, 4)
    Y = lsfc.apply_batch(M, B, 0)
    for j in range(4):
        assert rel_err(Y[j], o.apply_reduced(G2, nu, 5.0, B[j], dims)) < TOL, j
    M.close()


def test_apply_batch_n128_device_resident_and_interleaved_with_single_applies(lsfc):
    # 256-point lines in the tiled layout; device vectors; the work arrays grow for the batch and keep serving single applies
    import torch
    n = 128
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    k = 20.0
    g = np.exp(-40 * x ** 2)
    nu = (0.3 * g[:, None, None] * g[None, :, None] * g[None, None, :]).reshape(-1)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, k, nu)
    M.set_tuning(batch_fuse=1)
    G2 = o.reduced_symbol_gv3d(n, n, n, 1.0, k, patch_singular=False)
    B = _columns(n ** 3, 3)
    Bd = torch.from_numpy(B).cuda()
    y0 = M * Bd[0].contiguous()
    Yd = lsfc.apply_batch(M, Bd, 0)
    y0b = M * Bd[0].contiguous()
    M.synchronize()
    assert rel_err(y0.cpu().numpy(), y0b.cpu().numpy()) < 1e-15
    for j in range(3):
        assert rel_err(Yd[j].cpu().numpy(), o.apply_reduced(G2, nu, k, B[j], (n, n, n))) < TOL, j
    M.close()


def test_gmres_batch_two_incident_directions(lsfc):
    # tests/plasma_example.jl:160-176: two incident plane waves; here solved in lock step.  Each row must reproduce the
    # oracle's GMRES on that row (history, iteration count, solution), and the rows converge at different iterations.
    c = cases.case_3d("gv16k10")
    Mo, n, k = c["M"], c["n"], c["k"]
    M = lsfc.buildFastConvolution3D(c["x"], c["x"], c["x"], c["X"], c["Y"], c["Z"], c["h"], k, c["nu"])
    G2 = o.reduce_symbol(Mo.GFFT, (n, n, n))
    A = lambda v: o.apply_reduced(G2, Mo.nu, Mo.omega, v, (n, n, n))
    U_inc = np.stack([np.exp(1j * k * c["X"]), np.exp(1j * k * (0.6 * c["Y"] + 0.8 * c["Z"])), 0.01 * np.exp(1j * k * c["Z"])])
    RHS = -(lsfc.apply_batch(M, U_inc, 0) - U_inc)                   # examples/example3D.jl:71-72 for every direction
    X = np.zeros_like(RHS)
    X, hists = lsfc.gmres_batch_(X, M, RHS, restart=5, reltol=1e-9, abstol=2e-8, log=True)
    iters = []
    for j in range(3):
        uo = np.zeros(n ** 3, complex)
        uo, ho = o.gmres(uo, A, RHS[j], restart=5, reltol=1e-9, abstol=2e-8)
        assert hists[j].isconverged and abs(hists[j].iters - ho.iters) <= 1 and hists[j].mvps == ho.mvps + (hists[j].iters - ho.iters)
        kk = min(hists[j].iters, ho.iters)
        r, ro = hists[j]["resnorm"][:kk], np.array(ho.resnorm[:kk])
        big = ro > 1e3 * 1e-9 * ro[0]
        assert np.max(np.abs(r - ro)[big] / ro[big]) < 1e-6
        assert rel_err(X[j], uo) < 1e-6
        iters.append(hists[j].iters)
    assert len(set(iters)) > 1                                        # the small third right-hand side stops on abstol first
    # host preconditioner through the batched solve: same iterates as the single solve with the same callback
    d = 1.0 + Mo.omega ** 2 * 0.01 * Mo.nu

    def Pl(v):
        v /= d
    X2 = np.zeros_like(RHS)
    X2, h2 = lsfc.gmres_batch_(X2, M, RHS, Pl=Pl, restart=5, reltol=1e-9, log=True)
    for j in range(3):
        u1 = np.zeros(n ** 3, complex)
        u1, h1 = lsfc.gmres_(u1, M, RHS[j], Pl=Pl, restart=5, reltol=1e-9, log=True)
        assert h2[j].iters == h1.iters and rel_err(X2[j], u1) < 1e-10

    def bad(v):
        raise RuntimeError("boom")
    with pytest.raises(RuntimeError):
        lsfc.gmres_batch_(np.zeros_like(RHS), M, RHS, Pl=bad, restart=5)
    M.close()
