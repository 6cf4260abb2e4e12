"""GPU test of the slab-distributed operator with P logical ranks on ONE device (lsfc_dist_sim_*): the same kernels
(chunk-packed x passes, symbol slabs, tiled y/z passes) as the multi-GPU path; only the two exchanges are
device-to-device copies instead of RCCL all-to-all.  Compared against the oracle and the single-GPU plan."""
import numpy as np
import pytest

from oracle import lsfc_oracle as o
import cases
from cases import TOL
from conftest import rel_err
from fast_solver_lippmann_schwinger_amd.distributed import SimulatedRanks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nranks", [1, 2, 4])
def test_simulated_ranks_match_oracle(lsfc, nranks):
    c = cases.case_3d("gv32k10")
    Mo, b, n = c["M"], c["b"], c["n"]
    S = SimulatedRanks(n, n, n, c["h"], c["k"], Mo.nu, nranks)
    assert rel_err(S.apply(b), o.mul(Mo, b)) < TOL
    assert rel_err(S.convolve(b), o.fft_convolution(Mo, b)) < TOL
    S.close()


def test_simulated_ranks_noncubic_equals_single_gpu(lsfc):
    n, m, l, k = 64, 16, 32, 9.0
    h = 1.0 / n
    rng = np.random.default_rng(11)
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    x = -0.5 + h * np.arange(n)
    M = lsfc.buildFastConvolution3D(x, x[:m], x[:l], None, None, None, h, k, nu)
    ref = M * b
    for nranks in (2, 8):
        S = SimulatedRanks(n, m, l, h, k, nu, nranks)
        assert rel_err(S.apply(b), ref) < 1e-13
        assert rel_err(S.convolve(b, apply_nu=True), (ref - b) / k**2) < 1e-11
        S.close()


@pytest.mark.parametrize("split_edges", ["1", "0"])
def test_three_stream_pipeline_event_logic_single_rank(lsfc, monkeypatch, split_edges):
    # the overlapped production pipeline (compute stream + two exchange streams + events), forced on with one rank:
    # repeated applies on device-resident vectors must stay bitwise identical to the sequential path
    import torch
    from fast_solver_lippmann_schwinger_amd.distributed import build_distributed_3d
    n = 64
    h = 1.0 / n
    rng = np.random.default_rng(3)
    nu = rng.uniform(-0.3, 0.3, n ** 3)
    b = o.random_vector(n ** 3)
    monkeypatch.setenv("LSFC_DIST_CHUNKS", "4")
    Mseq = build_distributed_3d(n, h, 12.0, nu, 0, 1, 0)
    ref = Mseq * b
    monkeypatch.setenv("LSFC_DIST_FORCE_OVERLAP", "1")
    monkeypatch.setenv("LSFC_DIST_SPLIT_EDGES", split_edges)   # z-half split of the first exchange in / last exchange back
    Mov = build_distributed_3d(n, h, 12.0, nu, 0, 1, 0)
    xb = torch.from_numpy(b).cuda()
    yb = torch.empty_like(xb)
    for _ in range(20):                               # back-to-back applies: buffer reuse across iterations
        Mov.mul_(yb, xb)
    Mov.synchronize()
    assert np.array_equal(yb.cpu().numpy(), ref)
    x2 = torch.from_numpy(ref).cuda()                 # chained applies y = M(M x)
    Mov.mul_(yb, x2)
    Mov.synchronize()
    assert np.array_equal(yb.cpu().numpy(), Mseq * ref)


@pytest.mark.parametrize("chunks", ["1", "4"])
def test_distributed_plan_single_rank_production_path(lsfc, chunks, monkeypatch):
    # the real (non-simulated) distributed plan with one rank: chunk-packed x passes, chunk loop, profile stages, GMRES
    from fast_solver_lippmann_schwinger_amd.distributed import build_distributed_3d
    monkeypatch.setenv("LSFC_DIST_CHUNKS", chunks)
    c = cases.case_3d("gv32k10")
    Mo, b, n = c["M"], c["b"], c["n"]
    M = build_distributed_3d(n, c["h"], c["k"], Mo.nu, 0, 1, 0)
    assert rel_err(M * b, o.mul(Mo, b)) < TOL
    assert rel_err(lsfc.FFTconvolution(M, b), o.fft_convolution(Mo, b)) < TOL
    import torch
    xb = torch.from_numpy(b).cuda(); yb = torch.empty_like(xb)
    names = [s[0] for s in lsfc.profile_apply(M, xb, yb, 1)]
    assert names == ["xfwd", "alltoall_in", "yfwd", "zfused", "yinv", "alltoall_back", "xinv"]
    u_inc = cases.plane_wave(c["k"], c["X"])
    rhs = -(M * u_inc - u_inc)
    u = np.zeros(M.N, complex)
    u, hist = lsfc.gmres_(u, M, rhs, restart=10, reltol=1e-8, log=True)
    assert hist.isconverged and np.linalg.norm(o.mul(Mo, u) - rhs) / np.linalg.norm(rhs) < 1e-7


@pytest.mark.parametrize("shape,ranks", [((24, 20, 18), (1, 2)), ((48, 40, 96), (2, 4)), ((80, 24, 64), (4,))])
def test_simulated_ranks_grid_sizes_not_powers_of_two(lsfc, shape, ranks):
    # slab decomposition on mixed-radix working grids (24 x 20 x 18 -> 48^3, 48 x 40 x 96 -> 96 x 80 x 192,
    # 80 x 24 x 64 -> 160 x 48 x 128): the rank count must divide l and Lx/8; chunk widths need not be powers of two
    n, m, l = shape
    k = 7.0
    h = 1.0 / n
    rng = np.random.default_rng(12)
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    x = -0.5 + h * np.arange(n)
    M = lsfc.buildFastConvolution3D(x, x[:1].repeat(m), x[:1].repeat(l), None, None, None, h, k, nu)
    ref = M * b
    for nranks in ranks:
        S = SimulatedRanks(n, m, l, h, k, nu, nranks)
        assert rel_err(S.apply(b), ref) < 1e-13
        S.close()
