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
    monkeypatch.setenv("LSFC_DIST_COMPUTE_STREAMS", "2" if split_edges == "1" else "1")   # odd chunks on a second compute stream (opt-in form)
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


# ---------------------------------------------------------------------------------------------------------------
# single-process multi-device plan (lsfc_plan_create_gv3d_multi): the host is ONE process, as the reference's
# (examples/example3D.jl:54,78).  On a one-GPU box the device is listed P times: P logical ranks with their own
# streams, events and buffers, slab exchanges by source-issued device copies -- the same enqueue order, event graph
# and kernels as with P GPUs.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ndev,overlap", [(1, "1"), (2, "1"), (4, "1"), (4, "0"), (8, "1")])
def test_multi_device_plan_matches_oracle(lsfc, monkeypatch, ndev, overlap):
    from fast_solver_lippmann_schwinger_amd.distributed import MultiDeviceFastM3D
    if ndev in (2, 8):
        monkeypatch.setenv("LSFC_DIST_COMPUTE_STREAMS", "2")    # odd chunks on a second compute stream per rank (opt-in form)
    monkeypatch.setenv("LSFC_DIST_OVERLAP", overlap)
    c = cases.case_3d("gv32k10")
    Mo, b, n = c["M"], c["b"], c["n"]
    M = MultiDeviceFastM3D(n, c["h"], c["k"], Mo.nu, devices=[0] * ndev)
    assert M.N == n ** 3 and M.local_n == n ** 3 // ndev and M.pipeline == "pruned-hip"
    assert ("copies" in M.transport) == (ndev > 1)
    ref = o.mul(Mo, b)
    for _ in range(3):                                    # repeated applies re-use the exchange buffers and events
        assert rel_err(M * b, ref) < TOL
    assert rel_err(lsfc.FFTconvolution(M, b), o.fft_convolution(Mo, b)) < TOL
    y = np.empty_like(b)
    M.mul_(y, b)
    assert rel_err(y, ref) < TOL
    st = dict((s, ms) for s, ms, _ in lsfc.profile_apply(M, None, None, reps=1))
    assert set(st) == {"xfwd", "alltoall_in", "yfwd", "zfused", "yinv", "alltoall_back", "xinv"}
    M.close()


def test_multi_device_plan_device_resident_slabs_and_split_edges(lsfc, monkeypatch):
    import torch
    from fast_solver_lippmann_schwinger_amd.distributed import MultiDeviceFastM3D
    n, m, l, k, P = 64, 16, 32, 9.0, 4
    h = 1.0 / n
    rng = np.random.default_rng(11)
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    x = -0.5 + h * np.arange(n)
    M1 = lsfc.buildFastConvolution3D(x, x[:m], x[:l], None, None, None, h, k, nu)
    ref = M1 * b
    for edges in ("1", "0"):
        monkeypatch.setenv("LSFC_DIST_SPLIT_EDGES", edges)
        M = MultiDeviceFastM3D(n, h, k, nu, devices=[0] * P, m=m, l=l)
        ln = M.local_n
        xs = [torch.from_numpy(b[r * ln:(r + 1) * ln]).cuda() for r in range(P)]
        ys = [torch.empty_like(v) for v in xs]
        for _ in range(4):
            M.apply_dev(xs, ys)
        M.synchronize()
        got = np.concatenate([v.cpu().numpy() for v in ys])
        assert rel_err(got, ref) < 1e-13, edges
        M.apply_dev(xs, xs)                               # y may alias x, slab by slab
        M.synchronize()
        assert rel_err(np.concatenate([v.cpu().numpy() for v in xs]), ref) < 1e-13
        M.close()


@pytest.mark.parametrize("orth", ["ModifiedGramSchmidt", "ClassicalGramSchmidt", "DGKS"])
def test_multi_device_gmres_matches_single_gpu_and_oracle(lsfc, orth):
    # the Krylov basis spread over 4 ranks, inner products completed across them; host preconditioner on the whole vector
    from fast_solver_lippmann_schwinger_amd.distributed import MultiDeviceFastM3D
    c = cases.case_3d("gv16k10")
    Mo, n = c["M"], c["n"]
    M = MultiDeviceFastM3D(n, c["h"], c["k"], Mo.nu, devices=[0, 0, 0, 0])
    M1 = lsfc.FastM3D(Mo.GFFT, Mo.nu, Mo.ne, Mo.me, Mo.le, n, n, n, Mo.omega)
    u_inc = cases.plane_wave(c["k"], c["X"])
    rhs = -(M * u_inc - u_inc)
    d = 1.0 + Mo.omega ** 2 * 0.01 * Mo.nu
    calls = []

    def Pl(v):
        assert v.size == n ** 3                            # the WHOLE vector, as ldiv!(Pl, v) sees it
        calls.append(1)
        v /= d

    for pl in (None, Pl):
        u = np.zeros(M.N, complex)
        u, hist = lsfc.gmres_(u, M, rhs, Pl=pl, restart=5, reltol=1e-10, log=True, orth_meth=orth)
        u1 = np.zeros(M.N, complex)
        u1, h1 = lsfc.gmres_(u1, M1, rhs, Pl=pl, restart=5, reltol=1e-10, log=True, orth_meth=orth)
        assert hist.isconverged and hist.iters == h1.iters and hist.mvps == h1.mvps
        assert np.max(np.abs(hist["resnorm"] - h1["resnorm"]) / h1["resnorm"]) < 1e-6
        assert rel_err(u, u1) < 1e-9
        G2 = o.reduce_symbol(Mo.GFFT, (n, n, n))
        A = lambda v: o.apply_reduced(G2, Mo.nu, Mo.omega, v, (n, n, n))
        if pl is None:
            assert np.linalg.norm(A(u) - rhs) / np.linalg.norm(rhs) < 2e-10
    assert len(calls) > 0
    M.close()


def test_multi_device_plan_argument_errors(lsfc):
    from fast_solver_lippmann_schwinger_amd.distributed import MultiDeviceFastM3D
    nu = np.zeros(32 ** 3)
    with pytest.raises(lsfc.LsfcError):
        MultiDeviceFastM3D(32, 1 / 32, 5.0, nu, devices=[0, 0, 0])          # 3 ranks: not a power of two dividing l
    with pytest.raises(lsfc.LsfcError):
        MultiDeviceFastM3D(32, 1 / 32, 5.0, nu, devices=[0, 99])            # no such device


def test_bench_single_process_rehearsal(lsfc):
    # bench.py --single-process with two logical ranks on device 0 (peer-copy transport): the N > 1 line carries the
    # exchange object and says that it is a rehearsal
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--single-process", "--devices", "0,0", "--n", "64",
                        "--steps", "3", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and "REHEARSAL" in line["config"]["parallelism"]
    assert set(line["exchange"]["unoverlapped_ms"]) == {"alltoall_in", "alltoall_back"} and "copies" in line["exchange"]["transport"]
    assert line["value"] > 0 and line["roofline"]["kernel"] in ("zfused", "yfwd", "yinv", "xfwd", "xinv")
