"""Secondary measurements for the BASELINE.json configs that are not the bench.py headline:
  configs[1]  2D n=1024 fp64 apply on one MI355X
  configs[2]  3D n=256 fp64, full GMRES(30) solve to 1e-6 (rhs from a plane wave, examples/example3D.jl:71-78)
  configs[4]  3D n=512 at omega = 64 pi (lattice points on |s| = k, patched symbol): apply rate + finite check
usage: python tools/bench_configs.py [2d] [gmres] [hf] [hfgmres] [gmres512] [host]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402


def bump(*c):
    r2 = sum(x ** 2 for x in c)
    out = 0.3 * np.exp(-40 * r2)
    for x in c:
        out = out * (np.abs(x) < 0.48)
    return out


def bench_2d(n=1024):
    h = 1.0 / (n - 1)
    x = -0.5 + h * np.arange(n)
    k = 1.0 / h
    M = lsfc.buildFastConvolution(x, x, h, k, lambda X, Y: bump(X, Y), quadRule="Greengard_Vico")
    N = n * n
    xb = torch.randn(N, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
    lsfc.time_apply(M, xb, yb, 20)
    ms = min(lsfc.time_apply(M, xb, yb, 200) / 200 for _ in range(3))
    st = lsfc.profile_apply(M, xb, yb, 20)
    return {"config": f"2D n={n} fp64 apply", "pipeline": M.pipeline, "ms_per_apply": ms, "applies_per_s": 1e3 / ms,
            "algorithmic_GBps": 248.0 * N / (ms * 1e-3) / 1e9, "stages": {s: t for s, t, _ in st}}


def bench_gmres(n=256, restart=30, reltol=1e-6):
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    k = 1.0 / h
    X = np.tile(x, n * n)                                  # x fastest
    Y = np.tile(np.repeat(x, n), n)
    Z = np.repeat(x, n * n)
    nu = bump(X, Y, Z)
    t0 = time.time()
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, k, nu)
    t_plan = time.time() - t0
    u_inc = torch.from_numpy(np.exp(1j * k * X)).cuda()
    rhs = -(M * u_inc - u_inc)
    out = {}
    # (a one-step solve first: the Krylov workspace -- (restart + 2) vectors, 69 GB at 512^3 -- is allocated on first use, and
    # that hipMalloc takes 0.5-2 s; round 2's per-iteration figures of the first solve of a process included it)
    lsfc.gmres_(torch.zeros_like(rhs), M, rhs, restart=restart, maxiter=1, reltol=reltol, log=True)
    for orth in ["ModifiedGramSchmidt", "ClassicalGramSchmidt"]:
        u = torch.zeros_like(rhs)
        torch.cuda.synchronize(); t0 = time.time()
        u, hist = lsfc.gmres_(u, M, rhs, restart=restart, reltol=reltol, log=True, orth_meth=orth)
        torch.cuda.synchronize(); t = time.time() - t0
        res = float(torch.linalg.norm(M * u - rhs) / torch.linalg.norm(rhs))
        out[orth] = {"iters": hist.iters, "mvps": hist.mvps, "converged": hist.isconverged, "seconds": t,
                     "ms_per_iteration": 1e3 * t / max(hist.iters, 1), "true_relres": res, "final_resnorm": float(hist["resnorm"][-1])}
    yb = torch.empty_like(rhs)
    ms = lsfc.time_apply(M, rhs, yb, 20) / 20
    return {"config": f"3D n={n} GMRES({restart}) reltol={reltol:g}", "plan_seconds": t_plan, "apply_ms": ms, "solves": out}


def bench_small():
    """launch-latency territory: the reference's own example sizes (2D n=128 configs[0], 3D n=48 example3D.jl:20)"""
    out = []
    for dim, n in [(2, 128), (2, 256), (3, 48), (3, 64)]:
        if dim == 2:
            h = 1.0 / (n - 1); x = -0.5 + h * np.arange(n); k = 10 * np.pi if n == 128 else 1.0 / h
            M = lsfc.buildFastConvolution(x, x, h, k, lambda X, Y: bump(X, Y), quadRule="Greengard_Vico")
            X = np.tile(x, n)
        else:
            h = 1.0 / n; x = -0.5 + h * np.arange(n); k = 1.0 / h
            X = np.tile(x, n * n); Y = np.tile(np.repeat(x, n), n); Z = np.repeat(x, n * n)
            M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, k, bump(X, Y, Z))
        u_inc = torch.from_numpy(np.exp(1j * k * X)).cuda()
        rhs = -(M * u_inc - u_inc)
        yb = torch.empty_like(rhs)
        lsfc.time_apply(M, rhs, yb, 50)
        us = 1e3 * min(lsfc.time_apply(M, rhs, yb, 500) / 500 for _ in range(3))
        r = {"config": f"{dim}D n={n}", "pipeline": M.pipeline, "apply_us": us}
        for orth in ["ModifiedGramSchmidt", "ClassicalGramSchmidt"]:
            best = None
            for _ in range(3):
                u = torch.zeros_like(rhs)
                torch.cuda.synchronize(); t0 = time.time()
                u, hist = lsfc.gmres_(u, M, rhs, restart=30, reltol=1e-10, maxiter=60, log=True, orth_meth=orth)
                torch.cuda.synchronize(); t = time.time() - t0
                best = t if best is None else min(best, t)
            r[orth] = {"iters": hist.iters, "us_per_iteration": 1e6 * best / max(hist.iters, 1)}
        out.append(r)
    return out


def bench_hf(n=512):
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    k = 64 * np.pi
    rng = np.random.default_rng(0)
    nu = rng.uniform(-0.3, 0.3, n ** 3)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, k, nu, flags=4)          # LSFC_FLAG_PATCH_SINGULAR
    xb = torch.randn(n ** 3, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
    lsfc.time_apply(M, xb, yb, 2)
    ms = lsfc.time_apply(M, xb, yb, 10) / 10
    return {"config": f"3D n={n} omega=64pi patched symbol", "ms_per_apply": ms, "finite": bool(torch.isfinite(torch.view_as_real(yb)).all())}


def bench_hf_gmres(n=512, restart=30, maxiter=30):
    """configs[4] as the reference runs it: GPU operator apply, preconditioner applied ON THE HOST through the
    two-argument in-place callback (src/preconditioner.jl:147-170) -- the Krylov vector crosses PCIe twice per inner
    step.  The sparsifying factorisation itself is out of scope (and infeasible at 512^3): a diagonal
    stand-in of the same data movement is used, so this times the boundary, not a preconditioner's quality."""
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    k = 64 * np.pi
    X = np.tile(x, n * n); Y = np.tile(np.repeat(x, n), n); Z = np.repeat(x, n * n)
    nu = bump(X, Y, Z)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, k, nu, flags=4)
    u_inc = torch.from_numpy(np.exp(1j * k * X)).cuda()
    del X, Y, Z
    rhs = -(M * u_inc - u_inc)
    dinv = 1.0 / (1.0 + 0.05 * nu)
    t_cb = [0.0, 0]

    def Pl(v):
        t0 = time.time(); v *= dinv; t_cb[0] += time.time() - t0; t_cb[1] += 1

    out = {}
    for name, kw in [("host preconditioner callback", dict(Pl=Pl)), ("no preconditioner", {})]:
        u = torch.zeros_like(rhs)
        t_cb[0], t_cb[1] = 0.0, 0
        torch.cuda.synchronize(); t0 = time.time()
        u, hist = lsfc.gmres_(u, M, rhs, restart=restart, reltol=1e-6, maxiter=maxiter, log=True, orth_meth="ClassicalGramSchmidt", **kw)
        torch.cuda.synchronize(); t = time.time() - t0
        out[name] = {"iters": hist.iters, "seconds": t, "ms_per_iteration": 1e3 * t / max(hist.iters, 1),
                     "resnorm_first_last": [float(hist["resnorm"][0]), float(hist["resnorm"][-1])],
                     "callback_calls": t_cb[1], "host_ms_inside_callback_per_call": 1e3 * t_cb[0] / max(t_cb[1], 1)}
    return {"config": f"3D n={n} omega=64pi patched symbol, GMRES({restart}) capped at {maxiter} iterations", "solves": out}


def bench_precond(n=513):
    """SparsifyingPreconditioner apply b <- Msp^-1 (As b) (src/preconditioner.jl:132-170) on the
    device against the host sparse-LU solve, 2D n x n, synthetic (Msp, As) of the reference's structure (tests/cases.py);
    then the GMRES inner step with the preconditioner on the device, and on the host through the callback."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import cases
    import scipy.sparse.linalg as spla
    h = 1.0 / (n - 1); x = -0.5 + h * np.arange(n); k = 0.5 / h
    X = np.tile(x, n); Y = np.repeat(x, n)
    nu = bump(X, Y)
    Msp, As = cases.sparsifying_pair_2d(n, h, k, nu)
    t0 = time.time(); lu = spla.splu(Msp); t_lu = time.time() - t0
    t0 = time.time(); P = lsfc.SparsifyingPreconditioner(Msp, As, lu=lu); t_up = time.time() - t0
    N = n * n
    b = np.random.default_rng(0).standard_normal(N) + 1j * np.random.default_rng(1).standard_normal(N)
    t0 = time.time()
    for _ in range(3): ref = lu.solve(As @ b)
    t_host = (time.time() - t0) / 3
    v = torch.from_numpy(b).cuda(); P.ldiv_(v); torch.cuda.synchronize()
    err = float(np.linalg.norm(v.cpu().numpy() - ref) / np.linalg.norm(ref))
    v = torch.from_numpy(b).cuda(); torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): P.ldiv_(v)
    torch.cuda.synchronize(); t_dev = (time.time() - t0) / 20
    M = lsfc.buildFastConvolution(x, x, h, k, lambda X_, Y_: nu, quadRule="Greengard_Vico")
    u_inc = np.exp(1j * k * X)
    rhs = torch.from_numpy(-k**2 * lsfc.FFTconvolution(M, nu * u_inc)).cuda()
    out = {}
    for name, Pl in [("preconditioner on the device", P), ("preconditioner on the host (callback)", lambda v_: v_.__setitem__(slice(None), lu.solve(As @ v_))), ("none", None)]:
        u = torch.zeros_like(rhs); torch.cuda.synchronize(); t0 = time.time()
        u, hist = lsfc.gmres_(u, M, rhs, Pl=Pl, restart=30, reltol=1e-6, maxiter=30, log=True, orth_meth="ClassicalGramSchmidt")
        torch.cuda.synchronize(); t = time.time() - t0
        out[name] = {"iters": hist.iters, "converged": hist.isconverged, "ms_per_iteration": 1e3 * t / max(hist.iters, 1)}
    st = P.stats()
    return {"config": f"2D n={n} SparsifyingPreconditioner apply (synthetic Msp/As pair)", "N": N, "nnz_L+U": st["nnz_L"] + st["nnz_U"],
            "levels": [st["levels_L"], st["levels_U"]], "graph_kernel_nodes": st["launches"], "host_lu_factor_s": t_lu, "upload_and_analysis_s": t_up,
            "host_apply_ms": 1e3 * t_host, "device_apply_ms": 1e3 * t_dev, "device_vs_host_rel_err": err, "gmres": out}


def bench_host_vectors(n=512):
    """PCIe-inclusive rate: the same apply with HOST-resident x and y (LSFC_MEM_HOST), as the Julia wrapper's `*` does."""
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
    b = np.random.default_rng(1).standard_normal(n ** 3) + 0j
    y = np.empty_like(b)
    M.mul_(y, b)
    t0 = time.time()
    for _ in range(3):
        M.mul_(y, b)
    t = (time.time() - t0) / 3
    return {"config": f"3D n={n} apply with host vectors (PCIe inclusive, pageable numpy memory)", "ms_per_apply": 1e3 * t, "applies_per_s": 1 / t}


if __name__ == "__main__":
    what = sys.argv[1:] or ["2d", "gmres", "hf"]
    res = []
    if "2d" in what: res.append(bench_2d())
    if "gmres" in what: res.append(bench_gmres())
    if "hf" in what: res.append(bench_hf())
    if "gmres512" in what: res.append(bench_gmres(512))
    if "small" in what: res.extend(bench_small())
    if "hfgmres" in what: res.append(bench_hf_gmres())
    if "precond" in what: res.append(bench_precond())
    if "precond257" in what: res.append(bench_precond(257))
    if "host" in what: res.append(bench_host_vectors())
    for r in res:
        print(json.dumps(r), flush=True)
