"""Developer harness: per-rank kernel times of the slab-distributed apply at P ranks, measured on ONE device with
simulated ranks (same kernels, layouts, chunking and symbol slab as the RCCL path; no exchange is timed).
usage: python tools/prof_dist_sim.py [n] [P ...]"""
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402
from fast_solver_lippmann_schwinger_amd.distributed import SimulatedRanks  # noqa: E402


def run(n, P, reps=5):
    h = 1.0 / n
    rng = np.random.default_rng(0)
    lz = n // P
    # only rank 0 is timed: build just that rank's plan (the library checks rank < nranks, nothing else is shared)
    sim = SimulatedRanks.__new__(SimulatedRanks)
    import ctypes as C
    from fast_solver_lippmann_schwinger_amd import _lib as L
    nu = rng.uniform(-0.3, 0.3, n * n * lz)
    plan = C.c_void_p()
    L.check(L.load().lsfc_dist_sim_plan_create_gv3d(C.byref(plan), n, n, n, float(n * h), float(1.0 / h), nu.ctypes.data_as(C.c_void_p), 0, 0, 0, P))
    M = types.SimpleNamespace(_plan=plan, N=n * n * lz)
    xb = torch.randn(M.N, dtype=torch.complex128, device="cuda")
    yb = torch.empty_like(xb)
    lsfc.profile_apply(M, xb, yb, 2)
    st = lsfc.profile_apply(M, xb, yb, reps)
    tot = sum(t for _, t, _ in st)
    detail = " ".join(f"{s}={t:.3f}({b/(t*1e-3)/1e12:.2f})" for s, t, b in st)
    print(f"n={n} P={P} K={os.environ.get('LSFC_DIST_CHUNKS', 'auto')} compute/rank={tot:.3f} ms (x{P} = {tot*P:.2f}) | {detail}", flush=True)
    L.load().lsfc_plan_destroy(plan)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    for P in [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]:
        run(n, P)
