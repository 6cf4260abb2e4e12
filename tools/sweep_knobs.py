"""Developer harness: sweep of the fused-pass knobs (split exchange, symbol prefetch, LDS twiddle table) and of the
split exchanges of the x / y passes.  usage: python tools/sweep_knobs.py [2d] n [n ...]"""
import itertools
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402

BASE = dict(split_x=1, split_s=1, split_z=-1, sym_prefetch=-1, ytile_g=0, ytile_z=0, z_half=-1, tw_lds=1)
DIM = 2 if "2d" in sys.argv[1:] else 3
for n in [int(a) for a in sys.argv[1:] if a != "2d"]:
    h = 1.0 / n; x = -0.5 + h * np.arange(n); N = n ** DIM
    nu = np.random.default_rng(0).uniform(-0.3, 0.3, N)
    xb = torch.randn(N, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
    if DIM == 3:
        M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
    else:
        M = lsfc.buildFastConvolution(x, x, h, 1.0 / h, lambda X, Y: nu, quadRule="Greengard_Vico")
    for sz, pf, tw in itertools.product([1, 0], [0, 1], [1, 0]):
        kn = dict(BASE); kn.update(split_z=sz, sym_prefetch=pf, tw_lds=tw)
        try:
            M.set_tuning(**kn)
            lsfc.time_apply(M, xb, yb, 2)
            st = lsfc.profile_apply(M, xb, yb, 20 if DIM == 2 else 5)
            tot = lsfc.time_apply(M, xb, yb, 50) / 50
            st = st + [("apply", tot, 0)]
            print(f"n={n} L={M.padded_dims[DIM - 1]} split_z={sz} prefetch={pf} tw_lds={tw}: " + " ".join(f"{s}={t:.3f}" for s, t, b in st), flush=True)
        except Exception as e:
            print(f"n={n} split_z={sz} prefetch={pf} tw_lds={tw}: FAILED {str(e)[:80]}", flush=True)
    for ss in [0]:
        kn = dict(BASE); kn.update(split_s=ss, split_x=ss)
        M.set_tuning(**kn)
        lsfc.time_apply(M, xb, yb, 2)
        st = lsfc.profile_apply(M, xb, yb, 5)
        print(f"n={n} split_s=split_x={ss}: " + " ".join(f"{s}={t:.3f}" for s, t, b in st), flush=True)
    M.close()
