"""2D n (default 1024) apply under fused-pass knob variants.  usage: python tools/sweep_2d.py [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = 1.0 / (n - 1); x = -0.5 + h * np.arange(n)
M = lsfc.buildFastConvolution(x, x, h, 1.0 / h, lambda X, Y: 0.3 * np.exp(-40 * (X ** 2 + Y ** 2)), quadRule="Greengard_Vico")
xb = torch.randn(n * n, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
for sz in (-1, 0, 1):
    for pf in (-1, 0, 1):
        for tw in (1, 0):
            M.set_tuning(split_z=sz, sym_prefetch=pf, tw_lds=tw)
            lsfc.time_apply(M, xb, yb, 20)
            us = min(lsfc.time_apply(M, xb, yb, 200) / 200 for _ in range(3)) * 1e3
            st = lsfc.profile_apply(M, xb, yb, 20)
            print(f"n={n} split_z={sz:2d} sym_prefetch={pf:2d} tw_lds={tw}: {us:6.1f} us/apply | " + " ".join(f"{s}={t*1e3:.1f}" for s, t, _ in st), flush=True)
