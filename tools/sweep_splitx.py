"""x passes with re/im-split (default) or whole-complex LDS exchanges on small / 2D grids: python tools/sweep_splitx.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
def run(M, N, label):
    xb = torch.randn(N, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
    for rnd in range(2):
        for sx in (1, 0):
            for ss in ((1, 0) if M.ndim == 3 else (1,)):
                M.set_tuning(split_x=sx, split_s=ss)
                lsfc.time_apply(M, xb, yb, 20)
                us = min(lsfc.time_apply(M, xb, yb, 200) / 200 for _ in range(3)) * 1e3
                st = lsfc.profile_apply(M, xb, yb, 20)
                print(f"{label} split_x={sx} split_s={ss}: {us:7.1f} us/apply | " + " ".join(f"{s}={t*1e3:.1f}" for s, t, _ in st), flush=True)
for n in (1024, 512, 128):
    h = 1.0 / (n - 1); x = -0.5 + h * np.arange(n)
    M = lsfc.buildFastConvolution(x, x, h, 1.0 / h, lambda X, Y: 0.3 * np.exp(-40 * (X ** 2 + Y ** 2)), quadRule="Greengard_Vico"); M.ndim = 2
    run(M, n * n, f"2D n={n}")
for n in (48, 96, 128):
    h = 1.0 / n; x = -0.5 + h * np.arange(n)
    nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu); M.ndim = 3
    run(M, n ** 3, f"3D n={n}")
