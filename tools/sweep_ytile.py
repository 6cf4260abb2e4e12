"""Block-order tile (x'-groups x z planes) of the y passes and their split / whole-complex exchanges at 3D n (default 512):
python tools/sweep_ytile.py [n]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = 1.0 / n; x = -0.5 + h * np.arange(n)
nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
xb = torch.randn(n ** 3, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
for ss in (1, 0):
    for g, z in [(0, 0), (8, 8), (16, 8), (32, 4), (32, 8), (32, 16), (64, 4), (64, 8), (128, 1), (128, 2), (128, 8), (16, 32), (8, 64)]:
        M.set_tuning(split_s=ss, ytile_g=g, ytile_z=z)
        lsfc.time_apply(M, xb, yb, 2)
        st = lsfc.profile_apply(M, xb, yb, 5)
        d = {s: t for s, t, _ in st}
        print(f"n={n} split_s={ss} ytile={g:3d}x{z:2d}: yfwd={d['yfwd']:.3f} yinv={d['yinv']:.3f} sum={d['yfwd'] + d['yinv']:.3f}", flush=True)
