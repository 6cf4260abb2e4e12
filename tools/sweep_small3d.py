"""Small 3D cubes (default 48 64): apply time under fused-pass knob variants.  usage: python tools/sweep_small3d.py [n ...]"""
import os, sys, itertools
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
for n in ([int(a) for a in sys.argv[1:]] or [48, 64]):
    h = 1.0 / n; x = -0.5 + h * np.arange(n)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3))
    xb = torch.randn(n ** 3, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
    for tw, pf, zp, sz in itertools.product((1, 0), (-1, 0, 1), (-1, 3), (-1, 0, 1)):
        M.set_tuning(tw_lds=tw, sym_prefetch=pf, z_persist=zp, split_z=sz)
        lsfc.time_apply(M, xb, yb, 50)
        us = min(lsfc.time_apply(M, xb, yb, 300) / 300 for _ in range(3)) * 1e3
        st = lsfc.profile_apply(M, xb, yb, 50)
        print(f"n={n} tw_lds={tw} sym_prefetch={pf:2d} z_persist={zp:2d} split_z={sz:2d}: {us:6.1f} us | " + " ".join(f"{s}={t*1e3:.1f}" for s, t, _ in st), flush=True)
    M.close()
