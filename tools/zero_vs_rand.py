import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import fast_solver_lippmann_schwinger_amd as lsfc
n = 512; h = 1.0 / n; x = -0.5 + h * np.arange(n)
nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
yb = torch.empty(n ** 3, dtype=torch.complex128, device="cuda")
for name, xb in (("random operands", torch.randn(n ** 3, dtype=torch.complex128, device="cuda")), ("zero operands", torch.zeros(n ** 3, dtype=torch.complex128, device="cuda")),
                 ("random operands", torch.randn(n ** 3, dtype=torch.complex128, device="cuda"))):
    lsfc.time_apply(M, xb, yb, 3)
    ms = min(lsfc.time_apply(M, xb, yb, 10) / 10 for _ in range(3))
    st = lsfc.profile_apply(M, xb, yb, 5)
    print(name, f"apply {ms:.3f} ms |", " ".join(f"{s}={t:.3f}" for s, t, b in st), flush=True)
