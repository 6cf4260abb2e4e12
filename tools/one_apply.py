"""Run a few device applies at 3D n (default 512) -- target for rocprofv3 counter passes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h = 1.0 / n; x = -0.5 + h * np.arange(n)
nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
xb = torch.randn(n ** 3, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
for _ in range(reps):
    M.mul_(yb, xb)
M.synchronize()
print("done", M.pipeline)
