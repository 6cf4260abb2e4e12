"""Back-to-back device applies at a small grid (launch-bound regime): target for `rocprofv3 --kernel-trace` (per-kernel durations and
the gaps between dependent launches) and a host-side enqueue timing.  usage: python tools/small_trace.py [n=48] [reps=300] [dim=3]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dim = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if dim == 3:
    h = 1.0 / n; x = -0.5 + h * np.arange(n)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3))
else:
    h = 1.0 / (n - 1); x = -0.5 + h * np.arange(n)
    M = lsfc.buildFastConvolution(x, x, h, 1.0 / h, lambda X, Y: 0.3 * np.exp(-40 * (X ** 2 + Y ** 2)), quadRule="Greengard_Vico")
N = M.N
xb = torch.randn(N, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
for _ in range(20):
    M.mul_(yb, xb)
M.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    M.mul_(yb, xb)
t1 = time.perf_counter()
M.synchronize()
t2 = time.perf_counter()
ms = lsfc.time_apply(M, xb, yb, reps)
print(f"n={n} dim={dim} pipeline={M.pipeline} pads={M.padded_dims}: python enqueue {1e6*(t1-t0)/reps:.1f} us/apply, to completion {1e6*(t2-t0)/reps:.1f} us/apply; "
      f"lsfc_time_apply (C loop, HIP events) {1e3*ms/reps:.1f} us/apply")
