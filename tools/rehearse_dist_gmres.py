"""One-rank rehearsal of distributed GMRES through real RCCL communicators (see tools/rehearse_dist.sh)."""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as ls
from fast_solver_lippmann_schwinger_amd.distributed import build_distributed_3d
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
n = 64; h = 1.0 / n; k = 10.0
x = -0.5 + h * np.arange(n)
Z, Y, X = np.meshgrid(x, x, x, indexing="ij")
nu = (0.3 * np.exp(-40 * (X**2 + Y**2 + Z**2))).reshape(-1)
Md = build_distributed_3d(n, h, k, nu, 0, 1, 0)
Ms = ls.buildFastConvolution3D(x, x, x, None, None, None, h, k, nu)
u_inc = np.exp(1j * k * X.reshape(-1))
rhs = -(Ms * u_inc - u_inc)
assert np.array_equal(Md * u_inc, Ms * u_inc), "distributed apply differs from the single-GPU apply"
ud, hd = ls.gmres_(np.zeros(n**3, complex), Md, rhs, restart=10, reltol=1e-8, log=True)
us, hs = ls.gmres_(np.zeros(n**3, complex), Ms, rhs, restart=10, reltol=1e-8, log=True)
print("gmres dist", hd.iters, hd.isconverged, "single", hs.iters, hs.isconverged, "diff", np.linalg.norm(ud - us) / np.linalg.norm(us))
assert hd.isconverged and hd.iters == hs.iters and np.linalg.norm(ud - us) / np.linalg.norm(us) < 1e-12
dist.destroy_process_group()
print("rehearsal ok")
