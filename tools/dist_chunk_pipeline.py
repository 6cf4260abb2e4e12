"""One rank, the chunk pipeline of the distributed apply forced on (LSFC_DIST_FORCE_OVERLAP=1): the apply at 3D n with K chunks of the
x' range, one or two compute streams (LSFC_DIST_COMPUTE_STREAMS).  With K = 32 at n = 512 every chunk has the shape a rank of an
8-GPU job transforms (32 x' x 1024 x 512 z-lines), so 1/8 of the time is that rank's compute with its launch ramps and tails
(the self exchange is a device-to-device copy on the communication streams).
usage: python tools/dist_chunk_pipeline.py [n] [K ...]"""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
from fast_solver_lippmann_schwinger_amd.distributed import build_distributed_3d

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
Ks = [int(a) for a in sys.argv[2:]] or [4, 8, 16, 32]
h = 1.0 / n
nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
xb = torch.randn(n ** 3, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
x = -0.5 + h * np.arange(n)
M0 = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
lsfc.time_apply(M0, xb, yb, 3)
ms0 = min(lsfc.time_apply(M0, xb, yb, 10) / 10 for _ in range(3))
ref = yb.clone()
print(json.dumps({"n": n, "form": "single-GPU plan (no chunks)", "ms_per_apply": round(ms0, 3)}), flush=True)
M0.close()
os.environ["LSFC_DIST_FORCE_OVERLAP"] = "1"
for K in Ks:
    for streams in (1, 2, 1, 2):
        os.environ["LSFC_DIST_CHUNKS"] = str(K)
        os.environ["LSFC_DIST_COMPUTE_STREAMS"] = str(streams)
        M = build_distributed_3d(n, h, 1.0 / h, nu, 0, 1, 0)
        lsfc.time_apply(M, xb, yb, 3)
        ms = min(lsfc.time_apply(M, xb, yb, 10) / 10 for _ in range(3))
        err = float(torch.linalg.norm(yb - ref) / torch.linalg.norm(ref))
        print(json.dumps({"n": n, "form": "one rank, chunk pipeline", "K": K, "compute_streams": streams, "ms_per_apply": round(ms, 3),
                          "rel_diff_to_single_gpu_plan": err}), flush=True)
        M.close()
