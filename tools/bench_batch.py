"""Multi-right-hand-side apply: ms per right-hand side and bytes model for R = 1, 2, 4, 8 (lsfc_apply_batch, one fused pass per
group) against R single applies.  usage: python tools/bench_batch.py [n ...]   (3D cubes; default 48 128 256)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402


def run(n):
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    g = np.exp(-40 * x ** 2)
    nu = (0.3 * g[:, None, None] * g[None, :, None] * g[None, None, :]).reshape(-1)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
    N = n ** 3
    M.set_tuning(batch_fuse=1)            # measure the fused pass at every size (auto falls back to member-by-member above 256^3 padded points)
    out = []
    for R in (1, 2, 4, 8):
        B = torch.randn(R, N, dtype=torch.complex128, device="cuda")
        reps = max(3, min(200, int(2e9 / (N * R * 600))))
        for fused in (True, False):
            def once():
                if fused:
                    lsfc.apply_batch(M, B, 0)
                else:
                    for j in range(R):
                        M * B[j]
            once(); M.synchronize(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                once()
            M.synchronize(); torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / reps * 1e3 / R
            model = ((35 - 8 + 8.0 / R) * 16 + 8) if fused else 568.0
            out.append({"n": n, "R": R, "fused_batch": fused, "ms_per_rhs": ms, "bytes_per_point_model": model,
                        "algorithmic_GBps_of_568": 568.0 * N / (ms * 1e-3) / 1e9})
            print(json.dumps(out[-1]), flush=True)
    M.close()


if __name__ == "__main__":
    for n in ([int(a) for a in sys.argv[1:]] or [48, 128, 256]):
        run(n)
