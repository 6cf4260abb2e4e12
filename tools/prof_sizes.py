"""Developer harness: apply time of the hand-written pipeline at grid sizes that are not powers of two, against the
power-of-two embedding (LSFC_POW2_ONLY=1) and the rocFFT pipeline on the exact 2n grid.
usage: python tools/prof_sizes.py [n ...]   (3D cubes)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402


def run(n, reps=5):
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    N = n ** 3
    nu = np.random.default_rng(0).uniform(-0.3, 0.3, N)
    xb = torch.randn(N, dtype=torch.complex128, device="cuda")
    yb = torch.empty_like(xb)
    ref = None
    for name, env, flags in [("mixed radix", {}, 0), ("power-of-two embedding", {"LSFC_POW2_ONLY": "1"}, 0), ("rocFFT on the 2n grid", {}, 2)]:
        for k_ in list(os.environ):
            if k_.startswith("LSFC_"):
                del os.environ[k_]
        os.environ.update(env)
        M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu, flags=flags)
        lsfc.time_apply(M, xb, yb, 2)
        ms = min(lsfc.time_apply(M, xb, yb, reps) / reps for _ in range(3))
        if ref is None:
            ref = yb.clone()
        err = float(torch.linalg.norm(ref - yb) / torch.linalg.norm(ref))
        st = lsfc.profile_apply(M, xb, yb, reps)
        detail = " ".join(f"{s}={t:.3f}" for s, t, b in st)
        print(f"n={n} {name:24s} {M.pipeline:14s} grid={M.padded_dims} apply={ms:8.3f} ms  568B/pt -> {568.0 * N / (ms * 1e-3) / 1e12:5.2f} TB/s  diff={err:.1e} | {detail}", flush=True)
        M.close()


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [48, 96, 160, 192, 320, 384]:
        run(n)
