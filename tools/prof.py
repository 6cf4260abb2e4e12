"""Developer harness: time the device apply stage by stage under tuning variants.
usage: python tools/prof.py [n ...]   (3D cubes; env LSFC_* variants are swept in-process)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402

PEAK = 8.0e12


def run(n, variants, reps=5):
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    N = n ** 3
    rng = np.random.default_rng(0)
    nu = rng.uniform(-0.3, 0.3, N)
    for name, env in variants:
        for k_, v in env.items():
            os.environ[k_] = v
        t0 = time.time()
        M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
        t_plan = time.time() - t0
        xb = torch.randn(N, dtype=torch.complex128, device="cuda")
        yb = torch.empty_like(xb)
        lsfc.time_apply(M, xb, yb, 2)
        ms = lsfc.time_apply(M, xb, yb, reps) / reps
        frac = 568.0 * N / (ms * 1e-3) / PEAK
        print(f"n={n} {name:28s} pipeline={M.pipeline} plan={t_plan:.1f}s apply={ms:.3f} ms  {1e3/ms:.1f} applies/s  alg-roofline={frac*100:.1f}%", flush=True)
        for st, sms, sb in lsfc.profile_apply(M, xb, yb, reps):
            print(f"      {st:10s} {sms:8.3f} ms  {sb/1e9:7.2f} GB  {sb/(sms*1e-3)/1e12:6.2f} TB/s", flush=True)
        M.close()
        del xb, yb
        torch.cuda.empty_cache()


if __name__ == "__main__":
    ns = [int(a) for a in sys.argv[1:]] or [256]
    variants = [("split_x=1 split_s=1", {"LSFC_SPLIT_X": "1", "LSFC_SPLIT_S": "1"}),
                ("split_x=0 split_s=1", {"LSFC_SPLIT_X": "0", "LSFC_SPLIT_S": "1"}),
                ("split_x=1 split_s=0", {"LSFC_SPLIT_X": "1", "LSFC_SPLIT_S": "0"}),
                ("split_x=0 split_s=0", {"LSFC_SPLIT_X": "0", "LSFC_SPLIT_S": "0"})]
    for n in ns:
        run(n, variants)
