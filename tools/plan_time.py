"""Developer harness: wall time of plan creation (symbol generation included), first and second time in a process
(the first pays rocFFT's runtime kernel compilation unless its on-disk cache is warm).  usage: python tools/plan_time.py [n ...]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402

torch.zeros(1, device="cuda")
for n in [int(a) for a in sys.argv[1:]] or [256, 512]:
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
        torch.cuda.synchronize(); t = time.time() - t0
        print(f"n={n} plan creation #{rep + 1}: {t:.3f} s", flush=True)
        M.close()
