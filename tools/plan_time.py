import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import fast_solver_lippmann_schwinger_amd as lsfc
for n in (256, 512):
    h = 1.0 / n; x = -0.5 + h * np.arange(n)
    nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
        torch.cuda.synchronize(); print(f"n={n} rep={rep} plan creation {time.time() - t0:.3f} s", flush=True)
        M.close()
