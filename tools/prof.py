"""Developer harness: time the device apply stage by stage under tuning variants.
usage: python tools/prof.py [n ...]   (3D cubes).  Variants: (name, env at plan creation, runtime knobs)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc  # noqa: E402

PEAK = 8.0e12
BASE = dict(split_x=1, split_s=1, split_z=-1, sym_prefetch=-1, ytile_g=0, ytile_z=0, z_half=-1, tw_lds=1, z_persist=-1, xlane=-1)

VARIANTS = [("auto (persistent pipelined z pass)", {}, {}),
            ("one tile per workgroup (round 1)", {}, dict(z_persist=0)),
            ("persistent, whole-complex exchanges", {}, dict(z_persist=1)),
            ("persistent, split exchanges", {}, dict(z_persist=2)),
            ("persistent, whole, symbol after stage 0", {}, dict(z_persist=3)),
            ("persistent, split, symbol after stage 0", {}, dict(z_persist=4)),
            ("persistent half tiles, two workgroups per CU", {}, dict(z_persist=5)),
            ("persistent whole tiles by tickets", {}, dict(z_persist=6)),
            ("whole tiles by tickets, LDS exchanges only", {}, dict(z_persist=6, xlane=0)),
            ("whole tiles, lane exchange", {}, dict(z_persist=6, xlane=1)),
            ("whole tiles, lane exchange + mirror symbol from L2", {}, dict(z_persist=6, xlane=3)),
            ("whole tiles, lane exchange + row pairs share the symbol registers", {}, dict(z_persist=6, xlane=5)),

            ("half tiles by tickets, LDS exchanges only", {}, dict(z_persist=5, xlane=0))]
if os.environ.get("PROF_ONLY"):
    keep = [int(i) for i in os.environ["PROF_ONLY"].split(",")]
    VARIANTS = [v for i, v in enumerate(VARIANTS) if i in keep]


def run(n, reps=5):
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    N = n ** 3
    rng = np.random.default_rng(0)
    nu = rng.uniform(-0.3, 0.3, N)
    xb = torch.randn(N, dtype=torch.complex128, device="cuda")
    yb = torch.empty_like(xb)
    ref = None
    last_env, M = None, None
    for name, env, knobs in VARIANTS:
        if env != last_env:
            if M is not None:
                M.close()
            for k_ in list(os.environ):
                if k_.startswith("LSFC_"):
                    del os.environ[k_]
            os.environ.update(env)
            M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
            last_env = env
        kn = dict(BASE); kn.update(knobs)
        M.set_tuning(**kn)
        lsfc.time_apply(M, xb, yb, 2)
        ms = min(lsfc.time_apply(M, xb, yb, reps) / reps for _ in range(3))
        if ref is None:
            ref = yb.clone()
        same = bool(torch.equal(ref, yb)) or float(torch.linalg.norm(ref - yb) / torch.linalg.norm(ref))
        frac = 568.0 * N / (ms * 1e-3) / PEAK
        st = lsfc.profile_apply(M, xb, yb, reps)
        detail = " ".join(f"{s}={t:.3f}({b/(t*1e-3)/1e12:.2f})" for s, t, b in st)
        print(f"n={n} {name:34s} apply={ms:7.3f} ms {1e3/ms:7.1f}/s roof={frac*100:5.1f}% same={same} | {detail}", flush=True)
    M.close()


if __name__ == "__main__":
    for n in [int(a) for a in sys.argv[1:]] or [512]:
        run(n)
