"""Same-process A/B of one run-time tuning knob on the 3D apply: python tools/ab_knob.py <knob> <n> v0 v1 ... [rounds]
(apply time by HIP events, best of 3 x 10 applies per visit, the values visited round-robin; results compared bit for bit)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc
knob, n = sys.argv[1], int(sys.argv[2])
vals = [int(v) for v in sys.argv[3:]]
h = 1.0 / n; x = -0.5 + h * np.arange(n)
nu = np.random.default_rng(0).uniform(-0.3, 0.3, n ** 3)
M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
xb = torch.randn(n ** 3, dtype=torch.complex128, device="cuda"); yb = torch.empty_like(xb)
ref = None
for rnd in range(3):
    for v in vals:
        M.set_tuning(**{knob: v})
        lsfc.time_apply(M, xb, yb, 3)
        ms = min(lsfc.time_apply(M, xb, yb, 10) / 10 for _ in range(3))
        if ref is None: ref = yb.clone()
        print(f"round {rnd} {knob}={v:4d}  apply {ms:7.3f} ms  same={bool(torch.equal(ref, yb))}", flush=True)
