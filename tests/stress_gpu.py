"""Randomised cross-check of the pipelines (GPU): random grid shapes, symbols, contrasts and flags; the pruned HIP
pipeline, the rocFFT-reduced and rocFFT-literal pipelines, the simulated distributed ranks and the CPU oracle must agree.
usage: python tests/stress_gpu.py [seconds]   (not collected by pytest; lives under tests/ because it uses the oracle)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fast_solver_lippmann_schwinger_amd as ls          # noqa: E402
from fast_solver_lippmann_schwinger_amd.distributed import SimulatedRanks   # noqa: E402
from oracle import lsfc_oracle as o                      # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b)))


def even_symbol(rng, shape):
    """random symbol that is even in every axis (like the Green's symbols), centred layout"""
    G = rng.standard_normal(shape) + 1j * rng.standard_normal(shape)
    G = np.fft.ifftshift(G)
    for ax in range(G.ndim):
        idx = (-np.arange(G.shape[ax])) % G.shape[ax]
        G = 0.5 * (G + np.take(G, idx, axis=ax))
    return np.fft.fftshift(G)


def main(budget):
    rng = np.random.default_rng(int(time.time()))
    t_end = time.time() + budget
    it = 0
    worst = 0.0
    while time.time() < t_end:
        it += 1
        dim = rng.choice([2, 3])
        if rng.random() < 0.5:
            sizes = [int(rng.choice([16, 32, 64] if dim == 3 else [16, 32, 64, 128, 256])) for _ in range(dim)]
        else:                                          # arbitrary (odd, non-power-of-two) sizes
            sizes = [int(rng.integers(9, 81 if dim == 3 else 400)) for _ in range(dim)]
        N = int(np.prod(sizes))
        lit = tuple(4 * s for s in sizes)
        G = even_symbol(rng, lit) if rng.random() < 0.5 else (rng.standard_normal(lit) + 1j * rng.standard_normal(lit))
        nu = rng.uniform(-0.3, 0.3, N)
        b = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        k = float(rng.uniform(1, 30))
        G2 = o.reduce_symbol(G, tuple(sizes))
        ref = o.apply_reduced(G2, nu, k, b, tuple(sizes))
        errs = {}
        for flags, name in [(0, "pruned"), (2, "rocfft-reduced"), (1, "rocfft-literal")]:
            if dim == 3:
                M = ls.FastM3D(G, nu, *lit, *sizes, k, flags=flags)
            else:
                M = ls.FastM(G, nu, *lit, *sizes, k, quadRule="Greengard_Vico", flags=flags)
            errs[name] = rel(M * b, ref)
            M.close()
        even = all(v % 2 == 0 for v in sizes)
        Lx8 = ls._lib.load().lsfc_padded_length(sizes[0]) // 8
        ranks = [p for p in (1, 2, 4, 8) if sizes[-1] % p == 0 and Lx8 % p == 0]
        if dim == 3 and even and ranks:
            # builder + simulated ranks against the single-GPU builder (same generated symbol)
            n, m, l = sizes
            h = 1.0 / n
            x = -0.5 + h * np.arange(n)
            Mb = ls.buildFastConvolution3D(x, x[:1].repeat(m), x[:1].repeat(l), None, None, None, h, k, nu)
            yb = Mb * b
            P = int(rng.choice(ranks))
            S = SimulatedRanks(n, m, l, h, k, nu, P)
            errs[f"sim{P}"] = rel(S.apply(b), yb)
            S.close(); Mb.close()
        worst = max(worst, max(errs.values()))
        status = "ok" if max(errs.values()) < 1e-11 else "FAIL"
        print(f"{it:4d} dims={sizes} k={k:5.1f} " + " ".join(f"{n_}={e:.1e}" for n_, e in errs.items()) + f" {status}", flush=True)
        if status == "FAIL":
            sys.exit(1)
    print(f"stress ok: {it} cases, worst relative error {worst:.2e}")


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0)
