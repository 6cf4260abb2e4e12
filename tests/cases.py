"""Seeded parity cases shared by the golden generator, the oracle tests and the GPU tests.
Inputs are regenerated from seeds (numpy PCG64 streams are stable across versions); the
golden files hold the oracle's outputs only."""
import numpy as np

from oracle import lsfc_oracle as o

TOL = 1e-10          # BASELINE.json north_star: <= 1e-10 relative l2 vs the CPU reference


def nu_synthetic(dim, seed=1234):
    """sum of 8 Gaussians, centres U(-.3,.3)^d, amplitudes U(-.3,.3), beta U(20,80)."""
    rng = np.random.default_rng(seed)
    cen = rng.uniform(-0.3, 0.3, size=(8, dim))
    amp = rng.uniform(-0.3, 0.3, size=8)
    beta = rng.uniform(20, 80, size=8)

    def nu(*coords):
        out = np.zeros_like(coords[0])
        for c, a, b in zip(cen, amp, beta):
            r2 = sum((x - ci) ** 2 for x, ci in zip(coords, c))
            out = out + a * np.exp(-b * r2)
        return out
    return nu


def grid(n, inclusive):
    if inclusive:                      # examples/example.jl:35-36
        h = 1.0 / (n - 1)
        return -0.5 + h * np.arange(n), h
    h = 1.0 / n                        # examples/example3D.jl:27-29
    return -0.5 + h * np.arange(n), h


def case_2d(name):
    """-> dict(x, h, k, nu, quadRule, M (oracle FastM), b)"""
    spec = {
        "trap21":  dict(n=21, inclusive=True, k=None, quad="trapezoidal", nu=o.gaussian_bump),
        "gv33":    dict(n=33, inclusive=True, k=None, quad="Greengard_Vico", nu=o.gaussian_bump),
        "gv32":    dict(n=32, inclusive=False, k=20.0, quad="Greengard_Vico", nu=nu_synthetic(2)),
        "gv128":   dict(n=128, inclusive=True, k=10 * np.pi, quad="Greengard_Vico", nu=o.gaussian_bump),   # BASELINE configs[0]
    }[name]
    x, h = grid(spec["n"], spec["inclusive"])
    k = spec["k"] if spec["k"] is not None else 1.0 / h
    M = o.build_fast_convolution(x, x, h, k, spec["nu"], quadRule=spec["quad"])
    b = o.random_vector(spec["n"] ** 2)
    return dict(x=x, h=h, k=k, nu=spec["nu"], quadRule=spec["quad"], M=M, b=b, n=spec["n"])


def case_3d(name):
    spec = {
        "gv16":   dict(n=16, k=None, nu=o.gaussian_bump),
        "gv16k10": dict(n=16, k=10.0, nu=nu_synthetic(3)),
        "gv32":   dict(n=32, k=None, nu=o.gaussian_bump),
        "gv32k10": dict(n=32, k=10.0, nu=nu_synthetic(3)),
    }[name]
    n = spec["n"]
    x, h = grid(n, False)
    k = spec["k"] if spec["k"] is not None else 1.0 / h
    X, Y, Z = o.grid3d(x, x, x)
    M = o.build_fast_convolution3d(x, x, x, X, Y, Z, h, k, spec["nu"])
    b = o.random_vector(n ** 3)
    return dict(x=x, h=h, k=k, nu=spec["nu"], M=M, b=b, n=n, X=X, Y=Y, Z=Z)


def plane_wave(k, X):
    return np.exp(1j * k * X)


def sparsifying_pair_2d(n, h, k, nu_flat, eps=0.5):
    """Synthetic (Msp, As) of the reference's structure (src/preconditioner.jl:27-31; the assembly of the real pair,
    src/SparsifyingMatrix2D.jl, is out of scope): with (Lap + k^2) G = -delta the sparsifier As = -(Lap_h + k^2 + i eps)
    turns A = I + k^2 G nu into the sparse Msp = As + k^2 diag(nu) up to discretisation error (5-point Laplacian,
    Dirichlet box).  Column-major grid, x fastest."""
    import scipy.sparse as sp
    e = np.ones(n)
    T = sp.diags([e[:-1], -2 * e, e[:-1]], [-1, 0, 1]) / h**2
    I = sp.identity(n)
    lap = sp.kron(I, T) + sp.kron(T, I)
    N = n * n
    As = (-(lap + (k**2 + 1j * eps) * sp.identity(N))).tocsr().astype(np.complex128)
    Msp = (As + k**2 * sp.diags(np.asarray(nu_flat, dtype=np.float64))).tocsc().astype(np.complex128)
    return Msp, As


def gaussian_2d_quadrature_points(x, k, sig, points):
    """Independent known answer for the 2D Greengard-Vico branch: u(p) = int (i/4) H0^(1)(k |p - y|) f(y) dy for the
    unit-mass Gaussian f, by adaptive quadrature in polar coordinates around p (no FFT, no truncated-kernel symbol: the
    continuous operator the reference discretises, src/FastConvolution.jl:185-231 + src/Functions.jl:40-42).
    Returns [(i, j, u(x_i, x_j))]."""
    import scipy.integrate as si
    import scipy.special as sp

    def value(px, py):
        def integrand(th, r, part):
            yx, yy = px + r * np.cos(th), py + r * np.sin(th)
            fv = np.exp(-(yx ** 2 + yy ** 2) / (2 * sig ** 2)) / (2 * np.pi * sig ** 2)
            v = 0.25j * sp.hankel1(0, k * r) * fv * r
            return v.real if part == 0 else v.imag
        re = si.dblquad(integrand, 0, 1.0, 0, 2 * np.pi, args=(0,), epsabs=1e-11, epsrel=1e-11)[0]
        im = si.dblquad(integrand, 0, 1.0, 0, 2 * np.pi, args=(1,), epsabs=1e-11, epsrel=1e-11)[0]
        return re + 1j * im
    return [(i, j, value(x[i], x[j])) for i, j in points]
