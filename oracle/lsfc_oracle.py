"""CPU oracle for the Lippmann-Schwinger fast-convolution hot path.

TEST INFRASTRUCTURE ONLY.  This module is a numpy/scipy restatement of the
reference's Julia arithmetic (tanderson92/Fast_solver_Lippmann_Schwinger).  It
may be imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; the product path (the HIP library under
``fast_solver_lippmann_schwinger_amd/csrc``) never calls into it.

PARITY UNPINNED: the reference ships no assertions, golden vectors or fixtures
for this path (its ``tests/`` are driver scripts) and Julia is
not installed in the build image, so the reference cannot be executed.  The
oracle is pinned instead by independent identities that hold for the
reference's own formulas (tests/test_oracle.py):
  * dense identity   fastconvolution == b + k^2 * buildConvMatrix * (nu .* b)
  * analytic answer  FFTconvolution(FastM3D, gaussian) == -solRefHelmholtz
  * quadrature       FFTconvolution(FastM 2D Greengard-Vico, gaussian) == int (i/4) H0(k r) gaussian  (adaptive quadrature)
  * padding / shift identities (literal 4n == pre-shifted == reduced 2n)
The GMRES arithmetic follows IterativeSolvers.jl (not vendored, not pinned by
the reference) as documented upstream; its residual histories and solutions are
cross-checked against an independent restarted GMRES (scipy) in tests/test_oracle.py.

Conventions: every grid function is a flat vector in Julia's column-major
order (x fastest).  ``reshape(b, n, m)`` in Julia == ``b.reshape((n, m),
order="F")`` here.  Every function cites the reference file:line it follows
(paths relative to the reference root).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np
import scipy.fft as sfft
import scipy.special as sp

_WORKERS = int(os.environ.get("LSFC_ORACLE_WORKERS", os.cpu_count() or 1))


def _fftn(a):
    return sfft.fftn(a, workers=_WORKERS)


def _ifftn(a):
    return sfft.ifftn(a, workers=_WORKERS)


# ----------------------------------------------------------------------------
# structs: src/FastConvolution.jl:11-27 (FastM), src/FastConvolution3D.jl:7-26
# ----------------------------------------------------------------------------
@dataclass
class FastM:
    """src/FastConvolution.jl:11-27.  GFFT has shape (ne, me): centred
    (fftshift) order for "Greengard_Vico", plain FFT order for "trapezoidal"."""
    GFFT: np.ndarray
    nu: np.ndarray
    ne: int
    me: int
    n: int
    m: int
    omega: float
    quadRule: str = "trapezoidal"


@dataclass
class FastM3D:
    """src/FastConvolution3D.jl:7-26.  GFFT has shape (ne, me, le), centred."""
    GFFT: np.ndarray
    nu: np.ndarray
    ne: int
    me: int
    le: int
    n: int
    m: int
    l: int
    omega: float
    quadRule: str = "Greengard_Vico"


# ----------------------------------------------------------------------------
# truncated-kernel Fourier symbols: src/Functions.jl:40-51
# ----------------------------------------------------------------------------
def gtruncated2d(L, k, s, patch_singular=False):
    """src/Functions.jl:40-42.  ``patch_singular`` replaces the removable 0/0 at
    s == k by its analytic limit (documented deviation)."""
    s = np.asarray(s, dtype=np.float64)
    a = 1j * np.pi / 2 * L * sp.hankel1(0, L * k)
    b = 1j * np.pi / 2 * L * k * sp.hankel1(1, L * k)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = (1 + a * (s * sp.jv(1, L * s)) - b * sp.jv(0, L * s)) / (s**2 - k**2)
    if patch_singular:
        lim = (a * L * k * sp.jv(0, L * k) + b * L * sp.jv(1, L * k)) / (2 * k)
        g = np.where(s == k, lim, g)
    return g


def gtruncated3d(L, k, s, patch_singular=False):
    """src/Functions.jl:45-51; Julia sinc(x) = sin(pi x)/(pi x) == numpy.sinc."""
    s = np.asarray(s, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        g = (-1 + np.exp(1j * L * k) * (np.cos(L * s) - 1j * k * L * np.sinc(L * s / np.pi))) / (k**2 - s**2)
    if patch_singular:
        lim = (1j * L - (1j / k) * np.sin(L * k) * np.exp(1j * L * k)) / (2 * k)
        g = np.where(s == k, lim, g)
    return g


def sol_ref_helmholtz(x, y, z, sigma, k):
    """src/Functions.jl:32-36: closed form of G * (unit-mass centred Gaussian)."""
    r = np.sqrt(x**2 + y**2 + z**2)
    u = (np.exp(-sigma**2 * k**2 / 2) / (4 * np.pi * r)) * (
        np.real(np.exp(-1j * k * r) * sp.erf((2 * sigma**2 * 1j * k - 2 * r) / (2 * np.sqrt(2 * sigma**2))))
        - 1j * np.sin(k * r))
    return u


# ----------------------------------------------------------------------------
# trapezoidal (Duan-Rokhlin) kernel: src/FastConvolution.jl:407-469
# ----------------------------------------------------------------------------
def reference_vals_trap_rule():
    """src/FastConvolution.jl:407-415."""
    x = 2.0 ** (-np.arange(6.0))
    w = np.array([1 - 0.892j, 1 - 1.35j, 1 - 1.79j, 1 - 2.23j, 1 - 2.67j, 1 - 3.11j])
    return x, w


def build_gconv(x, y, h, n, m, D0, k):
    """src/FastConvolution.jl:425-469 (odd n only; the even branch of the
    reference dereferences an undefined variable).  The singular sample is the
    centre entry (index n-1, m-1 zero-based), as ``findall(R.==0)[1]`` finds."""
    if n % 2 != 1 or m % 2 != 1:
        raise ValueError("so far only works for n odd (src/FastConvolution.jl:455)")
    xe = x[0] - (n - 1) / 2 * h + h * np.arange(2 * n - 1)
    ye = y[0] - (m - 1) / 2 * h + h * np.arange(2 * m - 1)
    Xe, Ye = np.meshgrid(xe, ye, indexing="ij")
    R = np.sqrt(Xe**2 + Ye**2)
    R[n - 1, m - 1] = 1.0
    Ge = 1j / 4 * sp.hankel1(0, k * R) * h**2          # :398 alpha*hankelh1(0,k R)
    Ge[n - 1, m - 1] = 1j / 4 * D0 * h**2               # :465
    return Ge


def build_conv_matrix(k, X, Y, D0, h):
    """src/FastConvolution.jl:497-513: dense N x N oracle of the same kernel."""
    N = len(X)
    G = np.zeros((N, N), dtype=np.complex128)
    for ii in range(N):
        r = np.sqrt((X - X[ii])**2 + (Y - Y[ii])**2)
        r[ii] = 1.0
        G[ii, :] = 1j / 4 * sp.hankel1(0, k * r) * h**2
        G[ii, ii] = 1j / 4 * D0 * h**2
    return G


# ----------------------------------------------------------------------------
# builders: src/FastConvolution.jl:170-236, src/FastConvolution3D.jl:68-132
# ----------------------------------------------------------------------------
def grid2d(x, y):
    """X = repeat(x,1,m)[:], Y = repeat(y',n,1)[:]  (FastConvolution.jl:180-181)."""
    n, m = len(x), len(y)
    X = np.repeat(x[:, None], m, axis=1).reshape(-1, order="F")
    Y = np.repeat(y[None, :], n, axis=0).reshape(-1, order="F")
    return X, Y


def grid3d(x, y, z):
    """examples/example3D.jl:33-39 (i fastest)."""
    Xg, Yg, Zg = np.meshgrid(x, y, z, indexing="ij")
    return (Xg.reshape(-1, order="F"), Yg.reshape(-1, order="F"), Zg.reshape(-1, order="F"))


def build_fast_convolution(x, y, h, k, nu, quadRule="trapezoidal", patch_singular=False):
    """src/FastConvolution.jl:170-236.  ``nu`` is a callable nu(X, Y)."""
    n, m = len(x), len(y)
    X, Y = grid2d(x, y)
    if quadRule == "trapezoidal":
        _, D = reference_vals_trap_rule()
        idx = int(round(k * h))                          # :176 (1-based D[round(Int,k*h)]; Julia throws BoundsError outside 1..6)
        if not 1 <= idx <= len(D):
            raise IndexError(f"BoundsError: attempt to access {len(D)}-element Vector at index [{idx}]")
        D0 = D[idx - 1]
        Ge = build_gconv(x, y, h, n, m, D0, k)
        GFFT = _fftn(Ge)
        return FastM(GFFT, np.asarray(nu(X, Y), dtype=np.float64), 2 * n - 1, 2 * m - 1, n, m, float(k))
    if quadRule == "Greengard_Vico":
        Lp = 4 * (abs(x[-1] - x[0]) + h)                 # :187
        L = (abs(x[-1] - x[0]) + h) * 1.5                # :188
        kx = np.arange(-2 * n, 2 * n, dtype=np.float64)  # :195 / :220 (odd branch identical)
        ky = np.arange(-2 * m, 2 * m, dtype=np.float64)
        KX = (2 * np.pi / Lp) * np.repeat(kx[:, None], 4 * m, axis=1)
        KY = (2 * np.pi / Lp) * np.repeat(ky[None, :], 4 * n, axis=0)
        S = np.sqrt(KX**2 + KY**2)
        GFFT = gtruncated2d(L, float(k), S, patch_singular)
        return FastM(GFFT, np.asarray(nu(X, Y), dtype=np.float64), 4 * n, 4 * m, n, m, float(k),
                     quadRule="Greengard_Vico")
    raise ValueError(f"unknown quadRule {quadRule!r}")


def build_fast_convolution3d(x, y, z, X, Y, Z, h, k, nu, quadRule="Greengard_Vico", patch_singular=False):
    """src/FastConvolution3D.jl:68-101 (even n; the odd branch of the reference
    uses a different, unscaled lattice and is treated as broken)."""
    if quadRule != "Greengard_Vico":
        raise ValueError("buildFastConvolution3D only implements Greengard_Vico")
    n, m, l = len(x), len(y), len(z)
    if n % 2:
        raise ValueError("3D builder: even n only")
    Lp = 4 * (abs(x[-1] - x[0]) + h)                     # :72
    L = (abs(x[-1] - x[0]) + h) * 1.8                    # :73
    kx = (2 * np.pi / Lp) * np.arange(-2 * n, 2 * n, dtype=np.float64)
    ky = (2 * np.pi / Lp) * np.arange(-2 * m, 2 * m, dtype=np.float64)
    kz = (2 * np.pi / Lp) * np.arange(-2 * l, 2 * l, dtype=np.float64)
    S = np.sqrt(kx[:, None, None]**2 + ky[None, :, None]**2 + kz[None, None, :]**2)
    GFFT = gtruncated3d(L, float(k), S, patch_singular)  # :95-99
    return FastM3D(GFFT, np.asarray(nu(X, Y, Z), dtype=np.float64), 4 * n, 4 * m, 4 * l, n, m, l, float(k))


# ----------------------------------------------------------------------------
# the apply: src/FastConvolution.jl:43-154, src/FastConvolution3D.jl:31-63
# ----------------------------------------------------------------------------
def fastconvolution(M: FastM, b):
    """src/FastConvolution.jl:58-107: b + omega^2 * crop(ifft(GFFT .* fft(pad(nu .* b))))."""
    b = np.asarray(b, dtype=np.complex128)
    if M.quadRule == "trapezoidal":
        BExt = np.zeros((M.ne, M.me), dtype=np.complex128)
        BExt[:M.n, :M.m] = (M.nu * b).reshape((M.n, M.m), order="F")
        BFft = _fftn(BExt)
        BFft = M.GFFT * BFft
        BExt = _ifftn(BFft)
        B = M.omega**2 * BExt[M.n - 1:2 * M.n - 1, M.m - 1:2 * M.m - 1]   # :82
    elif M.quadRule == "Greengard_Vico":
        BExt = np.zeros((M.ne, M.me), dtype=np.complex128)
        BExt[:M.n, :M.m] = (M.nu * b).reshape((M.n, M.m), order="F")
        BFft = sfft.fftshift(_fftn(BExt))
        BFft = M.GFFT * BFft
        BExt = _ifftn(sfft.ifftshift(BFft))
        B = M.omega**2 * BExt[:M.n, :M.m]                                 # :101
    else:
        raise NameError("B not defined (src/FastConvolution.jl:106)")
    return b + B.reshape(-1, order="F")


def mul(M, b):
    """``*``: src/FastConvolution.jl:43-48, src/FastConvolution3D.jl:31-37."""
    if isinstance(M, FastM3D):
        b = np.asarray(b, dtype=np.complex128)
        return b + M.omega**2 * fft_convolution(M, M.nu * b)
    return fastconvolution(M, b)


def fft_convolution(M, b):
    """Bare convolution.  2D: src/FastConvolution.jl:110-154 (quirks kept: the
    trapezoidal branch multiplies by nu, the GV branch does not; both use ne
    and n for both dimensions).  3D: src/FastConvolution3D.jl:39-63."""
    b = np.asarray(b, dtype=np.complex128)
    if isinstance(M, FastM3D):
        BExt = np.zeros((M.ne, M.ne, M.le), dtype=np.complex128)           # :48 (ne twice)
        BExt[:M.n, :M.m, :M.l] = b.reshape((M.n, M.m, M.l), order="F")
        BFft = sfft.fftshift(_fftn(BExt))
        BFft = M.GFFT * BFft
        BExt = _ifftn(sfft.ifftshift(BFft))
        return BExt[:M.n, :M.m, :M.l].reshape(-1, order="F")
    if M.quadRule == "trapezoidal":
        ind = M.n - 1
        BExt = np.zeros((M.ne, M.ne), dtype=np.complex128)
        BExt[:M.n, :M.m] = (M.nu * b).reshape((M.n, M.m), order="F")
        BExt = _ifftn(M.GFFT * _fftn(BExt))
        B = BExt[ind:ind + M.n, ind:ind + M.n]
    elif M.quadRule == "Greengard_Vico":
        BExt = np.zeros((M.ne, M.ne), dtype=np.complex128)
        BExt[:M.n, :M.n] = b.reshape((M.n, M.n), order="F")
        BFft = sfft.fftshift(_fftn(BExt))
        BExt = _ifftn(sfft.ifftshift(M.GFFT * BFft))
        B = BExt[:M.n, :M.n]
    else:
        raise NameError("B not defined")
    return B.reshape(-1, order="F")


def size(M, dim=None):
    """src/FastConvolution.jl:31-37 (tuple-of-tuples quirk kept)."""
    if dim is not None:
        return M.nu.shape[0]
    return (M.nu.shape, M.nu.shape)


def eltype(M):
    """src/FastConvolution.jl:39-41."""
    return M.GFFT.dtype


# ----------------------------------------------------------------------------
# equivalent reduced pipelines -- used by tests to prove
# the identities the HIP build relies on, and as the CPU baseline at sizes where
# the literal (4n)^3 arrays do not fit in host memory.
# ----------------------------------------------------------------------------
def reduce_symbol(GFFT, dims):
    """Literal centred (4n)^d GV symbol -> (2n)^d symbol in plain FFT order that
    yields the same cropped convolution: T = ifft(ifftshift(GFFT)), keep offsets
    -n..n-1 per axis, fft on the (2n)^d grid."""
    T = _ifftn(sfft.ifftshift(GFFT))
    for ax, n in enumerate(dims):
        ne = T.shape[ax]
        idx = np.concatenate([np.arange(0, n), np.arange(ne - n, ne)])
        T = np.take(T, idx, axis=ax)
    return _fftn(T)


def convolve_reduced(G2, b, dims):
    """crop(ifft(G2 .* fft(pad_2n(b)))) on the reduced (2n)^d grid."""
    B = np.zeros(G2.shape, dtype=np.complex128)
    sl = tuple(slice(0, n) for n in dims)
    B[sl] = np.asarray(b, dtype=np.complex128).reshape(dims, order="F")
    B = _ifftn(G2 * _fftn(B))
    return B[sl].reshape(-1, order="F")


def apply_reduced(G2, nu, omega, b, dims):
    b = np.asarray(b, dtype=np.complex128)
    return b + omega**2 * convolve_reduced(G2, nu * b, dims)


def reduced_symbol_gv3d(n, m, l, box, k, patch_singular=True, chunk=8, threads=None):
    """Reduced (2n,2m,2l) FFT-order symbol of buildFastConvolution3D WITHOUT
    materialising the (4n)^3 cube: the inverse transform of the centred literal
    symbol is done plane-chunk by plane-chunk along x with pruned outputs.
    ``box`` = |x_end - x_1| + h.  Mirrors what the HIP symbol generator does.

    Two exact shortcuts keep the full-size cases (n = 256, 512) affordable on the host: the literal symbol depends on
    k only through kx^2 + ky^2 + kz^2, so the samples at +j along an axis are bit-identical copies of those at -j and
    are copied, not re-evaluated (one octant of closed-form evaluations instead of the cube); and independent plane chunks are evaluated on a thread pool (numpy and scipy.fft release the GIL)."""
    from concurrent.futures import ThreadPoolExecutor
    Lp, L = 4 * box, 1.8 * box
    kx = (2 * np.pi / Lp) * np.arange(-2 * n, 2 * n, dtype=np.float64)
    ky = (2 * np.pi / Lp) * np.arange(-2 * m, 2 * m, dtype=np.float64)
    kz = (2 * np.pi / Lp) * np.arange(-2 * l, 2 * l, dtype=np.float64)
    iy = np.concatenate([np.arange(0, m), np.arange(3 * m, 4 * m)])
    iz = np.concatenate([np.arange(0, l), np.arange(3 * l, 4 * l)])
    ix = np.concatenate([np.arange(0, n), np.arange(3 * n, 4 * n)])
    t2 = np.empty((4 * n, 2 * m, 2 * l), dtype=np.complex128)
    threads = threads or max(1, min(_WORKERS, 32))
    inner = max(1, _WORKERS // threads)
    ky2, kz2 = ky[None, :, None]**2, kz[None, None, :]**2

    def planes(x0):
        xs = slice(x0, min(2 * n + 1, x0 + chunk))                 # kx = -2n .. 0 only
        # the same holds along y and z: evaluate the quadrant ky <= 0, kz <= 0 and fill the rest with bit-identical copies
        S = np.sqrt(kx[xs, None, None]**2 + ky2[:, :2 * m + 1] + kz2[:, :, :2 * l + 1])     # src/FastConvolution3D.jl:98
        gq = gtruncated3d(L, float(k), S, patch_singular)
        g = np.empty((gq.shape[0], 4 * m, 4 * l), dtype=np.complex128)
        g[:, :2 * m + 1, :2 * l + 1] = gq
        g[:, 2 * m + 1:, :2 * l + 1] = gq[:, 2 * m - 1:0:-1, :]
        g[:, :, 2 * l + 1:] = g[:, :, 2 * l - 1:0:-1]
        g = sfft.ifftshift(g, axes=(1, 2))
        g = sfft.ifft(g, axis=2, workers=inner, overwrite_x=True)[:, :, iz]       # pruned outputs: z first, then y on half the data
        t2[xs] = sfft.ifft(g, axis=1, workers=inner, overwrite_x=True)[:, iy]

    with ThreadPoolExecutor(max_workers=threads) as pool:
        list(pool.map(planes, range(0, 2 * n + 1, chunk)))
    t2[2 * n + 1:] = t2[2 * n - 1:0:-1]                            # kx = +j  <-  kx = -j, j = 1 .. 2n-1
    t2 = sfft.ifft(sfft.ifftshift(t2, axes=0), axis=0, workers=_WORKERS, overwrite_x=True)[ix]
    return sfft.fftn(t2, workers=_WORKERS, overwrite_x=True)


# ----------------------------------------------------------------------------
# delta-source sampling: src/FastConvolution3D.jl:136-160, FastConvolution.jl:278-306
# ----------------------------------------------------------------------------
def sample_g_conv(indS, fastconv):
    """sampleGConv / sampleG3D(…, fastconv): one FFTconvolution per delta source."""
    N = fastconv.nu.shape[0]
    Gc = np.zeros((len(indS), N), dtype=np.complex128)
    for i, ii in enumerate(indS):
        e = np.zeros(N, dtype=np.complex128)
        e[ii] = 1.0
        Gc[i, :] = fft_convolution(fastconv, e)
    return Gc


# ----------------------------------------------------------------------------
# GMRES: IterativeSolvers.jl gmres! (external to the reference; call sites
# examples/example.jl:85,91, examples/example3D.jl:78).  Algorithm as documented
# upstream: restarted, left-preconditioned, modified
# Gram-Schmidt by default, residual estimated through the Hessenberg null vector,
# Givens least squares at restart/convergence.
# ----------------------------------------------------------------------------
@dataclass
class ConvergenceHistory:
    resnorm: list = field(default_factory=list)
    mvps: int = 0
    iters: int = 0
    isconverged: bool = False


def _givens(f, g):
    """LinearAlgebra.givensAlgorithm for complex (c real, s complex): returns
    (c, s, r) with [c s; -conj(s) c] [f; g] = [r; 0]."""
    if g == 0:
        return 1.0, 0.0 + 0.0j, f
    if f == 0:
        return 0.0, np.conj(g) / abs(g), abs(g)
    d = np.hypot(abs(f), abs(g))
    c = abs(f) / d
    s = (f / abs(f)) * np.conj(g) / d
    return c, s, (f / abs(f)) * d


def _solve_least_squares(H, beta, k):
    """hessenberg.jl: Givens QR of the k x (k-1) Hessenberg block, then the
    triangular solve; returns y (length k-1)."""
    width = k - 1
    Hh = H[:k, :width].copy()
    rhs = np.zeros(k, dtype=np.complex128)
    rhs[0] = beta
    for i in range(width):
        c, s, _ = _givens(Hh[i, i], Hh[i + 1, i])
        Hh[i, i] = c * Hh[i, i] + s * Hh[i + 1, i]
        for j in range(i + 1, width):
            tmp = -np.conj(s) * Hh[i, j] + c * Hh[i + 1, j]
            Hh[i, j] = c * Hh[i, j] + s * Hh[i + 1, j]
            Hh[i + 1, j] = tmp
        tmp = -np.conj(s) * rhs[i] + c * rhs[i + 1]
        rhs[i] = c * rhs[i] + s * rhs[i + 1]
        rhs[i + 1] = tmp
    y = np.zeros(width, dtype=np.complex128)
    for i in range(width - 1, -1, -1):
        y[i] = (rhs[i] - Hh[i, i + 1:width] @ y[i + 1:]) / Hh[i, i]
    return y


def gmres(x, A, b, Pl=None, abstol=0.0, reltol=None, restart=None, maxiter=None,
          initially_zero=False, orth_meth="ModifiedGramSchmidt"):
    """``gmres!(x, A, b; Pl, abstol, reltol, restart, maxiter, log=true)``.
    ``A`` is a callable v -> A v, ``Pl`` a callable v -> Pl \\ v (or None).
    x is updated in place; returns (x, ConvergenceHistory)."""
    N = b.shape[0]
    reltol = np.sqrt(np.finfo(np.float64).eps) if reltol is None else reltol
    restart = min(20, N) if restart is None else restart
    maxiter = N if maxiter is None else maxiter
    V = np.zeros((N, restart + 1), dtype=np.complex128, order="F")
    H = np.zeros((restart + 1, restart), dtype=np.complex128)
    nullvec = np.ones(restart + 1, dtype=np.complex128)
    hist = ConvergenceHistory()

    def init(skip_mv=False):
        V[:, 0] = b
        if not skip_mv:
            V[:, 0] -= A(x)
        if Pl is not None:
            V[:, 0] = Pl(V[:, 0])
        beta = np.linalg.norm(V[:, 0])
        V[:, 0] *= 1.0 / beta
        return beta

    hist.mvps = 1 if initially_zero else 0
    beta = init(skip_mv=initially_zero)
    current, accumulator = beta, 1.0
    nullvec[0] = 1.0
    tol = max(reltol * current, abstol)
    k, iteration = 1, 0
    while not (iteration >= maxiter or current <= tol):
        w = A(V[:, k - 1])                                   # expand!
        if Pl is not None:
            w = Pl(w)
        hist.mvps += 1
        h = H[:k, k - 1]
        if orth_meth == "ModifiedGramSchmidt":
            for i in range(k):
                h[i] = np.vdot(V[:, i], w)
                w = w - h[i] * V[:, i]
            nrm = np.linalg.norm(w)
        else:
            h[:] = V[:, :k].conj().T @ w
            w = w - V[:, :k] @ h
            nrm = np.linalg.norm(w)
            if orth_meth == "DGKS":
                # orthogonalize.jl: repeat `while nrm < projection_size / sqrt(2)`, projection_size being the norm of
                # the latest correction
                proj = np.linalg.norm(h)
                while nrm < proj / np.sqrt(2.0):
                    corr = V[:, :k].conj().T @ w
                    proj = np.linalg.norm(corr)
                    w = w - V[:, :k] @ corr
                    h += corr
                    nrm = np.linalg.norm(w)
        V[:, k] = w * (1.0 / nrm)
        H[k, k - 1] = nrm
        nullvec[k] = -np.conj(np.vdot(nullvec[:k], H[:k, k - 1]) / H[k, k - 1])   # update_residual!
        accumulator += abs(nullvec[k])**2
        current = beta / np.sqrt(accumulator)
        k += 1
        if k == restart + 1 or current <= tol:
            y = _solve_least_squares(H, beta, k)
            x += V[:, :k - 1] @ y
            k = 1
            if not current <= tol:
                beta = init()
                accumulator = 1.0
                nullvec[0] = 1.0
                hist.mvps += 1
        iteration += 1
        hist.resnorm.append(float(current))
    hist.iters = iteration
    hist.isconverged = bool(current <= tol)
    return x, hist


# ----------------------------------------------------------------------------
# synthetic benchmark inputs
# ----------------------------------------------------------------------------
def gaussian_bump(*coords):
    """examples/example.jl:48, examples/example3D.jl:43."""
    r2 = sum(c**2 for c in coords)
    out = 0.3 * np.exp(-40 * r2)
    for c in coords:
        out = out * (np.abs(c) < 0.48)
    return out


def random_vector(N, seed=20250224):
    rng = np.random.default_rng(seed)
    return rng.standard_normal(N) + 1j * rng.standard_normal(N)


# ----------------------------------------------------------------------------
# SparsifyingPreconditioner apply: src/preconditioner.jl:27-58 (lu(Msp)), :132-170 (ldiv!)
# ----------------------------------------------------------------------------
class SparsifyingPreconditioner:
    """``b <- MspInv \\ (As * b)`` with ``MspInv = lu(Msp)`` (src/preconditioner.jl:35, 139, 160).  scipy's SuperLU stands
    in for UMFPACK: any LU of Msp applies the same operator up to rounding."""

    def __init__(self, Msp, As):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        self.Msp = sp.csc_matrix(Msp, dtype=np.complex128)
        self.As = sp.csr_matrix(As, dtype=np.complex128)
        self.MspInv = spla.splu(self.Msp)

    def solve(self, b):                      # \\(M, b), :132-145
        return self.MspInv.solve(self.As @ np.asarray(b, dtype=np.complex128))

    def ldiv_(self, b):                      # ldiv!(M, b), :147-170
        b[:] = self.solve(b)
        return b

    def __call__(self, b):
        self.ldiv_(b)
