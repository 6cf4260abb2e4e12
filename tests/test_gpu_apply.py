"""GPU parity tests: the HIP operator, called through the C ABI (ctypes), against the CPU
oracle on identical seeded inputs and against the committed golden vectors.
Tolerance: <= 1e-10 relative l2 (BASELINE.json north_star); observed ~1e-15."""
import os

import numpy as np
import pytest

from oracle import lsfc_oracle as o
import cases
from cases import TOL
from conftest import rel_err

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
LITERAL, FORCE_ROCFFT, PATCH = 1, 2, 4
# padded line lengths of the hand-written pipeline (csrc/pruned.hip): 2^k, 3*2^k, 5*2^k
LENGTHS = sorted([32, 64, 128, 256, 512, 1024, 2048, 48, 96, 192, 384, 768, 1536, 80, 160, 320, 640, 1280])


def best_length(n):
    """smallest hand-written line length >= max(2n, 32) (mirror of pruned_best_length)"""
    return next(L for L in LENGTHS if L >= max(2 * n, 32))


# ---------------------------------------------------------------------------- 2D, literal-symbol constructor
@pytest.mark.parametrize("name,flags,pipeline", [
    ("trap21", 0, "pruned-hip"),               # odd n: embedded in the next power-of-two working grid (2*32)
    ("trap21", FORCE_ROCFFT, "rocfft-reduced"),
    ("trap21", LITERAL, "rocfft-literal"),     # the reference's literal (2n-1) grid and crop window
    ("gv33", 0, "pruned-hip"),
    ("gv33", FORCE_ROCFFT, "rocfft-reduced"),
    ("gv33", LITERAL, "rocfft-literal"),
    ("gv32", 0, "pruned-hip"),
    ("gv32", FORCE_ROCFFT, "rocfft-reduced"),
    ("gv32", LITERAL, "rocfft-literal"),
    ("gv128", 0, "pruned-hip"),
])
def test_fastm_2d_matches_oracle_and_golden(lsfc, name, flags, pipeline):
    c = cases.case_2d(name)
    Mo, b = c["M"], c["b"]
    M = lsfc.FastM(Mo.GFFT, Mo.nu, Mo.ne, Mo.me, Mo.n, Mo.m, Mo.omega, quadRule=Mo.quadRule, flags=flags)
    assert M.pipeline == pipeline
    g = np.load(os.path.join(GOLD, f"2d_{name}.npz"))
    y = M * b
    assert rel_err(y, o.fastconvolution(Mo, b)) < TOL
    assert rel_err(y, g["apply_random"]) < TOL
    assert rel_err(lsfc.FFTconvolution(M, b), g["conv_random"]) < TOL
    X, _ = o.grid2d(c["x"], c["x"])
    assert rel_err(lsfc.fastconvolution(M, cases.plane_wave(c["k"], X)), g["apply_planewave"]) < TOL
    # traits (src/FastConvolution.jl:31-41) and mul!
    N = Mo.n * Mo.m
    assert lsfc.size(M, 1) == N and lsfc.size(M) == ((N,), (N,)) and lsfc.eltype(M) == np.complex128
    Y = np.empty(N, complex)
    lsfc.mul_(Y, M, b)
    assert np.array_equal(Y, y)
    # run-to-run bitwise reproducibility (no float atomics anywhere in the apply)
    assert np.array_equal(M * b, y)


# ---------------------------------------------------------------------------- 3D, literal-symbol constructor
@pytest.mark.parametrize("name,flags", [("gv16", 0), ("gv16", FORCE_ROCFFT), ("gv16", LITERAL), ("gv16k10", 0), ("gv32", 0), ("gv32k10", 0)])
def test_fastm3d_matches_oracle_and_golden(lsfc, name, flags):
    c = cases.case_3d(name)
    Mo, b, n = c["M"], c["b"], c["n"]
    M = lsfc.FastM3D(Mo.GFFT, Mo.nu, Mo.ne, Mo.me, Mo.le, n, n, n, Mo.omega, flags=flags)
    assert M.pipeline == {0: "pruned-hip", FORCE_ROCFFT: "rocfft-reduced", LITERAL: "rocfft-literal"}[flags]
    g = np.load(os.path.join(GOLD, f"3d_{name}.npz"))
    y = M * b
    assert rel_err(y, g["apply_random"]) < TOL
    if "conv_random" in g:
        assert rel_err(lsfc.FFTconvolution(M, b), g["conv_random"]) < TOL
        assert rel_err(M * cases.plane_wave(c["k"], c["X"]), g["apply_planewave"]) < TOL
    assert np.array_equal(M * b, y)


def test_noncubic_3d_pruned(lsfc):
    # distinct n, m, l exercise every stride of the tiled intermediate layouts
    n, m, l = 16, 32, 64
    rng = np.random.default_rng(7)
    G = rng.standard_normal((4 * n, 4 * m, 4 * l)) + 1j * rng.standard_normal((4 * n, 4 * m, 4 * l))
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    k = 7.0
    G2 = o.reduce_symbol(G, (n, m, l))
    ref = o.apply_reduced(G2, nu, k, b, (n, m, l))
    for flags, pipe in [(0, "pruned-hip"), (FORCE_ROCFFT, "rocfft-reduced"), (LITERAL, "rocfft-literal")]:
        M = lsfc.FastM3D(G, nu, 4 * n, 4 * m, 4 * l, n, m, l, k, flags=flags)
        assert M.pipeline == pipe
        assert rel_err(M * b, ref) < TOL, pipe
        assert rel_err(lsfc.FFTconvolution(M, b), o.convolve_reduced(G2, b, (n, m, l))) < TOL, pipe


@pytest.mark.parametrize("dims", [(24, 20, 18), (48, 16, 30), (17, 33, 19), (23, 37, 45), (9, 9, 9)])
def test_sizes_that_are_not_powers_of_two_3d(lsfc, dims):
    # any grid size runs in the hand-written pipeline: lines are zero-extended in registers to half the padded length
    # L = smallest of 2^k, 3*2^k, 5*2^k that is >= 2n
    n, m, l = dims
    rng = np.random.default_rng(sum(dims))
    G = rng.standard_normal((4 * n, 4 * m, 4 * l)) + 1j * rng.standard_normal((4 * n, 4 * m, 4 * l))
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    G2 = o.reduce_symbol(G, (n, m, l))
    ref = o.apply_reduced(G2, nu, 5.0, b, (n, m, l))
    M = lsfc.FastM3D(G, nu, 4 * n, 4 * m, 4 * l, n, m, l, 5.0)
    Ls = [best_length(v) for v in dims]
    if np.prod(Ls) <= 4 * 8 * n * m * l:   # embedding costs at most 4x the points: hand-written pipeline
        assert M.pipeline == "pruned-hip" and M.padded_dims == tuple(Ls)
    else:                                  # (9, 9, 9): 32^3 / 18^3 = 5.6x -> rocFFT on the exact 2n grid
        assert dims == (9, 9, 9)
        assert M.pipeline == "rocfft-reduced" and M.padded_dims == tuple(2 * v for v in dims)
    assert rel_err(M * b, ref) < TOL
    assert rel_err(lsfc.FFTconvolution(M, b), o.convolve_reduced(G2, b, (n, m, l))) < TOL
    Mr = lsfc.FastM3D(G, nu, 4 * n, 4 * m, 4 * l, n, m, l, 5.0, flags=FORCE_ROCFFT)
    assert Mr.pipeline == "rocfft-reduced" and rel_err(Mr * b, ref) < TOL


def _random_shapes(seed, count, ndim, lo, hi):
    rng = np.random.default_rng(seed)
    return [tuple(int(v) for v in rng.integers(lo, hi + 1, size=ndim)) for _ in range(count)]


@pytest.mark.parametrize("dims", _random_shapes(2025, 16, 3, 1, 44) + [(1, 1, 1), (2, 1, 1), (1, 40, 1), (44, 1, 2), (5, 7, 3)])
def test_random_shapes_3d(lsfc, dims):
    # seeded random (n, m, l) in 1 .. 44, and degenerate ones: whatever pipeline the plan picks (hand-written lines zero-extended
    # in registers, or rocFFT on the exact 2n grid where the embedding would cost too much), apply and bare convolution against
    # the oracle on the reduced grid, host and device vectors, twice (plan state)
    import torch
    n, m, l = dims
    rng = np.random.default_rng(n * 10007 + m * 101 + l)
    G2 = rng.standard_normal((2 * n, 2 * m, 2 * l)) + 1j * rng.standard_normal((2 * n, 2 * m, 2 * l))
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = rng.standard_normal(n * m * l) + 1j * rng.standard_normal(n * m * l)
    M = lsfc.FastM3D(np.fft.fftshift(G2), nu, 2 * n, 2 * m, 2 * l, n, m, l, 3.0)
    ref = o.apply_reduced(G2, nu, 3.0, b, (n, m, l))
    y = M * b
    assert rel_err(y, ref) < TOL, M.pipeline
    assert np.array_equal(M * b, y)
    assert rel_err(lsfc.FFTconvolution(M, b), o.convolve_reduced(G2, b, (n, m, l))) < TOL
    xd = torch.from_numpy(b).cuda(); yd = torch.empty_like(xd)
    M.mul_(yd, xd)
    assert np.array_equal(yd.cpu().numpy(), y)
    M.close()


@pytest.mark.parametrize("dims", _random_shapes(7, 12, 2, 1, 300) + [(1, 1), (1, 257), (300, 1), (2, 2)])
def test_random_shapes_2d(lsfc, dims):
    n, m = dims
    rng = np.random.default_rng(n * 1009 + m)
    G2 = rng.standard_normal((2 * n, 2 * m)) + 1j * rng.standard_normal((2 * n, 2 * m))
    nu = rng.uniform(-0.3, 0.3, n * m)
    b = rng.standard_normal(n * m) + 1j * rng.standard_normal(n * m)
    M = lsfc.FastM(np.fft.fftshift(G2), nu, 2 * n, 2 * m, n, m, 2.0, quadRule="Greengard_Vico")
    y = M * b
    assert rel_err(y, o.apply_reduced(G2, nu, 2.0, b, (n, m))) < TOL, M.pipeline
    assert np.array_equal(M * b, y)
    M.close()


@pytest.mark.parametrize("dims", [(6, 16, 16), (4, 32, 32), (16, 6, 32)])
def test_builder_3d_tiny_axis_next_to_long_ones(lsfc, dims):
    # an axis of fewer than 8 points would need a working line (32) longer than the literal 4n lattice the slab-wise
    # Greengard-Vico generator samples: such plans fall back to rocFFT on the exact 2n grid instead of failing
    n, m, l = dims
    h = 1.0 / 16
    x, y, z = (-0.5 + h * np.arange(v) for v in dims)
    X, Y, Z = o.grid3d(x, y, z)
    k = 9.0
    M = lsfc.buildFastConvolution3D(x, y, z, X, Y, Z, h, k, o.gaussian_bump)
    assert M.pipeline == "rocfft-reduced" and M.padded_dims == (2 * n, 2 * m, 2 * l)
    Mo = o.build_fast_convolution3d(x, y, z, X, Y, Z, h, k, o.gaussian_bump)
    b = o.random_vector(n * m * l)
    # (the reference's FFTconvolution allocates (ne, ne, le), src/FastConvolution3D.jl:48, and cannot run a non-cubic
    # grid itself; its arithmetic on the reduced grid is the comparator)
    G2 = o.reduce_symbol(Mo.GFFT, dims)
    assert rel_err(M * b, o.apply_reduced(G2, Mo.nu, k, b, dims)) < TOL
    assert rel_err(lsfc.FFTconvolution(M, b), o.convolve_reduced(G2, b, dims)) < TOL


def test_reference_example_size_n48_builder(lsfc):
    # examples/example3D.jl uses n = 48 (h = 1/48, k = 48): builder on the device vs the oracle's literal builder
    n = 48
    x, h = cases.grid(n, False)
    k = 1.0 / h
    X, Y, Z = o.grid3d(x, x, x)
    M = lsfc.buildFastConvolution3D(x, x, x, X, Y, Z, h, k, o.gaussian_bump)
    assert M.pipeline == "pruned-hip" and M.padded_dims == (96, 96, 96)
    Mo = o.build_fast_convolution3d(x, x, x, X, Y, Z, h, k, o.gaussian_bump)
    b = o.random_vector(n ** 3)
    assert rel_err(M * b, o.mul(Mo, b)) < TOL


@pytest.mark.parametrize("tiled", ["0", "1", None])
@pytest.mark.parametrize("dims", [(256, 256), (512, 384), (200, 1000), (80, 1024), (24, 800)])
def test_2d_tiled_and_row_layouts(lsfc, dims, tiled, monkeypatch):
    # the 2D pipeline with the x'-expanded array in rows [m][Lx] or in tiles [Lx/8][m][8] (LSFC_2D_TILED; tiles are the
    # default from 2048-point lines on): same operator, checked against the oracle on an even (Green's-type) symbol.
    # (80, 1024), (24, 800): 2048-point y lines next to a short x axis -- Lx / 8 is not a multiple of 8, so the tiled form
    # (whole groups of 8 tiles) must fall back to natural rows also under the default policy (tiled = None)
    if tiled is None:
        monkeypatch.delenv("LSFC_2D_TILED", raising=False)
    else:
        monkeypatch.setenv("LSFC_2D_TILED", tiled)
    n, m = dims
    rng = np.random.default_rng(n + m)
    G2 = rng.standard_normal((2 * n, 2 * m)) + 1j * rng.standard_normal((2 * n, 2 * m))
    for ax in range(2):
        G2 = 0.5 * (G2 + np.roll(np.flip(G2, axis=ax), 1, axis=ax))
    nu = rng.uniform(-0.3, 0.3, n * m)
    b = o.random_vector(n * m)
    M = lsfc.FastM(np.fft.fftshift(G2), nu, 2 * n, 2 * m, n, m, 2.0, quadRule="Greengard_Vico")
    assert M.pipeline == "pruned-hip"
    # (the comparison runs on the working grid the plan chose: the kernel is resampled there, the operator is the same)
    assert rel_err(M * b, o.apply_reduced(G2, nu, 2.0, b, (n, m))) < TOL
    B = np.stack([b, 1j * b[::-1]])
    Y = lsfc.apply_batch(M, B, 0)
    assert rel_err(Y[1], o.apply_reduced(G2, nu, 2.0, B[1], (n, m))) < TOL


def test_noncubic_2d_pruned(lsfc):
    n, m = 64, 16
    rng = np.random.default_rng(8)
    G = rng.standard_normal((4 * n, 4 * m)) + 1j * rng.standard_normal((4 * n, 4 * m))
    nu = rng.uniform(-0.3, 0.3, n * m)
    b = o.random_vector(n * m)
    G2 = o.reduce_symbol(G, (n, m))
    M = lsfc.FastM(G, nu, 4 * n, 4 * m, n, m, 3.0, quadRule="Greengard_Vico")
    assert M.pipeline == "pruned-hip"
    assert rel_err(M * b, o.apply_reduced(G2, nu, 3.0, b, (n, m))) < TOL
    with pytest.raises(ValueError):           # the reference's FFTconvolution(::FastM) assumes n == m
        lsfc.FFTconvolution(M, b)


@pytest.mark.parametrize("n", [L // 2 for L in LENGTHS])
def test_every_line_length_2d(lsfc, n):
    # one case per hand-written factorisation (padded length 2n = 32 ... 2048: 2^k, 3*2^k, 5*2^k), random symbol
    rng = np.random.default_rng(100 + n)
    G2 = rng.standard_normal((2 * n, 2 * n)) + 1j * rng.standard_normal((2 * n, 2 * n))
    # hand the kernel a literal symbol whose reduction is G2: centred 2n grid (ne == 2n is allowed)
    G = np.fft.fftshift(G2)
    nu = rng.uniform(-0.3, 0.3, n * n)
    b = o.random_vector(n * n)
    M = lsfc.FastM(G, nu, 2 * n, 2 * n, n, n, 2.0, quadRule="Greengard_Vico")
    assert M.pipeline == "pruned-hip" and M.padded_dims[:2] == (2 * n, 2 * n)
    assert rel_err(M * b, o.apply_reduced(G2, nu, 2.0, b, (n, n))) < TOL


@pytest.mark.parametrize("axis", [0, 1, 2])
@pytest.mark.parametrize("L", [v for v in LENGTHS if v & (v - 1)] + [2048])
def test_every_mixed_radix_line_length_on_every_axis_3d(lsfc, L, axis):
    # the lines with a factor 3 or 5 in each of the three kinds of pass (contiguous x, strided y, fused z); the other
    # two axes are short so that the 2n-grid oracle stays small
    dims = [16, 16, 16]
    dims[axis] = L // 2
    n, m, l = dims
    rng = np.random.default_rng(L + axis)
    G2 = rng.standard_normal((2 * n, 2 * m, 2 * l)) + 1j * rng.standard_normal((2 * n, 2 * m, 2 * l))
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    M = lsfc.FastM3D(np.fft.fftshift(G2), nu, 2 * n, 2 * m, 2 * l, n, m, l, 2.0)
    # (the longest line, 2048, runs the fused z pass of the 3D layout in 4-line half tiles)
    assert M.pipeline == "pruned-hip"
    assert M.padded_dims == (2 * n, 2 * m, 2 * l)
    assert rel_err(M * b, o.apply_reduced(G2, nu, 2.0, b, (n, m, l))) < TOL


@pytest.mark.parametrize("short", [0, 6])
@pytest.mark.parametrize("L", [128, 192, 320, 384, 512, 640, 1024, 1280, 1536, 2048])
def test_fused_pass_forms_even_symbol_long_z_lines(lsfc, L, short):
    # every form of the fused z pass on the long lines, against the oracle: one tile per workgroup (0), persistent whole
    # tiles (3), ticketed half tiles with per-XCD queues (5; the default at L = 1280 and 1536), ticketed whole tiles (6; the default at 1024).  They need the z-even half symbol,
    # i.e. a symbol that is even in every axis (as the Green's symbols are): a random one is symmetrised.
    # short > 0: the z axis holds fewer points than half the line (the kernels' guarded loads and stores)
    n, m, l = 16, 16, L // 2 - short
    rng = np.random.default_rng(L)
    G2 = rng.standard_normal((2 * n, 2 * m, L)) + 1j * rng.standard_normal((2 * n, 2 * m, L))
    for ax in range(3):
        G2 = 0.5 * (G2 + np.roll(np.flip(G2, axis=ax), 1, axis=ax))
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    M = lsfc.FastM3D(np.fft.fftshift(G2), nu, 2 * n, 2 * m, L, n, m, l, 2.0)
    assert M.pipeline == "pruned-hip" and M.padded_dims == (2 * n, 2 * m, L)
    want = o.apply_reduced(G2, nu, 2.0, b, (n, m, l))
    got = {}
    for form in (0, 3, 5, 6):
        # xlane: the exchanges between stages of equal radix (8.8 of the 512-, 1024- and 1536-point lines, 4.4 of 128, 192, 320,
        # 384, 640, and both 4.4 of 1280 = 20.4.4.4 in its half-tile form) through the lanes of the wavefront or through LDS
        for xl in (5, 3, 1, 0):
            M.set_tuning(z_persist=form, xlane=xl)
            got[form] = M * b
            assert rel_err(got[form], want) < TOL, (form, xl)
    # twice in a row through the ticket counters (a fresh set per launch)
    M.set_tuning(z_persist=5)
    assert np.array_equal(M * b, got[5])


def test_mixed_radix_even_symbol_3d(lsfc):
    # an even (Green's-type) symbol on a 3 x 5 x 2^k grid: exercises the y-even / z-even half-symbol storage with the
    # mixed-radix storage orders (builder on the device vs the slab-wise oracle symbol)
    n, m, l = 24, 40, 48
    h = 1.0 / n
    x, y, z = (-0.5 + h * np.arange(v) for v in (n, m, l))
    k = 9.0
    X, Y, Z = o.grid3d(x, y, z)
    nuv = o.gaussian_bump(X, Y, Z)
    M = lsfc.buildFastConvolution3D(x, y, z, X, Y, Z, h, k, nuv)
    assert M.pipeline == "pruned-hip" and M.padded_dims == (48, 80, 96)
    # (the literal FFTconvolution of the reference allocates zeros(ne, ne, le), src/FastConvolution3D.jl:48, so its
    # non-cubic form is checked through the reduced-grid identity)
    G2 = o.reduced_symbol_gv3d(n, m, l, n * h, k, patch_singular=False)
    b = o.random_vector(n * m * l)
    assert rel_err(M * b, o.apply_reduced(G2, nuv, k, b, (n, m, l))) < TOL


# ---------------------------------------------------------------------------- builders (symbol generated on the device)
@pytest.mark.parametrize("dims", [(64, 64, 64), (24, 40, 48), (16, 96, 20)])
def test_builder_3d_through_symmetry_equals_full_builder(lsfc, dims, monkeypatch):
    # the default 3D builder evaluates the radial symbol on the non-negative frequencies only and mirrors half-axes
    # (symbol_gv3d_quarter); LSFC_SYMBOL_FULL=1 selects the literal-grid builder of round 1.  Same operator to rounding,
    # and both against the oracle.
    n, m, l = dims
    h = 1.0 / n
    x, y, z = (-0.5 + h * np.arange(v) for v in (n, m, l))
    k = 11.0
    X, Y, Z = o.grid3d(x, y, z)
    nuv = o.gaussian_bump(X, Y, Z)
    b = o.random_vector(n * m * l)
    M = lsfc.buildFastConvolution3D(x, y, z, X, Y, Z, h, k, nuv)
    got = M * b
    monkeypatch.setenv("LSFC_SYMBOL_FULL", "1")
    Mf = lsfc.buildFastConvolution3D(x, y, z, X, Y, Z, h, k, nuv)
    full = Mf * b
    monkeypatch.delenv("LSFC_SYMBOL_FULL")
    assert M.pipeline == Mf.pipeline == "pruned-hip" and M.padded_dims == Mf.padded_dims
    assert rel_err(got, full) < 1e-13
    G2 = o.reduced_symbol_gv3d(n, m, l, n * h, k, patch_singular=False)
    assert rel_err(got, o.apply_reduced(G2, nuv, k, b, (n, m, l))) < TOL


@pytest.mark.parametrize("name", ["trap21", "gv33", "gv32", "gv128"])
def test_build_fast_convolution_2d(lsfc, name):
    c = cases.case_2d(name)
    M = lsfc.buildFastConvolution(c["x"], c["x"], c["h"], c["k"], c["nu"], quadRule=c["quadRule"])
    g = np.load(os.path.join(GOLD, f"2d_{name}.npz"))
    assert rel_err(M * c["b"], g["apply_random"]) < TOL
    assert rel_err(lsfc.FFTconvolution(M, c["b"]), g["conv_random"]) < TOL
    assert (M.ne, M.n) == (c["M"].ne, c["M"].n)


@pytest.mark.parametrize("name", ["gv16", "gv16k10", "gv32", "gv32k10"])
def test_build_fast_convolution_3d(lsfc, name):
    c = cases.case_3d(name)
    M = lsfc.buildFastConvolution3D(c["x"], c["x"], c["x"], c["X"], c["Y"], c["Z"], c["h"], c["k"], c["nu"])
    assert M.pipeline == "pruned-hip"
    g = np.load(os.path.join(GOLD, f"3d_{name}.npz"))
    assert rel_err(M * c["b"], g["apply_random"]) < TOL


def test_builder_3d_n128_against_slabwise_oracle(lsfc):
    n = 128
    x, h = cases.grid(n, False)
    k = 1.0 / h
    X, Y, Z = o.grid3d(x, x, x)
    nuv = cases.nu_synthetic(3)(X, Y, Z)
    M = lsfc.buildFastConvolution3D(x, x, x, X, Y, Z, h, k, nuv)
    G2 = o.reduced_symbol_gv3d(n, n, n, 1.0, k, patch_singular=False)
    b = o.random_vector(n ** 3)
    assert rel_err(M * b, o.apply_reduced(G2, nuv, k, b, (n, n, n))) < TOL


def test_singular_omega_patch(lsfc):
    # omega = 8 pi on the half-open unit box with n = 16: lattice points sit exactly on |s| = k.
    # Unpatched the reference (and this build) produce non-finite output; PATCH uses the analytic limit.
    n = 16
    x, h = cases.grid(n, False)
    k = 8 * np.pi
    X, Y, Z = o.grid3d(x, x, x)
    b = o.random_vector(n ** 3)
    Mo = o.build_fast_convolution3d(x, x, x, X, Y, Z, h, k, o.gaussian_bump, patch_singular=True)
    assert np.isfinite(Mo.GFFT).all()
    with np.errstate(all="ignore"):
        assert not np.isfinite(o.build_fast_convolution3d(x, x, x, X, Y, Z, h, k, o.gaussian_bump).GFFT).all()
    M = lsfc.buildFastConvolution3D(x, x, x, X, Y, Z, h, k, o.gaussian_bump, flags=PATCH)
    assert rel_err(M * b, o.mul(Mo, b)) < TOL
    Mbad = lsfc.buildFastConvolution3D(x, x, x, X, Y, Z, h, k, o.gaussian_bump)
    assert not np.isfinite(Mbad * b).all()


# ---------------------------------------------------------------------------- properties and callers
def test_analytic_gaussian_known_answer(lsfc):
    n, k, sig = 64, 10.0, 0.05
    x, h = cases.grid(n, False)
    X, Y, Z = o.grid3d(x, x, x)
    M = lsfc.buildFastConvolution3D(x, x, x, X, Y, Z, h, k, o.gaussian_bump)
    f = (np.exp(-(X**2 + Y**2 + Z**2) / (2 * sig**2)) / ((2 * np.pi) ** 1.5 * sig**3)).astype(complex)
    with np.errstate(all="ignore"):
        ref = -o.sol_ref_helmholtz(X, Y, Z, sig, k)
    ok = np.isfinite(ref)
    assert rel_err(lsfc.FFTconvolution(M, f)[ok], ref[ok]) < 1e-10


def test_analytic_gaussian_2d_by_quadrature(lsfc):
    # the device 2D Greengard-Vico builder + FFTconvolution against the continuous operator itself (adaptive quadrature of
    # (i/4) H0(k r) * Gaussian at three points): no oracle, no FFT in the comparator
    n, k, sig = 128, 20.0, 0.06
    x, h = cases.grid(n, True)
    M = lsfc.buildFastConvolution(x, x, h, k, o.gaussian_bump, quadRule="Greengard_Vico")
    X, Y = o.grid2d(x, x)
    f = np.exp(-(X ** 2 + Y ** 2) / (2 * sig ** 2)) / (2 * np.pi * sig ** 2)
    u = lsfc.FFTconvolution(M, f.astype(complex)).reshape((n, n), order="F")
    for i, j, exact in cases.gaussian_2d_quadrature_points(x, k, sig, [(64, 64), (74, 57), (20, 90)]):
        assert abs(u[i, j] - exact) / abs(exact) < 1e-10, (i, j)


def test_sample_g3d_delta_sources(lsfc):
    c = cases.case_3d("gv16")
    Mo, n = c["M"], c["n"]
    M = lsfc.FastM3D(Mo.GFFT, Mo.nu, Mo.ne, Mo.me, Mo.le, n, n, n, Mo.omega)
    indS = [0, 17, n**3 - 1, 5 + n * (6 + n * 7)]
    assert rel_err(lsfc.sampleG3D(Mo.omega, None, None, None, indS, M), o.sample_g_conv(indS, Mo)) < TOL


def test_sample_sources_gather_2d_and_noncubic(lsfc):
    # GV 2D (sampleGConv) and a non-cubic 3D grid: gather path == one FFT convolution per source
    c = cases.case_2d("gv32")
    M = lsfc.buildFastConvolution(c["x"], c["x"], c["h"], c["k"], c["nu"], quadRule="Greengard_Vico")
    ind = [0, 5 + 32 * 7, 31, 32 * 32 - 1]
    assert rel_err(lsfc.sampleGConv(c["k"], None, None, ind, M), o.sample_g_conv(ind, c["M"])) < TOL
    n, m, l = 16, 32, 64
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    M3 = lsfc.buildFastConvolution3D(x, x[:1].repeat(m), x[:1].repeat(l), None, None, None, h, 6.0, np.zeros(n * m * l))
    ind3 = [0, 3 + n * (17 + m * 40), n * m * l - 1]
    rows = lsfc.sampleG3D(6.0, None, None, None, ind3, M3)
    for r, j in zip(rows, ind3):
        e = np.zeros(n * m * l, complex); e[j] = 1
        assert rel_err(r, lsfc.FFTconvolution(M3, e)) < 1e-12
    # trapezoidal keeps the reference's nu quirk (one convolution per source)
    ct = cases.case_2d("trap21")
    Mt = lsfc.buildFastConvolution(ct["x"], ct["x"], ct["h"], ct["k"], ct["nu"], quadRule="trapezoidal")
    assert rel_err(lsfc.sampleGConv(ct["k"], None, None, [7, 100], Mt), o.sample_g_conv([7, 100], ct["M"])) < TOL


def test_set_nu_and_aliasing(lsfc):
    c = cases.case_3d("gv16")
    Mo, b, n = c["M"], c["b"], c["n"]
    M = lsfc.FastM3D(Mo.GFFT, Mo.nu, Mo.ne, Mo.me, Mo.le, n, n, n, Mo.omega)
    nu2 = cases.nu_synthetic(3)(c["X"], c["Y"], c["Z"])
    M.set_nu(nu2)
    Mo.nu = nu2
    ref = o.mul(Mo, b)
    assert rel_err(M * b, ref) < TOL
    y = b.copy()
    M.mul_(y, y)                       # y may alias x
    assert rel_err(y, ref) < TOL


@pytest.mark.parametrize("K", ["0", "2", "8", "5"])
def test_host_vector_pipeline(lsfc, K, monkeypatch):
    # mul!(Y, M, b) with HOST vectors (src/FastConvolution.jl:50-54) as the chunked pipeline (x chunks upload under the x / y
    # passes, y chunks download under the inverse passes; default for vectors >= 64 MB, forced here on a small grid):
    # same bits as the one-copy-each-way form and the device-resident apply, also in place (Y aliases b), with page-locked
    # vectors, and for the bare convolution
    import torch
    n, m, l = 24, 20, 40                       # l = 40: 2, 5 and 8 chunks of planes (8 -> the largest divisor below: 8 | 40)
    rng = np.random.default_rng(5)
    G2 = rng.standard_normal((2 * n, 2 * m, 2 * l)) + 1j * rng.standard_normal((2 * n, 2 * m, 2 * l))
    nu = rng.uniform(-0.3, 0.3, n * m * l)
    b = o.random_vector(n * m * l)
    M = lsfc.FastM3D(np.fft.fftshift(G2), nu, 2 * n, 2 * m, 2 * l, n, m, l, 2.0)
    assert M.pipeline == "pruned-hip"
    want = o.apply_reduced(G2, nu, 2.0, b, (n, m, l))
    monkeypatch.setenv("LSFC_HOST_PIPELINE", K)
    y = M * b
    assert rel_err(y, want) < TOL
    xd = torch.from_numpy(b).cuda(); yd = torch.empty_like(xd)
    M.mul_(yd, xd)
    assert np.array_equal(yd.cpu().numpy(), y)
    z = b.copy()
    M.mul_(z, z)                               # in place on the host
    assert np.array_equal(z, y)
    # page-locked vectors owned by the runtime (lsfc_host_alloc): moved by DMA, no staging copies
    bp, yp = lsfc.host_empty(b.shape), lsfc.host_empty(b.shape)
    bp[:] = b
    M.mul_(yp, bp)
    assert np.array_equal(yp, y)
    M.mul_(bp, bp)                             # in place in page-locked memory
    assert np.array_equal(bp, y)
    assert rel_err(lsfc.FFTconvolution(M, b), o.convolve_reduced(G2, b, (n, m, l))) < TOL
    # several right-hand sides: two staging slots, the download of one overlapping the upload of the next
    B = np.stack([b, 1j * b[::-1], b.conj(), 0.5 * b, b[::-1]])
    Y = lsfc.apply_batch(M, B, 0)
    for r in range(B.shape[0]):
        assert np.array_equal(Y[r], M * B[r]), r
    Bp = lsfc.host_empty(B.shape)
    Bp[:] = B
    assert np.array_equal(lsfc.apply_batch(M, Bp, 0), Y)
    del Bp, bp, yp


def test_host_register_round_trip(lsfc):
    # lsfc_host_register / lsfc_host_unregister on a page-aligned anonymous mapping of whole pages (the form to prefer when the
    # caller's memory has to be pinned in place); the vectors then move by DMA.  Registering ranges of the process HEAP that share
    # pages with other objects is left out of the default suite on purpose: one session of round 3 ended in a GPU memory fault on a
    # page-aligned heap address while a registered numpy array next to a pageable one was being read (host vectors);
    # LSFC_TEST_HOST_REGISTER_HEAP=1 runs that form too.
    import mmap
    n = 16
    rng = np.random.default_rng(11)
    G2 = rng.standard_normal((2 * n, 2 * n, 2 * n)) + 1j * rng.standard_normal((2 * n, 2 * n, 2 * n))
    nu = rng.uniform(-0.3, 0.3, n ** 3)
    b = o.random_vector(n ** 3)
    M = lsfc.FastM3D(np.fft.fftshift(G2), nu, 2 * n, 2 * n, 2 * n, n, n, n, 2.0)
    y = M * b
    nbytes = -(-b.nbytes // mmap.PAGESIZE) * mmap.PAGESIZE
    mx, my = mmap.mmap(-1, nbytes), mmap.mmap(-1, nbytes)
    bx = np.frombuffer(mx, dtype=np.complex128, count=b.size); by = np.frombuffer(my, dtype=np.complex128, count=b.size)
    bx[:] = b; by[:] = 0
    import ctypes
    L = lsfc.load()
    px, py = bx.ctypes.data, by.ctypes.data
    assert px % mmap.PAGESIZE == 0 and py % mmap.PAGESIZE == 0
    dma = os.environ.get("LSFC_TEST_HOST_REGISTER_DMA") == "1" or os.environ.get("LSFC_TEST_HOST_REGISTER_HEAP") == "1"
    assert L.lsfc_host_register(ctypes.c_void_p(px), nbytes) == 0 and L.lsfc_host_register(ctypes.c_void_p(py), nbytes) == 0
    try:
        # the apply that reads / writes the pinned-in-place pages by DMA runs on request only (LSFC_TEST_HOST_REGISTER_DMA=1; it passed in
        # every session that ran it): the default suite, which the round-end driver runs once and unattended, checks the calls and their
        # errors and leaves device access to user pages pinned in place to the tools (tools/bench_host_path.py does it at 512^3)
        if dma:
            M.mul_(by, bx)
    finally:
        assert L.lsfc_host_unregister(ctypes.c_void_p(px)) == 0 and L.lsfc_host_unregister(ctypes.c_void_p(py)) == 0
    if dma:
        assert np.array_equal(by, y)
    assert L.lsfc_host_unregister(ctypes.c_void_p(px)) != 0            # not registered any more: an error, not a crash
    M.mul_(by, bx)                                                     # the same (now pageable) vectors: staged copies
    assert np.array_equal(by, y)
    if os.environ.get("LSFC_TEST_HOST_REGISTER_HEAP") == "1":
        hb, hy = b.copy(), np.empty_like(b)
        lsfc.host_register(hb); lsfc.host_register(hy)
        try:
            M.mul_(hy, hb)
        finally:
            lsfc.host_unregister(hb); lsfc.host_unregister(hy)
        assert np.array_equal(hy, y)
    del bx, by
    mx.close(); my.close()


def test_first_apply_of_a_fresh_process(lsfc):
    # The first host-vector apply of a PROCESS is where round 2's intermittent GPU fault sat: a staged
    # asynchronous copy in flight while the first launch loaded a code object.  The library now loads every code object at the
    # first plan creation.  One child process, one plan (the 2D trapezoidal n = 21 case of that fault), one apply; run once.
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("LSFC_EAGER_LOAD", "LSFC_HOST_COPY", "AMD_LOG_LEVEL"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "first_apply_trace.py"), "test"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "plan: pruned-hip (48, 48, 1)" in r.stdout and "finite: True repeatable: True" in r.stdout


def test_device_resident_vectors_torch(lsfc):
    import torch
    c = cases.case_3d("gv16")
    Mo, b, n = c["M"], c["b"], c["n"]
    M = lsfc.FastM3D(Mo.GFFT, Mo.nu, Mo.ne, Mo.me, Mo.le, n, n, n, Mo.omega)
    xb = torch.from_numpy(b).cuda()
    y = M * xb
    assert y.is_cuda
    M.synchronize()
    assert rel_err(y.cpu().numpy(), o.mul(Mo, b)) < TOL
    ms = lsfc.time_apply(M, xb, y, 3)
    assert ms > 0
    prof = lsfc.profile_apply(M, xb, y, 2)
    assert [p[0] for p in prof] == ["xfwd", "yfwd", "zfused", "yinv", "xinv"]


# ---------------------------------------------------------------------------- error behaviour and edge cases through the C ABI
def test_error_paths_and_edge_cases(lsfc):
    import ctypes as C
    from fast_solver_lippmann_schwinger_amd import _lib as L
    lib = lsfc.load()
    c = cases.case_3d("gv16")
    Mo, b, n = c["M"], c["b"], c["n"]
    M = lsfc.FastM3D(Mo.GFFT, Mo.nu, Mo.ne, Mo.me, Mo.le, n, n, n, Mo.omega)
    # DimensionMismatch on the vector length (the reference throws from reshape / broadcasting)
    with pytest.raises(ValueError):
        M * b[:-1]
    with pytest.raises(ValueError):
        lsfc.FastM3D(Mo.GFFT, Mo.nu[:-1], Mo.ne, Mo.me, Mo.le, n, n, n, Mo.omega)
    # C ABI: NULL arguments, bad quadRule, too-small padded grid, odd n in the 3D builder, even n in the trapezoidal builder
    plan = C.c_void_p()
    nu = np.zeros(n ** 3)
    assert lib.lsfc_apply(None, None, None, 0) == -1 and b"NULL" in lib.lsfc_last_error()
    G = np.asfortranarray(Mo.GFFT)
    assert lib.lsfc_plan_create_3d(C.byref(plan), n, n, n, 4 * n, 4 * n, 4 * n, nu.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), 1.0, 7, 0, 0) == -1
    assert b"quadRule" in lib.lsfc_last_error()
    assert lib.lsfc_plan_create_3d(C.byref(plan), n, n, n, n, n, n, nu.ctypes.data_as(C.c_void_p), G.ctypes.data_as(C.c_void_p), 1.0, 1, 0, 0) == -1
    assert lib.lsfc_plan_create_gv3d(C.byref(plan), 15, 16, 16, 1.0, 5.0, nu.ctypes.data_as(C.c_void_p), 0, 0) == -1
    assert lib.lsfc_plan_create_trap2d(C.byref(plan), 20, 21, -0.5, -0.5, 0.05, 20.0, 1.0, -0.892, nu.ctypes.data_as(C.c_void_p), 0, 0) == -1
    assert b"n odd" in lib.lsfc_last_error()
    assert lib.lsfc_plan_create_gv3d(C.byref(plan), 16, 16, 16, 1.0, 5.0, nu.ctypes.data_as(C.c_void_p), 0, 99) == -1     # no such device
    assert lib.lsfc_plan_destroy(None) == 0
    # unknown tuning key; sample index out of range
    assert lib.lsfc_plan_set_tuning(M._plan, b"nonsense", 1) == -1
    src = np.array([n ** 3], dtype=np.int64); out = np.empty(n ** 3, complex)
    assert lib.lsfc_sample_sources(M._plan, src.ctypes.data_as(C.c_void_p), 1, out.ctypes.data_as(C.c_void_p), 0) == -1
    # batch entry (one fused pass for the three vectors) vs single applies; zero vector maps to zero
    X = np.stack([b, 2j * b, np.zeros_like(b)])
    Y = np.empty_like(X)
    L.check(lib.lsfc_apply_batch(M._plan, X.ctypes.data_as(C.c_void_p), Y.ctypes.data_as(C.c_void_p), 3, 0, 0))
    y = M * b
    assert rel_err(Y[0], y) < 1e-14 and rel_err(Y[1], 2j * y) < 1e-14 and not Y[2].any()
    # n = 8 (padded length 16) is below the hand-written range: the rocFFT pipeline takes over, same results
    n8 = 8
    rng = np.random.default_rng(2)
    G8 = rng.standard_normal((4 * n8,) * 3) + 1j * rng.standard_normal((4 * n8,) * 3)
    nu8 = rng.uniform(-0.3, 0.3, n8 ** 3); b8 = o.random_vector(n8 ** 3)
    M8 = lsfc.FastM3D(G8, nu8, 4 * n8, 4 * n8, 4 * n8, n8, n8, n8, 2.0)
    assert M8.pipeline == "rocfft-reduced"
    assert rel_err(M8 * b8, o.apply_reduced(o.reduce_symbol(G8, (n8,) * 3), nu8, 2.0, b8, (n8,) * 3)) < TOL


def test_c_api_example_compiles_and_runs(lsfc, tmp_path):
    # the boundary is usable from plain C: examples/c_api_example.c built with gcc against liblsfc.so.  The child process
    # loads the system ROCm stack only (no torch): single-GPU solve, multi-device plan, slab plan over RCCL of /opt/rocm
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "fast_solver_lippmann_schwinger_amd")
    exe = str(tmp_path / "c_api_example")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"), os.path.join(root, "examples", "c_api_example.c"),
                        "-L", libdir, "-llsfc", "-lm", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([exe, "32"], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"converged=1" in r.stdout and b"multi-device plan on" in r.stdout and b"RCCL send/recv self exchange" in r.stdout
