"""GPU tests at the BASELINE headline size (3D n = 512, 1024^3 padded grid), where no CPU oracle fits in
the test budget: size-independent properties of the operator, checked on device-resident vectors.
  * analytic known answer: FFTconvolution(unit-mass Gaussian) == -solRefHelmholtz (src/Functions.jl:32-36),
    evaluated point-wise on sampled z-planes (exercises the slab-wise symbol generator and every kernel at full size)
  * linearity of the full operator
  * delta response: a unit source gives the same (shifted) kernel wherever it sits (what sampleG3D relies on,
    src/FastConvolution3D.jl:136-160), and the kernel is even
  * y may alias x; run-to-run bitwise reproducibility"""
import numpy as np
import pytest

from oracle import lsfc_oracle as o

pytestmark = pytest.mark.gpu

N512 = 512


@pytest.fixture(scope="module")
def op512(lsfc):
    n = N512
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    rng = np.random.default_rng(1)
    nu = rng.uniform(-0.3, 0.3, n ** 3)
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 10.0, nu)
    assert M.pipeline == "pruned-hip" and M.padded_dims == (1024, 1024, 1024)
    yield M, x, h
    M.close()


def test_analytic_gaussian_n512(lsfc, op512):
    import torch
    M, x, h = op512
    n, k, sig = N512, 10.0, 0.05
    X2, Y2 = np.meshgrid(x, x, indexing="ij")
    planes = [0, 137, 255, 256, 300, 511]
    f = torch.empty(n ** 3, dtype=torch.complex128, device="cuda")
    fv = f.view(n, n, n)                                   # [z][y][x]
    gx = torch.from_numpy(np.exp(-x ** 2 / (2 * sig ** 2))).cuda()
    fv[:] = (gx[:, None, None] * gx[None, :, None] * gx[None, None, :]) / ((2 * np.pi) ** 1.5 * sig ** 3)
    u = lsfc.FFTconvolution(M, f)
    M.synchronize()
    uv = u.view(n, n, n)
    worst = 0.0
    for kz in planes:
        got = uv[kz].cpu().numpy().T                       # [x][y]
        with np.errstate(all="ignore"):
            ref = -o.sol_ref_helmholtz(X2, Y2, np.full_like(X2, x[kz]), sig, k)
        ok = np.isfinite(ref)
        worst = max(worst, float(np.linalg.norm((got - ref)[ok]) / np.linalg.norm(ref[ok])))
    assert worst < 1e-10, worst


def test_linearity_aliasing_reproducibility_n512(lsfc, op512):
    import torch
    M, _, _ = op512
    N = N512 ** 3
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    x1 = torch.randn(N, dtype=torch.complex128, device="cuda", generator=g)
    x2 = torch.randn(N, dtype=torch.complex128, device="cuda", generator=g)
    a, b = 0.7 - 0.2j, -1.3 + 0.5j
    y1, y2 = M * x1, M * x2
    y12 = M * (a * x1 + b * x2)
    err = float(torch.linalg.norm(y12 - (a * y1 + b * y2)) / torch.linalg.norm(y12))
    assert err < 1e-13, err
    assert torch.equal(M * x1, y1)                          # bitwise reproducible
    z = x1.clone()
    M.mul_(z, z)                                            # y aliases x
    assert torch.equal(z, y1)


def test_delta_response_shift_invariance_n512(lsfc, op512):
    import torch
    M, _, _ = op512
    n = N512
    N = n ** 3

    def response(src):
        e = torch.zeros(N, dtype=torch.complex128, device="cuda")
        e[src[0] + n * (src[1] + n * src[2])] = 1.0
        return lsfc.FFTconvolution(M, e).view(n, n, n)     # [z][y][x]

    a, b = (100, 200, 300), (140, 170, 260)
    ra, rb = response(a), response(b)
    d = tuple(bi - ai for ai, bi in zip(a, b))             # rb[p + d] == ra[p] on the overlap
    w = 180
    sa = ra[a[2] - 90:a[2] - 90 + w, a[1] - 90:a[1] - 90 + w, a[0] - 90:a[0] - 90 + w]
    sb = rb[b[2] - 90:b[2] - 90 + w, b[1] - 90:b[1] - 90 + w, b[0] - 90:b[0] - 90 + w]
    assert float(torch.linalg.norm(sa - sb) / torch.linalg.norm(sa)) < 1e-12
    # the kernel is even: response at src + p equals response at src - p
    c = ra[a[2] - 50:a[2] + 51, a[1] - 50:a[1] + 51, a[0] - 50:a[0] + 51]
    assert float(torch.linalg.norm(c - torch.flip(c, dims=(0, 1, 2))) / torch.linalg.norm(c)) < 1e-12


@pytest.mark.parametrize("n", [320, 384])
def test_analytic_gaussian_mixed_radix_cubes(lsfc, n):
    # the same known answer on large cubes whose working grid is a mixed-radix one (640 = 5*2^7, 768 = 3*2^8): symbol
    # generator, mixed-radix passes and half symbols at full size.  (Cubes only: the reference's builder takes the
    # frequency lattice of every axis from the x extent, src/FastConvolution3D.jl:72-81, which is the physical kernel
    # only when the three axes have the same length.)
    import torch
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    k, sig = 12.0, 0.05
    M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, k, np.zeros(n ** 3))
    assert M.pipeline == "pruned-hip" and M.padded_dims == (2 * n, 2 * n, 2 * n)
    gx = torch.from_numpy(np.exp(-x ** 2 / (2 * sig ** 2))).cuda()
    f = ((gx[:, None, None] * gx[None, :, None] * gx[None, None, :]) / ((2 * np.pi) ** 1.5 * sig ** 3)).to(torch.complex128).reshape(-1)
    u = lsfc.FFTconvolution(M, f)
    M.synchronize()
    uv = u.view(n, n, n)                                    # [z][y][x]
    X2, Y2 = np.meshgrid(x, x, indexing="ij")
    worst = 0.0
    for kz in [0, 77, n // 2, n - 1]:
        got = uv[kz].cpu().numpy().T                        # [x][y]
        with np.errstate(all="ignore"):
            ref = -o.sol_ref_helmholtz(X2, Y2, np.full_like(X2, x[kz]), sig, k)
        ok = np.isfinite(ref)
        worst = max(worst, float(np.linalg.norm((got - ref)[ok]) / np.linalg.norm(ref[ok])))
    M.close()
    assert worst < 1e-10, worst
