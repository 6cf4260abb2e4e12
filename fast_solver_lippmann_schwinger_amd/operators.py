"""Host-side mirror of the reference's operator surface for the fast-convolution
path, over the lsfc C ABI (include/lsfc.h).

Names, argument meaning and error behaviour follow the Julia reference
(paths relative to the reference root):

  FastM, FastM3D                      src/FastConvolution.jl:11-27, src/FastConvolution3D.jl:7-26
  M * b, mul_(Y, M, b) (= mul!)       src/FastConvolution.jl:43-54, src/FastConvolution3D.jl:31-37
  size, eltype                        src/FastConvolution.jl:31-41
  fastconvolution, FFTconvolution     src/FastConvolution.jl:58-154, src/FastConvolution3D.jl:39-63
  buildFastConvolution[3D]            src/FastConvolution.jl:170-236, src/FastConvolution3D.jl:68-101
  referenceValsTrapRule               src/FastConvolution.jl:407-415
  sampleGConv / sampleG3D             src/FastConvolution.jl:278-306, src/FastConvolution3D.jl:136-160
  gmres_ (= gmres!)                   IterativeSolvers.jl, call sites examples/example.jl:85,91

Vectors are flat complex128 arrays in column-major (x fastest) order: numpy
arrays (host memory, copied through PCIe inside the call) or torch CUDA tensors
(device memory, zero-copy).  All arithmetic runs in HIP kernels; nothing here
computes on the CPU.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _is_torch(a):
    return type(a).__module__.split(".")[0] == "torch"


def _vec(a, N, name, count=1, plan=None):
    """-> (pointer, memspace, keepalive).  For CUDA tensors the plan is moved onto torch's current stream, so the
    library's kernels are ordered with the caller's torch work (stream 0 == the legacy default stream)."""
    if _is_torch(a):
        import torch
        if a.dtype != torch.complex128 or not a.is_contiguous() or a.numel() != N * count:
            raise TypeError(f"{name}: need a contiguous complex128 tensor with {N * count} entries")
        if a.is_cuda and plan is not None:
            L.check(L.load().lsfc_plan_set_stream(plan, C.c_void_p(torch.cuda.current_stream(a.device).cuda_stream)))
        return C.c_void_p(a.data_ptr()), (L.LSFC_MEM_DEVICE if a.is_cuda else L.LSFC_MEM_HOST), a
    arr = np.ascontiguousarray(a, dtype=np.complex128)
    if arr.size != N * count:
        raise ValueError(f"DimensionMismatch: {name} has {arr.size} entries, operator needs {N * count}")
    return arr.ctypes.data_as(C.c_void_p), L.LSFC_MEM_HOST, arr


_QUAD = {"trapezoidal": L.LSFC_QUAD_TRAPEZOIDAL, "Greengard_Vico": L.LSFC_QUAD_GREENGARD_VICO}


class _Operator:
    """Common behaviour of FastM / FastM3D: owns one lsfc plan."""
    _plan = None

    # -- traits: src/FastConvolution.jl:31-41 --------------------------------
    def size(self, dim=None):
        N = int(self.nu.shape[0])
        return N if dim is not None else ((N,), (N,))      # tuple-of-tuples quirk kept

    @property
    def shape(self):
        N = int(self.nu.shape[0])
        return (N, N)

    def eltype(self):
        return np.complex128

    dtype = np.complex128

    @property
    def N(self):
        return int(L.load().lsfc_plan_size(self._plan))

    @property
    def pipeline(self):
        return L.load().lsfc_plan_pipeline(self._plan).decode()

    @property
    def padded_dims(self):
        d, p = (C.c_int64 * 3)(), (C.c_int64 * 3)()
        L.check(L.load().lsfc_plan_dims(self._plan, d, p))
        return tuple(p)

    def _out_like(self, b, count=1):
        if _is_torch(b):
            import torch
            return torch.empty(self.N * count, dtype=torch.complex128, device=b.device)
        return np.empty(self.N * count, dtype=np.complex128)

    # -- the apply -------------------------------------------------------------
    def __mul__(self, b):
        return fastconvolution(self, b)

    __matmul__ = __mul__
    matvec = __mul__

    def mul_(self, Y, b):
        """LinearAlgebra.mul!(Y, M, b): Y[:] = M*b  (src/FastConvolution.jl:50-54)."""
        px, sx, kx = _vec(b, self.N, "b", plan=self._plan)
        if _is_torch(Y):
            py, sy, _ = _vec(Y, self.N, "Y")
            if sy != sx:
                raise TypeError("Y and b must live in the same memory space")
            L.check(L.load().lsfc_apply(self._plan, px, py, sx))
            return Y
        if not (isinstance(Y, np.ndarray) and Y.dtype == np.complex128 and Y.flags.c_contiguous and Y.size == self.N):
            raise TypeError("Y: need a contiguous complex128 array of the operator size")
        if sx != L.LSFC_MEM_HOST:
            raise TypeError("Y and b must live in the same memory space")
        L.check(L.load().lsfc_apply(self._plan, px, Y.ctypes.data_as(C.c_void_p), sx))
        return Y

    def set_nu(self, nu):
        nu = np.ascontiguousarray(nu, dtype=np.float64)
        if nu.size != self.N:
            raise ValueError("DimensionMismatch: nu")
        L.check(L.load().lsfc_plan_set_nu(self._plan, nu.ctypes.data_as(C.c_void_p), L.LSFC_MEM_HOST))
        self.nu = nu

    def working_symbol(self):
        """The symbol as the device pipeline stores it (tests / debugging)."""
        cnt = C.c_int64(0)
        L.check(L.load().lsfc_plan_get_symbol(self._plan, None, 0, C.byref(cnt)))
        out = np.empty(cnt.value, dtype=np.complex128)
        L.check(L.load().lsfc_plan_get_symbol(self._plan, out.ctypes.data_as(C.c_void_p), cnt.value, C.byref(cnt)))
        return out

    def set_tuning(self, **knobs):
        """Benchmark knobs of the pruned pipeline (results never depend on them)."""
        for key, value in knobs.items():
            L.check(L.load().lsfc_plan_set_tuning(self._plan, key.encode(), int(value)))

    def synchronize(self):
        L.check(L.load().lsfc_plan_synchronize(self._plan))

    def close(self):
        if self._plan is not None:
            L.load().lsfc_plan_destroy(self._plan)
            self._plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FastM(_Operator):
    """FastM(GFFT, nu, ne, me, n, m, k; quadRule="trapezoidal") -- src/FastConvolution.jl:11-27.
    GFFT: (ne, me) complex, centred order for Greengard_Vico, FFT order for trapezoidal."""

    def __init__(self, GFFT, nu, ne, me, n, m, k, quadRule="trapezoidal", flags=0, device=0, _plan=None):
        self.GFFT, self.ne, self.me, self.n, self.m = GFFT, int(ne), int(me), int(n), int(m)
        self.omega, self.quadRule = float(k), quadRule
        self.nu = np.ascontiguousarray(nu, dtype=np.float64).reshape(-1)
        if _plan is not None:
            self._plan = _plan
            return
        if quadRule not in _QUAD:
            # the reference fails at apply time with "B not defined" (src/FastConvolution.jl:106)
            raise NameError(f"UndefVarError: B not defined (unknown quadRule {quadRule!r})")
        G = np.asarray(GFFT, dtype=np.complex128)
        if G.shape != (self.ne, self.me):
            raise ValueError(f"DimensionMismatch: GFFT has shape {G.shape}, expected {(self.ne, self.me)}")
        if self.nu.size != self.n * self.m:
            raise ValueError("DimensionMismatch: nu")
        Gf = np.asfortranarray(G)
        plan = C.c_void_p()
        L.check(L.load().lsfc_plan_create_2d(C.byref(plan), self.n, self.m, self.ne, self.me,
                                             self.nu.ctypes.data_as(C.c_void_p), Gf.ctypes.data_as(C.c_void_p),
                                             self.omega, _QUAD[quadRule], flags, device))
        self._plan = plan


class FastM3D(_Operator):
    """FastM3D(GFFT, nu, ne, me, le, n, m, l, k; quadRule="Greengard_Vico") -- src/FastConvolution3D.jl:7-26."""

    def __init__(self, GFFT, nu, ne, me, le, n, m, l, k, quadRule="Greengard_Vico", flags=0, device=0, _plan=None):
        self.GFFT = GFFT
        self.ne, self.me, self.le, self.n, self.m, self.l = int(ne), int(me), int(le), int(n), int(m), int(l)
        self.omega, self.quadRule = float(k), quadRule
        self.nu = np.ascontiguousarray(nu, dtype=np.float64).reshape(-1)
        if _plan is not None:
            self._plan = _plan
            return
        if quadRule not in _QUAD:
            raise NameError(f"UndefVarError: B not defined (unknown quadRule {quadRule!r})")
        G = np.asarray(GFFT, dtype=np.complex128)
        if G.shape != (self.ne, self.me, self.le):
            raise ValueError(f"DimensionMismatch: GFFT has shape {G.shape}, expected {(self.ne, self.me, self.le)}")
        if self.nu.size != self.n * self.m * self.l:
            raise ValueError("DimensionMismatch: nu")
        Gf = np.asfortranarray(G)
        plan = C.c_void_p()
        L.check(L.load().lsfc_plan_create_3d(C.byref(plan), self.n, self.m, self.l, self.ne, self.me, self.le,
                                             self.nu.ctypes.data_as(C.c_void_p), Gf.ctypes.data_as(C.c_void_p),
                                             self.omega, _QUAD[quadRule], flags, device))
        self._plan = plan


# ---------------------------------------------------------------------------
# free functions with the reference's names
# ---------------------------------------------------------------------------
def size(M, dim=None):
    return M.size(dim)


def eltype(M):
    return M.eltype()


def fastconvolution(M, b):
    """b + omega^2 * G*(nu .* b)  (src/FastConvolution.jl:58-107; `*` of FastM3D)."""
    px, sx, keep = _vec(b, M.N, "b", plan=M._plan)
    y = M._out_like(keep)
    py, _, _ = _vec(y, M.N, "y")
    L.check(L.load().lsfc_apply(M._plan, px, py, sx))
    return y


def mul_(Y, M, b):
    """mul!(Y, M, b)."""
    return M.mul_(Y, b)


def FFTconvolution(M, b):
    """Bare convolution (src/FastConvolution.jl:110-154, src/FastConvolution3D.jl:39-63).  As in the
    reference, only the 2D trapezoidal branch multiplies by nu (src/FastConvolution.jl:122 vs :141)."""
    if isinstance(M, FastM) and M.n != M.m:
        # the reference allocates (ne, ne) and crops n in both dimensions (:120,:132): square grids only
        raise ValueError("DimensionMismatch: FFTconvolution(::FastM) assumes n == m")
    apply_nu = 1 if (isinstance(M, FastM) and M.quadRule == "trapezoidal") else 0
    px, sx, keep = _vec(b, M.N, "b", plan=M._plan)
    y = M._out_like(keep)
    py, _, _ = _vec(y, M.N, "y")
    L.check(L.load().lsfc_convolve(M._plan, px, py, apply_nu, sx))
    return y


def referenceValsTrapRule():
    """src/FastConvolution.jl:407-415."""
    x = 2.0 ** (-np.arange(6.0))
    w = np.array([1 - 0.892j, 1 - 1.35j, 1 - 1.79j, 1 - 2.23j, 1 - 2.67j, 1 - 3.11j])
    return x, w


def _grid2d(x, y):
    n, m = len(x), len(y)
    X = np.repeat(np.asarray(x, float)[:, None], m, axis=1).reshape(-1, order="F")
    Y = np.repeat(np.asarray(y, float)[None, :], n, axis=0).reshape(-1, order="F")
    return X, Y


def buildFastConvolution(x, y, h, k, nu, quadRule="trapezoidal", flags=0, device=0):
    """src/FastConvolution.jl:170-236.  ``nu`` is a callable nu(X, Y); the symbol is generated on the device."""
    x, y = np.asarray(x, float), np.asarray(y, float)
    n, m = len(x), len(y)
    X, Y = _grid2d(x, y)
    nuv = np.ascontiguousarray(nu(X, Y), dtype=np.float64)
    plan = C.c_void_p()
    if quadRule == "trapezoidal":
        _, D = referenceValsTrapRule()
        idx = int(round(k * h))                             # D[round(Int, k*h)], 1-based (src/FastConvolution.jl:175-176)
        if not 1 <= idx <= len(D):
            raise IndexError(f"BoundsError: attempt to access {len(D)}-element Vector at index [{idx}] "
                             "(D[round(Int, k*h)], src/FastConvolution.jl:176)")
        D0 = D[idx - 1]
        L.check(L.load().lsfc_plan_create_trap2d(C.byref(plan), n, m, float(x[0]), float(y[0]), float(h), float(k),
                                                 float(D0.real), float(D0.imag), nuv.ctypes.data_as(C.c_void_p), flags, device))
        return FastM(None, nuv, 2 * n - 1, 2 * m - 1, n, m, k, quadRule="trapezoidal", _plan=plan)
    if quadRule == "Greengard_Vico":
        box = abs(x[-1] - x[0]) + h
        L.check(L.load().lsfc_plan_create_gv2d(C.byref(plan), n, m, float(box), float(k), nuv.ctypes.data_as(C.c_void_p), flags, device))
        return FastM(None, nuv, 4 * n, 4 * m, n, m, k, quadRule="Greengard_Vico", _plan=plan)
    raise NameError(f"UndefVarError: unknown quadRule {quadRule!r}")


def buildFastConvolution3D(x, y, z, X, Y, Z, h, k, nu, quadRule="Greengard_Vico", flags=0, device=0):
    """src/FastConvolution3D.jl:68-101.  The (4n)^3 symbol is never materialised (137 GB at n=512):
    it is evaluated slab-wise on the device and reduced to the equivalent (2n)^3 grid."""
    if quadRule != "Greengard_Vico":
        raise NameError(f"UndefVarError: unknown quadRule {quadRule!r}")
    x = np.asarray(x, float)
    n, m, l = len(x), len(y), len(z)
    nuv = np.ascontiguousarray(nu(X, Y, Z) if callable(nu) else nu, dtype=np.float64).reshape(-1)
    box = abs(x[-1] - x[0]) + h
    plan = C.c_void_p()
    L.check(L.load().lsfc_plan_create_gv3d(C.byref(plan), n, m, l, float(box), float(k), nuv.ctypes.data_as(C.c_void_p), flags, device))
    return FastM3D(None, nuv, 4 * n, 4 * m, 4 * l, n, m, l, k, _plan=plan)


def sampleGConv(k, X, Y, indS, fastconv):
    """src/FastConvolution.jl:278-306: one FFTconvolution per delta source (batched on the device)."""
    return _sample(indS, fastconv)


def sampleG3D(k, X, Y, Z, indS, fastconv):
    """src/FastConvolution3D.jl:136-160."""
    return _sample(indS, fastconv)


def _sample(indS, M):
    N, ns = M.N, len(indS)
    out = np.empty((ns, N), dtype=np.complex128)
    if isinstance(M, FastM) and M.quadRule == "trapezoidal":
        # the reference's trapezoidal FFTconvolution multiplies the source by nu (src/FastConvolution.jl:122):
        # keep that quirk literally, one convolution per source
        R = np.zeros((ns, N), dtype=np.complex128)
        for i, ii in enumerate(indS):
            R[i, ii] = 1.0
        L.check(L.load().lsfc_apply_batch(M._plan, R.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), ns, 2, L.LSFC_MEM_HOST))
        return out
    # Greengard-Vico: rows of the Green's matrix = shifted copies of the (even) kernel -> one convolution + gathers
    src = np.ascontiguousarray(indS, dtype=np.int64)
    L.check(L.load().lsfc_sample_sources(M._plan, src.ctypes.data_as(C.c_void_p), ns, out.ctypes.data_as(C.c_void_p), L.LSFC_MEM_HOST))
    return out


# ---------------------------------------------------------------------------
# GMRES
# ---------------------------------------------------------------------------
class ConvergenceHistory:
    """The part of IterativeSolvers.ConvergenceHistory the reference scripts read."""

    def __init__(self, resnorm, iters, mvps, isconverged):
        self.data = {"resnorm": resnorm}
        self.iters, self.mvps, self.isconverged = iters, mvps, isconverged

    def __getitem__(self, key):
        return self.data[key.lstrip(":")]


class _DevPtr:
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer (complex128 vector)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<c16", "data": (ptr, False), "version": 2}


def _device_view(ptr, n):
    import torch
    return torch.as_tensor(_DevPtr(ptr, n), device="cuda")


_ORTH = {"ModifiedGramSchmidt": L.LSFC_ORTH_MGS, "ClassicalGramSchmidt": L.LSFC_ORTH_CGS, "DGKS": L.LSFC_ORTH_DGKS}


def gmres_(x, A, b, Pl=None, abstol=0.0, reltol=None, restart=None, maxiter=None, log=False,
           initially_zero=False, orth_meth="ModifiedGramSchmidt", Pl_on_device=False):
    """gmres!(x, A, b; Pl, abstol, reltol, restart, maxiter, log, initially_zero, orth_meth).

    ``Pl`` is a callable ``v -> None`` that overwrites the host numpy vector v with Pl \\ v -- the
    two-argument in-place ``ldiv!(Pl, v)`` of src/preconditioner.jl:147-170.  x is updated in place.
    ``Pl_on_device=True``: Pl instead receives a torch CUDA tensor aliasing the Krylov vector on the device (no PCIe
    round trip) and must work on torch's current stream."""
    N = A.N
    px, sx, keepx = _vec(x, N, "x", plan=A._plan)
    pb, sb, keepb = _vec(b, N, "b")
    if sx != sb:
        raise TypeError("x and b must live in the same memory space")
    if not _is_torch(x) and keepx is not x:
        raise TypeError("x must be a contiguous complex128 array (it is updated in place)")
    opts = L.GmresOpts()
    opts.restart = int(restart) if restart is not None else 0
    opts.maxiter = int(maxiter) if maxiter is not None else 0
    opts.reltol = float(reltol) if reltol is not None else -1.0
    opts.abstol = float(abstol)
    opts.orth = _ORTH[orth_meth]
    opts.initially_zero = 1 if initially_zero else 0
    err = []
    native_pc = getattr(Pl, "_pc", None) if Pl is not None else None
    if native_pc is not None:
        # device-resident SparsifyingPreconditioner: the library calls it directly, no Python in the loop; the Krylov
        # basis lives on the device wherever x and b live, so the preconditioner sees device vectors either way
        if sx == L.LSFC_MEM_DEVICE:
            import torch
            Pl.set_stream(torch.cuda.current_stream(x.device).cuda_stream)
        else:
            L.check(L.load().lsfc_plan_set_stream(A._plan, None))
            Pl.set_stream(0)
        lib = L.load()
        opts.precond = C.cast(lib.lsfc_precond_callback, L.PRECOND_FN)
        opts.precond_user = native_pc
        opts.precond_on_device = 1
    elif Pl is not None:
        def _cb(user, v, n):
            try:
                if Pl_on_device:
                    Pl(_device_view(C.cast(v, C.c_void_p).value, n))
                    return 0
                arr = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(2 * n,)).view(np.complex128)
                Pl(arr)
                return 0
            except Exception as e:      # never let a Python exception cross the C boundary
                err.append(e)
                return 1
        cb = L.PRECOND_FN(_cb)
        opts.precond = cb
        opts.precond_on_device = 1 if Pl_on_device else 0
    cap = int(maxiter) if maxiter is not None else N
    cap = max(1, min(cap, 1 << 20))
    res = L.GmresResult()
    resnorm = np.zeros(cap, dtype=np.float64)
    rc = L.load().lsfc_gmres(A._plan, px, pb, C.byref(opts), resnorm.ctypes.data_as(C.c_void_p), cap, C.byref(res), sx)
    if err:
        raise err[0]
    if rc not in (0, L.LSFC_ENOTCONV):
        L.check(rc)
    if log:
        return x, ConvergenceHistory(resnorm[:min(res.iters, cap)].copy(), int(res.iters), int(res.mvps), bool(res.converged))
    return x


def apply_batch(M, B, mode=0):
    """Rows of B (nrhs, N) through the operator in batched passes (lsfc_apply_batch): mode 0 ``M * b``, 1 bare
    convolution, 2 convolution of nu .* b.  The multi-vector form of src/FastConvolution3D.jl:146-159."""
    if _is_torch(B):
        nrhs = B.shape[0]
        pb, sb, keep = _vec(B.reshape(-1), M.N, "B", count=nrhs, plan=M._plan)
        import torch
        Y = torch.empty_like(keep)
        L.check(L.load().lsfc_apply_batch(M._plan, pb, C.c_void_p(Y.data_ptr()), nrhs, mode, sb))
        return Y.reshape(nrhs, M.N)
    B = np.ascontiguousarray(B, dtype=np.complex128)
    if B.ndim != 2 or B.shape[1] != M.N:
        raise ValueError(f"DimensionMismatch: B must be (nrhs, {M.N})")
    Y = np.empty_like(B)
    L.check(L.load().lsfc_apply_batch(M._plan, B.ctypes.data_as(C.c_void_p), Y.ctypes.data_as(C.c_void_p), B.shape[0], mode, L.LSFC_MEM_HOST))
    return Y


def gmres_batch_(X, A, B, Pl=None, abstol=0.0, reltol=None, restart=None, maxiter=None, log=False,
                 initially_zero=False, orth_meth="ModifiedGramSchmidt"):
    """``gmres!`` for several right-hand sides at once (rows of B, solutions in the rows of X, updated in place): the
    solves of tests/plasma_example.jl:160-176 (two incident directions, one after the other in the reference) run in
    lock step, one batched operator application per Arnoldi step.  Each row's iterates are those of ``gmres_`` on that
    row alone.  Returns X or (X, [ConvergenceHistory per row])."""
    torch_in = _is_torch(X)
    if torch_in:
        nrhs = X.shape[0]
        px, sx, keepx = _vec(X.reshape(-1), A.N, "X", count=nrhs, plan=A._plan)
        pb, sb, keepb = _vec(B.reshape(-1), A.N, "B", count=nrhs)
        if keepx.data_ptr() != X.data_ptr():
            raise TypeError("X must be contiguous (it is updated in place)")
    else:
        if not (isinstance(X, np.ndarray) and X.dtype == np.complex128 and X.flags.c_contiguous and X.ndim == 2 and X.shape[1] == A.N):
            raise TypeError(f"X must be a C-contiguous complex128 array of shape (nrhs, {A.N}) (it is updated in place)")
        nrhs = X.shape[0]
        Bc = np.ascontiguousarray(B, dtype=np.complex128)
        if Bc.shape != X.shape:
            raise ValueError("DimensionMismatch: B")
        px, sx, pb, sb, keepb = X.ctypes.data_as(C.c_void_p), L.LSFC_MEM_HOST, Bc.ctypes.data_as(C.c_void_p), L.LSFC_MEM_HOST, Bc
    if sx != sb:
        raise TypeError("X and B must live in the same memory space")
    opts = L.GmresOpts()
    opts.restart = int(restart) if restart is not None else 0
    opts.maxiter = int(maxiter) if maxiter is not None else 0
    opts.reltol = float(reltol) if reltol is not None else -1.0
    opts.abstol = float(abstol)
    opts.orth = _ORTH[orth_meth]
    opts.initially_zero = 1 if initially_zero else 0
    err = []
    if Pl is not None:
        def _cb(user, v, n):
            try:
                Pl(np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_double)), shape=(2 * n,)).view(np.complex128))
                return 0
            except Exception as e:
                err.append(e)
                return 1
        cb = L.PRECOND_FN(_cb)
        opts.precond = cb
    cap = max(1, min(int(maxiter) if maxiter is not None else A.N, 1 << 20))
    res = (L.GmresResult * nrhs)()
    resnorm = np.zeros((nrhs, cap), dtype=np.float64)
    rc = L.load().lsfc_gmres_batch(A._plan, px, pb, nrhs, C.byref(opts), resnorm.ctypes.data_as(C.c_void_p), cap, res, sx)
    if err:
        raise err[0]
    L.check(rc)
    if log:
        return X, [ConvergenceHistory(resnorm[j, :min(res[j].iters, cap)].copy(), int(res[j].iters), int(res[j].mvps), bool(res[j].converged))
                   for j in range(nrhs)]
    return X


# ---------------------------------------------------------------------------
# timing helpers used by bench.py (HIP events on the plan's own stream)
# ---------------------------------------------------------------------------
def time_apply(M, x_dev, y_dev, reps):
    px, sx, _ = _vec(x_dev, M.N, "x")
    py, sy, _ = _vec(y_dev, M.N, "y")
    if sx != L.LSFC_MEM_DEVICE or sy != L.LSFC_MEM_DEVICE:
        raise TypeError("time_apply needs device tensors")
    ms = C.c_double(0)
    L.check(L.load().lsfc_time_apply(M._plan, px, py, int(reps), C.byref(ms)))
    return ms.value


def profile_apply(M, x_dev, y_dev, reps=3):
    """(kernel / stage name, mean ms, algorithmic bytes) of one apply; a multi-device plan profiles on its own staging
    vectors (pass None, None)"""
    px = _vec(x_dev, M.N, "x")[0] if x_dev is not None else None
    py = _vec(y_dev, M.N, "y")[0] if y_dev is not None else None
    names = (C.c_char_p * 16)()
    ms, nb, ns = (C.c_double * 16)(), (C.c_double * 16)(), C.c_int(0)
    L.check(L.load().lsfc_profile_apply(M._plan, px, py, int(reps), 16, names, ms, nb, C.byref(ns)))
    return [(names[i].decode(), ms[i], nb[i]) for i in range(ns.value)]
