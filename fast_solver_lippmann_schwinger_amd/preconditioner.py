"""Device-resident apply of the reference's SparsifyingPreconditioner (src/preconditioner.jl:27-58, 132-170):

    P = SparsifyingPreconditioner(Msp, As)        # MspInv = lu(Msp) on the host      (:35)
    ldiv!(P, b):  b[:] = MspInv \\ (As * b)        # once per Arnoldi step             (:132-170)

The sparse LU stays on the host, as in the reference (UMFPACK there; scipy's SuperLU here -- any LU of Msp gives
the same operator).  Its factors and As are uploaded once; `ldiv_` then runs entirely on the device (csrc/precond.hip:
CSR SpMV + two level-scheduled sparse triangular solves replayed from one hipGraph), so under `gmres_` the Krylov
vector never crosses PCIe.  Assembling Msp / As (src/SparsifyingMatrix*.jl) is outside this package."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _csr_arrays(A):
    A = A.tocsr()
    A.sort_indices()
    return (np.ascontiguousarray(A.indptr, dtype=np.int64), np.ascontiguousarray(A.indices, dtype=np.int64),
            np.ascontiguousarray(A.data, dtype=np.complex128))


class SparsifyingPreconditioner:
    """SparsifyingPreconditioner(Msp, As; solverType="UMFPACK") -- src/preconditioner.jl:27-58.
    Msp, As: scipy.sparse matrices (N x N, complex).  ``lu``: optional pre-computed scipy.sparse.linalg.SuperLU of Msp."""

    def __init__(self, Msp, As, solverType="UMFPACK", device=0, lu=None):
        import scipy.sparse as sp
        import scipy.sparse.linalg as spla
        if solverType not in ("UMFPACK", "MKLPARDISO"):
            raise NameError(f"UndefVarError: unknown solverType {solverType!r}")
        Msp = sp.csc_matrix(Msp, dtype=np.complex128)
        As = sp.csr_matrix(As, dtype=np.complex128)
        N = Msp.shape[0]
        if Msp.shape != (N, N) or As.shape != (N, N):
            raise ValueError("DimensionMismatch: Msp and As must be square and of the same size")
        self.Msp, self.As, self.solverType, self.N = Msp, As, solverType, N
        if lu is None:
            lu = spla.splu(Msp)                      # Pr * Msp * Pc = L * U, no row scaling
        # scipy: Pr Msp Pc = L U with row i of Msp -> row perm_r[i] of L U, column j of Msp -> column perm_c[j] of L U;
        # the C ABI wants the inverse maps (row / column of Msp behind row / column k of L U)
        row_gather = np.empty(N, dtype=np.int64)
        row_gather[np.asarray(lu.perm_r, dtype=np.int64)] = np.arange(N, dtype=np.int64)
        col_scatter = np.empty(N, dtype=np.int64)
        col_scatter[np.asarray(lu.perm_c, dtype=np.int64)] = np.arange(N, dtype=np.int64)
        a_ptr, a_col, a_val = _csr_arrays(As)
        l_ptr, l_col, l_val = _csr_arrays(lu.L)
        u_ptr, u_col, u_val = _csr_arrays(lu.U)
        self.nnz_L, self.nnz_U = int(l_val.size), int(u_val.size)
        pc = C.c_void_p()
        p = lambda a: a.ctypes.data_as(C.c_void_p)   # noqa: E731
        L.check(L.load().lsfc_precond_create(C.byref(pc), N, p(a_ptr), p(a_col), p(a_val), p(l_ptr), p(l_col), p(l_val),
                                             p(u_ptr), p(u_col), p(u_val), p(row_gather), p(col_scatter), None, int(device)))
        self._pc = pc

    # -- ldiv!(P, b) / P \\ b -- src/preconditioner.jl:132-170 ---------------------------------------------------------
    def ldiv_(self, v):
        """in place: v <- Msp^{-1} (As v).  numpy vector (copied over PCIe) or torch CUDA tensor (stays on the device,
        enqueued on torch's current stream)."""
        from .operators import _vec, _is_torch
        pv, space, keep = _vec(v, self.N, "v")
        if not _is_torch(v) and keep is not v:
            raise TypeError("v must be a contiguous complex128 array (it is updated in place)")
        if space == L.LSFC_MEM_DEVICE:
            import torch
            self.set_stream(torch.cuda.current_stream(v.device).cuda_stream)
        L.check(L.load().lsfc_precond_apply(self._pc, pv, space))
        return v

    def __call__(self, v):                           # usable as the Pl callable of gmres_
        self.ldiv_(v)

    def solve(self, b):
        """P \\ b (out of place)"""
        from .operators import _is_torch
        v = b.clone() if _is_torch(b) else np.array(b, dtype=np.complex128)
        return self.ldiv_(v)

    def set_stream(self, stream):
        L.check(L.load().lsfc_precond_set_stream(self._pc, C.c_void_p(int(stream))))

    def stats(self):
        """dependency levels of the L and U solves, kernel launches captured in the graph"""
        a, b, c = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(L.load().lsfc_precond_stats(self._pc, C.byref(a), C.byref(b), C.byref(c)))
        return {"levels_L": a.value, "levels_U": b.value, "launches": c.value, "nnz_L": self.nnz_L, "nnz_U": self.nnz_U}

    def close(self):
        if getattr(self, "_pc", None):
            L.load().lsfc_precond_destroy(self._pc)
            self._pc = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
