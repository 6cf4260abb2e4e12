"""ctypes binding of liblsfc.so (include/lsfc.h).  There is no Python or CPU
fallback: if the HIP library is missing the import of the compute path fails
loudly, and every compute entry point returns LSFC_ENODEV without a GPU."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (developer override for A/B builds: LSFC_LIBRARY=/path/to/another/liblsfc.so; a missing file is an error either way)
LIB_PATH = os.environ.get("LSFC_LIBRARY") or os.path.join(_HERE, "liblsfc.so")

LSFC_QUAD_TRAPEZOIDAL, LSFC_QUAD_GREENGARD_VICO = 0, 1
LSFC_MEM_HOST, LSFC_MEM_DEVICE = 0, 1
LSFC_FLAG_DEFAULT, LSFC_FLAG_LITERAL_PAD, LSFC_FLAG_FORCE_ROCFFT, LSFC_FLAG_PATCH_SINGULAR = 0, 1, 2, 4
LSFC_ORTH_MGS, LSFC_ORTH_CGS, LSFC_ORTH_DGKS = 0, 1, 2
LSFC_ENODEV, LSFC_ENOTCONV = -2, -5
LSFC_UNIQUE_ID_BYTES = 128

PRECOND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64)


class GmresOpts(C.Structure):
    _fields_ = [("restart", C.c_int), ("maxiter", C.c_int64), ("reltol", C.c_double), ("abstol", C.c_double),
                ("orth", C.c_int), ("initially_zero", C.c_int), ("precond", PRECOND_FN), ("precond_user", C.c_void_p),
                ("precond_on_device", C.c_int)]


class GmresResult(C.Structure):
    _fields_ = [("iters", C.c_int64), ("mvps", C.c_int64), ("converged", C.c_int), ("final_resnorm", C.c_double)]


class LsfcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lsfc error {code}: {msg}")
        self.code = code


# every symbol include/lsfc.h declares: name -> (restype, argtypes)
_P, _I, _L, _D, _U = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_uint
_PP = C.POINTER(C.c_void_p)
SIGNATURES = {
    "lsfc_plan_create_2d": (_I, [_PP, _L, _L, _L, _L, _P, _P, _D, _I, _U, _I]),
    "lsfc_plan_create_3d": (_I, [_PP, _L, _L, _L, _L, _L, _L, _P, _P, _D, _I, _U, _I]),
    "lsfc_plan_create_gv2d": (_I, [_PP, _L, _L, _D, _D, _P, _U, _I]),
    "lsfc_plan_create_trap2d": (_I, [_PP, _L, _L, _D, _D, _D, _D, _D, _D, _P, _U, _I]),
    "lsfc_plan_create_gv3d": (_I, [_PP, _L, _L, _L, _D, _D, _P, _U, _I]),
    "lsfc_plan_destroy": (_I, [_P]),
    "lsfc_plan_size": (_L, [_P]),
    "lsfc_plan_dims": (_I, [_P, C.POINTER(_L), C.POINTER(_L)]),
    "lsfc_plan_pipeline": (C.c_char_p, [_P]),
    "lsfc_plan_set_nu": (_I, [_P, _P, _I]),
    "lsfc_plan_get_symbol": (_I, [_P, _P, _L, C.POINTER(_L)]),
    "lsfc_apply": (_I, [_P, _P, _P, _I]),
    "lsfc_convolve": (_I, [_P, _P, _P, _I, _I]),
    "lsfc_apply_batch": (_I, [_P, _P, _P, _L, _I, _I]),
    "lsfc_sample_sources": (_I, [_P, _P, _L, _P, _I]),
    "lsfc_gmres": (_I, [_P, _P, _P, C.POINTER(GmresOpts), _P, _L, C.POINTER(GmresResult), _I]),
    "lsfc_gmres_batch": (_I, [_P, _P, _P, _L, C.POINTER(GmresOpts), _P, _L, C.POINTER(GmresResult), _I]),
    "lsfc_precond_create": (_I, [_PP, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I]),
    "lsfc_precond_destroy": (_I, [_P]),
    "lsfc_precond_set_stream": (_I, [_P, _P]),
    "lsfc_precond_apply": (_I, [_P, _P, _I]),
    "lsfc_precond_callback": (_I, [_P, _P, _L]),
    "lsfc_precond_stats": (_I, [_P, C.POINTER(_L), C.POINTER(_L), C.POINTER(_L)]),
    "lsfc_plan_set_stream": (_I, [_P, _P]),
    "lsfc_plan_synchronize": (_I, [_P]),
    "lsfc_plan_set_tuning": (_I, [_P, C.c_char_p, _I]),
    "lsfc_time_apply": (_I, [_P, _P, _P, _I, C.POINTER(_D)]),
    "lsfc_profile_apply": (_I, [_P, _P, _P, _I, _I, C.POINTER(C.c_char_p), C.POINTER(_D), C.POINTER(_D), C.POINTER(_I)]),
    "lsfc_device_count": (_I, [C.POINTER(_I)]),
    "lsfc_malloc": (_I, [_PP, C.c_size_t, _I]),
    "lsfc_free": (_I, [_P]),
    "lsfc_memcpy_h2d": (_I, [_P, _P, C.c_size_t]),
    "lsfc_memcpy_d2h": (_I, [_P, _P, C.c_size_t]),
    "lsfc_host_register": (_I, [_P, C.c_size_t]),
    "lsfc_host_unregister": (_I, [_P]),
    "lsfc_host_alloc": (_I, [_PP, C.c_size_t]),
    "lsfc_host_free": (_I, [_P]),
    "lsfc_dist_unique_id": (_I, [_P]),
    "lsfc_dist_plan_create_gv3d": (_I, [_PP, _L, _L, _L, _D, _D, _P, _U, _I, _I, _I, _P]),
    "lsfc_dist_sim_plan_create_gv3d": (_I, [_PP, _L, _L, _L, _D, _D, _P, _U, _I, _I, _I]),
    "lsfc_dist_sim_apply": (_I, [_PP, _I, _PP, _PP, _I]),
    "lsfc_plan_create_gv3d_multi": (_I, [_PP, _L, _L, _L, _D, _D, _P, _U, C.POINTER(_I), _I]),
    "lsfc_multi_info": (_I, [_P, C.POINTER(_I), C.POINTER(_I), C.POINTER(_L), C.POINTER(C.c_char_p)]),
    "lsfc_multi_apply_dev": (_I, [_P, _PP, _PP, _I]),
    "lsfc_last_error": (C.c_char_p, []),
    "lsfc_padded_length": (_I, [C.c_int64]),
    "lsfc_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load liblsfc.so (built by ``__graft_entry__.build()`` / ``make -C csrc``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback for the lsfc operator")
        try:
            # PyTorch-ROCm bundles its own HIP runtime; when both live in one process torch's copy must be
            # loaded first, otherwise torch.cuda later reports "No HIP GPUs are available".
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise LsfcError(rc, load().lsfc_last_error().decode(errors="replace"))
    return rc


def host_register(a):
    """Page-lock a long-lived numpy vector (lsfc_host_register): host-vector applies then move it by DMA."""
    check(load().lsfc_host_register(a.ctypes.data_as(C.c_void_p), a.nbytes))


def host_unregister(a):
    check(load().lsfc_host_unregister(a.ctypes.data_as(C.c_void_p)))


class _PinnedBlock:
    """owner of one lsfc_host_alloc block; freed when the last numpy view of it is gone"""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        check(load().lsfc_host_alloc(C.byref(self.ptr), max(int(nbytes), 1)))
        self.nbytes = int(nbytes)

    def __del__(self):
        try:
            if self.ptr:
                load().lsfc_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


def host_empty(shape, dtype="complex128"):
    """numpy array in page-locked memory owned by the HIP runtime (lsfc_host_alloc): host-vector applies move it by DMA.
    The preferred form of a page-locked work vector (page-aligned, shares no page with the process heap)."""
    import numpy as np
    dt = np.dtype(dtype)
    count = int(np.prod(shape)) if np.ndim(shape) else int(shape)
    blk = _PinnedBlock(count * dt.itemsize)
    buf = (C.c_char * max(blk.nbytes, 1)).from_address(blk.ptr.value)
    buf._owner = blk                               # numpy keeps `buf` (its base) alive, `buf` keeps the block alive
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


def device_count():
    n = C.c_int(0)
    load().lsfc_device_count(C.byref(n))
    return n.value
