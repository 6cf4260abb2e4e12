"""Slab-distributed 3D operator: one process per GPU, z-slabs, RCCL all-to-all over xGMI
(csrc/dist.hip).  The reference has no distributed mode; this is the multi-GPU
form of `buildFastConvolution3D` + `*` (src/FastConvolution3D.jl:31-101).

torch.distributed is used only as plumbing: to ship the 128-byte RCCL unique id from rank 0 to
the other ranks.  The data-path collectives are issued by the HIP library itself."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from .operators import FastM3D


def slab_range(l, rank, nranks):
    """z planes [lo, hi) owned by `rank` (the library requires nranks | l)."""
    if l % nranks:
        raise ValueError(f"l = {l} is not divisible by the number of ranks {nranks}")
    lz = l // nranks
    return rank * lz, (rank + 1) * lz


def exchange_unique_id(rank, nranks, group=None):
    """Rank 0 creates the RCCL unique id; everyone receives it through torch.distributed.  With more than one rank an
    initialised process group is mandatory: without the broadcast the other ranks would hand ncclCommInitRank an id
    nobody else has and block forever (pass ``uid=`` to build_distributed_3d to ship the id some other way)."""
    import torch
    import torch.distributed as dist
    if nranks > 1 and not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError(f"exchange_unique_id: {nranks} ranks but torch.distributed is not initialised; call "
                           "dist.init_process_group first or pass the 128-byte id of lsfc_dist_unique_id() as uid=")
    buf = (C.c_ubyte * L.LSFC_UNIQUE_ID_BYTES)()
    if rank == 0:
        L.check(L.load().lsfc_dist_unique_id(buf))
    if dist.is_available() and dist.is_initialized():
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=0, group=group)
        buf = (C.c_ubyte * L.LSFC_UNIQUE_ID_BYTES)(*t.cpu().tolist())
    return buf


def build_distributed_3d(n, h, k, nu_local, rank, nranks, device, m=None, l=None, flags=0, group=None, uid=None):
    """Distributed `buildFastConvolution3D` on the half-open grid x = x0 + h*(0..n-1): this rank passes the contrast
    on its own z planes (flat, x fastest).  Returns a FastM3D whose vectors are the local slabs."""
    m = n if m is None else m
    l = n if l is None else l
    lo, hi = slab_range(l, rank, nranks)
    nuv = np.ascontiguousarray(nu_local, dtype=np.float64).reshape(-1)
    if nuv.size != n * m * (hi - lo):
        raise ValueError("DimensionMismatch: nu_local")
    box = n * h                                   # |x[end] - x[1]| + h
    if uid is None:
        uid = exchange_unique_id(rank, nranks, group)
    else:
        uid = (C.c_ubyte * L.LSFC_UNIQUE_ID_BYTES)(*bytes(uid))
    plan = C.c_void_p()
    L.check(L.load().lsfc_dist_plan_create_gv3d(C.byref(plan), n, m, l, float(box), float(k), nuv.ctypes.data_as(C.c_void_p),
                                                flags, device, rank, nranks, uid))
    return FastM3D(None, nuv, 4 * n, 4 * m, 4 * l, n, m, l, k, _plan=plan)


class MultiDeviceFastM3D(FastM3D):
    """`buildFastConvolution3D` on several GPUs driven from THIS one process (lsfc_plan_create_gv3d_multi): the form a
    single-process host -- the reference is one Julia process, examples/example3D.jl:54,78 -- uses to reach the
    multi-GPU path.  Behaves like FastM3D with host (numpy) vectors: ``M * b``, ``mul_``, ``FFTconvolution``,
    ``gmres_`` (host preconditioner callback).  ``devices`` may list a device more than once (logical ranks on one GPU,
    peer-copy transport): that is how the path is tested where only one GPU is visible."""

    def __init__(self, n, h, k, nu, devices, m=None, l=None, flags=0):
        m = n if m is None else m
        l = n if l is None else l
        nuv = np.ascontiguousarray(nu, dtype=np.float64).reshape(-1)
        if nuv.size != n * m * l:
            raise ValueError("DimensionMismatch: nu")
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        plan = C.c_void_p()
        L.check(L.load().lsfc_plan_create_gv3d_multi(C.byref(plan), n, m, l, float(n * h), float(k), nuv.ctypes.data_as(C.c_void_p),
                                                     flags, devs, len(devices)))
        super().__init__(None, nuv, 4 * n, 4 * m, 4 * l, n, m, l, k, _plan=plan)
        self.devices = [int(d) for d in devices]

    def _info(self):
        nd, ln, tr = C.c_int(0), C.c_int64(0), C.c_char_p()
        L.check(L.load().lsfc_multi_info(self._plan, C.byref(nd), None, C.byref(ln), C.byref(tr)))
        return nd.value, ln.value, tr.value.decode()

    @property
    def transport(self):
        return self._info()[2]

    @property
    def local_n(self):
        return self._info()[1]

    def apply_dev(self, xs, ys, mode=0):
        """device-resident slabs: xs[r], ys[r] = torch complex128 tensors on cuda:devices[r] (stream-ordered, no copy)"""
        P = len(self.devices)
        xp = (C.c_void_p * P)(*[x.data_ptr() for x in xs])
        yp = (C.c_void_p * P)(*[y.data_ptr() for y in ys])
        L.check(L.load().lsfc_multi_apply_dev(self._plan, xp, yp, mode))

    def bench(self, steps, warmup):
        """`steps` timed applies on device-resident slabs (host clock around the enqueue + a synchronisation of every
        device), and the per-stage profile: what bench.py --single-process reports"""
        import time
        import torch
        from .operators import profile_apply
        ln = self.local_n
        xs, ys = [], []
        for r, d in enumerate(self.devices):
            g = torch.Generator(device=f"cuda:{d}"); g.manual_seed(20250224 + r)
            xs.append(torch.randn(ln, dtype=torch.complex128, device=f"cuda:{d}", generator=g))
            ys.append(torch.empty(ln, dtype=torch.complex128, device=f"cuda:{d}"))
        for d in set(self.devices):
            torch.cuda.synchronize(d)
        for _ in range(warmup):
            self.apply_dev(xs, ys)
        self.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.apply_dev(xs, ys)
        self.synchronize()
        elapsed = time.perf_counter() - t0
        return {"elapsed_s": elapsed, "stages": profile_apply(self, None, None, reps=3)}


class SimulatedRanks:
    """`nranks` logical ranks on ONE device (lsfc_dist_sim_*): the test double of the multi-GPU operator.  Same
    kernels, layouts and symbol slabs; the two slab exchanges are device-to-device copies instead of RCCL."""

    def __init__(self, n, m, l, h, k, nu, nranks, device=0, flags=0):
        self.n, self.m, self.l, self.nranks = n, m, l, nranks
        nu = np.ascontiguousarray(nu, dtype=np.float64).reshape(-1)
        self.plans = (C.c_void_p * nranks)()
        self.bounds = [slab_range(l, r, nranks) for r in range(nranks)]
        for r, (lo, hi) in enumerate(self.bounds):
            loc = np.ascontiguousarray(nu[n * m * lo:n * m * hi])
            plan = C.c_void_p()
            L.check(L.load().lsfc_dist_sim_plan_create_gv3d(C.byref(plan), n, m, l, float(n * h), float(k),
                                                            loc.ctypes.data_as(C.c_void_p), flags, device, r, nranks))
            self.plans[r] = plan

    def _run(self, b, mode):
        n, m = self.n, self.m
        b = np.ascontiguousarray(b, dtype=np.complex128)
        xs = [np.ascontiguousarray(b[n * m * lo:n * m * hi]) for lo, hi in self.bounds]
        ys = [np.empty_like(x) for x in xs]
        xp = (C.c_void_p * self.nranks)(*[x.ctypes.data for x in xs])
        yp = (C.c_void_p * self.nranks)(*[y.ctypes.data for y in ys])
        L.check(L.load().lsfc_dist_sim_apply(self.plans, self.nranks, xp, yp, mode))
        return np.concatenate(ys)

    def apply(self, b):
        return self._run(b, 0)

    def convolve(self, b, apply_nu=False):
        return self._run(b, 2 if apply_nu else 1)

    def close(self):
        for r in range(self.nranks):
            if self.plans[r]:
                L.load().lsfc_plan_destroy(self.plans[r])
                self.plans[r] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
