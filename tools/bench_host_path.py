"""The drop-in path with HOST vectors (mul!(Y, M, b), src/FastConvolution.jl:50-54) at 3D n (default 512): ms per apply with
pageable and with page-locked (lsfc_host_register) numpy vectors, chunk-pipelined (default) and as one copy each way
(LSFC_HOST_PIPELINE=0), next to the device-resident apply and the PCIe floor.  One JSON line per case.
usage: python tools/bench_host_path.py [n]      (environment LSFC_HOST_PIPELINE=0|K is read by the library at first use)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fast_solver_lippmann_schwinger_amd as lsfc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
h = 1.0 / n
x = -0.5 + h * np.arange(n)
N = n ** 3
nu = np.random.default_rng(0).uniform(-0.3, 0.3, N)
M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, 1.0 / h, nu)
rng = np.random.default_rng(1)
b = rng.standard_normal(N) + 1j * rng.standard_normal(N)
y = np.empty_like(b)
tag = os.environ.get("LSFC_HOST_PIPELINE", "auto")


def timed(reps=4):
    M.mul_(y, b)
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); M.mul_(y, b); best = min(best, time.time() - t0)
    return best


t = timed()
ref = y.copy()
print(json.dumps({"n": n, "vectors": "pageable", "pipeline": tag, "ms_per_apply": round(1e3 * t, 2), "applies_per_s": round(1 / t, 2),
                  "GBps_each_way_if_copies_only": round(2 * N * 16 / t / 1e9, 1)}), flush=True)
lsfc.host_register(b); lsfc.host_register(y)
t = timed()
print(json.dumps({"n": n, "vectors": "page-locked (lsfc_host_register)", "pipeline": tag, "ms_per_apply": round(1e3 * t, 2), "applies_per_s": round(1 / t, 2),
                  "same_result": bool(np.array_equal(ref, y))}), flush=True)
lsfc.host_unregister(b); lsfc.host_unregister(y)
bo, yo = lsfc.host_empty(b.shape), lsfc.host_empty(b.shape)       # page-locked memory owned by the runtime (lsfc_host_alloc)
bo[:] = b; yo[:] = 0
b_keep, y_keep = b, y
b, y = bo, yo
t = timed()
print(json.dumps({"n": n, "vectors": "page-locked, runtime-owned (lsfc_host_alloc)", "pipeline": tag, "ms_per_apply": round(1e3 * t, 2), "applies_per_s": round(1 / t, 2),
                  "same_result": bool(np.array_equal(ref, y))}), flush=True)
b, y = b_keep, y_keep
del bo, yo
import torch
xb = torch.from_numpy(b).cuda(); yb = torch.empty_like(xb)
ms = lsfc.time_apply(M, xb, yb, 10) / 10
print(json.dumps({"n": n, "vectors": "device-resident", "ms_per_apply": round(ms, 3), "host_equals_device": bool(np.array_equal(yb.cpu().numpy(), ref))}), flush=True)

# several right-hand sides through lsfc_apply_batch with host vectors: across right-hand sides upload and download overlap
R = 4
B = np.stack([b * (1 + r) for r in range(R)]); Y = np.empty_like(B)
import ctypes as C
from fast_solver_lippmann_schwinger_amd import _lib as L
Y[:] = 0                                       # (touch the result pages once: first-touch page faults are not transfer time)
def timed_batch():
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        L.check(L.load().lsfc_apply_batch(M._plan, B.ctypes.data_as(C.c_void_p), Y.ctypes.data_as(C.c_void_p), R, 0, L.LSFC_MEM_HOST))
        best = min(best, time.time() - t0)
    return best, Y.copy()
t, Yl = timed_batch()
print(json.dumps({"n": n, "vectors": "pageable", "nrhs": R, "pipeline": tag, "ms_per_rhs": round(1e3 * t / R, 2), "first_row_equals_single_apply": bool(np.array_equal(Yl[0], ref))}), flush=True)
lsfc.host_register(B)
lsfc.host_register(Y)
best = 1e9
for _ in range(3):
    t0 = time.time()
    L.check(L.load().lsfc_apply_batch(M._plan, B.ctypes.data_as(C.c_void_p), Y.ctypes.data_as(C.c_void_p), R, 0, L.LSFC_MEM_HOST))
    best = min(best, time.time() - t0)
print(json.dumps({"n": n, "vectors": "page-locked", "nrhs": R, "pipeline": tag, "ms_per_rhs": round(1e3 * best / R, 2), "same_result": bool(np.array_equal(Y, Yl))}), flush=True)
lsfc.host_unregister(B); lsfc.host_unregister(Y)
