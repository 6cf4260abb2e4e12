"""Host <-> device copy rates of one MI355X box for the sizes of a 512^3 complex vector (2.15 GB): pageable and pinned
host memory, whole and in z-chunks, one direction and both at once.  Feeds the floor of the host-vector apply."""
import time, torch
N = 512 ** 3
dev = torch.device("cuda")
d = torch.empty(N, dtype=torch.complex128, device=dev)
d2 = torch.empty(N, dtype=torch.complex128, device=dev)
hp = torch.empty(N, dtype=torch.complex128).normal_()
t0 = time.time(); hq = torch.empty(N, dtype=torch.complex128).pin_memory(); t1 = time.time()
print(f"pin_memory of 2.15 GB: {1e3 * (t1 - t0):.0f} ms")
hq2 = torch.empty(N, dtype=torch.complex128).pin_memory()
GB = N * 16 / 1e9
def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t = time.time(); f(); torch.cuda.synchronize(); best = min(best, time.time() - t)
    return best
for name, h in (("pageable", hp), ("pinned", hq)):
    t = timed(lambda: d.copy_(h, non_blocking=True)); print(f"H2D {name:8s} whole : {1e3 * t:7.1f} ms  {GB / t:6.1f} GB/s")
    t = timed(lambda: h.copy_(d, non_blocking=True)); print(f"D2H {name:8s} whole : {1e3 * t:7.1f} ms  {GB / t:6.1f} GB/s")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): d.copy_(hq, non_blocking=True)
    with torch.cuda.stream(s2): hq2.copy_(d2, non_blocking=True)
t = timed(both); print(f"H2D + D2H pinned at once: {1e3 * t:7.1f} ms  {2 * GB / t:6.1f} GB/s total")
for K in (8, 32):
    c = N // K
    def chunks():
        for k in range(K): d[k * c:(k + 1) * c].copy_(hq[k * c:(k + 1) * c], non_blocking=True)
    t = timed(chunks); print(f"H2D pinned in {K} chunks: {1e3 * t:7.1f} ms  {GB / t:6.1f} GB/s")
    def chunksp():
        for k in range(K): d[k * c:(k + 1) * c].copy_(hp[k * c:(k + 1) * c], non_blocking=True)
    t = timed(chunksp); print(f"H2D pageable in {K} chunks: {1e3 * t:7.1f} ms  {GB / t:6.1f} GB/s")
t0 = time.time(); torch.cuda.cudart().cudaHostRegister(hp.data_ptr(), N * 16, 0); t1 = time.time()
print(f"hipHostRegister of 2.15 GB: {1e3 * (t1 - t0):.0f} ms")
t = timed(lambda: d.copy_(hp, non_blocking=True)); print(f"H2D registered whole : {1e3 * t:7.1f} ms  {GB / t:6.1f} GB/s")
t0 = time.time(); torch.cuda.cudart().cudaHostUnregister(hp.data_ptr()); t1 = time.time()
print(f"hipHostUnregister: {1e3 * (t1 - t0):.0f} ms")
