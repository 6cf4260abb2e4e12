"""CPU, world_size = 2, gloo: the z-slab decomposition of the operator (csrc/dist.hip) restated in numpy and run
over real torch.distributed ranks.  Checks the two layout facts the HIP path relies on --
  (1) writing the x-pass output as [dest rank][W][m][lz] packs the transpose, and the blocks received from all
      ranks, concatenated in rank order, ARE the natural array [W][m][l] on the owned x' range;
  (2) the way back: block p of that array is rank p's z range and lands in [src rank][W][m][lz], the layout the
      inverse x pass reads --
and that the result equals the single-process oracle.  (The GPU form of the same test, with simulated ranks on one
device, is tests/test_gpu_distributed.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import lsfc_oracle as o
from fast_solver_lippmann_schwinger_amd.distributed import slab_range


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _alltoall_blocks(send_blocks, rank, world):
    """all-to-all of equal numpy blocks over gloo (all_gather of everything, keep what is addressed to me)."""
    flat = torch.from_numpy(np.stack(send_blocks).view(np.float64).copy())
    gathered = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    return [g.numpy().view(np.complex128).reshape(np.stack(send_blocks).shape)[rank] for g in gathered]


def _worker(rank, world, port, n, m, l, k, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)                          # same data on every rank
        G2 = rng.standard_normal((2 * n, 2 * m, 2 * l)) + 1j * rng.standard_normal((2 * n, 2 * m, 2 * l))
        nu = rng.uniform(-0.3, 0.3, n * m * l)
        b = rng.standard_normal(n * m * l) + 1j * rng.standard_normal(n * m * l)
        lo, hi = slab_range(l, rank, world)
        lz, Lx, W = hi - lo, 2 * n, 2 * n // world
        xl = (nu * b).reshape((n, m, l), order="F")[:, :, lo:hi]
        # phase 1: x pass on own planes, written packed per destination rank
        A = np.fft.fft(np.concatenate([xl, np.zeros_like(xl)], axis=0), axis=0)            # [Lx][m][lz]
        S1 = [A[qd * W:(qd + 1) * W].reshape(-1, order="F") for qd in range(world)]         # block q = [W][m][lz]
        # exchange 1; fact (1): concatenation of the received blocks is the natural [W][m][l]
        R1 = np.concatenate(_alltoall_blocks(S1, rank, world)).reshape((W, m, l), order="F")
        # phase 2 on the owned x' range
        B = np.fft.fft(np.concatenate([R1, np.zeros_like(R1)], axis=1), axis=1)
        B = np.fft.fft(np.concatenate([B, np.zeros((W, 2 * m, l), complex)], axis=2), axis=2)
        B = B * G2[rank * W:(rank + 1) * W]
        B = np.fft.ifft(B, axis=2)[:, :, :l]
        B = np.fft.ifft(B, axis=1)[:, :m, :]
        # exchange 2; fact (2): block p of the flat natural array is rank p's z range
        flat = B.reshape(-1, order="F")
        blk = W * m * lz
        back = _alltoall_blocks([flat[p * blk:(p + 1) * blk] for p in range(world)], rank, world)
        full = np.concatenate([bb.reshape((W, m, lz), order="F") for bb in back], axis=0)    # [Lx][m][lz]
        yl = b.reshape((n, m, l), order="F")[:, :, lo:hi] + k**2 * np.fft.ifft(full, axis=0)[:n]
        ref = o.apply_reduced(G2, nu, k, b, (n, m, l)).reshape((n, m, l), order="F")[:, :, lo:hi]
        q.put((rank, float(np.linalg.norm(yl - ref) / np.linalg.norm(ref))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims", [(8, 4, 6), (16, 16, 16)])
def test_slab_decomposition_world2_gloo(dims):
    n, m, l = dims
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, m, l, 3.0, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0] < 1e-13 and res[1] < 1e-13, res


def test_slab_range():
    assert slab_range(512, 3, 8) == (192, 256)
    with pytest.raises(ValueError):
        slab_range(10, 0, 4)
