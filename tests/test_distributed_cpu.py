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


def _worker(rank, world, port, n, m, l, k, q, K=1, halves=False):
    """K: pipeline chunks of the owned x' range (csrc/dist.hip: S1 is [dest rank][chunk][Wc][m][lz], block index
    dest * K + chunk; R1 is [chunk][slot = source rank][Wc][m][lz]).  halves: every message is shipped as the two z
    halves of its block, as the split pipeline ends do (blocks are z-slowest, so a z half is a contiguous half)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)                          # same data on every rank
        G2 = rng.standard_normal((2 * n, 2 * m, 2 * l)) + 1j * rng.standard_normal((2 * n, 2 * m, 2 * l))
        nu = rng.uniform(-0.3, 0.3, n * m * l)
        b = rng.standard_normal(n * m * l) + 1j * rng.standard_normal(n * m * l)
        lo, hi = slab_range(l, rank, world)
        lz, Lx, W = hi - lo, 2 * n, 2 * n // world
        Wc = W // K
        B = Wc * m * lz                                          # elements of one (rank, chunk) block
        xl = (nu * b).reshape((n, m, l), order="F")[:, :, lo:hi]
        # phase 1: x pass on own planes, written packed per (destination rank, chunk): storage index s -> block s // Wc
        A = np.fft.fft(np.concatenate([xl, np.zeros_like(xl)], axis=0), axis=0)            # [Lx][m][lz]
        S1 = np.concatenate([A[blk * Wc:(blk + 1) * Wc].reshape(-1, order="F") for blk in range(world * K)])

        def ship(buf_blocks):
            """all-to-all of one block per destination; with `halves`, as two messages per block"""
            if not halves:
                return _alltoall_blocks(buf_blocks, rank, world)
            h = buf_blocks[0].size // 2
            lo_half = _alltoall_blocks([x[:h] for x in buf_blocks], rank, world)
            hi_half = _alltoall_blocks([x[h:] for x in buf_blocks], rank, world)
            return [np.concatenate([a, c]) for a, c in zip(lo_half, hi_half)]

        full_chunks = []
        for c in range(K):
            # exchange of chunk c: block (q * K + c) of S1 -> rank q; fact (1): the received blocks, in source-rank
            # order, concatenate to the natural [Wc][m][l] of this chunk's x' range
            got = ship([S1[(qd * K + c) * B:(qd * K + c + 1) * B] for qd in range(world)])
            R1 = np.concatenate(got).reshape((Wc, m, l), order="F")
            # phase 2 on the chunk's x' range (storage = natural order in this numpy model)
            xs = (rank * K + c) * Wc
            Bc = np.fft.fft(np.concatenate([R1, np.zeros_like(R1)], axis=1), axis=1)
            Bc = np.fft.fft(np.concatenate([Bc, np.zeros((Wc, 2 * m, l), complex)], axis=2), axis=2)
            Bc = Bc * G2[xs:xs + Wc]
            Bc = np.fft.ifft(Bc, axis=2)[:, :, :l]
            Bc = np.fft.ifft(Bc, axis=1)[:, :m, :]
            # exchange back; fact (2): slot p of the flat natural array of the chunk is rank p's z range
            flat = Bc.reshape(-1, order="F")
            back = ship([flat[p * B:(p + 1) * B] for p in range(world)])
            full_chunks.append(back)                             # back[src] = block (src * K + c) of S1
        # S1 layout on the way back: [source rank][chunk][Wc][m][lz] -> x' index (src * K + c) * Wc
        full = np.concatenate([full_chunks[c][src].reshape((Wc, m, lz), order="F") for src in range(world) for c in range(K)], axis=0)
        yl = b.reshape((n, m, l), order="F")[:, :, lo:hi] + k**2 * np.fft.ifft(full, axis=0)[:n]
        ref = o.apply_reduced(G2, nu, k, b, (n, m, l)).reshape((n, m, l), order="F")[:, :, lo:hi]
        q.put((rank, float(np.linalg.norm(yl - ref) / np.linalg.norm(ref))))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,K,halves", [((8, 4, 6), 1, False), ((16, 16, 16), 1, False), ((16, 8, 12), 2, False), ((16, 8, 12), 4, True)])
def test_slab_decomposition_world2_gloo(dims, K, halves):
    n, m, l = dims
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, m, l, 3.0, q, K, halves)) for r in range(2)]
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


def test_bench_self_launch_command_and_relay(tmp_path):
    # `python bench.py --gpus N` started directly must launch N ranks itself, BEFORE touching a GPU, and relay rank 0's
    # line: the launcher path is exercised here with a stand-in rank script (gloo, CPU) through the same code
    import subprocess
    import sys
    import bench as B
    cmd = B.launcher_command(["--gpus", "2", "--steps", "3"], 2)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and cmd[-5].endswith("bench.py")
    # relay: a fake two-rank job that prints one JSON line from rank 0 and noise from rank 1
    script = tmp_path / "fake_rank.py"
    script.write_text(
        "import os, json, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "r = dist.get_rank()\n"
        "print('noise from rank', r, flush=True)\n"
        "if r == 0: print(json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': dist.get_world_size()}), flush=True)\n"
        "dist.destroy_process_group()\n")
    out = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, %r); import bench as B\n"
                          "B.launcher_command = lambda argv, n: %r\n"
                          "B.self_launch([], 2)" % (os.path.dirname(os.path.abspath(B.__file__)),
                                                    cmd[:cmd.index(os.path.abspath(B.__file__))] + [str(script)])],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and '"n_gpus": 2' in lines[0]              # ONE line on stdout; the noise went to stderr
    assert "noise from rank" in out.stderr
    # a failing child must give a non-zero exit code and no line
    bad = tmp_path / "bad_rank.py"
    bad.write_text("raise SystemExit(3)\n")
    out = subprocess.run([sys.executable, "-c",
                          "import sys; sys.path.insert(0, %r); import bench as B\n"
                          "B.launcher_command = lambda argv, n: %r\n"
                          "B.self_launch([], 2)" % (os.path.dirname(os.path.abspath(B.__file__)),
                                                    cmd[:cmd.index(os.path.abspath(B.__file__))] + [str(bad)])],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and not out.stdout.strip()
