"""CPU, world_size = 2, gloo: the z-slab decomposition of the operator (csrc/dist.hip) run over real torch.distributed
ranks -- the passes restated in numpy, the HOST LOGIC (chunk planning, block offsets, the message list of every exchange)
taken from the product's own csrc/dist_schedule.hpp through ctypes.  Checks the two layout facts the HIP path relies on --
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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build_schedule_lib(tmpdir):
    """csrc/dist_schedule.hpp -- the message list dist.hip posts through RCCL -- as a shared object for ctypes"""
    import subprocess
    so = os.path.join(tmpdir, "libdsched.so")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-shared", "-fPIC", "-DLSFC_SCHED_SHARED", "-o", so,
                        os.path.join(ROOT, "tests", "emu", "dist_schedule_test.cpp")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    return so


class _Schedule:
    def __init__(self, so):
        import ctypes as C
        self.C, self.lib = C, C.CDLL(so)

    def plan_chunks(self, Lx, P, req):
        C = self.C
        W, K, Wc = C.c_int(), C.c_int(), C.c_int()
        self.lib.lsfc_sched_plan_chunks(Lx, P, req, C.byref(W), C.byref(K), C.byref(Wc))
        return W.value, K.value, Wc.value

    def messages(self, rank, P, K, c, back, part, B):
        C = self.C
        cap = 2 * P
        peer, send, ins1 = (C.c_int * cap)(), (C.c_int * cap)(), (C.c_int * cap)()
        off, cnt = (C.c_int64 * cap)(), (C.c_int64 * cap)()
        n = self.lib.lsfc_sched_exchange(rank, P, K, c, int(back), part, C.c_int64(B), cap, peer, send, ins1, off, cnt)
        assert n == 2 * P
        return [(peer[i], bool(send[i]), bool(ins1[i]), off[i], cnt[i]) for i in range(n)]


def _exchange(sched, S1, R1, rank, world, K, c, back, part, B):
    """one exchange exactly as dist.hip posts it: the local copy, then the grouped point-to-point messages of the
    pairwise schedule (csrc/dist_schedule.hpp), here over gloo"""
    msgs = sched.messages(rank, world, K, c, back, part, B)
    buf = lambda m: (S1 if m[2] else R1)
    (_, _, _, so, sc), (_, _, _, do, dc) = msgs[0], msgs[1]
    buf(msgs[1])[do:do + dc] = buf(msgs[0])[so:so + sc]
    reqs, landing = [], []
    for m in msgs[2:]:
        peer, send, _, off, cnt = m
        if send:
            t = torch.from_numpy(buf(m)[off:off + cnt].view(np.float64).copy())
            reqs.append(dist.isend(t, dst=peer))
        else:
            t = torch.empty(2 * cnt, dtype=torch.float64)
            reqs.append(dist.irecv(t, src=peer))
            landing.append((m, t))
    for r in reqs:
        r.wait()
    for m, t in landing:
        buf(m)[m[3]:m[3] + m[4]] = t.numpy().view(np.complex128)


def _worker(rank, world, port, n, m, l, k, q, Kreq, edges, so):
    """The slab-distributed apply over real gloo ranks with the HOST LOGIC OF dist.hip: chunk planning, block offsets in the
    message buffers S1 [dest rank][chunk][Wc][m][lz] / R1 [chunk][source rank][Wc][m][lz] and the message list of every
    exchange come from csrc/dist_schedule.hpp through ctypes; the passes are numpy.  edges: the pipeline ends ship the two
    z halves of their blocks as separate messages (dist_convolve_dev's split of the first exchange in and the last back)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sched = _Schedule(so)
        rng = np.random.default_rng(5)                          # same data on every rank
        G2 = rng.standard_normal((2 * n, 2 * m, 2 * l)) + 1j * rng.standard_normal((2 * n, 2 * m, 2 * l))
        nu = rng.uniform(-0.3, 0.3, n * m * l)
        b = rng.standard_normal(n * m * l) + 1j * rng.standard_normal(n * m * l)
        lo, hi = slab_range(l, rank, world)
        lz, Lx = hi - lo, 2 * n
        W, K, Wc = sched.plan_chunks(Lx, world, Kreq)
        B = Wc * m * lz                                          # elements of one (rank, chunk) block
        xl = (nu * b).reshape((n, m, l), order="F")[:, :, lo:hi]
        # phase 1: x pass on own planes, written packed per (destination rank, chunk): storage index s -> block s // Wc
        A = np.fft.fft(np.concatenate([xl, np.zeros_like(xl)], axis=0), axis=0)            # [Lx][m][lz]
        S1 = np.concatenate([A[blk * Wc:(blk + 1) * Wc].reshape(-1, order="F") for blk in range(world * K)])
        R1 = np.full(K * world * B, np.nan + 0j)
        for c in range(K):
            if edges and c == 0:
                _exchange(sched, S1, R1, rank, world, K, c, False, 0, B)
                _exchange(sched, S1, R1, rank, world, K, c, False, 1, B)
            else:
                _exchange(sched, S1, R1, rank, world, K, c, False, -1, B)
        S1[:] = np.nan                                           # everything that comes back must be written by a message
        for c in range(K):
            # fact (1): the blocks received from all ranks, in source-rank order, ARE the natural [Wc][m][l] of the chunk
            Rc = R1[c * world * B:(c + 1) * world * B].reshape((Wc, m, l), order="F")
            xs = (rank * K + c) * Wc
            Bc = np.fft.fft(np.concatenate([Rc, np.zeros_like(Rc)], axis=1), axis=1)
            Bc = np.fft.fft(np.concatenate([Bc, np.zeros((Wc, 2 * m, l), complex)], axis=2), axis=2)
            Bc = Bc * G2[xs:xs + Wc]
            Bc = np.fft.ifft(Bc, axis=2)[:, :, :l]
            Bc = np.fft.ifft(Bc, axis=1)[:, :m, :]
            # fact (2): slot p of the flat natural array of the chunk is rank p's z range
            R1[c * world * B:(c + 1) * world * B] = Bc.reshape(-1, order="F")
            if edges and c == K - 1:
                _exchange(sched, S1, R1, rank, world, K, c, True, 0, B)
                _exchange(sched, S1, R1, rank, world, K, c, True, 1, B)
            else:
                _exchange(sched, S1, R1, rank, world, K, c, True, -1, B)
        assert np.isfinite(S1).all()
        # S1 on the way back: [source rank][chunk][Wc][m][lz] -> x' index (src * K + c) * Wc, the layout the inverse x pass reads
        full = np.concatenate([S1[(src * K + c) * B:(src * K + c + 1) * B].reshape((Wc, m, lz), order="F") for src in range(world) for c in range(K)], axis=0)
        yl = b.reshape((n, m, l), order="F")[:, :, lo:hi] + k**2 * np.fft.ifft(full, axis=0)[:n]
        ref = o.apply_reduced(G2, nu, k, b, (n, m, l)).reshape((n, m, l), order="F")[:, :, lo:hi]
        q.put((rank, float(np.linalg.norm(yl - ref) / np.linalg.norm(ref)), K))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dims,K,edges", [((8, 4, 6), 1, False), ((16, 16, 16), 1, False), ((16, 8, 12), 2, True), ((32, 8, 8), 4, True)])
def test_slab_decomposition_world2_gloo(dims, K, edges, tmp_path):
    n, m, l = dims
    so = _build_schedule_lib(str(tmp_path))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, m, l, 3.0, q, K, edges, so)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res = {r: e for r, e, _ in got}
    assert res[0] < 1e-13 and res[1] < 1e-13, res
    assert all(kk == (K if (2 * n // 2) % K == 0 and (2 * n // 2 // K) % 8 == 0 else kk) for _, _, kk in got)


def test_exchange_schedule_for_2_4_8_ranks(tmp_path):
    # the message list dist.hip posts (csrc/dist_schedule.hpp) for P = 2, 4, 8, K = 1, 2, 4, both directions, whole blocks and
    # z halves: every send matched by one receive of equal size, no overlap, the buffers tiled exactly, all peers busy in
    # every step of the pairwise schedule -- under AddressSanitizer / UBSan
    import subprocess
    exe = str(tmp_path / "dsched")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(ROOT, "tests", "emu", "dist_schedule_test.cpp")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([exe], capture_output=True, timeout=300)
    assert r.returncode == 0 and b"failures: 0" in r.stdout, r.stdout.decode()[-2000:]


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
