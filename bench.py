#!/usr/bin/env python3
"""Headline benchmark: fastconv applies/s on the 3D n=512^3 fp64 operator (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path, y = x + omega^2 * G * (nu .* x) (FastM3D `*`,
src/FastConvolution3D.jl:31-37 of the reference), on a synthetic contrast with the vectors
already resident in HBM.  N = 1: the whole 512^3 volume on one MI355X.  N > 1: the same volume
slab-partitioned over N GPUs (strong scaling) with the RCCL all-to-all pencil transposes, one
process per GPU.  Started under torch.distributed.run the process is one rank of the job; started
directly with --gpus N > 1 it launches the N ranks itself (torch.distributed.run as a child process,
before anything in this process touches the GPU) and relays rank 0's line.  `--single-process` runs
the N-GPU job from ONE host process instead (lsfc_plan_create_gv3d_multi: the form a single Julia
host uses).  Rank 0 prints ONE JSON line.

Before any throughput is reported the same build is checked against the CPU oracle on the bench's own
(nu, omega) recipe at n = 64 and n = 128 (n = 64 only when N > 1): relative l2 > 1e-10 aborts the run.
The CPU baseline (the numpy/scipy oracle, a port of the reference arithmetic -- the reference itself
is Julia and cannot run here) is timed on rank 0 at N = 1 only, on a bounded sample, and is never
the thing measured as `value`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BYTES_PER_POINT = 568.0         # 35 complex + 1 real per grid point per apply
PARITY_TOL = 1e-10              # BASELINE.json north_star


def synthetic_nu(n, lo, hi):
    """Sum of 8 Gaussians on z-planes [lo, hi) of the half-open unit box grid."""
    rng = np.random.default_rng(1234)
    cen = rng.uniform(-0.3, 0.3, size=(8, 3))
    amp = rng.uniform(-0.3, 0.3, size=8)
    beta = rng.uniform(20, 80, size=8)
    h = 1.0 / n
    x = -0.5 + h * np.arange(n)
    z = x[lo:hi]
    out = np.zeros((hi - lo, n, n))          # [z][y][x]  == column-major (x fastest) flat order
    for c, a, b in zip(cen, amp, beta):
        gx = np.exp(-b * (x - c[0]) ** 2)
        gy = np.exp(-b * (x - c[1]) ** 2)
        gz = np.exp(-b * (z - c[2]) ** 2)
        out += a * gz[:, None, None] * gy[None, :, None] * gx[None, None, :]
    return out.reshape(-1)


def bench_vector(n, lo, hi):
    """x = g1 + i g2, planes [lo, hi); the same on every rank layout"""
    rng = np.random.default_rng(20250224)
    v = rng.standard_normal(2 * n ** 3).view(np.complex128) if n <= 128 else None
    if v is not None:
        return v[lo * n * n:hi * n * n].copy()
    rng = np.random.default_rng(20250224 + lo)
    return rng.standard_normal(2 * (hi - lo) * n * n).view(np.complex128)


# --------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with N > 1 and no torchrun environment
# --------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launcher_command(argv, n):
    """the child command that runs this script as n ranks on one node (no GPU is touched by the parent)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, n):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    # a rank that blocks for ever in a collective would otherwise hold the whole job until the driver's own limit: the launcher is
    # given a deadline (the ranks carry their own watchdog as well, see main) and a late job is reported as failed, with its output
    deadline = float(os.environ.get("LSFC_BENCH_DEADLINE_S", "1500"))
    try:
        proc = subprocess.run(launcher_command(argv, n), env=env, stdout=subprocess.PIPE, text=True, timeout=deadline)
    except subprocess.TimeoutExpired as e:
        out = e.stdout if isinstance(e.stdout, str) else (e.stdout or b"").decode(errors="replace")
        print(out, file=sys.stderr)
        print(f"[bench] the {n}-rank job did not finish within {deadline:.0f} s (a blocked collective?); no throughput reported", file=sys.stderr, flush=True)
        raise SystemExit(3)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if proc.returncode != 0 or line is None:
        print(f"[bench] the {n}-rank job failed (exit code {proc.returncode})", file=sys.stderr, flush=True)
        raise SystemExit(proc.returncode or 1)
    print(line, flush=True)
    raise SystemExit(0)


# --------------------------------------------------------------------------------------------------------------
# CPU baseline and parity gate (the only users of oracle/ in this file)
# --------------------------------------------------------------------------------------------------------------
def cpu_baseline(n_target, nu, xvec):
    """Oracle (port of the reference arithmetic, reduced-2n variant -- the literal (4n)^3 arrays are 137 GB each at
    n=512) timed on this box's host cores on a bounded sample of the SAME workload: the bench's contrast and vector at
    the real size when the host can hold it (~70 GB of work arrays at n=512), else n=256 measured and scaled by
    N log N.  The symbol handed to the timed apply is a cheap deterministic array of the right shape (evaluating the
    true 1024^3 symbol on the host takes minutes and its values do not change the FFT time); results are NOT compared
    here -- parity of this build is established by the gate above and by tests/."""
    from oracle import lsfc_oracle as o
    cores = os.cpu_count() or 1
    avail_gb = 0.0
    try:
        import psutil
        avail_gb = psutil.virtual_memory().available / 1e9
    except Exception:
        pass

    def timed(n, nu_n, x_n, max_reps, budget_s):
        a = np.linspace(0.1, 1.0, 2 * n)
        G2 = (a[:, None, None] + 1j * a[None, :, None]) * a[None, None, :]
        times, t_all = [], time.time()
        while len(times) < max_reps and (not times or time.time() - t_all + times[-1] < budget_s):
            t0 = time.time()
            o.apply_reduced(G2, nu_n, float(n), x_n, (n, n, n))
            times.append(time.time() - t0)
        return times

    need_gb = 5.5 * 16 * (2 * n_target) ** 3 / 1e9
    n, scale = n_target, 1.0
    if not (cores >= 32 and avail_gb > need_gb) and n_target > 256:
        n = 256
    if n != n_target:
        nu, xvec = synthetic_nu(n, 0, n), bench_vector(n, 0, n)
        Nt, Ns = (2 * n_target) ** 3, (2 * n) ** 3
        scale = (Nt * np.log2(Nt)) / (Ns * np.log2(Ns))
    try:
        times = timed(n, nu, xvec, 4, 80.0)        # 1 warm-up + 3 timed at ~14 s each on the GPU box's 256 cores
    except MemoryError:
        return None
    best = min(times[1:]) if len(times) > 1 else times[0]
    note = (f"oracle reduced-2n apply (scipy.fft, workers={cores}), 3D n={n}, bench contrast and vector, stand-in symbol of the right shape "
            "(its values do not change the FFT time), "
            + (f"1 warm-up + {len(times) - 1} timed, best-of" if len(times) > 1 else "one apply, no warm-up (a second apply did not fit the 80-s sample budget)"))
    if n != n_target:
        note += f"; extrapolated to n={n_target} by N log N (x{scale:.2f}): the host cannot hold an n={n_target} apply"
    return {"value": 1.0 / (best * scale), "unit": "applies/s", "cores": cores, "kind": "port", "sample": note,
            "measured_s_per_apply_sample": best, "all_sample_s": times, "sample_n": n}


def parity_gate(lsfc, sizes, rank, world, local_rank, dist, multi_devices=None):
    """the bench's (nu, omega = 1/h) recipe at small n on THIS build and device(s) vs the CPU oracle"""
    from oracle import lsfc_oracle as o
    worst = {}
    for n in sizes:
        h = 1.0 / n
        omega = 1.0 / h
        lo, hi = rank * n // world, (rank + 1) * n // world
        b = bench_vector(n, 0, n)
        nu = synthetic_nu(n, 0, n)
        if multi_devices is not None:
            from fast_solver_lippmann_schwinger_amd.distributed import MultiDeviceFastM3D
            M = MultiDeviceFastM3D(n, h, omega, nu, devices=multi_devices)
        elif world == 1:
            xg = -0.5 + h * np.arange(n)
            M = lsfc.buildFastConvolution3D(xg, xg, xg, None, None, None, h, omega, nu, device=local_rank)
        else:
            from fast_solver_lippmann_schwinger_amd.distributed import build_distributed_3d
            M = build_distributed_3d(n, h, omega, nu[lo * n * n:hi * n * n], rank, world, local_rank)
        y = M * b[lo * n * n:hi * n * n]
        M.close()
        G2 = o.reduced_symbol_gv3d(n, n, n, n * h, omega, patch_singular=False)
        ref = o.apply_reduced(G2, nu, omega, b, (n, n, n))[lo * n * n:hi * n * n]
        err = float(np.linalg.norm(y - ref) / np.linalg.norm(ref))
        if dist is not None:
            import torch
            t = torch.tensor([err], dtype=torch.float64, device=f"cuda:{local_rank}")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            err = float(t.item())
        worst[f"n{n}"] = err
        if not err <= PARITY_TOL:
            raise SystemExit(f"[bench] PARITY GATE FAILED at n={n}: relative l2 {err:.3e} > {PARITY_TOL:g} vs the CPU oracle; no throughput reported")
    return worst


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=int(os.environ.get("LSFC_BENCH_N", 512)))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-gate", action="store_true", help="skip the oracle check (profiling runs only; the line then says so)")
    ap.add_argument("--force-dist", action="store_true", help="use the distributed plan even with one rank (rehearsal of the multi-GPU path)")
    ap.add_argument("--single-process", action="store_true", help="drive all --gpus devices from this one process (multi-device plan)")
    ap.add_argument("--devices", type=str, default="", help="with --single-process: comma-separated device ids, one per rank; a device may repeat "
                    "(logical ranks on one GPU, peer-copy transport): a rehearsal of the path where fewer GPUs are visible, not a measurement")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "RANK" not in os.environ and not args.single_process:
        self_launch(sys.argv[1:], args.gpus)             # never returns; nothing above has touched the GPU
    if "RANK" in os.environ and args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")

    # watchdog of a multi-rank run: a rank stuck in a collective (first execution of the RCCL path on a new node) ends the job with a
    # message instead of idling until somebody's outer limit; the thread never touches the GPU
    if world > 1:
        import threading
        limit = float(os.environ.get("LSFC_BENCH_WATCHDOG_S", "1200"))

        def _watchdog():
            time.sleep(limit)
            print(f"[bench] rank {rank}: no result after {limit:.0f} s -- giving up (blocked collective or a far slower exchange than "
                  "modelled); no throughput reported", file=sys.stderr, flush=True)
            os._exit(3)
        threading.Thread(target=_watchdog, daemon=True).start()

    import torch
    import fast_solver_lippmann_schwinger_amd as lsfc

    n = args.n
    N = n ** 3
    h = 1.0 / n
    omega = 1.0 / h                      # omega = 1/h = 512, min|s-k| = 9.3e-5
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    dist = None
    if world > 1 or (args.force_dist and "RANK" in os.environ):
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    parity = None
    multi_devs = None
    if args.single_process and args.gpus > 1:
        multi_devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
        if len(multi_devs) != args.gpus:
            raise SystemExit(f"--devices lists {len(multi_devs)} ids for --gpus {args.gpus}")
    if not args.no_parity_gate:
        parity = parity_gate(lsfc, [64, 128] if (world == 1 and multi_devs is None) else [64], rank, world, local_rank,
                             dist if world > 1 else None, multi_devs)

    nu_host = None
    multi = None
    if args.single_process and args.gpus > 1:
        from fast_solver_lippmann_schwinger_amd.distributed import MultiDeviceFastM3D
        nu_host = synthetic_nu(n, 0, n)
        multi = MultiDeviceFastM3D(n, h, omega, nu_host, devices=multi_devs)
        world_eff = args.gpus
    elif world == 1 and not args.force_dist:
        nu_host = synthetic_nu(n, 0, n)
        x = -0.5 + h * np.arange(n)
        M = lsfc.buildFastConvolution3D(x, x, x, None, None, None, h, omega, nu_host, device=local_rank)
        nloc = N
        world_eff = 1
    else:
        from fast_solver_lippmann_schwinger_amd.distributed import build_distributed_3d
        lo, hi = rank * n // world, (rank + 1) * n // world
        M = build_distributed_3d(n, h, omega, synthetic_nu(n, lo, hi), rank, world, local_rank)
        nloc = (hi - lo) * n * n
        world_eff = world

    if multi is not None:
        out = multi.bench(args.steps, args.warmup)
        elapsed, stages = out["elapsed_s"], out["stages"]
        pipeline, padded = "pruned-hip", list(multi.padded_dims)
        nloc = N // args.gpus
    else:
        lo = rank * n // world_eff
        xb = torch.from_numpy(bench_vector(n, lo, lo + nloc // (n * n))).to(dev)
        yb = torch.empty_like(xb)

        def sync():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()

        for _ in range(args.warmup):
            M.mul_(yb, xb)
        M.synchronize()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            M.mul_(yb, xb)
        M.synchronize()
        sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        # per-kernel roofline: HIP events on the plan's stream around every stage of one apply.  With more than one
        # rank the stages contain collectives, so a failure must take the whole job down (no per-rank try/except:
        # a rank that skipped them would leave the others blocked in ncclRecv).
        if world_eff > 1:
            stages = lsfc.profile_apply(M, xb, yb, reps=5)
        else:
            try:
                stages = lsfc.profile_apply(M, xb, yb, reps=5)
            except Exception as e:                   # never lose the headline line to the optional per-stage profile
                print(f"[bench] per-stage profile unavailable: {e}", file=sys.stderr, flush=True)
                stages = [("apply (whole; per-stage profile unavailable)", elapsed / args.steps * 1e3, BYTES_PER_POINT * nloc)]
        pipeline, padded = M.pipeline, list(M.padded_dims)

    ms_per_step = elapsed / args.steps * 1e3
    value = args.steps / elapsed

    # dominant COMPUTE kernel (the un-overlapped all-to-all stages of the multi-GPU profile are listed, not ranked:
    # they are bounded by the xGMI links, not by HBM)
    dom = max((s for s in stages if not s[0].startswith("alltoall")), key=lambda s: s[1])
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath) and world_eff == 1 and n == 512:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from csrc_hash import csrc_hash
            rec = json.load(open(tpath))
            if rec.get("_csrc_sha256") == csrc_hash(ROOT):
                traffic = rec.get(dom[0])
                traffic_src = ("profiles/traffic_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/one_apply.py 512 "
                               "(tools/pmc_traffic.sh) on the kernel sources of THIS build (sha256 of csrc/ matches; per launch; not collected "
                               "in this run -- counters need the profiler)")
            else:
                traffic_src = ("null: profiles/traffic_latest.json was measured on other kernel sources (sha256 of csrc/ differs) -- re-run "
                               "tools/pmc_traffic.sh on this build")
        except Exception as e:
            traffic, traffic_src = None, f"null: {e}"
    roofline = {"bound": "hbm", "kernel": dom[0], "achieved": dom[2] / (dom[1] * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": dom[2] / (dom[1] * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                "traffic_over_algorithmic": (traffic / dom[2]) if traffic else None,
                "algorithmic_bytes_per_launch": dom[2], "avg_launch_ms": dom[1],
                "stages": [{"kernel": s[0], "ms": s[1], "GBps": s[2] / (s[1] * 1e-3) / 1e9} for s in stages]}
    whole = BYTES_PER_POINT * N / (ms_per_step * 1e-3) / 1e9

    out = {"metric": "fastconv applies/sec, 3D n=512^3 fp64 (FastM3D apply)" if n == 512 else f"fastconv applies/sec, 3D n={n}^3 fp64",
           "value": value, "unit": "applies/s", "n_gpus": world_eff, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"3D n={n}^3 complex fp64 operator apply y = x + w^2 G*(nu.*x), Greengard-Vico truncated-kernel symbol, "
                                  f"omega=1/h={omega:g}, sum-of-8-Gaussians contrast, vectors resident in HBM",
                      "n": n, "omega": omega, "pipeline": pipeline, "padded_grid": padded,
                      "parallelism": "single GPU" if world_eff == 1 else
                                     (f"z-slabs over {world_eff} ranks, 2 all-to-all slab transposes per apply, "
                                      + ((f"ONE host process driving devices {multi.devices}" + (" -- REHEARSAL: ranks share a GPU, not a multi-GPU measurement" if len(set(multi.devices)) < len(multi.devices) else ""))
                                         if multi is not None else "one process per GPU, RCCL send/recv"))},
           "achieved_algorithmic_GBps_per_gpu": whole / world_eff, "hbm_roofline_frac_whole_apply": whole / world_eff / HBM_PEAK_GBPS,
           "parity_rel_l2": parity if parity is not None else "gate skipped (--no-parity-gate)",
           "parity_tol": PARITY_TOL,
           "roofline": roofline}
    if world_eff > 1:
        # the two slab transposes of an apply move 2 * 2N complex in total; rank p sends 2N*16 B*(P-1)/P^2 per transpose
        # over its P-1 direct xGMI links (one link per GPU pair).  A model, not a measurement: see DESIGN.md section 5.
        per_link = 2 * (2.0 * N * 16.0) / world_eff ** 2
        out["link_model"] = {"bytes_per_link_per_direction_per_apply": per_link,
                             "ms_at_76.8_GBps_per_direction": per_link / 76.8e9 * 1e3,
                             "note": "xGMI time floor of the two all-to-all transposes at the nominal per-direction link rate; "
                                     "the apply cannot be faster than this however well it overlaps"}
        # what the transport saw: the un-overlapped exchange stages of the per-stage profile
        ex = {s[0]: s[1] for s in stages if s[0].startswith("alltoall")}
        out["exchange"] = {"nranks": world_eff, "transport": (multi.transport if multi is not None else "rccl send/recv, pairwise schedule"),
                           "unoverlapped_ms": ex,
                           "achieved_GBps_per_link_per_direction": {k: (per_link / 2) / (v * 1e-3) / 1e9 for k, v in ex.items() if v > 0},
                           "note": "each exchange ships bytes_per_link_per_direction_per_apply / 2 over every one of the P-1 links of a GPU; "
                                   "production applies overlap the exchanges with the y/z passes (K pipeline chunks)"}
    if world_eff == 1 and rank == 0 and not args.no_cpu_baseline:
        cb = cpu_baseline(n, nu_host, xb.cpu().numpy())
        if cb is not None:
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = value / cb["value"]
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
