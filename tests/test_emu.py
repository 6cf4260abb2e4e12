"""CPU test of the gfx950 FFT line transform's index algebra: tests/emu/fft_emu.cpp compiles the
device header (fft_core.hpp) for the host, executes every factorisation thread by thread with
emulated LDS phases, and compares with a naive long-double DFT (forward, permutation table,
pruned forward/inverse round trip); the forms of the persistent fused pass (swizzled half-tile buffer,
exchange stores issued from inside the stages, the exchange through the lanes on an emulated wavefront) and the ticket ->
tile decode of its work queues; built with -fsanitize=address,undefined."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_line_transform_emulation(tmp_path):
    exe = str(tmp_path / "fft_emu")
    # AddressSanitizer + UBSan on the CPU build: the index algebra of every layout (padded, XOR-swizzled), the in-stage exchange
    # stores and the lane-exchange schedule run under them
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", exe,
                        os.path.join(ROOT, "tests", "emu", "fft_emu.cpp")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([exe], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    out = r.stdout.decode()
    assert "worst=" in out and "ticket decode / lane exchange checks failed: 0" in out and "+ lane exchanges between stages of equal radix" in out
