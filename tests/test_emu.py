"""CPU test of the gfx950 FFT line transform's index algebra: tests/emu/fft_emu.cpp compiles the
device header (fft_core.hpp) for the host, executes every factorisation thread by thread with
emulated LDS phases, and compares with a naive long-double DFT (forward, permutation table,
pruned forward/inverse round trip)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fft_line_transform_emulation(tmp_path):
    exe = str(tmp_path / "fft_emu")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "emu", "fft_emu.cpp")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([exe], capture_output=True, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert "worst=" in r.stdout.decode()
