"""Kernel-level unit checks that need no oracle: the register-resident lag scan (demod_lagsys_kernel: samples handed from
lane to lane by DPP, lag 32 taken from the next symbol's lag 0, edge waves walked sample by sample) must give the
amplitudes of the LDS-tiled kernel that computes every (symbol, lag) on its own, bit for bit, for candidates inside the
record and hanging over either end of it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _tool(name):
    exe = os.path.join(ROOT, "tools", name + ".bin")
    src = os.path.join(ROOT, "tools", name + ".hip")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                        "-fno-fast-math", "-I", os.path.join(ROOT, "rtlsdr-wsprd_amd", "csrc", "kernels"), src, "-o", exe],
                       check=True, capture_output=True)
    return exe


@pytest.mark.parametrize("samples", [45000, 30000])
def test_register_resident_lag_scan_equals_the_tiled_kernel(samples):
    r = subprocess.run([_tool("lagsys_check"), str(samples)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "lagsys == tile kernel bit for bit" in r.stdout, r.stdout[-2000:] + r.stderr[-500:]


def test_dpp_wave_shift_semantics():
    """v_mov_b32_dpp wave_shl:1 -- lane l takes lane l + 1, lane 63 keeps `old`: what the lag scan's hand-over relies on."""
    r = subprocess.run([_tool("dpp_probe")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("wave_shl1:")][0]
    assert [int(x) for x in line.split()[1:]] == list(range(1, 64)) + [1063]
