"""bench.py pieces that can be checked without a GPU: the algorithmic byte counts behind the roofline
(SURVEY 8d), the CPU-share helper, and the command-line contract."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_algorithmic_bytes_are_the_surveys():
    import bench
    assert bench.K1_BYTES == 360000 + 4 * 417 * 347 == 938796          # IQ read once + ps rows 48..464 written
    assert bench.STAGE_BYTES == 1517592                                   # K1 + (ps read once by K2/K3)
    assert bench.HBM_PEAK_GBS == 8000.0
    assert bench.NS == 45000


def test_cpu_share_is_positive_and_bounded():
    import bench
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_command_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--inflight"):
        assert flag in out.stdout
