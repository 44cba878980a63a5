"""examples/wspr_host.c: the reference application's decoder-side modes (playback -r, self-test -t, a raw receiver stream
through the session) as a C program over include/wspr_mi355x.h -- the reference's own language above the C ABI.
CPU: it compiles as C11 with -Wall -Wextra -Werror against the public header and links the product library; without a
GPU it refuses loudly.  GPU: its stdout is the reference-held line of documentation/bug-fix/REPORT.md:202 for the
reference's own recording, several files give the same spots in one batch call, the self-test passes
(rtlsdr_wsprd.c:782-788's check), and two slots of raw 2.4 Msps bytes come out as two timestamped spots."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "wspr_host")
GOLDEN = os.path.join(ROOT, "tests", "golden", "refSignalSnr0dB.iq")
REPORT_LINE = "Spot :  -0.07   0.01 144.490550  0    K1JT   FN20 20"          # reference documentation/bug-fix/REPORT.md:202


@pytest.fixture(scope="module")
def exe():
    subprocess.run(["make", "-s", "-B", "-C", os.path.join(ROOT, "examples")], check=True)
    return EXE


def run(exe, *args, cwd=None, stdin=None):
    return subprocess.run([exe, *args], capture_output=True, text=True, cwd=cwd, stdin=stdin, timeout=600)


def test_c_host_builds_warning_free_and_checks_its_arguments(exe):
    r = run(exe)
    assert r.returncode == 2 and "use:" in r.stderr
    r = run(exe, "-r")
    assert r.returncode == 2
    r = run(exe, "-x")
    assert r.returncode == 2
    out = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libwspr_mi355x.so" in out and "oracle" not in out and "lab" not in out


def test_c_host_refuses_loudly_without_a_gpu(exe):
    import rtlsdr_wsprd_amd as w
    if w.lib().wspr_device_ready() == 1:
        pytest.skip("a GPU is present")
    r = run(exe, "-r", GOLDEN)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr and "Spot" not in r.stdout


@pytest.mark.gpu
def test_c_host_playback_prints_the_reference_held_line(exe):
    r = run(exe, "-f", "144489000", "-r", GOLDEN)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0] == "Number of samples: 45000"
    assert lines[1] == "        SNR      DT        Freq Dr    Call    Loc Pwr"
    assert lines[2:] == [REPORT_LINE]


@pytest.mark.gpu
def test_c_host_self_test_and_batch_playback(exe, tmp_path):
    r = run(exe, "-t", cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Self-test SUCCESS!" in r.stdout and "K1JT" in r.stdout and "FN20" in r.stdout
    made = tmp_path / "selftest.iq"
    assert made.stat().st_size == 45000 * 8                       # writeRawIQfile's format, rtlsdr_wsprd.c:595-617
    # the file it wrote and the reference's recording, each alone and both in one batch call: the same spot lines
    alone = [run(exe, "-f", "144489000", "-r", f).stdout.splitlines()[2:] for f in (GOLDEN, str(made))]
    both = run(exe, "-f", "144489000", "-r", GOLDEN, str(made))
    assert both.returncode == 0, both.stderr
    out = both.stdout.splitlines()
    cut = out.index(str(made))
    assert out[0] == GOLDEN and out[3:cut] == alone[0] == [REPORT_LINE]
    assert out[cut + 3:] == alone[1] and len(alone[1]) == 1 and "K1JT" in alone[1][0]
    r = run(exe, "-r", str(tmp_path / "nothing.wav"))
    assert r.returncode == 2 and "Not a valid extension" in r.stderr


@pytest.mark.gpu
def test_c_host_raw_stream_through_the_session(exe, tmp_path):
    import torch
    import bench
    dev = torch.device("cuda", 0)
    raw, expected = bench.synth_raw_gpu(2, 97531, dev, snr_db=-15.0)
    path = tmp_path / "two_slots.u8"
    other = tmp_path / "one_slot.u8"
    with open(path, "wb") as fh:
        for s in range(2):
            fh.write(raw[s].cpu().numpy().tobytes())
    with open(other, "wb") as fh:
        fh.write(raw[1].cpu().numpy().tobytes())
    del raw
    torch.cuda.empty_cache()
    t0 = 1700000040                                              # 2023-11-14 22:14:00 UTC, an even minute
    r = run(exe, "-f", "14095600", "-i", str(path), "-T", str(t0))
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert len(lines) == 2, r.stdout
    for line, stamp, (msg,) in zip(lines, ("2023-11-14 22:14z", "2023-11-14 22:16z"), expected):
        call, loc, pwr = msg.split()
        assert line.startswith("Spot :  " + stamp), line
        assert line.split()[-3:] == [call, loc, pwr.lstrip("0") or "0"] or line.split()[-3:] == [call, loc, pwr], line
    # two receivers: callbacks in turn, both completed buffers decoded together at the even minute
    r3 = run(exe, "-f", "14095600", "-i", str(path), "-i", str(other), "-T", str(t0))
    assert r3.returncode == 0, r3.stderr
    l3 = r3.stdout.splitlines()
    assert len(l3) == 4, r3.stdout
    assert l3[0] == "[0] " + lines[0] and l3[2] == "[0] " + lines[1]
    assert l3[1].startswith("[1] Spot :  2023-11-14 22:14z") and l3[1].split()[-3:] == lines[1].split()[-3:]
    assert l3[3].startswith("[1] Signal too short")
    # the same bytes on stdin
    with open(path, "rb") as fh:
        r2 = run(exe, "-f", "14095600", "-i", "-", "-T", str(t0), stdin=fh)
    assert r2.returncode == 0 and r2.stdout == r.stdout
