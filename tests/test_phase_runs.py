"""Exact jump-ahead of the float phase accumulator (phase_runs.h, used by the GPU subtraction K7):
every sample's phase rebuilt from ~200 runs must equal the reference's serial float walk
(wsprd.c:340-351) bit for bit -- random signals, sign changes, ties, stuck and tiny increments."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TWOPIDT = 2.0 * np.pi / 375.0


@pytest.fixture(scope="module")
def pr(tmp_path_factory):
    so = tmp_path_factory.mktemp("pr") / "phase_runs_check.so"
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(so),
                    os.path.join(ROOT, "tests", "helpers", "phase_runs_check.cpp")], check=True)
    L = C.CDLL(str(so))
    L.phase_runs_check.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    L.phase_runs_chained_diff.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.phase_runs_chained_diff.restype = C.c_long
    return L


def _check(pr, dphi, sps_log2=8, max_runs=0):
    d = np.ascontiguousarray(dphi, np.float32)
    bad = C.c_long(-1)
    nr = pr.phase_runs_check(d.ctypes.data, d.size, sps_log2, C.addressof(bad), None, max_runs)
    # the chained construction (a wave of the GPU kernel probes 64 symbols at once) gives the same tables
    for width in (64, 5):
        assert pr.phase_runs_chained_diff(d.ctypes.data, d.size, sps_log2, max_runs, width) == 0, (width, d[:4])
    return nr, bad.value


def _wspr_dphi(rng, f0, drift):
    cs = rng.integers(0, 4, 162)
    i = np.arange(162, dtype=np.float64)
    arg = np.float64(np.float32(f0)) + (np.float64(np.float32(drift)) / 2.0) * (i - 81.0) / 81.0 + (cs - 1.5) * 375.0 / 256.0
    return (TWOPIDT * arg).astype(np.float32)


def test_wspr_signals(pr):
    rng = np.random.default_rng(1)
    runs = []
    for t in range(400):
        f0 = rng.uniform(-110, 110) if t % 4 else rng.uniform(-3, 3)      # small |f0|: the increment changes sign
        drift = float(rng.integers(-4, 5))
        nr, bad = _check(pr, _wspr_dphi(rng, f0, drift))
        assert bad == 0 and 162 <= nr <= 512, (t, f0, drift, nr, bad)
        runs.append(nr)
    assert np.mean(runs) < 260          # about one run per symbol plus the binade crossings
    print("runs per signal: mean %.0f, max %d" % (np.mean(runs), max(runs)))
    # a table that is too small is reported and the per-symbol fallback is exact as well
    nr, bad = _check(pr, _wspr_dphi(rng, 37.5, 1.0), max_runs=100)
    assert nr == -1 and bad == 0


def test_adversarial_increments(pr):
    rng = np.random.default_rng(2)
    cases = []
    # few significant bits: long stretches of exact ties
    for k in range(-12, 3):
        cases.append(np.full(162, 2.0 ** k, np.float32))
        cases.append(np.full(162, -3 * 2.0 ** k, np.float32))
        cases.append(np.where(np.arange(162) % 2 == 0, 5 * 2.0 ** k, -(2.0 ** k)).astype(np.float32))
    # increments far below the accumulator's spacing (stuck), zero, alternating signs around zero
    cases.append(np.concatenate([np.full(40, 3.7, np.float32), np.full(122, 1e-9, np.float32)]))
    cases.append(np.zeros(162, np.float32))
    cases.append(np.concatenate([np.full(81, 0.731, np.float32), np.full(81, -0.731, np.float32)]))
    cases.append(np.tile(np.array([1.25, -1.25, 0.0, 2.5e-4], np.float32), 41)[:162])
    # random magnitudes over many binades, random signs
    for _ in range(200):
        mag = 10.0 ** rng.uniform(-6, 1, 162)
        cases.append((mag * rng.choice([-1.0, 1.0], 162)).astype(np.float32))
    for _ in range(100):
        cases.append((rng.integers(1, 64, 162) * 2.0 ** float(rng.integers(-14, -2))).astype(np.float32))
    worst = 0
    for c in cases:
        nr, bad = _check(pr, c)
        assert bad == 0, (c[:4], nr, bad)
        worst = max(worst, nr)
        assert nr == -1 or nr <= 512      # (an overflowing walk was checked through the fallback)
    assert worst > 0
