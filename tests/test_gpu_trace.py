"""The production K4/K5 kernels per candidate (round-2 verdict, weak #2): the exported sync_and_demodulate() drives the
GENERAL demod kernel, so its parity tests say nothing about demod_lagsys / demod_drift / freq_scalar / freq_drift / the
tiled ladder kernel, and spot-level tests only see candidates that decode.  wspr_decode_batch_trace() records what those
kernels produced for every candidate the reference's loop enters; it must equal the oracle's trace field for field.
The library's environment switches are read once per process, so the alternatives that are kept in the tree are run
through the same check in subprocesses."""
import os
import subprocess
import sys

import pytest

import oracle_lib as ol
import trace_parity as tp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def w():
    import rtlsdr_wsprd_amd as mod
    assert mod.lib().wspr_device_ready() == 1
    return mod


@pytest.mark.parametrize("opts", [dict(), dict(quickmode=1), dict(subtraction=0), dict(npasses=1), dict(npasses=3)])
def test_trace_of_the_parity_batch_equals_oracle(w, opts):
    I, Q = tp.parity_batch()
    total, undecoded = tp.check(I, Q, w, ol, opts, "parity")
    assert total >= 8 and undecoded >= 1          # candidates that never decode are compared too


def test_trace_of_random_scenes_equals_oracle(w):
    """120 random scenes (a longer soak: WSPR_TRACE_SCENES=300 python tests/trace_parity.py scenes)."""
    from test_gpu_parity import random_scenes
    I, Q = random_scenes(120)
    total, undecoded = tp.check(I, Q, w, ol, None, "scenes")
    assert total > 300 and undecoded > 90


def test_trace_of_config3_segments_equals_oracle(w):
    """192 segments of configs[2] (256 until round 5, when the suite gained 15 tests and had to stay near six and a half minutes; soak: WSPR_TRACE_CONFIG3=1024) (ten overlapping signals, -10..-28 dB): ~3 000 candidate visits over two passes, the
    subtractions in between, most of the ladder walks ending in Fano time-outs."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    torch.cuda.set_device(0)
    n = int(os.environ.get("WSPR_TRACE_CONFIG3", "192"))
    I, Q, _ = bench.synth_batch_gpu(n, 4321, torch.device("cuda", 0), 10, -10.0, -28.0, 0.3)
    total, undecoded = tp.check(I.cpu().numpy(), Q.cpu().numpy(), w, ol, None, "config3")
    assert total >= 9 * n and undecoded >= n


def _run_with(env, sets):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "trace_parity.py")] + sets, env=e,
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "TRACE PARITY OK" in r.stdout, (env, r.stdout[-1500:], r.stderr[-3000:])


@pytest.mark.parametrize("env", [{"WSPR_K4_LAG": "tile"}, {"WSPR_K4_DRIFT": "tile"}, {"WSPR_K4_FREQ": "nocentre"},
                                 {"WSPR_FANO_DEVICE": "1"},
                                 {"WSPR_FANO_DEVICE": "0"}, {"WSPR_K3_KERNEL": "lane", "WSPR_K1_FUSED": "1"},
                                 {"WSPR_K3_KERNEL": "waves", "WSPR_K1_FUSED": "0", "WSPR_SLOTS": "1"}],
                         ids=lambda e: ",".join("%s=%s" % kv for kv in e.items()))
def test_trace_under_the_kept_environment_switches(env):
    """Every alternative kernel / placement that stays selectable must give the same per-candidate values."""
    os.environ.setdefault("WSPR_TRACE_SCENES", "20")
    _run_with(env, ["parity", "scenes"])
