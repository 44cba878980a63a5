"""Deterministic synthetic scenes shared by the golden-vector generator and the tests: a seed fixes the
number of signals (0-6), their messages (types 1-3), SNR, drift, start time and carrier offset."""
import numpy as np

import oracle_lib as ol
import synth

NS = synth.NS
GOLDEN_SEEDS = [11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22]
_T23 = ["PJ4/K1ABC 37", "K1ABC/7 33", "<PJ4/K1ABC> FK52UD 37"]


def _symbols(msg):
    ok, s = ol.channel_symbols(msg)
    assert ok, msg
    return s


def make_scene(seed):
    rng = np.random.default_rng(1000003 * seed + 7)
    sigma = np.sqrt((375.0 / 2500.0) / 2.0)
    I = rng.normal(0, sigma, NS); Q = rng.normal(0, sigma, NS)
    nsig = int(rng.integers(0, 7)) if seed % 6 else 0
    for k in range(nsig):
        msg = _T23[int(rng.integers(0, 3))] if rng.random() < 0.25 else synth.message_for(int(rng.integers(0, 1 << 20)))
        amp = 10.0 ** (rng.uniform(-29, 4) / 20.0)
        si, sq = synth.tone_signal(_symbols(msg), rng.uniform(-120, 120), rng.uniform(0.3, 3.7), amp,
                                   drift=float(rng.integers(-3, 4)))
        I += si; Q += sq
    return synth.normalise(I.astype(np.float32), Q.astype(np.float32))


def spot_record(s):
    """Everything a spot reports; floats as hex so that the comparison is exact."""
    return {"message": s.message.decode(), "call": s.call.decode(), "loc": s.loc.decode(), "pwr": s.pwr.decode(),
            "cycles": int(s.cycles), "jitter": int(s.jitter), "drift": float(s.drift).hex(),
            "sync": float(s.sync).hex(), "dt": float(s.dt).hex(), "freq": float(s.freq).hex(),
            "snr": round(float(s.snr), 3)}
