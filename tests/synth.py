"""Synthetic WSPR segments ("wsprsim frames") for tests and bench -- numpy version.

Signal model = the reference self-test generator (rtlsdr_wsprd.c:743-760: continuous-
phase 4-FSK, double-precision phase) placed in complex AWGN whose power is 1 in a
2500 Hz bandwidth (sigma^2 per rail = (375/2500)/2), amplitude 10^(SNR/20), followed by
the receiver's max-abs normalisation (rtlsdr_wsprd.c:290-305).  SURVEY §8(d) configs 2/3.
"""
import numpy as np

CALLS = ["K1JT", "W1AW", "VA2GKA", "G4ABC", "JA1XYZ", "K9AN", "DL0ABC", "VK2AB", "ZS6BKW", "F5XYZ",
         "OH2ABC", "PY2AB", "LU1ABC", "EA4XY", "I2ABC", "SM5XYZ", "UA3ABC", "BV2AB", "HL1ABC", "ZL2XY"]
GRIDS = ["FN20", "FN31", "FN35", "IO91", "PM95", "EN50", "JO62", "QF56", "KG33", "JN18",
         "KP20", "GG66", "GF05", "IN80", "JN45", "JO89", "KO85", "PL05", "PM37", "RE78"]
POWERS = [0, 3, 7, 10, 13, 17, 20, 23, 27, 30, 33, 37, 40, 43, 47, 50, 53, 57, 60]
NS = 45000
# compound calls (type 2) and the 6-character locators their stations send in the hashed form (type 3), for -H traffic
STATIONS = [("PJ4/K1ABC", "FK52UD", 37), ("K1ABC/7", "DN40AB", 30), ("VP9/W1AW", "FM72PH", 23), ("G4ABC/P", "IO91WM", 27),
            ("F/DL0ABC", "JN18DU", 33), ("ZS6BKW/5", "KG33XX", 20), ("EA8/OH2AB", "IL18QI", 40), ("JA1XYZ/1", "PM95RR", 10)]


def station_message(st, seg):
    """Station st's transmission in segment seg: stations alternate between their type-2 and type-3 messages from slot to
    slot, as real ones do."""
    call, grid6, pwr = STATIONS[st % len(STATIONS)]
    return ("%s %d" % (call, pwr)) if (seg + st) % 2 == 0 else ("<%s> %s %d" % (call, grid6, pwr))


def message_for(idx):
    return "%s %s %d" % (CALLS[idx % len(CALLS)], GRIDS[(idx // 3) % len(GRIDS)],
                         POWERS[(idx // 7) % len(POWERS)])


_L = "ABCDEFGHIJKLMNOPQRSTUVWXYZ"


def message_wide(idx):
    """A type-1 message drawn from (nearly) the whole space a receiver can hear: 26^5 x 10 six-character calls with the
    digit at index 2 and 26^4 x 10 + 26^3 x 10 shorter ones with the digit at index 1 (the two shapes pack_call takes
    without overflow), 18 x 18 x 100 locators, 19 powers -- every signal of a benchmark batch a different
    message, so that nothing host-side can be answered from a cache of earlier decodes.  idx: any non-negative integer."""
    r = int(idx)
    shape, r = r % 4, r // 4
    if shape:                                           # "KA1ABC"
        c = [_L[r % 26]]; r //= 26
        c.append(_L[r % 26]); r //= 26
        c.append(str(r % 10)); r //= 10
        for _ in range(3):
            c.append(_L[r % 26]); r //= 26
    else:                                               # "K1AB", "K1ABC" (a type-1 call has four characters or more,
        c = [_L[r % 26]]; r //= 26                      # wsprsim_utils.c:201)
        c.append(str(r % 10)); r //= 10
        n = 2 + r % 2; r //= 2
        for _ in range(n):
            c.append(_L[r % 26]); r //= 26
    g = _L[r % 18]; r //= 18
    g += _L[r % 18]; r //= 18
    g += "%02d" % (r % 100); r //= 100
    return "%s %s %d" % ("".join(c), g, POWERS[r % len(POWERS)])


def expected_text(msg):
    """Decoder prints a type-1 power with two digits (wsprd_utils.c:259); type-2 / type-3 texts come back as sent (the
    hashed call resolved)."""
    if "/" in msg or msg.startswith("<"):
        return msg
    c, g, p = msg.split()
    return "%s %s %02d" % (c, g, int(p))


def tone_signal(symbols, f0, t0, amp, drift=0.0):
    """Complex baseband CP-4FSK, 162 symbols x 256 samples starting at t0 seconds."""
    out_i = np.zeros(NS, np.float64)
    out_q = np.zeros(NS, np.float64)
    df, dt = 375.0 / 256.0, 1.0 / 375.0
    k = np.arange(162 * 256)
    sym = np.repeat(np.asarray(symbols, np.float64), 256)
    fdrift = (drift / 2.0) * (np.repeat(np.arange(162), 256) - 81.0) / 81.0
    dphi = 2.0 * np.pi * dt * (f0 + fdrift + (sym - 1.5) * df)
    phi = np.concatenate(([0.0], np.cumsum(dphi)[:-1]))
    start = int(round(t0 / dt))
    idx = start + k
    ok = (idx >= 0) & (idx < NS)
    out_i[idx[ok]] = amp * np.cos(phi[ok])
    out_q[idx[ok]] = amp * np.sin(phi[ok])
    return out_i, out_q


def normalise(I, Q):
    peak = max(np.float32(1e-24), np.abs(I).max(), np.abs(Q).max())
    scale = np.float32(0.5 / float(peak))
    return (I * scale).astype(np.float32), (Q * scale).astype(np.float32)


def make_segment(seed, symbols_of, n_signals=1, snr_db=-20.0, snr_span=0.0, f_span=100.0,
                 t_jitter=1.0, drift=0.0):
    """Returns (I, Q, [(message, f0, t0, snr)]).  symbols_of(message) -> 162 channel symbols."""
    rng = np.random.default_rng(seed)
    sigma = np.sqrt((375.0 / 2500.0) / 2.0)
    I = rng.normal(0.0, sigma, NS)
    Q = rng.normal(0.0, sigma, NS)
    truth = []
    if n_signals == 1:
        f0s = [rng.uniform(-f_span, f_span)]
    else:
        slots = np.linspace(-f_span, f_span, n_signals)
        f0s = list(slots + rng.uniform(-2.0, 2.0, n_signals))
    for s in range(n_signals):
        msg = message_for(int(rng.integers(0, 1 << 20)))
        snr = snr_db - (snr_span * s / max(1, n_signals - 1))
        t0 = 2.0 + rng.uniform(-t_jitter, t_jitter)
        si, sq = tone_signal(symbols_of(msg), f0s[s], t0, 10.0 ** (snr / 20.0), drift)
        I += si
        Q += sq
        truth.append((msg, f0s[s], t0, snr))
    I32, Q32 = normalise(I.astype(np.float32), Q.astype(np.float32))
    return I32, Q32, truth
