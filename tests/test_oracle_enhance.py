"""Row N3 (speech enhancement output): the oracle's restatement of sigOUT (src/io/out.cc:346-451) against an
independent numpy re-computation (np.fft instead of the oracle's own transforms).

PARITY UNPINNED for this row: neither the reference's tests nor SURVEY.md hold a recorded output of the enhancement
mode, so what is pinned here is internal consistency (analysis -> NR off -> synthesis) and the output length law."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from ctucopy_amd import config_dims
from tests.util import sig

X = sig("CS3")[:40000]
BASE = "-fs 16000 -format_in raw -format_out raw -nr_mode none -fea_kind none -fb_definition none".split()


def numpy_sigout(x, window, wshift, wfft, preem, remove_dc):
    x = x.astype(np.float64)
    T = (len(x) - (window - wshift)) // wshift
    j = np.arange(window)
    ham = 0.54 - 0.46 * np.cos(2 * np.pi * j / (window - 1.0))
    corr = max(sum(0.54 - 0.46 * np.cos(2 * np.pi * k / (window - 1.0)) for k in range(i, window, wshift)) for i in range(wshift))
    out = np.zeros(T * wshift + window)
    prev = 0.0
    for t in range(T):
        seg = x[t * wshift:t * wshift + window]
        d = seg - preem * np.concatenate([[prev], seg[:-1]]) if preem > 0 else seg.copy()
        f = d * ham
        if remove_dc:
            f = f - f.mean()
        prev = x[t * wshift + wshift - 1]
        S = np.fft.rfft(f, wfft)
        mag, ph = np.abs(S), np.angle(S)
        if remove_dc:
            mag[0] = np.sqrt(1e-10)
        Y = mag * np.exp(1j * ph)
        Y[0] = mag[0]            # DC and Nyquist go back as positive reals (out.cc:416-419)
        Y[-1] = mag[-1]
        y = np.fft.irfft(Y, wfft)
        out[t * wshift:t * wshift + window] += y[:window]
    n = T * wshift + window - wshift
    v = np.floor(out[:n] / corr)
    return np.where(np.abs(v) > 32767, np.sign(v) * 32767, v).astype(np.int16)


@pytest.mark.parametrize("w,s,extra", [(32, 16, []), (25, 10, ["-preem", "0.97"]), (32, 8, ["-remove_dc", "off"]), (16, 8, [])])
def test_oracle_equals_numpy_resynthesis(w, s, extra):
    cfg = BASE + ["-w", str(w), "-s", str(s)] + extra
    o = Oracle(cfg)
    y = o.enhance(X)
    d = config_dims(cfg)
    assert d.signal_out == 1 and d.row_floats == 0
    T = o.num_frames(len(X))
    assert len(y) == T * d.wshift + d.window - d.wshift
    preem = 0.97 if "-preem" in extra else 0.0
    ref = numpy_sigout(X, d.window, d.wshift, d.wfft, preem, "-remove_dc" not in extra)
    diff = np.abs(y.astype(int) - ref.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3  # floor() of values computed two ways


def test_exten_preset_attenuates_and_keeps_length():
    o = Oracle("-fs 16000 -format_in raw -format_out raw -preset exten".split())
    y = o.enhance(X)
    assert len(y) == o.num_frames(len(X)) * 256 + 256
    assert 0.2 < np.abs(y.astype(float)).mean() / np.abs(X.astype(float)).mean() < 0.9
    assert np.array_equal(y, o.enhance(X))  # NR state is per file
