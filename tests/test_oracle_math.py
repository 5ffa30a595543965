"""Independent numpy cross-checks of the oracle's arithmetic (FFT restatement, filter bank, DCT)."""
import numpy as np

from oracle.oracle import Oracle, burg_cepstrum
from tests.util import C2, C3, sig, synth_utt


def _frame_power(o, x, t):
    d = o.dims
    W = o.hamming()
    p = np.float64(np.float32(0.97))
    s = t * d.wshift
    seg = x[s:s + d.window].astype(np.float64)
    prev = np.concatenate(([x[s - 1] if s > 0 else 0.0], seg[:-1]))
    y = W * (seg - p * prev)
    y -= y.sum() / d.window
    X = np.fft.rfft(y, d.wfft)
    P = X.real ** 2 + X.imag ** 2
    P[0] = 1e-10
    P[-1] = X.real[-1] ** 2
    return P


def test_last_frame_power_matches_numpy_fft():
    o = Oracle(C2)
    x = sig("CS0")
    o.process(x)
    T = o.num_frames(x.size)
    P = _frame_power(o, x, T - 1)
    np.testing.assert_allclose(o.last_power(), P, rtol=1e-9, atol=1e-6)


def test_hamming_and_mel_bank_properties():
    o = Oracle(C2)
    W = o.hamming()
    assert W.shape == (400,) and abs(W[0] - 0.08) < 1e-12 and abs(W[-1] - 0.08) < 1e-12
    mat, first, last = o.fbank()
    assert mat.shape == (26, 257)
    np.testing.assert_allclose(mat.sum(axis=1), 1.0, rtol=1e-12)   # fb_norm: unit area
    assert (np.diff(first) >= 0).all() and (np.diff(last) >= 0).all()
    for b in range(26):
        nz = np.nonzero(mat[b])[0]
        assert first[b] == nz[0] and last[b] == nz[-1]
    # every interior bin is covered by at most two triangles
    assert ((mat > 0).sum(axis=0) <= 2).all()


def test_mfcc_row_from_numpy():
    o = Oracle(C2)
    x = synth_utt(3, 16000)
    rows = o.process(x)
    mat, _, _ = o.fbank()
    t = 17
    Y = np.log(mat @ _frame_power(o, x, t))
    B = 26
    k = np.arange(1, B + 1)
    c = np.array([np.sqrt(2.0 / B) * np.sum(Y * np.cos(np.pi * i * (k - 0.5) / B)) for i in range(13)])
    c[1:] *= 1 + 11 * np.sin(np.pi * np.arange(1, 13) / 22)
    expect = np.concatenate((c[1:], c[:1])).astype(np.float32)
    np.testing.assert_allclose(rows[t], expect, rtol=2e-6, atol=2e-6)


def test_plp_levinson_against_numpy_solve():
    o = Oracle(C3 + ["-fea_lifter", "0"])
    x = synth_utt(5, 8000)
    rows = o.process(x)
    Y = o.last_fbank()          # (mat.P)^0.33 of the last frame
    B = Y.size
    N = 2 * (B - 1)
    full = np.concatenate((Y, Y[-2:0:-1]))
    R = np.fft.ifft(full).real[:13]
    assert N == full.size
    a = np.linalg.solve(np.array([[R[abs(i - j)] for j in range(12)] for i in range(12)]), -R[1:13])
    err = R[0] + np.dot(a, R[1:13])
    c = np.zeros(13)
    c[0] = np.log(err)
    for n in range(1, 13):
        c[n] = -a[n - 1] - sum((n - k) * c[n - k] * a[k - 1] for k in range(1, n)) / n
    expect = np.concatenate((c[1:], c[:1]))
    np.testing.assert_allclose(rows[-1], expect, rtol=1e-4, atol=1e-5)


def test_burg_on_ar_process():
    rng = np.random.default_rng(0)
    e = rng.normal(size=4000)
    x = np.zeros_like(e)
    for i in range(2, x.size):
        x[i] = 1.3 * x[i - 1] - 0.6 * x[i - 2] + e[i]
    a, c, alpha = burg_cepstrum(x, 3)
    assert abs(a[1] + 1.3) < 0.05 and abs(a[2] - 0.6) < 0.05 and abs(alpha - 1.0) < 0.1
    assert abs(c[1] + a[1]) < 1e-12
