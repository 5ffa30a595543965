"""Row N2 (CMS part): cms_POST (src/fea/post_impl.cc:159-240) as restated in the oracle, against closed forms.
The reference keeps the running mean and the block ring in `float`, so the closed forms (in double) agree to float
rounding of values up to ~60 (c0), not better."""
import numpy as np
import pytest

from oracle.oracle import Oracle, OracleError
from ctucopy_amd import config_dims
from tests.util import C2, C3, sig

PCM = sig("CS0")


def block_cms(x, L):
    out = x.copy()
    for t in range(L - 1, x.shape[0]):
        out[t] = x[t] - x[t - L + 1:t + 1].mean(0)
    return out


def exp_cms(x, z):
    m = np.zeros(x.shape[1])
    out = np.zeros_like(x)
    for t in range(x.shape[0]):
        m = np.float32(m * z + x[t] * (1 - z)).astype(np.float64)
        out[t] = x[t] - m
    return out


def test_exponential_mean():
    base = Oracle(C2).process(PCM).astype(np.float64)
    for Z in (2000.0, 300.0):
        z = float(np.float32(np.float32(1) - (2 * 10.0) / np.float32(Z)))  # src/io/opts.cc:273-274
        r = Oracle(C2 + ["-fea_Z_exp", str(Z)]).process(PCM).astype(np.float64)
        assert np.abs(r - exp_cms(base, z)).max() < 1e-4


@pytest.mark.parametrize("Zb", [100.0, 500.0, 2000.0])
def test_block_mean(Zb):
    L = int(np.floor((Zb - 25.0) / 10.0)) + 1  # src/io/opts.cc:270-271
    base = Oracle(C2).process(PCM).astype(np.float64)
    r = Oracle(C2 + ["-fea_Z_block", str(Zb)]).process(PCM).astype(np.float64)
    assert np.array_equal(r[:L - 1], base[:L - 1])  # nothing is subtracted until the ring is full
    assert np.abs(r - block_cms(base, L)).max() < 1e-4


def test_cms_touches_block0_only_and_runs_after_the_delta_chain():
    cfg = C2 + ["-fea_delta", "d_a", "-fea_E", "on"]
    plain = Oracle(cfg).process(PCM).astype(np.float64)
    r = Oracle(cfg + ["-fea_Z_block", "500"]).process(PCM).astype(np.float64)
    assert np.array_equal(r[:, 13:], plain[:, 13:])  # deltas and E come from the un-normalised cepstra
    assert np.abs(r[:, :13] - block_cms(plain[:, :13], 48)).max() < 1e-4


def test_state_is_per_utterance_and_c0_off_rows_skip_c0():
    o = Oracle(C2 + ["-fea_Z_exp", "500"])
    a = o.process(PCM)
    b = o.process(PCM)
    assert np.array_equal(a, b)
    full = Oracle(C2 + ["-fea_Z_exp", "500"]).process(PCM)
    noc0 = Oracle(C2 + ["-fea_c0", "off", "-fea_Z_exp", "500"]).process(PCM)
    assert noc0.shape[1] == 12 and np.array_equal(noc0, full[:, :12])


def test_plp_cepstra_and_both_flags():
    r = Oracle(C3 + ["-fea_Z_block", "300"]).process(PCM)
    assert r.shape[1] == Oracle(C3).dims.D and np.isfinite(r).all()
    # both given: block wins (post_impl.cc:163-167)
    a = Oracle(C2 + ["-fea_Z_exp", "500", "-fea_Z_block", "300"]).process(PCM)
    b = Oracle(C2 + ["-fea_Z_block", "300"]).process(PCM)
    assert np.array_equal(a, b)


def test_refusals():
    with pytest.raises(OracleError, match="non-cepstral"):
        Oracle(C2 + ["-fea_kind", "logspec", "-fea_Z_exp", "500"])
    with pytest.raises(OracleError, match="stacked"):
        Oracle(C2 + ["-fea_trap", "5", "-fea_Z_exp", "500"])
    with pytest.raises(OracleError, match="CMVN"):
        Oracle(C2 + ["-stat_cmvn", "x.txt"])
    with pytest.raises(OracleError, match="shorter than one frame"):
        Oracle(C2 + ["-fea_Z_block", "10"])
    assert config_dims(C2 + ["-fea_Z_exp", "500"]).row_floats == 13
