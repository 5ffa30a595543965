"""Synthetic input sets (include/ctu_synth.h): the C generator in libctu_engine.so and the numpy
definition give the same samples; the sets have the properties the survey asks for."""
import numpy as np
import pytest

from ctucopy_amd import build as cbuild
from ctucopy_amd import synth


@pytest.fixture(scope="module", autouse=True)
def _built():
    cbuild.build_engine()


@pytest.mark.parametrize("set_id", [synth.SET_SPEECH, synth.SET_NOISY])
def test_c_and_numpy_generators_agree_bit_for_bit(set_id):
    for idx, mini in ((0, True), (3, True), (15, True), (98765, True), (2, False)):
        a, c = synth.utterance(set_id, idx, mini), synth.utterance_c(set_id, idx, mini)
        assert a.dtype == np.int16 and np.array_equal(a, c)


def test_set_properties():
    fs = 16000
    L = synth.lengths(synth.SET_SPEECH, range(200))
    assert L.min() >= 3 * fs and L.max() <= 15 * fs and L.std() > fs
    x = synth.utterance_c(synth.SET_SPEECH, 11)
    assert 9000 < np.abs(x).max() < 16000
    frames = x[:x.size // 400 * 400].reshape(-1, 400).astype(np.float64)
    assert frames.std(axis=1).min() > 100          # never digitally silent
    y = synth.utterance_c(synth.SET_NOISY, 11)
    assert y.size >= 3 * 8000
    head, rest = y[:4000].astype(np.float64), y[4000:].astype(np.float64)
    assert 200 < head.std() < 2500 and rest.std() > head.std()   # first 0.5 s is noise only


def test_arena_fill_matches_single_utterances():
    idx = np.array([5, 0, 9, 2])
    L = synth.lengths(synth.SET_NOISY, idx, mini=True)
    off = np.concatenate([[8], 8 + np.cumsum((L + 7) // 8 * 8)])
    arena = synth.fill_arena(synth.SET_NOISY, idx, off[:-1], off[-1] + 64, mini=True, threads=3)
    for k, i in enumerate(idx):
        assert np.array_equal(arena[off[k]:off[k] + L[k]], synth.utterance(synth.SET_NOISY, int(i), True))
