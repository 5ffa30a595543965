"""Committed 16-utterance miniatures of the synthetic sets (tests/golden/*_mini_pcm.npz) and the oracle's outputs on
them for BASELINE.json's configurations C2..C5 (tests/golden/mini_expected.npz; made by tests/golden/make_fixtures.py).

CPU: the generator reproduces the committed PCM and the oracle reproduces the committed outputs, bit for bit.
GPU: the HIP path reproduces the committed outputs within 1e-4 (norm-wise) with identical VAD strings.
"""
import os

import numpy as np
import pytest

from ctucopy_amd import synth
from oracle.oracle import Oracle
from tests.util import C2, C3, C4, C5, GOLDEN

TOL = 1e-4
CASES = (("c2", C2, "smfcc", 16), ("c3", C3, "smfcc", 16), ("c4", C4, "snoisy", 16), ("c5", C5, "smfcc", 2))


def _pcm(name):
    z = np.load(os.path.join(GOLDEN, f"{name}_mini_pcm.npz"))
    return [z[f"u{i:02d}"] for i in range(16)]


def _expected():
    return np.load(os.path.join(GOLDEN, "mini_expected.npz"))


@pytest.mark.parametrize("name,set_id", [("smfcc", synth.SET_SPEECH), ("snoisy", synth.SET_NOISY)])
def test_generator_reproduces_the_committed_pcm(name, set_id):
    from ctucopy_amd import build as cbuild
    cbuild.build_engine()
    for i, u in enumerate(_pcm(name)):
        assert np.array_equal(u, synth.utterance_c(set_id, i, mini=True))
    assert np.array_equal(_pcm(name)[3], synth.utterance(set_id, 3, mini=True))  # and the numpy definition


@pytest.mark.parametrize("tag,cfg,src,n", CASES)
def test_oracle_reproduces_the_committed_outputs(tag, cfg, src, n):
    exp, pcm, orc = _expected(), _pcm(src), Oracle(cfg)
    for i in range(n):
        if orc.dims.do_vad:
            rows, vad = orc.process(pcm[i], want_vad=True)
            assert np.array_equal(vad, exp[f"{tag}_vad_{i:02d}"])
        else:
            rows = orc.process(pcm[i])
        assert rows.dtype == np.float32 and np.array_equal(rows, exp[f"{tag}_rows_{i:02d}"])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,cfg,src,n", CASES)
def test_gpu_matches_the_committed_outputs(tag, cfg, src, n):
    import torch
    assert torch.cuda.is_available()
    from ctucopy_amd import Engine
    exp, pcm = _expected(), _pcm(src)
    eng = Engine(cfg)
    got = eng.extract(pcm[:n], want_vad=True)
    ones = total = 0
    for i in range(n):
        ref = exp[f"{tag}_rows_{i:02d}"]
        assert got[0][i].shape == ref.shape
        err = np.abs(got[0][i] - ref) / np.maximum(np.abs(ref), 1.0)
        assert err.max() <= TOL, (tag, i, float(err.max()))
        if eng.dims.has_vad:  # decisions are discontinuous: the strings must be identical, not close
            assert np.array_equal(got[1][i], exp[f"{tag}_vad_{i:02d}"]), (tag, i)
            ones += int((got[1][i] == ord("1")).sum())
            total += got[1][i].size
    if eng.dims.has_vad:
        assert 0.1 < ones / total < 0.9  # both classes occur on the gated noisy set
