"""The oracle's restatement of the VAD behind a delta / stacking chain and behind CMS (src/io/batch.cc:172-241), pinned by
relations that do not depend on how it is written: with a memoryless threshold and no median filter the detector's byte for
call m is the plain detector's byte for input frame min(m + delay, T-1); CMS changes rows, not decisions."""
import numpy as np
import pytest

from oracle.oracle import Oracle
from tests.util import C2, synth_utt

MEMORYLESS = "-vad_out_mode vad -vad_cri_mode energy -vad_thr_mode absolute -vad_filter_order 1".split()


@pytest.mark.parametrize("chain,delay", [(["-fea_delta", "d"], 2), (["-fea_delta", "d_a"], 4), (["-fea_delta", "d_a_t"], 6),
                                         (["-fea_delta", "d", "-d_win", "5"], 5), (["-fea_trap", "9"], 4),
                                         # trapdct: process_frame() is false for the first half context and flush_frame() delivers the last one
                                         # (src/fea/fea_trap.cc:53-127): the same delayed writer, half = (traplen - 1) / 2 frames
                                         (["-fb_definition", "23filters", "-fea_kind", "trapdct,101,16"], 50),
                                         (["-fb_definition", "23filters", "-fea_kind", "trapdct,31,8"], 15)])
def test_detector_behind_a_chain_sees_the_newest_input_frame(chain, delay):
    u = synth_utt(77, 30000)
    # pick a threshold that splits the frames: scan a few levels until both decisions occur in the plain run
    for level in (150.0, 160.0, 170.0, 175.0, 180.0, 185.0, 190.0, 195.0):
        cfg = C2 + MEMORYLESS + ["-vad_absolute_thr", str(level)]
        _, plain = Oracle(cfg).process(u, want_vad=True)
        if 0.2 < (plain == ord("1")).mean() < 0.8:
            break
    else:
        pytest.skip("no splitting threshold found")
    rows, chained = Oracle(cfg + chain).process(u, want_vad=True)
    T = plain.size
    assert chained.size == T and rows.shape[0] == T
    want = plain[np.minimum(np.arange(T) + delay, T - 1)]
    assert np.array_equal(chained, want)


def test_cms_leaves_the_spectral_criteria_alone():
    u = synth_utt(78, 30000)
    base = C2 + "-vad_out_mode vad -vad_cri_mode energy".split()
    r0, v0 = Oracle(base).process(u, want_vad=True)
    r1, v1 = Oracle(base + ["-fea_Z_exp", "1500"]).process(u, want_vad=True)
    assert np.array_equal(v0, v1) and r0.shape == r1.shape and not np.allclose(r0[:, :12], r1[:, :12])
    # and dropping rows happens after the subtraction: the kept rows are the CMS rows of the kept frames
    r2, v2 = Oracle(base + ["-fea_Z_exp", "1500", "-vad_apply_mode", "drop"]).process(u, want_vad=True)
    assert np.array_equal(v2, v1)
    keep = v1 == ord("1")
    assert r2.shape[0] == int(keep.sum())


def test_apply_mode_silence_changes_nothing_on_the_feature_path():
    # VAD::silence_frame (src/vad/vad.cc:727-736) zeroes in->_Xsabs behind a non-speech frame - after the frame's features left and
    # before get_frame() rewrites the whole vector: rows and decisions are those of `none`; with the *ss modes (which seed the next
    # file from that vector, src/nr/nr.cc:212-221) the oracle refuses; with signal output the reference has no VAD object at all
    from oracle.oracle import OracleError
    from tests.util import C4
    u = synth_utt(78, 16000, fs=8000)
    rows_n, vad_n = Oracle(C4 + ["-vad_apply_mode", "none"]).process(u, want_vad=True)
    rows_s, vad_s = Oracle(C4 + ["-vad_apply_mode", "silence"]).process(u, want_vad=True)
    assert np.array_equal(rows_n, rows_s) and np.array_equal(vad_n, vad_s) and 0 < (vad_s == ord("1")).sum() < vad_s.size
    with pytest.raises(OracleError, match="silence"):
        Oracle(C2 + "-nr_mode fwss -vad burg -vad_out_mode vad -vad_apply_mode silence".split())
    with pytest.raises(OracleError, match="crashes in the reference"):
        Oracle("-fs 16000 -format_in raw -format_out raw -preset exten -vad_out_mode vad".split())
