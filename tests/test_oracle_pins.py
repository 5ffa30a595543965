"""Pins the CPU oracle to outputs of the compiled reference recorded in SURVEY.md 8(c) / Appendix A.

The reference cannot be rebuilt here (FFTW3 absent), so these recorded facts are the only
reference-produced anchors available: frame counts, HTK headers, file sizes, kind codes and the
Burg-VAD decision count (a discontinuous, value-sensitive quantity).
"""
import numpy as np
import pytest

from oracle.oracle import Oracle, OracleError, htk_bytes
from tests.util import C1, C2, C3, C4, C4_NOVAD, C5, sig


def test_c1_frame_counts_header_and_sizes():
    o = Oracle(C1)
    d = o.dims
    assert (d.window, d.wshift, d.wfft, d.K, d.B, d.D) == (400, 160, 512, 257, 30, 13)
    assert (d.period, 4 * d.D, d.htk_kind) == (100000, 52, 8198)  # kind=8198 (020006)
    sizes = {}
    for name, frames in (("CS0", 594), ("CS3", 592)):
        rows = o.process(sig(name))
        assert rows.shape == (frames, 13)
        assert np.isfinite(rows).all()
        img = htk_bytes(rows, d.period, d.htk_kind)
        sizes[name] = len(img)
        hdr = np.frombuffer(img[:12], dtype="<u4")
        assert hdr[0] == frames and hdr[1] == 100000
        assert tuple(np.frombuffer(img[8:12], dtype="<u2")) == (52, 8198)
    assert sizes == {"CS0": 30900, "CS3": 30796}


def test_c2_preset_mfcc_geometry():
    d = Oracle(C2).dims
    assert (d.window, d.wshift, d.wfft, d.K, d.B, d.D, d.htk_kind) == (400, 160, 512, 257, 26, 13, 0o20006)


def test_c3_plp_geometry():
    o = Oracle(C3)
    d = o.dims
    assert d.B == 19 and d.D == 13 and d.htk_kind == 0o20013
    assert Oracle("-fs 8000 -format_in raw -format_out htk -preset plpc".split()).dims.B == 15
    rows = o.process(sig("CS0"))
    assert rows.shape == (594, 13) and np.isfinite(rows).all()


def test_energy_kind_bit():
    d = Oracle(C2 + ["-fea_E", "on"]).dims
    assert d.D == 14 and d.htk_kind == 0o20106


def test_c5_trapdct_geometry():
    o = Oracle(C5)
    d = o.dims
    assert d.B == 23 and d.D == 368 and d.htk_kind == 0o20011  # 592 x 368
    rows = o.process(sig("CS3"))
    assert rows.shape == (592, 368) and np.isfinite(rows).all()


def test_c4_burg_vad_decision_count():
    # VAD file length = frame count (1186); 626 ones; rows unchanged by the VAD.
    o = Oracle(C4)
    assert (o.dims.window, o.dims.wshift, o.dims.wfft, o.dims.K) == (200, 80, 256, 129)
    rows, vad = o.process(sig("CS3"), want_vad=True)
    assert rows.shape == (1186, 13) and vad.size == 1186
    assert set(np.unique(vad)) <= {ord("0"), ord("1")}
    assert int((vad == ord("1")).sum()) == 626
    assert np.array_equal(rows, Oracle(C4_NOVAD).process(sig("CS3")))


def test_vad_cepdist_needs_phase():
    args = [a for a in C4 if a not in ("-vad", "burg")]
    with pytest.raises(OracleError, match="cannot perform iFFT"):
        Oracle(args)


def test_short_signal_and_frame_count_rule():
    o = Oracle(C2)
    assert o.num_frames(239) == -1          # < window - wshift: "IO: Signal shorter than one frame!"
    assert o.num_frames(240) == 0
    assert o.num_frames(399) == 0
    assert o.num_frames(400) == 1
    assert o.num_frames(95382) == 594
    with pytest.raises(OracleError):
        o.process(np.zeros(100, dtype=np.int16))


def test_option_order_and_errors():
    # -preset is an order-dependent macro (src/io/opts.cc:832): later flags override it, earlier ones are lost
    assert Oracle("-fs 16000 -preset mfcc -fb_definition 23filters".split()).dims.B == 23
    assert Oracle("-fs 16000 -fb_definition 23filters -preset mfcc".split()).dims.B == 26
    with pytest.raises(OracleError, match="sampling rate"):
        Oracle(["-preset", "mfcc"])
    with pytest.raises(OracleError, match="Syntax error"):
        Oracle("-fs 16000 -no_such_flag 1".split())
    with pytest.raises(OracleError, match="Preemphasis"):
        Oracle("-fs 16000 -preset mfcc -preem 1.5".split())
    assert abs(Oracle(C2).preem() - 0.9700000286102295) < 1e-16  # float preem, src/io/opts.h:47
