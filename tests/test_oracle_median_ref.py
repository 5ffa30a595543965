"""Pins the oracle's majority filter - decisions, the delay of the feature vectors through its ring and the flush - against the
reference's own medianFilter.

src/vad/vad.h:79-176 is header-only and FFTW-free (it pulls in io/opts.h and base/types.h): oracle/Makefile compiles it where it lies
into oracle/_ref/libref_median.so (git-ignored, travels to the GPU box), driven as VAD::process_frame / BATCH::flush_vad drive it
(src/vad/vad.cc:692-699, 742-745; src/io/batch.cc:230-249).  The raw decisions vad0 do not depend on the filter (the thresholds and the
background update consume vad0, src/vad/vad.cc:696-697): the oracle's run with -vad_filter_order 1 supplies them and the vectors that
enter the filter, and its runs with orders 3, 5, 7 must equal the reference class fed with those.  Skips when neither the .so nor the
reference exist.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle.oracle import Oracle, build
from tests.util import sig, synth_utt

REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_median.so")


def _ref():
    if not os.path.exists(REF_SO) and os.path.exists("/root/reference/src/vad/vad.h"):
        build(force=True)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_median.so not built and /root/reference absent")
    L = ctypes.CDLL(REF_SO)
    L.ref_median_run.restype = ctypes.c_int
    L.ref_median_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    return L


def _reference_filter(L, order, feats, vad0):
    T, nf = feats.shape
    feats = np.ascontiguousarray(feats, dtype=np.float64)
    vad0 = np.ascontiguousarray(vad0, dtype=np.uint8)
    dec = np.zeros(T + order, dtype=np.uint8)
    out = np.zeros((T + order, nf), dtype=np.float64)
    n = L.ref_median_run(order, nf, feats.ctypes.data, vad0.ctypes.data, T, dec.ctypes.data, out.ctypes.data)
    return dec[:n], out[:n]


CRITERIA = ["-vad_cri_mode energy -vad_thr_mode adapt", "-vad_cri_mode energy -vad_thr_mode dyn",
            "-vad burg -vad_cri_mode cepdist -vad_cepdist_mode lpc -vad_thr_mode adapt"]


@pytest.mark.parametrize("cri", CRITERIA)
@pytest.mark.parametrize("order", [3, 5, 7])
def test_majority_filter_and_feature_delay_identical_to_reference_header(cri, order):
    L = _ref()
    base = "-fs 8000 -format_in raw -format_out htk -preset mfcc -vad_out_mode vad".split() + cri.split()
    for u in (sig("CS0")[::2].copy(), synth_utt(31, 8000 * 3 + 5, fs=8000), synth_utt(32, 120 + 80 * 2, fs=8000)):   # the last one: two frames
        rows1, vad1 = Oracle(base + ["-vad_filter_order", "1"]).process(u, want_vad=True)
        rows, vad = Oracle(base + ["-vad_filter_order", str(order)]).process(u, want_vad=True)
        vad0 = (np.asarray(vad1) == ord("1")).astype(np.uint8)
        T = vad0.size
        assert rows1.shape[0] == T
        dec, out = _reference_filter(L, order, rows1.astype(np.float64), vad0)
        # the filter delays by (order-1)/2 frames and the flush makes up for it: frame count unchanged - unless the file has no more
        # frames than the delay: the filter never gets `ready`, BATCH::flush_vad's loop does not start, nothing is written at all
        assert dec.size == (T if T > (order - 1) // 2 else 0)
        assert rows.shape[0] == dec.size and len(vad) == dec.size
        assert np.array_equal(dec, (np.asarray(vad) == ord("1")).astype(np.uint8))
        assert np.array_equal(out.astype(np.float32), rows)      # the vectors the writer sees: delayed by (order-1)/2 frames, the ring's
                                                                 # oldest entries at the flush


def test_order_one_is_the_identity():
    L = _ref()
    rng = np.random.default_rng(1)
    feats, vad0 = rng.standard_normal((50, 4)), rng.integers(0, 2, 50).astype(np.uint8)
    dec, out = _reference_filter(L, 1, feats, vad0)
    assert np.array_equal(dec, vad0) and np.array_equal(out, feats)


def _ref_list(L, order, frames):
    fr = np.array(frames, dtype=np.int32)
    out = np.zeros(int(fr.sum()) + 2 * order * len(frames) + 8)
    n = np.zeros(len(frames), dtype=np.int32)
    L.ref_median_list.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.ref_median_list(order, fr.ctypes.data, len(frames), out.ctypes.data, n.ctypes.data)
    res, w = [], 0
    for k in range(len(frames)):
        res.append(out[w:w + n[k]].astype(int))
        w += n[k]
    return res


@pytest.mark.parametrize("order", [3, 5])
def test_the_filter_ring_runs_on_from_file_to_file(order):
    """One VAD object serves the whole list and VAD::clean() = cleanFilter() resets `start`, the ring and the decisions but neither
    historyIdx nor historySize (src/vad/vad.h:110-121): the next file's vectors land in ring slots out of phase with the slots its outputs
    are read from.  The reference class driven over lists of files (value = frame index + 1, 0 = an untouched slot) against the oracle's
    list mode: which frame's vector each written row carries, all-zero rows and row counts included."""
    L = _ref()
    base = "-fs 8000 -format_in raw -format_out htk -preset mfcc -fea_kind spec -vad_out_mode vad -vad_cri_mode energy -vad_thr_mode adapt".split()
    cfg = base + ["-vad_filter_order", str(order)]
    rng = np.random.default_rng(order)
    for trial in range(6):
        frames = [int(x) for x in rng.integers(0, 12, 7)] + [9]
        utts = [synth_utt(200 + 10 * trial + i, 120 + 80 * T + (0 if T else 40), fs=8000) for i, T in enumerate(frames)]
        want = _ref_list(L, order, frames)
        o = Oracle(cfg)
        got = o.process_list(utts, want_vad=True)
        alone = [Oracle(cfg).process(u) for u in utts]           # every file as the first of its own process: the undisturbed vectors
        for k, (T, (rows, vad), src) in enumerate(zip(frames, got, want)):
            assert rows.shape[0] == src.size == len(vad), (order, frames, k)
            for r, t1 in zip(rows, src):
                if t1 == 0:
                    assert not r.any()                           # a ring slot this file has not written yet: zeros
                else:
                    # the frame's own vector; a file no longer than the delay writes nothing when alone, so take it from a longer run
                    full = alone[k] if alone[k].shape[0] == T else None
                    if full is not None:
                        assert np.array_equal(r, full[t1 - 1]), (order, frames, k, t1)
    # the first file of a process, and process() by default, start in phase
    assert Oracle(cfg).vad_ring() == (0, 0)


def test_reference_list_behaviour_at_the_default_order():
    """What a user of the reference gets at -vad_filter_order 3 (its default): rows of a file shifted by one frame either way, depending on
    the frame counts of the files in front of it."""
    L = _ref()
    assert [r.tolist() for r in _ref_list(L, 3, [6, 6, 6, 7])] == [[1, 2, 3, 4, 5, 6], [0, 1, 2, 3, 4, 5], [2, 3, 4, 5, 6, 4], [1, 2, 3, 4, 5, 6, 7]]
    assert [r.tolist() for r in _ref_list(L, 3, [1, 6])] == [[], [2, 3, 4, 5, 6, 4]]
