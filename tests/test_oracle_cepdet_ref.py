"""Pins the oracle's CepstralDetector restatement (ctuo_cepdet_*) against the reference's own header.

src/vdet/CepstralDet.h with Burg.h / Fft.h / Fft.cc / Complex.h is FFTW-free: oracle/Makefile compiles it where it lies
into oracle/_ref/libref_cepdet.so (git-ignored, travels to the GPU box).  Decisions must agree frame for frame over
whole recordings, across detector restarts and option sets.  Skips when neither the .so nor the reference exist.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle.oracle import CepstralDetector, build
from tests.util import sig

REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_cepdet.so")


def _ref():
    if not os.path.exists(REF_SO) and os.path.exists("/root/reference/src/vdet/CepstralDet.h"):
        build(force=True)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_cepdet.so not built and /root/reference absent")
    L = ctypes.CDLL(REF_SO)
    L.ref_cepdet_new.restype = ctypes.c_void_p
    L.ref_cepdet_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    L.ref_cepdet_run.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    L.ref_cepdet_delete.argtypes = [ctypes.c_void_p]
    return L


def _frames(x, npoints, hop):
    n = (x.size - npoints) // hop + 1
    return np.ascontiguousarray(np.stack([x[t * hop:t * hop + npoints] for t in range(n)]).astype(np.float64))


@pytest.mark.parametrize("npoints,hop,ninit,ncoefs,p,q", [(200, 80, 10, 12, 0.95, 0.99), (400, 160, 10, 12, 0.95, 0.99),
                                                         (200, 80, 40, 10, 0.8, 0.97), (256, 128, 3, 14, 0.9, 0.95)])
def test_detector_decisions_identical_to_reference_header(npoints, hop, ninit, ncoefs, p, q):
    L = _ref()
    total = ones = 0
    for name in ("CS0", "CS3"):
        fr = _frames(sig(name), npoints, hop)
        ref = np.zeros(fr.shape[0], dtype=np.uint8)
        h = L.ref_cepdet_new(npoints, ninit, ncoefs, p, q)
        L.ref_cepdet_run(h, fr.ctypes.data, npoints, fr.shape[0], ref.ctypes.data)
        L.ref_cepdet_delete(h)
        det = CepstralDetector(npoints, ninit, ncoefs, p, q)
        got = np.array([det.process(f) for f in fr], dtype=np.uint8)
        assert np.array_equal(got, ref)
        total += ref.size
        ones += int(ref.sum())
    assert 0 < ones < total  # both classes occur, so the comparison is not vacuous
