"""Pins ctuo_burg_cepstrum() against the reference's own Burg estimator.

src/vdet/Burg.h is the one piece of the reference that compiles in this image (no <fftw3.h>);
oracle/Makefile builds it where it lies into oracle/_ref/libref_burg.so.  The .so travels to the GPU
box (it is git-ignored, not gpurun-ignored); the test skips when neither the .so nor the reference exist.
"""
import ctypes
import os

import numpy as np
import pytest

from oracle.oracle import build, burg_cepstrum
from tests.util import sig

REF_SO = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libref_burg.so")


def _ref():
    if not os.path.exists(REF_SO) and os.path.exists("/root/reference/src/vdet/Burg.h"):
        build(force=True)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_burg.so not built and /root/reference absent")
    L = ctypes.CDLL(REF_SO)
    L.ref_burg_cepstrum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p]
    return L


@pytest.mark.parametrize("npoints,ncoefs", [(200, 14), (400, 14), (200, 10), (64, 5)])
def test_burg_bit_identical_to_reference_header(npoints, ncoefs):
    L = _ref()
    x = sig("CS3").astype(np.float64)
    rng = np.random.default_rng(1)
    for start in rng.integers(0, x.size - npoints, size=8):
        seg = np.ascontiguousarray(x[start:start + npoints] * np.hamming(npoints))
        a_ref, c_ref, al = np.zeros(ncoefs), np.zeros(ncoefs), ctypes.c_double()
        L.ref_burg_cepstrum(seg.ctypes.data, npoints, ncoefs, a_ref.ctypes.data, c_ref.ctypes.data, ctypes.byref(al))
        a, c, alpha = burg_cepstrum(seg, ncoefs)
        assert np.array_equal(a, a_ref) and np.array_equal(c, c_ref) and alpha == al.value
