"""Two identities the device relies on, checked in numpy:
 * the N-point spectrum of a frame that is zero beyond its window is every (256/N)-th bin of its 256-point spectrum
   (FFT sizes 32..128 ride on the 256-point mode, ctu_engine_create);
 * a float split into two fp16 terms (h = fp16(x), l = fp16(x - h)) reproduces x to max(2^-21 |x|, 3e-8) (the second
   bound is half the spacing of fp16 subnormals, where the residual of small values lands), so the three products hh, hl,
   lh of two such operands carry a sum of products to ~2^-20 of its largest term (trapdct_split16_kernel<true>)."""
import numpy as np
import pytest


@pytest.mark.parametrize("n", [32, 64, 128])
def test_small_spectrum_is_a_subset_of_the_256_point_spectrum(n):
    rng = np.random.default_rng(n)
    w = int(rng.integers(n // 2 + 1, n + 1))  # any window the reference maps to this size
    x = np.zeros(256)
    x[:w] = rng.standard_normal(w)
    small = np.fft.rfft(x[:n])        # n/2 + 1 bins
    big = np.fft.rfft(x)              # 129 bins
    assert np.allclose(small, big[:: 256 // n], rtol=0, atol=1e-12)


def test_two_term_fp16_split():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(100000) * 8).astype(np.float32)           # centred log-mel values: a few units
    g = (rng.uniform(-1, 1, 100000) * rng.uniform(1e-3, 1, 100000)).astype(np.float32)  # table entries, |G| <= 1
    def split(v):
        h = v.astype(np.float16)
        l = (v - h.astype(np.float32)).astype(np.float16)
        return h.astype(np.float64), l.astype(np.float64)
    xh, xl = split(x)
    gh, gl = split(g)
    assert np.all(np.abs((xh + xl) - x) <= np.maximum(2.0 ** -21 * np.abs(x), 3.0e-8))
    assert np.all(np.abs((gh + gl) - g) <= np.maximum(2.0 ** -21 * np.abs(g), 3.0e-8))
    prod = gh * xh + gh * xl + gl * xh     # what the three MFMAs accumulate (exactly, in fp32 the products are exact)
    ref = g.astype(np.float64) * x.astype(np.float64)
    assert np.max(np.abs(prod - ref)) <= 2.0 ** -20 * np.max(np.abs(ref))
