"""Deterministic synthetic input sets: S-MFCC / S-PLP / S-TRAP (16 kHz) and S-NOISY (8 kHz).

Integer arithmetic only, so this numpy definition and the C one (ctucopy_amd/csrc/synth.cc, include/ctu_synth.h) give
the same int16 samples bit for bit (tests/test_synth.py).  `utterance` is the readable definition; `fill_arena` calls
the C implementation (csrc/synth.cc, built on its own as libctu_synth.so; threads) for the benchmark's 10 000 utterances.
"""
import ctypes

import numpy as np

SET_SPEECH, SET_NOISY = 0, 1
SEED0 = 20260101
_M64 = (1 << 64) - 1
_GAMMA = 0x9E3779B97F4A7C15
_COEF = (0, 32768, 16384, 10923, 8192, 6554)          # 32768 / h
_SNR_MUL = (29569, 26353, 23487, 20933, 18657, 16628, 14819, 13208, 11772, 10491, 9350)  # 5..15 dB, see synth.cc


def _mix(z):
    """splitmix64 finaliser on numpy uint64 arrays (or a Python int)."""
    if isinstance(z, int):
        z &= _M64
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _param(base, k):
    return _mix(base + (k + 1) * _GAMMA)


def _psin(p):
    """parabolic sine of a 32-bit phase (int64 array): +-32768 peak."""
    q = p >> 16
    hi = q >= 32768
    q = np.where(hi, q - 32768, q)
    y = (q * (32768 - q)) >> 13
    return np.where(hi, -y, y)


def fs_of(set_id):
    return 8000 if set_id == SET_NOISY else 16000


def length(set_id, index, mini=False):
    fs = fs_of(set_id)
    lo, hi = (fs * 6 // 10, fs * 2) if mini else (fs * 3, fs * 15)
    return lo + _param(SEED0 + index, 0) % (hi - lo + 1)


def utterance(set_id, index, mini=False):
    """int16 samples of utterance `index` of a set (numpy definition)."""
    fs = fs_of(set_id)
    base = SEED0 + index
    n = length(set_id, index, mini)
    i = np.arange(n, dtype=np.int64)
    H = 3 + _param(base, 1) % 3
    inc_mid, inc_dev = (170 << 32) // fs, (80 << 32) // fs
    ginc = (10 << 32) // (fs * (13 + _param(base, 2) % 21))
    gl = (int(_param(base, 3) & 0xffffffff) + i * ginc) & 0xffffffff
    tri = np.abs(np.where(gl >= (1 << 31), gl - (1 << 32), gl)) - (1 << 30)
    inc = inc_mid + ((inc_dev * tri) >> 30)
    ph = (int(_param(base, 4) & 0xffffffff) + np.cumsum(inc)) & 0xffffffff
    v = np.zeros(n, dtype=np.int64)
    for h in range(1, H + 1):
        php = (h * ph + int(_param(base, 8 + h) & 0xffffffff)) & 0xffffffff
        v += (_psin(php) * _COEF[h]) >> 15
    pam = (int(_param(base, 5) & 0xffffffff) + i * ((4 << 32) // fs)) & 0xffffffff
    am = 19661 + ((13107 * _psin(pam)) >> 15)
    speech = (((v * 5200) >> 15) * am) >> 15
    iu = (i + 1).astype(np.uint64)
    z = _mix(np.uint64(_param(base, 100)) + iu * np.uint64(_GAMMA))
    m16 = np.uint64(0xffff)
    g = ((z & m16) + ((z >> np.uint64(16)) & m16) + ((z >> np.uint64(32)) & m16) + (z >> np.uint64(48))).astype(np.int64) - 131070
    if set_id == SET_SPEECH:
        x = speech + ((g * 130) >> 14)
    else:
        pink = np.zeros(n, dtype=np.int64)
        for r in range(6):
            zr = _mix(np.uint64(_param(base, 101 + r)) + ((i >> r) + 1).astype(np.uint64) * np.uint64(_GAMMA))
            pink += (zr & m16).astype(np.int64) - 32768
        gate = np.zeros(n, dtype=np.int64)
        t, j = fs // 2, 0                                  # first 0.5 s: noise only
        while t < n:
            on = fs * 3 // 10 + _param(base, 200 + 2 * j) % (fs * 12 // 10 + 1)
            gate[t:t + on] = 1
            t += on + fs * 3 // 10 + _param(base, 201 + 2 * j) % (fs * 12 // 10 + 1)
            j += 1
        x = gate * speech + (((g + pink) * _SNR_MUL[_param(base, 6) % 11]) >> 20)
    return np.clip(x, -32768, 32767).astype(np.int16)


_synth = None


def _lib():
    """ctucopy_amd/libctu_synth.so: csrc/synth.cc on its own (the engine library exports the same symbols for C callers,
    but loading it pulls in the HIP runtime - the CPU baseline's workers must not pay for that, nor initialise it)."""
    global _synth
    if _synth is None:
        import os
        from . import build as _build
        if not os.path.exists(_build.SYNTH_LIB):  # built by __graft_entry__.build(); never rebuilt here (many processes load it at once)
            _build.build_synth()
        _synth = ctypes.CDLL(_build.SYNTH_LIB)
    L = _synth
    if not getattr(L, "_synth_ready", False):
        L.ctu_synth_length.restype = ctypes.c_int64
        L.ctu_synth_length.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]
        L.ctu_synth_fill.restype = ctypes.c_int64
        L.ctu_synth_fill.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int64]
        L.ctu_synth_fill_arena.restype = ctypes.c_int64
        L.ctu_synth_fill_arena.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_int32]
        L._synth_ready = True
    return L


def lengths(set_id, indices, mini=False):
    L = _lib()
    return np.array([L.ctu_synth_length(set_id, int(k), int(mini)) for k in indices], dtype=np.int64)


def utterance_c(set_id, index, mini=False):
    L = _lib()
    n = int(L.ctu_synth_length(set_id, index, int(mini)))
    out = np.empty(n, dtype=np.int16)
    L.ctu_synth_fill(set_id, index, int(mini), out.ctypes.data, n)
    return out


def fill_arena(set_id, indices, sample_off, total_samples, mini=False, threads=0, out=None):
    """Packed int16 arena with utterance indices[i] at sample_off[i] (C implementation, host threads)."""
    L = _lib()
    idx = np.ascontiguousarray(indices, dtype=np.int64)
    so = np.ascontiguousarray(sample_off[:idx.size], dtype=np.int64)
    arena = np.zeros(int(total_samples), dtype=np.int16) if out is None else out
    L.ctu_synth_fill_arena(set_id, idx.ctypes.data, int(mini), int(idx.size), so.ctypes.data, arena.ctypes.data, int(threads))
    return arena
