"""ctypes wrapper around oracle/_build/libctu_oracle.so.

TEST INFRASTRUCTURE ONLY -- see oracle/ctu_oracle.h.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libctu_oracle.so")
_lib = None


class Dims(ctypes.Structure):
    _fields_ = [
        ("fs", ctypes.c_int), ("window", ctypes.c_int), ("wshift", ctypes.c_int), ("wfft", ctypes.c_int),
        ("K", ctypes.c_int), ("B", ctypes.c_int), ("nfea", ctypes.c_int), ("D", ctypes.c_int),
        ("htk_kind", ctypes.c_int), ("period", ctypes.c_uint), ("do_vad", ctypes.c_int),
        ("phase_needed", ctypes.c_int), ("fb_power", ctypes.c_int), ("swap_out", ctypes.c_int),
    ]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "ctu_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        L.ctuo_create.restype = ctypes.c_void_p
        L.ctuo_create.argtypes = [ctypes.c_int, ctypes.POINTER(ctypes.c_char_p), ctypes.c_char_p, ctypes.c_int]
        L.ctuo_destroy.argtypes = [ctypes.c_void_p]
        L.ctuo_get_dims.argtypes = [ctypes.c_void_p, ctypes.POINTER(Dims)]
        L.ctuo_num_frames.restype = ctypes.c_long
        L.ctuo_num_frames.argtypes = [ctypes.c_void_p, ctypes.c_long]
        L.ctuo_out_samples.restype = ctypes.c_long
        L.ctuo_out_samples.argtypes = [ctypes.c_void_p, ctypes.c_long]
        L.ctuo_enhance.restype = ctypes.c_long
        L.ctuo_enhance.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
        L.ctuo_set_vad_ring.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.ctuo_get_vad_ring.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.ctuo_process.restype = ctypes.c_long
        L.ctuo_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
        L.ctuo_error.restype = ctypes.c_char_p
        L.ctuo_error.argtypes = [ctypes.c_void_p]
        L.ctuo_hamming.restype = ctypes.POINTER(ctypes.c_double)
        L.ctuo_hamming.argtypes = [ctypes.c_void_p]
        L.ctuo_preem.restype = ctypes.c_float
        L.ctuo_preem.argtypes = [ctypes.c_void_p]
        L.ctuo_fb_row.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.POINTER(ctypes.c_double)),
                                  ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.ctuo_last_power.restype = ctypes.POINTER(ctypes.c_double)
        L.ctuo_last_power.argtypes = [ctypes.c_void_p]
        L.ctuo_last_fbank.restype = ctypes.POINTER(ctypes.c_double)
        L.ctuo_last_fbank.argtypes = [ctypes.c_void_p]
        L.ctuo_burg_cepstrum.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_void_p, ctypes.c_void_p]
        L.ctuo_cepdet_new.restype = ctypes.c_void_p
        L.ctuo_cepdet_new.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]
        L.ctuo_cepdet_process.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.ctuo_cepdet_last_distance.restype = ctypes.c_double
        L.ctuo_cepdet_last_distance.argtypes = [ctypes.c_void_p]
        L.ctuo_cepdet_free.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


class Oracle:
    """One configured reference chain; `args` is the ctucopy command line (list of str, no argv[0])."""

    def __init__(self, args):
        L = lib()
        self.args = [str(a) for a in args]
        arr = (ctypes.c_char_p * len(self.args))(*[a.encode() for a in self.args])
        err = ctypes.create_string_buffer(512)
        self._h = L.ctuo_create(len(self.args), arr, err, 512)
        if not self._h:
            raise OracleError(err.value.decode())
        d = Dims()
        L.ctuo_get_dims(self._h, ctypes.byref(d))
        self.dims = d

    def close(self):
        if getattr(self, "_h", None):
            lib().ctuo_destroy(self._h)
            self._h = None

    __del__ = close

    def num_frames(self, nsamples):
        return int(lib().ctuo_num_frames(self._h, int(nsamples)))

    def process(self, pcm, want_vad=False, first_in_process=True):
        """pcm: int16 array of one utterance -> float32 [rows, D] (and the VAD '0'/'1' bytes).

        first_in_process: the VAD's majority filter starts as in a fresh process (historyIdx = historySize = 0); False keeps what the
        previous call left, as the reference does from file to file (src/vad/vad.h:110-121; process_list)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        if first_in_process:
            lib().ctuo_set_vad_ring(self._h, 0, 0)
        T = self.num_frames(pcm.size)
        if T < 0:
            raise OracleError("IO: Signal shorter than one frame!")
        # a file can write more rows than it has frames: a majority filter left half drained by the file before it (list mode, orders >= 5)
        rows = np.empty((T + 64, self.dims.D), dtype=np.float32)
        vad = np.zeros(T + 64, dtype=np.uint8)
        n = lib().ctuo_process(self._h, pcm.ctypes.data, pcm.size, rows.ctypes.data, vad.ctypes.data)
        if n < 0:
            raise OracleError(lib().ctuo_error(self._h).decode())
        out = rows[:n].copy()
        if want_vad:
            return out, vad[vad != 0].copy()  # '0' / '1' per decision written; none for a file the majority filter never got ready on
        return out

    def process_list(self, utterances, want_vad=False):
        """The utterances as the files of ONE list of one process: the majority filter's ring index runs on from file to file."""
        return [self.process(u, want_vad=want_vad, first_in_process=(i == 0)) for i, u in enumerate(utterances)]

    def set_vad_ring(self, hidx, hsize=0):
        """Start the next process(..., first_in_process=False) where a list's earlier files left the majority filter."""
        lib().ctuo_set_vad_ring(self._h, int(hidx), int(hsize))

    def vad_ring(self):
        """(historyIdx, historySize) the last processed file left behind."""
        a, b = ctypes.c_int(0), ctypes.c_int(0)
        lib().ctuo_get_vad_ring(self._h, ctypes.byref(a), ctypes.byref(b))
        return a.value, b.value

    def enhance(self, pcm):
        """-format_out raw|wave: int16 samples of the enhanced utterance (what rawOUT/waveOUT write, host byte order)."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        L = lib()
        n = L.ctuo_out_samples(self._h, pcm.size)
        if n < 0:
            raise OracleError("IO: Signal shorter than one frame!")
        out = np.zeros(max(n, 1), dtype=np.int16)
        got = L.ctuo_enhance(self._h, pcm.ctypes.data, pcm.size, out.ctypes.data)
        if got < 0:
            raise OracleError(L.ctuo_error(self._h).decode())
        return out[:got].copy()

    def hamming(self):
        return np.ctypeslib.as_array(lib().ctuo_hamming(self._h), shape=(self.dims.window,)).copy()

    def preem(self):
        return float(lib().ctuo_preem(self._h))

    def fbank(self):
        """Dense [B, K] float64 weights plus first/last non-zero bin per band."""
        B, K = self.dims.B, self.dims.K
        mat = np.zeros((B, K))
        first = np.zeros(B, dtype=np.int32)
        last = np.zeros(B, dtype=np.int32)
        for b in range(B):
            p = ctypes.POINTER(ctypes.c_double)()
            f = ctypes.c_int()
            l = ctypes.c_int()
            lib().ctuo_fb_row(self._h, b, ctypes.byref(p), ctypes.byref(f), ctypes.byref(l))
            mat[b] = np.ctypeslib.as_array(p, shape=(K + 2,))[:K]
            first[b], last[b] = f.value, l.value
        return mat, first, last

    def last_power(self):
        return np.ctypeslib.as_array(lib().ctuo_last_power(self._h), shape=(self.dims.K,)).copy()

    def last_fbank(self):
        return np.ctypeslib.as_array(lib().ctuo_last_fbank(self._h), shape=(self.dims.B,)).copy()


def burg_cepstrum(x, ncoefs):
    x = np.ascontiguousarray(x, dtype=np.float64)
    a = np.zeros(ncoefs)
    c = np.zeros(ncoefs)
    alpha = ctypes.c_double()
    lib().ctuo_burg_cepstrum(x.ctypes.data, x.size, ncoefs, a.ctypes.data, c.ctypes.data, ctypes.byref(alpha))
    return a, c, alpha.value


class CepstralDetector:
    """CepstralDetector<BurgCepstrumEstimator> of src/vdet/CepstralDet.h:92-217 (one instance per file in hwss / fwss)."""

    def __init__(self, npoints, ninit=40, ncoefs=10, p=0.8, q=0.97):
        self._h = lib().ctuo_cepdet_new(npoints, ninit, ncoefs, p, q)
        self.npoints = npoints

    def process(self, frame):
        f = np.ascontiguousarray(frame, dtype=np.float64)
        assert f.size == self.npoints
        return int(lib().ctuo_cepdet_process(self._h, f.ctypes.data))

    def last_distance(self):
        return float(lib().ctuo_cepdet_last_distance(self._h))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ctuo_cepdet_free(self._h)
            self._h = None


def htk_bytes(rows, period, kind, big_endian=False):
    """HTK file image as htkOUT writes it (src/io/out.cc:115-213): 12-byte header + float32 rows."""
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    n, d = rows.shape if rows.ndim == 2 else (0, 0)
    e = ">" if big_endian else "<"
    hdr = np.array([n], dtype=e + "u4").tobytes() + np.array([period], dtype=e + "u4").tobytes() + \
        np.array([4 * d], dtype=e + "u2").tobytes() + np.array([kind], dtype=e + "u2").tobytes()
    return hdr + rows.astype(e + "f4").tobytes()


# ---------------------------------------------------------------------------------------------------------------
# Per-speaker CMVN (row N2): numpy restatement of cmvn_POST (src/fea/post_impl.cc:51-142) and of the three passes
# BATCH::process makes for it (src/io/batch.cc:331-419).  Works on the rows the oracle emits for the same options
# without the -stat_cmvn / -apply_cmvn flags (float32 casts of the doubles the reference accumulates).
def cmvn_slot_columns(ncep, n_blocks):
    """Row column of statistic slot k.  Slot k holds internal vector entry k+1, the last slot entry 0
    (post_impl.cc:56-62); entry i of block j sits at row column fc*j + (i-1), c0 (i == 0) at fc*j + fc-1 (out.cc:188-201)."""
    fc = ncep + 1
    X = fc * n_blocks
    cols = []
    for k in range(X):
        i = (k + 1) % X
        j, ii = divmod(i, fc)
        cols.append(fc * j + (fc - 1 if ii == 0 else ii - 1))
    return np.array(cols)


def cmvn_speakers(ids):
    """add_spk (post_impl.cc:120-142): speakers are numbered in order of first appearance in the list."""
    table, out = [], []
    for s in ids:
        if s not in table:
            table.append(s)
        out.append(table.index(s))
    return table, np.array(out, dtype=np.int32)


def cmvn_stats(rows_by_utt, spk_of_utt, n_spk, slot_cols):
    """sum_fea + stat_cm, then sum_cv + stat_cv (post_impl.cc:51-102): mean over all frames of a speaker, then the sum
    of squared deviations from that mean divided by count - 1."""
    X = len(slot_cols)
    mean = np.zeros((n_spk, X))
    count = np.zeros(n_spk)
    for r, s in zip(rows_by_utt, spk_of_utt):
        mean[s] += r[:, slot_cols].astype(np.float64).sum(0)
        count[s] += r.shape[0]
    mean /= count[:, None]
    var = np.zeros((n_spk, X))
    for r, s in zip(rows_by_utt, spk_of_utt):
        d = r[:, slot_cols].astype(np.float64) - mean[s]
        var[s] += (d * d).sum(0)
    var /= (count[:, None] - 1)
    return mean, var, count


def cmvn_apply(rows, spk, mean, var, slot_cols):
    """process_frame (post_impl.cc:104-118): (F - mean) / var - the variance, not its root."""
    out = rows.astype(np.float64).copy()
    out[:, slot_cols] = (out[:, slot_cols] - mean[spk]) / var[spk]
    return out.astype(np.float32)


def cmvn_stat_text(ids, mean, var):
    """cmvnOUT::save_frame (src/io/out.cc:591-613): "<id>\nmean\t%f %f ...\nvar\t%f ...\n" per speaker."""
    out = []
    for i, name in enumerate(ids):
        out.append("%s\nmean\t%s\nvar\t%s\n" % (name, " ".join("%f" % v for v in mean[i]), " ".join("%f" % v for v in var[i])))
    return "".join(out)
