"""ctypes binding of include/ctu_engine.h -- the only way Python reaches the hot path.

PyTorch is used purely as plumbing (device buffers and streams); the computation is the HIP library.
There is no CPU fallback: creating an Engine without the built library or without a GPU raises.
"""
import ctypes
import os
import weakref

import numpy as np

from . import build as _build

CTU_OK, CTU_ERR_OPTS, CTU_ERR_UNSUPPORTED, CTU_ERR_DEVICE, CTU_ERR_INPUT = 0, -1, -2, -3, -4


class CtuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class Dims(ctypes.Structure):
    _fields_ = [("fs", ctypes.c_int32), ("window", ctypes.c_int32), ("wshift", ctypes.c_int32),
                ("wfft", ctypes.c_int32), ("nbins", ctypes.c_int32), ("nbands", ctypes.c_int32),
                ("row_floats", ctypes.c_int32), ("htk_kind", ctypes.c_int32), ("htk_period", ctypes.c_uint32),
                ("has_vad", ctypes.c_int32), ("swap_out", ctypes.c_int32), ("pcm_align", ctypes.c_int32),
                ("signal_out", ctypes.c_int32)]


# every symbol include/ctu_engine.h declares (checked by tests/test_abi.py)
EXPORTS = ["ctu_engine_create", "ctu_engine_destroy", "ctu_create_error", "ctu_last_error", "ctu_engine_dims",
           "ctu_config_dims", "ctu_config_table", "ctu_num_frames", "ctu_plan_create", "ctu_plan_destroy", "ctu_plan_sample_offsets",
           "ctu_plan_row_offsets", "ctu_arena_layout", "ctu_plan_total_samples", "ctu_plan_total_frames", "ctu_engine_run",
           "ctu_engine_run_host", "ctu_host_alloc", "ctu_host_free", "ctu_engine_reset_chain", "ctu_engine_set_vad_stream", "ctu_vad_ring_step", "ctu_plan_set_vad_ring", "ctu_vad_ring_rows", "ctu_decode_g711", "ctu_engine_last_kernel_ms", "ctu_engine_kernel_name", "ctu_cmvn_cols", "ctu_cmvn_accumulate", "ctu_cmvn_apply",
           "ctu_cmvn_accumulate_host", "ctu_cmvn_apply_host", "ctu_plan_out_samples", "ctu_engine_run_signal",
           "ctu_engine_run_signal_host"]

_lib = None


def load_library():
    """Loads ctucopy_amd/libctu_engine.so (must have been built: __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CTU_ENGINE_LIB", _build.LIB)  # override only for A/B experiments on kernel builds
    if not os.path.exists(path):
        raise CtuError(CTU_ERR_DEVICE, f"{path} is missing: run __graft_entry__.build() (hipcc, gfx950) first; "
                                       "there is no CPU fallback")
    L = ctypes.CDLL(path)
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    argv_t = ctypes.POINTER(ctypes.c_char_p)
    L.ctu_engine_create.argtypes = [ctypes.c_int, argv_t, ctypes.c_int, ctypes.POINTER(vp)]
    L.ctu_engine_destroy.argtypes = [vp]
    L.ctu_create_error.restype = ctypes.c_char_p
    L.ctu_last_error.restype = ctypes.c_char_p
    L.ctu_last_error.argtypes = [vp]
    L.ctu_engine_dims.argtypes = [vp, ctypes.POINTER(Dims)]
    L.ctu_config_dims.argtypes = [ctypes.c_int, argv_t, ctypes.POINTER(Dims)]
    L.ctu_config_table.restype = i64
    L.ctu_config_table.argtypes = [ctypes.c_int, argv_t, ctypes.c_char_p, vp, i64]
    L.ctu_num_frames.restype = i64
    L.ctu_num_frames.argtypes = [vp, i64]
    L.ctu_plan_create.argtypes = [vp, ctypes.POINTER(i64), i32, ctypes.POINTER(vp)]
    L.ctu_plan_destroy.argtypes = [vp]
    L.ctu_plan_sample_offsets.restype = ctypes.POINTER(i64)
    L.ctu_plan_sample_offsets.argtypes = [vp]
    L.ctu_plan_row_offsets.restype = ctypes.POINTER(i64)
    L.ctu_plan_row_offsets.argtypes = [vp]
    L.ctu_arena_layout.restype = i64
    L.ctu_arena_layout.argtypes = [ctypes.POINTER(i64), ctypes.c_int32, ctypes.POINTER(i64)]
    L.ctu_plan_total_samples.restype = i64
    L.ctu_plan_total_samples.argtypes = [vp]
    L.ctu_plan_total_frames.restype = i64
    L.ctu_plan_total_frames.argtypes = [vp]
    L.ctu_engine_run.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ctu_engine_run_host.argtypes = [vp, vp, vp, vp, vp, vp]
    L.ctu_engine_reset_chain.argtypes = [vp]
    L.ctu_engine_set_vad_stream.argtypes = [vp, ctypes.c_char_p, i64]
    L.ctu_vad_ring_step.restype = None
    L.ctu_vad_ring_step.argtypes = [ctypes.c_int32, i64, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
    L.ctu_plan_set_vad_ring.argtypes = [vp, ctypes.POINTER(ctypes.c_int32)]
    L.ctu_vad_ring_rows.restype = i64
    L.ctu_vad_ring_rows.argtypes = [ctypes.c_int32, i64, ctypes.c_int32, ctypes.POINTER(ctypes.c_int32)]
    L.ctu_decode_g711.argtypes = [vp, vp, i64, ctypes.c_int, vp, vp]
    L.ctu_host_alloc.restype = vp
    L.ctu_host_alloc.argtypes = [ctypes.c_size_t]
    L.ctu_host_free.argtypes = [vp]
    L.ctu_engine_last_kernel_ms.restype = ctypes.c_float
    L.ctu_engine_last_kernel_ms.argtypes = [vp]
    L.ctu_engine_kernel_name.restype = ctypes.c_char_p
    L.ctu_engine_kernel_name.argtypes = [vp]
    L.ctu_cmvn_cols.argtypes = [vp]
    L.ctu_cmvn_accumulate.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    L.ctu_cmvn_apply.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
    L.ctu_cmvn_accumulate_host.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    L.ctu_plan_out_samples.restype = ctypes.POINTER(i64)
    L.ctu_plan_out_samples.argtypes = [vp]
    L.ctu_engine_run_signal.argtypes = [vp, vp, vp, vp, vp]
    L.ctu_engine_run_signal_host.argtypes = [vp, vp, vp, vp]
    L.ctu_cmvn_apply_host.argtypes = [vp, vp, vp, vp, i32, vp, vp]
    _lib = L
    return L


def host_alloc(shape, dtype):
    """numpy array over page-locked host memory from ctu_host_alloc (DMA at the link rate in the *_host calls)."""
    L = load_library()
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    ptr = L.ctu_host_alloc(n)
    if not ptr:
        raise CtuError(CTU_ERR_DEVICE, "ctu_host_alloc failed")
    buf = (ctypes.c_char * n).from_address(ptr)
    # The block lives as long as the ctypes object every derived array ends up holding as its buffer: np.asarray(),
    # np.ascontiguousarray(), slices and .view() of the result all keep `buf` reachable through .base, whatever happens to
    # the first array object (an owner attribute on an ndarray subclass is dropped by those calls while DMA may still run).
    weakref.finalize(buf, L.ctu_host_free, ptr)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def _argv(args):
    args = [str(a).encode() for a in args]
    return len(args), (ctypes.c_char_p * len(args))(*args)


def config_dims(args):
    """Geometry of a command line without touching a GPU (parse + host-side table design)."""
    L = load_library()
    d = Dims()
    n, arr = _argv(args)
    rc = L.ctu_config_dims(n, arr, ctypes.byref(d))
    if rc != CTU_OK:
        raise CtuError(rc, L.ctu_create_error().decode())
    return d


def config_table(args, name):
    """Host-designed table (float64 numpy) by name; see ctu_config_table in include/ctu_engine.h."""
    L = load_library()
    n, arr = _argv(args)
    cnt = L.ctu_config_table(n, arr, name.encode(), None, 0)
    if cnt < 0:
        raise CtuError(int(cnt), L.ctu_create_error().decode())
    out = np.zeros(int(cnt), dtype=np.float64)
    L.ctu_config_table(n, arr, name.encode(), out.ctypes.data, cnt)
    return out


class Plan:
    """Batch layout: where each utterance sits in the packed PCM arena and in the output rows."""

    def __init__(self, engine, nsamples):
        L = load_library()
        self.engine = engine
        ns = np.ascontiguousarray(nsamples, dtype=np.int64)
        self.n_utt = int(ns.size)
        h = ctypes.c_void_p()
        rc = L.ctu_plan_create(engine._h, ns.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), self.n_utt, ctypes.byref(h))
        if rc != CTU_OK:
            raise CtuError(rc, L.ctu_last_error(engine._h).decode())
        self._h = h
        self.nsamples = ns
        self.sample_off = np.ctypeslib.as_array(L.ctu_plan_sample_offsets(h), shape=(self.n_utt + 1,)).copy()
        self.row_off = np.ctypeslib.as_array(L.ctu_plan_row_offsets(h), shape=(self.n_utt + 1,)).copy()
        self.total_samples = int(L.ctu_plan_total_samples(h))
        self.total_frames = int(L.ctu_plan_total_frames(h))

    def set_vad_ring(self, hidx):
        """The historyIdx of the VAD's majority filter every utterance starts with (None: all in phase): the reference's list behaviour,
        see include/ctu_engine.h."""
        L = load_library()
        if hidx is None:
            rc = L.ctu_plan_set_vad_ring(self._h, None)
        else:
            a = np.ascontiguousarray(hidx, dtype=np.int32)
            assert a.size == self.n_utt
            rc = L.ctu_plan_set_vad_ring(self._h, a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        if rc != CTU_OK:
            raise CtuError(rc, L.ctu_last_error(self.engine._h).decode())

    def pack(self, utterances):
        """int16 arena (numpy) holding the utterances at their planned offsets."""
        arena = np.zeros(self.total_samples, dtype=np.int16)
        for u, off in zip(utterances, self.sample_off[:-1]):
            arena[off:off + len(u)] = u
        return arena

    def close(self):
        if getattr(self, "_h", None):
            load_library().ctu_plan_destroy(self._h)
            self._h = None

    __del__ = close


class Engine:
    """One configured chain on one GPU; `args` is the ctucopy command line (list of str)."""

    def __init__(self, args, device=0):
        L = load_library()
        n, arr = _argv(args)
        h = ctypes.c_void_p()
        rc = L.ctu_engine_create(n, arr, int(device), ctypes.byref(h))
        if rc != CTU_OK:
            raise CtuError(rc, L.ctu_create_error().decode())
        self._h = h
        self.device = int(device)
        self.dims = Dims()
        L.ctu_engine_dims(h, ctypes.byref(self.dims))

    def close(self):
        if getattr(self, "_h", None):
            load_library().ctu_engine_destroy(self._h)
            self._h = None

    __del__ = close

    def num_frames(self, nsamples):
        return int(load_library().ctu_num_frames(self._h, int(nsamples)))

    def plan(self, nsamples):
        return Plan(self, nsamples)

    def _check(self, rc):
        if rc != CTU_OK:
            raise CtuError(rc, load_library().ctu_last_error(self._h).decode())

    def run_device(self, plan, pcm, rows=None, vad=None, stream=None):
        """pcm: torch int16 CUDA tensor [plan.total_samples]; returns the float32 rows tensor (async)."""
        import torch
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.numel() >= plan.total_samples
        if rows is None:
            rows = torch.empty((plan.total_frames, self.dims.row_floats), dtype=torch.float32, device=pcm.device)
        s = stream if stream is not None else torch.cuda.current_stream(pcm.device)
        self._check(load_library().ctu_engine_run(self._h, plan._h, pcm.data_ptr(), rows.data_ptr(),
                                                  vad.data_ptr() if vad is not None else None, s.cuda_stream))
        return rows

    def run_host(self, plan, arena, want_vad=False, rows_out=None):
        """numpy int16 arena in, numpy float32 rows out (H2D + kernels + D2H inside the library).

        With want_vad: (rows, vad bytes '0'/'1' per frame, rows actually kept per utterance)."""
        arena = np.ascontiguousarray(arena, dtype=np.int16)
        assert arena.size >= plan.total_samples
        rows = rows_out if rows_out is not None else np.empty((plan.total_frames, self.dims.row_floats), dtype=np.float32)
        per = np.zeros(plan.n_utt, dtype=np.int64)
        vad = np.zeros(max(plan.total_frames, 1), dtype=np.uint8)
        self._check(load_library().ctu_engine_run_host(self._h, plan._h, arena.ctypes.data, rows.ctypes.data,
                                                       vad.ctypes.data if self.dims.has_vad else None, per.ctypes.data))
        if want_vad:
            return rows, vad[:plan.total_frames], per
        return rows

    # ---- speech enhancement output (-format_out raw|wave)
    def enhance(self, utterances):
        """list of int16 arrays -> list of int16 arrays (frames*wshift + window-wshift samples each)."""
        L = load_library()
        plan = self.plan([len(u) for u in utterances])
        arena = plan.pack(utterances)
        out = np.zeros(plan.total_samples, dtype=np.int16)
        self._check(L.ctu_engine_run_signal_host(self._h, plan._h, arena.ctypes.data, out.ctypes.data))
        n = L.ctu_plan_out_samples(plan._h)
        res = [out[plan.sample_off[i]:plan.sample_off[i] + n[i]].copy() for i in range(plan.n_utt)]
        plan.close()
        return res

    def enhance_device(self, plan, pcm, out=None, stream=None):
        """pcm: torch int16 CUDA tensor [plan.total_samples] -> int16 CUDA tensor of the same layout (async)."""
        import torch
        assert pcm.is_cuda and pcm.dtype == torch.int16 and pcm.numel() >= plan.total_samples
        if out is None:
            out = torch.zeros(plan.total_samples, dtype=torch.int16, device=pcm.device)
        s = stream if stream is not None else torch.cuda.current_stream(pcm.device)
        self._check(load_library().ctu_engine_run_signal(self._h, plan._h, pcm.data_ptr(), out.data_ptr(), s.cuda_stream))
        return out

    # ---- per-speaker CMVN over device-resident rows (include/ctu_engine.h; src/fea/post_impl.cc:51-118)
    def cmvn_cols(self):
        return int(load_library().ctu_cmvn_cols(self._h))

    def cmvn_accumulate(self, plan, rows, spk_of_utt, n_spk, mean=None, acc=None, stream=None):
        """acc[n_spk, cols+1] (float64) += sums (mean is None) or sums of squared deviations from `mean`, and counts."""
        import torch
        cols = self.cmvn_cols()
        spk = np.ascontiguousarray(spk_of_utt, dtype=np.int32)
        if acc is None:
            acc = np.zeros((n_spk, cols + 1), dtype=np.float64)
        assert acc.shape == (n_spk, cols + 1) and acc.dtype == np.float64 and acc.flags.c_contiguous
        m = None if mean is None else np.ascontiguousarray(mean, dtype=np.float64)
        s = stream if stream is not None else torch.cuda.current_stream(rows.device)
        self._check(load_library().ctu_cmvn_accumulate(self._h, plan._h, rows.data_ptr(), spk.ctypes.data, int(n_spk),
                                                       m.ctypes.data if m is not None else None, acc.ctypes.data, s.cuda_stream))
        return acc

    def cmvn_apply(self, plan, rows, spk_of_utt, n_spk, mean, var, stream=None):
        import torch
        spk = np.ascontiguousarray(spk_of_utt, dtype=np.int32)
        m = np.ascontiguousarray(mean, dtype=np.float64)
        v = np.ascontiguousarray(var, dtype=np.float64)
        s = stream if stream is not None else torch.cuda.current_stream(rows.device)
        self._check(load_library().ctu_cmvn_apply(self._h, plan._h, rows.data_ptr(), spk.ctypes.data, int(n_spk),
                                                  m.ctypes.data, v.ctypes.data, s.cuda_stream))
        return rows

    def decode_g711(self, codes, alaw=True, stream=None):
        """torch uint8 CUDA tensor of G.711 codes -> int16 CUDA tensor (the reference's expansion, on the device)."""
        import torch
        assert codes.is_cuda and codes.dtype == torch.uint8
        out = torch.empty(codes.numel(), dtype=torch.int16, device=codes.device)
        s = stream if stream is not None else torch.cuda.current_stream(codes.device)
        self._check(load_library().ctu_decode_g711(self._h, codes.data_ptr(), codes.numel(), 1 if alaw else 0, out.data_ptr(), s.cuda_stream))
        return out

    def set_vad_stream(self, data):
        """-vad file=...: replaces the byte stream the hwss / fwss / 2fwss decisions are read from (one byte per frame) and rewinds it."""
        data = bytes(data)
        self._check(load_library().ctu_engine_set_vad_stream(self._h, data, len(data)))

    def reset_chain(self):
        """hwss / fwss / 2fwss: forget the spectrum vector the previous run left behind (see include/ctu_engine.h)."""
        self._check(load_library().ctu_engine_reset_chain(self._h))

    def last_kernel_ms(self):
        return float(load_library().ctu_engine_last_kernel_ms(self._h))

    def kernel_name(self):
        return load_library().ctu_engine_kernel_name(self._h).decode()

    def vad_ring_of_list(self, nsamples, order):
        """historyIdx each file of a list starts with when the list is one process's (ctu_vad_ring_step from 0, 0)."""
        L = load_library()
        hi, hs = ctypes.c_int32(0), ctypes.c_int32(0)
        out = []
        for n in nsamples:
            out.append(hi.value)
            L.ctu_vad_ring_step(int(order), max(self.num_frames(int(n)), 0), ctypes.byref(hi), ctypes.byref(hs))
        return out

    def extract(self, utterances, want_vad=False, as_list_of_one_process=None):
        """Convenience: list of int16 arrays -> list of [rows, D] float32 arrays (and the VAD byte strings).

        as_list_of_one_process = the VAD's filter order: the utterances are the files of one list, the majority filter's ring index runs
        on from file to file as in the reference (Plan.set_vad_ring)."""
        plan = self.plan([len(u) for u in utterances])
        if as_list_of_one_process:
            plan.set_vad_ring(self.vad_ring_of_list([len(u) for u in utterances], as_list_of_one_process))
        rows, vad, per = self.run_host(plan, plan.pack(utterances), want_vad=True)
        out = [rows[plan.row_off[i]:plan.row_off[i] + per[i]] for i in range(plan.n_utt)]
        vads = [vad[plan.row_off[i]:plan.row_off[i + 1]] for i in range(plan.n_utt)]
        vads = [v[v != 0] for v in vads]  # NUL = nothing written (an utterance the majority filter never got ready on)
        plan.close()
        return (out, vads) if want_vad else out
