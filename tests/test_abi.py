"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol, parses like the oracle,
designs bit-identical tables, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

import ctucopy_amd
from ctucopy_amd import CtuError, config_dims, config_table
from ctucopy_amd import build as cbuild
from ctucopy_amd import engine as ceng
from oracle.oracle import Oracle, OracleError
from tests.util import C1, C2, C3, C4, C5

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built():
    cbuild.build_engine()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "ctu_engine.h")).read()
    declared = set(re.findall(r"\b(ctu_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(ceng.EXPORTS)
    synth_hdr = open(os.path.join(ROOT, "include", "ctu_synth.h")).read()
    declared |= set(re.findall(r"\b(ctu_synth_[a-z0-9_]+)\s*\(", synth_hdr))
    assert {"ctu_synth_length", "ctu_synth_fill", "ctu_synth_fill_arena"} <= declared
    lib = ctypes.CDLL(cbuild.LIB)
    for name in declared:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("cfg", [C1, C2, C3, C5, C2 + ["-fea_E", "on"], C2 + ["-fea_c0", "off"],
                                 "-fs 8000 -preset plpc".split(), "-fs 16000 -preset mfcc -fea_kind logspec".split(),
                                 "-fs 16000 -preset mfcc -fb_shape rect -fb_definition 0-4000Hz:1-8/8filters,4000-8000Hz:1-4/4filters".split()])
def test_geometry_matches_oracle(cfg):
    d, o = config_dims(cfg), Oracle(cfg).dims
    assert (d.fs, d.window, d.wshift, d.wfft, d.nbins, d.nbands, d.row_floats, d.htk_kind, d.htk_period) == \
        (o.fs, o.window, o.wshift, o.wfft, o.K, o.B, o.D, o.htk_kind, o.period)


@pytest.mark.parametrize("cfg", [C1, C2, C3, C5, "-fs 8000 -preset plpc".split(),
                                 "-fs 16000 -preset mfcc -fb_scale bark -fb_eqld on".split(),
                                 "-fs 16000 -preset mfcc -fb_scale expolog -fb_norm off".split(),
                                 "-fs 16000 -preset mfcc -fb_scale lin -fb_shape rect -fb_definition 12filters".split()])
def test_host_design_tables_bit_identical_to_oracle(cfg):
    o = Oracle(cfg)
    assert np.array_equal(config_table(cfg, "hamming"), o.hamming())
    mat, first, last = o.fbank()
    assert np.array_equal(config_table(cfg, "fbank").reshape(mat.shape), mat)
    assert np.array_equal(config_table(cfg, "fb_first"), first)
    assert np.array_equal(config_table(cfg, "fb_last"), last)


def test_error_texts_follow_the_reference():
    for cfg, pat in ((["-preset", "mfcc"], "sampling rate"), ("-fs 16000 -bogus 1".split(), "Syntax error"),
                     ("-fs 16000 -preset nope".split(), "Unknown preset"),
                     ("-fs 16000 -preset mfcc -preem 1.0".split(), "Preemphasis"),
                     ("-fs 16000 -preset mfcc -fb_definition 3x".split(), "parse error"),
                     ("-fs 16000 -preset mfcc -fea_kind trapdct,100,5".split(), "must be odd")):
        with pytest.raises(CtuError, match=pat):
            config_dims(cfg)
        with pytest.raises(OracleError, match=pat):
            Oracle(cfg)


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(CtuError) as ei:
        ctucopy_amd.Engine(C2)
    assert ei.value.code == ceng.CTU_ERR_DEVICE and "no CPU fallback" in str(ei.value)


def test_unsupported_configurations_are_reported_not_approximated():
    for cfg in (C2 + ["-dither", "1.0"], C2 + ["-nr_mode", "hwss", "-vad", "burg", "-w", "30"], C2 + ["-stat_cmvn", "stat.txt", "-fea_c0", "off"], C2 + ["-apply_cmvn", "s", "-fea_Z_exp", "500"],
                C2 + ["-fea_kind", "spec", "-fea_Z_exp", "500"],
                C2 + ["-fea_delta", "d_a", "-fea_c0", "off"], C2 + ["-fea_kind", "logspec", "-fea_delta", "d"],
                C2 + ["-fea_delta", "d", "-d_win", "17"],
                C2 + ["-w", "300"],                               # 4800 samples: an 8192-point FFT
                C2 + ["-w", "100", "-nr_mode", "exten", "-nr_when", "afterFB"],   # exten after the filter bank with a 2048-point FFT (on the spectrum it runs since round 4)
                C2 + ["-w", "40", "-nr_mode", "fwss", "-vad", "burg"],   # the *ss modes with a 1024-point FFT
                C2 + ["-nr_mode", "fwss", "-vad", "burg", "-stat_cmvn", "s.txt"],   # the *ss modes with CMVN's two passes over the list
                C2 + ["-remove_dc1", "on", "-w", "25", "-s", "2"]):   # 12 frames over a sample
        with pytest.raises(CtuError) as ei:
            ctucopy_amd.Engine(cfg)
        assert ei.value.code == ceng.CTU_ERR_UNSUPPORTED


@pytest.mark.parametrize("cfg", [C1, C2, C3, C4, C5, "-fs 8000 -preset plpc".split(),
                                 C2 + ["-fb_definition", "40filters"], C2 + ["-fb_shape", "rect", "-fb_definition", "12filters"],
                                 C2 + ["-fb_definition", "100-4000Hz:1-10/10filters,4000-8000Hz:3-6/8filters"],
                                 C2 + ["-w", "20", "-s", "5"], C2 + ["-fb_scale", "lin", "-fb_definition", "3filters"]])
def test_phase2_chunk_tables_reproduce_the_filter_bank(cfg):
    # the kernel's band-group / chunk tables, rebuilt into dense rows on the host, must equal the float bank exactly
    cfg = [a for a in cfg if a not in ("-vad", "burg")]
    err, nchunks, nslots = config_table(cfg, "phase2_check")
    assert err == 0.0 and nchunks >= 1 and nslots >= 1


def _random_cfg(rng):
    fs = int(rng.choice([8000, 11025, 16000, 22050, 44100]))
    w = float(rng.choice([10, 16, 20, 25, 32]))
    s = float(rng.choice([5, 8, 10, 12.5, 16]))
    scale = str(rng.choice(["mel", "bark", "lin", "expolog"]))
    shape = str(rng.choice(["triang", "rect", "trapez"]))
    nf = int(rng.integers(3, 41))
    if rng.random() < 0.3:
        lo, mid, hi = sorted(int(x) for x in rng.choice(np.arange(100, fs // 2, 50), 3, replace=False))
        n1, n2 = int(rng.integers(2, 12)), int(rng.integers(2, 12))
        definition = f"{lo}-{mid}Hz:1-{n1}/{n1}filters,{mid}-{hi}Hz:2-{n2}/{n2}filters"
    else:
        definition = f"{nf}filters"
    cfg = ["-fs", str(fs), "-format_in", "raw", "-format_out", "htk", "-w", str(w), "-s", str(s), "-fb_scale", scale,
           "-fb_shape", shape, "-fb_definition", definition, "-fb_norm", str(rng.choice(["on", "off"])),
           "-fb_eqld", str(rng.choice(["on", "off"])), "-fea_kind", str(rng.choice(["dctc", "logspec", "lpc"])),
           "-fea_ncepcoefs", str(int(rng.integers(4, 17))), "-fea_lporder", str(int(rng.integers(4, 17))),
           "-fea_lifter", str(int(rng.choice([0, 1, 22, 30])))]
    return cfg


def test_random_configurations_design_the_same_tables_or_fail_alike():
    # 60 seeded random front-end configurations: both sides either reject the configuration, or agree bit-for-bit on
    # geometry, window, bank (values and first / last bin), DCT / cosine-iDFT matrix and lifter
    rng = np.random.default_rng(20260101)
    ok = 0
    for _ in range(60):
        cfg = _random_cfg(rng)
        try:
            o = Oracle(cfg)
        except OracleError:
            with pytest.raises(CtuError):
                config_dims(cfg)
            continue
        d, od = config_dims(cfg), o.dims
        assert (d.window, d.wshift, d.wfft, d.nbins, d.nbands, d.row_floats, d.htk_kind) == \
            (od.window, od.wshift, od.wfft, od.K, od.B, od.D, od.htk_kind), cfg
        assert np.array_equal(config_table(cfg, "hamming"), o.hamming()), cfg
        mat, first, last = o.fbank()
        assert np.array_equal(config_table(cfg, "fbank").reshape(mat.shape), mat), cfg
        assert np.array_equal(config_table(cfg, "fb_first"), first) and np.array_equal(config_table(cfg, "fb_last"), last), cfg
        ok += 1
    assert ok >= 30


def test_arena_layout_rule():
    """ctu_arena_layout (no engine, no device): utterance starts are multiples of pcm_align behind 8 samples of padding, regions do not
    overlap, 512 samples of padding close the arena - what ctu_plan_create reports (the GPU tests compare the two on real plans)."""
    L = ceng.load_library()
    i64 = ctypes.c_int64
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 17, 400):
        ns = rng.integers(0, 50000, size=n).astype(np.int64)
        if n > 2:
            ns[1] = 0
        off = np.full(n + 1, -1, dtype=np.int64)
        total = L.ctu_arena_layout(ns.ctypes.data_as(ctypes.POINTER(i64)), n, off.ctypes.data_as(ctypes.POINTER(i64)))
        assert off[0] == 8 and total == off[n] + 512
        assert np.all(off % 8 == 0) and np.all(off[1:] - off[:-1] >= ns) and np.all(off[1:] - off[:-1] < ns + 8)
        assert L.ctu_arena_layout(ns.ctypes.data_as(ctypes.POINTER(i64)), n, None) == total  # offsets optional
    bad = np.array([5, -1], dtype=np.int64)
    assert L.ctu_arena_layout(bad.ctypes.data_as(ctypes.POINTER(i64)), 2, None) < 0
    assert L.ctu_arena_layout(None, 3, None) < 0


def test_vad_ring_step_follows_the_filter_from_file_to_file():
    """ctu_vad_ring_step (pure): historyIdx / historySize of the VAD's majority filter after a file, as the oracle's list mode leaves them
    (pinned against the reference's own class in tests/test_oracle_median_ref.py)."""
    from tests.util import synth_utt
    L = ceng.load_library()
    rng = np.random.default_rng(8)
    for order in (1, 3, 5, 7):
        cfg = "-fs 8000 -format_in raw -format_out htk -preset mfcc -vad_out_mode vad -vad_cri_mode energy -vad_thr_mode adapt".split() + ["-vad_filter_order", str(order)]
        o = Oracle(cfg)
        hi, hs = ctypes.c_int32(0), ctypes.c_int32(0)
        first = True
        for T in [int(x) for x in rng.integers(0, 9, 12)]:
            o.process(synth_utt(3, 120 + 80 * T + (0 if T else 40), fs=8000), first_in_process=first)
            first = False
            L.ctu_vad_ring_step(order, T, ctypes.byref(hi), ctypes.byref(hs))
            assert (hi.value, hs.value) == o.vad_ring(), (order, T)
