#!/usr/bin/env python3
"""Diagnostic: (1) hwss / fwss / 2fwss rows against the oracle by both measures of tests/test_gpu_parity.py::_assert_rows
(element-wise, and against the row's largest value); (2) the 16 kHz VAD modes file by file: bytes that differ and where the
first one is (a decision that flips changes the detector's state for the rest of the file)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ctucopy_amd import Engine, synth
from oracle.oracle import Oracle
from tests.util import C2, sig, synth_utt

SS8 = "-fs 8000 -format_in raw -format_out htk -preset mfcc -preem 0.97 -vad burg".split()
utts = [synth.utterance_c(synth.SET_NOISY, i, True) for i in (2, 5, 9, 12)] + [synth_utt(17, 2000, fs=8000), synth_utt(18, 120, fs=8000), synth_utt(19, 9000, fs=8000)]
for extra in (["-nr_mode", "hwss", "-fea_kind", "spec"], ["-nr_mode", "hwss"], ["-nr_mode", "hwss", "-fea_kind", "logspec"], ["-nr_mode", "fwss"],
              ["-nr_mode", "fwss", "-fea_kind", "spec"], ["-nr_mode", "2fwss"]):
    cfg = SS8 + extra
    got, orc = Engine(cfg).extract(utts), Oracle(cfg)
    el, rn = 0.0, 0.0
    for u, g in zip(utts, got):
        ref = orc.process(u)
        if not ref.size:
            continue
        el = max(el, float((np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)).max()))
        rn = max(rn, float((np.abs(g - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0)).max()))
    print("ss", " ".join(extra), f"element-wise {el:.3e} row-norm {rn:.3e}", flush=True)

modes = [["-vad_out_mode", "vad", "-vad_cri_mode", "energy", "-vad_thr_mode", "perc"],
         ["-vad_out_mode", "vad", "-vad_cri_mode", "energy", "-vad_thr_mode", "adapt"],
         ["-vad_out_mode", "vad", "-vad_cri_mode", "energy", "-vad_thr_mode", "dyn", "-vad_filter_order", "5"],
         ["-vad_out_mode", "vad", "-vad_cri_mode", "energy", "-vad_thr_mode", "absolute", "-vad_absolute_thr", "150"],
         ["-vad_out_mode", "vad", "-vad_cri_mode", "cepdist", "-vad_cepdist_mode", "fea", "-vad_thr_mode", "adapt"],
         ["-vad", "burg", "-vad_out_mode", "vad", "-vad_cri_mode", "cepdist", "-vad_thr_mode", "adapt"]]
files = [sig("CS0")[:48000], synth_utt(92, 30000), sig("CS3")[:48000], synth_utt(93, 50000)] + [synth.utterance_c(synth.SET_SPEECH, i, True) for i in range(6)]
for extra in modes:
    cfg = C2 + extra
    rows, vads = Engine(cfg).extract(files, want_vad=True)
    orc = Oracle(cfg)
    rep = []
    for u, v in zip(files, vads):
        _, rv = orc.process(u, want_vad=True)
        d = np.nonzero(v != rv)[0]
        rep.append(f"{d.size}/{v.size}" + (f"@{d[0]}" if d.size else ""))
    print("vad16", " ".join(extra[2:]), "differing bytes per file:", " ".join(rep), flush=True)
