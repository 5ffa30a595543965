import sys, numpy as np
sys.path.insert(0, '.')
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C2, C4, sig, synth_utt
for name, cfg, utts in (("C4", C4, [sig("CS3"), sig("CS0"), synth_utt(91, 24000, fs=8000), synth_utt(92, 64000, fs=8000)]),
                        ("C2+burg adapt", C2 + ["-vad", "burg", "-vad_out_mode", "vad", "-vad_cri_mode", "cepdist", "-vad_thr_mode", "adapt"], [sig("CS0"), sig("CS3"), synth_utt(93, 60000)])):
    rows, vads = Engine(cfg).extract(utts, want_vad=True)
    orc = Oracle(cfg)
    agree = total = 0
    for u, v in zip(utts, vads):
        _, rv = orc.process(u, want_vad=True)
        agree += int((v == rv).sum()); total += v.size
    print(name, "agreement %.5f" % (agree / total), "of", total, "ones in first:", int((vads[0] == ord('1')).sum()))
