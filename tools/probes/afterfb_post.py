"""-nr_when afterFB (exten on the band energies, src/io/batch.cc:207-210) followed by delta / stacking / CMS: errors against the oracle by both
measures of tests/test_gpu_parity.py::_assert_rows.  python tools/probes/afterfb_post.py   (GPU box)"""
import os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ctucopy_amd import Engine, CtuError
from oracle.oracle import Oracle
from tests.util import sig, synth_utt
import tests.test_gpu_parity as T
base = "-fs 8000 -format_in raw -format_out htk -preset mfcc -nr_mode exten -nr_when afterFB".split()
utts = [sig("CS3")[::2].copy(), synth_utt(77, 8000 * 3 + 123, fs=8000), synth_utt(78, 120 + 80 * 40, fs=8000)]
for extra in (["-fea_delta", "d_a"], ["-fea_delta", "d_a_t", "-fea_E", "on"], ["-fea_trap", "4"], ["-fea_Z_exp", "0.98"], ["-fea_Z_block", "50"], ["-fea_delta", "d", "-fea_Z_exp", "0.95"],
              ["-fs", "16000", "-fea_delta", "d_a", "-nr_a", "1"]):
    cfg = base + extra
    try:
        eng, orc = Engine(cfg), Oracle(cfg)
    except CtuError as e:
        print(extra, "REFUSED", e); continue
    uu = utts if "16000" not in extra else [sig("CS0"), synth_utt(79, 50000)]
    worst = rown = 0
    for u, g in zip(uu, eng.extract(uu)):
        ref = orc.process(u)
        assert g.shape == ref.shape, (extra, g.shape, ref.shape)
        worst = max(worst, float((np.abs(g - ref) / np.maximum(np.abs(ref), 1.0)).max()))
        rown = max(rown, float((np.abs(g - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0)).max()))
    print(extra, "worst element-wise %.3g, row-norm %.3g" % (worst, rown))
