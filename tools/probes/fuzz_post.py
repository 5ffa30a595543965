"""The suite's seeded random post-processing chains (delta / stacking / CMS) and enhancement configurations under other seeds:
np.random.default_rng is wrapped so that the tests' fixed seeds are offset.  python tools/probes/fuzz_post.py [n_offsets]   (GPU box)"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tests.test_gpu_parity as T
from ctucopy_amd import Engine as E, load_library
load_library()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
real = np.random.default_rng
bad = []
for off in range(1, n + 1):
    np.random.default_rng = lambda seed=None, _o=off: real(None if seed is None else seed + 1000 * _o)
    for name in ("test_random_post_processing_chains", "test_random_enhancement_configurations"):
        try:
            getattr(T, name)(E)
        except AssertionError as e:
            bad.append((name, off, str(e).split("\n")[0][:500]))
        except Exception as e:
            bad.append((name + " EXC", off, repr(e)[:500]))
    print("offset", off, "failures so far", len(bad), flush=True)
np.random.default_rng = real
print("offsets %d, mismatches / errors %d" % (n, len(bad)))
for b in bad:
    print(b)
