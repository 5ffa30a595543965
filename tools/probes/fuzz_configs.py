"""A wider net than the suite's three seeds: the random front-end configurations of tests/test_gpu_parity.py
(test_random_configurations_match_the_oracle, ..._on_1024_points) for many seeds, failures collected instead of stopping at the first.
python tools/probes/fuzz_configs.py [first_seed] [n_seeds]          (GPU box; a few minutes for 30 seeds)"""
import os, sys, traceback
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import tests.test_gpu_parity as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
from ctucopy_amd import Engine as E, load_library
load_library()
bad = []
runs = 0
for seed in range(first, first + n):
    for fs in (16000, 8000):
        try:
            T.test_random_configurations_match_the_oracle.__wrapped__(E, seed, fs) if hasattr(T.test_random_configurations_match_the_oracle, "__wrapped__") else T.test_random_configurations_match_the_oracle(E, seed, fs)
            runs += 1
        except AssertionError as e:
            msg = str(e).split("\n")[0][:600]
            if msg.startswith("(") and "refused" not in msg and len(msg) < 12:  # "ran >= 20" bookkeeping, not a mismatch
                runs += 1
                continue
            bad.append(("main", seed, fs, msg))
        except Exception as e:
            bad.append(("main-exc", seed, fs, repr(e)[:600]))
    try:
        T.test_random_configurations_on_1024_points(E, seed)
        runs += 1
    except AssertionError as e:
        msg = str(e).split("\n")[0][:600]
        if len(msg) < 6:
            runs += 1
        else:
            bad.append(("1024", seed, 16000, msg))
    except Exception as e:
        bad.append(("1024-exc", seed, 16000, repr(e)[:600]))
    print("seed", seed, "done; failures so far:", len(bad), flush=True)
print("sweeps without a mismatch: %d; mismatches / errors: %d" % (runs, len(bad)))
for b in bad:
    print(b)
