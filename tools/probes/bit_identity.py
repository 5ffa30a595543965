"""Are the headline rows of two engine builds bit-identical?  The in-tree library against ctucopy_amd/_variants/lib_prev.so (CTU_ENGINE_LIB), 300 S-MFCC
utterances, device-resident run.  python tools/probes/bit_identity.py"""
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if len(sys.argv) > 1:
    from ctucopy_amd import Engine, synth
    from tests.util import C2
    import torch
    eng = Engine(C2)
    idx = list(range(300))
    plan = eng.plan(synth.lengths(synth.SET_SPEECH, idx))
    host = synth.fill_arena(synth.SET_SPEECH, idx, plan.sample_off, plan.total_samples)
    pcm = torch.from_numpy(np.asarray(host)).cuda()
    rows = eng.run_device(plan, pcm)
    torch.cuda.synchronize()
    np.save(sys.argv[1], rows.cpu().numpy())
else:
    for name, lib in (("new", None), ("old", "ctucopy_amd/_variants/lib_prev.so")):
        env = dict(os.environ)
        if lib: env["CTU_ENGINE_LIB"] = lib
        subprocess.run([sys.executable, __file__, "/tmp/rows_%s.npy" % name], check=True, env=env)
    a, b = np.load("/tmp/rows_new.npy"), np.load("/tmp/rows_old.npy")
    print("rows", a.shape, "bit-identical:", bool(np.array_equal(a.view(np.uint32), b.view(np.uint32))), "max abs diff", float(np.abs(a - b).max()))
