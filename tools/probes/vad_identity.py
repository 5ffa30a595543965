"""Do two engine builds take the same VAD decisions and write the same rows?  The in-tree library against ctucopy_amd/_variants/lib_prev.so
(CTU_ENGINE_LIB) on N utterances of S-NOISY at 8 kHz (C4) and of S-MFCC at 16 kHz with the Burg-cepstral VAD, device-resident runs.
python tools/probes/vad_identity.py [N]"""
import os, sys, subprocess, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
N = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 3000
if len(sys.argv) > 2:
    from ctucopy_amd import Engine, synth
    from tests.util import C2, C4
    import torch
    out = {}
    for name, cfg, sid in (("c4", C4, synth.SET_NOISY), ("c2vad16", C2 + "-vad burg -vad_out_mode vad -vad_cri_mode cepdist -vad_cepdist_mode lpc -vad_thr_mode adapt".split(), synth.SET_SPEECH)):
        eng = Engine(cfg)
        idx = list(range(N))
        plan = eng.plan(synth.lengths(sid, idx))
        host = synth.fill_arena(sid, idx, plan.sample_off, plan.total_samples)
        pcm = torch.from_numpy(np.asarray(host)).cuda()
        vad = torch.zeros(plan.total_frames, dtype=torch.uint8, device="cuda")
        rows = eng.run_device(plan, pcm, vad=vad)
        torch.cuda.synchronize()
        out[name + "_rows"] = rows.cpu().numpy()
        out[name + "_vad"] = vad.cpu().numpy()
    np.savez(sys.argv[2], **out)
else:
    for name, lib in (("new", None), ("old", "ctucopy_amd/_variants/lib_prev.so")):
        env = dict(os.environ)
        if lib: env["CTU_ENGINE_LIB"] = lib
        subprocess.run([sys.executable, __file__, str(N), "/tmp/vadid_%s.npz" % name], check=True, env=env)
    a, b = np.load("/tmp/vadid_new.npz"), np.load("/tmp/vadid_old.npz")
    for k in a.files:
        same = np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))
        extra = "" if same else "  differing entries: %d of %d" % (int((a[k] != b[k]).sum()), a[k].size)
        print("%-14s %s %s%s" % (k, a[k].shape, "identical" if same else "DIFFERENT", extra))
    print("speech frames: c4 %d of %d, c2vad16 %d of %d" % (int((a["c4_vad"] == ord("1")).sum()), a["c4_vad"].size, int((a["c2vad16_vad"] == ord("1")).sum()), a["c2vad16_vad"].size))
