#!/usr/bin/env python3
"""File-to-file throughput of the command-line host (bin/ctucopy): N seeded synthetic utterances written as raw PCM files,
one `-S` list, wall time of the whole process (engine creation, reading, H2D, kernels, D2H, writing HTK files).

  python tools/cli_e2e.py --utts 2000 [--bin bin/ctucopy] [--dir /tmp/ctu_e2e] [--set S-MFCC] [--runs 3] [-- extra CLI flags]

Prints one JSON line per run.  The first run of a directory also pays for the page cache; later ones read cached files."""
import argparse, json, os, subprocess, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ctucopy_amd import synth  # noqa: E402

FLAGS = {"S-MFCC": ["-fs", "16000", "-format_in", "raw", "-format_out", "htk", "-preset", "mfcc", "-preem", "0.97"],
         "S-NOISY": ["-fs", "8000", "-format_in", "raw", "-format_out", "htk", "-preset", "mfcc", "-nr_mode", "exten", "-nr_a", "2"],
         # C4 with its VAD: a four-column list (the VAD files go beside the HTK files)
         "S-NOISY-VAD": ["-fs", "8000", "-format_in", "raw", "-format_out", "htk", "-preset", "mfcc", "-nr_mode", "exten", "-nr_a", "2", "-vad", "burg", "-vad_out_mode", "vad",
                         "-vad_cri_mode", "cepdist", "-vad_cepdist_mode", "lpc", "-vad_thr_mode", "adapt"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2000)
    ap.add_argument("--set", default="S-MFCC")
    ap.add_argument("--dir", default="/tmp/ctu_e2e")
    ap.add_argument("--bin", default=os.path.join(ROOT, "bin", "ctucopy"))
    ap.add_argument("--runs", type=int, default=3)
    ap.add_argument("extra", nargs="*")
    a = ap.parse_args()
    din, dout = os.path.join(a.dir, "in"), os.path.join(a.dir, "out")
    os.makedirs(din, exist_ok=True)
    os.makedirs(dout, exist_ok=True)
    idx = list(range(a.utts))
    sid = synth.SET_NOISY if a.set.startswith("S-NOISY") else synth.SET_SPEECH
    ns = synth.lengths(sid, idx)
    off = np.zeros(a.utts + 1, dtype=np.int64)
    off[1:] = np.cumsum((np.asarray(ns) + 7) // 8 * 8)
    t0 = time.time()
    arena = synth.fill_arena(sid, idx, off, int(off[-1]))
    arena = np.asarray(arena)
    lst = os.path.join(a.dir, "list.scp")
    with open(lst, "w") as f:
        for i in idx:
            p = os.path.join(din, "u%05d.raw" % i)
            if not os.path.exists(p) or os.path.getsize(p) != 2 * ns[i]:
                arena[off[i]:off[i] + ns[i]].tofile(p)
            if a.set.endswith("-VAD"):
                f.write("%s %s spk %s\n" % (p, os.path.join(dout, "u%05d.htk" % i), os.path.join(dout, "u%05d.vad" % i)))
            else:
                f.write("%s %s\n" % (p, os.path.join(dout, "u%05d.htk" % i)))
    fs = synth.fs_of(sid)
    win, hop = fs * 25 // 1000, fs // 100
    frames = int(sum((n - (win - hop)) // hop for n in ns))
    print("generated %d files, %.1f MB, %d frames in %.1f s" % (a.utts, 2 * sum(ns) / 1e6, frames, time.time() - t0), file=sys.stderr)
    for r in range(a.runs):
        for f in os.listdir(dout):
            os.unlink(os.path.join(dout, f))
        t = time.time()
        cp = subprocess.run([a.bin] + FLAGS[a.set] + ["-S", lst] + a.extra, capture_output=True, text=True, env=dict(os.environ, CTU_HOST_TIMING="1"))
        wall = time.time() - t
        nout = len(os.listdir(dout))
        size = sum(os.path.getsize(os.path.join(dout, f)) for f in os.listdir(dout))
        print(json.dumps({"bin": os.path.basename(a.bin), "extra": a.extra, "run": r, "rc": cp.returncode, "wall_s": round(wall, 3), "files_out": nout,
                          "bytes_out": size, "frames": frames, "frames_per_s": round(frames / wall), "pcm_MB_per_s": round(2 * sum(ns) / 1e6 / wall, 1),
                          "stderr": cp.stderr.strip()[-200:]}))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
