"""The `ctucopy` executable: command line, list format, decoders and the HTK / ark / pfile writers.

Byte-level anchors come from outputs of the compiled reference recorded in SURVEY.md 8(b)/(c):
HTK files of 30900 / 30796 bytes with header (n, 100000, 52, 8198); ark offsets 5 and 30913; a pfile of
103940 bytes for the same two utterances.
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from ctucopy_amd import build as cbuild
from oracle.oracle import Oracle, htk_bytes
from tests.util import C1, GOLDEN, sig

CLI = cbuild.CLI
TOL = 1e-4


@pytest.fixture(scope="module", autouse=True)
def _built():
    cbuild.build_cli()


def run(args, cwd=None):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True, cwd=cwd)


def test_errors_without_gpu_work(tmp_path):
    r = run([])
    assert r.returncode == 255 and "No command line options" in r.stderr
    r = run(["-preset", "mfcc", "-S", "x"])
    assert r.returncode == 255 and "sampling rate" in r.stderr
    r = run("-fs 16000 -format_in raw -format_out htk -preset mfcc -S /nonexistent/list".split())
    assert r.returncode == 255 and "Cannot open list file" in r.stderr
    r = run("-fs 16000 -format_in raw -format_out raw -preset exten -S x".split())
    assert r.returncode == 255
    assert run(["-h"]).returncode == 0


def _list(tmp_path, names, cols=2):
    lines = []
    for n in names:
        cols_ = [os.path.join(GOLDEN, "SA000CB1." + n), str(tmp_path / (n + ".out"))]
        if cols > 2:
            cols_ += ["spk", str(tmp_path / (n + ".vad"))]
        lines.append(" ".join(cols_))
    p = tmp_path / "list.scp"
    p.write_text("\n".join(lines) + "\n")
    return str(p)


def _close(a, b):
    return (np.abs(a - b) / np.maximum(np.abs(b), 1.0)).max() <= TOL


@pytest.mark.gpu
def test_htk_output_matches_reference_anchors(tmp_path):
    lst = _list(tmp_path, ["CS0", "CS3"])
    r = run(C1 + ["-S", lst, "-v"])
    assert r.returncode == 0, r.stderr
    assert "594 frames" in r.stderr and "592 frames" in r.stderr
    orc = Oracle(C1)
    for name, size, frames in (("CS0", 30900, 594), ("CS3", 30796, 592)):
        img = (tmp_path / (name + ".out")).read_bytes()
        assert len(img) == size
        assert struct.unpack("<IIHH", img[:12]) == (frames, 100000, 52, 8198)
        got = np.frombuffer(img[12:], dtype="<f4").reshape(frames, 13)
        ref = orc.process(sig(name))
        assert _close(got, ref)
        assert img[:12] == htk_bytes(ref, 100000, 8198)[:12]


@pytest.mark.gpu
def test_big_endian_and_single_file_mode(tmp_path):
    out = tmp_path / "be.htk"
    r = run(C1[:8] + ["-endian_out", "big"] + C1[8:] + ["-endian_out", "big", "-i", os.path.join(GOLDEN, "SA000CB1.CS3"),
                                                       "-o", str(out)])
    assert r.returncode == 0, r.stderr
    img = out.read_bytes()
    assert struct.unpack(">IIHH", img[:12]) == (592, 100000, 52, 8198)
    got = np.frombuffer(img[12:], dtype=">f4").reshape(592, 13)
    assert _close(got.astype(np.float32), Oracle(C1).process(sig("CS3")))


@pytest.mark.gpu
def test_ark_scp_and_pfile(tmp_path):
    lst = tmp_path / "l.scp"
    lst.write_text(f"{GOLDEN}/SA000CB1.CS0 uttA\n{GOLDEN}/SA000CB1.CS3 uttB\n")
    args = [a for a in C1 if a not in ("-format_out", "htk")]
    ark = tmp_path / "t.ark"
    r = run(args + ["-format_out", f"ark={ark}", "-S", str(lst)])
    assert r.returncode == 0, r.stderr
    scp = (tmp_path / "t.scp").read_text().split("\n")
    assert scp[0] == f"uttA {ark}:5" and scp[1] == f"uttB {ark}:30913"
    b = ark.read_bytes()
    assert b[:11] == b"uttA \0BFM \4" and struct.unpack("<i", b[11:15])[0] == 594 and b[15] == 4
    assert struct.unpack("<i", b[16:20])[0] == 13
    pfile = tmp_path / "t.pfile"
    r = run(args + ["-format_out", f"pfile={pfile}", "-S", str(lst)])
    assert r.returncode == 0, r.stderr
    pb = pfile.read_bytes()
    assert len(pb) == 103940
    hdr = pb[:32768].split(b"\0")[0].decode()
    assert "-num_sentences 2" in hdr and "-num_frames 1186" in hdr and "-num_features 13" in hdr
    row0 = struct.unpack(">II13f", pb[32768:32768 + 60])
    assert row0[:2] == (0, 0)
    assert _close(np.array(row0[2:], dtype=np.float32), Oracle(C1).process(sig("CS0"))[0])
    assert struct.unpack(">3I", pb[-12:]) == (0, 594, 1186)


@pytest.mark.gpu
def test_wave_and_alaw_inputs(tmp_path):
    x = sig("CS3")[:32000]
    wav = tmp_path / "a.wav"
    hdr = b"RIFF" + struct.pack("<I", 36 + 2 * x.size) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 16000, 32000, 2, 16) \
        + b"data" + struct.pack("<I", 2 * x.size)
    wav.write_bytes(hdr + x.astype("<i2").tobytes())
    args = [a for a in C1]
    args[args.index("raw")] = "wave"
    out = tmp_path / "w.htk"
    r = run(args + ["-i", str(wav), "-o", str(out)])
    assert r.returncode == 0, r.stderr
    got = np.frombuffer(out.read_bytes()[12:], dtype="<f4").reshape(-1, 13)
    assert _close(got, Oracle(C1).process(x))
    # a-law: decode table restated from the reference formula (src/io/amulaw.h:20-53)
    codes = np.arange(256, dtype=np.uint8)
    a = codes.astype(np.int8).astype(np.int32)
    sgn = (~(a >> 7)) & 1
    chord = ((a ^ 0x55) >> 4) & 7
    step = (a ^ 0x55) & 0xF
    mag = (step << 1) + 1
    mag = np.where(chord > 0, mag + 32, mag)
    chord = np.where(chord > 0, chord, 1)
    mag = mag << chord
    outv = ((1 - 2 * sgn) * mag) & 0xFFFF
    outv = (outv << 2) & 0xFFFF
    table = np.where(outv & 0x8000, outv - 65536, outv).astype(np.int16)
    ref_so = os.path.join(os.path.dirname(GOLDEN), "..", "oracle", "_ref", "libref_amulaw.so")
    if os.path.exists(ref_so):  # the reference's own expander compiled in place (tests/test_host_decoders.py)
        import ctypes
        rt = np.zeros(256, dtype=np.int16)
        ctypes.CDLL(ref_so).ref_amulaw_table(1, rt.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(rt, table)
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, size=24000, dtype=np.uint8)
    (tmp_path / "x.al").write_bytes(raw.tobytes())
    args2 = [a for a in C1]
    args2[args2.index("raw")] = "alaw"
    out2 = tmp_path / "al.htk"
    r = run(args2 + ["-i", str(tmp_path / "x.al"), "-o", str(out2)])
    assert r.returncode == 0, r.stderr
    got = np.frombuffer(out2.read_bytes()[12:], dtype="<f4").reshape(-1, 13)
    assert _close(got, Oracle(C1).process(table[raw]))


@pytest.mark.gpu
def test_short_file_aborts_like_the_reference(tmp_path):
    (tmp_path / "s.raw").write_bytes(np.zeros(100, dtype="<i2").tobytes())
    r = run(C1 + ["-i", str(tmp_path / "s.raw"), "-o", str(tmp_path / "s.htk")])
    assert r.returncode == 255 and "Signal shorter than one frame" in r.stderr


@pytest.mark.gpu
def test_vad_file_and_four_column_list(tmp_path):
    from tests.util import C4
    lst = _list(tmp_path, ["CS3"], cols=4)
    r = run(C4 + ["-S", lst])
    assert r.returncode == 0, r.stderr
    vad = (tmp_path / "CS3.vad").read_bytes()
    assert len(vad) == 1186 and set(vad) <= {ord("0"), ord("1")}
    assert vad.count(b"1") == 626                   # the compiled reference wrote 626 ones
    assert len((tmp_path / "CS3.out").read_bytes()) == 12 + 1186 * 52
    # a two-column list is a format error when the VAD is on (src/io/batch.cc:356)
    r = run(C4 + ["-S", _list(tmp_path, ["CS3"], cols=2)])
    assert r.returncode == 255 and "Bad list format" in r.stderr


@pytest.mark.gpu
def test_delta_and_stacking_headers(tmp_path):
    # MFCC_0_D_A: 39 floats, kind 6|_0|_D|_A (src/io/out.cc:146-159)
    lst = _list(tmp_path, ["CS0", "CS3"])
    cfg = C1 + ["-fea_delta", "d_a"]
    r = run(cfg + ["-S", lst])
    assert r.returncode == 0, r.stderr
    orc = Oracle(cfg)
    for name, frames in (("CS0", 594), ("CS3", 592)):
        img = (tmp_path / (name + ".out")).read_bytes()
        assert struct.unpack("<IIHH", img[:12]) == (frames, 100000, 156, 6 | 0o20000 | 0o400 | 0o1000)
        got = np.frombuffer(img[12:], dtype="<f4").reshape(frames, 39)
        assert _close(got, orc.process(sig(name)))
    # -fea_trap: the reference's writers switch fea_kind to "spec" while saving the first frame (out.cc:182), so the
    # first file's header says MFCC (6) and every later one says MELSPEC (8); the qualifier bits stay
    cfg = C1 + ["-fea_trap", "5"]
    r = run(cfg + ["-S", lst])
    assert r.returncode == 0, r.stderr
    orc = Oracle(cfg)
    for name, frames, base in (("CS0", 594, 6), ("CS3", 592, 8)):
        img = (tmp_path / (name + ".out")).read_bytes()
        assert struct.unpack("<IIHH", img[:12]) == (frames, 100000, 65 * 4, base | 0o20000 | 0o400)
        got = np.frombuffer(img[12:], dtype="<f4").reshape(frames, 65)
        assert _close(got, orc.process(sig(name)))


@pytest.mark.gpu
def test_cmvn_statistics_file_and_three_pass_apply(tmp_path):
    from oracle.oracle import cmvn_apply, cmvn_slot_columns, cmvn_stat_text, cmvn_stats
    # statistics only: "<in> <speaker>" lines, no feature files (src/io/batch.cc:154-156,358-364)
    lst = tmp_path / "stat.scp"
    lst.write_text("".join("%s %s\n" % (os.path.join(GOLDEN, "SA000CB1." + n), s) for n, s in (("CS0", "spkA"), ("CS3", "spkB"), ("CS0", "spkB"))))
    stat = tmp_path / "cmvn.stat"
    r = run(C1 + ["-stat_cmvn", str(stat), "-S", str(lst)])
    assert r.returncode == 0, r.stderr
    orc = Oracle(C1)
    rows = [orc.process(sig(n)) for n in ("CS0", "CS3", "CS0")]
    spk = np.array([0, 1, 1])
    cols = cmvn_slot_columns(12, 1)
    mean, var, _ = cmvn_stats(rows, spk, 2, cols)
    text = stat.read_text()
    want = cmvn_stat_text(["spkA", "spkB"], mean, var)
    assert [l.split("\t")[0] for l in text.splitlines()] == [l.split("\t")[0] for l in want.splitlines()]
    got_vals = np.array([[float(v) for v in l.split("\t")[1].split()] for l in text.splitlines() if "\t" in l])
    want_vals = np.array([[float(v) for v in l.split("\t")[1].split()] for l in want.splitlines() if "\t" in l])
    assert got_vals.shape == (4, 13) and np.abs(got_vals - want_vals).max() <= 2e-6 + 1e-6 * np.abs(want_vals).max()
    assert not list(tmp_path.glob("*.out"))
    # compute + apply: "<in> <out> <speaker>", statistics land in the -apply_cmvn file (src/io/batch.cc:131-151)
    lst2 = tmp_path / "apply.scp"
    lst2.write_text("".join("%s %s %s\n" % (os.path.join(GOLDEN, "SA000CB1." + n), tmp_path / (t + ".out"), s)
                            for n, t, s in (("CS0", "a", "spkA"), ("CS3", "b", "spkB"), ("CS0", "c", "spkB"))))
    stat2 = tmp_path / "cmvn2.stat"
    r = run(C1 + ["-apply_cmvn", str(stat2), "-S", str(lst2)])
    assert r.returncode == 0, r.stderr
    assert "Stat. cmvn file is being created" in r.stdout
    assert stat2.read_text().splitlines()[0] == "spkA"
    for t, i in (("a", 0), ("b", 1), ("c", 2)):
        img = (tmp_path / (t + ".out")).read_bytes()
        n = rows[i].shape[0]
        assert struct.unpack("<IIHH", img[:12]) == (n, 100000, 52, 8198)
        got = np.frombuffer(img[12:], dtype="<f4").reshape(n, 13)
        assert _close(got, cmvn_apply(rows[i], spk[i], mean, var, cols))
    # an existing statistics file is refused (the reference's reader is broken, see main.cc)
    r = run(C1 + ["-apply_cmvn", str(stat2), "-S", str(lst2)])
    assert r.returncode != 0 and "existing CMVN statistics" in r.stderr


@pytest.mark.gpu
def test_enhancement_raw_and_wave_files(tmp_path):
    # the shipped example egs/conf/21_exten.ctuconf: -fs 16000 -format_in raw -format_out raw -preset exten
    cfg = "-fs 16000 -format_in raw -format_out raw -preset exten".split()
    lst = _list(tmp_path, ["CS0", "CS3"])
    r = run(cfg + ["-S", lst, "-v"])
    assert r.returncode == 0, r.stderr
    orc = Oracle(cfg)
    for name in ("CS0", "CS3"):
        ref = orc.enhance(sig(name))
        got = np.frombuffer((tmp_path / (name + ".out")).read_bytes(), dtype="<i2")
        d = np.abs(got.astype(int) - ref.astype(int))
        assert got.shape == ref.shape and d.max() <= 2 and d.mean() < 0.3
    # WAVE container and big-endian raw
    r = run([a if a != "raw" or i < 4 else "wave" for i, a in enumerate(cfg)] + ["-S", lst])
    assert r.returncode == 0, r.stderr
    img = (tmp_path / "CS3.out").read_bytes()
    n = len(orc.enhance(sig("CS3")))
    assert img[:4] == b"RIFF" and img[8:16] == b"WAVEfmt " and img[36:40] == b"data"
    assert struct.unpack("<I", img[4:8])[0] == 2 * n + 36 and struct.unpack("<I", img[40:44])[0] == 2 * n
    assert struct.unpack("<IHHIIHH", img[16:36]) == (16, 1, 1, 16000, 32000, 2, 16)
    wav = np.frombuffer(img[44:], dtype="<i2")
    r = run(cfg + ["-endian_out", "big", "-S", lst])
    assert r.returncode == 0, r.stderr
    be = np.frombuffer((tmp_path / "CS3.out").read_bytes(), dtype=">i2")
    assert np.array_equal(wav, be.astype(np.int16))


def _device_count():
    import torch
    return torch.cuda.device_count()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["mfcc", "vad", "cmvn", "enhance"])
def test_multi_engine_host_path_writes_what_one_engine_writes(tmp_path, kind):
    """bin/ctucopy --gpus N (host/main.cc: one thread and one engine per GPU, LPT shards, files written in list order).
    With two devices the engines sit on devices 0 and 1; on a one-GPU box --gpu-map 0,0 puts both on device 0, which runs
    the same threaded path (per-engine tables, plans, streams, attribute handling)."""
    from tests.util import C2, C4, synth_utt
    fs = 8000 if kind == "vad" else 16000
    lens = [30000, 8000, 52000, 16000, 9000, 41000, 12345]
    names = []
    for i, n in enumerate(lens):
        f = tmp_path / f"u{i}.raw"
        synth_utt(300 + i, n, fs=fs).astype("<i2").tofile(f)
        names.append(f)
    cfg = {"mfcc": C2 + ["-fea_delta", "d_a"], "vad": C4, "cmvn": C2 + ["-apply_cmvn", str(tmp_path / "STAT")],
           "enhance": "-fs 16000 -format_in raw -format_out raw -preset exten".split()}[kind]
    outs = {}
    for tag, extra in (("one", []), ("two", ["--gpus", "2"] + ([] if _device_count() >= 2 else ["--gpu-map", "0,0"]))):
        d = tmp_path / tag
        d.mkdir()
        lines = []
        for i, f in enumerate(names):
            cols = [str(f), str(d / f"u{i}.out")]
            if kind == "vad":
                cols += ["spk", str(d / f"u{i}.vad")]
            if kind == "cmvn":
                cols += [f"spk{i % 3}"]
            lines.append(" ".join(cols))
        (d / "list").write_text("\n".join(lines) + "\n")
        stat = tmp_path / "STAT"
        if stat.exists():
            stat.unlink()
        r = run(cfg + ["-S", str(d / "list")] + extra)
        assert r.returncode == 0, r.stderr
        outs[tag] = {p.name: p.read_bytes() for p in sorted(d.iterdir()) if p.name != "list"}
        if kind == "cmvn":
            outs[tag]["STAT"] = stat.read_bytes()
    assert outs["one"].keys() == outs["two"].keys() and len(outs["one"]) >= len(lens)
    for k in outs["one"]:
        if kind == "cmvn":
            # statistics are summed per engine and then across engines: the order of the double additions differs
            if k == "STAT":
                ta, tb = outs["one"][k].decode().split(), outs["two"][k].decode().split()
                assert len(ta) == len(tb)
                for x, y in zip(ta, tb):
                    try:
                        assert abs(float(x) - float(y)) <= 1e-9 * max(1.0, abs(float(x)))
                    except ValueError:
                        assert x == y
            else:
                a = np.frombuffer(outs["one"][k][12:], dtype="<f4")
                b = np.frombuffer(outs["two"][k][12:], dtype="<f4")
                assert outs["one"][k][:12] == outs["two"][k][:12] and np.allclose(a, b, rtol=0, atol=1e-5)
        else:
            assert outs["one"][k] == outs["two"][k], k


@pytest.mark.gpu
def test_bench_two_ranks_smoke():
    """bench.py under torch.distributed.run with two ranks (RCCL): one list LPT-sharded, one JSON line from rank 0."""
    import json
    import sys
    if _device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(29000 + os.getpid() % 2000), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "1", "--utts", "300"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and len(j["config"]["frames_per_rank"]) == 2 and j["scaling"] == "weak"
    assert j["validated"]["rows_finite"] and j["validated"]["worst_rel_err"] <= 1e-4
    assert abs(j["config"]["frames_per_rank"][0] - j["config"]["frames_per_rank"][1]) <= 1500


@pytest.mark.gpu
def test_spectral_subtraction_list_is_one_chain(tmp_path):
    """-nr_mode fwss -vad burg through the CLI at 16 kHz: the files of the list are one chain through the noise seed
    (src/nr/nr.cc:212-221), a file without a frame scales the stale vector (nr.cc:219), --gpus N falls back to one GPU."""
    from tests.util import C2, synth_utt
    cfg = C2 + ["-vad", "burg", "-nr_mode", "fwss"]
    lens = [30000, 12000, 240, 20000, 9000]            # 240 samples: the pre-load and no hop
    lines, utts = [], []
    for i, n in enumerate(lens):
        f = tmp_path / f"u{i}.raw"
        u = synth_utt(500 + i, n)
        u.astype("<i2").tofile(f)
        utts.append(u)
        lines.append(f"{f} {tmp_path / f'u{i}.htk'}")
    (tmp_path / "list").write_text("\n".join(lines) + "\n")
    r = run(cfg + ["-S", str(tmp_path / "list"), "--gpus", "2", "-v"])
    assert r.returncode == 0, r.stderr
    assert "chains the files of the list" in r.stderr
    orc = Oracle(cfg)
    for i, u in enumerate(utts):
        ref = orc.process(u)
        raw = (tmp_path / f"u{i}.htk").read_bytes()
        n, period, nbytes, kind = struct.unpack("<IIHH", raw[:12])
        assert (n, nbytes) == (ref.shape[0], 52)
        got = np.frombuffer(raw[12:], dtype="<f4").reshape(-1, 13)
        if ref.size:
            e = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
            rn = np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1.0)
            assert e.max() <= 1e-3 and rn.max() <= 1e-4, (i, float(e.max()), float(rn.max()))   # the conditioning class of an NR configuration


def _many_files(tmp_path, n, seed=7):
    """n short raw files (1 .. 4 s of 16 kHz noise: a dozen to the MiB) and a list that sends their rows to <tmp>/o/"""
    rng = np.random.default_rng(seed)
    (tmp_path / "i").mkdir()
    (tmp_path / "o").mkdir()
    lines = []
    for k in range(n):
        x = (rng.standard_normal(int(rng.integers(16000, 64000))) * 2000).astype("<i2")
        x.tofile(tmp_path / "i" / ("f%03d.raw" % k))
        lines.append("%s %s" % (tmp_path / "i" / ("f%03d.raw" % k), tmp_path / "o" / ("f%03d.htk" % k)))
    (tmp_path / "list.scp").write_text("\n".join(lines) + "\n")
    return str(tmp_path / "list.scp")


@pytest.mark.gpu
def test_pipelined_host_loop_is_independent_of_batching_and_threads(tmp_path):
    """The three-stage host loop (reader / engines / writer, main.cc): whatever the batch size and the thread counts, the
    files come out byte for byte the same - many small batches exercise every hand-over and the page-locked buffer pool."""
    lst = _many_files(tmp_path, 60)
    base = "-fs 16000 -format_in raw -format_out htk -preset mfcc -S".split() + [lst]
    r = run(base + ["--io-threads", "1", "--write-threads", "1"])
    assert r.returncode == 0, r.stderr
    want = {f: (tmp_path / "o" / f).read_bytes() for f in sorted(os.listdir(tmp_path / "o"))}
    assert len(want) == 60
    for extra in (["--batch-mib", "1"], ["--batch-mib", "1", "--io-threads", "8", "--write-threads", "3"], ["--gpus", "2", "--gpu-map", "0,0", "--batch-mib", "1"]):
        for f in want:
            os.unlink(tmp_path / "o" / f)
        r = run(base + extra)
        assert r.returncode == 0, r.stderr
        got = {f: (tmp_path / "o" / f).read_bytes() for f in sorted(os.listdir(tmp_path / "o"))}
        assert got.keys() == want.keys() and all(got[f] == want[f] for f in want), extra
    # ark / scp through the ordered writer: entries in list order whatever the batching
    for extra in ([], ["--batch-mib", "1", "--io-threads", "8"]):
        r = run("-fs 16000 -format_in raw -preset mfcc -format_out".split() + ["ark=" + str(tmp_path / ("a%d.ark" % len(extra))), "-S", lst] + extra)
        assert r.returncode == 0, r.stderr
    a0, a1 = (tmp_path / "a0.ark").read_bytes(), (tmp_path / "a4.ark").read_bytes()
    assert a0 == a1 and len(a0) > 60 * 13 * 4 * 28
    assert (tmp_path / "a0.scp").read_text().replace("a0.ark", "a4.ark") == (tmp_path / "a4.scp").read_text()
    keys = [l.split()[0] for l in (tmp_path / "a0.scp").read_text().splitlines()]
    assert keys == [str(tmp_path / "o" / ("f%03d.htk" % k)) for k in range(60)]


@pytest.mark.gpu
def test_a_bad_file_stops_the_list_after_the_batches_in_front_of_it(tmp_path):
    """The reference aborts the batch at the first bad file with its message and exit status -1, having written the files in
    front of it (src/main.cpp:54-60, src/io/batch.cc:326-421).  The pipelined loop writes every batch that lies wholly in
    front of the bad file and nothing behind it."""
    lst = _many_files(tmp_path, 40)
    os.unlink(tmp_path / "i" / "f025.raw")
    r = run("-fs 16000 -format_in raw -format_out htk -preset mfcc -S".split() + [lst, "--batch-mib", "1", "--io-threads", "4"])
    assert r.returncode == 255 and "Cannot open data file" in r.stderr
    done = sorted(os.listdir(tmp_path / "o"))
    assert done == ["f%03d.htk" % k for k in range(len(done))] and 0 < len(done) <= 25
    # a file too short for one frame: the reference's text (src/io/in.cc:277)
    np.zeros(100, dtype="<i2").tofile(tmp_path / "i" / "f025.raw")
    r = run("-fs 16000 -format_in raw -format_out htk -preset mfcc -S".split() + [lst, "--batch-mib", "1"])
    assert r.returncode == 255 and "Signal shorter than one frame" in r.stderr


@pytest.mark.gpu
def test_vad_decisions_from_a_file_through_the_cli(tmp_path):
    """-nr_mode fwss -vad file=<f> (src/nr/nr.cc:205-209, 297-302): one byte per frame out of one stream for the whole list,
    spread over several small batches by --batch-mib; a stream one byte short ends the run with the reference's text."""
    from tests.util import C2, synth_utt
    lens = [30000, 12000, 240, 20000, 9000, 16000, 31000]
    lines, utts = [], []
    for i, n in enumerate(lens):
        f = tmp_path / f"u{i}.raw"
        u = synth_utt(600 + i, n)
        u.astype("<i2").tofile(f)
        utts.append(u)
        lines.append(f"{f} {tmp_path / f'u{i}.htk'}")
    (tmp_path / "list").write_text("\n".join(lines) + "\n")
    frames = [max((n - 240) // 160, 0) for n in lens]
    stream = np.random.default_rng(4).integers(0, 2, sum(frames)).astype(np.uint8)
    (tmp_path / "vad.bin").write_bytes(bytes(stream))
    cfg = C2 + ["-nr_mode", "fwss", "-vad", f"file={tmp_path / 'vad.bin'}"]
    r = run(cfg + ["-S", str(tmp_path / "list"), "--batch-mib", "1"])
    assert r.returncode == 0, r.stderr
    orc = Oracle(cfg)
    for i, u in enumerate(utts):
        ref = orc.process(u)
        raw = (tmp_path / f"u{i}.htk").read_bytes()
        n = struct.unpack("<I", raw[:4])[0]
        assert n == ref.shape[0] == frames[i]
        if n:
            got = np.frombuffer(raw[12:], dtype="<f4").reshape(n, -1)
            assert np.all(np.abs(got - ref) <= 1e-3 * np.maximum(np.abs(ref), 1.0)) and np.abs(got - ref).max() <= 1e-4 * np.abs(ref).max() * 10
    (tmp_path / "vad.bin").write_bytes(bytes(stream[:-1]))
    r = run(cfg + ["-S", str(tmp_path / "list")])
    assert r.returncode == 255 and "Unexpected end of VAD file" in r.stderr
    r = run(C2 + ["-nr_mode", "fwss", "-vad", f"file={tmp_path / 'nope'}", "-S", str(tmp_path / "list")])
    assert r.returncode == 255 and "Unable to open VAD file" in r.stderr


@pytest.mark.gpu
def test_one_frame_file_with_the_vad_writes_an_empty_htk_file_and_no_decision(tmp_path):
    """A file with no more frames than the VAD's majority filter delays (default order 3: one frame) never gets the filter ready: the
    reference writes a header with 0 frames and an empty VAD file for it (src/vad/vad.h:126-136, src/io/batch.cc:230-249)."""
    from tests.util import C4, synth_utt
    lines = []
    for i, T in enumerate((1, 30)):
        synth_utt(70 + i, 120 + 80 * T, fs=8000).astype("<i2").tofile(tmp_path / f"u{i}.raw")
        lines.append(f"{tmp_path / f'u{i}.raw'} {tmp_path / f'u{i}.htk'} spk {tmp_path / f'u{i}.vad'}")
    (tmp_path / "list").write_text("\n".join(lines) + "\n")
    r = run(C4 + ["-S", str(tmp_path / "list")])
    assert r.returncode == 0, r.stderr
    assert struct.unpack("<I", (tmp_path / "u0.htk").read_bytes()[:4])[0] == 0 and (tmp_path / "u0.vad").read_bytes() == b""
    assert struct.unpack("<I", (tmp_path / "u1.htk").read_bytes()[:4])[0] == 30 and len((tmp_path / "u1.vad").read_bytes()) == 30


@pytest.mark.gpu
def test_vad_list_through_the_cli_is_the_reference_list(tmp_path):
    """bin/ctucopy walks its list as one process like the reference: the VAD's majority filter keeps its ring index from file to file
    (src/vad/vad.h:110-121), also across the batches of the host loop and over two engines."""
    from tests.util import C4, synth_utt
    frames = [17, 9, 31, 8, 1, 25, 14, 40, 22, 13]
    utts, lines = [], []
    for i, T in enumerate(frames):
        u = synth_utt(500 + i, 120 + 80 * T + 3 * i, fs=8000)
        u.astype("<i2").tofile(tmp_path / f"u{i}.raw")
        utts.append(u)
        lines.append(f"{tmp_path / f'u{i}.raw'} {tmp_path / f'u{i}.htk'} spk {tmp_path / f'u{i}.vad'}")
    (tmp_path / "list").write_text("\n".join(lines) + "\n")
    ref = Oracle(C4).process_list(utts, want_vad=True)
    for extra in ([], ["--gpus", "2", "--gpu-map", "0,0"]):
        r = run(C4 + ["-S", str(tmp_path / "list")] + extra)
        assert r.returncode == 0, r.stderr
        for i, (rows, rv) in enumerate(ref):
            raw = (tmp_path / f"u{i}.htk").read_bytes()
            n = struct.unpack("<I", raw[:4])[0]
            assert n == rows.shape[0] and (tmp_path / f"u{i}.vad").read_bytes() == bytes(rv), (extra, i)
            if n:
                got = np.frombuffer(raw[12:], dtype="<f4").reshape(n, -1)
                assert np.array_equal(~got.any(axis=1), ~rows.any(axis=1))
                assert np.all(np.abs(got - rows) <= 1e-3 * np.maximum(np.abs(rows), 1.0)), (extra, i)


@pytest.mark.gpu
@pytest.mark.parametrize("apply_mode", ["none", "drop"])
def test_cmvn_together_with_the_vad(tmp_path, apply_mode):
    """-apply_cmvn with the VAD on (src/io/batch.cc:193-204,230-241): the passes that sum the statistics never call save_frame, so the
    statistics are over every frame; the last pass normalises a vector and then hands it to save_frame - the VAD's ring (whose index runs on
    along the list), its decision, the drop.  Expected: the oracle's rows and decisions of every file on its own, numpy CMVN over all of
    them, then the ring mapping (ctu_vad_ring_rows, itself checked against the oracle's list mode and the reference's class) and the drop."""
    import ctypes
    from ctucopy_amd import engine as ceng
    from ctucopy_amd import synth
    from oracle.oracle import cmvn_apply, cmvn_slot_columns, cmvn_stats
    from tests.util import C4
    cfg = C4 + ["-vad_apply_mode", apply_mode]
    utts = [synth.utterance_c(synth.SET_NOISY, k, True) for k in range(6)] + [np.zeros(120 + 80, np.int16) + 3]   # the last: one frame
    spk = [0, 1, 0, 1, 1, 0, 1]
    lines = []
    for i, u in enumerate(utts):
        u.astype("<i2").tofile(tmp_path / f"u{i}.raw")
        lines.append(f"{tmp_path / f'u{i}.raw'} {tmp_path / f'u{i}.htk'} spk{spk[i]} {tmp_path / f'u{i}.vad'}")
    (tmp_path / "list").write_text("\n".join(lines) + "\n")
    r = run(cfg + ["-apply_cmvn", str(tmp_path / "stat"), "-S", str(tmp_path / "list")])
    assert r.returncode == 0, r.stderr
    # expected
    plain = C4 + ["-vad_apply_mode", "none", "-vad_filter_order", "1"]      # every frame's row, undelayed; decisions do not depend on the filter...
    orc1, orc = Oracle(plain), Oracle(C4 + ["-vad_apply_mode", "none"])     # ... and the filtered decisions of each file on its own
    rows = [orc1.process(u) for u in utts]
    dec = [np.asarray(orc.process(u, want_vad=True)[1]) for u in utts]
    cols = cmvn_slot_columns(12, 1)
    mean, var, _ = cmvn_stats(rows, np.array(spk), 2, cols)
    L = ceng.load_library()
    hi, hs = ctypes.c_int32(0), ctypes.c_int32(0)
    for i, (u, rw) in enumerate(zip(utts, rows)):
        T = rw.shape[0]
        norm = cmvn_apply(rw, spk[i], mean, var, cols)
        src = np.full(max(T, 1), -1, dtype=np.int32)
        n_out = L.ctu_vad_ring_rows(3, T, hi.value, src.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        L.ctu_vad_ring_step(3, T, ctypes.byref(hi), ctypes.byref(hs))
        assert n_out == len(dec[i])
        want = np.stack([norm[src[q]] if src[q] >= 0 else np.zeros(13, np.float32) for q in range(n_out)]) if n_out else np.zeros((0, 13), np.float32)
        if apply_mode == "drop" and n_out:
            want = want[dec[i] == ord("1")]
        raw = (tmp_path / f"u{i}.htk").read_bytes()
        n = struct.unpack("<I", raw[:4])[0]
        assert n == want.shape[0], (i, n, want.shape)
        assert (tmp_path / f"u{i}.vad").read_bytes() == bytes(dec[i])
        if n:
            got = np.frombuffer(raw[12:], dtype="<f4").reshape(n, 13)
            assert np.array_equal(~got.any(axis=1), ~want.any(axis=1))
            assert np.all(np.abs(got - want) <= 2e-3 * np.maximum(np.abs(want), 1.0)), (i, float(np.abs(got - want).max()))
