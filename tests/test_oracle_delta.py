"""Row N1 (delta / acceleration / third differences, -fea_trap stacking): the oracle replays the reference's
streaming ring (src/fea/fea_delta.cc) literally; these tests pin what that ring computes against closed forms, the
same closed forms the GPU pass implements (ctucopy_amd/csrc/post_kernels.h: post_kernel), and check geometry and header
bits against src/io/out.cc:95-113,157-159 and the shipped example configs egs/conf/15_*, 17_*."""
import numpy as np
import pytest

from oracle.oracle import Oracle, OracleError
from ctucopy_amd import config_dims
from tests.util import C2, sig

PCM = sig("CS0")[: 16000 * 2]


def delta(x, w, quirk=True):
    T = x.shape[0]
    idx = np.arange(T)
    out = np.zeros_like(x)
    for i in range(1, w + 1):
        out += i * (x[np.minimum(idx + i, T - 1)] - x[np.maximum(idx - i, 0)])
    out /= 2 * sum(i * i for i in range(1, w + 1))
    if quirk and w == 1:
        out[T - 1] = 0  # the flush of a w=1 stage writes the last frame twice (num_c stays 2, fea_delta.cc:81-83,132-144)
    return out


def win_args(spec, ws):
    a = ["-fea_delta", spec]
    for flag, w in zip(("-d_win", "-a_win", "-t_win"), ws):
        a += [flag, str(w)]
    return a


@pytest.mark.parametrize("spec,ws", [("d", (2,)), ("d_a", (2, 2)), ("d_a_t", (2, 2, 2)), ("d_a", (1, 3)), ("d_a_t", (3, 1, 2)),
                                     ("d", (1,)), ("d_a_t", (1, 1, 1)), ("d_a", (4, 1)), ("d_a_t", (8, 8, 8))])
def test_ring_equals_clamped_regression(spec, ws):
    base = Oracle(C2).process(PCM).astype(np.float64)
    r = Oracle(C2 + win_args(spec, ws)).process(PCM).astype(np.float64)
    blocks = [base]
    for w in ws:
        blocks.append(delta(blocks[-1], w))
    exp = np.concatenate(blocks, 1)
    assert r.shape == exp.shape
    assert np.array_equal(r[:, :13], base)           # block 0 is the undelayed base row, same slot order (c1..c12, c0)
    assert np.abs(r - exp).max() < 2e-5               # base is float32-rounded here, the ring runs on doubles
    if 1 in ws:                                       # ... and the w == 1 rule is needed to get there
        blocks = [base]
        for w in ws:
            blocks.append(delta(blocks[-1], w, quirk=False))
        assert np.abs(r - np.concatenate(blocks, 1)).max() > 1e-3


def test_energy_column_lags_by_the_chain_delay():
    cfg = C2 + ["-fea_E", "on"]
    base = Oracle(cfg).process(PCM)
    T = base.shape[0]
    r = Oracle(cfg + win_args("d_a", (2, 3))).process(PCM)
    assert r.shape == (T, 40)
    assert np.array_equal(r[:, 39], base[np.minimum(np.arange(T) + 5, T - 1), 13])


@pytest.mark.parametrize("tw", [3, 5, 7, 9])
def test_stacking_layout_and_edge_rows(tw):
    cfg = C2 + ["-fea_E", "on"]
    base = Oracle(cfg).process(PCM)
    T = base.shape[0]
    fv = np.concatenate([base[:, 12:13], base[:, :12]], 1)  # fvec order: c0, c1..c12
    w = (tw - 1) // 2
    L = 2 * w + 1
    o = Oracle(cfg + ["-fea_trap", str(tw)])
    r = o.process(PCM)
    exp = np.zeros((T, 13 * L + 1), np.float32)
    for t in range(T):
        for j in range(L):
            f = (0 if j < w else max(1, j - w)) if t == 0 else min(max(t - w + j, 0), T - 1)
            if w == 1 and t == T - 1:
                f = T - 1
            exp[t, j:13 * L:L] = fv[f]
        if t == 0 or t >= T - w:
            exp[t, :13] = fv[t]
        exp[t, 13 * L] = base[min(t + w, T - 1), 13]
    assert np.array_equal(r, exp)
    assert o.dims.htk_kind == (6 | 0o20000 | 0o100 | 0o400)  # -fea_trap also sets fea_delta / n_order=1 (opts.cc:694-703)


def test_geometry_and_kind_bits():
    # egs/conf/15_*: MFCC_0_D_A, 39 floats
    c15 = ("-fs 16000 -format_in raw -format_out htk -w 25 -s 10 -preem 0.97 -fb_scale mel -fb_shape triang -fb_power on "
           "-fb_definition 30filters -nr_mode none -fb_eqld off -fb_inld off -fea_kind dctc -fea_ncepcoefs 12 -fea_c0 on -fea_E off "
           "-fea_lifter 22 -fea_rawenergy off -fea_delta d_a -d_win 2 -a_win 2 -t_win 2").split()
    def oracle_dims(c):
        d = Oracle(c).dims
        return d.D, d.htk_kind, d.B

    def engine_dims(c):
        d = config_dims(c)
        return d.row_floats, d.htk_kind, d.nbands

    for mk in (oracle_dims, engine_dims):
        assert mk(c15) == (39, 6 | 0o20000 | 0o400 | 0o1000, 30)
        assert mk(C2 + ["-fea_delta", "d_a_t", "-fea_E", "on"])[:2] == (53, 6 | 0o20000 | 0o100 | 0o400 | 0o1000 | 0o100000)
        assert mk(C2 + ["-fea_trap", "5"])[:2] == (65, 6 | 0o20000 | 0o400)
        k = mk("-fs 8000 -preset plpc -fea_delta d".split())[1]
        assert k & 0o77 == 11 and k & 0o400
    # option order as in src/io/opts.cc:686-703: -fea_trap after -fea_delta is ignored; -fea_delta after -fea_trap
    # switches stacking off but keeps the d_win the trap option derived
    assert config_dims(C2 + ["-fea_delta", "d", "-fea_trap", "9"]).row_floats == 26
    a = Oracle(C2 + ["-fea_trap", "9", "-fea_delta", "d"]).process(PCM).astype(np.float64)
    b = Oracle(C2 + ["-fea_delta", "d", "-d_win", "4"]).process(PCM).astype(np.float64)
    assert np.array_equal(a, b)


def test_refusals():
    with pytest.raises(OracleError, match="Delta window size"):
        Oracle(C2 + ["-fea_delta", "d", "-d_win", "0"])
    with pytest.raises(OracleError, match="non-cepstral"):
        Oracle(C2 + ["-fea_kind", "logspec", "-fea_delta", "d"])
    with pytest.raises(OracleError, match="fea_c0"):
        Oracle(C2 + ["-fea_c0", "off", "-fea_delta", "d"])
    o = Oracle(C2 + ["-fea_delta", "d", "-d_win", "4"])
    with pytest.raises(OracleError, match="fewer than window"):
        o.process(PCM[: 240 + 160 * 4])  # 4 frames < w+1
    assert o.process(PCM[: 240 + 160 * 5]).shape == (5, 26)


def test_window_plus_one_frames_is_not_the_closed_form():
    # With exactly w+1 frames a stage never runs its steady-state branch, `end` stays 0 and the flush starts from ring
    # slot 0 (fea_delta.cc:118,183-186): rows mix frames.  The oracle replays that literally; the engine refuses such
    # inputs (CTU_ERR_INPUT) instead of imitating it - this test documents why.
    pcm = PCM[: 240 + 160 * 3]
    base = Oracle(C2).process(pcm)
    r = Oracle(C2 + ["-fea_delta", "d"]).process(pcm)
    assert r.shape == (3, 26) and np.array_equal(r[0, :13], base[0])
    assert not np.array_equal(r[1, :13], base[1]) and np.array_equal(r[1, :13], base[2])
