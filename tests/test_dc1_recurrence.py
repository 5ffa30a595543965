"""-remove_dc1 (src/io/in.cc:343-350): the closed form the device uses (ctucopy_amd/csrc/decode_kernels.h) against a
literal simulation of the reference's circular buffer."""
import numpy as np
import pytest


def simulate(x, W, S):
    """The reference's loop on a linear copy of the signal: every frame subtracts its buffer mean from the buffer."""
    x = x.astype(np.float64).copy()
    T = (len(x) - (W - S)) // S
    offs, frames, pre = [], [], []
    for t in range(T):
        s = t * S
        o = x[s:s + W].mean()
        x[s:s + W] -= o
        offs.append(o)
        frames.append(x[s:s + W].copy())
        pre.append(x[s + S - 1])   # preemtmp for the next frame
    return np.array(offs), np.array(frames), np.array(pre)


@pytest.mark.parametrize("W,S", [(400, 160), (200, 80), (400, 80), (512, 256), (256, 128), (300, 40)])
def test_offsets_follow_the_recurrence_and_samples_their_accumulated_offsets(W, S):
    rng = np.random.default_rng(W * 1000 + S)
    x = (rng.integers(-3000, 3000, size=W + S * 40) + 500).astype(np.int16)
    offs, frames, pre = simulate(x, W, S)
    T = len(offs)
    xr = x.astype(np.float64)
    # o_t = m_t - sum_{j>=1, j*S < W} (W - j*S)/W * o_{t-j}
    o = np.zeros(T)
    for t in range(T):
        m = xr[t * S:t * S + W].mean()
        acc = m
        j = 1
        while j * S < W:
            if t - j >= 0:
                acc -= (W - j * S) / W * o[t - j]
            j += 1
        o[t] = acc
    assert np.allclose(o, offs, rtol=0, atol=1e-9)
    # sample at position i of frame t: x - o_t - sum_{j>=1, i <= W-1-j*S} o_{t-j}
    J = W // S
    for t in (0, 1, 2, T // 2, T - 1):
        i = np.arange(W)
        cum = np.full(W, o[t])
        for j in range(1, J + 1):
            if t - j >= 0:
                cum += np.where(i <= W - 1 - j * S, o[t - j], 0.0)
        assert np.allclose(xr[t * S:t * S + W] - cum, frames[t], rtol=0, atol=1e-9)
        # the sample ahead of the frame (pre-emphasis history): position -1, without the frame's own offset
        if t >= 1:
            c = 0.0
            for j in range(1, J + 1):
                if t - j >= 0 and -1 <= W - 1 - j * S:
                    c += o[t - j]
            assert abs((xr[t * S - 1] - c) - pre[t - 1]) < 1e-9
