#!/usr/bin/env python3
"""Error probe (numpy, no GPU) for the matrix-pipe transform of the headline kernel, ahead of writing it:
the 512-point real FFT of a frame as two 256-point real FFTs (even / odd samples), each in two radix-16 stages that are
matrix products on v_mfma_f32_16x16x32_f16 with fp32 accumulation:
  stage 1: int16 PCM split EXACTLY into (signed high byte, unsigned low byte) as fp16; window, pre-emphasis and the 16-point
           DFT over n1 folded into per-n2 matrices split into two fp16 terms (4 products),
  stage 2: stage-1 outputs split into two fp16 terms; twiddles, 16-point DFT over n2 and W512^k folded into per-k1 matrices
           in two fp16 terms (3 products, or 4 with --four),
  DC removal as a rank-one correction of the spectrum, |.|^2, mel bank, log, DCT in fp32.
Prints the error of the resulting MFCC rows against the float64 oracle on the committed S-MFCC miniatures."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
from tests.util import C2, GOLDEN, sig

ap = argparse.ArgumentParser()
ap.add_argument("--four", action="store_true", help="4 products in stage 2")
ap.add_argument("--rtn", action="store_true", help="round-to-nearest high term in the stage-2 data split (default: toward zero)")
ap.add_argument("--s1", type=float, default=1 / 16., help="scale of the stage-1 outputs (high plane: in the matrix; low plane: in the data)")
ap.add_argument("--chunk", type=int, default=4)
ap.add_argument("--terms", type=int, default=3, help="fp16 terms of the stage-1 matrices")
ap.add_argument("--blockk", action="store_true", help="K order [xe(16) | xp(16)] instead of interleaved pairs")
ap.add_argument("--desc", action="store_true", help="largest products first (default: smallest first)")
ap.add_argument("--x1", action="store_true", help="stage 1 exact (float64)")
ap.add_argument("--x2", action="store_true", help="stage 2 exact (float64)")
ap.add_argument("--f64", action="store_true", help="control: the same factorisation in float64 (no splits)")
a = ap.parse_args()

orc = Oracle(C2)
W, SH, NF = 400, 160, 512
win = orc.hamming()
pre = orc.preem()
fb, first, last = orc.fbank()
B = fb.shape[0]

def f16split(M, rtz=False):
    M = np.asarray(M, dtype=np.float64)
    if rtz:
        h = M.astype(np.float16)
        # toward zero: step back where rounding went away from zero
        away = np.abs(h.astype(np.float64)) > np.abs(M)
        h = np.where(away, np.nextafter(h, np.float16(0)), h)
    else:
        h = M.astype(np.float16)
    l = (M - h.astype(np.float64)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)

def fsplit(M, n):
    M = np.asarray(M, dtype=np.float64)
    out = []
    for _ in range(n):
        t = M.astype(np.float16).astype(np.float64)
        out.append(t)
        M = M - t
    return out


def mm32(A, X, acc=None):
    """What one MFMA is assumed to do: exact products, exact sums over `--chunk` consecutive k, each chunk sum added to the
    fp32 accumulator with one rounding.  acc: the accumulator coming in (chained MFMAs)."""
    A = A.astype(np.float64); X = X.astype(np.float64)
    K = A.shape[1]
    out = np.zeros((A.shape[0], X.shape[1]), dtype=np.float32) if acc is None else acc.astype(np.float32)
    for k0 in range(0, K, a.chunk):
        out = (out.astype(np.float64) + A[:, k0:k0 + a.chunk] @ X[k0:k0 + a.chunk]).astype(np.float32)
    return out

# ---- stage-1 matrices, per n2: rows [E0, E8, Re E1, Im E1, ..., Im E7], K = [xe(n1) n1=0..15 | xp(n1) n1=0..15] (E form), [xo | xe] (O form)
def rows16(col):  # col[n1] real weights -> 16 x 16 real matrix of the DFT over n1 (outputs k1 = 0, 8, 1..7 re/im)
    n1 = np.arange(16)
    M = np.zeros((16, 16))
    M[0] = col
    M[1] = col * np.cos(np.pi * n1)
    for k1 in range(1, 8):
        M[2 * k1] = col * np.cos(2 * np.pi * n1 * k1 / 16)
        M[2 * k1 + 1] = -col * np.sin(2 * np.pi * n1 * k1 / 16)
    return M
wz = np.concatenate([win, np.zeros(NF + 2 - W)])
AE, AO = [], []
for n2 in range(16):
    i = 32 * np.arange(16) + 2 * n2
    me, mo = rows16(wz[i]), rows16(wz[i + 1])
    AE.append(np.hstack([me, -pre * me]) * a.s1)
    AO.append(np.hstack([mo, -pre * mo]) * a.s1)

# ---- stage-2 matrices, per k1: E: rows (k2 re, im) x cols (n2 re, im)
def stage2(k1, with_w512):
    n2 = np.arange(16)
    if k1 in (0, 8):
        k2s = np.arange(9) if k1 == 0 else np.arange(8)
        rows = []
        for k2 in k2s:
            k = k1 + 16 * k2
            c = np.exp(-2j * np.pi * (n2 * k1 / 256 + n2 * k2 / 16)) * (np.exp(-2j * np.pi * k / 512) if with_w512 else 1)
            rows.append((k, c))
        return rows, True
    rows = []
    for k2 in range(16):
        k = k1 + 16 * k2
        c = np.exp(-2j * np.pi * (n2 * k1 / 256 + n2 * k2 / 16)) * (np.exp(-2j * np.pi * k / 512) if with_w512 else 1)
        rows.append((k, c))
    return rows, False

def apply_stage2(S, k1, with_w512):
    """S: stage-1 rows of this k1 for all n2 and frames: real [16, F] (k1 = 0, 8) or complex parts [(re, im)] -> dict bin -> complex [F]"""
    rows, real_in = stage2(k1, with_w512)
    if real_in:
        X = S  # [16 n2, F]
        M = np.zeros((2 * len(rows), 16))
        for r, (k, c) in enumerate(rows):
            M[2 * r], M[2 * r + 1] = c.real, c.imag
    else:
        X = np.empty((32, S[0].shape[1]), dtype=np.float64 if a.f64 else np.float32)
        X[0::2], X[1::2] = S[0], S[1]
        M = np.zeros((2 * len(rows), 32))
        for r, (k, c) in enumerate(rows):
            M[2 * r, 0::2], M[2 * r, 1::2] = c.real, -c.imag
            M[2 * r + 1, 0::2], M[2 * r + 1, 1::2] = c.imag, c.real
    if a.f64 or a.x2:
        Y = M @ X.astype(np.float64)
    else:
        Mh, Ml = f16split(M)
        Xh, Xl = f16split(X, rtz=not a.rtn)
        if a.desc:
            Y = mm32(Ml, Xh, mm32(Mh, Xl, mm32(Mh, Xh)))
            if a.four:
                Y = mm32(Ml, Xl, Y)
        else:
            Y = mm32(Ml, Xl) if a.four else None
            Y = mm32(Mh, Xh, mm32(Mh, Xl, mm32(Ml, Xh, Y)))
    return {k: Y[2 * r] + 1j * Y[2 * r + 1] for r, (k, c) in enumerate(rows)}

# rectangle spectra for the DC correction
ne, no = (W + 1) // 2, W // 2
Re = np.array([np.exp(-2j * np.pi * np.arange(ne) * k / 256).sum() for k in range(256)])
Ro = np.array([np.exp(-2j * np.pi * np.arange(no) * k / 256).sum() for k in range(256)])

def spectrum(u):
    T = (u.size - (W - SH)) // SH
    fr = np.zeros((T, NF + 1), dtype=np.int64)   # x[-1 .. 510]
    up = np.concatenate([[0], u.astype(np.int64), np.zeros(NF, dtype=np.int64)])
    for t in range(T):
        fr[t] = up[t * SH: t * SH + NF + 1]
    xm1, x = fr[:, :-1], fr[:, 1:]              # x[i-1], x[i], i = 0..511
    def split(v):
        hi = v >> 8
        return hi.astype(np.float32), (v - 256 * hi).astype(np.float32)
    Es = np.zeros((16, 16, T), dtype=np.float64 if a.f64 else np.float32)
    Os = np.zeros_like(Es)
    for n2 in range(16):
        i = 32 * np.arange(16) + 2 * n2
        ke = np.vstack([x[:, i].T, xm1[:, i].T])         # [xe | xp]  (32, T)
        ko = np.vstack([x[:, i + 1].T, x[:, i].T])       # [xo | xe]
        for dst, A, kk in ((Es, AE[n2], ke), (Os, AO[n2], ko)):
            if a.f64 or a.x1:
                dst[:, n2] = A @ kk
            else:
                if not a.blockk:  # (xe(n1), xp(n1)) adjacent: every pair of products is one pre-emphasised sample
                    perm = np.arange(32).reshape(2, 16).T.ravel()
                    A, kk = A[:, perm], kk[perm]
                h, l = split(kk)
                # scale s1 = 1/16 without pushing matrix terms into fp16's subnormal range: the high plane's matrix is
                # 256 s1 A (<= 16), the low plane keeps A itself and its DATA carry s1 (lo / 16 is exact in fp16)
                t256 = fsplit(A / a.s1 * 256.0, a.terms)   # both planes: scale in the data (hi / 16, lo / 16 are exact in fp16)
                t1 = fsplit(A / a.s1, a.terms)
                # smallest products first: only the last MFMA's rounding happens at the result's full magnitude
                chain = [(t256[j], h) for j in range(a.terms)] + [(t1[j], l) for j in range(a.terms - 1)]
                order = [2, 4, 1, 3, 0] if a.terms == 3 and not a.desc else range(len(chain))
                acc = None
                for j in order:
                    acc = mm32(chain[j][0], chain[j][1] * a.s1, acc)
                dst[:, n2] = acc
    P = np.zeros((T, 257))
    E, Tt = {}, {}
    for k1 in range(9):
        if k1 in (0, 8):
            r = 0 if k1 == 0 else 1
            E.update(apply_stage2(Es[r], k1, False))
            Tt.update(apply_stage2(Os[r], k1, True))
        else:
            E.update(apply_stage2((Es[2 * k1], Es[2 * k1 + 1]), k1, False))
            Tt.update(apply_stage2((Os[2 * k1], Os[2 * k1 + 1]), k1, True))
    m = ((E[0] + Tt[0]).real / W).astype(np.float64 if a.f64 else np.float32)   # mean of the windowed frame (scaled by s1)
    f = (lambda v: v) if a.f64 else (lambda v: v.astype(np.complex64))
    for k in E:
        if k == 0 or k > 255:
            continue
        e = f(E[k]) - m * f(Re[k % 256])
        t = f(Tt[k]) - m * f(Ro[k % 256] * np.exp(-2j * np.pi * k / 512))
        P[:, k] = np.abs(e + t) ** 2
        P[:, 256 - k] = np.abs(e - t) ** 2
    e0, t0 = E[0].real - m * Re[0].real, Tt[0].real - m * Ro[0].real
    P[:, 0] = 1e-10 * a.s1 ** 2
    P[:, 256] = (e0 - t0) ** 2
    return (P / a.s1 ** 2).astype(np.float64 if a.f64 else np.float32)

def mfcc(P):
    Y = (P.astype(np.float32) @ fb.T.astype(np.float32)).astype(np.float64) if not a.f64 else P @ fb.T
    L = np.log(Y)
    i = np.arange(13)[:, None]; k = np.arange(1, B + 1)[None, :]
    D = np.sqrt(2.0 / B) * np.cos(np.pi * i * (2 * k - 1) / (2 * B))
    c = L @ D.T
    n = np.arange(1, 13)
    c[:, 1:] *= 1 + 11 * np.sin(np.pi * n / 22)
    return np.hstack([c[:, 1:], c[:, :1]])

pcm = np.load(os.path.join(GOLDEN, "smfcc_mini_pcm.npz"))
utts = [pcm[k] for k in sorted(pcm.files, key=lambda s: int(''.join(ch for ch in s if ch.isdigit()) or 0))][:8] + [sig("CS0")[:40000]]
worst, allerr = 0.0, []
for u in utts:
    ref = orc.process(u)
    got = mfcc(spectrum(u))
    e = np.abs(got - ref) / np.maximum(np.abs(ref), 1.0)
    allerr.append(e.ravel()); worst = max(worst, e.max())
    if os.environ.get("PERUTT"):
        t_, c_ = np.unravel_index(e.argmax(), e.shape)
        print(f"  utt len {u.size}: max {e.max():.3e} at frame {t_} coef {c_} (ref {ref[t_, c_]:.4f}); p99 {np.quantile(e, .99):.2e}; peak |x| {np.abs(u.astype(int)).max()}")
e = np.concatenate(allerr)
print(f"four={a.four} rtn={a.rtn} s1={a.s1} f64={a.f64}: entries {e.size} max {e.max():.3e} p99.9 {np.quantile(e, .999):.3e} p99 {np.quantile(e, .99):.3e} median {np.median(e):.3e}")
if os.environ.get("DBG"):
    u = utts[0]
    P = spectrum(u)
    T = P.shape[0]
    up = np.concatenate([[0], u.astype(np.float64)])
    worstp = 0
    for t in range(T):
        x = up[t * SH + 1: t * SH + 1 + W]; xm = up[t * SH: t * SH + W]
        y = win * (x - pre * xm); y = y - y.sum() / W
        X = np.fft.rfft(y, NF); Pd = np.abs(X) ** 2; Pd[0] = 1e-10
        r = np.abs(P[t] - Pd) / Pd.max()
        if r.max() > worstp: worstp = r.max(); wt = (t, int(r.argmax()), P[t][r.argmax()], Pd[r.argmax()])
    print("worst spectrum deviation rel. to frame max", worstp, wt)
if os.environ.get("DBG1"):
    u = utts[0]
    T = (u.size - (W - SH)) // SH
    up = np.concatenate([[0], u.astype(np.int64), np.zeros(NF, dtype=np.int64)])
    fr = np.stack([up[t * SH: t * SH + NF + 1] for t in range(T)])
    xm1, x = fr[:, :-1], fr[:, 1:]
    n2 = 5
    i = 32 * np.arange(16) + 2 * n2
    kk = np.vstack([x[:, i].T, xm1[:, i].T])
    A = AE[n2] / a.s1
    exact = A @ kk * a.s1
    perm = np.arange(32).reshape(2, 16).T.ravel()
    Ap, kp = A[:, perm], kk[perm]
    hi = kp >> 8; lo = kp - 256 * hi
    for nt in (2, 3, 4):
        t256 = fsplit(Ap * 256.0, nt); t1 = fsplit(Ap, nt)
        full = sum(t @ (hi * a.s1) for t in t256) + sum(t @ (lo * a.s1) for t in t1[:nt - 1])   # float64 sums: matrix error only
        print(nt, "terms: matrix-only error / rms(exact):", np.abs(full - exact).max() / np.sqrt((exact ** 2).mean()))
    print("rms exact", np.sqrt((exact ** 2).mean()), "rms raw partial", np.sqrt(((Ap[:, 0::2] @ kp[0::2]) ** 2).mean()) * a.s1)
if os.environ.get("DBG2"):
    u = utts[0]
    T = (u.size - (W - SH)) // SH
    up = np.concatenate([[0], u.astype(np.int64), np.zeros(NF, dtype=np.int64)])
    fr = np.stack([up[t * SH: t * SH + NF + 1] for t in range(T)])
    xm1, x = fr[:, :-1], fr[:, 1:]
    for n2 in (0, 5, 11):
        i = 32 * np.arange(16) + 2 * n2
        kk = np.vstack([x[:, i].T, xm1[:, i].T])
        A = AE[n2] / a.s1
        exact = A @ kk * a.s1
        perm = np.arange(32).reshape(2, 16).T.ravel()
        Ap, kp = A[:, perm], kk[perm]
        for shift in (8, 6):
            hi = kp >> shift; lo = kp - (hi << shift)
            for nt, ntl in ((3, 2), (3, 3), (4, 3)):
                tH = fsplit(Ap * float(1 << shift), nt); tL = fsplit(Ap, ntl)
                chain = [(t, hi) for t in tH] + [(t, lo) for t in tL]
                chain.sort(key=lambda c: np.abs(c[0]).max() * np.abs(c[1]).max())
                acc = None
                for M_, d_ in chain:
                    acc = mm32(M_, d_ * a.s1, acc)
                full = sum(M_ @ (d_ * a.s1) for M_, d_ in chain)
                rms = np.sqrt((exact ** 2).mean())
                print(f"n2 {n2} shift {shift} terms hi {nt} lo {ntl}: matrix-only err rms {np.sqrt(((full - exact) ** 2).mean()) / rms:.2e} max {np.abs(full - exact).max() / rms:.2e} | with fp32 acc: rms {np.sqrt(((acc - exact) ** 2).mean()) / rms:.2e} max {np.abs(acc - exact).max() / rms:.2e} | fp32(exact) rms {np.sqrt(((exact.astype(np.float32) - exact) ** 2).mean()) / rms:.2e}")
