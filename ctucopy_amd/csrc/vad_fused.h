// Burg-cepstral VAD criterion fused into the front end (256-point mode): the wave that computed a step's eight spectra
// also rebuilds the eight time-domain frames the detector looks at and runs the Burg lattices, so neither the complex
// spectra nor the time-domain frames go through HBM.  Included by frontend_kernel.h.
//
// Reference chain (src/vad/vad.cc:220-236): fft_in = Xsabs[k] cos / sin (Xsph[k]) with Xsabs the vector AFTER noise
// reduction (a power spectrum when fb_power is on) and Xsph the phase of the ORIGINAL spectrum (src/io/in.cc:396-401);
// unnormalised FFTW_HC2R; the first `window` samples go to Burg (src/vdet/Burg.h:49-95) and on to cepstra (:141-152).
#pragma once

namespace {

// In-register 16-point inverse DFT (e^{+i...}): IDFT(x) = swap(DFT(swap(x))) with swap = exchange of re and im.
__device__ __forceinline__ void idft16(float2 (&v)[16]) {
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = make_float2(v[i].y, v[i].x);
    dft16(v);
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = make_float2(v[i].y, v[i].x);
}

constexpr int VF_NC = 14;         // cepstral coefficients of the fused path (-vad_lpc_coefs default: Burg order 13)
constexpr int SS_NC = 16;         // the *ss modes' detector uses -fea_ncepcoefs coefficients (src/nr/nr.cc:266-268: 12 in the presets): the lattice is
                                  // unrolled for up to 16 (one coefficient per lane of a frame's 16) and stops at the run-time count (KParams::ss_nc)
constexpr int VF_WINDOW = 200;    // the window the fused path is built for (8 kHz, 25 ms); other windows take the separate kernels
constexpr int VF_SPL = 13;        // samples per lane of a frame: 16 lanes x 13 = 208 >= window
constexpr int VF_LW = (VF_WINDOW - 1) / VF_SPL, VF_JW = (VF_WINDOW - 1) % VF_SPL;  // lane / register of the window's last sample
constexpr int VF8_SPL = 25;       // the same window as 8 lanes x 25 samples (the VAD module's float lattice, frontend_kernel.h VF8)
constexpr int VF_FSTRIDE = 216;   // floats per time-domain frame in the LDS staging area: 2 x 216 = 16 (mod 32), so the two
                                  // frame groups of a half wave read different banks; 8 x 216 <= the wave's 8 x 260
// The 512-point mode (16 kHz, 25 ms): one frame per 16-lane group, four frames per half step
constexpr int VF0_WINDOW = 400;   // the window the fused path of the 512-point mode is built for
constexpr int VF0_SPL = 25;       // 16 lanes x 25 = 400 samples
constexpr int VF0_LW = (VF0_WINDOW - 1) / VF0_SPL, VF0_JW = (VF0_WINDOW - 1) % VF0_SPL;
constexpr int VF0_FSTRIDE = 432;  // floats per staged frame: >= 416 (13 rows of 32 samples are written) and = 16 (mod 32): the two frame
                                  // groups of a half wave read 25 l16 + j from different banks
constexpr int VF0_STAGE = 4 * VF0_FSTRIDE;  // floats of staging per wave (behind the tables: these instantiations run one workgroup per CU)

// Stage A.  vz[r] = Z[l16 + 16 r] of the forward transform of (frame A + i frame B).  Every bin is scaled to the
// magnitude the noise reduction left (P rows of the two frames) while it keeps its direction:
//   XA'[k] = gA XA[k], XB'[k] = gB XB[k], g = Xsabs_after_NR / |X|  (bins 1..127; DC and Nyquist are real, see below)
//   Z'[k] = XA'[k] + i XB'[k] = hA (sr, si) + hB (dr, di),  h = g / 2,  (sr, si, dr, di) the untangle sums of Z[k], Z[256-k]
// and the same with the mirror's gains for k > 128 (Hermitian extension).
// SYN: speech synthesis (sigOUT, src/io/out.cc:405-427) instead of the detector's frame: magnitudes times `scale` (1/N),
// DC and Nyquist as positive reals (the reference stores them before its sign fix-up, out.cc:416-419).
template <bool SYN = false>
__device__ __forceinline__ void vf_scale_spectra(const float2 (&vz)[16], float2 (&vn)[16], const float *rowA, const float *rowB,
                                                 int l16, int partner, float scale = 1.f) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
        float br = mirror_fetch(vz[15 - r].x, partner);
        float bi = mirror_fetch(vz[15 - r].y, partner);
        if (l16 == 0) {  // lane 0 holds its own mirrors: bin 16 r <-> bin 16 (16 - r)
            br = vz[(16 - r) & 15].x;
            bi = vz[(16 - r) & 15].y;
        }
        const int k = l16 + 16 * r, kt = r < 8 ? k : 256 - k;  // kt <= 128: where the gains live
        const float pa = SYN ? rowA[kt] * scale : rowA[kt], pb = SYN ? rowB[kt] * scale : rowB[kt];  // Xsabs after NR
        const float ar = vz[r].x, ai = vz[r].y;
        const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
        const float ma = sr * sr + si * si, mb = dr * dr + di * di;  // 4 |XA|^2, 4 |XB|^2
        // 1 / (2 |X|) = rsqrt(4 |X|^2): v_rsq + one Newton step, as exact as its float input
        float ia = __builtin_amdgcn_rsqf(ma), ib = __builtin_amdgcn_rsqf(mb);
        ia = ia * (1.5f - 0.5f * ma * ia * ia);
        ib = ib * (1.5f - 0.5f * mb * ib * ib);
        const float ha = pa * ia, hb = pb * ib;
        float xar = ha * sr, xai = ha * si;      // XA'[k]
        float xbr = hb * di, xbi = -hb * dr;     // XB'[k] = gB (di, -dr) / 2
        // |X| = 0: c_ph(0, 0) = -pi/2 (src/io/in.cc:191-193), so X' = -i Xsabs (+i on the mirrored side)
        if (!(ma > 0.f)) { xar = 0.f; xai = r < 8 ? -pa : pa; }
        if (!(mb > 0.f)) { xbr = 0.f; xbi = r < 8 ? -pb : pb; }
        if (l16 == 0 && (r == 0 || r == 8)) {
            // DC: phase 0; Nyquist: 0 or pi by the sign of the real part (src/io/in.cc:398-399); both purely real.
            // Z[0] = XA[0] + i XB[0] and Z[128] likewise, with XA, XB real there.
            xar = (r == 0 || SYN) ? pa : (ar >= 0.f ? pa : -pa);
            xai = 0.f;
            xbr = (r == 0 || SYN) ? pb : (ai >= 0.f ? pb : -pb);
            xbi = 0.f;
        }
        vn[r] = make_float2(xar - xbi, xai + xbr);
    }
}

// Stage A of the 512-point mode (one real frame per 16-lane group, packed as z[n] = x[2n] + i x[2n+1]):
// vz[r] = Z[l16 + 16 r].  The forward untangle gives u = 2 X[k] and v with X[256-k] = conj(v) / 2 for the lane's bins
// k = l16 + 16 k2 and their mirrors; both are re-scaled to the magnitudes in `row` (times `scale`) and tangled back:
//   s' = (u' + v') / 2,  t' = (u' - v') / 2,  d' = i conj(w) t',  Z'[k] = (s' + d') / 2,  Z'[256-k] = conj((s' - d') / 2)
// The second one belongs to the mirror lane's register 15 - k2 and is exchanged by ds_bpermute (lane 0 owns its mirrors).
// SYN: synthesis conventions - DC and Nyquist (bin 256) as positive reals (sigOUT, src/io/out.cc:416-419); otherwise the
// detector's (src/vad/vad.cc:227-230 with the phases of src/io/in.cc:396-401): DC phase 0, Nyquist 0 or pi by the sign of its
// real part.  HC2R's x = 2 * IDFT_256-sum(Z'): `scale` carries that 2.
template <bool SYN = true>
__device__ __forceinline__ void vf_scale_tangle0(const float2 (&vz)[16], float2 (&vn)[16], const float *row, const float4 *ltw4, int l16,
                                                 int partner, float scale) {
    float2 bp[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
        const float4 u4q = ltw4[8 + (k2 >> 1)];
        float br = mirror_fetch(vz[15 - k2].x, partner);
        float bi = mirror_fetch(vz[15 - k2].y, partner);
        if (l16 == 0) {
            br = vz[(16 - k2) & 15].x;
            bi = vz[(16 - k2) & 15].y;
        }
        const float wr = (k2 & 1) ? u4q.z : u4q.x, wi = (k2 & 1) ? u4q.w : u4q.y;
        const float ar = vz[k2].x, ai = vz[k2].y;
        const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
        const float tr = wr * di + wi * dr, ti = wi * di - wr * dr;
        const float ur = sr + tr, ui = si + ti, vr = sr - tr, vi = si - ti;
        const int k = l16 + 16 * k2;
        const float tk = row[k] * scale, tm = row[256 - k] * scale;
        const float mu = ur * ur + ui * ui, mv = vr * vr + vi * vi;  // 4 |X[k]|^2, 4 |X[256-k]|^2
        float iu = __builtin_amdgcn_rsqf(mu), iv = __builtin_amdgcn_rsqf(mv);
        iu = iu * (1.5f - 0.5f * mu * iu * iu);
        iv = iv * (1.5f - 0.5f * mv * iv * iv);
        const float gu = 2.f * tk * iu, gv = 2.f * tm * iv;   // u' = 2 X'[k] = 2 tk u / |u|
        float upr = gu * ur, upi = gu * ui, vpr = gv * vr, vpi = gv * vi;
        if (!(mu > 0.f)) { upr = 0.f; upi = -2.f * tk; }      // c_ph(0, 0) = -pi/2: X' = -i t
        if (!(mv > 0.f)) { vpr = 0.f; vpi = 2.f * tm; }       // v' = conj(2 X'[256-k])
        if (l16 == 0 && k2 == 0) {                             // bins 0 and 256 are real: u = 2 X[0], v = 2 X[256]
            upr = 2.f * tk; upi = 0.f;
            vpr = (SYN || vr >= 0.f) ? 2.f * tm : -2.f * tm; vpi = 0.f;
        }
        const float spr = 0.5f * (upr + vpr), spi = 0.5f * (upi + vpi), tpr = 0.5f * (upr - vpr), tpi = 0.5f * (upi - vpi);
        const float dpr = wi * tpr - wr * tpi, dpi = wr * tpr + wi * tpi;   // i conj(w) t'
        vn[k2] = make_float2(0.5f * (spr + dpr), 0.5f * (spi + dpi));
        bp[k2] = make_float2(0.5f * (spr - dpr), -0.5f * (spi - dpi));
    }
    // bin 128 (lane 0, register 8): X[128] = conj(Z[128]), so Z'[128] = g Z[128]
    float2 z128;
    {
        const float m = vz[8].x * vz[8].x + vz[8].y * vz[8].y, t = row[128] * scale * 0.5f;  // Z' carries half of HC2R's 2 here: see below
        float im = __builtin_amdgcn_rsqf(m);
        im = im * (1.5f - 0.5f * m * im * im);
        // |X[128]| = |Z[128]|; the factor 2 of `scale` belongs to the u / v convention (u = 2X): here X' = tgt X / |X| directly
        z128 = m > 0.f ? make_float2(2.f * t * im * vz[8].x, 2.f * t * im * vz[8].y) : make_float2(0.f, 2.f * t);
    }
#pragma unroll
    for (int r = 8; r < 16; r++) {
        const float orr = mirror_fetch(bp[15 - r].x, partner);
        const float ori = mirror_fetch(bp[15 - r].y, partner);
        const float2 l0 = r == 8 ? z128 : bp[(16 - r) & 7];
        vn[r] = l16 == 0 ? l0 : make_float2(orr, ori);
    }
}

// Stage B.  z[n] = sum_k Z'[k] e^{+2 pi i n k / 256} (unnormalised, FFTW_HC2R of the two Hermitian spectra at once): the
// forward transform run backwards.  In: vn[k2] = Z'[k1 + 16 k2] in lane k1.  IDFT16 over k2 -> register m2; twiddle
// W256^{-k1 m2} (the conjugate of the forward table, lane = k1); transpose; IDFT16 over k1 -> register m1:
// out vn[m1] = z[l16 + 16 m1] = (frame A sample, frame B sample).
__device__ __forceinline__ void vf_inverse_fft(float2 (&vn)[16], const float4 *ltw4, uint32_t sbase, const float *rd) {
    idft16(vn);
#pragma unroll
    for (int h = 0; h < 8; h++) {
        const float4 tw = ltw4[h];  // W256^(l16 (2h+1)), W256^(l16 (2h+2))
        vn[2 * h + 1] = cmul(vn[2 * h + 1], make_float2(tw.x, -tw.y));
        if (2 * h + 2 < 16) vn[2 * h + 2] = cmul(vn[2 * h + 2], make_float2(tw.z, -tw.w));
    }
    wave_transpose16(vn, sbase, rd);
    idft16(vn);
}

// Stages C + D.  One real frame per 16-lane group, 13 consecutive samples per lane (sample i = 13 l16 + j), zero beyond
// `window`.  Burg lattice (src/vdet/Burg.h:49-95) with the order loop unrolled: the prediction coefficients are plain
// registers, the same in every lane of the group; the two sums of an order are 16-lane DPP all-reduces.  Then the a -> c
// recursion (Burg.h:141-152).  cc[0..NC-1] are the cepstra (cc[0] = ln alpha).
//
// Range handling without per-sample masks (every lane runs the same instructions):
//   * the sums of order ik run over i >= ik (Burg.h:70-74).  Samples below 13 >= ik live in lane 0 only; ef[i < ik] and
//     eb[i < ik - 1] are dead by then, so lane 0 zeroes ef[ik-1] and eb[ik-2] at the start of the order and they drop out;
//   * eb[window-1] is written by every order but only ever read for the sample after the window: it is kept at zero
//     (lane `lw`, register `jw`), which also keeps ef / eb at zero beyond the window;
//   * eb[i-1] of a lane's first sample is the previous lane's last register (row_ror:1); lane 0 receives lane 15's,
//     which lies beyond the window: zero.
// JW >= 0: the register of eb[window-1] is known at compile time (window 200: lane 15, register 4); JW < 0: run-time jw.
// NC = number of cepstral coefficients (orders 1..NC-1), a compile-time constant: straight-line code, no joins.
// T = float (the VAD module's criterion: identical decisions to the double oracle on every test recording) or double
// (the *ss modes' detector: it sees spectra raised to the power a, whose frames are nearly sinusoidal - reflection
// coefficients within 1e-6 of +-1 - and a float lattice then loses the cepstra's digits that the threshold test needs).
#ifndef CTU_BURG_RCP
#define CTU_BURG_RCP 1  // 0: the reflection coefficient by the IEEE division sequence (A/B)
#endif
#ifndef CTU_BURG_DREC
#define CTU_BURG_DREC 0  // 1: the denominator of order m+1 from that of order m, D' = (1 - k^2) D - f[m]^2 - b[N-1]^2, instead of the sum
#endif
// The tail of the Burg estimator behind the lattice: from the reflection coefficients (FROM_RC: the polynomial is rebuilt first,
// a_i <- a_i + k a_{m-i}, Burg.h:88-93) to the LPC cepstrum c_0 = ln alpha, c_m = -a_m - (1/m) sum_{k<m} (m-k) c_{m-k} a_k
// (Burg.h:143-151).  One definition for the front end (all lanes of a frame) and for vad_a2c_kernel (one frame per lane), so that
// both round alike.
template <int NC, class T, bool FROM_RC>
__device__ __forceinline__ void vf_lattice_to_cepstrum(T alpha, T (&a)[NC], T (&cc)[NC]) {
    if constexpr (FROM_RC) {  // a[m] holds the reflection coefficient of order m on entry
        T rcs[NC];
#pragma unroll
        for (int i = 0; i < NC; i++) {
            rcs[i] = a[i];
            a[i] = i == 0 ? (T)1 : (T)0;
        }
#pragma unroll
        for (int ik = 1; ik < NC; ik++) {
            const T rc = rcs[ik];
            T an[NC];
#pragma unroll
            for (int i = 1; i < ik; i++) an[i] = a[i] + rc * a[ik - i];
#pragma unroll
            for (int i = 1; i < ik; i++) a[i] = an[i];
            a[ik] = rc;
        }
    }
    if constexpr (sizeof(T) == 8) cc[0] = log(alpha);
    else cc[0] = __builtin_amdgcn_logf(alpha) * 0.69314718056f;
#pragma unroll
    for (int m = 1; m < NC; m++) {
        T sum = 0;
#pragma unroll
        for (int k = 1; k < m; k++) sum += (T)(m - k) * cc[m - k] * a[k];
        cc[m] = -a[m] - sum * ((T)1 / (T)m);
    }
}

// RC_ONLY: stop at the lattice - cc[0] = the residual energy alpha, cc[m] = reflection coefficient of order m.  The coefficient
// recursion (Burg.h:88-93) and a -> c (Burg.h:143-151) are one short sequential recursion per FRAME which all sixteen lanes of a
// frame would repeat: vad_a2c_kernel (vad_kernels.h) finishes them one frame per lane, with these very expressions.
// LPF: lanes per frame.  16: a frame fills a DPP row; 8: two frames per row (the 200-sample window as 8 lanes x 25 samples, all eight
// frames of a step in one call) - the sums then stay inside the half rows.
// RC_ONLY with FOLD: cc[h] of the frame's lane i ends up holding entry LPF h + i of {alpha, k_1 .. k_{NC-1}} - what that lane stores - and
// every coefficient is dead as soon as its order is done (thirteen registers fewer across the lattice than keeping them for a final select).
// nc_rt: the number of coefficients wanted (<= NC, wave-uniform): orders nc_rt .. NC - 1 are skipped (their a[] stay zero and their
// cepstra are not read).
template <int NC, int JW, class T, int SPL = VF_SPL, bool RC_ONLY = false, int LPF = 16, bool FOLD = false>
__device__ __forceinline__ void vf_burg_cepstrum(const float (&x)[SPL], int l16, int lw, int jw, T inv_w, T (&cc)[NC], int nc_rt = NC) {
    constexpr int VF_SPL = SPL;  // samples per lane (13: 256-point mode, 25: 512-point mode); shadows the global of the same name
    T ef[VF_SPL], eb[VF_SPL];
    T part = 0, part1 = 0;
#pragma unroll
    for (int j = 0; j < VF_SPL; j++) {
        ef[j] = eb[j] = (T)x[j];
        if (j & 1) part1 += ef[j] * ef[j];
        else part += ef[j] * ef[j];
    }
    auto row_sum = [](T v) {
        if constexpr (LPF == 8) {
            v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
            v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
            v += dpp_mov<0x141>(v);  // row_half_mirror: the other quad of the half row
        } else {
            v += dpp_mov<0x128>(v);  // row_ror:8
            v += dpp_mov<0x124>(v);  // row_ror:4
            v += dpp_mov<0x122>(v);  // row_ror:2
            v += dpp_mov<0x121>(v);  // row_ror:1
        }
        return v;
    };
    T alpha = row_sum(part + part1) * inv_w;  // Energy.h:38-44 / Burg.h:62
    const bool lane0 = l16 == 0, lanew = l16 == lw;
    auto clear_last = [&] {
        if constexpr (JW >= 0) eb[JW] = lanew ? (T)0 : eb[JW];
        else {
#pragma unroll
            for (int j = 0; j < VF_SPL; j++)
                if (j == jw) eb[j] = lanew ? (T)0 : eb[j];
        }
    };
    clear_last();
    T a[NC];
    T den_next = 0;
    if constexpr (RC_ONLY && FOLD) {
#pragma unroll
        for (int h = 0; h < (NC + LPF - 1) / LPF; h++) cc[h] = 0;
    }
#pragma unroll
    for (int i = 0; i < NC; i++) a[i] = i == 0 ? (T)1 : (T)0;
#pragma unroll
    for (int ik = 1; ik < NC; ik++) {
        if (ik < nc_rt) {
            if (ik - 1 < VF_SPL) ef[ik - 1] = lane0 ? (T)0 : ef[ik - 1];
            if (ik >= 2 && ik - 2 < VF_SPL) eb[ik - 2] = lane0 ? (T)0 : eb[ik - 2];
            // row_ror:1: the lane before this one; a frame's first lane reads the last lane of a neighbour - the entry clear_last() keeps
            // at zero (LPF = 8: SPL * LPF = window exactly) or a sample of the padding beyond the window (zero as well)
            const T eb_prev = dpp_mov<0x121>(eb[VF_SPL - 1]);
            constexpr bool DREC = CTU_BURG_DREC && sizeof(T) == 4 && JW >= 0 && LPF == 16;
            T n0 = 0, n1 = 0, d0 = 0, d1 = 0;
#pragma unroll
            for (int j = 0; j < VF_SPL; j++) {
                const T bm = j == 0 ? eb_prev : eb[j - 1];
                if (j & 1) {
                    n1 += ef[j] * bm;
                    if (!DREC || ik == 1) {
                        d1 += ef[j] * ef[j];
                        d1 += bm * bm;
                    }
                } else {
                    n0 += ef[j] * bm;
                    if (!DREC || ik == 1) {
                        d0 += ef[j] * ef[j];
                        d0 += bm * bm;
                    }
                }
            }
            const T num = row_sum(n0 + n1);
            const T den = (!DREC || ik == 1) ? row_sum(d0 + d1) : den_next;
            T rc;
            if constexpr (CTU_BURG_RCP && sizeof(T) == 4) {
                // the float lattice's quotient by v_rcp_f32 and one correction of the quotient (the residual by FMA): as exact as the
                // IEEE sequence for a denominator in the normal range (a sum of squares of samples), four instructions instead of ten
                // on the chain every lane of the frame waits for
                const float r = __builtin_amdgcn_rcpf(den), n2 = -2.f * num;
                const float q = n2 * r;
                rc = __builtin_fmaf(__builtin_fmaf(-den, q, n2), r, q);
            } else
                rc = -((T)2 * num) / den;
            alpha *= (T)1 - rc * rc;
            T carry = eb_prev;
#pragma unroll
            for (int j = 0; j < VF_SPL; j++) {  // both updates from the old values (Burg.h:80-86)
                const T bm = carry;
                carry = eb[j];
                const T nef = ef[j] + rc * bm, neb = bm + rc * ef[j];
                ef[j] = nef;
                eb[j] = neb;
            }
            if constexpr (DREC) {
                // sample ik's forward error (lane 0, or lane 1's first register) and the window's last backward error
                const T fe = ik < VF_SPL ? (lane0 ? ef[ik < VF_SPL ? ik : 0] : (T)0) : (l16 == 1 ? ef[0] : (T)0);
                const T be = lanew ? eb[JW] : (T)0;
                den_next = ((T)1 - rc * rc) * den - row_sum(fe * fe + be * be);
            }
            clear_last();
            if constexpr (RC_ONLY && FOLD) cc[ik / LPF] = l16 == ik % LPF ? rc : cc[ik / LPF];
            else if constexpr (RC_ONLY) cc[ik] = rc;
            else {
                T an[NC];
#pragma unroll
                for (int i = 1; i < ik; i++) an[i] = a[i] + rc * a[ik - i];
#pragma unroll
                for (int i = 1; i < ik; i++) a[i] = an[i];
                a[ik] = rc;
            }
        }
    }
    if constexpr (RC_ONLY && FOLD) cc[0] = lane0 ? alpha : cc[0];
    else if constexpr (RC_ONLY) cc[0] = alpha;
    else vf_lattice_to_cepstrum<NC, T, false>(alpha, a, cc);
}

// X^a and X^(1/a) of the *ss modes (src/nr/nr.cc:229-234, 252-257): exact for a = 1, 2, pow otherwise.
__device__ __forceinline__ float ss_pow(float x, float a) { return a == 1.0f ? x : (a == 2.0f ? x * x : powf(x, a)); }
__device__ __forceinline__ float ss_root(float x, float a) { return a == 1.0f ? x : (a == 2.0f ? sqrtf(x) : powf(x, 1.0f / a)); }

// CepstralDetector<BurgCepstrumEstimator>::Process without the cepstrum estimation (src/vdet/CepstralDet.h:140-194),
// one frame at a time, every lane of the wave in step: lane i keeps c0[i], the rest is wave-uniform.
struct CepDetRun {
    double c0;  // background cepstrum, coefficient `lane`
    double dMean, dMean2, threshold;
    int nseg;
};
__device__ __forceinline__ void cepdet_reset(CepDetRun &d) {
    d.c0 = d.dMean = d.dMean2 = d.threshold = 0.0;
    d.nseg = 0;
}
// cil: coefficient `lane` of the frame's cepstrum (0 beyond nc).  Returns the decision (1 = speech).
__device__ __forceinline__ int cepdet_frame(CepDetRun &d, double cil, int lane, int nc, int ninit, double pcoef, double qcoef) {
    int result = 0;
    if (d.nseg == 0) d.c0 = cil;
    else {
        if (d.nseg == 1) d.c0 = (d.c0 + cil) / 2.0;
        const double dl = (lane >= 1 && lane < nc) ? cil - d.c0 : 0.0;  // the first coefficient is skipped (CepstralDet.h:64-72)
        const double dist = 4.3429 * sqrt(2 * wave_sum_fast(dl * dl));
        if (d.nseg == 1) {
            d.dMean = dist;
            d.dMean2 = dist * dist;
            d.threshold = d.dMean;
        } else {
            result = (d.nseg > ninit && dist >= d.threshold);
            if (!result) {
                d.c0 = pcoef * d.c0 + (1 - pcoef) * cil;
                d.dMean = qcoef * d.dMean + (1 - qcoef) * dist;
                d.dMean2 = qcoef * d.dMean2 + (1 - qcoef) * dist * dist;
                const double dVar = d.dMean2 - d.dMean * d.dMean;
                d.threshold = d.dMean + 2.0 * sqrt(dVar);
            }
        }
    }
    ++d.nseg;
    return result;
}

// The coefficient recursion and a -> c of the fused Burg-cepstral criterion, one frame per lane (see RC_ONLY above): a row of the
// scratch holds {alpha, k_1 .. k_{NC-1}} when the front end leaves it and the NC cepstra afterwards, which is what vad_lanes_kernel
// reads.  The front end's sixteen lanes per frame spent 65 vector instructions per frame repeating this; here it is 4.
template <int NC>
__global__ __launch_bounds__(256) void vad_a2c_kernel(float *__restrict__ cf, int64_t total_frames) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total_frames) return;
    float4 *row = reinterpret_cast<float4 *>(cf + t * VFC_STRIDE);
    float v[VFC_STRIDE];
#pragma unroll
    for (int q = 0; q < VFC_STRIDE / 4; q++) {
        const float4 x = row[q];
        v[4 * q] = x.x;
        v[4 * q + 1] = x.y;
        v[4 * q + 2] = x.z;
        v[4 * q + 3] = x.w;
    }
    float a[NC], cc[NC];
#pragma unroll
    for (int i = 0; i < NC; i++) a[i] = v[i];
    vf_lattice_to_cepstrum<NC, float, true>(v[0], a, cc);
#pragma unroll
    for (int i = 0; i < NC; i++) v[i] = cc[i];
    // what the distance does not sum - coefficient 0 and the row's padding - leaves as zeros: vad_lanes_kernel needs no masks then
    // (its background cepstrum of such an entry stays zero, and so does the entry's term)
    v[0] = 0.f;
#pragma unroll
    for (int i = NC; i < VFC_STRIDE; i++) v[i] = 0.f;
#pragma unroll
    for (int q = 0; q < VFC_STRIDE / 4; q++) row[q] = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
}

}  // namespace
