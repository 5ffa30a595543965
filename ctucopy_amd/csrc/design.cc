// Host-side table design (double precision); see design.h for what each table restates.
#include "design.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace ctu {
namespace {

// Frequency warping used both for the bin axis and for band edges (src/fea/fb.cc:105-131,313-337).
double warp(const std::string &scale, double f) {
    if (scale == "lin") return f;
    if (scale == "bark") return 6. * std::log(f / 600. + std::sqrt((f / 600.) * (f / 600.) + 1.));
    if (scale == "expolog") return f <= 2000 ? 700. * (std::pow(10., f / 3988.) - 1.) : 2595. * std::log10(1. + f / 700.);
    if (scale == "mel") return 2595 * std::log10(1. + f / 700.);
    throw DesignError("FB: Unknown frequency scale!");
}

// Centre of a band back in Hz, for the equal-loudness point (src/fea/fb.cc:358-372).
double unwarp_mid(const std::string &scale, double w) {
    if (scale == "lin") return w;
    if (scale == "bark") return 600 * std::sinh(w / 6.);
    if (scale == "expolog") return w <= 1521.4 ? 3988. * std::log10(1. + (w / 700.)) : 700. * (std::pow(10., w / 2595.) - 1);
    return 700. * (std::pow(10., w / 2595.) - 1.);  // mel
}

// 40 dB equal-loudness curve, with the extra pole above 10 kHz sampling (src/fea/fb.cc:157-164,376-383).
double equal_loudness(double om, int fs) {
    // Products are taken left to right exactly as written there: om^4 is ((om*om)*om)*om, not (om^2)^2,
    // because the tables must agree with the reference to the last bit in double.
    const double num = om * om * om * om * (om * om + 5.68e7);
    double den = (om * om + 6.3e6) * (om * om + 6.3e6) * (om * om + 3.8e8);
    if (fs > 10000) den = den * (om * om * om * om * om * om + 9.58e26);
    return num / den;
}

struct SubBank {
    double f_lo, f_hi;
    int of, first, last;
};

size_t span_of(const std::string &s, size_t pos, const char *set) { return std::strspn(s.c_str() + pos, set); }

// Grammar: token = [[X-YHz:]K-L/]Nfilters, tokens separated by ',' (src/fea/fb.cc:186-253).
std::vector<SubBank> parse_definition(const std::string &def, int fs) {
    const DesignError bad("FB: Filter bank specification parse error!");
    std::vector<SubBank> out;
    size_t p = 0;
    int nfilt = 0;
    while (p <= def.size()) {
        size_t q = def.find(',', p);
        std::string tok = def.substr(p, q == std::string::npos ? std::string::npos : q - p);
        p = (q == std::string::npos) ? def.size() + 1 : q + 1;
        if (tok.empty()) continue;  // strtok skips empty tokens
        SubBank sb{0., fs / 2., 0, 1, 0};
        size_t at = 0;
        const double x = std::atof(tok.c_str());
        at += span_of(tok, at, ".1234567890");
        if (tok.compare(at, 1, "-") == 0) {
            at++;
            const double y = std::atof(tok.c_str() + at);
            at += span_of(tok, at, ".1234567890");
            if (tok.compare(at, 3, "Hz:") == 0) {
                at += 3;
                sb.f_lo = x;
                sb.f_hi = y;
                sb.first = std::atoi(tok.c_str() + at);
                at += span_of(tok, at, "1234567890");
                if (tok.compare(at, 1, "-") != 0) throw bad;
                sb.last = std::atoi(tok.c_str() + ++at);
                at += span_of(tok, at, "1234567890");
                if (tok.compare(at, 1, "/") != 0) throw bad;
                nfilt = std::atoi(tok.c_str() + ++at);
                at += span_of(tok, at, "1234567890");
                if (tok.compare(at, 7, "filters") != 0) throw bad;
            } else if (tok.compare(at, 1, "/") == 0) {
                sb.first = (int)x;
                sb.last = (int)y;
                at++;
                nfilt = std::atoi(tok.c_str() + at);
                at += span_of(tok, at, "1234567890");
                if (tok.compare(at, 7, "filters") != 0) throw bad;
            } else throw bad;
        } else if (tok.compare(at, 7, "filters") == 0) {
            sb.last = (int)x;
            nfilt = sb.last;
        } else throw bad;
        sb.of = nfilt;
        out.push_back(sb);
    }
    return out;
}

}  // namespace

Design::Design(const Opts &opts) : o(opts) {
    window = o.window;
    wshift = o.wshift;
    wfft = o.wfft;
    K = o.wfftby2;
    if (wfft < 8 || window < 2 || wshift < 1) throw DesignError("OPTS: unusable window / shift");

    // ---- Hamming window; pi = 2*asin(1) as in src/io/in.cc:141
    hamming.resize(window);
    {
        const double pi = 2. * std::asin(1.);
        for (int j = 0; j < window; j++) hamming[j] = 0.54 - (1 - 0.54) * std::cos(2 * pi * j / (window - 1.));
    }

    signal_out = (o.format_out == "raw" || o.format_out == "wave");
    if (signal_out) {
        // row N3: no FB, no FEA.  OLA correction = maximum over the phases of the shift of the summed windows
        // (src/io/out.cc:355-377; pi = 2*asin(1) there too)
        const double pi = 2. * std::asin(1.);
        ola_corr = 0.;
        for (int i = 0; i < wshift; i++) {
            double y = 0.;
            for (int x = i; x < window; x += wshift) y += 0.54 - (1 - 0.54) * std::cos(2 * pi * (double)x / (window - 1.));
            if (y > ola_corr) ola_corr = y;
        }
        kind = FeaKind::None;
        B = 0; nfea = 0; D = Dbase = 0; htk_kind = 0;
        period = (unsigned)std::floor(.5 + 10000000. * wshift / (double)o.fs);
        return;
    }
    // ---- filter bank
    const bool plp = (o.fb_shape == "trapez");
    if (plp) {  // src/fea/fb.cc:44-54
        o.fb_scale = "bark";
        o.fb_inld = true;
        o.fb_eqld = true;
    }
    std::vector<double> axis(K, 0.0);
    if (o.fb_scale == "lin" || o.fb_scale == "bark" || o.fb_scale == "expolog" || o.fb_scale == "mel")
        for (int i = 0; i < K; i++) axis[i] = warp(o.fb_scale, (double)i * o.fs / (double)wfft);
    if (plp) {  // src/fea/fb.cc:134-184
        const double max_bark = 6 * std::log(o.fs / 1200. + std::sqrt((o.fs / 1200.) * (o.fs / 1200.) + 1.));
        const int n_bark = (int)std::floor(max_bark + .5);
        const double step = max_bark / (double)n_bark;
        for (int i = 0; i < n_bark - 1; i++) {
            const double centre = (i + 1) * step;
            const double el = equal_loudness(3.1415926535898 * 1200 * std::sinh(centre / 6), o.fs);
            std::vector<double> row(K);
            for (int k = 0; k < K; k++) {
                const double d = axis[k] - centre;
                double v;
                if (d >= -1.3 && d <= -.5) v = std::pow(10., 2.5 * (0.5 + d));
                else if (std::fabs(d) < 0.5) v = 1;
                else if (d >= 0.5 && d <= 2.5) v = std::pow(10., 0.5 - d);
                else v = 0;
                if (o.fb_eqld) v *= el;
                row[k] = v;
            }
            fb.push_back(std::move(row));
        }
    } else {
        const bool rect = (o.fb_shape == "rect");
        if (!rect && o.fb_shape != "triang") throw DesignError("FB: Unknown filter shape!");
        std::vector<SubBank> banks = parse_definition(o.fb_definition, o.fs);
        if (rect) {  // abutting rectangular sub-banks must not share their edge bin (src/fea/fb.cc:257-279)
            const double df = o.fs / (double)wfft;
            for (SubBank &a : banks) {
                bool joined = false;
                for (const SubBank &b : banks) joined |= (a.f_hi == b.f_lo);
                if (!joined) a.f_hi += df;
            }
        }
        for (const SubBank &sb : banks) {
            const double w_lo = warp(o.fb_scale, sb.f_lo), w_hi = warp(o.fb_scale, sb.f_hi);
            for (int b = sb.first; b <= sb.last; b++) {
                if (fb.size() >= 998) throw DesignError("FB: Too many filters in FB!");
                double w0, w1;
                if (rect) {
                    w0 = w_lo + (b - 1.) * (w_hi - w_lo) / (double)(sb.of);
                    w1 = w_lo + (b + 0.) * (w_hi - w_lo) / (double)(sb.of);
                } else {
                    w0 = w_lo + (b - 1.) * (w_hi - w_lo) / (double)(sb.of + 1);
                    w1 = w_lo + (b + 1.) * (w_hi - w_lo) / (double)(sb.of + 1);
                }
                double el = 1.;
                if (o.fb_eqld) {
                    const double mid = w0 + (w1 - w0) / 2.;
                    el = equal_loudness(2 * 3.141592653589793 * unwarp_mid(o.fb_scale, mid), o.fs);
                }
                std::vector<double> row(K);
                double area = 0;
                for (int i = 0; i < K; i++) {
                    if (rect) {
                        const bool in = axis[i] >= w0 && axis[i] < w1;
                        row[i] = in ? 1 : 0;
                        if (in) area++;
                    } else if (axis[i] < w0 || axis[i] > w1) {
                        row[i] = 0;
                    } else {
                        const double mid = w0 + (w1 - w0) / 2.;
                        row[i] = 1. - 2. * std::fabs(mid - axis[i]) / (w1 - w0);
                        area += row[i];
                    }
                }
                const double g = o.fb_norm ? el / area : el;
                for (int i = 0; i < K; i++) row[i] *= g;
                fb.push_back(std::move(row));
            }
        }
    }
    B = (int)fb.size();
    if (B < 1) throw DesignError("FB: empty filter bank");
    fb_first.resize(B);
    fb_last.resize(B);
    for (int b = 0; b < B; b++) {  // first non-zero, then the end of that non-zero run (src/fea/fb.cc:432-447)
        int k = 0;
        // a filter that covers no bin has area 0; with -fb_norm the reference divides by it: NaN weights, NaN features
        for (int i = 0; i < K; i++)
            if (fb[b][i] != fb[b][i]) throw DesignError("FB: filter with no spectral bin (NaN weights in the reference)");
        while (k < K && fb[b][k] == 0) k++;
        if (k == K) throw DesignError("FB: filter with no spectral bin (undefined in the reference)");
        fb_first[b] = k++;
        while (k < K && fb[b][k] != 0) k++;
        fb_last[b] = k - 1;
    }

    // ---- feature kind and its tables
    const std::string &fk = o.fea_kind;
    const int ncep = o.fea_ncepcoefs, p = o.fea_lporder;
    if (fk == "spec") kind = FeaKind::Spec;
    else if (fk == "logspec") kind = FeaKind::LogSpec;
    else if (fk == "dctc") kind = FeaKind::Dctc;
    else if (fk == "lpa") kind = FeaKind::Lpa;
    else if (fk == "lpc") kind = FeaKind::Lpc;
    else if (fk == "trapdct") kind = FeaKind::TrapDct;
    else throw DesignError("FEA: Unknown feature kind!");

    if (kind == FeaKind::Dctc || kind == FeaKind::Lpc) {
        if (ncep < 1) throw DesignError("FEA: fea_ncepcoefs must be positive");
        lifter.resize(ncep);
        for (int n = 0; n < ncep; n++)
            lifter[n] = 1 + ((double)o.fea_lifter) / 2 * std::sin(3.141592653589793 * (n + 1.) / ((double)o.fea_lifter));
    }
    switch (kind) {
        case FeaKind::Spec:
        case FeaKind::LogSpec:
            nfea = B;
            break;
        case FeaKind::Dctc: {  // c_i = sqrt(2/B) sum_k X_{k-1} wdct[(2k-1) i mod 4B]  (src/fea/fea_impl.cc:92-122)
            nfea = ncep + 1;
            std::vector<double> wdct(4 * B);
            for (int i = 0; i < 4 * B; i++) wdct[i] = std::cos(3.1415926535898 * (double)i / (2 * B));
            const double norm = std::sqrt(2.0 / B);
            dct.assign((size_t)nfea * B, 0.0);
            for (int i = 0; i < nfea; i++)
                for (int k = 1; k <= B; k++) {
                    double v = wdct[(2 * k - 1) * i % (4 * B)] * norm;
                    if (i >= 1 && o.fea_lifter > 1) v *= lifter[i - 1];
                    dct[(size_t)i * B + (k - 1)] = v;
                }
            break;
        }
        case FeaKind::Lpa:
        case FeaKind::Lpc: {  // R[k] = (2/N)(Y0/2 + sum Y_n cos(2 pi n k / N) + (-1)^k Y_{B-1}/2), N = 2(B-1)
            if (B < 2) throw DesignError("FEA: LPC needs at least two bands");
            if (p < 1) throw DesignError("FEA: fea_lporder must be positive");
            nfea = (kind == FeaKind::Lpa) ? p + 1 : ncep + 1;
            const int N = 2 * (B - 1);
            std::vector<double> wre(N);
            for (int i = 0; i < N; i++) wre[i] = std::cos(2 * 3.141592653589793 * i / N);
            idft.assign((size_t)(p + 1) * B, 0.0);
            for (int k = 0; k <= p; k++) {
                idft[(size_t)k * B + 0] = 0.5 / ((double)N / 2);
                for (int n = 1; n < B - 1; n++) idft[(size_t)k * B + n] = wre[(n * k) % N] / ((double)N / 2);
                idft[(size_t)k * B + B - 1] += (1 - 2 * (k % 2)) * 0.5 / ((double)N / 2);
            }
            break;
        }
        case FeaKind::TrapDct: {
            const int T = o.fea_trapdct_traplen, nd = o.fea_trapdct_ndct;
            if (T % 2 == 0) throw DesignError("FEA: TRAP length must be odd!");
            if (nd >= T) throw DesignError("FEA: Number of DCT coeffs must be less than TRAP length (c0 is not output)!");
            if (nd < 1) throw DesignError("FEA: TRAP needs at least one DCT coefficient");
            nfea = B * nd;
            // out_k = 2 sum_j (x_j - mean) h_j cos(pi (j+1/2) k / T)  ==  sum_j x_j G[k][j]
            std::vector<double> h(T);
            for (int i = 0; i < T; i++) h[i] = 0.54 - (1 - 0.54) * std::cos(2 * 3.14159265359 * i / (T - 1.));
            trap.assign((size_t)nd * T, 0.0);
            for (int k = 1; k <= nd; k++) {
                double s = 0;
                std::vector<double> hc(T);
                for (int j = 0; j < T; j++) {
                    hc[j] = 2.0 * h[j] * std::cos(3.14159265358979323846 * (j + 0.5) * k / T);
                    s += hc[j];
                }
                for (int j = 0; j < T; j++) trap[(size_t)(k - 1) * T + j] = hc[j] - s / T;
            }
            break;
        }
        case FeaKind::None: break;  // not reached: the signal path returned above
    }

    // ---- output row layout (src/io/out.cc:95-113,174-203) and HTK header fields (145-171)
    if ((kind == FeaKind::Lpa || kind == FeaKind::Spec || kind == FeaKind::LogSpec) && o.fea_c0) o.fea_c0 = false;
    row_slot.assign(nfea, -1);
    int size = nfea;
    if (kind == FeaKind::Spec || kind == FeaKind::LogSpec || kind == FeaKind::TrapDct) {
        for (int i = 0; i < nfea; i++) row_slot[i] = i;
        if (o.fea_E) e_slot = nfea;
    } else {
        // coefficients 1..ncep first, c0 (or a0: never) after them, E last; the reference indexes this
        // with fea_ncepcoefs even for lpa, so lpa is only well-defined when fea_ncepcoefs == fea_lporder.
        if (kind == FeaKind::Lpa && ncep != p) throw DesignError("OUT: fea_kind lpa needs -fea_ncepcoefs equal to -fea_lporder");
        for (int i = 1; i <= ncep && i < nfea; i++) row_slot[i] = i - 1;
        if (kind == FeaKind::Lpa) size -= 1;
        else if (o.fea_c0) row_slot[0] = ncep;
        else size -= 1;
        if (o.fea_E) e_slot = (o.fea_c0 && kind != FeaKind::Lpa) ? nfea : ncep;
    }
    if (o.fea_E) size++;
    Dbase = D = size;
    if (o.fea_delta || o.fea_trap) {
        // BATCH::init_delta (src/io/batch.cc:122-130): n_order chained deltaFEA stages, each sized on
        // fea_c = fea_ncepcoefs+1 (src/fea/fea_delta.cc:22-28); OUT then sees fea_c*(n_order+1) values, or
        // fea_c*(2*d_win+1) for -fea_trap, plus E (src/io/out.cc:95-113 with Xsize of the last stage).
        const int fea_c = ncep + 1;
        post_order = o.n_order;
        post_stack = o.fea_trap;
        post_w[0] = o.d_win; post_w[1] = o.a_win; post_w[2] = o.t_win;
        for (int j = 0; j < post_order; j++)
            if (post_w[j] < 1) throw DesignError(o.fea_trap ? "FEA: Trap window size must be >= 3!" : "FEA: Delta window size must be > 1!");
        if (post_order > 0) {
            int xs = post_stack ? fea_c * (2 * o.d_win + 1) : fea_c * (post_order + 1);
            if ((kind == FeaKind::Lpc || kind == FeaKind::Dctc) && !o.fea_c0) xs--;
            if (kind == FeaKind::Lpa) xs--;
            D = xs + (o.fea_E ? 1 : 0);
        }
    }
    if (o.fea_Z_exp > 0) cms = 1;
    if (o.fea_Z_block > 0) cms = 2;  // block wins when both are given (src/fea/post_impl.cc:163-167)
    if (cms) cms_cols = ncep + (o.fea_c0 ? 1 : 0);
    if (D > 32767) throw DesignError("OUT: HTK format does not support more than 32767 features!");
    period = (unsigned)std::floor(.5 + 10000000. * wshift / (double)o.fs);
    int kcode = 9;
    if (kind == FeaKind::Lpc) kcode = 11;
    else if (kind == FeaKind::Dctc) kcode = 6;
    else if (kind == FeaKind::Spec) kcode = 8;
    else if (kind == FeaKind::LogSpec) kcode = 7;
    if (o.fea_c0) kcode |= 020000;
    if (o.fea_E) kcode |= 000100;
    if (o.fea_delta && o.n_order >= 1) kcode |= 000400;   // src/io/out.cc:157-159 (-fea_trap sets both fields too)
    if (o.fea_delta && o.n_order >= 2) kcode |= 001000;
    if (o.fea_delta && o.n_order == 3) kcode |= 0100000;
    htk_kind = kcode;
}

}  // namespace ctu
