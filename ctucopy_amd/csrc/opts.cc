// Flag-table implementation of the ctucopy command line (see opts.h for the reference citations).
#include "opts.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace ctu {
namespace {

enum class Kind {
    Str,      // "-x v": copy string; silently ignored when the value is missing (reference: `if(r)`)
    Int,      // atoi
    Dbl,      // atof
    Flt,      // (float)atof
    OnOff,    // "on"/"off"; other words leave the field alone; missing value = syntax error
    Set,      // no value needed: field = true
    Special,  // handled in Opts::apply
};

struct Flag {
    const char *name;
    Kind kind;
    size_t off;
};

#define F(name, kind, field) {name, Kind::kind, offsetof(Opts, field)}
// The order carries no meaning; lookup is by name.
const Flag kFlags[] = {
    F("-S", Str, list),
    F("-i", Str, in),
    F("-o", Str, out),
    F("-format_in", Str, format_in),
    F("-online_in", Set, pipe_in),
    F("-online_out", Set, pipe_out),
    F("-fb_printself", Set, fb_printself),
    F("-preem", Flt, preem),
    F("-fea_Z_exp", Flt, fea_Z_exp),
    F("-fea_Z_block", Flt, fea_Z_block),
    F("-fs", Int, fs),
    F("-dither", Dbl, dither),
    F("-remove_dc", OnOff, remove_dc),
    F("-remove_dc1", OnOff, remove_dc1),
    F("-w", Dbl, window_ms),
    F("-s", Dbl, wshift_ms),
    F("-fb_scale", Str, fb_scale),
    F("-fb_shape", Str, fb_shape),
    F("-fb_norm", OnOff, fb_norm),
    F("-fb_power", OnOff, fb_power),
    F("-fb_eqld", OnOff, fb_eqld),
    F("-fb_inld", OnOff, fb_inld),
    F("-fb_definition", Str, fb_definition),
    F("-nr_mode", Str, nr_mode),
    F("-nr_p", Dbl, nr_p),
    F("-nr_q", Dbl, nr_q),
    F("-nr_a", Dbl, nr_a),
    F("-nr_b", Dbl, nr_b),
    F("-nr_initsegs", Int, nr_initsegs),
    F("-d_win", Int, d_win),
    F("-a_win", Int, a_win),
    F("-t_win", Int, t_win),
    F("-fea_lporder", Int, fea_lporder),
    F("-fea_ncepcoefs", Int, fea_ncepcoefs),
    F("-nfeacoefs", Int, nfeacoefs),
    F("-fea_c0", OnOff, fea_c0),
    F("-fea_E", OnOff, fea_E),
    F("-fea_rawenergy", OnOff, fea_rawenergy),
    F("-weight_of_td_iir_mfcc_bank", Flt, weight_of_td_iir_mfcc_bank),
    F("-fea_lifter", Int, fea_lifter),
    F("-vad_apply_mode", Str, vad_apply_mode),
    F("-vad_out_mode", Str, vad_out_mode),
    F("-vad_out", Str, vad_out),
    F("-vad_cri_mode", Str, vad_cri_mode),
    F("-vad_thr_mode", Str, vad_thr_mode),
    F("-vad_energy_db", OnOff, vad_energy_db),
    F("-vad_cepdist_mode", Str, vad_cepdist_mode),
    F("-vad_cepdist_p", Dbl, vad_cepdist_p),
    F("-vad_cepdist_init", Int, vad_cepdist_init),
    F("-vad_lpc_coefs", Int, vad_lpc_coefs),
    F("-vad_absolute_thr", Dbl, vad_absolute_thr),
    F("-vad_perc_init", Int, vad_perc_init),
    F("-vad_perc_thr", Dbl, vad_perc_thr),
    F("-vad_adapt_init", Int, vad_adapt_init),
    F("-vad_adapt_q", Dbl, vad_adapt_q),
    F("-vad_adapt_za", Dbl, vad_adapt_za),
    F("-vad_dyn_init", Int, vad_dyn_init),
    F("-vad_dyn_perc", Dbl, vad_dyn_perc),
    F("-vad_dyn_min", Dbl, vad_dyn_min),
    F("-vad_dyn_qmaxinc", Dbl, vad_dyn_qmaxinc),
    F("-vad_dyn_qmaxdec", Dbl, vad_dyn_qmaxdec),
    F("-vad_dyn_qmindec", Dbl, vad_dyn_qmindec),
    F("-vad_dyn_qmininc", Dbl, vad_dyn_qmininc),
    F("-vad_filter_order", Int, vad_filter_order),
    F("-C", Str, config),
    {"-format_out", Kind::Special, 0},
    {"-endian_in", Kind::Special, 0},
    {"-endian_out", Kind::Special, 0},
    {"-stat_cmvn", Kind::Special, 0},
    {"-apply_cmvn", Kind::Special, 0},
    {"-fea_delta", Kind::Special, 0},
    {"-fea_trap", Kind::Special, 0},
    {"-filters", Kind::Special, 0},
    {"-vad", Kind::Special, 0},
    {"-nr_rasta", Kind::Special, 0},
    {"-nr_when", Kind::Special, 0},
    {"-fea_kind", Kind::Special, 0},
    {"-preset", Kind::Special, 0},
    {"-verbose", Kind::Special, 0},
    {"-v", Kind::Special, 0},
    {"-quiet", Kind::Special, 0},
    {"-info", Kind::Special, 0},
    {"-h", Kind::Special, 0},
    {"--help", Kind::Special, 0},
};
#undef F

[[noreturn]] void syntax_error(const std::string &flag, const std::string *value) {
    // src/io/opts.cc:840-844
    std::string m = "OPTS: Syntax error in option \"" + flag;
    if (value) m += " " + *value;
    throw OptsError(m + "\".");
}

template <class T>
T &field(Opts *o, size_t off) {
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(o) + off);
}

}  // namespace

void Opts::apply(const std::string &flag, const std::string *value) {
    const Flag *f = nullptr;
    for (const Flag &c : kFlags)
        if (flag == c.name) {
            f = &c;
            break;
        }
    if (!f) syntax_error(flag, value);
    switch (f->kind) {
        case Kind::Str:
            if (value) field<std::string>(this, f->off) = *value;
            return;
        case Kind::Int:
            if (value) field<int>(this, f->off) = std::atoi(value->c_str());
            return;
        case Kind::Dbl:
            if (value) field<double>(this, f->off) = std::atof(value->c_str());
            return;
        case Kind::Flt:
            if (value) field<float>(this, f->off) = (float)std::atof(value->c_str());
            return;
        case Kind::OnOff:
            if (!value) syntax_error(flag, value);
            if (*value == "on") field<bool>(this, f->off) = true;
            else if (*value == "off") field<bool>(this, f->off) = false;
            return;
        case Kind::Set:
            field<bool>(this, f->off) = true;
            return;
        case Kind::Special:
            break;
    }
    const std::string &l = flag;
    auto need = [&]() -> const std::string & {
        if (!value) syntax_error(flag, value);
        return *value;
    };
    if (l == "-format_out") {  // src/io/opts.cc:650-663
        const std::string &r = need();
        size_t eq = r.find('=');
        if (r.find("pfile=") != std::string::npos) {
            pfilename = r.substr(eq + 1);
            format_out = "pfile";
        } else if (r.find("ark=") != std::string::npos) {
            arkfilename = r.substr(eq + 1);
            format_out = "ark";
        } else format_out = r;
    } else if (l == "-endian_in" || l == "-endian_out") {
        const std::string &r = need();
        bool &dst = (l == "-endian_in") ? endian_in_little : endian_out_little;
        if (r == "big") dst = false;
        else if (r == "little") dst = true;
    } else if (l == "-stat_cmvn") {
        fcmvn_stat_out = need();
        stat_cmvn = true;
    } else if (l == "-apply_cmvn") {
        fcmvn_stat_in = need();
        apply_cmvn = true;
    } else if (l == "-fea_delta") {  // src/io/opts.cc:686-693
        const std::string &r = need();
        fea_delta = true;
        fea_trap = false;
        if (r == "d") n_order = 1;
        else if (r == "d_a") n_order = 2;
        else if (r == "d_a_t") n_order = 3;
        else fea_delta = false;
    } else if (l == "-fea_trap") {  // src/io/opts.cc:694-704
        const std::string &r = need();
        if (!fea_delta) {
            fea_trap = true;
            trap_win = std::atoi(r.c_str());
            fea_delta = true;
            n_order = 1;
            d_win = (trap_win - 1) / 2;
        }
    } else if (l == "-filters") {
        ffilters = need();
    } else if (l == "-vad") {  // src/io/opts.cc:740-749
        const std::string &r = need();
        if (r == "burg") vadmode = "burg";
        else if (r.find("file=") != std::string::npos) {
            filevad = r.substr(r.find('=') + 1);
            vadmode = "file";
        } else throw OptsError("OPTS: Syntax error in option -vad !");
    } else if (l == "-nr_rasta") {
        rasta = true;
        nr_rasta = need();
    } else if (l == "-nr_when") {
        const std::string &r = need();
        if (r == "beforeFB") nr_when_afterFB = false;
        else if (r == "afterFB") nr_when_afterFB = true;
    } else if (l == "-fea_kind") {  // src/io/opts.cc:764-780
        if (!value) throw OptsError("OPTS: Missing argument to '-fea_kind' option!");
        const std::string &r = *value;
        if (r.find("trapdct") != std::string::npos) {
            const char *msg = "OPTS: Syntax error in option -fea_kind! (should be -fea_kind trapdct,<X>,<Y>)";
            size_t c1 = r.find(',');
            if (c1 == std::string::npos) throw OptsError(msg);
            size_t c2 = r.find(',', c1 + 1);
            fea_kind = r.substr(0, c1);
            fea_trapdct_traplen = std::atoi(r.c_str() + c1 + 1);
            if (c2 == std::string::npos) throw OptsError(msg);
            fea_trapdct_ndct = std::atoi(r.c_str() + c2 + 1);
        } else fea_kind = r;
    } else if (l == "-preset") {
        if (value) {
            preset = *value;
            set_preset();
        }
    } else if (l == "-verbose" || l == "-v") {
        verbose = true;
        quiet = false;
        info = true;
    } else if (l == "-quiet") {
        quiet = true;
        verbose = false;
        info = false;
    } else if (l == "-info") {
        info = true;
        quiet = false;
    } else if (l == "-h" || l == "--help") {
        help = true;
    }
}

void Opts::set_preset() {  // src/io/opts.cc:196-253
    if (preset == "mfcc") {
        fb_scale = "mel";
        fb_shape = "triang";
        fb_power = true;
        fb_definition = "1-26/26filters";
        nr_mode = "none";
        rasta = false;
        fb_eqld = fb_inld = false;
        fea_kind = "dctc";
        fea_ncepcoefs = 12;
        fea_c0 = true;
        fea_E = false;
        fea_lifter = 22;
        fea_rawenergy = false;
    } else if (preset == "plpc") {
        fb_scale = "bark";
        fb_shape = "trapez";
        fb_power = true;
        fb_definition = "1-15/15filters";
        nr_mode = "none";
        rasta = false;
        fb_eqld = fb_inld = true;
        fea_kind = "lpc";
        fea_lporder = 12;
        fea_ncepcoefs = 12;
        fea_c0 = true;
        fea_E = false;
        fea_lifter = 22;
        fea_rawenergy = false;
    } else if (preset == "exten") {
        window_ms = 32.;
        wshift_ms = 16.;
        fb_definition = fb_scale = fb_shape = "none";
        nr_a = 2.;
        fb_eqld = fb_inld = fb_power = fb_norm = false;
        nr_mode = "exten";
        fea_kind = "none";
        fea_c0 = fea_E = false;
        fea_lifter = 0;
        fea_rawenergy = false;
    } else {
        throw OptsError("OPTS: Unknown preset!");
    }
}

void Opts::check_config() {  // src/io/opts.cc:255-325
    if (fs == 0) throw OptsError("OPTS: Please specify sampling rate!");
    window = (int)std::floor(.5 + window_ms / 1000. * (double)fs);
    wshift = (int)std::floor(.5 + wshift_ms / 1000. * (double)fs);
    // FFT size: the power of two p with window/p == 1, doubled unless the window is exactly p
    wfft = 0;
    for (int p = 1048576; p > 4; p /= 2)
        if (window / p == 1) wfft = p * (1 + (window % p != 0));
    wfftby2 = wfft / 2 + 1;
    if (fea_Z_block != -1) length_b = (int)std::floor((fea_Z_block - window_ms) / wshift_ms) + 1;
    if (fea_Z_exp != -1) fea_Z_exp = (float)1 - (2 * wshift_ms) / fea_Z_exp;
    const bool natural_little = true;  // gfx950 hosts are x86-64
    swap_in = endian_in_little != natural_little;
    swap_out = endian_out_little != natural_little;
    const bool signal_out = (format_out == "raw" || format_out == "wave");
    phase_needed = signal_out || vadmode == "burg";  // "Brutal hack for Burg VAD", src/io/opts.cc:293-294
    if (preem >= 1.0 || preem < 0.0) throw OptsError("OPTS: Preemphasis not in range <0,1)!");
    if ((pipe_in || !in.empty()) != (pipe_out || !out.empty()))
        throw OptsError("OPTS: Single file mode has to be set at both sides (input and output)!");
    if (pipe_out && (format_out == "wave" || format_out == "pfile"))
        throw OptsError("OPTS: Online output available only for raw and htk formats!");
    if (signal_out && fb_power) {
        fb_power = false;
        warn_power_forced_off = true;
    }
}

Opts Opts::from_args(const std::vector<std::string> &args) {
    Opts o;
    if (args.empty()) throw OptsError("OPTS: No command line options!");
    // -C first (src/io/opts.cc:158-182): one option per line, '#' starts a comment
    for (size_t j = 0; j + 1 < args.size(); j++)
        if (args[j] == "-C") {
            o.config = args[j + 1];
            std::ifstream cfg(o.config);
            if (!cfg) throw OptsError("OPTS: Cannot open config file!");
            std::string line;
            while (std::getline(cfg, line)) {
                size_t h = line.find('#');
                if (h != std::string::npos) line.resize(h);
                std::istringstream ss(line);
                std::string l, r;
                if (!(ss >> l)) continue;
                if (ss >> r) o.apply(l, &r);
                else o.apply(l, nullptr);
            }
        }
    // then the command line: a word starting with '-' is a flag, the next word its value unless it
    // starts with '-' too (src/io/opts.cc:185-192)
    for (size_t j = 0; j < args.size(); j++) {
        if (args[j].empty() || args[j][0] != '-') continue;
        if (j + 1 < args.size() && !(args[j + 1].size() && args[j + 1][0] == '-')) o.apply(args[j], &args[j + 1]);
        else o.apply(args[j], nullptr);
    }
    if (!o.help) o.check_config();
    return o;
}

std::string Opts::usage() const {
    return "ctucopy (MI355X engine) -- CtuCopy-compatible speech feature extraction\n"
           "usage: ctucopy -C <config> | <options>  -S <list>   (batch)   or   -i <in> -o <out>\n"
           "  I/O:      -format_in raw|alaw|mulaw|wave  -format_out htk|pfile=<f>|ark=<f>  -fs <Hz>\n"
           "            -endian_in|-endian_out big|little  -preem <0..1)  -remove_dc on|off\n"
           "  framing:  -w <ms>  -s <ms>\n"
           "  bank:     -fb_scale mel|bark|lin|expolog  -fb_shape triang|rect|trapez  -fb_norm|-fb_power|\n"
           "            -fb_eqld|-fb_inld on|off  -fb_definition <[[X-YHz:]K-L/]Nfilters,...>\n"
           "  NR:       -nr_mode none|exten  -nr_p <p>  -nr_a <a>\n"
           "  features: -fea_kind spec|logspec|dctc|lpa|lpc|trapdct,<len>,<ndct>  -fea_lporder <n>\n"
           "            -fea_ncepcoefs <n>  -fea_c0|-fea_E|-fea_rawenergy on|off  -fea_lifter <L>\n"
           "  VAD:      -vad burg  -vad_out_mode none|vad  -vad_apply_mode none|silence|drop  -vad_cri_mode energy|cepdist ...\n"
           "  presets:  -preset mfcc|plpc   (order matters: later options override the preset)\n"
           "  misc:     -v -quiet -info -h --gpus <n>\n";
}

}  // namespace ctu
