// Host-side configuration of the engine: the ctucopy command line.
//
// Mirrors the *interface* of the reference's `class opts` (src/io/opts.h:41-164): same flag names,
// defaults (src/io/opts.cc:31-146), order-dependent `-preset` macro (src/io/opts.cc:196-253,832),
// `-C` config-file syntax (src/io/opts.cc:158-182), derived sizes (src/io/opts.cc:255-325) and error
// texts.  The implementation is a flag table, not a translation of the reference's if-chain.
#pragma once

#include <stdexcept>
#include <string>
#include <vector>

namespace ctu {

struct OptsError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Opts {
    // --- I/O
    std::string format_in, format_out, list, in, out, pfilename, arkfilename;
    float preem = 0.0f;  // float on purpose (src/io/opts.h:47): "-preem 0.97" is 0.9700000286...
    int fs = 0;
    double dither = 0.0;
    bool endian_in_little = true, endian_out_little = true;
    bool pipe_in = false, pipe_out = false;
    bool remove_dc = true, remove_dc1 = false;
    // --- segmentation
    double window_ms = 25., wshift_ms = 10.;
    // --- filter bank
    std::string fb_scale = "mel", fb_shape = "triang", fb_definition = "26filters";
    bool fb_power = true, fb_norm = true, fb_eqld = true, fb_inld = true, fb_printself = false;
    // --- noise reduction
    std::string vadmode = "none", filevad, nr_mode = "none", nr_rasta;
    double nr_p = 0.95, nr_q = 0.99, nr_a = 1., nr_b = 1.;
    int nr_initsegs = 10;
    bool rasta = false, nr_when_afterFB = false;
    // --- parametrisation
    std::string fea_kind = "lpc", ffilters, preset = "user";
    int fea_lporder = 12, fea_ncepcoefs = 12;
    bool fea_c0 = true, fea_E = false, fea_rawenergy = false;
    int fea_lifter = 22;
    int fea_trapdct_traplen = 0, fea_trapdct_ndct = 0;
    float fea_Z_exp = -1, fea_Z_block = -1;  // after check_config fea_Z_exp holds the forgetting factor (src/io/opts.cc:273-274)
    int length_b = 0;                         // frames in the block-CMS window (src/io/opts.cc:270-271)
    bool stat_cmvn = false, apply_cmvn = false;
    std::string fcmvn_stat_out, fcmvn_stat_in;
    int d_win = 2, a_win = 2, t_win = 2;
    bool fea_delta = false, fea_trap = false;
    int n_order = 0, trap_win = 5, nfeacoefs = 13;
    float weight_of_td_iir_mfcc_bank = 2.026f;
    // --- VAD module
    std::string vad_apply_mode = "none", vad_out_mode = "none", vad_out, vad_cri_mode = "energy",
                vad_thr_mode = "perc", vad_cepdist_mode = "lpc";
    bool vad_energy_db = true;
    double vad_cepdist_p = 0.8;
    int vad_cepdist_init = 4, vad_lpc_coefs = 14;
    double vad_absolute_thr = 1.0;
    int vad_perc_init = 10;
    double vad_perc_thr = 50.0;
    int vad_adapt_init = 20;
    double vad_adapt_q = 0.9, vad_adapt_za = 2.0;
    int vad_dyn_init = 5;
    double vad_dyn_perc = 50.0, vad_dyn_min = 1.0, vad_dyn_qmaxinc = 0.8, vad_dyn_qmaxdec = 0.995,
           vad_dyn_qmindec = 0.8, vad_dyn_qmininc = 0.9999;
    int vad_filter_order = 3;
    // --- misc
    bool verbose = false, quiet = false, info = false, help = false;
    std::string config;

    // --- derived by check_config()
    int window = 0, wshift = 0, wfft = 0, wfftby2 = 0;
    bool swap_in = false, swap_out = false, phase_needed = false;
    bool warn_power_forced_off = false;

    // Parses `args` (the command line without argv[0]): the -C file first, then the arguments, then
    // check_config().  Throws OptsError with the reference's message text.
    static Opts from_args(const std::vector<std::string> &args);

    void apply(const std::string &flag, const std::string *value);  // one "-flag [value]" pair
    void set_preset();
    void check_config();
    bool do_vad() const { return vad_apply_mode != "none" || vad_out_mode != "none"; }  // src/io/batch.cc:34-38
    std::string usage() const;
};

}  // namespace ctu
