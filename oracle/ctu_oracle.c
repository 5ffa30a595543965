/*
 * ctu_oracle.c -- float64 CPU restatement of the CtuCopy 4.0.2 per-frame chain.
 *
 * TEST INFRASTRUCTURE ONLY (see ctu_oracle.h for the pin status and the rules
 * on who may load this).  Written from the reference's behaviour, not copied:
 * every function cites the reference lines it restates (relative to
 * /root/reference).  All arithmetic is IEEE double, as in the reference; the
 * (float) cast happens once, when a row is emitted (src/io/out.cc:178-201).
 */
#include "ctu_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAXB 999 /* src/fea/fb.cc:31 "hope 999 filters is enough" */

/* ------------------------------------------------------------------ options */
/* Field names follow class opts (src/io/opts.h:41-164). */
typedef struct {
    char format_in[64], format_out[64], list[1024], in[1024], out[1024];
    char pfilename[1024], arkfilename[1024];
    float preem; /* float on purpose: src/io/opts.h:47 */
    int fs;
    double dither;
    int endian_in_little, endian_out_little, pipe_in, pipe_out;
    int remove_dc, remove_dc1;
    double window_ms, wshift_ms;
    char fb_scale[64], fb_shape[64], fb_definition[1024];
    int fb_power, fb_norm, fb_eqld, fb_inld, fb_printself;
    char vadmode[64], filevad[1024], nr_mode[64];
    double nr_p, nr_q, nr_a, nr_b;
    int nr_initsegs, rasta, nr_when_afterFB;
    char fea_kind[64];
    int fea_lporder, fea_ncepcoefs, fea_c0, fea_E, fea_rawenergy, fea_lifter;
    int fea_trapdct_traplen, fea_trapdct_ndct;
    float fea_Z_exp, fea_Z_block;
    int length_b;
    int stat_cmvn, apply_cmvn, d_win, a_win, t_win, fea_delta, n_order, fea_trap, trap_win, nfeacoefs;
    char preset[64];
    char vad_apply_mode[64], vad_out_mode[64], vad_out[1024], vad_cri_mode[64], vad_thr_mode[64];
    int vad_energy_db;
    char vad_cepdist_mode[64];
    double vad_cepdist_p;
    int vad_cepdist_init, vad_lpc_coefs;
    double vad_absolute_thr;
    int vad_perc_init;
    double vad_perc_thr;
    int vad_adapt_init;
    double vad_adapt_q, vad_adapt_za;
    int vad_dyn_init;
    double vad_dyn_perc, vad_dyn_min, vad_dyn_qmaxinc, vad_dyn_qmaxdec, vad_dyn_qmindec, vad_dyn_qmininc;
    int vad_filter_order;
    int verbose, quiet, info;
    char config[1024];
    /* derived (check_config) */
    int window, wshift, wfft, wfftby2, swap_in, swap_out, phase_needed;
} opts_t;

struct ctuo {
    opts_t o;
    char err[256];
    /* tables */
    double *W;          /* Hamming[window] */
    int B;              /* bank size */
    double **mat;       /* B rows x (K+2); last two columns as in fb.cc:432-447 */
    double *warp, *hz;
    double *wdct;       /* 4*B          (fea_impl.cc:92-93) */
    double *lift;       /* ncep         (fea_impl.cc:96-97,247-248) */
    double normcoef;
    double *WRe;        /* 2*(B-1)      (fea_impl.cc:157) */
    double *trap_hamm;  /* traplen      (fea_trap.cc:42-43) */
    /* fft plan (power of two) */
    int fftn;
    double *tw_re, *tw_im; /* n/2 twiddles of the half-size complex FFT */
    double *ut_re, *ut_im; /* untangle twiddles exp(-2 pi i k / n), k<=n/4.. */
    /* geometry */
    int nfea, D, htk_kind;
    int signal_out;     /* -format_out raw|wave: enhanced speech instead of features (sigOUT, out.cc:346-451) */
    double ola_corr;    /* OLA window-sum maximum (out.cc:355-377) */
    int16_t *sig_buf;   /* destination of the current ctuo_enhance call */
    int post_order;     /* 0 = none, 1..3 = delta stages chained after FEA (batch.cc:122-130) */
    int post_stack;     /* -fea_trap: one stage that stacks 2*d_win+1 frames (fea_delta.cc:166-176) */
    int Xsize;          /* size of the vector OUT sees: nfea, fea_c*(n_order+1) or fea_c*(2*d_win+1) */
    unsigned period;
    int do_vad;
    /* debug taps */
    double *last_power, *last_fbank;
    /* hwss / fwss / 2fwss: what survives from file to file.  new_file() seeds the noise estimate from the spectrum vector
     * as the previous file left it (src/nr/nr.cc:212-221,402-409; all zeros before the first file, base/types.h:35-38). */
    double *ss_stale;   /* K: _Xsabs after the last process_frame */
    int ss_mode;        /* 0 none, 1 hwss, 2 fwss, 3 2fwss */
    /* -vad file=<f> (nr.cc:205-209,297-302): ONE byte stream for all files of the process, one byte per frame, opened by the
     * constructor and never rewound ("no separation of files in VAD file (yes, it is dangerous, but...)", nr.cc:273) */
    unsigned char *vad_stream;
    long vad_len, vad_pos;
    int vad_from_file;
    /* The VAD's majority filter lives as long as the process (one VAD object per BATCH): VAD::clean() = cleanFilter() resets `start`,
     * `ready`, the ring and the decisions at the end of a file but NOT historyIdx / historySize (src/vad/vad.h:110-121).  The next
     * file's pushes therefore land in ring slots that are out of phase with the slots its outputs are read from: its rows come out
     * shifted, with an all-zero or a repeated row (pinned against the reference's own class, tests/test_oracle_median_ref.py). */
    int med_hidx, med_hsize;
};

static void set_err(ctuo_t *c, const char *msg) {
    snprintf(c->err, sizeof c->err, "%s", msg);
}

/* src/io/opts.cc:31-146 */
static void opts_defaults(opts_t *o) {
    memset(o, 0, sizeof *o);
    o->preem = 0.0f;
    o->fs = 0;
    o->dither = 0.0;
    o->endian_in_little = 1;
    o->endian_out_little = 1;
    o->remove_dc = 1;
    o->remove_dc1 = 0;
    o->window_ms = 25.;
    o->wshift_ms = 10.;
    strcpy(o->fb_scale, "mel");
    strcpy(o->fb_shape, "triang");
    o->fb_power = 1;
    o->fb_norm = 1;
    o->fb_eqld = 1;
    o->fb_inld = 1;
    strcpy(o->fb_definition, "26filters");
    strcpy(o->vadmode, "none");
    strcpy(o->nr_mode, "none");
    o->nr_p = 0.95;
    o->nr_q = 0.99;
    o->nr_a = 1.;
    o->nr_b = 1.;
    o->nr_initsegs = 10;
    o->nr_when_afterFB = 0;
    strcpy(o->fea_kind, "lpc");
    o->fea_lporder = 12;
    o->fea_ncepcoefs = 12;
    o->fea_c0 = 1;
    o->fea_E = 0;
    o->fea_Z_exp = -1;
    o->fea_Z_block = -1;
    o->d_win = o->a_win = o->t_win = 2;
    o->trap_win = 5;
    o->nfeacoefs = 13;
    o->fea_rawenergy = 0;
    o->fea_lifter = 22;
    strcpy(o->preset, "user");
    strcpy(o->vad_apply_mode, "none");
    strcpy(o->vad_out_mode, "none");
    strcpy(o->vad_cri_mode, "energy");
    strcpy(o->vad_thr_mode, "perc");
    o->vad_energy_db = 1;
    strcpy(o->vad_cepdist_mode, "lpc");
    o->vad_cepdist_p = 0.8;
    o->vad_cepdist_init = 4;
    o->vad_lpc_coefs = 14;
    o->vad_absolute_thr = 1.0;
    o->vad_perc_init = 10;
    o->vad_perc_thr = 50.0;
    o->vad_adapt_init = 20;
    o->vad_adapt_q = 0.9;
    o->vad_adapt_za = 2.0;
    o->vad_dyn_init = 5;
    o->vad_dyn_perc = 50.0;
    o->vad_dyn_min = 1.0;
    o->vad_dyn_qmaxinc = 0.8;
    o->vad_dyn_qmaxdec = 0.995;
    o->vad_dyn_qmindec = 0.8;
    o->vad_dyn_qmininc = 0.9999;
    o->vad_filter_order = 3;
}

/* src/io/opts.cc:196-253 */
static int opts_set_preset(ctuo_t *c) {
    opts_t *o = &c->o;
    if (!strcmp(o->preset, "mfcc")) {
        strcpy(o->fb_scale, "mel");
        strcpy(o->fb_shape, "triang");
        o->fb_power = 1;
        strcpy(o->fb_definition, "1-26/26filters");
        strcpy(o->nr_mode, "none");
        o->rasta = 0;
        o->fb_eqld = 0;
        o->fb_inld = 0;
        strcpy(o->fea_kind, "dctc");
        o->fea_ncepcoefs = 12;
        o->fea_c0 = 1;
        o->fea_E = 0;
        o->fea_lifter = 22;
        o->fea_rawenergy = 0;
    } else if (!strcmp(o->preset, "plpc")) {
        strcpy(o->fb_scale, "bark");
        strcpy(o->fb_shape, "trapez");
        o->fb_power = 1;
        strcpy(o->fb_definition, "1-15/15filters");
        strcpy(o->nr_mode, "none");
        o->rasta = 0;
        o->fb_eqld = 1;
        o->fb_inld = 1;
        strcpy(o->fea_kind, "lpc");
        o->fea_lporder = 12;
        o->fea_ncepcoefs = 12;
        o->fea_c0 = 1;
        o->fea_E = 0;
        o->fea_lifter = 22;
        o->fea_rawenergy = 0;
    } else if (!strcmp(o->preset, "exten")) {
        o->window_ms = 32.;
        o->wshift_ms = 16.;
        strcpy(o->fb_definition, "none");
        strcpy(o->fb_scale, "none");
        strcpy(o->fb_shape, "none");
        o->nr_a = 2.;
        o->fb_eqld = 0;
        o->fb_inld = 0;
        o->fb_power = 0;
        o->fb_norm = 0;
        strcpy(o->nr_mode, "exten");
        strcpy(o->fea_kind, "none");
        o->fea_c0 = 0;
        o->fea_E = 0;
        o->fea_lifter = 0;
        o->fea_rawenergy = 0;
    } else {
        set_err(c, "OPTS: Unknown preset!");
        return -1;
    }
    return 0;
}

static int onoff(const char *r, int *dst) {
    if (!strcmp(r, "on")) *dst = 1;
    else if (!strcmp(r, "off")) *dst = 0;
    return 0;
}

/* src/io/opts.cc:644-846.  r may be NULL. */
static int opts_parse(ctuo_t *c, const char *l, const char *r_in) {
    opts_t *o = &c->o;
    char rbuf[2048];
    char *r = NULL;
    if (r_in) {
        snprintf(rbuf, sizeof rbuf, "%s", r_in);
        r = rbuf;
    }
#define S(flag, field) else if (!strcmp(l, flag)) { if (r) snprintf(o->field, sizeof o->field, "%s", r); }
#define I(flag, field) else if (!strcmp(l, flag)) { if (r) o->field = atoi(r); }
#define Dbl(flag, field) else if (!strcmp(l, flag)) { if (r) o->field = atof(r); }
#define OO(flag, field) else if (!strcmp(l, flag) && r) { onoff(r, &o->field); }
    if (!strcmp(l, "-S")) { if (r) snprintf(o->list, sizeof o->list, "%s", r); }
    S("-i", in)
    S("-o", out)
    S("-format_in", format_in)
    else if (!strcmp(l, "-format_out") && r) {
        char *eq;
        if (strstr(r, "pfile=") != NULL) {
            eq = strchr(r, '=');
            snprintf(o->pfilename, sizeof o->pfilename, "%s", eq + 1);
            strcpy(o->format_out, "pfile");
        } else if (strstr(r, "ark=") != NULL) {
            eq = strchr(r, '=');
            snprintf(o->arkfilename, sizeof o->arkfilename, "%s", eq + 1);
            strcpy(o->format_out, "ark");
        } else snprintf(o->format_out, sizeof o->format_out, "%s", r);
    }
    else if (!strcmp(l, "-endian_in") && r) {
        if (!strcmp(r, "big")) o->endian_in_little = 0;
        else if (!strcmp(r, "little")) o->endian_in_little = 1;
    }
    else if (!strcmp(l, "-endian_out") && r) {
        if (!strcmp(r, "big")) o->endian_out_little = 0;
        else if (!strcmp(r, "little")) o->endian_out_little = 1;
    }
    else if (!strcmp(l, "-online_in")) o->pipe_in = 1;
    else if (!strcmp(l, "-online_out")) o->pipe_out = 1;
    else if (!strcmp(l, "-fb_printself")) o->fb_printself = 1;
    else if (!strcmp(l, "-preem")) { if (r) o->preem = (float)atof(r); }
    else if (!strcmp(l, "-fea_Z_exp")) { if (r) o->fea_Z_exp = (float)atof(r); }
    else if (!strcmp(l, "-fea_Z_block")) { if (r) o->fea_Z_block = (float)atof(r); }
    else if (!strcmp(l, "-stat_cmvn") && r) o->stat_cmvn = 1;
    else if (!strcmp(l, "-apply_cmvn") && r) o->apply_cmvn = 1;
    else if (!strcmp(l, "-fea_delta") && r) {
        o->fea_delta = 1;
        o->fea_trap = 0;
        if (!strcmp(r, "d")) o->n_order = 1;
        else if (!strcmp(r, "d_a")) o->n_order = 2;
        else if (!strcmp(r, "d_a_t")) o->n_order = 3;
        else o->fea_delta = 0;
    }
    else if (!strcmp(l, "-fea_trap") && r) {
        if (!o->fea_delta) {
            o->fea_trap = 1;
            o->trap_win = atoi(r);
            o->fea_delta = 1;
            o->n_order = 1;
            o->d_win = (o->trap_win - 1) / 2;
        }
    }
    else if (!strcmp(l, "-filters") && r) { /* td-iir-mfcc coefficient file: out of scope */ }
    I("-fs", fs)
    Dbl("-dither", dither)
    OO("-remove_dc", remove_dc)
    OO("-remove_dc1", remove_dc1)
    Dbl("-w", window_ms)
    Dbl("-s", wshift_ms)
    S("-fb_scale", fb_scale)
    S("-fb_shape", fb_shape)
    OO("-fb_norm", fb_norm)
    OO("-fb_power", fb_power)
    OO("-fb_eqld", fb_eqld)
    OO("-fb_inld", fb_inld)
    S("-fb_definition", fb_definition)
    else if (!strcmp(l, "-vad") && r) {
        if (!strcmp(r, "burg")) strcpy(o->vadmode, "burg");
        else if (strstr(r, "file=") != NULL) {
            char *eq = strchr(r, '=');
            snprintf(o->filevad, sizeof o->filevad, "%s", eq + 1);
            strcpy(o->vadmode, "file");
        } else { set_err(c, "OPTS: Syntax error in option -vad !"); return -1; }
    }
    S("-nr_mode", nr_mode)
    Dbl("-nr_p", nr_p)
    Dbl("-nr_q", nr_q)
    Dbl("-nr_a", nr_a)
    Dbl("-nr_b", nr_b)
    I("-nr_initsegs", nr_initsegs)
    else if (!strcmp(l, "-nr_rasta")) { o->rasta = 1; }
    else if (!strcmp(l, "-nr_when") && r) {
        if (!strcmp(r, "beforeFB")) o->nr_when_afterFB = 0;
        else if (!strcmp(r, "afterFB")) o->nr_when_afterFB = 1;
    }
    else if (!strcmp(l, "-fea_kind")) {
        if (!r) { set_err(c, "OPTS: Missing argument to '-fea_kind' option!"); return -1; }
        if (strstr(r, "trapdct") != NULL) {
            char *next = strchr(r, ',');
            if (!next) { set_err(c, "OPTS: Syntax error in option -fea_kind! (should be -fea_kind trapdct,<X>,<Y>)"); return -1; }
            *next = 0;
            snprintf(o->fea_kind, sizeof o->fea_kind, "%s", r);
            o->fea_trapdct_traplen = atoi(next + 1);
            next = strchr(next + 1, ',');
            if (!next) { set_err(c, "OPTS: Syntax error in option -fea_kind! (should be -fea_kind trapdct,<X>,<Y>)"); return -1; }
            o->fea_trapdct_ndct = atoi(next + 1);
        } else snprintf(o->fea_kind, sizeof o->fea_kind, "%s", r);
    }
    I("-d_win", d_win)
    I("-a_win", a_win)
    I("-t_win", t_win)
    I("-fea_lporder", fea_lporder)
    I("-fea_ncepcoefs", fea_ncepcoefs)
    I("-nfeacoefs", nfeacoefs)
    OO("-fea_c0", fea_c0)
    OO("-fea_E", fea_E)
    OO("-fea_rawenergy", fea_rawenergy)
    else if (!strcmp(l, "-weight_of_td_iir_mfcc_bank")) { }
    I("-fea_lifter", fea_lifter)
    S("-vad_apply_mode", vad_apply_mode)
    S("-vad_out_mode", vad_out_mode)
    S("-vad_out", vad_out)
    S("-vad_cri_mode", vad_cri_mode)
    S("-vad_thr_mode", vad_thr_mode)
    OO("-vad_energy_db", vad_energy_db)
    S("-vad_cepdist_mode", vad_cepdist_mode)
    Dbl("-vad_cepdist_p", vad_cepdist_p)
    I("-vad_cepdist_init", vad_cepdist_init)
    I("-vad_lpc_coefs", vad_lpc_coefs)
    Dbl("-vad_absolute_thr", vad_absolute_thr)
    I("-vad_perc_init", vad_perc_init)
    Dbl("-vad_perc_thr", vad_perc_thr)
    I("-vad_adapt_init", vad_adapt_init)
    Dbl("-vad_adapt_q", vad_adapt_q)
    Dbl("-vad_adapt_za", vad_adapt_za)
    I("-vad_dyn_init", vad_dyn_init)
    Dbl("-vad_dyn_perc", vad_dyn_perc)
    Dbl("-vad_dyn_min", vad_dyn_min)
    Dbl("-vad_dyn_qmaxinc", vad_dyn_qmaxinc)
    Dbl("-vad_dyn_qmaxdec", vad_dyn_qmaxdec)
    Dbl("-vad_dyn_qmindec", vad_dyn_qmindec)
    Dbl("-vad_dyn_qmininc", vad_dyn_qmininc)
    I("-vad_filter_order", vad_filter_order)
    else if (!strcmp(l, "-preset")) {
        if (r) {
            snprintf(o->preset, sizeof o->preset, "%s", r);
            if (opts_set_preset(c)) return -1;
        }
    }
    else if (!strcmp(l, "-verbose") || !strcmp(l, "-v")) { o->verbose = 1; o->quiet = 0; o->info = 1; }
    else if (!strcmp(l, "-quiet")) { o->quiet = 1; o->verbose = 0; o->info = 0; }
    else if (!strcmp(l, "-info")) { o->info = 1; o->quiet = 0; }
    S("-C", config)
    else {
        snprintf(c->err, sizeof c->err, "OPTS: Syntax error in option \"%s%s%s\".", l, r ? " " : "", r ? r : "");
        return -1;
    }
#undef S
#undef I
#undef Dbl
#undef OO
    return 0;
}

/* src/io/opts.cc:255-325 */
static int opts_check_config(ctuo_t *c) {
    opts_t *o = &c->o;
    if (o->fs == 0) { set_err(c, "OPTS: Please specify sampling rate!"); return -1; }
    o->window = (int)floor(.5 + o->window_ms / 1000. * (double)o->fs);
    o->wshift = (int)floor(.5 + o->wshift_ms / 1000. * (double)o->fs);
    o->wfft = 0;
    for (int i = 1048576; i > 4; i /= 2)
        if ((o->window / i) == 1) o->wfft = i * (1 + ((o->window % i) != 0));
    o->wfftby2 = o->wfft / 2 + 1;
    /* CMS constants, src/io/opts.cc:270-274 (the float field is overwritten by the forgetting factor) */
    if (o->fea_Z_block != -1) o->length_b = (int)floor((o->fea_Z_block - o->window_ms) / o->wshift_ms) + 1;
    if (o->fea_Z_exp != -1) o->fea_Z_exp = (float)1 - (2 * o->wshift_ms) / o->fea_Z_exp;
    /* natural_little is true on every target of this build */
    o->swap_in = (o->endian_in_little == 1) ^ 1;
    o->swap_out = (o->endian_out_little == 1) ^ 1;
    o->phase_needed = (!strcmp(o->format_out, "raw") || !strcmp(o->format_out, "wave"));
    if (!strcmp(o->vadmode, "burg")) o->phase_needed = 1;
    if (o->preem >= 1.0 || o->preem < 0.0) { set_err(c, "OPTS: Preemphasis not in range <0,1)!"); return -1; }
    if ((!strcmp(o->format_out, "raw") || !strcmp(o->format_out, "wave")) && o->fb_power) o->fb_power = 0;
    return 0;
}

/* ------------------------------------------------------------------ filter bank */
static double warp_of(const char *scale, double f) {
    /* src/fea/fb.cc:105-131, 313-337 (the same formulas are used for the axis and for the edges) */
    if (!strcmp(scale, "lin")) return f;
    if (!strcmp(scale, "bark")) return 6. * log(f / 600. + sqrt((f / 600.) * (f / 600.) + 1.));
    if (!strcmp(scale, "expolog")) {
        if (f <= 2000) return 700. * (pow(10., f / 3988.) - 1.);
        return 2595. * log10(1. + f / 700.);
    }
    /* mel */
    return 2595 * log10(1. + f / 700.);
}

static double eqloud_of(double om, int fs) {
    /* src/fea/fb.cc:157-164, 376-383 */
    double eqnum = om * om * om * om * (om * om + 5.68e7);
    double eqden;
    if (fs <= 10000) eqden = (om * om + 6.3e6) * (om * om + 6.3e6) * (om * om + 3.8e8);
    else eqden = (om * om + 6.3e6) * (om * om + 6.3e6) * (om * om + 3.8e8) * (om * om * om * om * om * om + 9.58e26);
    return eqnum / eqden;
}

typedef struct { double f_start, f_stop; int bands, band_first, band_last; } subbank_t;

/* src/fea/fb.cc:186-253 */
static int fb_parse(ctuo_t *c, subbank_t *bank, int *nsub) {
    char copy[1024];
    snprintf(copy, sizeof copy, "%s", c->o.fb_definition);
    int nfilt = 0;
    *nsub = 0;
    for (char *tok = strtok(copy, ","); tok; tok = strtok(NULL, ",")) {
        int off = 0;
        double flow = 0., fhigh = c->o.fs / 2.;
        int start = 1, stop = 0;
        double x = atof(tok + off);
        off += (int)strspn(tok + off, ".1234567890");
        if (!strncmp(tok + off, "-", 1)) {
            off++;
            double y = atof(tok + off);
            off += (int)strspn(tok + off, ".1234567890");
            if (!strncmp(tok + off, "Hz:", 3)) {
                off += 3;
                flow = x;
                fhigh = y;
                start = atoi(tok + off);
                off += (int)strspn(tok + off, "1234567890");
                if (strncmp(tok + off, "-", 1)) goto bad;
                stop = atoi(tok + ++off);
                off += (int)strspn(tok + off, "1234567890");
                if (strncmp(tok + off, "/", 1)) goto bad;
                nfilt = atoi(tok + ++off);
                off += (int)strspn(tok + off, "1234567890");
                if (strncmp(tok + off, "filters", 7)) goto bad;
            } else if (!strncmp(tok + off, "/", 1)) {
                start = (int)x;
                stop = (int)y;
                off++;
                nfilt = atoi(tok + off);
                off += (int)strspn(tok + off, "1234567890");
                if (strncmp(tok + off, "filters", 7)) goto bad;
            } else goto bad;
        } else if (!strncmp(tok + off, "filters", 7)) {
            stop = (int)x;
            nfilt = stop;
        } else goto bad;
        if (*nsub >= MAXB) goto bad;
        bank[*nsub].f_start = flow;
        bank[*nsub].f_stop = fhigh;
        bank[*nsub].bands = nfilt;
        bank[*nsub].band_first = start;
        bank[*nsub].band_last = stop;
        (*nsub)++;
    }
    return 0;
bad:
    set_err(c, "FB: Filter bank specification parse error!");
    return -1;
}

/* src/fea/fb.cc:306-429 */
static int fb_get_filter(ctuo_t *c, double *vec, double f_start, double f_stop, int bands, int band_index) {
    const opts_t *o = &c->o;
    int K = o->wfftby2;
    if (strcmp(o->fb_scale, "lin") && strcmp(o->fb_scale, "bark") && strcmp(o->fb_scale, "expolog") && strcmp(o->fb_scale, "mel")) {
        set_err(c, "FB: Unknown frequency scale!");
        return -1;
    }
    double w_high = warp_of(o->fb_scale, f_stop), w_low = warp_of(o->fb_scale, f_start);
    double w_start, w_end;
    int rect = !strcmp(o->fb_shape, "rect");
    if (rect) {
        w_start = w_low + (band_index - 1.) * (w_high - w_low) / (double)(bands);
        w_end = w_low + (band_index + 0.) * (w_high - w_low) / (double)(bands);
    } else if (!strcmp(o->fb_shape, "triang")) {
        w_start = w_low + (band_index - 1.) * (w_high - w_low) / (double)(bands + 1);
        w_end = w_low + (band_index + 1.) * (w_high - w_low) / (double)(bands + 1);
    } else { set_err(c, "FB: Unknown filter shape!"); return -1; }

    double eqloud = 1.;
    if (o->fb_eqld) {
        double w_mid = w_start + (w_end - w_start) / 2.;
        double f_mid = 0;
        if (!strcmp(o->fb_scale, "lin")) f_mid = w_mid;
        if (!strcmp(o->fb_scale, "bark")) f_mid = 600 * sinh(w_mid / 6.);
        if (!strcmp(o->fb_scale, "expolog")) {
            if (w_mid <= 1521.4) f_mid = 3988. * log10(1. + (w_mid / 700.));
            else f_mid = 700. * (pow(10., w_mid / 2595.) - 1);
        }
        if (!strcmp(o->fb_scale, "mel")) f_mid = 700. * (pow(10., w_mid / 2595.) - 1.);
        double om = 2 * 3.141592653589793 * f_mid;
        eqloud = eqloud_of(om, o->fs);
    }
    double area = 0;
    if (rect) {
        for (int i = 0; i < K; i++) {
            if (c->warp[i] >= w_start && c->warp[i] < w_end) { vec[i] = 1; area++; }
            else vec[i] = 0;
        }
    } else {
        for (int i = 0; i < K; i++) {
            if (c->warp[i] < w_start || c->warp[i] > w_end) vec[i] = 0;
            else {
                double w_mid = w_start + (w_end - w_start) / 2.;
                vec[i] = 1. - 2. * fabs(w_mid - c->warp[i]) / (w_end - w_start);
                area += vec[i];
            }
        }
    }
    if (o->fb_norm) for (int i = 0; i < K; i++) vec[i] *= eqloud / area;
    else for (int i = 0; i < K; i++) vec[i] *= eqloud;
    return 0;
}

/* src/fea/fb.cc:20-66, 100-184, 255-303, 432-447 */
static int fb_design(ctuo_t *c) {
    opts_t *o = &c->o;
    int K = o->wfftby2;
    int PLP = !strcmp(o->fb_shape, "trapez");
    c->hz = calloc(K, sizeof(double));
    c->warp = calloc(K, sizeof(double));
    c->mat = calloc(MAXB, sizeof(double *));
    if (PLP) { /* fb.cc:44-54 */
        strcpy(o->fb_scale, "bark");
        o->fb_inld = 1;
        o->fb_eqld = 1;
    }
    for (int i = 0; i < K; i++) c->hz[i] = (double)i * o->fs / (double)o->wfft;
    if (!strcmp(o->fb_scale, "lin") || !strcmp(o->fb_scale, "bark") || !strcmp(o->fb_scale, "expolog") || !strcmp(o->fb_scale, "mel"))
        for (int i = 0; i < K; i++) c->warp[i] = warp_of(o->fb_scale, c->hz[i]);
    /* else: axis stays zero, as in fb.cc:100-132 (no branch taken) */
    int size = 0;
    if (PLP) { /* fb.cc:134-184 */
        double maxBark = 6 * log(o->fs / 1200. + sqrt((o->fs / 1200.) * (o->fs / 1200.) + 1.));
        int nBark = (int)(floor(maxBark + .5));
        int plpsize = nBark - 1;
        double Barkstep = maxBark / (double)nBark;
        for (int i = 0; i < plpsize; i++) {
            c->mat[i] = calloc(K + 2, sizeof(double));
            double Om = (i + 1) * Barkstep;
            double om = 3.1415926535898 * 1200 * sinh(Om / 6);
            double eqloud = eqloud_of(om, o->fs);
            for (int k = 0; k < K; k++) {
                double diff = c->warp[k] - Om, v;
                if (diff >= -1.3 && diff <= -.5) v = pow(10., 2.5 * (0.5 + diff));
                else if (fabs(diff) < 0.5) v = 1;
                else if (diff >= 0.5 && diff <= 2.5) v = pow(10., 0.5 - diff);
                else v = 0;
                if (o->fb_eqld) v *= eqloud;
                c->mat[i][k] = v;
            }
        }
        size = plpsize;
    } else {
        subbank_t *bank = calloc(MAXB, sizeof *bank);
        int nsub = 0;
        if (fb_parse(c, bank, &nsub)) { free(bank); return -1; }
        if (!strcmp(o->fb_shape, "rect")) { /* fb.cc:257-279 */
            double df = o->fs / (double)o->wfft;
            for (int i = 0; i < nsub; i++) {
                int no_join = 1;
                for (int j = 0; j < nsub; j++) if (bank[i].f_stop == bank[j].f_start) no_join = 0;
                if (no_join) bank[i].f_stop += df;
            }
        }
        for (int sb = 0; sb < nsub; sb++)
            for (int b = bank[sb].band_first; b <= bank[sb].band_last; b++) {
                if (size >= MAXB - 1) { free(bank); set_err(c, "FB: Too many filters in FB!"); return -1; }
                c->mat[size] = calloc(K + 2, sizeof(double));
                if (fb_get_filter(c, c->mat[size], bank[sb].f_start, bank[sb].f_stop, bank[sb].bands, b)) { free(bank); return -1; }
                size++;
            }
        free(bank);
    }
    if (size < 1) { set_err(c, "FB: empty filter bank"); return -1; }
    /* fb.cc:432-447 -- first / last non-zero element; an all-zero row runs off the
     * end of the array in the reference (undefined), so it is rejected here. */
    for (int b = 0; b < size; b++) {
        int k = 0;
        /* a filter that covers no bin has area 0; with -fb_norm the reference divides by it (0/0): every weight of the
         * row is NaN and so is every feature.  Rejected like the all-zero row. */
        for (int i = 0; i < K; i++)
            if (c->mat[b][i] != c->mat[b][i]) { set_err(c, "FB: filter with no spectral bin (NaN weights in the reference)"); return -1; }
        while (k < K && c->mat[b][k] == 0) k++;
        if (k == K) { set_err(c, "FB: filter with no spectral bin (undefined in the reference)"); return -1; }
        c->mat[b][K + 1] = k;
        k++;
        while (k < K && c->mat[b][k] != 0) k++;
        c->mat[b][K] = k - 1;
    }
    c->B = size;
    return 0;
}

/* ------------------------------------------------------------------ FFTW r2r restatements */
/* In-place iterative radix-2 complex FFT, sign = -1 forward / +1 inverse, unnormalised. */
static void cfft(double *re, double *im, int n, int sign) {
    for (int i = 1, j = 0; i < n; i++) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    const double pi = 3.14159265358979323846;
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        for (int k = 0; k < half; k++) {
            double ang = sign * 2.0 * pi * k / len;
            double wr = cos(ang), wi = sin(ang);
            for (int i = k; i < n; i += len) {
                int j = i + half;
                double xr = re[j] * wr - im[j] * wi, xi = re[j] * wi + im[j] * wr;
                re[j] = re[i] - xr; im[j] = im[i] - xi;
                re[i] += xr; im[i] += xi;
            }
        }
    }
}

/* Plan for FFTW_R2HC of size n (power of two): half-size complex FFT + untangle. */
static void r2hc_plan(ctuo_t *c, int n) {
    const double pi = 3.14159265358979323846;
    int h = n / 2;
    c->fftn = n;
    c->tw_re = malloc(sizeof(double) * (h / 2 + 1));
    c->tw_im = malloc(sizeof(double) * (h / 2 + 1));
    for (int k = 0; k <= h / 2; k++) { c->tw_re[k] = cos(-2 * pi * k / h); c->tw_im[k] = sin(-2 * pi * k / h); }
    c->ut_re = malloc(sizeof(double) * (h + 1));
    c->ut_im = malloc(sizeof(double) * (h + 1));
    for (int k = 0; k <= h; k++) { c->ut_re[k] = cos(-2 * pi * k / n); c->ut_im[k] = sin(-2 * pi * k / n); }
}

/* Forward real DFT: Xre[k], Xim[k], k=0..n/2, X_k = sum_j x_j exp(-2 pi i j k / n)
 * (FFTW_R2HC semantics: out[k]=Re X_k, out[n-k]=Im X_k; src/io/in.cc:229,388). */
static void r2hc(const ctuo_t *c, const double *x, double *Xre, double *Xim, double *zr, double *zi) {
    int n = c->fftn, h = n / 2;
    for (int j = 0; j < h; j++) { zr[j] = x[2 * j]; zi[j] = x[2 * j + 1]; }
    /* half-size FFT with table twiddles (decimation in time, bit reversal first) */
    for (int i = 1, j = 0; i < h; i++) {
        int bit = h >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { double t = zr[i]; zr[i] = zr[j]; zr[j] = t; t = zi[i]; zi[i] = zi[j]; zi[j] = t; }
    }
    for (int len = 2; len <= h; len <<= 1) {
        int half = len >> 1, step = h / len;
        for (int k = 0; k < half; k++) {
            double wr = c->tw_re[k * step], wi = c->tw_im[k * step];
            for (int i = k; i < h; i += len) {
                int j = i + half;
                double xr = zr[j] * wr - zi[j] * wi, xi = zr[j] * wi + zi[j] * wr;
                zr[j] = zr[i] - xr; zi[j] = zi[i] - xi;
                zr[i] += xr; zi[i] += xi;
            }
        }
    }
    for (int k = 0; k <= h; k++) {
        int k1 = k % h, k2 = (h - k) % h;
        double ar = zr[k1], ai = zi[k1], br = zr[k2], bi = -zi[k2]; /* b = conj Z[h-k] */
        double er = 0.5 * (ar + br), ei = 0.5 * (ai + bi);          /* even part  */
        double dr = 0.5 * (ar - br), di = 0.5 * (ai - bi);          /* (Z - conj Z')/2 */
        /* odd part = -i * d ; X = e + w^k * (-i d) */
        double orr = di, oi = -dr;
        double wr = c->ut_re[k], wi = c->ut_im[k];
        Xre[k] = er + (orr * wr - oi * wi);
        Xim[k] = ei + (orr * wi + oi * wr);
    }
    Xim[0] = 0.0;
    Xim[h] = 0.0;
}

/* FFTW_HC2R, unnormalised: x_j = sum_k X_k exp(+2 pi i j k / n) for the Hermitian
 * extension of (re[k], im[k]), k=0..n/2 (src/vad/vad.cc:176,232; src/nr/nr.cc:199). */
static void hc2r(const double *re, const double *im, int n, double *x, double *wr, double *wi) {
    int h = n / 2;
    wr[0] = re[0]; wi[0] = 0.0;
    wr[h] = re[h]; wi[h] = 0.0;
    for (int k = 1; k < h; k++) { wr[k] = re[k]; wi[k] = im[k]; wr[n - k] = re[k]; wi[n - k] = -im[k]; }
    cfft(wr, wi, n, +1);
    for (int j = 0; j < n; j++) x[j] = wr[j];
}

/* ------------------------------------------------------------------ create */
static int design_all(ctuo_t *c) {
    opts_t *o = &c->o;
    if (o->dither != 0.) { set_err(c, "oracle: -dither != 0 makes outputs depend on file order (src/io/in.cc:205,454); not restated"); return -1; }
    if (o->wfft < 8) { set_err(c, "oracle: window too short"); return -1; }
    c->signal_out = (!strcmp(o->format_out, "raw") || !strcmp(o->format_out, "wave"));
    if (!strcmp(o->format_in, "htk") || !strcmp(o->fea_kind, "td-iir-mfcc") || (!strcmp(o->fea_kind, "none") && !c->signal_out)) {
        set_err(c, "oracle: only the spectral paths (raw PCM in; features or enhanced speech out) are restated");
        return -1;
    }
    c->ss_mode = !strcmp(o->nr_mode, "hwss") ? 1 : !strcmp(o->nr_mode, "fwss") ? 2 : !strcmp(o->nr_mode, "2fwss") ? 3 : 0;
    if (strcmp(o->nr_mode, "none") && strcmp(o->nr_mode, "exten") && !c->ss_mode) { set_err(c, "NR: Unknown noise reduction mode!"); return -1; }
    if (c->ss_mode) { /* src/nr/nr.cc:181-442 */
        if (!strcmp(o->vadmode, "none")) { set_err(c, "NR: Please specify Voice Activity Detector!"); return -1; } /* nr.cc:271 */
        if (!strcmp(o->vadmode, "file")) { /* hwssNR::hwssNR, nr.cc:205-209: fopen(filevad, "r") */
            FILE *f = fopen(o->filevad, "rb");
            if (!f) { set_err(c, "NR: Unable to open VAD file!\n"); return -1; }
            long cap = 1 << 16, n = 0;
            unsigned char *buf = malloc((size_t)cap);
            size_t got;
            while ((got = fread(buf + n, 1, (size_t)(cap - n), f)) > 0) {
                n += (long)got;
                if (n == cap) { cap *= 2; buf = realloc(buf, (size_t)cap); }
            }
            fclose(f);
            c->vad_stream = buf; c->vad_len = n; c->vad_pos = 0; c->vad_from_file = 1;
        } else if (strcmp(o->vadmode, "burg")) { set_err(c, "NR: Unknown VAD mode!"); return -1; } /* nr.cc:276 */
        if (!c->vad_from_file && o->nr_when_afterFB) { set_err(c, "NR: Cannot use Burg detector after filter bank!"); return -1; } /* nr.cc:194-195 */
        if (c->vad_from_file && o->nr_when_afterFB) { set_err(c, "oracle: hwss/fwss/2fwss after the filter bank are not restated"); return -1; }
        if (o->rasta) { set_err(c, "oracle: -nr_rasta is not restated"); return -1; }
        /* VAD::silence_frame (vad.cc:727-736) zeroes in->_Xsabs behind a non-speech frame: invisible elsewhere (the frame's features are
         * out, the next get_frame() rewrites the vector), but these modes seed the next file from that vector (nr.cc:212-221) */
        if (!strcmp(o->vad_apply_mode, "silence")) { set_err(c, "oracle: -vad_apply_mode silence together with hwss/fwss/2fwss is not restated"); return -1; }
    }
    if (o->stat_cmvn || o->apply_cmvn) {
        set_err(c, "oracle: CMVN post-processing is outside the restated path");
        return -1;
    }
    if (o->fea_Z_exp > 0 || o->fea_Z_block > 0) { /* row N2, CMS part: src/fea/post_impl.cc:159-240 */
        if (strcmp(o->fea_kind, "dctc") && strcmp(o->fea_kind, "lpc")) {
            set_err(c, "oracle: CMS on non-cepstral kinds is not restated (cms_POST walks fea_ncepcoefs+1 entries whatever the vector holds, post_impl.cc:203-240)");
            return -1;
        }
        if (o->fea_trap) { set_err(c, "oracle: CMS on stacked vectors is not restated (it would subtract a mean from the first fea_ncepcoefs+1 context slots only)"); return -1; }
        if (o->fea_Z_block > 0 && o->length_b < 1) { set_err(c, "oracle: -fea_Z_block shorter than one frame (the reference allocates a ring of length_b <= 0 rows)"); return -1; }
    }
    if (o->fea_delta || o->fea_trap) { /* row N1: restated for the layouts the reference writes completely */
        if (strcmp(o->fea_kind, "dctc") && strcmp(o->fea_kind, "lpc")) {
            set_err(c, "oracle: delta / stacking on non-cepstral kinds is not restated (deltaFEA sizes its vectors as fea_ncepcoefs+1, fea_delta.cc:22-28)");
            return -1;
        }
        if (!o->fea_c0) {
            set_err(c, "oracle: delta / stacking without -fea_c0 is not restated (the writers leave slots of the row unwritten, out.cc:190-201)");
            return -1;
        }
    }
    /* Hamming, src/io/in.cc:139-144 (alpha = 0.54, in.cc:206) */
    c->W = malloc(sizeof(double) * o->window);
    {
        double pi = 2. * asin(1.);
        for (int j = 0; j < o->window; j++) c->W[j] = 0.54 - (1 - 0.54) * cos(2 * pi * j / (o->window - 1.));
    }
    if (c->signal_out) { /* row N3: IN -> NR -> sigOUT; no FB, no FEA (batch.cc:62-65) */
        /* BATCH constructs its VAD in init_out() only (batch.cc:70-76), which the signal path never reaches (batch.cc:62-66): save_frame()
         * then calls through an unassigned pointer (batch.cc:230-241) - the reference crashes, nothing to restate */
        if (strcmp(o->vad_apply_mode, "none") || strcmp(o->vad_out_mode, "none")) { set_err(c, "oracle: VAD together with signal output crashes in the reference (batch.cc:62-66,230-241)"); return -1; }
        r2hc_plan(c, o->wfft);
        /* OLA correction, out.cc:355-377: the largest sum of overlapping Hamming windows over all phases of the shift */
        double pi = 2. * asin(1.), min = 999.;
        c->ola_corr = 0.;
        for (int i = 0; i < o->wshift; i++) {
            int xx = i;
            double y = 0.;
            while (xx < o->window) { y += 0.54 - (1 - 0.54) * cos(2 * pi * (double)xx / (o->window - 1.)); xx += o->wshift; }
            if (y > c->ola_corr) c->ola_corr = y;
            if (y < min) min = y;
        }
        c->B = 0; c->nfea = 0; c->D = 0; c->htk_kind = 0; c->do_vad = 0;
        c->period = (unsigned)floor(.5 + 10000000. * o->wshift / (double)o->fs);
        c->last_power = calloc(o->wfftby2, sizeof(double));
        c->last_fbank = calloc(1, sizeof(double));
        return 0;
    }
    if (fb_design(c)) return -1;
    r2hc_plan(c, o->wfft);
    int B = c->B;
    const char *k = o->fea_kind;
    c->do_vad = (strcmp(o->vad_apply_mode, "none") || strcmp(o->vad_out_mode, "none")); /* batch.cc:34-38 */
    if (!strcmp(k, "spec") || !strcmp(k, "logspec")) c->nfea = B;
    else if (!strcmp(k, "dctc") || !strcmp(k, "lpc")) c->nfea = o->fea_ncepcoefs + 1;
    else if (!strcmp(k, "lpa")) c->nfea = o->fea_lporder + 1;
    else if (!strcmp(k, "trapdct")) {
        if (o->fea_trapdct_traplen % 2 == 0) { set_err(c, "FEA: TRAP length must be odd!"); return -1; }
        if (o->fea_trapdct_ndct >= o->fea_trapdct_traplen) { set_err(c, "FEA: Number of DCT coeffs must be less than TRAP length (c0 is not output)!"); return -1; }
        c->nfea = B * o->fea_trapdct_ndct;
        c->trap_hamm = malloc(sizeof(double) * o->fea_trapdct_traplen);
        for (int i = 0; i < o->fea_trapdct_traplen; i++) /* fea_trap.cc:42-43 */
            c->trap_hamm[i] = 0.54 - (1 - 0.54) * cos(2 * 3.14159265359 * i / (o->fea_trapdct_traplen - 1.));
    } else { set_err(c, "FEA: Unknown feature kind!"); return -1; }
    if (!strcmp(k, "dctc")) { /* fea_impl.cc:81-102 */
        c->wdct = malloc(sizeof(double) * 4 * B);
        for (int i = 0; i < 4 * B; i++) c->wdct[i] = cos(3.1415926535898 * (double)i / (2 * B));
        c->normcoef = sqrt(2.0 / B);
    }
    if (!strcmp(k, "dctc") || !strcmp(k, "lpc")) {
        int n = o->fea_ncepcoefs;
        c->lift = malloc(sizeof(double) * (n > 0 ? n : 1));
        for (int i = 0; i < n; i++)
            c->lift[i] = 1 + ((double)o->fea_lifter) / 2 * sin(3.141592653589793 * (i + 1.) / ((double)o->fea_lifter));
    }
    if (!strcmp(k, "lpa") || !strcmp(k, "lpc")) { /* fea_impl.cc:141-161 */
        int Nfull = (B - 1) * 2;
        if (Nfull < 2) { set_err(c, "oracle: LPC needs at least 2 bands"); return -1; }
        if (!strcmp(k, "lpc") && o->fea_ncepcoefs > o->fea_lporder && 0) { /* reference reads a[] out of bounds only in lpa copy */ }
        c->WRe = malloc(sizeof(double) * Nfull);
        for (int i = 0; i < Nfull; i++) c->WRe[i] = cos(2 * 3.141592653589793 * i / Nfull);
    }
    /* output geometry, src/io/out.cc:95-113 (htkOUT::get_fea_size), 146-159 */
    if ((!strcmp(k, "lpa") || !strcmp(k, "spec") || !strcmp(k, "logspec")) && o->fea_c0) o->fea_c0 = 0;
    c->post_order = 0; c->post_stack = 0; c->Xsize = c->nfea;
    if (o->fea_delta || o->fea_trap) {
        /* BATCH::init_delta, batch.cc:122-130; deltaFEA ctor, fea_delta.cc:20-60 */
        int fea_c = o->fea_ncepcoefs + 1;
        c->post_order = o->n_order;
        c->post_stack = o->fea_trap ? 1 : 0;
        int w[3] = { o->d_win, o->a_win, o->t_win };
        for (int j = 0; j < o->n_order; j++)
            if (w[j] < 1) { set_err(c, o->fea_trap ? "FEA: Trap window size must be >= 3!" : "FEA: Delta window size must be > 1!"); return -1; }
        c->Xsize = o->fea_trap ? fea_c * (2 * o->d_win + 1) : fea_c * (o->n_order + 1);
    }
    int size = c->Xsize;
    if (!strcmp(k, "lpa")) size -= 1;
    if (!strcmp(k, "lpc") && !o->fea_c0) size--;
    if (!strcmp(k, "dctc") && !o->fea_c0) size--;
    if (o->fea_E) size++;
    c->D = size;
    if (size > 32767) { set_err(c, "OUT: HTK format does not support more than 32767 features!"); return -1; }
    c->period = (unsigned)floor(.5 + 10000000. * o->wshift / (double)o->fs);
    int kind;
    if (!strcmp(k, "lpc")) kind = 11;
    else if (!strcmp(k, "dctc")) kind = 6;
    else if (!strcmp(k, "trapdct")) kind = 9;
    else if (!strcmp(k, "spec")) kind = 8;
    else if (!strcmp(k, "logspec")) kind = 7;
    else kind = 9;
    if (o->fea_c0) kind |= 020000;
    if (o->fea_E) kind |= 000100;
    if (o->fea_delta && o->n_order >= 1) kind |= 000400; /* out.cc:157-159; -fea_trap sets fea_delta and n_order=1 too */
    if (o->fea_delta && o->n_order >= 2) kind |= 001000;
    if (o->fea_delta && o->n_order == 3) kind |= 0100000;
    c->htk_kind = kind;
    if (c->do_vad) {
        if (!strcmp(o->vad_cri_mode, "cepdist") && !strcmp(o->vad_cepdist_mode, "lpc") && !o->phase_needed) {
            set_err(c, "VADcri_cepdist: cannot perform iFFT!"); /* vad.cc:167-168: phase vector missing unless -vad burg */
            return -1;
        }
        if (o->vad_filter_order < 1 || (o->vad_filter_order % 2) == 0) { set_err(c, "medianFilter: filter order must be positive, odd number!"); return -1; }
    }
    c->last_power = calloc(o->wfftby2, sizeof(double));
    c->last_fbank = calloc(B, sizeof(double));
    return 0;
}

ctuo_t *ctuo_create(int argc, const char *const *argv, char *err, int errlen) {
    ctuo_t *c = calloc(1, sizeof *c);
    int fail = 0;
    opts_defaults(&c->o);
    /* src/io/opts.cc:158-182: config file first */
    for (int j = 0; j < argc && !fail; j++)
        if (!strcmp(argv[j], "-C") && j + 1 < argc) {
            FILE *f = fopen(argv[j + 1], "r");
            if (!f) { set_err(c, "OPTS: Cannot open config file!"); fail = 1; break; }
            char line[9999];
            while (!fail && fgets(line, sizeof line, f)) {
                line[strcspn(line, "\r\n")] = 0;
                char *h = strstr(line, "#");
                if (h) *h = 0;
                char *l = strtok(line, " \t");
                if (l) {
                    char *r = strtok(NULL, " \t");
                    if (opts_parse(c, l, r)) fail = 1;
                }
            }
            fclose(f);
        }
    /* src/io/opts.cc:184-192 */
    for (int j = 0; j < argc && !fail; j++)
        if (!strncmp(argv[j], "-", 1)) {
            const char *r = NULL;
            if (j + 1 < argc && strncmp(argv[j + 1], "-", 1)) r = argv[j + 1];
            if (opts_parse(c, argv[j], r)) fail = 1;
        }
    if (!fail && opts_check_config(c)) fail = 1;
    if (!fail && design_all(c)) fail = 1;
    if (fail) {
        if (err && errlen > 0) snprintf(err, errlen, "%s", c->err);
        ctuo_destroy(c);
        return NULL;
    }
    return c;
}

void ctuo_destroy(ctuo_t *c) {
    if (!c) return;
    free(c->W);
    if (c->mat) { for (int i = 0; i < MAXB; i++) free(c->mat[i]); free(c->mat); }
    free(c->warp); free(c->hz); free(c->wdct); free(c->lift); free(c->WRe); free(c->trap_hamm);
    free(c->tw_re); free(c->tw_im); free(c->ut_re); free(c->ut_im);
    free(c->last_power); free(c->last_fbank); free(c->ss_stale); free(c->vad_stream);
    free(c);
}

void ctuo_get_dims(const ctuo_t *c, ctuo_dims_t *d) {
    d->fs = c->o.fs; d->window = c->o.window; d->wshift = c->o.wshift; d->wfft = c->o.wfft; d->K = c->o.wfftby2;
    d->B = c->B; d->nfea = c->nfea; d->D = c->D; d->htk_kind = c->htk_kind; d->period = c->period;
    d->do_vad = c->do_vad; d->phase_needed = c->o.phase_needed; d->fb_power = c->o.fb_power; d->swap_out = c->o.swap_out;
}

long ctuo_num_frames(const ctuo_t *c, long n) {
    /* src/io/in.cc:264-279 (pre-load window-wshift), 314,438 (one hop per frame, short read ends the file) */
    long pre = c->o.window - c->o.wshift;
    if (n < pre) return -1;
    return (n - pre) / c->o.wshift;
}

void ctuo_set_vad_ring(ctuo_t *c, int hidx, int hsize) { c->med_hidx = hidx; c->med_hsize = hsize; }
void ctuo_get_vad_ring(const ctuo_t *c, int *hidx, int *hsize) { *hidx = c->med_hidx; *hsize = c->med_hsize; }

const char *ctuo_error(const ctuo_t *c) { return c->err; }
const double *ctuo_hamming(const ctuo_t *c) { return c->W; }
float ctuo_preem(const ctuo_t *c) { return c->o.preem; }
const double *ctuo_last_power(const ctuo_t *c) { return c->last_power; }
const double *ctuo_last_fbank(const ctuo_t *c) { return c->last_fbank; }
int ctuo_fb_row(const ctuo_t *c, int b, const double **w, int *first, int *last) {
    if (b < 0 || b >= c->B) return -1;
    *w = c->mat[b];
    *first = (int)c->mat[b][c->o.wfftby2 + 1];
    *last = (int)c->mat[b][c->o.wfftby2];
    return 0;
}

/* ------------------------------------------------------------------ Burg (src/vdet/Burg.h:49-152) */
void ctuo_burg_cepstrum(const double *x, int np, int nc, double *a_out, double *c_out, double *alpha_out) {
    double *ef = malloc(sizeof(double) * np), *eb = malloc(sizeof(double) * np);
    double *efo = malloc(sizeof(double) * np), *ebo = malloc(sizeof(double) * np);
    double *a = calloc(nc, sizeof(double)), *aa = calloc(nc, sizeof(double));
    double en = 0.0;
    for (int i = 0; i < np; i++) { ef[i] = eb[i] = x[i]; en += pow(x[i], 2.0); } /* Energy.h:38-44 */
    double alpha = en / np;
    a[0] = 1.0;
    for (int ik = 1; ik < nc; ik++) {
        double num = 0.0, den = 0.0;
        for (int i = ik; i < np; i++) {
            den += ef[i] * ef[i] + eb[i - 1] * eb[i - 1];
            num += ef[i] * eb[i - 1];
        }
        num *= 2.0;
        double rc = -num / den;
        a[ik] = rc;
        alpha *= 1 - rc * rc;
        for (int i = 0; i < np; i++) { efo[i] = ef[i]; ebo[i] = eb[i]; }
        for (int i = 1; i < np; i++) {
            ef[i] = ef[i] + (rc * ebo[i - 1]);
            eb[i] = ebo[i - 1] + (rc * efo[i]);
        }
        for (int i = 1; i < ik; i++) a[i] = aa[i] + rc * aa[ik - i];
        for (int i = 1; i <= ik; i++) aa[i] = a[i];
    }
    /* Burg2Cepstrum, Burg.h:141-152 */
    for (int n = 1; n < nc; n++) {
        double sum = 0.0;
        for (int k = 1; k < n; k++) sum += (n - k) * c_out[n - k] * a[k];
        c_out[n] = -a[n] - sum / n;
    }
    c_out[0] = log(alpha);
    if (a_out) memcpy(a_out, a, sizeof(double) * nc);
    if (alpha_out) *alpha_out = alpha;
    free(ef); free(eb); free(efo); free(ebo); free(a); free(aa);
}

/* ------------------------------------------------------------------ CepstralDetector<BurgCepstrumEstimator>
 * src/vdet/CepstralDet.h:92-217 (the detector hwss / fwss / 2fwss create per file, src/nr/nr.cc:263-276).
 * Pinned bit for bit against the reference's own header compiled in place (tests/test_oracle_cepdet_ref.py). */
struct ctuo_cepdet {
    int np, ninit, nc;
    double p, q;
    double *han, *x, *c0, *ci;
    double dMean, dMean2, dVar, threshold, last_dist;
    int nseg;
};

ctuo_cepdet_t *ctuo_cepdet_new(int npoints, int ninit, int ncoefs, double p, double q) {
    ctuo_cepdet_t *d = calloc(1, sizeof *d);
    d->np = npoints; d->ninit = ninit; d->nc = ncoefs; d->p = p; d->q = q;
    d->han = malloc(sizeof(double) * npoints);
    d->x = malloc(sizeof(double) * npoints);
    d->c0 = calloc(ncoefs, sizeof(double));
    d->ci = calloc(ncoefs, sizeof(double));
    double m = 2 * 3.141592653 / npoints;                       /* CepstralDet.h:133 (this literal) */
    for (int i = 0; i < npoints; i++) d->han[i] = 0.5 * (1 - cos(m * i));
    return d;
}

void ctuo_cepdet_free(ctuo_cepdet_t *d) {
    if (!d) return;
    free(d->han); free(d->x); free(d->c0); free(d->ci); free(d);
}

static double cepdet_distance(const ctuo_cepdet_t *d) {      /* CepstralDet.h:64-80: the first coefficient is skipped */
    double sum = 0.0;
    for (int i = 1; i < d->nc; i++) sum += (d->ci[i] - d->c0[i]) * (d->ci[i] - d->c0[i]);
    return 4.3429 * sqrt(2 * sum);
}

int ctuo_cepdet_process(ctuo_cepdet_t *d, const double *frame) {  /* CepstralDet.h:140-194 */
    int result = 0;
    for (int i = 0; i < d->np; i++) d->x[i] = d->han[i] * frame[i];
    ctuo_burg_cepstrum(d->x, d->np, d->nc, NULL, d->ci, NULL);
    if (d->nseg == 0) {
        for (int i = 0; i < d->nc; i++) d->c0[i] = d->ci[i];
    } else if (d->nseg == 1) {
        for (int i = 0; i < d->nc; i++) d->c0[i] = (d->c0[i] + d->ci[i]) / 2.0;
        double dist = cepdet_distance(d);
        d->last_dist = dist;
        d->dMean = dist;
        d->dMean2 = dist * dist;
        d->threshold = d->dMean;
    } else {
        double dist = cepdet_distance(d);
        d->last_dist = dist;
        result = (d->nseg > d->ninit && dist >= d->threshold);
        if (!result) {
            for (int i = 0; i < d->nc; i++) d->c0[i] = d->p * d->c0[i] + (1 - d->p) * d->ci[i];
            d->dMean = d->q * d->dMean + (1 - d->q) * dist;
            d->dMean2 = d->q * d->dMean2 + (1 - d->q) * dist * dist;
            d->dVar = d->dMean2 - d->dMean * d->dMean;
            d->threshold = d->dMean + 2.0 * sqrt(d->dVar);
        }
    }
    ++d->nseg;
    return result;
}

double ctuo_cepdet_last_distance(const ctuo_cepdet_t *d) { return d->last_dist; }

/* ------------------------------------------------------------------ phase (src/io/in.cc:187-200) */
static double c_ph(double re, double im) {
    static const double hpi = 1.57079632679490;
    static const double pi = 3.14159265358979;
    if (re == 0.0) return im > 0.0 ? hpi : -hpi;
    double y = atan(im / re);
    if (re < 0.0 && im >= 0.0) y += pi;
    if (re < 0.0 && im < 0.0) y -= pi;
    return y;
}

/* ------------------------------------------------------------------ VAD state (src/vad/vad.cc, vad.h) */
typedef struct {
    /* cepdist */
    double *c0, *ci;
    int csize;
    /* thresholds */
    double crimin, crimax;                 /* perc  */
    double crimean, crimean2, crivar, thr; /* adapt */
    int adapt_vad;
    double dmin, dmax;                     /* dyn   */
    /* median filter */
    int *history;
    int hidx, hsize;
    /* feature delay ring: order x nfea */
    double *ring;
    int start, ready;
} vad_state_t;

static int thr_process(const opts_t *o, vad_state_t *v, int t, double cri) {
    const char *m = o->vad_thr_mode;
    if (!strcmp(m, "absolute")) return cri >= o->vad_absolute_thr; /* vad.cc:329-331 */
    if (!strcmp(m, "perc")) { /* vad.cc:384-398 ; opt_init is a double there */
        if (t == 0 || (double)t < (double)o->vad_perc_init) { v->crimin = cri; v->crimax = cri; }
        else { if (cri < v->crimin) v->crimin = cri; if (cri > v->crimax) v->crimax = cri; }
        double thr = v->crimin + (o->vad_perc_thr / 100.0) * (v->crimax - v->crimin);
        return cri >= thr;
    }
    if (!strcmp(m, "adapt")) { /* vad.cc:469-495 */
        if (t == 0) {
            v->thr = cri; v->crimean = cri; v->crimean2 = cri * cri; v->crivar = 0.0; v->adapt_vad = 0;
        } else {
            v->thr = v->crimean + o->vad_adapt_za * sqrt(v->crivar);
            if ((cri < v->thr) || (t <= o->vad_adapt_init)) {
                v->crimean = o->vad_adapt_q * v->crimean + (1.0 - o->vad_adapt_q) * cri;
                v->crimean2 = o->vad_adapt_q * v->crimean2 + (1.0 - o->vad_adapt_q) * cri * cri;
                v->crivar = v->crimean2 - v->crimean * v->crimean;
                v->adapt_vad = 0;
            } else v->adapt_vad = 1;
        }
        return v->adapt_vad;
    }
    /* dyn, vad.cc:578-625 */
    int init = o->vad_dyn_init > 1 ? o->vad_dyn_init : 1;
    if (t < init) { v->dmax = cri; v->dmin = cri; return 0; }
    if (t == init) {
        if (cri > v->dmax) v->dmax = cri;
        v->dmax += o->vad_dyn_min / 10.0;
        if (cri < v->dmin) v->dmin = cri;
        v->dmin -= o->vad_dyn_min / 10.0;
        return 0;
    }
    if (v->dmax < cri) v->dmax = o->vad_dyn_qmaxinc * v->dmax + (1.0 - o->vad_dyn_qmaxinc) * cri;
    else v->dmax = o->vad_dyn_qmaxdec * v->dmax + (1.0 - o->vad_dyn_qmaxdec) * cri;
    if (v->dmin > cri) v->dmin = o->vad_dyn_qmindec * v->dmin + (1.0 - o->vad_dyn_qmindec) * cri;
    else v->dmin = o->vad_dyn_qmininc * v->dmin + (1.0 - o->vad_dyn_qmininc) * cri;
    double dyn = v->dmax - v->dmin;
    double thr = v->dmin + (o->vad_dyn_perc / 100.0) * dyn;
    return (cri > thr) && (dyn > o->vad_dyn_min);
}

/* ------------------------------------------------------------------ row assembly (src/io/out.cc:174-203) */
static void emit_row(const ctuo_t *c, const double *X, double E, float *row) {
    const opts_t *o = &c->o;
    const char *k = o->fea_kind;
    int Xsize = c->nfea;
    if (!strcmp(k, "spec") || !strcmp(k, "logspec") || !strcmp(k, "trapdct")) {
        for (int i = 0; i < Xsize; i++) row[i] = (float)X[i];
        if (o->fea_E) row[Xsize] = (float)E;
    } else if (c->post_stack) { /* out.cc:182: -fea_trap switches the writers to the straight "spec" copy */
        Xsize = c->Xsize;
        for (int i = 0; i < Xsize; i++) row[i] = (float)X[i];
        if (o->fea_E) row[Xsize] = (float)E;
    } else { /* out.cc:188-201: block j = (c1..cN, c0) of the j-th derivative; E once, after the last block */
        int nc = o->fea_ncepcoefs;
        Xsize = c->Xsize;
        for (int j = 0; j <= c->post_order; j++) {
            for (int i = (nc + 1) * j + 1; i < (nc + 1) * (j + 1); i++) row[i - 1] = (float)X[i];
            if (o->fea_c0) {
                row[nc * (j + 1) + j] = (float)X[(nc + 1) * j];
                if (o->fea_E) row[Xsize] = (float)E;
            } else if (o->fea_E) row[nc * (j + 1) + j] = (float)E;
        }
    }
}

/* ------------------------------------------------------------------ cms_POST (src/fea/post_impl.cc:159-240)
 * Running cepstral mean over the first fea_ncepcoefs+1 entries of the vector OUT is about to write.  The mean and the
 * block ring are `float` in the reference, the vector is double; BATCH resets the state per file (batch.cc:388-392). */
typedef struct {
    int type;      /* 0 off, 1 exp, 2 block (block wins when both are set, post_impl.cc:163-167) */
    int n, L, num_frame;
    float *sumM, *ring;
} cms_t;

static void cms_init(cms_t *m, const opts_t *o) {
    memset(m, 0, sizeof *m);
    if (o->fea_Z_exp > 0) m->type = 1;
    if (o->fea_Z_block > 0) m->type = 2;
    if (!m->type) return;
    m->n = o->fea_ncepcoefs + 1;
    m->L = o->length_b > 0 ? o->length_b : 1;
    m->sumM = calloc(m->n, sizeof(float));
    m->ring = calloc((size_t)m->L * m->n, sizeof(float));
}

static void cms_free(cms_t *m) { free(m->sumM); free(m->ring); }

static void cms_apply(cms_t *m, const opts_t *o, double *F) {
    if (m->type == 1) { /* post_impl.cc:207-212 */
        for (int i = 0; i < m->n; i++) {
            m->sumM[i] = m->sumM[i] * (o->fea_Z_exp) + F[i] * (1 - o->fea_Z_exp);
            F[i] -= m->sumM[i];
        }
    } else if (m->type == 2) { /* post_impl.cc:214-236 */
        int x = m->num_frame % m->L;
        for (int j = 0; j < m->n; j++) m->ring[(size_t)x * m->n + j] = F[j];
        if (m->num_frame >= m->L - 1) {
            for (int i = 0; i < m->n; i++) m->sumM[i] = 0;
            for (int i = 0; i < m->n; i++)
                for (int xx = 0; xx < m->L; xx++) m->sumM[i] += m->ring[(size_t)xx * m->n + i];
            for (int i = 0; i < m->n; i++) m->sumM[i] /= m->L;
        }
        for (int i = 0; i < m->n; i++) F[i] -= m->sumM[i];
    }
    m->num_frame++;
}

/* ------------------------------------------------------------------ deltaFEA (src/fea/fea_delta.cc)
 * One stage of the chain BATCH::init_delta builds (batch.cc:122-125).  The reference runs it as a streaming
 * ring of 2w+1 input vectors; the ring, its priming (first frame w times, second frame twice) and the flush
 * (last input repeated) are kept as they are so that every edge effect of the original falls out of the same
 * bookkeeping instead of being described by hand. */
typedef struct {
    int w, L;            /* delta_w, dwlen */
    int start, end, frame, endframe, avail, index, num_c, den;
    int n_order, nfea, fea_c, stack;
    const double *in;    /* upstream vector (nfea values are consumed) */
    double *out;         /* fea_c * n_order values */
    double *ring;        /* L x nfea */
} dstage_t;

static void dstage_new_file(dstage_t *s) { /* fea_delta.cc:62-72 */
    s->start = 0; s->end = 0; s->endframe = 1; s->frame = 1;
    s->avail = s->w + 1; s->index = s->avail - 1; s->num_c = 1;
}

static void dstage_init(dstage_t *s, const opts_t *o, const double *in, int n_order, int w) { /* fea_delta.cc:20-60 */
    memset(s, 0, sizeof *s);
    s->in = in; s->fea_c = o->fea_ncepcoefs + 1;
    s->w = w; s->L = 2 * w + 1; s->n_order = n_order; s->stack = o->fea_trap;
    s->nfea = s->fea_c * (n_order - 1);
    if (o->fea_trap) s->n_order = s->L;
    s->out = calloc((size_t)s->fea_c * s->n_order, sizeof(double));
    s->ring = calloc((size_t)s->L * s->nfea, sizeof(double));
    for (int i = 1; i <= w; i++) s->den += i * i;
    s->den *= 2;
    dstage_new_file(s);
}

static void dstage_free(dstage_t *s) { free(s->out); free(s->ring); }

static void dstage_prime(dstage_t *s) { /* init_cbuffer, fea_delta.cc:132-144 */
    for (int r = 0; r < s->num_c; r++) {
        memcpy(s->ring + (size_t)s->index * s->nfea, s->in, sizeof(double) * s->nfea);
        s->index = (s->index + 1) % s->L;
    }
}

static void dstage_compute(dstage_t *s) {
    const int L = s->L, w = s->w, nf = s->nfea, fc = s->fea_c;
    if (s->stack) { /* trap(), fea_delta.cc:166-176: coefficient-major stack of the ring in time order */
        for (int i = 0; i < fc; i++)
            for (int j = 0; j < L; j++) s->out[i * L + j] = s->ring[(size_t)((s->start + j) % L) * nf + i];
        return;
    }
    /* delta(), fea_delta.cc:146-164: regression over the newest block of the input vector */
    for (int j = fc * (s->n_order - 2); j < fc * (s->n_order - 1); j++) {
        double x = 0;
        for (int i = 1; i <= w; i++)
            x += i * (s->ring[(size_t)((s->start + w + i) % L) * nf + j] - s->ring[(size_t)((s->start + w - i) % L) * nf + j]);
        s->out[j + fc] = x / s->den;
    }
}

static int dstage_process(dstage_t *s) { /* fea_delta.cc:74-130 */
    const int L = s->L, w = s->w, nf = s->nfea;
    if (s->avail != 0) {
        s->num_c = s->frame == 1 ? w : (s->frame == 2 ? 2 : 1);
        dstage_prime(s);
        s->avail--;
        int ready = 0;
        if (!s->avail) {
            dstage_compute(s);
            memcpy(s->ring + (size_t)(L - 1) * nf, s->ring + (size_t)(L - 2) * nf, sizeof(double) * nf);
            memcpy(s->out, s->ring + (size_t)w * nf, sizeof(double) * nf);
            ready = 1;
        }
        s->start = (s->start + 1) % L;
        s->frame++;
        return ready;
    }
    int fx = s->start == 0 ? L - 1 : s->start - 1;
    memcpy(s->ring + (size_t)fx * nf, s->in, sizeof(double) * nf);
    dstage_compute(s);
    int ox = fx - w;
    if (ox < 0) ox += L;
    if (!s->stack) memcpy(s->out, s->ring + (size_t)ox * nf, sizeof(double) * nf);
    s->end = s->start;
    s->start = (s->start + 1) % L;
    s->frame++;
    return 1;
}

static int dstage_flush(dstage_t *s) { /* fea_delta.cc:178-206 */
    const int L = s->L, w = s->w, nf = s->nfea;
    if (s->endframe) { s->index = s->end; s->endframe = 0; }
    if (s->avail < w) {
        dstage_prime(s);
        dstage_compute(s);
        int ox = s->start - w - 1;
        if (ox < 0) ox += L;
        memcpy(s->out, s->ring + (size_t)ox * nf, sizeof(double) * nf);
        s->start = (s->start + 1) % L;
        s->avail++;
        return 1;
    }
    return 0;
}

/* ------------------------------------------------------------------ the chain */
long ctuo_process(ctuo_t *c, const int16_t *pcm, long nsamples, float *rows, unsigned char *vadout);

long ctuo_out_samples(const ctuo_t *c, long nsamples) { /* T frames leave T*wshift samples, close() adds window-wshift */
    long T = ctuo_num_frames(c, nsamples);
    if (T < 0) return -1;
    return T * c->o.wshift + (c->o.window - c->o.wshift);
}

long ctuo_enhance(ctuo_t *c, const int16_t *pcm, long nsamples, int16_t *out) {
    if (!c->signal_out) { set_err(c, "oracle: -format_out raw|wave needed for ctuo_enhance"); return -1; }
    c->sig_buf = out;
    long n = ctuo_process(c, pcm, nsamples, NULL, NULL);
    c->sig_buf = NULL;
    return n;
}

long ctuo_process(ctuo_t *c, const int16_t *pcm, long nsamples, float *rows, unsigned char *vadout) {
    opts_t *o = &c->o;
    const int window = o->window, wshift = o->wshift, wfft = o->wfft, K = o->wfftby2, B = c->B;
    const char *kind = o->fea_kind;
    long T = ctuo_num_frames(c, nsamples);
    if (T < 0) { set_err(c, "IO: Signal shorter than one frame!"); return -1; }
    if (c->signal_out && !c->sig_buf) { set_err(c, "oracle: this configuration writes speech, use ctuo_enhance"); return -1; }
    const int is_trap = !strcmp(kind, "trapdct");
    const int traplen = o->fea_trapdct_traplen, ndct = o->fea_trapdct_ndct, htrap = (traplen + 1) / 2;
    if (is_trap && T > 0 && T < htrap) {
        set_err(c, "oracle: trapdct on fewer than (traplen+1)/2 frames reads never-written ring rows in the reference (src/fea/fea_trap.cc:64-70,111-127); undefined");
        return -1;
    }
    const int is_post = c->post_order > 0;
    if (is_post && T > 0) {
        int wmax = o->d_win;
        if (c->post_order >= 2 && o->a_win > wmax) wmax = o->a_win;
        if (c->post_order >= 3 && o->t_win > wmax) wmax = o->t_win;
        if (T < wmax + 1) {
            set_err(c, "oracle: delta / stacking on fewer than window+1 frames emits rows built from never-written ring slots in the reference (src/fea/fea_delta.cc:74-130,178-206); undefined");
            return -1;
        }
    }
    cms_t cms;
    cms_init(&cms, o);
    /* the circular buffer of in.cc is kept here as a linear double copy of the signal;
     * remove_dc1 mutates it persistently exactly as cbuffer is mutated (in.cc:343-350). */
    double *x = malloc(sizeof(double) * (nsamples > 0 ? nsamples : 1));
    for (long i = 0; i < nsamples; i++) x[i] = (double)pcm[i]; /* in.cc:456-457 */
    double *fft_in = calloc(wfft, sizeof(double));
    double *Xre = malloc(sizeof(double) * K), *Xim = malloc(sizeof(double) * K);
    double *zr = malloc(sizeof(double) * wfft), *zi = malloc(sizeof(double) * wfft);
    double *Xabs = malloc(sizeof(double) * K), *Xph = malloc(sizeof(double) * K);
    double *Y = malloc(sizeof(double) * B);
    double *fvec = calloc(c->nfea > B ? c->nfea : B, sizeof(double));
    double *Navg = malloc(sizeof(double) * K), *Yavg = malloc(sizeof(double) * K);
    double *trapbuf = NULL, *trapE = NULL, *tin = NULL;
    double *postbuf = NULL, *postE = NULL;
    if (is_post) {
        postbuf = malloc(sizeof(double) * (size_t)(T > 0 ? T : 1) * c->nfea);
        postE = malloc(sizeof(double) * (size_t)(T > 0 ? T : 1));
    }
    if (is_trap) {
        trapbuf = malloc(sizeof(double) * (size_t)(T > 0 ? T : 1) * B);
        trapE = malloc(sizeof(double) * (size_t)(T > 0 ? T : 1));
        tin = malloc(sizeof(double) * traplen);
    }
    /* NR new_file, nr.cc:86-93 */
    for (int i = 0; i < K; i++) { Navg[i] = 0.95; Yavg[i] = 0.05; }
    /* hwss / fwss / 2fwss new_file (nr.cc:212-221, 402-409): a fresh detector, the noise estimate seeded from the stale vector */
    ctuo_cepdet_t *ss_det = NULL;
    double *ss_nr = NULL, *ss_re = NULL, *ss_im = NULL, *ss_t = NULL, *ss_w1 = NULL, *ss_w2 = NULL;
    int ss_ninit = o->nr_initsegs;
    if (c->ss_mode) {
        if (!c->ss_stale) c->ss_stale = calloc(K, sizeof(double));
        ss_det = ctuo_cepdet_new(window, ss_ninit, o->fea_ncepcoefs, o->nr_p, o->nr_q); /* vad_init, nr.cc:263-276 */
        ss_nr = calloc(K, sizeof(double));
        ss_re = malloc(sizeof(double) * K); ss_im = malloc(sizeof(double) * K);
        ss_t = malloc(sizeof(double) * wfft); ss_w1 = malloc(sizeof(double) * wfft); ss_w2 = malloc(sizeof(double) * wfft);
        for (int i = 0; i < K; i++) {
            Navg[i] = c->ss_mode == 3 ? c->ss_stale[i] : pow(c->ss_stale[i], o->nr_a);
            c->ss_stale[i] *= 0.1; /* nr.cc:219, :406 "init phase -> out = 0.1 * in": the first get_frame() overwrites the vector, but a file
                                      without a frame (window - wshift <= N < window) leaves it scaled for the file behind it */
        }
    }
    int lporder = o->fea_lporder;
    double *RRe = calloc(lporder + 2, sizeof(double)), *rc = calloc(lporder + 2, sizeof(double));
    double *a = calloc(lporder + 2, sizeof(double)), *aa = calloc(lporder + 2, sizeof(double)), *P = calloc(lporder + 2, sizeof(double));
    /* VAD */
    vad_state_t vs;
    memset(&vs, 0, sizeof vs);
    const int do_vad = c->do_vad, order = o->vad_filter_order, hdelay = (order - 1) / 2;
    const int vlen = c->Xsize > c->nfea ? c->Xsize : c->nfea; /* the vector the VAD's ring delays is the one OUT sees (batch.cc:76,122-130) */
    double cur_en = 0.0;      /* energy criterion of the newest input frame */
    double *en_t = NULL, *ci_t = NULL; /* delta / stacking: the criteria of every input frame, for the replay below */
    double *hw1 = NULL, *hw2 = NULL, *tsig = NULL, *hre = NULL, *him = NULL;
    if (do_vad) {
        vs.history = calloc(order, sizeof(int));
        vs.ring = calloc((size_t)order * vlen, sizeof(double));
        vs.hidx = c->med_hidx % order;   /* what the previous file of this process left (0, 0 for the first) */
        vs.hsize = c->med_hsize;
        if (!strcmp(o->vad_cri_mode, "cepdist")) {
            vs.csize = !strcmp(o->vad_cepdist_mode, "lpc") ? o->vad_lpc_coefs : c->nfea;
            vs.c0 = calloc(vs.csize, sizeof(double));
            vs.ci = calloc(vs.csize, sizeof(double));
            hw1 = malloc(sizeof(double) * wfft); hw2 = malloc(sizeof(double) * wfft); tsig = malloc(sizeof(double) * wfft);
            hre = malloc(sizeof(double) * K); him = malloc(sizeof(double) * K);
        }
    }
    double E_last = -1.; /* E is NOT delayed by the median filter: out reads it through a pointer at save time */

    /* sigOUT state (out.cc:397-403): OLA ring of `window` doubles, `start` = ring position of the frame's first sample */
    double *ola = NULL, *ytime = NULL, *sw1 = NULL, *sw2 = NULL, *sre = NULL, *sim = NULL;
    long ola_start = 0, nout = 0;
    if (c->signal_out) {
        ola = calloc(window, sizeof(double));
        ytime = malloc(sizeof(double) * wfft); sw1 = malloc(sizeof(double) * wfft); sw2 = malloc(sizeof(double) * wfft);
        sre = malloc(sizeof(double) * K); sim = malloc(sizeof(double) * K);
    }
    double preemtmp = 0.; /* in.cc:274 */
    long nrows = 0, nvad = 0;
    int fail = 0;

    if ((is_post || is_trap) && do_vad) {
        en_t = calloc((size_t)(T > 0 ? T : 1), sizeof(double));
        if (vs.csize && !strcmp(o->vad_cepdist_mode, "lpc")) ci_t = calloc((size_t)(T > 0 ? T : 1) * vs.csize, sizeof(double));
    }
    /* One call of BATCH::save_frame with the VAD on (batch.cc:230-241): VAD::process_frame (vad.cc:692-703) on the criterion
     * of the newest INPUT frame (cur_en / vs.ci; the `fea` mode takes the vector OUT sees), threshold, consume_vad, median
     * filter with its ring of output vectors (vad.h:126-150), then the writer.  TC = number of calls so far (the VAD's
     * frame_index), FV = the vector OUT sees at this call, EV = the energy the writer reads through its pointer. */
#define VAD_CALL(TC, FV, EV) do {                                                                                              \
        double cri_;                                                                                                           \
        if (!strcmp(o->vad_cri_mode, "energy")) cri_ = cur_en;                                                                 \
        else {                                                                                                                 \
            if (!strcmp(o->vad_cepdist_mode, "fea")) for (int i_ = 0; i_ < vs.csize; i_++) vs.ci[i_] = (FV)[i_];                \
            if ((TC) == 0) {                                                                                                   \
                for (int i_ = 0; i_ < vs.csize; i_++) vs.c0[i_] = vs.ci[i_];                                                   \
                cri_ = 0.0;                                                                                                    \
            } else {                                                                                                           \
                if ((TC) == 1) for (int i_ = 0; i_ < vs.csize; i_++) vs.c0[i_] = (vs.c0[i_] + vs.ci[i_]) / 2.0;                \
                double sum_ = 0.0;                                                                                             \
                for (int i_ = 1; i_ < vs.csize; i_++) sum_ += (vs.ci[i_] - vs.c0[i_]) * (vs.ci[i_] - vs.c0[i_]);               \
                cri_ = 4.3429 * sqrt(2 * sum_);                                                                                \
            }                                                                                                                  \
        }                                                                                                                      \
        int vad0_ = thr_process(o, &vs, (int)(TC), cri_);                                                                      \
        if (vs.c0 && !(vad0_ && ((TC) > o->vad_cepdist_init))) /* consume_vad, vad.cc:288-294 */                               \
            for (int i_ = 0; i_ < vs.csize; i_++) vs.c0[i_] = o->vad_cepdist_p * vs.c0[i_] + (1.0 - o->vad_cepdist_p) * vs.ci[i_]; \
        vs.history[vs.hidx] = vad0_;                                                                                           \
        memcpy(vs.ring + (size_t)vs.hidx * vlen, (FV), sizeof(double) * vlen);                                                 \
        vs.hidx = (vs.hidx + 1) % order;                                                                                       \
        if (vs.hsize < hdelay) { vs.hsize++; break; } /* not ready: nothing saved (batch.cc:235-236) */                        \
        vs.ready = 1;                                                                                                          \
        double hs_ = 0.0;                                                                                                      \
        for (int i_ = 0; i_ < order; i_++) hs_ += vs.history[i_] ? 1.0 : 0.0;                                                  \
        const double *fo_ = vs.ring + (size_t)(vs.start % order) * vlen;                                                       \
        vs.start++;                                                                                                            \
        int dec_ = (hs_ / (double)order) >= 0.5;                                                                               \
        if (vadout && strcmp(o->vad_out_mode, "none")) vadout[nvad] = dec_ ? '1' : '0';                                        \
        nvad++;                                                                                                                \
        if (!(!dec_ && !strcmp(o->vad_apply_mode, "drop"))) {                                                                  \
            emit_row(c, fo_, (EV), rows + (size_t)nrows * c->D);                                                               \
            nrows++;                                                                                                           \
        }                                                                                                                      \
    } while (0)

/* hwssNR / fwssNR / dfwssNR::process_frame (nr.cc:223-261, 331-369, 418-442) on the vector XVEC (K values); `break`s out of the frame loop on a failed VAD stream */ \
#define SS_STEP(XVEC) { \
 /* hwssNR / fwssNR / dfwssNR::process_frame, nr.cc:223-261, 331-369, 418-442 */ \
                    const double aexp = o->nr_a, bsub = o->nr_b, p = o->nr_p; \
                    double *X = (XVEC); \
                    if (c->ss_mode == 1) ss_ninit--;                       /* hwss counts down first (nr.cc:225) */ \
                    if (c->ss_mode != 3) {                                 /* dynamic expansion */ \
                        if (aexp == 2.0) for (int i = 0; i < K; i++) X[i] *= X[i]; \
                        else if (aexp != 1.0) for (int i = 0; i < K; i++) X[i] = pow(X[i], aexp); \
                    } \
                    int vad; \
                    if (c->vad_from_file) { \
                        /* nr.cc:297-302: `char vad = fgetc(fvad); if (vad != EOF) return bool(vad); else throw ...` - every byte but NUL \
                         * is speech (an ASCII '0' too), and a byte 0xFF compares equal to EOF in the (signed) char and ends the run */ \
                        const int ch = c->vad_pos < c->vad_len ? (int)c->vad_stream[c->vad_pos++] : EOF; \
                        const char vc = (char)ch; \
                        if (vc == EOF) { set_err(c, "NR: Unexpected end of VAD file!"); fail = 1; break; } \
                        vad = vc != 0; \
                    } else { \
                        /* vad_get_frame, nr.cc:278-295: back to the time domain with the original phase, first `window` samples */ \
                        for (int i = 0; i < K; i++) { ss_re[i] = X[i] * cos(Xph[i]); ss_im[i] = (i == 0 || i == K - 1) ? 0.0 : X[i] * sin(Xph[i]); } \
                        hc2r(ss_re, ss_im, wfft, ss_t, ss_w1, ss_w2); \
                        vad = ctuo_cepdet_process(ss_det, ss_t); \
                    } \
                    if (vad == 0 || ss_ninit > 0) for (int i = 0; i < K; i++) Navg[i] = p * Navg[i] + (1 - p) * X[i]; \
                    if (c->ss_mode == 3) { \
                        for (int i = 0; i < K; i++) { X[i] -= Navg[i]; if (X[i] < 0.) X[i] = -X[i]; } \
                        if (vad == 0 || ss_ninit > 0) for (int i = 0; i < K; i++) ss_nr[i] = p * ss_nr[i] + (1 - p) * X[i]; \
                        for (int i = 0; i < K; i++) { X[i] -= ss_nr[i]; if (X[i] < 0.) X[i] = -X[i]; } \
                    } else { \
                        for (int i = 0; i < K; i++) { \
                            X[i] -= bsub * Navg[i]; \
                            if (X[i] < 0.) X[i] = c->ss_mode == 1 ? 0. : -X[i]; \
                        } \
                        if (aexp == 2.) for (int i = 0; i < K; i++) X[i] = sqrt(X[i]); \
                        else if (aexp != 1.) for (int i = 0; i < K; i++) X[i] = pow(X[i], 1.0 / aexp); \
                    } \
                    if (c->ss_mode != 1) ss_ninit--; \
                    memcpy(c->ss_stale, X, sizeof(double) * K); \
}

    for (long t = 0; t < T && !fail; t++) {
        const long s = t * (long)wshift;
        /* ---- rawIN::get_frame, in.cc:305-419 */
        if (o->remove_dc1) {
            double off = 0;
            for (int i = 0; i < window; i++) off += x[s + i];
            off /= window;
            for (int i = 0; i < window; i++) x[s + i] -= off;
        }
        double E_in = -1.;
        if (o->fea_E && o->fea_rawenergy) {
            E_in = 0;
            for (int i = 1; i < window; i++) E_in += x[s + i] * x[s + i];
            E_in = log(E_in);
        }
        if (o->preem > 0.) {
            fft_in[0] = c->W[0] * (x[s] - o->preem * preemtmp);
            for (int i = 1; i < window; i++) fft_in[i] = c->W[i] * (x[s + i] - o->preem * x[s + i - 1]);
        } else
            for (int i = 0; i < window; i++) fft_in[i] = c->W[i] * x[s + i];
        if (o->remove_dc) {
            double off = 0;
            for (int i = 0; i < window; i++) off += fft_in[i];
            off /= window;
            for (int i = 0; i < window; i++) fft_in[i] -= off;
        }
        preemtmp = x[s + wshift - 1];
        for (int i = window; i < wfft; i++) fft_in[i] = 0.;
        r2hc(c, fft_in, Xre, Xim, zr, zi);
        if (o->remove_dc) Xabs[0] = 1e-10;
        else Xabs[0] = Xre[0] * Xre[0];
        Xabs[K - 1] = Xre[K - 1] * Xre[K - 1];
        for (int i = 1; i < K - 1; i++) Xabs[i] = Xre[i] * Xre[i] + Xim[i] * Xim[i];
        if (o->phase_needed) {
            Xph[0] = 0;
            Xph[K - 1] = (Xre[K - 1] >= 0) ? 0 : 3.14159265358979;
            for (int i = 1; i < K - 1; i++) Xph[i] = c_ph(Xre[i], Xim[i]);
        }
        if (o->fea_E && !o->fea_rawenergy) {
            double E = Xabs[0] / 2. + Xabs[K - 1] / 2.;
            for (int i = 1; i < K - 1; i++) E += Xabs[i];
            E_in = log(E * 2.);
        }
        if (!o->fb_power) for (int i = 0; i < K; i++) Xabs[i] = sqrt(Xabs[i]);

        if (c->signal_out) {
            /* BATCH::process_frame, batch.cc:223-227: nr->process_frame(); save_frame() -> sigOUT::save_frame */
            if (!strcmp(o->nr_mode, "exten")) { /* nr.cc:95-140 */
                double aexp = o->nr_a, p = o->nr_p;
                for (int i = 0; i < K; i++) {
                    double H;
                    if (aexp == 1.0) H = Navg[i] / (Navg[i] + Yavg[i]);
                    else if (aexp == 2.0) H = Navg[i] / sqrt(Navg[i] * Navg[i] + Yavg[i] * Yavg[i]);
                    else H = Navg[i] / pow(pow(Navg[i], aexp) + pow(Yavg[i], aexp), 1. / aexp);
                    double N = H * Xabs[i];
                    Navg[i] = p * Navg[i] + (1 - p) * N;
                    if (Xabs[i] > Navg[i]) Yavg[i] = Xabs[i] - Navg[i];
                    else Yavg[i] = Navg[i] - Xabs[i];
                    Xabs[i] -= N;
                }
            }
            if (c->ss_mode) SS_STEP(Xabs)   /* the same NR object on in->_Xsabs (batch.cc:62-66) */
            /* out.cc:405-434.  The ring slots that fall out of the window are cleared, the spectrum goes back to
             * Re/Im with the ORIGINAL phase and a 1/N factor (DC and Nyquist keep their magnitude as a positive real:
             * the sign flip at out.cc:419 comes after the value was stored), HC2R, overlap-add of the first `window`
             * samples, then `wshift` finished samples leave through floor(x / correction) with +-32767 clipping. */
            for (int i = 0; i < wshift; i++) ola[(ola_start + window - wshift + i) % window] = 0.;
            if (o->fb_power) for (int i = 0; i < K; i++) Xabs[i] = sqrt(Xabs[i]);
            sre[0] = Xabs[0] / (double)wfft; sim[0] = 0;
            sre[K - 1] = Xabs[K - 1] / (double)wfft; sim[K - 1] = 0;
            for (int i = 1; i < K - 1; i++) {
                double ampl = Xabs[i] / (double)wfft;
                sre[i] = ampl * cos(Xph[i]);
                sim[i] = ampl * sin(Xph[i]);
            }
            /* out.cc:419: `if(Xp[size-1]!=0) Xa[size-1]=-Xa[size-1];` - after its value went into the transform.  Nothing reads the vector
             * again in this file, but hwss / fwss / 2fwss seed the NEXT file's noise estimate from it (nr.cc:212-221) */
            if (c->ss_mode && Xph[K - 1] != 0) c->ss_stale[K - 1] = -c->ss_stale[K - 1];
            hc2r(sre, sim, wfft, ytime, sw1, sw2);
            for (int i = 0; i < window; i++) ola[(ola_start + i) % window] += ytime[i];
            for (int i = 0; i < wshift; i++) {
                int value = (int)floor(ola[(ola_start + i) % window] / c->ola_corr);
                c->sig_buf[nout++] = (fabs((float)value) > 32767) ? (value < 0 ? -32767 : 32767) : (int16_t)value;
            }
            ola_start += wshift;
            continue;
        }
        /* ---- NR / FB order, batch.cc:205-213 */
        double E_nr = -1.;
        double *nrvec = o->nr_when_afterFB ? Y : Xabs;
        int nrsize = o->nr_when_afterFB ? B : K;
        for (int pass = 0; pass < 2; pass++) {
            int do_fb = o->nr_when_afterFB ? (pass == 0) : (pass == 1);
            if (do_fb) { /* FB::project_frame, fb.cc:72-86 */
                for (int b = 0; b < B; b++) {
                    double acc = 0;
                    int k0 = (int)c->mat[b][K + 1], k1 = (int)c->mat[b][K];
                    for (int k = k0; k <= k1; k++) acc += Xabs[k] * c->mat[b][k];
                    if (o->fb_inld) acc = pow(acc, 0.33);
                    Y[b] = acc;
                }
            } else {
                if (!strcmp(o->nr_mode, "exten")) { /* nr.cc:95-140 (state sized K in the reference even afterFB: size = Xabs->get_size()) */
                    double aexp = o->nr_a, p = o->nr_p;
                    for (int i = 0; i < nrsize; i++) {
                        double H;
                        if (aexp == 1.0) H = Navg[i] / (Navg[i] + Yavg[i]);
                        else if (aexp == 2.0) H = Navg[i] / sqrt(Navg[i] * Navg[i] + Yavg[i] * Yavg[i]);
                        else H = Navg[i] / pow(pow(Navg[i], aexp) + pow(Yavg[i], aexp), 1. / aexp);
                        double N = H * nrvec[i];
                        Navg[i] = p * Navg[i] + (1 - p) * N;
                        if (nrvec[i] > Navg[i]) Yavg[i] = nrvec[i] - Navg[i];
                        else Yavg[i] = Navg[i] - nrvec[i];
                        nrvec[i] -= N;
                    }
                }
                if (c->ss_mode) SS_STEP(nrvec)
                if (o->fea_E && !o->fea_rawenergy) { /* _NR::compute_E, nr.cc:36-45 */
                    double E = nrvec[0] * nrvec[0] / 2. + nrvec[nrsize - 1] * nrvec[nrsize - 1] / 2.;
                    for (int i = 1; i < nrsize - 1; i++) E += nrvec[i] * nrvec[i];
                    E_nr = log(E * 2.);
                }
            }
        }
        memcpy(c->last_power, Xabs, sizeof(double) * K);
        memcpy(c->last_fbank, Y, sizeof(double) * B);

        /* ---- FEA */
        double E_fea = -1.;
        if (!strcmp(kind, "spec") || !strcmp(kind, "logspec") || is_trap) { /* fea_impl.cc:37-77 */
            for (int i = 0; i < B; i++) fvec[i] = (!strcmp(kind, "spec")) ? Y[i] : log(Y[i]);
            if (o->fea_E && !o->fea_rawenergy) {
                double E = Y[0] * Y[0] / 2. + Y[B - 1] * Y[B - 1] / 2.;
                for (int i = 1; i < B - 1; i++) E += Y[i] * Y[i];
                E_fea = log(E * 2.);
            }
        } else if (!strcmp(kind, "dctc")) { /* fea_impl.cc:104-131 */
            int Nout = o->fea_ncepcoefs + 1;
            for (int i = 0; i < B; i++) Y[i] = log(Y[i]);
            for (int i = 0; i < Nout; i++) {
                double acc = 0;
                for (int k = 1; k <= B; k++) acc += Y[k - 1] * c->wdct[(2 * k - 1) * i % (4 * B)];
                fvec[i] = acc * c->normcoef;
            }
            if (o->fea_lifter > 1) for (int n = 1; n < Nout; n++) fvec[n] *= c->lift[n - 1];
        } else { /* lpa / lpc, fea_impl.cc:163-222, 251-284 */
            int Nfull = (B - 1) * 2;
            if (!o->fb_inld) for (int i = 0; i < B; i++) Y[i] *= Y[i];
            for (int k = 0; k <= lporder; k++) {
                double r = Y[0] / 2.;
                for (int n = 1; n < B - 1; n++) r += Y[n] * c->WRe[(n * k) % Nfull];
                r += (1 - 2 * (k % 2)) * Y[B - 1] / 2.;
                r /= (double)Nfull / 2;
                RRe[k] = r;
            }
            P[0] = RRe[0];
            rc[1] = -RRe[1] / RRe[0];
            P[1] = P[0] * (1 - rc[1] * rc[1]);
            aa[1] = rc[1];
            a[0] = aa[0] = 1;
            for (int ik = 2; ik <= lporder; ik++) {
                double dm = RRe[ik];
                for (int n = 1; n <= ik - 1; n++) dm += aa[n] * RRe[ik - n];
                rc[ik] = -dm / P[ik - 1];
                a[ik] = rc[ik];
                for (int n = 1; n <= ik - 1; n++) a[n] = aa[n] + rc[ik] * aa[ik - n];
                for (int n = 1; n <= ik; n++) aa[n] = a[n];
                P[ik] = P[ik - 1] * (1 - rc[ik] * rc[ik]);
            }
            E_fea = log(RRe[0]);
            if (!strcmp(kind, "lpa")) {
                for (int i = 0; i <= lporder; i++) fvec[i] = a[i];
            } else {
                int Nout = o->fea_ncepcoefs + 1;
                fvec[0] = log(P[lporder]);
                for (int n = 1; n < Nout; n++) {
                    double sum = 0;
                    if (n <= lporder) {
                        for (int k = 1; k <= n - 1; k++) sum += (n - k) * fvec[n - k] * a[k];
                        fvec[n] = -a[n] - sum / n;
                    } else {
                        for (int k = 1; k <= lporder; k++) sum += (n - k) * fvec[n - k] * a[k];
                        fvec[n] = -sum / n;
                    }
                }
                if (o->fea_lifter > 1) for (int n = 1; n < Nout; n++) fvec[n] *= c->lift[n - 1];
            }
        }
        /* energy routing, batch.cc:70-120 */
        double E_out;
        if (!do_vad && o->fea_rawenergy) E_out = E_in;
        else if (!strcmp(kind, "dctc")) E_out = o->nr_when_afterFB ? E_in : E_nr;
        else E_out = E_fea;

        if (do_vad) {
            /* criterion inputs of this input frame: VADcri_energy (vad.cc:96-107) / VADcri_cepdist lpc (vad.cc:220-276) */
            if (!strcmp(o->vad_cri_mode, "energy")) {
                double en = 0.0;
                for (int i = 0; i < K; i++) en += Xabs[i] * Xabs[i];
                if (o->vad_energy_db) en = 10.0 * log10(2.2250738585072014e-308 + en);
                cur_en = en;
            } else if (!strcmp(o->vad_cri_mode, "cepdist")) {
                if (!strcmp(o->vad_cepdist_mode, "lpc")) {
                    for (int i = 0; i < K; i++) { hre[i] = Xabs[i] * cos(Xph[i]); him[i] = Xabs[i] * sin(Xph[i]); }
                    hc2r(hre, him, wfft, tsig, hw1, hw2);
                    ctuo_burg_cepstrum(tsig, window, vs.csize, NULL, vs.ci, NULL);
                } else if (strcmp(o->vad_cepdist_mode, "fea")) { set_err(c, "VADcri_cepdist: vad_cepdist_mode=in and in->_fvec is not available!"); fail = 1; break; }
            } else { set_err(c, "VAD: unknown vad_cri_mode!"); fail = 1; break; }
        }
        if (is_trap) { /* log-mel kept; TRAP computed after the loop (edge replication, fea_trap.cc:53-127) */
            memcpy(trapbuf + (size_t)t * B, fvec, sizeof(double) * B);
            trapE[t] = E_out;
            if (do_vad) { /* as behind a delta chain: the detector runs when a vector comes out, on the newest input frame's criterion */
                if (!strcmp(o->vad_cri_mode, "cepdist") && !strcmp(o->vad_cepdist_mode, "fea")) {
                    set_err(c, "oracle: the `fea` VAD criterion on TRAP vectors is not restated");
                    fail = 1;
                    break;
                }
                en_t[t] = cur_en;
                if (ci_t) memcpy(ci_t + (size_t)t * vs.csize, vs.ci, sizeof(double) * vs.csize);
            }
            continue;
        }
        if (is_post) { /* deltaFEA chain replayed after the loop (it only consumes fvec and E) */
            memcpy(postbuf + (size_t)t * c->nfea, fvec, sizeof(double) * c->nfea);
            postE[t] = E_out;
            if (do_vad) { /* the detector is called when a delayed vector comes out, on the newest input frame's criterion */
                en_t[t] = cur_en;
                if (ci_t) memcpy(ci_t + (size_t)t * vs.csize, vs.ci, sizeof(double) * vs.csize);
            }
            continue;
        }

        E_last = E_out;
        /* ---- BATCH::save_frame, batch.cc:230-241 */
        if (!do_vad) {
            if (cms.type) cms_apply(&cms, o, fvec); /* BATCH::cmvn_stat, batch.cc:198-200: post, then save */
            emit_row(c, fvec, E_out, rows + (size_t)nrows * c->D);
            nrows++;
            continue;
        }
        if (cms.type) cms_apply(&cms, o, fvec); /* BATCH::cmvn_stat, batch.cc:198-200: post, then save (the VAD sees the vector after it) */
        VAD_CALL(t, fvec, E_out);
    }

    if (!fail && is_trap) {
        /* trapdctFEA: output frame t uses log-mel frames t-50..t+50 with first/last
         * frame replication (fea_trap.cc:64-70,111-127); mean removal, Hamming, REDFT10,
         * keep k=1..ndct, band-major layout (fea_trap.cc:83-107). */
        int half = htrap - 1;
        for (long t = 0; t < T; t++) {
            for (int b = 0; b < B; b++) {
                double sum = 0;
                for (int j = 0; j < traplen; j++) {
                    long u = t - half + j;
                    if (u < 0) u = 0;
                    if (u > T - 1) u = T - 1;
                    tin[j] = trapbuf[(size_t)u * B + b];
                    sum += tin[j];
                }
                sum /= traplen;
                for (int j = 0; j < traplen; j++) tin[j] = (tin[j] - sum) * c->trap_hamm[j];
                for (int kk = 1; kk <= ndct; kk++) {
                    double acc = 0;
                    for (int j = 0; j < traplen; j++) acc += tin[j] * cos(3.14159265358979323846 * (j + 0.5) * kk / traplen);
                    fvec[b * ndct + (kk - 1)] = 2.0 * acc;
                }
            }
            /* E is read through a pointer at save time: it belongs to the newest frame
             * fed in, i.e. frame min(t+half, T-1) (batch.cc:116, fea_trap.cc:76-79). */
            long te = t + half; if (te > T - 1) te = T - 1;
            if (do_vad) {
                /* trapdctFEA::process_frame is false for the first `half` frames and flush_frame delivers the last `half` vectors
                 * (fea_trap.cc:53-127), so BATCH::save_frame's call t (batch.cc:230-241) finds input frame min(t + half, T - 1) in
                 * in->_Xsabs: the criterion the detector reads is that frame's */
                cur_en = en_t[te];
                if (ci_t) memcpy(vs.ci, ci_t + (size_t)te * vs.csize, sizeof(double) * vs.csize);
                VAD_CALL(t, fvec, trapE[te]);
                E_last = trapE[te];
                continue;
            }
            emit_row(c, fvec, trapE[te], rows + (size_t)nrows * c->D);
            nrows++;
        }
    }

    if (c->signal_out) { /* rawOUT::close / waveOUT::close, out.cc:487-491,539-541: the last window-wshift samples */
        for (int i = 0; i < window - wshift; i++) {
            int value = (int)floor(ola[(ola_start + i) % window] / c->ola_corr);
            c->sig_buf[nout++] = (fabs((float)value) > 32767) ? (value < 0 ? -32767 : 32767) : (int16_t)value;
        }
        nrows = nout;
    }
    if (!fail && is_post && T > 0) {
        /* BATCH::fea_delta (batch.cc:172-192) per frame, then BATCH::flush_fea (batch.cc:251-291).  The writers read
         * E through a pointer, so a row carries the energy of the newest frame fed in, not of the frame it describes. */
        const int n = c->post_order;
        dstage_t st[3];
        double *in0 = calloc(c->nfea, sizeof(double));
        dstage_init(&st[0], o, in0, 2, o->d_win);
        if (n >= 2) dstage_init(&st[1], o, st[0].out, 3, o->a_win);
        if (n >= 3) dstage_init(&st[2], o, st[1].out, 4, o->t_win);
        const double *X = st[n - 1].out;
        double E_cur = -1.;
        long tcur = 0, ncall = 0; /* newest input frame; save_frame calls so far (the VAD's frame_index) */
#define POST_EMIT() do {                                                                                   \
        if (cms.type) cms_apply(&cms, o, (double *)X);                                                     \
        if (do_vad) {                                                                                      \
            cur_en = en_t[tcur];                                                                           \
            if (ci_t) memcpy(vs.ci, ci_t + (size_t)tcur * vs.csize, sizeof(double) * vs.csize);            \
            VAD_CALL(ncall, X, E_cur);                                                                     \
            ncall++;                                                                                       \
        } else {                                                                                           \
            emit_row(c, X, E_cur, rows + (size_t)nrows * c->D);                                            \
            nrows++;                                                                                       \
        }                                                                                                  \
    } while (0)
        for (long t = 0; t < T; t++) {
            memcpy(in0, postbuf + (size_t)t * c->nfea, sizeof(double) * c->nfea);
            E_cur = postE[t];
            tcur = t;
            if (!dstage_process(&st[0])) continue;
            if (n >= 2) {
                if (!dstage_process(&st[1])) continue;
                if (n >= 3 && !dstage_process(&st[2])) continue;
            }
            POST_EMIT();
        }
        while (dstage_flush(&st[0])) {
            if (n >= 2) {
                if (!dstage_process(&st[1])) continue;
                if (n >= 3 && !dstage_process(&st[2])) continue;
            }
            POST_EMIT();
        }
        if (n >= 2)
            while (dstage_flush(&st[1])) {
                if (n >= 3 && !dstage_process(&st[2])) continue;
                POST_EMIT();
            }
        if (n >= 3)
            while (dstage_flush(&st[2])) POST_EMIT();
#undef POST_EMIT
        for (int j = 0; j < n; j++) dstage_free(&st[j]);
        free(in0);
        E_last = postE[T - 1];
    }

    if (!fail && do_vad) { /* BATCH::flush_vad, batch.cc:243-249; medianFilter::flush_frame, vad.h:156-175 */
        /* `while (vad->flush_frame())` runs on the filter's `ready` (vad.cc:742-745), which only a push that found the history
         * full sets (vad.h:126-136): a file with no more frames than the filter delays - (order-1)/2 - never gets there, the loop
         * does not start, and neither a row nor a decision is written for it (pinned by tests/test_oracle_median_ref.py) */
        for (;;) { /* `while (vad->flush_frame())`: VAD::flush_frame calls the filter's flush_frame and returns its `ready` */
            int dec = 0;
            const double *fo = NULL;
            if (vs.hsize > 0) {
                vs.history[vs.hidx] = 0;
                vs.hidx = (vs.hidx + 1) % order;
                double sum = 0.0;
                for (int i = 0; i < order; i++) sum += vs.history[i] ? 1.0 : 0.0;
                fo = vs.ring + (size_t)(vs.start % order) * vlen;
                vs.start++;
                dec = (sum / (double)order) >= 0.5;
                vs.hsize--;
            } else vs.ready = 0;
            if (!vs.ready) break; /* also for a filter that never got ready: one flush_frame ran, nothing is written */
            if (vadout && strcmp(o->vad_out_mode, "none")) vadout[nvad] = dec ? '1' : '0';
            nvad++;
            if (!(!dec && !strcmp(o->vad_apply_mode, "drop"))) {
                /* E pointer still holds the last frame's energy */
                emit_row(c, fo, E_last, rows + (size_t)nrows * c->D);
                nrows++;
            }
        }
        c->med_hidx = vs.hidx; /* VAD::clean() leaves both as they are */
        c->med_hsize = vs.hsize;
    }

    free(x); free(fft_in); free(Xre); free(Xim); free(zr); free(zi); free(Xabs); free(Xph); free(Y); free(fvec);
    ctuo_cepdet_free(ss_det); free(ss_nr); free(ss_re); free(ss_im); free(ss_t); free(ss_w1); free(ss_w2);
    free(Navg); free(Yavg); free(trapbuf); free(trapE); free(tin); free(postbuf); free(postE); free(en_t); free(ci_t);
    cms_free(&cms);
    free(ola); free(ytime); free(sw1); free(sw2); free(sre); free(sim);
    free(RRe); free(rc); free(a); free(aa); free(P);
    free(vs.history); free(vs.ring); free(vs.c0); free(vs.ci); free(hw1); free(hw2); free(tsig); free(hre); free(him);
    return fail ? -1 : nrows;
}
