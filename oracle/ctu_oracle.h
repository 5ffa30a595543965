/*
 * ctu_oracle.h -- CPU oracle for the CtuCopy per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a float64 restatement, written from
 * scratch, of the algorithm in pmizera/ctucopy 4.0.2 (paths cited as
 * src/...:line are relative to the reference tree).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and
 * only as the checker / reported CPU baseline -- never as the product path.
 *
 * PARITY PIN STATUS.  The reference cannot be built in this image: every
 * translation unit includes <fftw3.h> (src/io/stdafx.h:30) and FFTW3 is not
 * installed; the reference ships no tests and no golden vectors.  What pins
 * this oracle to the reference are the outputs of the compiled reference that
 * SURVEY.md section 8(c) / Appendix A recorded at survey time (frame counts
 * 594/592, HTK header (period 100000, 52 bytes, kind 8198), file sizes
 * 30900/30796, kind codes 020013/020011/020106, ark offsets 5/30913, pfile
 * size 103940, Burg-VAD 626 ones out of 1186) -- see tests/test_oracle_pins.py
 * -- plus tests/test_oracle_burg_ref.py, which compiles the reference's
 * FFTW-free header src/vdet/Burg.h in place into oracle/_ref/ and compares.
 * Cepstral VALUES are otherwise "parity unpinned": no reference-produced
 * value vector exists in this container.
 *
 * Third-party arithmetic restated here: FFTW3 (version unpinned by the
 * reference, src/objects.mk:7) r2r kinds R2HC, HC2R (unnormalised DFT pair,
 * halfcomplex layout r0..r_{n/2}, i_{n/2-1}..i_1) and REDFT10
 * (Y_k = 2 sum_j x_j cos(pi (j+1/2) k / n)).
 */
#ifndef CTU_ORACLE_H
#define CTU_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctuo ctuo_t;

typedef struct {
    int fs;
    int window;        /* samples per frame            (src/io/opts.cc:259) */
    int wshift;        /* hop                          (src/io/opts.cc:260) */
    int wfft;          /* FFT size                     (src/io/opts.cc:277-280) */
    int K;             /* wfft/2+1                     (src/io/opts.cc:283) */
    int B;             /* filter bank size             (src/fea/fb.cc:289-302,183) */
    int nfea;          /* internal feature vector size (fvec) */
    int D;             /* floats per output row        (src/io/out.cc:95-113) */
    int htk_kind;      /* HTK parameter kind           (src/io/out.cc:147-159) */
    unsigned period;   /* HTK sample period, 100 ns    (src/io/out.cc:146) */
    int do_vad;        /* src/io/batch.cc:34-38 */
    int phase_needed;  /* src/io/opts.cc:290-294 */
    int fb_power;
    int swap_out;      /* src/io/opts.cc:287 */
} ctuo_dims_t;

/* argv: the ctucopy command line WITHOUT argv[0]; parsed as src/io/opts.cc:158-193. */
ctuo_t *ctuo_create(int argc, const char *const *argv, char *err, int errlen);
void ctuo_destroy(ctuo_t *);
void ctuo_get_dims(const ctuo_t *, ctuo_dims_t *);

/* floor((N-(window-wshift))/wshift); -1 when N < window-wshift
 * ("IO: Signal shorter than one frame!", src/io/in.cc:277). */
long ctuo_num_frames(const ctuo_t *, long nsamples);

/* One utterance (one line of the -S list).  rows: capacity num_frames*D floats,
 * written in writer order (src/io/out.cc:174-203) after the (float) cast.
 * vad: optional, one '0'/'1' char per frame as written to the VAD file
 * (src/vad/vad.h:67-70).  Returns the number of rows written (fewer than
 * num_frames only with -vad_apply_mode drop), or -1 (message via ctuo_error). */
long ctuo_process(ctuo_t *, const int16_t *pcm, long nsamples, float *rows, unsigned char *vad);

/* -format_out raw|wave (speech enhancement, src/io/out.cc:346-451): `out` receives ctuo_out_samples(nsamples) int16
 * samples - what rawOUT / waveOUT write for one file, before byte order and the RIFF header.  Returns their number. */
long ctuo_out_samples(const ctuo_t *, long nsamples);
long ctuo_enhance(ctuo_t *, const int16_t *pcm, long nsamples, int16_t *out);

/* The majority filter's historyIdx / historySize as the previous file of the process left them (src/vad/vad.h:110-121: cleanFilter does
 * not reset them); (0, 0) = the first file of a process.  ctuo_process carries them from call to call. */
void ctuo_set_vad_ring(ctuo_t *, int hidx, int hsize);
void ctuo_get_vad_ring(const ctuo_t *, int *hidx, int *hsize);

const char *ctuo_error(const ctuo_t *);

/* Introspection used by the host-design tests. */
const double *ctuo_hamming(const ctuo_t *);                       /* W[window]   src/io/in.cc:139-144 */
int ctuo_fb_row(const ctuo_t *, int b, const double **w, int *first, int *last); /* src/fea/fb.cc:432-447 */
float ctuo_preem(const ctuo_t *);
/* Intermediate taps of the LAST frame processed (debug / unit tests). */
const double *ctuo_last_power(const ctuo_t *);  /* K values after NR */
const double *ctuo_last_fbank(const ctuo_t *);  /* B values after FB (before log) */

/* Stand-alone Burg lattice + a->c, restating src/vdet/Burg.h:49-152 (for the pin test). */
void ctuo_burg_cepstrum(const double *x, int npoints, int ncoefs, double *a_out, double *c_out, double *alpha_out);

/* Stand-alone CepstralDetector<BurgCepstrumEstimator>, restating src/vdet/CepstralDet.h:92-217 (for the pin test
 * and for the hwss / fwss / 2fwss noise-reduction modes, src/nr/nr.cc:263-295). */
typedef struct ctuo_cepdet ctuo_cepdet_t;
ctuo_cepdet_t *ctuo_cepdet_new(int npoints, int ninit, int ncoefs, double p, double q);
int ctuo_cepdet_process(ctuo_cepdet_t *, const double *frame);   /* 1 = speech */
double ctuo_cepdet_last_distance(const ctuo_cepdet_t *);
void ctuo_cepdet_free(ctuo_cepdet_t *);

#ifdef __cplusplus
}
#endif
#endif
