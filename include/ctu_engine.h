/*
 * ctu_engine.h -- C ABI of the MI355X (gfx950) framewise feature engine.
 *
 * This is the drop-in boundary for CtuCopy's per-frame hot path.  The reference has no FFI; its
 * "operator API" is the set of per-stage objects BATCH wires together by pointer and drives once per
 * frame (src/io/batch.cc:24-69 wiring, :205-228 per-frame dispatch, :326-421 list loop).  The calls
 * below replace, for a whole list of files at once:
 *
 *   new opts(argc,argv) + new BATCH(o)        -> ctu_engine_create        (src/io/opts.cc:27-194, src/io/batch.cc:24-69)
 *   scanning the -S list / IN::new_file       -> ctu_plan_create          (src/io/batch.cc:349-371, src/io/in.cc:264-279)
 *   while(in->get_frame()) process_frame();   -> ctu_engine_run[_host]    (src/io/batch.cc:402-407 incl. flush_fea)
 *     rawIN::get_frame after loadframe           (src/io/in.cc:343-417)
 *     NR::process_frame                          (src/nr/nr.cc:95-140)
 *     FB::project_frame                          (src/fea/fb.cc:72-86)
 *     FEA::process_frame / flush_frame           (src/fea/fea_impl.cc:104-131,163-284, src/fea/fea_trap.cc:53-127)
 *     VAD::process_frame .. flush                (src/vad/vad.cc:692-745, src/io/batch.cc:230-249)
 *     htkOUT::save_frame reorder + (float)       (src/io/out.cc:174-203; the fwrite stays with the caller)
 *   throw "literal" caught in main()          -> negative return + ctu_last_error (src/main.cpp:54-60)
 *
 * File decoding (rawIN::loadframe, a-law/mu-law, WAVE headers) and the HTK/pfile/ark writers stay on
 * the caller's side of this ABI ("src/io left as-is"); the bundled `ctucopy` executable provides them.
 *
 * Plain pointers and sizes only.  Device pointers are HIP device addresses on the engine's device.
 * There is no CPU fallback: every entry point that computes fails with CTU_ERR_DEVICE when no gfx950
 * device is usable.
 */
#ifndef CTU_ENGINE_H
#define CTU_ENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ctu_engine ctu_engine;
typedef struct ctu_plan ctu_plan;

enum {
    CTU_OK = 0,
    CTU_ERR_OPTS = -1,        /* command line rejected (message = the reference's text) */
    CTU_ERR_UNSUPPORTED = -2, /* valid ctucopy configuration outside the accelerated path */
    CTU_ERR_DEVICE = -3,      /* no usable HIP device / HIP runtime error */
    CTU_ERR_INPUT = -4,       /* bad argument, or "IO: Signal shorter than one frame!" */
};

/* Geometry of a configured chain (all derived as in the reference, see each field). */
typedef struct {
    int32_t fs;
    int32_t window;     /* src/io/opts.cc:259 */
    int32_t wshift;     /* src/io/opts.cc:260 */
    int32_t wfft;       /* src/io/opts.cc:277-280 */
    int32_t nbins;      /* wfft/2+1, src/io/opts.cc:283 */
    int32_t nbands;     /* src/fea/fb.cc:183,289-302 */
    int32_t row_floats; /* floats per output row incl. c0/E, src/io/out.cc:95-113 */
    int32_t htk_kind;   /* src/io/out.cc:147-159 */
    uint32_t htk_period;/* src/io/out.cc:146 */
    int32_t has_vad;    /* a VAD byte per frame is produced, src/io/batch.cc:34-38 */
    int32_t swap_out;   /* caller must byte-swap on write (-endian_out big), src/io/opts.cc:287 */
    int32_t pcm_align;  /* utterance starts inside the packed PCM arena are multiples of this many samples */
    int32_t signal_out; /* -format_out raw|wave: the engine writes enhanced speech (ctu_engine_run_signal), no rows */
} ctu_dims;

/* argv = the ctucopy command line without argv[0] (flags of src/io/opts.cc:644-846, incl. -C <file>).
 * device = HIP ordinal.  On failure *out is NULL and ctu_create_error() holds the message. */
int ctu_engine_create(int argc, const char *const *argv, int device, ctu_engine **out);
void ctu_engine_destroy(ctu_engine *);
const char *ctu_create_error(void);
const char *ctu_last_error(const ctu_engine *);
int ctu_engine_dims(const ctu_engine *, ctu_dims *out);

/* Parse + design only (no device needed): used by host-side tools and CPU tests. */
int ctu_config_dims(int argc, const char *const *argv, ctu_dims *out);

/* Host-designed tables in double precision (no device needed), for inspection and tests:
 * name = "hamming" [window] | "fbank" [nbands*nbins] | "fb_first" | "fb_last" [nbands] |
 *        "dct" [(ncep+1)*nbands] | "idft" [(p+1)*nbands] | "trap" [ndct*traplen] | "lifter" [ncep].
 * Returns the number of values (written up to cap), or a negative error code. */
int64_t ctu_config_table(int argc, const char *const *argv, const char *name, double *out, int64_t cap);

/* floor((n - (window-wshift)) / wshift); -1 when n < window-wshift (src/io/in.cc:277,314). */
int64_t ctu_num_frames(const ctu_engine *, int64_t nsamples);

/* A plan fixes the batch layout: utterance i has utt_nsamples[i] samples and lives in the packed
 * int16 arena at sample offset ctu_plan_sample_offsets()[i] (offsets are multiples of pcm_align; the
 * arena has 8 samples of padding ahead of the first utterance and 512 behind the last one, and frames DO read past the
 * end of their window - into the gap, the next utterance or that padding - under zero weights: every byte of
 * ctu_plan_total_samples() must be readable, its contents between utterances do not matter).  Its output rows start at row ctu_plan_row_offsets()[i].  Both offset arrays
 * have n_utt+1 entries and stay valid until ctu_plan_destroy. */
int ctu_plan_create(ctu_engine *, const int64_t *utt_nsamples, int32_t n_utt, ctu_plan **out);
/* The arena layout rule of ctu_plan_create on its own (no engine, no device): writes the n_utt+1 sample offsets a plan over
 * these lengths will report (sample_off may be NULL) and returns its ctu_plan_total_samples(), or a negative error code.  A host
 * loop that reads files straight into a page-locked arena (the counterpart of rawIN::loadframe, src/io/in.cc:434-460) lays the
 * next batch out with it while the engine is busy with the current one. */
int64_t ctu_arena_layout(const int64_t *utt_nsamples, int32_t n_utt, int64_t *sample_off);
void ctu_plan_destroy(ctu_plan *);
const int64_t *ctu_plan_sample_offsets(const ctu_plan *);
const int64_t *ctu_plan_row_offsets(const ctu_plan *);
int64_t ctu_plan_total_samples(const ctu_plan *);
int64_t ctu_plan_total_frames(const ctu_plan *);

/* Device-resident run: d_pcm = packed arena (int16, total_samples), d_rows = total_frames*row_floats
 * floats in writer order (c1..cN,c0[,E]), d_vad = total_frames bytes '0'/'1' or NULL.
 * stream = hipStream_t (NULL = default stream).  Asynchronous: returns after enqueueing. */
int ctu_engine_run(ctu_engine *, const ctu_plan *, const int16_t *d_pcm, float *d_rows, uint8_t *d_vad, void *stream);

/* Page-locked host memory for the host-buffer calls below: buffers from ctu_host_alloc are DMA-ed asynchronously at the
 * link rate; any other (pageable) pointer is accepted too and goes through the runtime's staging. */
void *ctu_host_alloc(size_t bytes);
void ctu_host_free(void *);

/* Host-buffer convenience: H2D, run, D2H, synchronised on return.  The device copies are kept in the plan.  Batches of at
 * least 16 utterances and 32 MiB in page-locked buffers go in eight utterance ranges on two streams, so the upload of a
 * range overlaps the kernels and the download of the previous one (environment CTU_HOST_CHUNKS=<n> overrides, 1 = one
 * range; the hwss / fwss / 2fwss chain always runs in one).  rows_per_utt (optional, n_utt entries) receives the number
 * of rows actually produced per utterance: < frames with -vad_apply_mode drop, and 0 for an utterance with no more frames than the VAD's
 * majority filter delays ((vad_filter_order-1)/2: the filter never gets ready, src/vad/vad.h:126-136, and the reference writes neither a
 * row nor a decision; the utterance's bytes in h_vad / d_vad are NUL then, '0' / '1' otherwise). */
int ctu_engine_run_host(ctu_engine *, const ctu_plan *, const int16_t *h_pcm, float *h_rows, uint8_t *h_vad,
                        int64_t *rows_per_utt);

/* -format_in alaw | mulaw on the device: n G.711 codes -> n int16 samples, the expansion of src/io/amulaw.h:20-53
 * (alaw2lin, called from src/io/in.cc:481-560) bit for bit.  The caller lays the codes of an utterance at the same offsets
 * as its samples (ctu_plan_sample_offsets) and may decode the whole arena at once.  Asynchronous on `stream`. */
int ctu_decode_g711(ctu_engine *, const uint8_t *d_codes, int64_t n, int alaw, int16_t *d_pcm, void *stream);

/* hwss / fwss / 2fwss (-nr_mode, with -vad burg): hwssNR::new_file seeds a file's noise estimate from the spectrum vector
 * as the previous file left it (src/nr/nr.cc:212-221), so the list is one chain.  A run processes its utterances in plan
 * order as that chain (synchronously: the seeds are iterated to their fixed point) and the engine keeps the last vector
 * for the next run, as the reference keeps it for the life of the process; this call forgets it (a new process). */
int ctu_engine_reset_chain(ctu_engine *);
/* -vad file=<f> with those modes (hwssNR::vad_get_frame, src/nr/nr.cc:297-302): the decisions come from ONE byte stream for all files
 * of the process, one byte per frame in list order - every byte but NUL is speech, a byte 0xFF ends the run like the end of the
 * stream does ("NR: Unexpected end of VAD file!").  ctu_engine_create reads <f> (src/nr/nr.cc:205-209); this call replaces the
 * stream and rewinds it, ctu_engine_reset_chain rewinds it. */
int ctu_engine_set_vad_stream(ctu_engine *, const unsigned char *bytes, int64_t n);

/* The VAD's majority filter belongs to the process, not to the file: VAD::clean() = medianFilter::cleanFilter() resets `start`, the ring
 * and the decisions at the end of a file but neither historyIdx nor historySize (src/vad/vad.h:110-121), so the vectors of the next file
 * land in ring slots out of phase with the slots its rows are read from - its rows come out shifted by a frame or two, with an all-zero
 * or a repeated row - depending on the frame counts of the files in front of it (decisions are not affected).  By default every
 * utterance of a plan is treated as the first file of its own process (in phase).  A host that wants the reference's list behaviour
 * walks its list with ctu_vad_ring_step (pure: the state after a file of `frames` frames; start from 0, 0) and hands each plan the
 * historyIdx its utterances start with; the run then delivers the rows the reference would write.  bin/ctucopy does.  Not reproduced:
 * a historySize left over by a file with no more frames than the delay, at filter orders of 5 and more. */
void ctu_vad_ring_step(int32_t order, int64_t frames, int32_t *hidx, int32_t *hsize);
int ctu_plan_set_vad_ring(ctu_plan *, const int32_t *hidx /* n_utt entries, or NULL: all in phase */);
/* The same mapping for one file on its own (pure; what ctu_plan_set_vad_ring applies on the device): the number of rows the file writes - its
 * frames, or 0 when it has no more frames than the filter delays - and, if src is not NULL, for each of them the frame whose vector it carries
 * (-1: an untouched ring slot, zeros).  For hosts that pass rows through stages of their own between the engine and the writer (CMVN). */
int64_t ctu_vad_ring_rows(int32_t order, int64_t frames, int32_t hidx0, int32_t *src);

/* Timing of the last ctu_engine_run on this engine, measured with HIP events on the run's stream
 * around the dominant (front-end) kernel; blocks until that run has finished.  Returns < 0 if none. */
float ctu_engine_last_kernel_ms(ctu_engine *);
/* Name of that kernel: the instantiation this engine's configuration runs its front end on (profiles and the bench's roofline
 * line are keyed by it).  Valid for the life of the engine. */
const char *ctu_engine_kernel_name(const ctu_engine *);

/* ---- per-speaker CMVN over rows that are resident on the device -----------------------------------------------
 * Replaces cmvn_POST::sum_fea / stat_cm / sum_cv / stat_cv / process_frame (src/fea/post_impl.cc:51-118) and the
 * three passes over the list that BATCH::process makes for them (src/io/batch.cc:331-419): with the corpus' rows
 * kept in HBM the passes run over rows, not over audio.  The caller owns the speaker table (add_spk,
 * post_impl.cc:120-142: list order of first appearance) and, with several GPUs, sums `acc` over ranks.
 *
 * Statistics are in the reference's own order (the order of the -stat_cmvn file, src/io/out.cc:591-613):
 * slot k holds internal vector entry k+1, the last slot holds entry 0 (c0 of the base block); the energy column is
 * not part of the vector.  ctu_cmvn_cols = number of slots. */
int ctu_cmvn_cols(const ctu_engine *);

/* acc[n_spk][cols+1] (host, double) += per-speaker sums over the plan's rows and, in the last slot, the frame count.
 * mean == NULL: sums of the values (pass 1);  mean != NULL ([n_spk][cols]): sums of (value - mean)^2 (pass 2).
 * Synchronises on `stream`. */
int ctu_cmvn_accumulate(ctu_engine *, const ctu_plan *, const float *d_rows, const int32_t *spk_of_utt, int32_t n_spk,
                        const double *mean, double *acc, void *stream);

/* rows = (rows - mean) / var in place - the reference divides by the variance, not by its root
 * (src/fea/post_impl.cc:104-118).  mean, var: [n_spk][cols] (host, double).  Asynchronous on `stream`. */
int ctu_cmvn_apply(ctu_engine *, const ctu_plan *, float *d_rows, const int32_t *spk_of_utt, int32_t n_spk,
                   const double *mean, const double *var, void *stream);

/* ---- speech enhancement output (-format_out raw|wave; e.g. -preset exten) ----------------------------------
 * Replaces `while(in->get_frame()) { nr->process_frame(); out->save_frame(); }` plus the writer's close() for every
 * file: src/io/batch.cc:223-227,402-407 with sigOUT::save_frame / fill_cache (src/io/out.cc:405-451) and
 * rawOUT::close (out.cc:487-491).  d_out is an int16 arena laid out like the PCM arena: utterance i's samples start
 * at ctu_plan_sample_offsets()[i] and ctu_plan_out_samples()[i] = frames*wshift + window - wshift of them are
 * written (host byte order; the RIFF header and -endian_out are the caller's).  Asynchronous on `stream`. */
const int64_t *ctu_plan_out_samples(const ctu_plan *);
int ctu_engine_run_signal(ctu_engine *, const ctu_plan *, const int16_t *d_pcm, int16_t *d_out, void *stream);
int ctu_engine_run_signal_host(ctu_engine *, const ctu_plan *, const int16_t *h_pcm, int16_t *h_out);

/* Host-buffer conveniences (H2D of the rows, the call above, D2H for apply; synchronised on return) for callers
 * that hold the rows in host memory, like the `ctucopy` executable. */
int ctu_cmvn_accumulate_host(ctu_engine *, const ctu_plan *, const float *h_rows, const int32_t *spk_of_utt, int32_t n_spk,
                             const double *mean, double *acc);
int ctu_cmvn_apply_host(ctu_engine *, const ctu_plan *, float *h_rows, const int32_t *spk_of_utt, int32_t n_spk,
                        const double *mean, const double *var);

#ifdef __cplusplus
}
#endif
#endif
