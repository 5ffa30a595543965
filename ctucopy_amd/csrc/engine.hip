// MI355X (gfx950) engine: host side of the C ABI (include/ctu_engine.h) - table design for the kernels, plans
// (batch layout, tile chains), launches.  The kernels live in the headers included below, in one translation unit:
//   kernel_common.h    constants, the kernel parameter block, DPP / LDS helpers, tile records
//   frontend_kernel.h  PCM -> pre-emphasis * Hamming -> 512/256-pt real FFT in registers -> |.|^2 (P tile in LDS)
//                      -> exten NR -> banded filter bank -> ^0.33 / log -> DCT-II + lifter | cosine iDFT + Levinson
//   vad_kernels.h      Burg-cepstral criterion (packed inverse FFT + lattice), the detector's recurrences per utterance (one wave, or one
//                      lane for the fused path)
//   lp_tail_kernel.h   Levinson-Durbin and a -> c, one frame per lane
//   wave1k_kernel.h    1024-point frames, one wave per frame
//   trap_kernel.h      TRAP-DCT as fp32 MFMA Toeplitz contraction
//   post_kernels.h     delta chain / stacking, CMS, per-speaker CMVN over resident rows
//   signal_kernels.h   speech-enhancement output: inverse transform, overlap-add
//
// Data layout in HBM
//   pcm   : one packed int16 arena; utterance i starts at sample_off[i] (multiple of 8 samples)
//   rows  : float32 [total_frames][D] in writer order (c1..cN, c0[, E]), utterance i at row_off[i]
//   tiles : 32-byte records; a tile is <= 64 consecutive frames of ONE utterance, workgroups walk chains of tiles
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <queue>
#include <set>
#include <string>
#include <vector>

#include "../../include/ctu_engine.h"
#include "design.h"
#include "opts.h"


#include "kernel_common.h"
#include "vad_kernels.h"
#include "vad_fused.h"
#include "frontend_kernel.h"
#include "trap_kernel.h"
#include "decode_kernels.h"
#include "bigfft_kernel.h"
#include "lp_tail_kernel.h"
#include "wave1k_kernel.h"

namespace {

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local std::string g_create_error;

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    void upload(const std::vector<T> &h) {
        release();
        n = h.size();
        if (!n) return;
        HIP_TRY(hipMalloc(&p, n * sizeof(T)));
        HIP_TRY(hipMemcpy(p, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
    }
    void alloc(size_t count) {
        release();
        n = count;
        if (n) HIP_TRY(hipMalloc(&p, n * sizeof(T)));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

}  // namespace

#include "post_kernels.h"
#include "signal_kernels.h"

struct ctu_engine {
    std::unique_ptr<ctu::Design> design;
    int device = 0;
    int n_cu = 256;
    int user_wfft = 0, user_K = 0;
    int kstride = 1;            // > 1: an FFT size below 256 carried by the 256-point mode (ctu_engine_create)
    std::string err, kname;
    int feat = FEAT_DCTC;
    int nz = 16;
    int mode = 0;  // 0: 512-point FFT, 1: 256-point FFT (two frames per complex transform)
    DevBuf<float> lanec, ftab, trapG;
    DevBuf<uint4> trapG16;   // TRAP on the bf16 matrix pipe: A fragments [8 phases][4 k-steps][3 terms][64 lanes] (trap_kernel.h)
    bool trap_bf16 = false;
    DevBuf<int> itab;
    // FFT sizes of 1024 to 4096 points (bigfft_kernel.h)
    bool big = false;
    bool wave1k = false;        // 1024-point frames on wave1k_kernel (one wave per frame); CTU_WAVE1K=0 keeps them on bigfft_kernel
    int big_fb_total = 0;
    DevBuf<float> big_win, big_fbw, big_coef, big_lifter;
    DevBuf<float2> big_tw;
    DevBuf<int> big_range, big_slot, big_seg;
    DevBuf<double> big_coef_d;
    int lift_off = 0, tab_floats = 0, ck_off = 0, cf_off = 0, cfd_off = 0, am_off = 0, NS = 0, CW = 4, ncoef_out = 0;
    bool md = false;        // DCT tail on the matrix cores (frontend_kernel<..., MD>): tables are laid out for its lane map
    bool half_window = false;  // the headline instantiation (DUAL): the window table is scaled by 1/2, which is the 1/4 of its power spectrum
    bool vf = false;        // Burg-cepstral VAD criterion fused into the front end (frontend_kernel<..., VF>)
    bool sy = false;        // speech-enhancement output with the inverse transform inside the front end (frontend_kernel<..., SY>)
    int ss = 0;             // hwss / fwss / 2fwss (1 / 2 / 3) on frontend_kernel<..., SS>
    std::vector<float> ss_stale;  // the spectrum vector the last file of the previous run left behind (zeros at first)
    // -vad file=<f>: ONE byte stream for all files of the process, a byte per frame, never rewound (nr.cc:205-209, 273, 297-302)
    bool ss_file = false;
    std::vector<unsigned char> vad_stream;
    int64_t vad_pos = 0;
    int han_off = 0;
    bool per_wave = false;  // chains per wave (state along an utterance lives in a wave's registers)
    size_t lds_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool host_timed = false;    // the last run was a host run cut into ranges: host_kernel_ms = sum over the ranges' front-end launches
    float host_kernel_ms = 0.f;
    bool in_signal_call = false;
    // CMVN (row N2): statistic slot <-> row column maps, and per-call scratch
    std::vector<int> col_of_slot, slot_of_col;
    DevBuf<int> d_col_of_slot, d_slot_of_col, d_spk;
    DevBuf<double> d_stat_a, d_stat_b;
    DevBuf<unsigned long long> stamps;
    std::set<const void *> attr_done;  // kernels whose dynamic-LDS limit has been raised on this engine's device
    bool do_vad = false;
    VadParams vp;
};

struct ctu_plan {
    ctu_engine *eng = nullptr;
    int n_utt = 0;
    std::vector<int64_t> nsamples, sample_off, row_off, frames;
    int64_t total_samples = 0, total_frames = 0;
    int n_tiles = 0;
    int grid = 0;               // workgroups of the front-end launch (tile chains are built for it)
    DevBuf<TileRec> tiles;
    DevBuf<int> wg_first;
    // host-buffer runs: device copies of the arena / rows / VAD bytes, kept for the life of the plan
    DevBuf<int16_t> h_pcm;
    DevBuf<float> h_rows;
    DevBuf<uint8_t> h_vad;
    // host-buffer runs of large batches: consecutive utterance ranges as plans of their own, run on two streams so that
    // the upload of one range overlaps the kernels and the download of the previous one (ctu_engine_run_host)
    std::vector<std::unique_ptr<ctu_plan>> parts;
    std::vector<int> part_first;       // first utterance of every part, n_utt at the end
    int parts_for = 0;                 // the range count `parts` was built for (ctu_engine_run_host)
    // VAD majority filter out of phase (ctu_plan_set_vad_ring): which frame's vector every written row carries (-1: an untouched ring
    // slot = zeros; absolute row numbers), and a copy of the rows to gather from
    std::vector<int> ring_hidx;        // per utterance, empty = all in phase
    DevBuf<int> ring_src;              // [total_frames]
    DevBuf<float> ring_tmp;            // [total_frames][D]
    hipStream_t part_stream[2] = {nullptr, nullptr};
    ~ctu_plan() {
        for (hipStream_t st : part_stream)
            if (st) (void)hipStreamDestroy(st);
    }
    DevBuf<int> tile_utt;            // utterance of every tile (SS)
    DevBuf<float> ss_seed, ss_last;  // SS: noise seeds per utterance [n_utt][K] and the vectors the utterances leave behind
    DevBuf<unsigned char> ss_dirty;  // SS: utterances a pass of the seed iteration recomputes
    DevBuf<unsigned char> ss_vbits;  // SS: the detector's decision of every frame (first pass), reused by the later passes
    DevBuf<float2> xri;         // VAD scratch
    DevBuf<float> pnr;
    DevBuf<double> vad_ci;
    DevBuf<float> vad_cf;      // fused Burg-cepstral VAD: cepstra of every frame [total_frames][VFC_STRIDE] ahead of vad_lanes_kernel
    DevBuf<int> vf_order;      // utterances with at least one frame, longest first (a wave of vad_lanes_kernel takes 16 in a row)
    int n_live = 0;
    DevBuf<int64_t> d_row_off;
    DevBuf<double> dc1m;       // -remove_dc1: frame means, then
    DevBuf<float> dc1;         // the offsets the frames subtract (decode_kernels.h)
    int max_frames = 0;        // longest utterance of the plan
    // scratch between the kernels of one run; owned by the plan, so plans can run concurrently on different streams
    DevBuf<float> logmel;      // TRAP: log-mel rows [total_frames][B]
    DevBuf<double> lp_r;       // LP kinds: autocorrelation lags [total_frames][lporder + 1] (used as float rows by the fp32 path) ahead of lp_tail_kernel
    DevBuf<float> base_rows;   // front-end rows ahead of the delta / stacking / CMS passes [total_frames][Dbase]
    DevBuf<float> ybuf;        // signal output: time-domain frames ahead of the overlap-add [total_frames][window]
    std::vector<int64_t> out_samples;   // signal output: samples written per utterance
    DevBuf<long long> d_sample_off;
    // TRAP
    DevBuf<int4> utt_info;
    DevBuf<int> trap_chunks, trap_chunks128;   // (utterance, first frame) of every 64- / 128-frame chunk
    int n_trap_chunks = 0, n_trap_chunks128 = 0;
};

namespace {

void set_error(ctu_engine *e, const std::string &m) { e->err = m; }

bool ss_eligible(const ctu::Design &d);
int ss_mode_of(const ctu::Opts &o);

// reasons a valid ctucopy configuration is outside the accelerated path
std::string unsupported_reason(const ctu::Design &d) {
    const ctu::Opts &o = d.o;
    if (d.signal_out) {  // row N3: IN -> NR -> sigOUT
        if (o.format_in == "htk") return "HTK feature input with signal output";
        if (o.fea_kind == "td-iir-mfcc") return "fea_kind outside the spectral path";
        if (o.dither != 0.) return "-dither != 0 makes outputs depend on file order (src/io/in.cc:205,454)";
        if (o.remove_dc1) {
            if (d.window / d.wshift > 8) return "-remove_dc1 with more than 8 frames over a sample (window / shift above 8)";
            if (o.fea_E && o.fea_rawenergy) return "-remove_dc1 together with -fea_rawenergy";
        }
        if (o.nr_mode != "none" && o.nr_mode != "exten" && !ss_eligible(d))
            return "hwss / fwss / 2fwss with signal output outside the fused detector path (windows of 129 .. 208 samples on 256 points or 257 .. 400 on 512, -vad burg with 2 to 16 cepstral coefficients or -vad file=..., DC removal on)";
        if (o.rasta) return "-nr_rasta";
        // BATCH only constructs its VAD on the feature paths (init_out, src/io/batch.cc:70-76); with signal output save_frame() calls
        // through the never-assigned pointer (batch.cc:230-241): the reference crashes, there is nothing to reproduce
        if (o.do_vad()) return "VAD together with signal output (the reference dereferences a VAD it never constructs there, src/io/batch.cc:62-66,230-241)";
        if (d.wfft > 4096) return "FFT size above 4096";
        if (d.wfft >= 1024) {  // bigfft_kernel exports the spectra, bigsynth_kernel transforms back: the plain chain and exten
            if (o.nr_mode != "none" && o.nr_mode != "exten") return "hwss / fwss / 2fwss with signal output at an FFT size above 512";
        }
        else if (d.wfft != 512 && d.wfft != 256) return "signal output at an FFT size below 256";
        if (d.window % 2 && d.wfft < 1024) return "odd window length with signal output at an FFT size below 1024";
        if (d.window < 32) return "window shorter than 32 samples";
        return "";
    }
    if (o.format_in == "htk") return "HTK feature input (-format_in htk) bypasses the spectral path";
    if (o.fea_kind == "td-iir-mfcc" || o.fea_kind == "none") return "fea_kind outside the spectral feature path";
    if (o.dither != 0.) return "-dither != 0 makes outputs depend on file order (src/io/in.cc:205,454)";
    if (o.remove_dc1) {
        if (d.window / d.wshift > 8) return "-remove_dc1 with more than 8 frames over a sample (window / shift above 8)";
        if (o.fea_E && o.fea_rawenergy) return "-remove_dc1 together with -fea_rawenergy";
        if (o.do_vad() && !(o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "fea")) return "-remove_dc1 together with a VAD criterion on the spectrum";
    }
    if (o.nr_mode != "none" && o.nr_mode != "exten") {
        // -vad_apply_mode silence zeroes in->_Xsabs behind a non-speech frame (src/vad/vad.cc:727-736).  The features of the frame are
        // out by then and the next get_frame() rewrites the vector, so on every other chain the mode changes nothing - but these
        // modes seed the next file's noise estimate from that very vector (src/nr/nr.cc:212-221)
        if (o.vad_apply_mode == "silence") return "-vad_apply_mode silence together with hwss / fwss / 2fwss (it zeroes the vector the next file's noise estimate starts from)";
        if (!ss_eligible(d)) return "hwss / fwss / 2fwss outside the fused detector path (windows of 129 .. 208 samples on 256 points or 257 .. 400 on 512, -vad burg with 2 to 16 cepstral coefficients or -vad file=..., DC removal on, at most 16 cepstral / LP coefficients, no trapdct, no -nr_when afterFB, no CMVN, no VAD module beside it)";
    }
    if (o.nr_when_afterFB) {
        if (d.signal_out) return "-nr_when afterFB together with signal output";
        if (d.B > 64) return "-nr_when afterFB with more than 64 bands";
    }
    if (o.rasta) return "-nr_rasta";
    if (d.post_order > 0) {
        if (d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "delta / stacking on non-cepstral kinds (the reference sizes the chain as fea_ncepcoefs+1, src/fea/fea_delta.cc:22-28)";
        if (!o.fea_c0) return "delta / stacking without -fea_c0 (the reference's writers leave slots of the row unwritten, src/io/out.cc:190-201)";
        int wsum = 0;
        for (int j = 0; j < d.post_order; j++) {
            if (d.post_w[j] > 16) return "delta / stacking window above 16 frames";
            wsum += d.post_w[j];
        }
        if (wsum > 24) return "delta windows adding up to more than 24 frames (LDS tile of the chain)";
    }
    if (o.stat_cmvn || o.apply_cmvn) {
        if (d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "CMVN on non-cepstral kinds";
        if (!o.fea_c0) return "CMVN without -fea_c0 (c0 is part of the statistics but not of the written row)";
        if (d.post_stack) return "CMVN on stacked vectors";
        if (d.cms) return "CMVN together with CMS (the reference warns and lets CMVN win, src/io/opts.cc:262-264)";
        // the statistics are taken over every frame and the VAD runs on the normalised vectors of the last pass (src/io/batch.cc:193-204,230-241):
        // a criterion on those vectors would need the statistics first
        if (o.do_vad() && o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "fea") return "the `fea` VAD criterion together with CMVN";
    }
    if (d.cms) {
        if (d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "CMS on non-cepstral kinds (the reference walks fea_ncepcoefs+1 entries whatever the vector holds, src/fea/post_impl.cc:203-240)";
        if (d.post_stack) return "CMS on stacked vectors";
        if (d.cms == 2 && (o.length_b < 1 || o.length_b > 512)) return "block CMS window outside 1..512 frames";
        if (d.cms_cols > 32) return "more than 32 CMS columns";
        if (d.cms == 2 && (size_t)(64 + o.length_b - 1) * d.cms_cols * sizeof(float) > 64 * 1024) return "block CMS tile above 64 KiB of LDS";
    }
    if (o.fea_E && d.kind == ctu::FeaKind::TrapDct) return "-fea_E with trapdct (the energy lags the features by 50 frames in the reference)";
    if (o.do_vad()) {
        // trapdct delays the writer by half a context (src/fea/fea_trap.cc:53-127): the detector runs when a vector comes out, on the
        // newest input frame's criterion - the delta chains' mechanism (vp.delay); the `fea` criterion would read 368-entry vectors
        if (d.kind == ctu::FeaKind::TrapDct && o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode != "lpc") return "the `fea` VAD criterion on TRAP vectors";
        if (o.vad_cri_mode != "energy" && o.vad_cri_mode != "cepdist") return "";  // rejected with the reference's text at create
        if (o.vad_cri_mode == "cepdist") {
            if (o.vad_cepdist_mode == "in") return "-vad_cepdist_mode in (HTK feature input)";
            if (o.vad_cepdist_mode == "fea" && d.kind != ctu::FeaKind::Dctc && d.kind != ctu::FeaKind::Lpc) return "-vad_cepdist_mode fea on non-cepstral features";
            const int nc = o.vad_cepdist_mode == "lpc" ? o.vad_lpc_coefs : d.nfea;
            if (nc > 32 || nc < 2) return "more than 32 (or fewer than 2) VAD cepstral coefficients";
        }
        if (o.vad_filter_order > 31) return "VAD filter order above 31";
        if (o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "lpc" && d.window <= 128) return "Burg-cepstral VAD with an FFT size below 256 (the detector's inverse transform has the reference's size)";
    }
    if (d.wfft >= 1024) {  // bigfft_kernel.h: the plain chain
        if (d.wfft > 4096) return "FFT size above 4096";
        // exten on the spectrum at 1024 points: wave1k_kernel carries the recurrence along per-wave chains of utterances
        // exten on the spectrum at 1024 .. 4096 points: wave1k_kernel / bigfft_kernel carry the recurrence along chains of whole utterances
        const bool big_exten = o.nr_mode == "exten" && !o.nr_when_afterFB && !d.signal_out;
        if ((o.nr_mode != "none" || o.nr_when_afterFB) && !big_exten) return "noise reduction with an FFT size above 512 (exten on the spectrum excepted)";
        // the VAD on 1024-point frames: the criteria that need no spectrum behind the front end - the energy of the vector the NR left
        // (wave1k_kernel stores it per frame) and the cepstral distance on the output vectors
        const bool big_vad = !d.signal_out && (o.vad_cri_mode == "energy" || (o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "fea"));
        if (o.do_vad() && !big_vad) return "VAD with an FFT size above 512 (the energy criterion and -vad_cepdist_mode fea excepted)";
        if (d.B > 64) return "more than 64 bands with an FFT size above 512";
    }
    else if (d.wfft != 512 && d.wfft != 256) return "FFT size below 32";
    if (d.window < 32) return "window shorter than 32 samples";
    if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) {
        if (o.fea_lporder >= d.B) return "LP order not below the number of bands: the normal equations are singular and the reference's output is rounding noise";
        if (o.fea_lporder > MAX_LP || o.fea_ncepcoefs > MAX_LP) return "LP order / cepstral order above 23 (the front end accumulates 24 lags per frame)";
    }
    if (d.kind == ctu::FeaKind::Dctc && d.nfea > MAXC) return "more cepstral coefficients than the kernel accumulates";
    if (d.B > 512) return "more than 512 filter bank channels";
    if (d.kind == ctu::FeaKind::TrapDct && o.fea_trapdct_ndct > 32) return "more than 32 TRAP DCT coefficients";
    if (d.kind == ctu::FeaKind::TrapDct && o.fea_trapdct_traplen > 255) return "TRAP longer than 255 frames";
    if (d.kind == ctu::FeaKind::TrapDct && (size_t)(64 + 256) * (d.B | 1) * 4 > 64 * 1024) return "too many bands for the TRAP tile";
    return "";
}

// Host-side image of the phase-2 LDS tables (see KParams for the layout).
struct Phase2Tables {
    std::vector<float> ft;   // LDS image followed by the lifter
    std::vector<int> it;     // slot_chunk[NS+1] | row_slot[nfea]
    std::vector<int> cells, slot_chunk;
    int lift_off = 0, tab_floats = 0, ck_off = 0, cf_off = 0, cfd_off = 0, am_off = 0, han_off = 0, NS = 0, CW = 4, ncoef_out = 0;
    bool md = false;  // lane map of the MFMA tail: lane = frame + 8 h + 16 kk, group = kk + 4 h
};

// The DCT tail runs on the matrix cores for the plain cepstral chains (what launch_vx instantiates with MD).
#ifndef CTU_MD
#define CTU_MD 1
#endif
#ifndef CTU_VF
#define CTU_VF 1
#endif
#ifndef CTU_SY
#define CTU_SY 1  // 0: speech-enhancement output through the exported spectra and synth_kernel (round 1's path)
#endif
// the plain cepstral chain: what the specialised instantiations (GEN_PLAIN / GEN_EXTEN with MD) cover
bool plain_cepstral(const ctu::Design &d) {
    const ctu::Opts &o = d.o;
    return !o.nr_when_afterFB && d.kind == ctu::FeaKind::Dctc && d.nfea <= 16 && !o.fea_E && o.fb_power && o.remove_dc && !o.remove_dc1 && !o.fb_inld && !d.signal_out;
}
// the frame shapes the fused detector paths are built for: 8 kHz / 25 ms (256-point mode, two frames per complex transform) and
// 16 kHz / 25 ms (512-point mode, 16 lanes x 25 samples) - vad_fused.h
bool fused_frame_shape(const ctu::Design &d) {
    return (d.wfft == 256 && d.window == VF_WINDOW) || (d.wfft == 512 && d.window == VF0_WINDOW);
}
// the *ss modes' detector: any window its lanes' 13 / 25 samples cover in the two transform sizes (the run-time-flag instantiations look
// the window's end up at run time); the shapes above keep their straight-line instantiations on the plain chain
bool ss_frame_shape(const ctu::Design &d) {
    return (d.wfft == 256 && d.window > 128 && d.window <= 16 * VF_SPL) || (d.wfft == 512 && d.window > 256 && d.window <= 16 * VF0_SPL);
}
// Burg-cepstral VAD criterion fused into the front end (vad_fused.h): 14 coefficients (the preset's detector)
bool vf_eligible(const ctu::Design &d) {
    const ctu::Opts &o = d.o;
    return CTU_VF && CTU_MD && plain_cepstral(d) && o.do_vad() && o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "lpc" &&
           fused_frame_shape(d) && o.vad_lpc_coefs == VF_NC && d.post_order == 0;
}
// the compressed-band LP chain (PLP and friends: -fb_inld, at most 16 lags, no energy column, no NR, no VAD): its cosine iDFT
// (src/fea/fea_impl.cc:181-198) is the same contraction over the bands as the DCT and takes the same MFMA tail
#ifndef CTU_LP_MD
#define CTU_LP_MD 1  // 0: the lags on the VALU (cell_accumulate + cells_reduce), for A/B
#endif
bool lp_md_eligible(const ctu::Design &d) {
    const ctu::Opts &o = d.o;
    return CTU_MD && CTU_LP_MD && (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) && o.fb_inld && o.fea_lporder + 1 <= 16 && !o.nr_when_afterFB && !o.fea_E &&
           o.fb_power && o.remove_dc && !o.remove_dc1 && !d.signal_out && !o.do_vad() && o.nr_mode == "none";
}
bool md_eligible(const ctu::Design &d) {
    // (the *ss modes on another window than the presets' run the run-time-flag instantiation, which has no MFMA tail)
    return (CTU_MD && plain_cepstral(d) && (!d.o.do_vad() || vf_eligible(d)) && !(ss_mode_of(d.o) && !fused_frame_shape(d))) || lp_md_eligible(d);
}
// hwss / fwss / 2fwss with the Burg cepstral detector (frontend_kernel<..., SS>): 25 ms frames at 8 or 16 kHz, the
// presets' 12 cepstral coefficients for the detector, the plain chain into cepstra or band energies
#ifndef CTU_SS_CACHE
#define CTU_SS_CACHE 1  // 0: every pass of the seed iteration runs the detector again (A/B)
#endif
int ss_mode_of(const ctu::Opts &o) { return o.nr_mode == "hwss" ? 1 : o.nr_mode == "fwss" ? 2 : o.nr_mode == "2fwss" ? 3 : 0; }
// the same modes ahead of sigOUT (-format_out raw|wave): the NR object works on in->_Xsabs - magnitudes, -fb_power is forced off there -
// and the enhanced frames go back to the time domain inside the front end (frontend_kernel<..., SS, SY>)
bool ss_signal_eligible(const ctu::Design &d) {
    const ctu::Opts &o = d.o;
    const bool det_ok = (o.vadmode == "burg" && o.fea_ncepcoefs >= 2 && o.fea_ncepcoefs <= SS_NC) || o.vadmode == "file";
    return CTU_SY && d.signal_out && ss_mode_of(o) && det_ok && ss_frame_shape(d) && o.remove_dc && !o.remove_dc1 && !o.rasta && !o.do_vad();
}
bool ss_eligible(const ctu::Design &d) {
    const ctu::Opts &o = d.o;
    if (d.signal_out) return ss_signal_eligible(d);
    // cepstra / LP coefficients in the sixteen-row accumulators, or band energies; everything behind the front end (the LP tail, delta /
    // stacking, CMS, CMVN) runs on the rows of the last pass of the seed iteration
    const bool lp = d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa;
    const bool kind_ok = (d.kind == ctu::FeaKind::Dctc && d.nfea <= 16) || (lp && o.fea_lporder + 1 <= 16) || d.kind == ctu::FeaKind::Spec || d.kind == ctu::FeaKind::LogSpec;
    // -vad file=<f> (nr.cc:205-209, 297-302): the decisions come from a byte stream instead of the detector; same kernel, same frame shapes
    const bool det_ok = (o.vadmode == "burg" && o.fea_ncepcoefs >= 2 && o.fea_ncepcoefs <= SS_NC) || o.vadmode == "file";
    // CMVN is two passes over the list in the reference (statistics, then the rows): the second starts from the noise vector the first
    // left behind, and no oracle restates that - refused rather than guessed
    return CTU_MD && ss_mode_of(o) && det_ok && !o.nr_when_afterFB && ss_frame_shape(d) && kind_ok && o.remove_dc && !o.remove_dc1 && !o.do_vad() && !d.signal_out && !o.rasta &&
           !o.stat_cmvn && !o.apply_cmvn;
}

void build_phase2(const ctu::Design &d, Phase2Tables &t) {
    t.md = md_eligible(d);
    // ---- phase-2 tables.  Bands are dealt to (slot, group) cells: sorted by width, eight per slot, so that the
    // eight lanes of a frame walk bands of similar width in lock step.  A band is cut into 4-bin chunks whose
    // bin range stays inside [0,K); weights outside the band's own [first,last] are zero.
    const int B = d.B;
    std::vector<int> order(B);
    for (int b = 0; b < B; b++) order[b] = b;
    auto width = [&](int b) { return d.fb_last[b] - d.fb_first[b] + 1; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return width(x) > width(y); });
    const int NS = (B + 7) / 8;
    int ncoef = 0;
    const std::vector<double> *coef_tab = nullptr;
    if (d.kind == ctu::FeaKind::Dctc) { coef_tab = &d.dct; ncoef = d.nfea; }
    else if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) { coef_tab = &d.idft; ncoef = d.o.fea_lporder + 1; }
    if (ncoef > MAXC) throw std::runtime_error("more cepstral / LP coefficients than the kernel accumulates");
    const int CW = ncoef <= 16 ? 16 : MAXC;  // the kernel has straight-line code for these two widths
    std::vector<int> slot_chunk(NS + 1, 0);
    std::vector<float> cw;                  // chunk weights [NC][8][4]
    std::vector<int> cell(NS * 8 * 2, 0);   // {first bin of the chunk run, band index or -1}
    const int CWS = CW + 4;  // row stride: 20 or 28 floats = 5 or 7 16-byte units, distinct mod 16 over the 8 groups
    std::vector<float> cf((size_t)NS * 8 * CWS, 0.f);
    // LP analysis on uncompressed band energies (no -fb_inld): the same rows in double for the double tail (FEAT_LPD)
    const bool lpd = (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) && !d.o.fb_inld;
    std::vector<double> cfd(lpd ? (size_t)NS * 8 * CW : 0, 0.0);
    // DCTC: coefficient row r of a cell is the value written to output slot r (c1..cN, then c0)
    std::vector<int> coef_of_slot;
    if (d.kind == ctu::FeaKind::Dctc) {
        coef_of_slot.assign(d.nfea, -1);
        int nout = 0;
        for (int i = 0; i < d.nfea; i++)
            if (d.row_slot[i] >= 0) {
                coef_of_slot[d.row_slot[i]] = i;
                nout = std::max(nout, d.row_slot[i] + 1);
            }
        t.ncoef_out = nout;
    }
    auto chunks_of = [&](int b) {  // chunks of 4 bins from the aligned start of the band to its last bin
        const int k0 = d.fb_first[b] & ~3;
        return (d.fb_last[b] - k0) / 4 + 1;
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return chunks_of(x) > chunks_of(y); });
    for (int sl = 0; sl < NS; sl++) {
        int nch = 0;
        for (int g = 0; g < 8 && sl * 8 + g < B; g++) nch = std::max(nch, chunks_of(order[sl * 8 + g]));
        if (4 * nch > PSTRIDE) throw std::runtime_error("filter band wider than the spectrum");
        slot_chunk[sl] = (int)cw.size() / 32;
        // Phase 2 reads P with one ds_read_b128 per lane; the 16-byte unit a lane touches is
        // (frame + kstart/4 + chunk) mod 16 (rows are 65 units apart).  A b128 wave access is served in four groups of
        // 16 lanes = {frame f: groups 0-3, f+1: groups 4-7, f+2: groups 4-7, f+3: groups 0-3}; collisions depend only
        // on a_g = kstart_g/4.  Search the assignment of this slot's bands to groups (and up to `slack` leading
        // zero chunks) for the fewest colliding lane pairs.
        std::vector<int> perm(8), best_perm(8), lead(8, 0), best_lead(8, 0);
        for (int g = 0; g < 8; g++) perm[g] = best_perm[g] = sl * 8 + g < B ? order[sl * 8 + g] : -1;
        auto a_of = [&](int b, int ld) { return b < 0 ? -1000 : (std::min(d.fb_first[b] & ~3, PSTRIDE - 4 * nch) / 4 - ld); };
        auto cost = [&](const std::vector<int> &pm, const std::vector<int> &ld) {
            int c = 0, a[8];
            for (int g = 0; g < 8; g++) a[g] = a_of(pm[g], ld[g]);
            // a ds_read_b128 is served in four groups of 16 lanes: {0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32
            static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                           {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
            for (int q = 0; q < 4; q++) {
                int unit[16];
                for (int i = 0; i < 16; i++) {
                    const int lane = grp[q & 1][i] + 32 * (q >> 1);
                    const int f8 = t.md ? (lane & 7) : (lane >> 3), g = t.md ? (((lane >> 3) & 1) * 4 + (lane >> 4)) : (lane & 7);
                    unit[i] = a[g] < -500 ? -1 - i : ((a[g] + f8) % 16 + 16) % 16;  // rows are 65 units apart
                }
                for (int i = 0; i < 16; i++)
                    for (int j = i + 1; j < 16; j++) c += (unit[i] >= 0 && unit[i] == unit[j]);
            }
            return c;
        };
        {
            int best = cost(perm, lead);
            uint32_t rng = 12345u + sl;
            for (int it = 0; it < 4000 && best > 0; it++) {
                std::vector<int> pm = best_perm, ld = best_lead;
                rng = rng * 1664525u + 1013904223u;
                const int i = (rng >> 8) % 8, j = (rng >> 16) % 8;
                std::swap(pm[i], pm[j]);
                std::swap(ld[i], ld[j]);
                rng = rng * 1664525u + 1013904223u;
                const int g = (rng >> 8) % 8;
                if (pm[g] >= 0) {
                    // leading zero chunks are allowed while the run still starts at >= 0 and ends beyond the last bin
                    const int k0 = std::min(d.fb_first[pm[g]] & ~3, PSTRIDE - 4 * nch);
                    const int mx = std::min({3, k0 / 4, (k0 + 4 * nch - (d.fb_last[pm[g]] + 1)) / 4});
                    ld[g] = mx > 0 ? (int)((rng >> 20) % (mx + 1)) : 0;
                }
                const int c = cost(pm, ld);
                if (c <= best) {
                    best = c;
                    best_perm = pm;
                    best_lead = ld;
                }
            }
        }
        std::vector<int> kstart(8, 0);
        for (int g = 0; g < 8; g++) {
            cell[(sl * 8 + g) * 2 + 1] = -1;
            if (best_perm[g] < 0) continue;
            const int b = best_perm[g];
            // aligned start; the run of nch chunks must end inside the row's PSTRIDE floats (bins >= K get weight 0
            // but are read, so the kernel keeps them finite: it zeroes the row padding once per workgroup)
            kstart[g] = std::min(d.fb_first[b] & ~3, PSTRIDE - 4 * nch) - 4 * best_lead[g];
            cell[(sl * 8 + g) * 2] = kstart[g];
            cell[(sl * 8 + g) * 2 + 1] = b;
            for (int i = 0; i < ncoef; i++) {
                const int src = (d.kind == ctu::FeaKind::Dctc) ? coef_of_slot[i] : i;
                if (src >= 0) cf[((size_t)sl * 8 + g) * CWS + i] = (float)(*coef_tab)[(size_t)src * B + b];
                if (src >= 0 && lpd) cfd[((size_t)sl * 8 + g) * CW + i] = (*coef_tab)[(size_t)src * B + b];
            }
        }
        for (int ch = 0; ch < nch; ch++)
            for (int g = 0; g < 8; g++)
                for (int i = 0; i < 4; i++) {
                    float w = 0.f;
                    if (best_perm[g] >= 0) {
                        const int b = best_perm[g], k = kstart[g] + 4 * ch + i;
                        if (k >= d.fb_first[b] && k <= d.fb_last[b]) w = (float)d.fb[b][k];
                    }
                    cw.push_back(w);
                }
    }
    slot_chunk[NS] = (int)cw.size() / 32;
    std::vector<float> ft(cw);
    t.cells = cell;
    t.slot_chunk = slot_chunk;
    auto push_ints = [&](const std::vector<int> &v) {
        for (int x : v) {
            float f;
            std::memcpy(&f, &x, 4);
            ft.push_back(f);
        }
        while (ft.size() & 3) ft.push_back(0.f);
    };
    t.ck_off = (int)ft.size();
    {   // the kernel's records: {first bin, band or -1, first chunk of the slot, chunks of the slot}; one more slot of idle cells, which
        // the walk prefetches behind the last one
        std::vector<int> rec4((size_t)(NS + 1) * 8 * 4, 0);
        for (int sl = 0; sl <= NS; sl++)
            for (int g = 0; g < 8; g++) {
                int *r = &rec4[(size_t)(sl * 8 + g) * 4];
                r[0] = sl < NS ? cell[(sl * 8 + g) * 2] : 0;
                r[1] = sl < NS ? cell[(sl * 8 + g) * 2 + 1] : -1;
                r[2] = slot_chunk[std::min(sl, NS)];
                r[3] = sl < NS ? slot_chunk[sl + 1] - slot_chunk[sl] : 0;
            }
        push_ints(rec4);
    }
    t.cf_off = (int)ft.size();
    if (!t.md) ft.insert(ft.end(), cf.begin(), cf.end());  // the MFMA tail reads the am table below instead
    while (ft.size() & 3) ft.push_back(0.f);
    if (t.md) {
        // A operands of the DCT MFMAs: for slot sl and half h, lane m + 16 kk holds row m of the folded DCT table (LP chain: of
        // the cosine iDFT table) at the band of cell (sl, group kk + 4 h); zero for idle cells and for rows >= the number of output slots
        t.am_off = (int)ft.size();
        for (int sl = 0; sl < NS; sl++)
            for (int h = 0; h < 2; h++)
                for (int lane = 0; lane < 64; lane++) {
                    const int m = lane & 15, g = (lane >> 4) + 4 * h;
                    const int b = cell[(sl * 8 + g) * 2 + 1];
                    float v = 0.f;
                    if (d.kind == ctu::FeaKind::Dctc) {
                        if (b >= 0 && m < t.ncoef_out && coef_of_slot[m] >= 0) v = (float)(*coef_tab)[(size_t)coef_of_slot[m] * B + b];
                    } else if (b >= 0 && m < ncoef) v = (float)(*coef_tab)[(size_t)m * B + b];  // LP: row m = lag m
                    ft.push_back(v);
                }
    }
    if (lpd) {
        t.cfd_off = (int)ft.size();  // a multiple of 4 floats: 8-byte aligned in LDS
        for (double v : cfd) {
            float h[2];
            std::memcpy(h, &v, 8);
            ft.push_back(h[0]);
            ft.push_back(h[1]);
        }
        while (ft.size() & 3) ft.push_back(0.f);
    }
    if (ss_eligible(d)) {
        // Hann window of the *ss modes' detector, han[i] = 0.5 (1 - cos(2 * 3.141592653 / window * i)) (src/vdet/CepstralDet.h:133-136)
        t.han_off = (int)ft.size();
        const double m = 2 * 3.141592653 / d.window;
        for (int i = 0; i < 16 * (d.wfft == 512 ? VF0_SPL : VF_SPL); i++) ft.push_back(i < d.window ? (float)(0.5 * (1 - std::cos(m * i))) : 0.f);
    }
    t.tab_floats = (int)ft.size();
    t.NS = NS;
    t.CW = CW;
    t.lift_off = (int)ft.size();
    for (double v : d.lifter) ft.push_back((float)v);
    ft.push_back(0.f);
    t.ft = ft;
    std::vector<int> it(slot_chunk);
    it.insert(it.end(), d.row_slot.begin(), d.row_slot.end());
    t.it = it;
}

// Rebuilds every band's dense weight row from the chunk tables and returns the largest deviation from the
// float-rounded filter bank (0 when the tables are consistent).
double check_phase2(const ctu::Design &d, const Phase2Tables &t) {
    double worst = 0;
    std::vector<int> seen(d.B, 0);
    for (int sl = 0; sl < t.NS; sl++)
        for (int g = 0; g < 8; g++) {
            const int kstart = t.cells[(sl * 8 + g) * 2], b = t.cells[(sl * 8 + g) * 2 + 1];
            if (b < 0) continue;
            seen[b]++;
            std::vector<double> row(PSTRIDE, 0.0);
            for (int ch = t.slot_chunk[sl]; ch < t.slot_chunk[sl + 1]; ch++)
                for (int i = 0; i < 4; i++) {
                    const int k = kstart + 4 * (ch - t.slot_chunk[sl]) + i;
                    if (k < 0 || k >= PSTRIDE) return 1e30;
                    row[k] += t.ft[(size_t)(ch * 8 + g) * 4 + i];
                }
            for (int k = 0; k < PSTRIDE; k++) {
                const double want = (k < d.K && k >= d.fb_first[b] && k <= d.fb_last[b]) ? (double)(float)d.fb[b][k] : 0.0;
                if (std::fabs(row[k] - want) > 0 && getenv("CTU_P2_DEBUG")) fprintf(stderr, "sl %d g %d band %d k %d kstart %d first %d last %d row %g want %g nch %d\n", sl, g, b, k, kstart, d.fb_first[b], d.fb_last[b], row[k], want, t.slot_chunk[sl+1]-t.slot_chunk[sl]);
                worst = std::max(worst, std::fabs(row[k] - want));
            }
        }
    for (int b = 0; b < d.B; b++)
        if (seen[b] != 1) return 1e30;
    return worst;
}

#ifndef CTU_TRAP_BF16
#define CTU_TRAP_BF16 1  // 0: TRAP-DCT on the fp32 matrix pipe only (trapdct_mfma_kernel)
#endif
#ifndef CTU_TRAP_F16
#define CTU_TRAP_F16 1   // 1: two fp16 terms, three products; 0: three bf16 terms, six products (trap_kernel.h)
#endif
// A operands of trapdct_bf16_kernel: G shifted by the frame phase c, zero-padded to 128 taps, each value split into three
// bf16 terms (round to nearest even at every step, as the device splits the data)
void build_trap_bf16(ctu_engine *e) {
    const ctu::Design &d = *e->design;
    const int tl = d.o.fea_trapdct_traplen, nd = d.o.fea_trapdct_ndct, half = (tl - 1) / 2;
    e->trap_bf16 = CTU_TRAP_BF16 && nd <= 16 && half <= TB_OFF && 7 + (TB_OFF - half) + tl <= 128 && d.B <= 24;  // 24 bands: 62 KB of LDS (tile 38 KB + two fragment buffers 24 KB): two workgroups per CU
    if (!e->trap_bf16) return;
    auto rn = [](float v) {
        uint32_t u;
        std::memcpy(&u, &v, 4);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    auto up = [](uint16_t h) {
        uint32_t u = (uint32_t)h << 16;
        float f;
        std::memcpy(&f, &u, 4);
        return f;
    };
    const bool f16 = CTU_TRAP_F16;
    const int NT = f16 ? 2 : 3;
    std::vector<uint4> tab((size_t)8 * 4 * NT * 64);
    for (int c = 0; c < 8; c++)
        for (int s_ = 0; s_ < 4; s_++)
            for (int lane = 0; lane < 64; lane++) {
                const int m = lane & 15, q = lane >> 4;
                uint16_t term[3][8];
                for (int i = 0; i < 8; i++) {
                    const int j = 32 * s_ + 8 * q + i - c - (TB_OFF - half);
                    const float v = (m < nd && j >= 0 && j < tl) ? (float)d.trap[(size_t)m * tl + j] : 0.f;
                    if (f16) {
                        const _Float16 h = (_Float16)v, l = (_Float16)(v - (float)h);
                        std::memcpy(&term[0][i], &h, 2);
                        std::memcpy(&term[1][i], &l, 2);
                        term[2][i] = 0;
                    } else {
                        term[0][i] = rn(v);
                        const float r1 = v - up(term[0][i]);
                        term[1][i] = rn(r1);
                        term[2][i] = rn(r1 - up(term[1][i]));
                    }
                }
                for (int sp = 0; sp < NT; sp++) {
                    uint4 w;
                    w.x = term[sp][0] | ((uint32_t)term[sp][1] << 16);
                    w.y = term[sp][2] | ((uint32_t)term[sp][3] << 16);
                    w.z = term[sp][4] | ((uint32_t)term[sp][5] << 16);
                    w.w = term[sp][6] | ((uint32_t)term[sp][7] << 16);
                    tab[(((size_t)c * 4 + s_) * NT + sp) * 64 + lane] = w;
                }
            }
    e->trapG16.upload(tab);
}

void build_big_tables(ctu_engine *e) {
    const ctu::Design &d = *e->design;
    const double pi = 3.14159265358979323846;
    std::vector<float> win(d.hamming.begin(), d.hamming.end());
    e->big_win.upload(win);
    std::vector<float2> tw((size_t)d.wfft / 2);
    for (int m = 0; m < d.wfft / 2; m++) {
        const double a = 2.0 * pi * (double)m / (double)d.wfft;
        tw[m] = make_float2((float)std::cos(a), (float)-std::sin(a));
    }
    e->big_tw.upload(tw);
    std::vector<float> fbw;
    std::vector<int> range((size_t)d.B * 3);
    for (int b = 0; b < d.B; b++) {
        range[3 * b] = d.fb_first[b];
        range[3 * b + 1] = d.fb_last[b];
        range[3 * b + 2] = (int)fbw.size();
        for (int k = d.fb_first[b]; k <= d.fb_last[b]; k++) fbw.push_back((float)d.fb[b][k]);
    }
    e->big_fb_total = (int)fbw.size();
    if (fbw.empty()) fbw.push_back(0.f);
    e->big_fbw.upload(fbw);
    e->big_range.upload(range);
    {
        // wave1k_kernel's bank: the bands' bin runs cut into at most 64 segments of at most S bins (the smallest S that fits)
        std::vector<int> seg(256 + 2 * (size_t)std::max(d.B, 1), 0);
        if (d.B <= 64) {
            int S = 1;
            for (;; S++) {
                int n = 0;
                for (int b = 0; b < d.B; b++) n += std::max(1, (d.fb_last[b] - d.fb_first[b] + 1 + S - 1) / S);
                if (n <= 64) break;
            }
            int lane = 0;
            for (int b = 0; b < d.B; b++) {
                const int w = d.fb_last[b] - d.fb_first[b] + 1, ns = std::max(1, (w + S - 1) / S);
                seg[256 + 2 * b] = lane;
                seg[256 + 2 * b + 1] = ns;
                for (int i = 0; i < ns; i++, lane++) {
                    const int k0 = d.fb_first[b] + i * S, cnt = std::max(0, std::min(S, w - i * S));
                    seg[4 * lane] = b;
                    seg[4 * lane + 1] = k0;
                    seg[4 * lane + 2] = cnt;
                    seg[4 * lane + 3] = range[3 * b + 2] + i * S;
                }
            }
        }
        e->big_seg.upload(seg);
    }
    e->ncoef_out = 0;
    std::vector<float> coef(4, 0.f);
    std::vector<double> coef_d(4, 0.0);
    std::vector<int> slot(4, -1);
    if (d.kind == ctu::FeaKind::Dctc) {
        // row r of the table is the value written to output slot r (c1..cN, then c0): norm and lifter are in d.dct
        std::vector<int> coef_of_slot(d.nfea, -1);
        int nout = 0;
        for (int i = 0; i < d.nfea; i++)
            if (d.row_slot[i] >= 0) {
                coef_of_slot[d.row_slot[i]] = i;
                nout = std::max(nout, d.row_slot[i] + 1);
            }
        e->ncoef_out = nout;
        coef.assign((size_t)std::max(nout, 1) * d.B, 0.f);
        slot.assign(std::max(nout, 1), -1);
        for (int r = 0; r < nout; r++) {
            slot[r] = coef_of_slot[r];
            if (coef_of_slot[r] >= 0)
                for (int b = 0; b < d.B; b++) coef[(size_t)r * d.B + b] = (float)d.dct[(size_t)coef_of_slot[r] * d.B + b];
        }
    } else if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) {
        coef_d.assign(d.idft.begin(), d.idft.end());
        slot.assign(d.row_slot.begin(), d.row_slot.end());
        while (slot.size() < (size_t)MAX_LP + 2) slot.push_back(-1);
    }
    e->big_coef.upload(coef);
    e->big_coef_d.upload(coef_d);
    e->big_slot.upload(slot);
    std::vector<float> lif(d.lifter.begin(), d.lifter.end());
    lif.push_back(0.f);
    e->big_lifter.upload(lif);
    e->lds_bytes = 0;
    e->mode = 0;
    e->nz = 16;
    // the tables of the 512 / 256-point kernels stay empty
    e->ftab.upload(std::vector<float>(4, 0.f));
    e->itab.upload(std::vector<int>(4, 0));
    e->lanec.upload(std::vector<float>(16 * LANEC, 0.f));
    if (d.kind == ctu::FeaKind::TrapDct) {
        std::vector<float> g(d.trap.begin(), d.trap.end());
        e->trapG.upload(g);
        build_trap_bf16(e);
    }
    switch (d.kind) {
        case ctu::FeaKind::Dctc: e->feat = FEAT_DCTC; break;
        case ctu::FeaKind::Lpc:
        case ctu::FeaKind::Lpa: e->feat = FEAT_LP; break;
        default: e->feat = FEAT_BANDS; break;
    }
}

void build_tables(ctu_engine *e) {
    const ctu::Design &d = *e->design;
    const double pi = 3.14159265358979323846;
    e->big = d.wfft >= 1024;
    if (e->big) {
        // (-remove_dc1 at 1024 points takes bigfft_kernel<4>, which reads the frames' offsets)
        e->wave1k = d.wfft == 1024 && !d.o.remove_dc1 && !d.signal_out && !(getenv("CTU_WAVE1K") && atoi(getenv("CTU_WAVE1K")) == 0);  // (and speech output: the spectra's export is bigfft_kernel's)
        build_big_tables(e);
        return;
    }
    // ---- per-lane constant records (see LC_* above)
    const bool mode1 = d.wfft == 256;
    e->mode = mode1 ? 1 : 0;
    std::vector<float> lc(16 * LANEC, 0.f);
    for (int l = 0; l < 16; l++) {
        float *r = lc.data() + l * LANEC;
        for (int j = 0; j < 16; j++)
            for (int h = 0; h < 2; h++) {
                // MODE 0: lane l holds samples 32j+2l, +1 of row j; MODE 1: sample 16j+l (second slot unused)
                const int i = mode1 ? (h ? d.window : 16 * j + l) : 32 * j + 2 * l + h;
                r[LC_WIN + 2 * j + h] = i < d.window ? (float)d.hamming[i] : 0.f;
                r[LC_MASK + 2 * j + h] = i < d.window ? 1.f : 0.f;
            }
        for (int k1 = 1; k1 < 16; k1++) {
            const double a = -2 * pi * (double)(k1 * l) / 256.0;
            r[LC_TW + 2 * (k1 - 1)] = (float)std::cos(a);
            r[LC_TW + 2 * (k1 - 1) + 1] = (float)std::sin(a);
        }
        for (int k2 = 0; k2 < 8; k2++) {
            const double a = -2 * pi * (double)(l + 16 * k2) / 512.0;
            r[LC_UT + 2 * k2] = (float)std::cos(a);
            r[LC_UT + 2 * k2 + 1] = (float)std::sin(a);
        }
    }
    e->lanec.upload(lc);
    if (d.signal_out) {  // nothing is projected: empty table area
        e->ncoef_out = 0; e->ck_off = 0; e->cf_off = 0; e->tab_floats = 0; e->NS = 0; e->CW = 16; e->lift_off = 0;
        e->ftab.upload(std::vector<float>(4, 0.f));
        e->itab.upload(std::vector<int>(4, 0));
        e->lds_bytes = ((size_t)TILE * PSTRIDE + LTW_FLOATS) * sizeof(float);
        e->feat = FEAT_BANDS;
        e->nz = e->mode ? (d.window + 15) / 16 : (d.window + 31) / 32;
        e->sy = CTU_SY;
        e->ss = ss_eligible(d) ? ss_mode_of(d.o) : 0;
        e->ss_file = e->ss && d.o.vadmode == "file";
        if (e->ss) {  // the detector's Hann window is the only table (as build_phase2 lays it out for the feature path)
            std::vector<float> ft;
            const double m = 2 * 3.141592653 / d.window;
            for (int i = 0; i < 16 * (d.wfft == 512 ? VF0_SPL : VF_SPL); i++) ft.push_back(i < d.window ? (float)(0.5 * (1 - std::cos(m * i))) : 0.f);
            e->han_off = 0;
            e->tab_floats = (int)ft.size();
            ft.push_back(0.f);  // (where the lifter sits on the feature path)
            e->lift_off = e->tab_floats;
            e->ftab.upload(ft);
            e->lds_bytes = ((size_t)TILE * PSTRIDE + e->tab_floats + LTW_FLOATS + (!e->mode ? NWAVE * VF0_STAGE : 0)) * sizeof(float);
        }
        return;
    }
    Phase2Tables t;
    build_phase2(d, t);
    if (check_phase2(d, t) != 0.0) throw std::runtime_error("internal: phase-2 chunk tables do not reproduce the filter bank");
    e->ncoef_out = t.ncoef_out;
    e->ck_off = t.ck_off;
    e->cf_off = t.cf_off;
    e->cfd_off = t.cfd_off;
    e->am_off = t.am_off;
    e->md = t.md;
    e->vf = vf_eligible(d);
    e->ss = ss_eligible(d) ? ss_mode_of(d.o) : 0;
    e->ss_file = e->ss && d.o.vadmode == "file";
    e->han_off = t.han_off;
    e->tab_floats = t.tab_floats;
    e->NS = t.NS;
    e->CW = t.CW;
    e->lift_off = t.lift_off;
    e->ftab.upload(t.ft);
    e->itab.upload(t.it);
    e->lds_bytes = ((size_t)TILE * PSTRIDE + e->tab_floats + LTW_FLOATS + (d.o.nr_when_afterFB ? NWAVE * 128 : 0) +
                    ((e->vf || e->ss) && !e->mode ? NWAVE * VF0_STAGE : 0)) * sizeof(float);  // 512-point detector paths: staged frames per wave
    if (e->lds_bytes > 160 * 1024) throw std::runtime_error("configuration needs more than 160 KiB of LDS");
    if (d.kind == ctu::FeaKind::TrapDct) {
        std::vector<float> g(d.trap.begin(), d.trap.end());
        e->trapG.upload(g);
        build_trap_bf16(e);
    }
    switch (d.kind) {
        case ctu::FeaKind::Spec:
        case ctu::FeaKind::LogSpec:
        case ctu::FeaKind::TrapDct: e->feat = FEAT_BANDS; break;
        case ctu::FeaKind::Dctc: e->feat = FEAT_DCTC; break;
        case ctu::FeaKind::Lpc:
        case ctu::FeaKind::Lpa: e->feat = d.o.fb_inld ? FEAT_LP : FEAT_LPD; break;
        case ctu::FeaKind::None: e->feat = FEAT_BANDS; break;  // not reached: the signal path returns above
    }
    e->nz = e->mode ? (d.window + 15) / 16 : (d.window + 31) / 32;  // rows of samples per lane that can be non-zero
    // The DUAL instantiations (frontend_kernel.h: the plain chain, with or without the intensity-loudness law) take their window scaled by 1/2: the packed transform's untangle owes the
    // power spectrum a factor 1/4, and a power of two on the window goes through every rounding of the chain unchanged - the rows are
    // bit for bit those of 0.25f * (re^2 + im^2), two multiplications per bin pair cheaper.  launch_vx checks that the instantiation
    // it launches is the one the table was scaled for.
    {   // the same decision tree as launch_vx, from the options
        const ctu::Opts &o = d.o;
        const bool vx = o.do_vad() && !e->vf && (o.vad_cri_mode == "energy" || (o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "lpc"));
        const bool base = !vx && !o.fea_E && o.fb_power && o.remove_dc && !o.remove_dc1 && !o.nr_when_afterFB;
        const bool exten = o.nr_mode == "exten", narrow = e->CW == 16;
        bool dual = false;
        if (e->md && e->feat == FEAT_LP) dual = true;
        else if (e->md) dual = !exten;
        else if (vx) dual = false;
        else if (base && !o.fb_inld && !exten && e->feat != FEAT_LPD) dual = true;
        else if (base && o.fb_inld && !exten && narrow && e->feat == FEAT_LP) dual = true;
        e->half_window = CTU_DUAL && dual && !e->vf && !e->ss && !e->sy && !o.remove_dc1 && e->mode == 0 && e->nz == 13;
    }
    if (e->half_window) {
        for (int l = 0; l < 16; l++)
            for (int j = 0; j < 32; j++) lc[(size_t)l * LANEC + LC_WIN + j] *= 0.5f;
        e->lanec.upload(lc);
    }
}

// Kernels whose dynamic LDS may pass the 64 KiB default: the 160 KiB attribute is set once per engine (= per device) and instantiation.
template <class K>
void allow_big_lds(ctu_engine *e, K kern) {
    const void *fp = reinterpret_cast<const void *>(kern);
    if (!e->attr_done.count(fp)) {
        HIP_TRY(hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        e->attr_done.insert(fp);
    }
}
// One front-end launch.
template <class K>
void launch_fe(ctu_engine *e, K kern, dim3 grid, hipStream_t s, const KParams &kp) {
    allow_big_lds(e, kern);
    hipLaunchKernelGGL(kern, grid, dim3(WG), e->lds_bytes, s, kp);
}

template <int NZ, int MODE, bool VX, int GEN>
void launch_nz(ctu_engine *e, dim3 grid, hipStream_t s, const KParams &kp) {
    const bool wide = kp.CW != 16;  // coefficient rows of MAXC entries (more than 16 cepstra / LP lags)
    const int feat = e->feat;
    if (feat == FEAT_BANDS) launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, VX, 16, GEN>, grid, s, kp);
    else if (feat == FEAT_DCTC && !wide) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, VX, 16, GEN>, grid, s, kp);
    else if (feat == FEAT_DCTC) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, VX, MAXC, GEN>, grid, s, kp);
    else if (feat == FEAT_LPD) {
        if constexpr (GEN == GEN_FULL || GEN == GEN_DC1) {
            if (!wide) launch_fe(e, &frontend_kernel<NZ, FEAT_LPD, MODE, VX, 16, GEN>, grid, s, kp);
            else launch_fe(e, &frontend_kernel<NZ, FEAT_LPD, MODE, VX, MAXC, GEN>, grid, s, kp);
        } else throw std::runtime_error("internal: the double LP tail has run-time flags only");
    }
    else if (!wide) launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, VX, 16, GEN>, grid, s, kp);
    else launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, VX, MAXC, GEN>, grid, s, kp);
}

template <int NZ, int MODE>
void launch_vx(ctu_engine *e, dim3 grid, hipStream_t s, const KParams &kp) {
    // Specialised instantiations (see GEN in frontend_kernel.h): the plain chain, plain + intensity-loudness law for the
    // 16-coefficient LP path (PLP), plain + exten for 16-coefficient DCT / band outputs; the cepstral ones of these run
    // their DCT tail on the matrix cores (MD, tables laid out for it by build_phase2).  Everything else, and every run
    // with the VAD export, reads its flags at run time.
    const bool vx = kp.vad_export != 0;
    const int feat = e->feat;
    const bool base = !vx && kp.e_mode == 0 && kp.fb_power && kp.remove_dc && !kp.remove_dc1 && !kp.dbg && !kp.skip_phase2 && !kp.nr_after_fb;
    const bool narrow = kp.CW == 16;
    {   // the window table of this engine is scaled for exactly one instantiation (ctu_engine::half_window)
        constexpr bool dual_shape = CTU_DUAL && MODE == 0 && NZ < 16;
        bool dual = false;  // does the tree below end on a DUAL instantiation (GEN_PLAIN or GEN_INLD without export / detector / synthesis)?
        if (dual_shape && !(e->sy && kp.skip_phase2) && !kp.remove_dc1 && !e->ss && !e->vf) {
            if (e->md && feat == FEAT_LP) dual = true;
            else if (e->md) dual = !kp.nr_exten;
            else if (vx) dual = false;
            else if (base && !kp.fb_inld && !kp.nr_exten && feat != FEAT_LPD) dual = true;
            else if (base && kp.fb_inld && !kp.nr_exten && narrow && feat == FEAT_LP) dual = true;
        }
        if (dual != e->half_window) throw std::runtime_error("internal: window table scaled for another instantiation");
    }
    if (e->sy && kp.skip_phase2) {
        if (kp.remove_dc1) {
            if constexpr (NZ == 16) launch_fe(e, &frontend_kernel<16, FEAT_BANDS, MODE, false, 16, GEN_DC1, 0, false, false, false, true>, grid, s, kp);
        }
        else if (e->ss) {
            if constexpr (MODE == 1 || NZ == 13) launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, false, 16, GEN_FULL, 0, false, false, true, true>, grid, s, kp);
            else throw std::runtime_error("internal: SS engine without an SS instantiation");
        }
        else launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, false, 16, GEN_FULL, 0, false, false, false, true>, grid, s, kp);
    }
    else if (kp.remove_dc1) {  // the generic row count only (ctu_engine_run picks NZ = 16), no spectrum export
        if constexpr (NZ == 16) launch_nz<16, MODE, false, GEN_DC1>(e, grid, s, kp);
    }
    else if (e->ss) {
        if constexpr (MODE == 1 || NZ == 13) {
            // the plain chain: the detector's lattice unrolled for the presets' 12 coefficients (LPO = 12) or for up to 16
            const bool preset_win = kp.window == (MODE == 1 ? VF_WINDOW : VF0_WINDOW);  // the straight-line lattices name the window's last sample
            if (!preset_win) {
                if constexpr (NZ == 13) {
                    if (!narrow) throw std::runtime_error("internal: SS engine without an SS instantiation");
                    if (feat == FEAT_BANDS) launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                    else if (feat == FEAT_DCTC) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                    else if (feat == FEAT_LP) launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                    else launch_fe(e, &frontend_kernel<NZ, FEAT_LPD, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                } else throw std::runtime_error("internal: SS engine without an SS instantiation");
            }
            else if (e->md && feat == FEAT_DCTC && kp.ss_nc == 12) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_PLAIN, 12, true, false, true>, grid, s, kp);
            else if (e->md && feat == FEAT_DCTC) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_PLAIN, 0, true, false, true>, grid, s, kp);
            else if (feat == FEAT_BANDS && base && !kp.fb_inld && kp.ss_nc == 12) launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, false, 16, GEN_PLAIN, 12, false, false, true>, grid, s, kp);
            else if (feat == FEAT_BANDS && base && !kp.fb_inld) launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, false, 16, GEN_PLAIN, 0, false, false, true>, grid, s, kp);
            else if (!narrow) throw std::runtime_error("internal: SS engine without an SS instantiation");
            // energy columns, -fb_inld, -fb_power off, the LP kinds: the flags at run time (the 25 ms frame shapes only: NZ = 13)
            else if constexpr (NZ == 13) {
                if (feat == FEAT_BANDS) launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                else if (feat == FEAT_DCTC) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                else if (feat == FEAT_LP) launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                else if (feat == FEAT_LPD) launch_fe(e, &frontend_kernel<NZ, FEAT_LPD, MODE, false, 16, GEN_FULL, 0, false, false, true>, grid, s, kp);
                else throw std::runtime_error("internal: SS engine without an SS instantiation");
            }
            else throw std::runtime_error("internal: SS engine without an SS instantiation");
        }
        else throw std::runtime_error("internal: SS engine without an SS instantiation");
    }
    else if (e->vf) {
        if (!(kp.e_mode == 0 && kp.fb_power && kp.remove_dc && !kp.fb_inld && feat == FEAT_DCTC && narrow && (MODE == 1 || NZ == 13) && e->md))
            throw std::runtime_error("internal: VF engine without the VF instantiation");
        if constexpr (MODE == 1 || NZ == 13) {
            if (kp.nr_exten) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_EXTEN, 0, true, true>, grid, s, kp);
            else launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_PLAIN, 0, true, true>, grid, s, kp);
        }
    }
    else if (e->md && feat == FEAT_LP) {  // the compressed-band LP chain: lags by the MFMA tail, the recursions in lp_tail_kernel
        if (!(base && kp.fb_inld && !kp.nr_exten && narrow)) throw std::runtime_error("internal: MD tables without the MD instantiation");
        if (kp.lporder == 12 && kp.ncep == 12 && !kp.lp_is_lpa) launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, false, 16, GEN_INLD, 12, true>, grid, s, kp);
        else launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, false, 16, GEN_INLD, 0, true>, grid, s, kp);
    }
    else if (e->md) {
        if (!(base && !kp.fb_inld && feat == FEAT_DCTC && narrow)) throw std::runtime_error("internal: MD tables without the MD instantiation");
        if (kp.nr_exten) launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_EXTEN, 0, true>, grid, s, kp);
        else launch_fe(e, &frontend_kernel<NZ, FEAT_DCTC, MODE, false, 16, GEN_PLAIN, 0, true>, grid, s, kp);
    }
    else if (vx) launch_nz<NZ, MODE, true, GEN_FULL>(e, grid, s, kp);
    else if (base && !kp.fb_inld && !kp.nr_exten && feat != FEAT_LPD) launch_nz<NZ, MODE, false, GEN_PLAIN>(e, grid, s, kp);
    else if (base && kp.fb_inld && !kp.nr_exten && narrow && feat == FEAT_LP && kp.lporder == 12 && kp.ncep == 12 && !kp.lp_is_lpa)
        // the PLP preset exactly: order and number of cepstra fixed at compile time (LPO)
        launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, false, 16, GEN_INLD, 12>, grid, s, kp);
    else if (base && kp.fb_inld && !kp.nr_exten && narrow && feat == FEAT_LP) launch_fe(e, &frontend_kernel<NZ, FEAT_LP, MODE, false, 16, GEN_INLD>, grid, s, kp);
    else if (base && !kp.fb_inld && kp.nr_exten && narrow && feat == FEAT_BANDS) launch_fe(e, &frontend_kernel<NZ, FEAT_BANDS, MODE, false, 16, GEN_EXTEN>, grid, s, kp);
    else launch_nz<NZ, MODE, false, GEN_FULL>(e, grid, s, kp);
}

std::vector<std::string> to_args(int argc, const char *const *argv) {
    std::vector<std::string> a;
    for (int i = 0; i < argc; i++) a.emplace_back(argv[i] ? argv[i] : "");
    return a;
}

void fill_dims(const ctu::Design &d, ctu_dims *out) {
    out->fs = d.o.fs;
    out->window = d.window;
    out->wshift = d.wshift;
    out->wfft = d.wfft;
    out->nbins = d.K;
    out->nbands = d.B;
    out->row_floats = d.D;
    out->htk_kind = d.htk_kind;
    out->htk_period = d.period;
    out->has_vad = d.o.do_vad() ? 1 : 0;
    out->swap_out = d.o.swap_out ? 1 : 0;
    out->pcm_align = PCM_ALIGN;
    out->signal_out = d.signal_out ? 1 : 0;
}

}  // namespace

extern "C" {

const char *ctu_create_error(void) { return g_create_error.c_str(); }

int ctu_config_dims(int argc, const char *const *argv, ctu_dims *out) {
    try {
        ctu::Opts o = ctu::Opts::from_args(to_args(argc, argv));
        ctu::Design d(o);
        fill_dims(d, out);
        return CTU_OK;
    } catch (const std::exception &ex) {
        g_create_error = ex.what();
        return CTU_ERR_OPTS;
    }
}

int64_t ctu_config_table(int argc, const char *const *argv, const char *name, double *out, int64_t cap) {
    try {
        ctu::Opts o = ctu::Opts::from_args(to_args(argc, argv));
        ctu::Design d(o);
        std::vector<double> v;
        const std::string n = name ? name : "";
        if (n == "hamming") v = d.hamming;
        else if (n == "fbank") for (const auto &row : d.fb) v.insert(v.end(), row.begin(), row.end());
        else if (n == "fb_first") v.assign(d.fb_first.begin(), d.fb_first.end());
        else if (n == "fb_last") v.assign(d.fb_last.begin(), d.fb_last.end());
        else if (n == "dct") v = d.dct;
        else if (n == "idft") v = d.idft;
        else if (n == "trap") v = d.trap;
        else if (n == "lifter") v = d.lifter;
        else if (n == "phase2_check") {
            Phase2Tables t;
            build_phase2(d, t);
            v = {check_phase2(d, t), (double)t.slot_chunk.back(), (double)t.NS};
        }
        else {
            g_create_error = "unknown table name";
            return CTU_ERR_INPUT;
        }
        for (int64_t i = 0; i < (int64_t)v.size() && i < cap; i++) out[i] = v[i];
        return (int64_t)v.size();
    } catch (const std::exception &ex) {
        g_create_error = ex.what();
        return CTU_ERR_OPTS;
    }
}

int ctu_engine_create(int argc, const char *const *argv, int device, ctu_engine **out) {
    if (!out) return CTU_ERR_INPUT;
    *out = nullptr;
    std::unique_ptr<ctu_engine> e(new ctu_engine);
    try {
        ctu::Opts o = ctu::Opts::from_args(to_args(argc, argv));
        e->design.reset(new ctu::Design(o));
        // FFT sizes below 256 (windows up to 128 samples): the N-point spectrum of a frame is every (256/N)-th bin of its
        // 256-point spectrum, because the frame is zero beyond the window either way.  The engine runs the 256-point mode
        // with the filter bank spread onto those bins (zero weight in between); sums over bins step by kstride.
        ctu::Design &d = *e->design;
        e->user_wfft = d.wfft;  // what ctu_engine_dims reports: the configuration's own FFT size and bin count (= ctu_config_dims)
        e->user_K = d.K;
        if (d.wfft < 256 && d.wfft >= 32 && !d.signal_out) {
            const int S = 256 / d.wfft;
            e->kstride = S;
            for (auto &row : d.fb) {
                std::vector<double> wide(129, 0.0);
                for (int k = 0; k < d.K; k++) wide[(size_t)k * S] = row[k];
                row.swap(wide);
            }
            for (int &f : d.fb_first) f *= S;
            for (int &l : d.fb_last) l *= S;
            d.K = 129;
            d.wfft = 256;
        }
    } catch (const std::exception &ex) {
        g_create_error = ex.what();
        return CTU_ERR_OPTS;
    }
    {
        const ctu::Opts &o = e->design->o;
        if (o.do_vad()) {  // constructor checks of src/vad/vad.cc:639-660, :149-195 and src/vad/vad.h:96-99
            const char *bad = nullptr;
            if (o.vad_cri_mode != "energy" && o.vad_cri_mode != "cepdist") bad = "VAD: unknown vad_cri_mode!";
            else if (o.vad_thr_mode != "absolute" && o.vad_thr_mode != "perc" && o.vad_thr_mode != "adapt" && o.vad_thr_mode != "dyn") bad = "VAD: unknown vad_thr_mode!";
            else if (o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode != "lpc" && o.vad_cepdist_mode != "fea" && o.vad_cepdist_mode != "in") bad = "VADcri_cepdist: unknown vad_cepdist_mode!";
            else if (o.vad_cri_mode == "cepdist" && o.vad_cepdist_mode == "lpc" && !o.phase_needed) bad = "VADcri_cepdist: cannot perform iFFT!";
            else if (o.vad_filter_order < 1 || o.vad_filter_order % 2 == 0) bad = "medianFilter: filter order must be positive, odd number!";
            if (bad) {
                g_create_error = bad;
                return CTU_ERR_OPTS;
            }
        }
    }
    const std::string why = unsupported_reason(*e->design);
    if (!why.empty()) {
        g_create_error = "ENGINE: configuration not on the accelerated path: " + why;
        return CTU_ERR_UNSUPPORTED;
    }
    if (ss_mode_of(e->design->o) && e->design->o.vadmode == "file") {  // hwssNR::hwssNR opens it (nr.cc:205-209); ctu_engine_set_vad_stream replaces it
        FILE *f = std::fopen(e->design->o.filevad.c_str(), "rb");
        if (!f) {
            g_create_error = "NR: Unable to open VAD file!\n";
            return CTU_ERR_OPTS;
        }
        unsigned char tmp[1 << 16];
        size_t got;
        while ((got = std::fread(tmp, 1, sizeof tmp, f)) > 0) e->vad_stream.insert(e->vad_stream.end(), tmp, tmp + got);
        std::fclose(f);
    }
    try {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) throw std::runtime_error("no HIP device (the engine has no CPU fallback)");
        if (device < 0 || device >= ndev) throw std::runtime_error("HIP device ordinal out of range");
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        e->device = device;
        e->n_cu = prop.multiProcessorCount;
        build_tables(e.get());
        {
            const ctu::Opts &o = e->design->o;
            const ctu::Design &d = *e->design;
            e->do_vad = o.do_vad();
            e->per_wave = o.nr_mode == "exten" || e->ss;  // state along an utterance (exten, the *ss modes): a wave per chain
            VadParams &vp = e->vp;
            std::memset(&vp, 0, sizeof vp);
            vp.K = d.K; vp.wfft = d.wfft; vp.window = d.window;
            vp.cri = o.vad_cri_mode == "energy" ? 0 : (o.vad_cepdist_mode == "lpc" ? 1 : 2);
            vp.ncoef = vp.cri == 1 ? o.vad_lpc_coefs : d.nfea;
            vp.thr = o.vad_thr_mode == "absolute" ? 0 : o.vad_thr_mode == "perc" ? 1 : o.vad_thr_mode == "adapt" ? 2 : 3;
            vp.energy_db = o.vad_energy_db; vp.cep_init = o.vad_cepdist_init; vp.filter_order = o.vad_filter_order;
            vp.cep_p = o.vad_cepdist_p; vp.abs_thr = o.vad_absolute_thr; vp.perc_thr = o.vad_perc_thr;
            vp.adapt_q = o.vad_adapt_q; vp.adapt_za = o.vad_adapt_za; vp.dyn_perc = o.vad_dyn_perc; vp.dyn_min = o.vad_dyn_min;
            vp.qmaxinc = o.vad_dyn_qmaxinc; vp.qmaxdec = o.vad_dyn_qmaxdec; vp.qmindec = o.vad_dyn_qmindec; vp.qmininc = o.vad_dyn_qmininc;
            vp.perc_init = o.vad_perc_init; vp.adapt_init = o.vad_adapt_init; vp.dyn_init = o.vad_dyn_init;
            vp.D = d.D; vp.ncep = o.fea_ncepcoefs; vp.c0_slot = d.post_stack ? -2 : (d.row_slot.empty() ? -1 : d.row_slot[0]);  // -2: stacked rows keep the internal order (c0 first)
            vp.delay = 0;
            for (int j = 0; j < d.post_order; j++) vp.delay += d.post_w[j];
            if (d.kind == ctu::FeaKind::TrapDct) vp.delay = (o.fea_trapdct_traplen - 1) / 2;
            vp.e_slot = o.fea_E ? (d.post_order > 0 ? d.D - 1 : d.e_slot) : -1;
            vp.e_delay = (o.vad_filter_order - 1) / 2;
        }
        HIP_TRY(hipEventCreate(&e->ev0));
        HIP_TRY(hipEventCreate(&e->ev1));
    } catch (const std::exception &ex) {
        g_create_error = std::string("ENGINE: ") + ex.what();
        return CTU_ERR_DEVICE;
    }
    *out = e.release();
    return CTU_OK;
}

void ctu_engine_destroy(ctu_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    delete e;
}

const char *ctu_last_error(const ctu_engine *e) { return e ? e->err.c_str() : "null engine"; }

int ctu_engine_dims(const ctu_engine *e, ctu_dims *out) {
    if (!e || !out) return CTU_ERR_INPUT;
    fill_dims(*e->design, out);
    out->wfft = e->user_wfft;
    out->nbins = e->user_K;
    return CTU_OK;
}

int64_t ctu_num_frames(const ctu_engine *e, int64_t n) {
    const int pre = e->design->window - e->design->wshift;
    if (n < pre) return -1;
    return (n - pre) / e->design->wshift;
}

int64_t ctu_arena_layout(const int64_t *utt_nsamples, int32_t n_utt, int64_t *sample_off) {
    if (n_utt < 0 || (n_utt && !utt_nsamples)) return CTU_ERR_INPUT;
    int64_t so = PCM_HEAD;
    for (int i = 0; i < n_utt; i++) {
        if (utt_nsamples[i] < 0) return CTU_ERR_INPUT;
        if (sample_off) sample_off[i] = so;
        so += (utt_nsamples[i] + PCM_ALIGN - 1) / PCM_ALIGN * PCM_ALIGN;
    }
    if (sample_off) sample_off[n_utt] = so;
    return so + PCM_TAIL;  // loads run to the end of the last 32-sample row of a frame
}

int ctu_plan_create(ctu_engine *e, const int64_t *utt_nsamples, int32_t n_utt, ctu_plan **out) {
    if (!e || !out || n_utt < 0 || (n_utt && !utt_nsamples)) return CTU_ERR_INPUT;
    *out = nullptr;
    std::unique_ptr<ctu_plan> pl(new ctu_plan);
    pl->eng = e;
    pl->n_utt = n_utt;
    pl->nsamples.assign(utt_nsamples, utt_nsamples + n_utt);
    pl->sample_off.resize(n_utt + 1);
    pl->row_off.resize(n_utt + 1);
    pl->frames.resize(n_utt);
    const ctu::Design &d = *e->design;
    int64_t ro = 0;
    pl->total_samples = ctu_arena_layout(utt_nsamples, n_utt, pl->sample_off.data());
    if (pl->total_samples < 0) {
        set_error(e, "ENGINE: negative utterance length");
        return CTU_ERR_INPUT;
    }
    std::vector<TileRec> tiles;
    std::vector<int> uts(n_utt + 1, 0);
    std::vector<int4> uinfo(n_utt);
    std::vector<int> chunks;
    const int trap_chunk = 64;  // output frames per TRAP workgroup
    for (int i = 0; i < n_utt; i++) {
        const int64_t T = ctu_num_frames(e, utt_nsamples[i]);
        if (T < 0) {
            set_error(e, "IO: Signal shorter than one frame!");  // src/io/in.cc:277
            return CTU_ERR_INPUT;
        }
        if (d.post_order > 0 && T > 0) {
            int wmax = 0;
            for (int j = 0; j < d.post_order; j++) wmax = std::max(wmax, d.post_w[j]);
            // With exactly window+1 frames a stage never takes its steady-state branch, so its flush starts from ring
            // slot 0 instead of the oldest slot and the emitted rows mix frames (src/fea/fea_delta.cc:118,183-186);
            // with fewer it reads slots that were never written.  Neither is a feature worth reproducing.
            if (T < wmax + 2) {
                set_error(e, "ENGINE: delta / stacking on fewer than window+2 frames is ill-defined in the reference (src/fea/fea_delta.cc:74-130,178-206)");
                return CTU_ERR_INPUT;
            }
        }
        if (d.kind == ctu::FeaKind::TrapDct && T > 0 && T < (d.o.fea_trapdct_traplen + 1) / 2) {
            set_error(e, "ENGINE: trapdct on fewer than (traplen+1)/2 frames is undefined in the reference (src/fea/fea_trap.cc:64-70)");
            return CTU_ERR_INPUT;
        }
        const int64_t so = pl->sample_off[i];
        pl->row_off[i] = ro;
        pl->frames[i] = T;
        uts[i] = (int)tiles.size();
        for (int64_t t0 = 0; t0 < T; t0 += TILE) {
            TileRec r;
            r.sbase = so + t0 * d.wshift;
            r.rbase = ro + t0;
            r.nvalid = (int)std::min<int64_t>(TILE, T - t0);
            r.t0 = (int)t0;
            r.next = -1;
            r.T = (int)T;
            tiles.push_back(r);
        }
        uinfo[i] = make_int4((int)(ro & 0xffffffff), (int)(ro >> 32), (int)T, 0);
        for (int64_t tc = 0; tc < T; tc += trap_chunk) {
            chunks.push_back(i);
            chunks.push_back((int)tc);
        }
        ro += T;
    }
    uts[n_utt] = (int)tiles.size();
    pl->row_off[n_utt] = ro;
    pl->total_frames = ro;
    pl->n_tiles = (int)tiles.size();
    // Each workgroup walks a chain of tiles.  Stateless chains stride over the tile list; with a
    // per-utterance recurrence (exten) a workgroup takes whole utterances, tile after tile.
    std::vector<int> wg_first;
    // workgroups that fit a CU at once: two when the instantiation keeps to 128 VGPRs and 80 KB of LDS; the synthesis and the
    // 512-point detector instantiations take 256 VGPRs (fe_waves_per_simd)
    const bool wide_regs = (e->sy && CTU_SY_LB < 4) || ((e->vf || e->ss) && (!e->mode || CTU_VF1_LB < 4));
    const int max_wg = e->n_cu * ((CTU_LB >= 4 && e->lds_bytes <= (size_t)LDS_2WG && !wide_regs) ? 2 : 1);
    if (e->per_wave) {
        // chains per wave: whole utterances, longest first onto the least loaded chain (LPT), tile after tile
        std::vector<int> live;  // utterances that have at least one frame
        for (int i = 0; i < n_utt; i++)
            if (uts[i + 1] > uts[i]) live.push_back(i);
        std::stable_sort(live.begin(), live.end(), [&](int a, int b) { return pl->frames[a] > pl->frames[b]; });
        // (bigfft_kernel walks a chain with a whole workgroup of 256 threads: eight of them fit a CU at once)
        const int slots = (e->big && !e->wave1k) ? e->n_cu * 8 : max_wg * NWAVE;
        const int C = std::max(1, std::min<int>((int)live.size(), slots));
        const int G = (C + NWAVE - 1) / NWAVE;
        wg_first.assign((size_t)G * NWAVE, -1);
        std::vector<int> tail(C, -1);  // last tile of each chain so far
        std::priority_queue<std::pair<int64_t, int>, std::vector<std::pair<int64_t, int>>, std::greater<std::pair<int64_t, int>>> load;
        // chain c lives in wave c / G of workgroup c % G: the chains of one workgroup are spread over the length ranks
        for (int c = 0; c < C; c++) load.push({0, c});
        for (int u : live) {
            const auto top = load.top();
            load.pop();
            const int c = top.second, slot = (c % G) * NWAVE + c / G;
            for (int t = uts[u]; t + 1 < uts[u + 1]; t++) tiles[t].next = t + 1;
            tiles[uts[u + 1] - 1].next = -1;
            if (tail[c] < 0) wg_first[slot] = uts[u];
            else tiles[tail[c]].next = uts[u];
            tail[c] = uts[u + 1] - 1;
            load.push({top.first + pl->frames[u], c});
        }
        pl->grid = G;
    } else {
        const int G = std::max(1, std::min(pl->n_tiles, max_wg));
        wg_first.assign(G, -1);
        for (int t = 0; t < pl->n_tiles; t++) tiles[t].next = (t + G < pl->n_tiles) ? t + G : -1;
        for (int g = 0; g < G && g < pl->n_tiles; g++) wg_first[g] = g;
        pl->grid = G;
    }
    try {
        HIP_TRY(hipSetDevice(e->device));
        pl->tiles.upload(tiles);
        pl->wg_first.upload(wg_first);
        if (e->ss) {
            std::vector<int> tu(tiles.size());
            for (int i = 0; i < n_utt; i++)
                for (int t = uts[i]; t < uts[i + 1]; t++) tu[t] = i;
            pl->tile_utt.upload(tu);
            pl->ss_seed.alloc((size_t)std::max(n_utt, 1) * d.K);
            pl->ss_last.alloc((size_t)std::max(n_utt, 1) * d.K);
            pl->ss_dirty.alloc((size_t)std::max(n_utt, 1));
            pl->ss_vbits.alloc((size_t)std::max<int64_t>(ro, 1));
        }
        if (e->do_vad) {
            pl->d_row_off.upload(pl->row_off);
            if (e->vp.cri == 1) {
                if (!e->vf) {
                    pl->xri.alloc((size_t)ro * d.K);
                    pl->pnr.alloc((size_t)ro * d.K);
                }
                if (!e->vf) pl->vad_ci.alloc((size_t)ro * e->vp.ncoef);
                else {
                    pl->vad_cf.alloc((size_t)std::max<int64_t>(ro, 1) * VFC_STRIDE);
                    std::vector<int> ord;
                    for (int i = 0; i < n_utt; i++)
                        if (pl->frames[i] > 0) ord.push_back(i);
                    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return pl->frames[a] > pl->frames[b]; });
                    pl->n_live = (int)ord.size();
                    if (ord.empty()) ord.push_back(0);
                    pl->vf_order.upload(ord);
                }
            } else if (e->vp.cri == 0) pl->pnr.alloc((size_t)ro);
        }
        if (d.signal_out) {
            if (!e->sy) {
                pl->xri.alloc((size_t)ro * d.K);
                pl->pnr.alloc((size_t)ro * d.K);
            }
            pl->utt_info.upload(uinfo);
            std::vector<long long> so64(pl->sample_off.begin(), pl->sample_off.end());
            pl->d_sample_off.upload(so64);
            pl->out_samples.resize(n_utt);
            for (int i = 0; i < n_utt; i++) pl->out_samples[i] = pl->frames[i] * d.wshift + (d.window - d.wshift);
            pl->ybuf.alloc((size_t)ro * d.window);
        }
        for (int i = 0; i < n_utt; i++) pl->max_frames = std::max<int>(pl->max_frames, (int)pl->frames[i]);
        if (d.o.remove_dc1) {
            pl->utt_info.upload(uinfo);
            std::vector<long long> so64(pl->sample_off.begin(), pl->sample_off.end());
            pl->d_sample_off.upload(so64);
            pl->dc1m.alloc((size_t)std::max<int64_t>(ro, 1));
            pl->dc1.alloc((size_t)std::max<int64_t>(ro, 1));
        }
        if (d.kind == ctu::FeaKind::TrapDct || d.post_order > 0 || d.cms || d.o.stat_cmvn || d.o.apply_cmvn) {
            pl->utt_info.upload(uinfo);
            pl->trap_chunks.upload(chunks);
            pl->n_trap_chunks = (int)chunks.size() / 2;
            std::vector<int> c128;
            for (size_t k = 0; k + 1 < chunks.size(); k += 2)
                if (!(chunks[k + 1] & 64)) {
                    c128.push_back(chunks[k]);
                    c128.push_back(chunks[k + 1]);
                }
            pl->n_trap_chunks128 = (int)c128.size() / 2;
            if (c128.empty()) c128.assign(2, 0);
            pl->trap_chunks128.upload(c128);
        }
        if (d.kind == ctu::FeaKind::TrapDct) pl->logmel.alloc((size_t)ro * d.B);
        if ((d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) && (!e->big || e->wave1k))
            pl->lp_r.alloc(((size_t)std::max<int64_t>(ro, 1) * (d.o.fea_lporder + 1) * ((e->feat == FEAT_LPD || e->big) ? 8 : 4) + 7) / 8);
        if (d.post_order > 0 || d.cms) pl->base_rows.alloc((size_t)ro * d.Dbase);
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    *out = pl.release();
    return CTU_OK;
}

void ctu_plan_destroy(ctu_plan *p) { delete p; }
const int64_t *ctu_plan_sample_offsets(const ctu_plan *p) { return p->sample_off.data(); }
const int64_t *ctu_plan_row_offsets(const ctu_plan *p) { return p->row_off.data(); }
int64_t ctu_plan_total_samples(const ctu_plan *p) { return p->total_samples; }
int64_t ctu_plan_total_frames(const ctu_plan *p) { return p->total_frames; }

void ctu_vad_ring_step(int32_t order, int64_t frames, int32_t *hidx, int32_t *hsize) {
    // medianFilter::push / flush_frame / cleanFilter as VAD::process_frame, BATCH::flush_vad and VAD::clean drive them over one file
    // (src/vad/vad.h:110-175, src/vad/vad.cc:692-699,705-708,742-745): cleanFilter resets neither historyIdx nor historySize
    if (order < 1 || !hidx || !hsize) return;
    const int delay = (order - 1) / 2;
    int hi = ((*hidx % order) + order) % order, hs = *hsize;
    bool ready = false;
    for (int64_t t = 0; t < frames; t++) {
        hi = (hi + 1) % order;
        if (hs < delay) hs++;
        else ready = true;
    }
    for (;;) {
        if (hs > 0) {
            hi = (hi + 1) % order;
            hs--;
        } else
            ready = false;
        if (!ready) break;
    }
    *hidx = hi;
    *hsize = hs;
}

int64_t ctu_vad_ring_rows(int32_t order, int64_t frames, int32_t hidx0, int32_t *src) {
    // the ring of one file: push t writes slot (hidx0 + t) % order; the k-th written row reads slot k % order - at push k + delay, or after
    // the last push for the rows of the flush (src/vad/vad.h:126-175 with `start` reset by cleanFilter, historyIdx not)
    if (order < 1 || frames < 0) return CTU_ERR_INPUT;
    const int delay = (order - 1) / 2;
    if (frames <= delay) return 0;  // the filter never gets ready: nothing is written
    if (!src) return frames;
    const int h0 = ((hidx0 % order) + order) % order;
    std::vector<int32_t> slot((size_t)order, -1);
    int64_t k = 0;
    for (int64_t t = 0; t < frames; t++) {
        slot[(size_t)((h0 + t) % order)] = (int32_t)t;
        if (t >= delay) {
            src[k] = slot[(size_t)(k % order)];
            k++;
        }
    }
    for (; k < frames; k++) src[k] = slot[(size_t)(k % order)];
    return frames;
}

int ctu_plan_set_vad_ring(ctu_plan *pl, const int32_t *hidx) {
    if (!pl) return CTU_ERR_INPUT;
    ctu_engine *e = pl->eng;
    const ctu::Design &d = *e->design;
    pl->ring_hidx.clear();
    if (!hidx || !e->do_vad || d.o.vad_filter_order <= 1) return CTU_OK;
    const int order = d.o.vad_filter_order, delay = (order - 1) / 2;
    bool any = false;
    for (int i = 0; i < pl->n_utt; i++) any = any || (hidx[i] % order) != 0;
    if (!any) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        pl->ring_hidx.assign(hidx, hidx + pl->n_utt);
        std::vector<int> src((size_t)std::max<int64_t>(pl->total_frames, 1), -1);
        std::vector<int32_t> rel;
        for (int i = 0; i < pl->n_utt; i++) {
            const int64_t T = pl->frames[i], r0 = pl->row_off[i];
            if (T <= delay) continue;  // nothing is written for it
            rel.assign((size_t)T, -1);
            ctu_vad_ring_rows(order, T, hidx[i], rel.data());
            for (int64_t k = 0; k < T; k++) src[(size_t)(r0 + k)] = rel[(size_t)k] < 0 ? -1 : (int)(r0 + rel[(size_t)k]);
        }
        pl->ring_src.upload(src);
        if (pl->ring_tmp.n < (size_t)std::max<int64_t>(pl->total_frames, 1) * d.D) pl->ring_tmp.alloc((size_t)std::max<int64_t>(pl->total_frames, 1) * d.D);
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        pl->ring_hidx.clear();
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

int ctu_engine_run(ctu_engine *e, const ctu_plan *pl, const int16_t *d_pcm, float *d_rows, uint8_t *d_vad, void *stream) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    if (pl->n_tiles == 0) return CTU_OK;
    const ctu::Design &d = *e->design;
    const bool signal = d.signal_out;
    if (signal && !e->in_signal_call) {
        set_error(e, "ENGINE: this configuration writes speech (-format_out raw|wave): use ctu_engine_run_signal");
        return CTU_ERR_INPUT;
    }
    if (!d_pcm || (!signal && !d_rows) || (e->do_vad && !d_vad)) {
        set_error(e, "ENGINE: null device buffer");
        return CTU_ERR_INPUT;
    }
    hipStream_t s = (hipStream_t)stream;
    try {
        HIP_TRY(hipSetDevice(e->device));
        KParams kp;
        std::memset(&kp, 0, sizeof kp);
        kp.pcm = d_pcm;
        kp.rows = (d.post_order > 0 || d.cms) ? pl->base_rows.p : d_rows;
        kp.logmel = pl->logmel.p;
        kp.xri = pl->xri.p;
        kp.pnr = pl->pnr.p;
        kp.band_log = d.kind != ctu::FeaKind::Spec;
        kp.band_to_scratch = d.kind == ctu::FeaKind::TrapDct;
        kp.lp_is_lpa = d.kind == ctu::FeaKind::Lpa;
        kp.ybuf = pl->ybuf.p;
        kp.syn_scale = 1.0f / (float)d.wfft;
        kp.vad_export = (signal && !e->sy) ? 1 : ((!e->do_vad || e->vf || signal) ? 0 : (e->vp.cri == 1 ? 1 : (e->vp.cri == 0 ? 2 : 0)));
        kp.vad_ci = pl->vad_ci.p;
        kp.vad_nc = e->vp.ncoef;
        kp.ss_mode = e->ss;
        kp.ss_init = d.o.nr_initsegs;
        kp.ss_nc = d.o.fea_ncepcoefs;
        kp.han_off = e->han_off;
        kp.nr_b = (float)d.o.nr_b;
        kp.ss_q = d.o.nr_q;
        kp.ss_seed = pl->ss_seed.p;
        kp.ss_last = pl->ss_last.p;
        kp.tile_utt = pl->tile_utt.p;
        kp.vad_cf = pl->vad_cf.p;
        kp.skip_phase2 = signal ? 1 : 0;
        kp.tiles = pl->tiles.p;
        kp.wg_first = pl->wg_first.p;
        kp.lanec = e->lanec.p;
        kp.ftab = e->ftab.p;
        kp.itab = e->itab.p;
        kp.tab_floats = e->tab_floats;
        kp.ck_off = e->ck_off;
        kp.cf_off = e->cf_off;
        kp.cfd_off = e->cfd_off;
        kp.NS = e->NS;
        kp.CW = e->CW;
        kp.ncoef_out = e->ncoef_out;
        kp.K = d.K;
        kp.window = d.window;
        kp.e_slot = d.e_slot;
        kp.e_mode = 0;
        if (d.o.fea_E) {  // energy routing of src/io/batch.cc:98-119
            if (d.o.fea_rawenergy) kp.e_mode = 4;
            else if (d.kind == ctu::FeaKind::Dctc) kp.e_mode = d.o.nr_when_afterFB ? 5 : 1;
            else if (d.kind == ctu::FeaKind::Lpc || d.kind == ctu::FeaKind::Lpa) kp.e_mode = 2;
            else kp.e_mode = 3;
        }
        if (signal) kp.e_mode = 0;
        kp.wshift = d.wshift;
        kp.B = d.B;
        kp.nfea = d.nfea;
        kp.D = d.Dbase;
        kp.ncep = d.o.fea_ncepcoefs;
        kp.lporder = d.o.fea_lporder;
        kp.lp_r = pl->lp_r.p;
        kp.lp_stride = d.o.fea_lporder + 1;
        kp.lift_off = e->lift_off;
        kp.preem = d.o.preem;
        kp.inv_window = 1.0f / (float)d.window;
        kp.inv_window_d = 1.0 / (double)d.window;
        kp.remove_dc = d.o.remove_dc;
        kp.kstride = e->kstride;
        kp.remove_dc1 = d.o.remove_dc1 ? 1 : 0;
        kp.dc1_J = d.window / d.wshift;
        kp.dc1 = pl->dc1.p;
        kp.fb_power = d.o.fb_power;
        kp.fb_inld = d.o.fb_inld;
        kp.lifter_on = d.o.fea_lifter > 1;
        kp.nr_exten = d.o.nr_mode == "exten";
        kp.nr_after_fb = d.o.nr_when_afterFB ? 1 : 0;
        kp.nr_p = (float)d.o.nr_p;
        kp.nr_p_d = d.o.nr_p;
        kp.nr_a = (float)d.o.nr_a;
        kp.per_wave = e->per_wave ? 1 : 0;
        kp.am_off = e->am_off;
#ifdef CTU_DIAG  // phase ablation (1 = phase 1 only, 2 = phase 2 only): diagnostic builds only
        kp.dbg = getenv("CTU_DEBUG_MODE") ? atoi(getenv("CTU_DEBUG_MODE")) : 0;
#endif
        const int grid = pl->grid;
#if CTU_STAMP
        if (e->stamps.n < (size_t)grid * NWAVE * 16) e->stamps.alloc((size_t)grid * NWAVE * 16);
        HIP_TRY(hipMemsetAsync(e->stamps.p, 0, e->stamps.n * 8, s));
        kp.stamps = e->stamps.p;
#endif
        if (d.o.remove_dc1) {
            const int gx = std::max(1, std::min((pl->max_frames + 3) / 4, 64));
            hipLaunchKernelGGL(dc1_means_kernel, dim3(gx, pl->n_utt), dim3(256), 0, s, d_pcm, pl->utt_info.p, pl->d_sample_off.p, pl->dc1m.p, pl->n_utt, d.window, d.wshift);
            hipLaunchKernelGGL(dc1_offsets_kernel, dim3((pl->n_utt + 63) / 64), dim3(64), 0, s, pl->dc1m.p, pl->utt_info.p, pl->dc1.p, pl->n_utt, d.window, d.wshift);
        }
        HIP_TRY(hipEventRecord(e->ev0, s));
        auto launch = [&] {
            // (the 13-row instantiations serve every window of at most 13 rows - the window table is zero beyond the window - and the *ss
            // modes are instantiated for 13 rows only)
            switch (kp.remove_dc1 ? 16 : (e->ss && e->nz <= 13) ? 13 : e->nz) {
                case 13: e->mode ? launch_vx<13, 1>(e, dim3(grid), s, kp) : launch_vx<13, 0>(e, dim3(grid), s, kp); break;
                default: e->mode ? launch_vx<16, 1>(e, dim3(grid), s, kp) : launch_vx<16, 0>(e, dim3(grid), s, kp); break;
            }
        };
        if (e->big) {
            BigParams bp;
            std::memset(&bp, 0, sizeof bp);
            bp.pcm = d_pcm; bp.rows = kp.rows; bp.logmel = kp.logmel; bp.tiles = pl->tiles.p; bp.n_tiles = pl->n_tiles;
            bp.win = e->big_win.p; bp.tw = e->big_tw.p; bp.fbw = e->big_fbw.p; bp.fb_range = e->big_range.p;
            bp.coef = e->big_coef.p; bp.coef_d = e->big_coef_d.p; bp.lifter = e->big_lifter.p; bp.row_slot = e->big_slot.p;
            bp.wfft = d.wfft; bp.K = d.K; bp.window = d.window; bp.wshift = d.wshift; bp.B = d.B; bp.D = kp.D;
            bp.ncoef_out = e->ncoef_out; bp.feat = e->feat; bp.e_mode = kp.e_mode; bp.e_slot = kp.e_slot;
            bp.remove_dc = kp.remove_dc; bp.fb_power = kp.fb_power; bp.fb_inld = kp.fb_inld; bp.band_log = kp.band_log;
            bp.band_to_scratch = kp.band_to_scratch; bp.lp_is_lpa = kp.lp_is_lpa; bp.lporder = kp.lporder; bp.ncep = kp.ncep;
            bp.lifter_on = kp.lifter_on; bp.preem = kp.preem;
            bp.fb_total = e->big_fb_total;
            bp.seg = e->big_seg.p;
            bp.nr_exten = kp.nr_exten; bp.nr_p = kp.nr_p; bp.nr_a = kp.nr_a;
            bp.vad_en = (e->do_vad && e->vp.cri == 0) ? pl->pnr.p : nullptr;
            bp.dc1 = kp.remove_dc1 ? pl->dc1.p : nullptr; bp.dc1_J = kp.dc1_J;
            bp.xri = signal ? pl->xri.p : nullptr; bp.pnr = signal ? pl->pnr.p : nullptr;
            bp.chain_first = pl->wg_first.p; bp.n_chains = (int)pl->wg_first.n;
            const size_t shm = (size_t)d.wfft * 8 + (size_t)((d.K + 3) & ~3) * 4 + 64 * 4 + 4 * 8 + (size_t)d.wfft / 2 * 8 +
                               (size_t)((d.window + 3) & ~3) * 4 + (size_t)((e->big_fb_total + 3) & ~3) * 4 +
                               (size_t)(e->feat == FEAT_LP ? (d.o.fea_lporder + 1) * d.B : 0) * 8 +
                               (size_t)(((e->feat == FEAT_DCTC ? e->ncoef_out * d.B : 0) + 3) & ~3) * 4 + (size_t)((3 * d.B + 3) & ~3) * 4 + 64 * 4;
            if (shm > 160 * 1024) throw std::runtime_error("filter bank too wide for the LDS tables of the large-FFT kernel");
            const void *kfn = kp.nr_exten ? (d.wfft == 1024 ? (const void *)bigfft_kernel<4, true> : d.wfft == 2048 ? (const void *)bigfft_kernel<8, true> : (const void *)bigfft_kernel<16, true>)
                                          : (d.wfft == 1024 ? (const void *)bigfft_kernel<4> : d.wfft == 2048 ? (const void *)bigfft_kernel<8> : (const void *)bigfft_kernel<16>);
            if (shm > 64 * 1024 && !e->attr_done.count(kfn)) {
                HIP_TRY(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                e->attr_done.insert(kfn);
            }
            const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / shm));
            const int g = std::max(1, std::min(pl->n_tiles, e->n_cu * per_cu));
            if (e->wave1k) {
                // 1024 points: one wave per frame, the transform in registers (wave1k_kernel.h); tiles are dealt to waves
                const size_t wshm = ((size_t)W1K_WAVES * W1K_WAVE_FLOATS + 1024 + W1K_TW_FLOATS + (size_t)((e->big_fb_total + 3) & ~3) + 2) * 4 +
                                    (size_t)(e->feat == FEAT_LP ? (d.o.fea_lporder + 1) * d.B : 0) * 8 +
                                    (size_t)(((e->feat == FEAT_DCTC ? e->ncoef_out * d.B : 0) + 3) & ~3) * 4 + (size_t)((3 * d.B + 3) & ~3) * 4 + (size_t)(256 + 2 * d.B) * 4 + 64;
                if (wshm > 160 * 1024) throw std::runtime_error("filter bank too wide for the LDS tables of the 1024-point kernel");
                const void *wfn = bp.nr_exten ? (const void *)wave1k_kernel<true> : (const void *)wave1k_kernel<false>;
                if (wshm > 64 * 1024 && !e->attr_done.count(wfn)) {
                    HIP_TRY(hipFuncSetAttribute(wfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                    e->attr_done.insert(wfn);
                }
                const int wper_cu = (int)std::max<size_t>(1, std::min<size_t>(CTU_W1K_LB * 4 / W1K_WAVES, (160 * 1024) / wshm));
                if (bp.nr_exten && !e->per_wave) throw std::runtime_error("internal: exten without per-wave chains");
                // exten: one wave per chain of utterances, every chain gets its wave whatever fits the chip at once
                const int wg = bp.nr_exten ? std::max(1, (bp.n_chains + W1K_WAVES - 1) / W1K_WAVES)
                                           : std::max(1, std::min((pl->n_tiles + W1K_WAVES - 1) / W1K_WAVES, e->n_cu * wper_cu));
                if (bp.nr_exten) hipLaunchKernelGGL(wave1k_kernel<true>, dim3(wg), dim3(64 * W1K_WAVES), wshm, s, bp, (void *)pl->lp_r.p, d.o.fea_lporder + 1);
                else hipLaunchKernelGGL(wave1k_kernel<false>, dim3(wg), dim3(64 * W1K_WAVES), wshm, s, bp, (void *)pl->lp_r.p, d.o.fea_lporder + 1);
            }
            else if (bp.nr_exten) {  // one workgroup per chain of utterances
                if (!e->per_wave) throw std::runtime_error("internal: exten without chains");
                const dim3 gx((unsigned)std::max(1, bp.n_chains));
                if (d.wfft == 1024) hipLaunchKernelGGL((bigfft_kernel<4, true>), gx, dim3(256), shm, s, bp);
                else if (d.wfft == 2048) hipLaunchKernelGGL((bigfft_kernel<8, true>), gx, dim3(256), shm, s, bp);
                else hipLaunchKernelGGL((bigfft_kernel<16, true>), gx, dim3(256), shm, s, bp);
            }
            else if (d.wfft == 1024) hipLaunchKernelGGL((bigfft_kernel<4>), dim3(g), dim3(256), shm, s, bp);
            else if (d.wfft == 2048) hipLaunchKernelGGL((bigfft_kernel<8>), dim3(g), dim3(256), shm, s, bp);
            else hipLaunchKernelGGL((bigfft_kernel<16>), dim3(g), dim3(256), shm, s, bp);
        }
        else if (!e->ss) launch();
        else {
            // hwss / fwss / 2fwss: a file's noise estimate starts from the vector the previous file of the list left
            // behind (src/nr/nr.cc:212-221), which chains the whole list.  Everything but that seed is independent per
            // file, so the list is run with the seeds known so far until they stop changing: pass j fixes the seeds of
            // the first j files for good, and a seed's influence dies out as p^(noise updates), so real lists settle in
            // two or three passes.  Synchronous: each pass reads the vectors back.
            const size_t nk = (size_t)pl->n_utt * d.K;
            std::vector<float> seed(nk, 0.f), last(nk, 0.f), next(nk, 0.f);
            if (e->ss_stale.size() != (size_t)d.K) e->ss_stale.assign(d.K, 0.f);
            // new_file() seeds the estimate from the vector and then scales the vector by 0.1 (nr.cc:217-220, 402-407); a
            // file with a frame overwrites it in its first get_frame(), a file without one (window - wshift <= N < window)
            // leaves it scaled.  So file i starts from what the last file with a frame ahead of it left - or, ahead of the
            // first such file, what the previous run left (the reference keeps the vector for the life of the process,
            // base/types.h:35-38) - times 0.1 for every frameless file in between.
            auto scaled = [&](float *dst, const float *src, int skipped) {
                double f = 1.0;
                for (int z = 0; z < skipped; z++) f *= 0.1;
                for (int k = 0; k < d.K; k++) dst[k] = (float)((double)src[k] * f);
            };
            auto propagate = [&](std::vector<float> &dst) {  // seeds of every file from `last` (files with frames) and e->ss_stale
                int prev = -1, skipped = 0;
                for (int i = 0; i < pl->n_utt; i++) {
                    scaled(&dst[(size_t)i * d.K], prev >= 0 ? &last[(size_t)prev * d.K] : e->ss_stale.data(), skipped);
                    if (pl->frames[i] > 0) {
                        prev = i;
                        skipped = 0;
                    } else
                        skipped++;
                }
                return std::make_pair(prev, skipped);
            };
            propagate(seed);  // `last` is still zero: only the seeds ahead of the first file with a frame are final
            // An utterance's rows and last vector depend on its samples and its seed only: a pass recomputes just the utterances
            // whose seed changed since the pass before (all of them in the first one).
            std::vector<unsigned char> dirty(std::max(pl->n_utt, 1), 1);
            kp.ss_dirty = pl->ss_dirty.p;
            kp.ss_vbits = pl->ss_vbits.p;
            if (e->ss_file) {
                // -vad file=<f>: `char vad = fgetc(fvad); if (vad != EOF) return bool(vad); else throw` (nr.cc:297-302), one byte per frame in
                // list order, the stream running on from file to file and from run to run.  Every byte but NUL is speech; a byte 0xFF
                // compares equal to EOF in the reference's (signed) char and ends the run like the end of the file does.
                const int64_t nf = pl->total_frames;
                std::vector<unsigned char> vb((size_t)std::max<int64_t>(nf, 1), 0);
                for (int64_t i = 0; i < nf; i++) {
                    if (e->vad_pos + i >= (int64_t)e->vad_stream.size() || e->vad_stream[(size_t)(e->vad_pos + i)] == 0xFF) {
                        set_error(e, "NR: Unexpected end of VAD file!");
                        return CTU_ERR_INPUT;
                    }
                    vb[(size_t)i] = e->vad_stream[(size_t)(e->vad_pos + i)] != 0;
                }
                HIP_TRY(hipMemcpyAsync(pl->ss_vbits.p, vb.data(), (size_t)nf, hipMemcpyHostToDevice, s));
                HIP_TRY(hipStreamSynchronize(s));  // vb leaves scope
                e->vad_pos += nf;
            }
            for (int iter = 0;; iter++) {
                // the detector never sees a subtracted spectrum (nr.cc:278-295): its decisions are those of the first pass
                kp.ss_cached = e->ss_file ? 1 : (iter > 0 ? CTU_SS_CACHE : 0);
                HIP_TRY(hipMemcpyAsync(pl->ss_seed.p, seed.data(), nk * sizeof(float), hipMemcpyHostToDevice, s));
                HIP_TRY(hipMemcpyAsync(pl->ss_dirty.p, dirty.data(), dirty.size(), hipMemcpyHostToDevice, s));
                launch();
                HIP_TRY(hipMemcpyAsync(last.data(), pl->ss_last.p, nk * sizeof(float), hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                propagate(next);
                bool any = false;
                for (int i = 0; i < pl->n_utt; i++) {
                    dirty[i] = std::memcmp(&next[(size_t)i * d.K], &seed[(size_t)i * d.K], d.K * sizeof(float)) != 0;
                    any = any || dirty[i];
                }
                if (!any) break;
                if (iter > pl->n_utt) throw std::runtime_error("internal: noise seeds of the *ss chain did not settle");
                seed = next;
            }
            {   // what this run leaves for the next one
                const auto tail = propagate(next);
                std::vector<float> keep(d.K);
                scaled(keep.data(), tail.first >= 0 ? &last[(size_t)tail.first * d.K] : e->ss_stale.data(), tail.second);
                e->ss_stale = keep;
            }
        }
        HIP_TRY(hipEventRecord(e->ev1, s));
        e->timed = true;
        e->host_timed = false;
        HIP_TRY(hipGetLastError());
#if CTU_STAMP
        if (const char *sf = getenv("CTU_STAMP_FILE")) {
            HIP_TRY(hipStreamSynchronize(s));
            std::vector<unsigned long long> h(e->stamps.n);
            HIP_TRY(hipMemcpy(h.data(), e->stamps.p, h.size() * 8, hipMemcpyDeviceToHost));
            double sum[16] = {0};
            for (size_t w = 0; w < (size_t)grid * NWAVE; w++)
                for (int i = 0; i < 16; i++) sum[i] += (double)h[w * 16 + i];
            if (FILE *f = fopen(sf, "w")) {
                double tot = 0;
                for (int i = 0; i < 16; i++) tot += sum[i];
                for (int i = 0; i < 16; i++) fprintf(f, "seg %2d  mean cycles per wave %12.0f  share %.3f\n", i, sum[i] / (grid * NWAVE), sum[i] / tot);
                fclose(f);
            }
        }
#endif
        if ((e->feat == FEAT_LP || e->feat == FEAT_LPD) && (!e->big || e->wave1k) && !signal) {
            LpTailParams tp;
            tp.lags = pl->lp_r.p;
            tp.rows = kp.rows;
            tp.lifter = e->big ? e->big_lifter.p : e->ftab.p + e->lift_off;
            tp.row_slot = e->big ? e->big_slot.p : e->itab.p + e->NS + 1;
            tp.total_frames = pl->total_frames;
            tp.stride = kp.lp_stride; tp.D = kp.D; tp.lporder = kp.lporder; tp.ncep = kp.ncep; tp.is_lpa = kp.lp_is_lpa;
            tp.lifter_on = kp.lifter_on; tp.e_mode = kp.e_mode; tp.e_slot = kp.e_slot;
            tp.inv_stride = (unsigned)((1ull << 32) / (unsigned)tp.stride) + 1u;
            tp.inv_D = (unsigned)((1ull << 32) / (unsigned)tp.D) + 1u;
            const dim3 tg((unsigned)std::max<int64_t>(1, std::min<int64_t>((pl->total_frames + 255) / 256, (int64_t)e->n_cu * 8)));
            const size_t tshm = (size_t)256 * ((tp.stride | 1) * ((e->feat == FEAT_LPD || e->big) ? 8 : 4) + (tp.D | 1) * 4);
            // double lags at orders 20..23 pass 64 KiB (order 23 with -fea_E: 75 KiB)
            if (e->feat == FEAT_LPD || e->big) {
                if (tshm > 64 * 1024) allow_big_lds(e, &lp_tail_kernel<double, 0>);
                hipLaunchKernelGGL((lp_tail_kernel<double, 0>), tg, dim3(256), tshm, s, tp);
            } else if (kp.lporder == 12 && kp.ncep == 12 && !kp.lp_is_lpa) hipLaunchKernelGGL((lp_tail_kernel<float, 12>), tg, dim3(256), tshm, s, tp);
            else {
                if (tshm > 64 * 1024) allow_big_lds(e, &lp_tail_kernel<float, 0>);
                hipLaunchKernelGGL((lp_tail_kernel<float, 0>), tg, dim3(256), tshm, s, tp);
            }
            HIP_TRY(hipGetLastError());
        }
        if (e->do_vad) {
            if (e->vp.cri == 1 && !e->vf) {
                const dim3 g((unsigned)std::min<int64_t>((pl->total_frames + 3) / 4, (int64_t)e->n_cu * 4));
                const size_t bshm = (512 + (size_t)4 * 2 * (d.wfft / 2 + 4)) * 2 * sizeof(vreal);
#define BURG_LAUNCH(Q, NC) hipLaunchKernelGGL((vad_burg_kernel<Q, NC>), g, dim3(256), bshm, s, pl->xri.p, pl->pnr.p, pl->vad_ci.p, e->vp, pl->total_frames)
                if (d.window <= 256) {
                    if (e->vp.ncoef <= 16) BURG_LAUNCH(4, 16);
                    else BURG_LAUNCH(4, 32);
                } else {
                    if (e->vp.ncoef <= 16) BURG_LAUNCH(8, 16);
                    else BURG_LAUNCH(8, 32);
                }
#undef BURG_LAUNCH
            }
            HIP_TRY(hipGetLastError());
        }
        if (d.kind == ctu::FeaKind::TrapDct) {
            const int tl = d.o.fea_trapdct_traplen, nd = d.o.fea_trapdct_ndct;
            const int ns = (tl + 3) / 4;
            const size_t shm = (size_t)(64 + 4 * (ns <= 26 ? 26 : ns)) * (d.B | 1) * sizeof(float);
#define TRAP_LAUNCH(NRB, NSM)                                                                                          \
    hipLaunchKernelGGL((trapdct_mfma_kernel<NRB, NSM>), dim3(pl->n_trap_chunks), dim3(256), shm, s, pl->logmel.p, d_rows, \
                       e->trapG.p, pl->utt_info.p, pl->trap_chunks.p, d.B, tl, nd, d.D)
            if (e->trap_bf16) {
                constexpr int NT16 = CTU_TRAP_F16 ? 2 : 3;
                const size_t shm16 = (size_t)NT16 * d.B * TB_TT * 2 + (size_t)((d.B + 3) & ~3) * 4 + 2 * (4 * NT16 * 64) * 16;
                hipLaunchKernelGGL((trapdct_split16_kernel<CTU_TRAP_F16 != 0>), dim3(std::max(pl->n_trap_chunks128, 1)), dim3(512), shm16, s,
                                   pl->logmel.p, d_rows, e->trapG16.p, pl->utt_info.p, pl->trap_chunks128.p, pl->n_trap_chunks128, d.B, nd, d.D);
            }
            else if (nd <= 16 && ns <= 26) TRAP_LAUNCH(1, 26);
            else if (nd <= 16) TRAP_LAUNCH(1, 64);
            else if (ns <= 26) TRAP_LAUNCH(2, 26);
            else TRAP_LAUNCH(2, 64);
#undef TRAP_LAUNCH
            HIP_TRY(hipGetLastError());
        }
        if (d.post_order > 0) {
            PostParams pp;
            std::memset(&pp, 0, sizeof pp);
            pp.fea_c = d.o.fea_ncepcoefs + 1;
            pp.Dbase = d.Dbase;
            pp.D = d.D;
            pp.order = d.post_order;
            pp.stack = d.post_stack ? 1 : 0;
            pp.has_e = d.o.fea_E ? 1 : 0;
            int H = 0;
            for (int j = 0; j < d.post_order; j++) {
                pp.w[j] = d.post_w[j];
                int den = 0;
                for (int i = 1; i <= d.post_w[j]; i++) den += i * i;
                pp.inv_den[j] = (float)(1.0 / (2.0 * den));
                H += d.post_w[j];
            }
            const int R = 64 + 2 * H;
            const size_t shm = ((size_t)R * d.Dbase + (size_t)(d.post_stack ? 0 : d.post_order) * R * pp.fea_c) * sizeof(float);
            if ((size_t)R * d.Dbase > 256 * 12) throw std::runtime_error("delta tile larger than the prefetch registers");
            const int pgrid = std::min(pl->n_trap_chunks, e->n_cu * 8);
            const bool std39 = !d.post_stack && d.post_order == 2 && pp.fea_c == 13 && d.Dbase == 13 && d.D == 39 && pp.w[0] == 2 && pp.w[1] == 2;
            if (std39)
                hipLaunchKernelGGL(post_kernel<true>, dim3(pgrid), dim3(256), shm, s, pl->base_rows.p, d_rows,
                                   pl->utt_info.p, pl->trap_chunks.p, pl->n_trap_chunks, pp);
            else
                hipLaunchKernelGGL(post_kernel<false>, dim3(pgrid), dim3(256), shm, s, pl->base_rows.p, d_rows,
                                   pl->utt_info.p, pl->trap_chunks.p, pl->n_trap_chunks, pp);
            HIP_TRY(hipGetLastError());
        }
        if (d.cms) {
            CmsParams cp;
            std::memset(&cp, 0, sizeof cp);
            cp.ncols = d.cms_cols;
            cp.Dbase = d.Dbase;
            cp.D = d.D;
            cp.copy_rest = d.post_order > 0 ? 0 : 1;
            cp.L = d.o.length_b;
            cp.z = d.o.fea_Z_exp;
            cp.omz = 1 - d.o.fea_Z_exp;
            if (d.cms == 1)
                hipLaunchKernelGGL(cms_exp_kernel, dim3((pl->n_utt + 1) / 2), dim3(64), 0, s, pl->base_rows.p, d_rows,
                                   pl->utt_info.p, pl->n_utt, cp);
            else
                hipLaunchKernelGGL(cms_block_kernel, dim3(pl->n_trap_chunks), dim3(256),
                                   (size_t)(64 + cp.L - 1) * cp.ncols * sizeof(float), s, pl->base_rows.p, d_rows,
                                   pl->utt_info.p, pl->trap_chunks.p, cp);
            HIP_TRY(hipGetLastError());
        }
        if (e->do_vad && e->vf && pl->n_live > 0) {
            // the fused path left the lattice's output of every frame behind: the coefficient recursion and a -> c one frame per lane,
            // then the detector's recurrences, sixteen utterances per wave
            if (CTU_VF_A2C)
                hipLaunchKernelGGL((vad_a2c_kernel<VF_NC>), dim3((unsigned)((pl->total_frames + 255) / 256)), dim3(256), 0, s, pl->vad_cf.p, (int64_t)pl->total_frames);
#define LANES_LAUNCH(THR) hipLaunchKernelGGL((vad_lanes_kernel<VF_NC, THR>), dim3((pl->n_live + 15) / 16), dim3(64), 0, s, pl->vad_cf.p, pl->vf_order.p, pl->n_live, pl->d_row_off.p, d_vad, e->vp)
            switch (e->vp.thr) {
                case 0: LANES_LAUNCH(0); break;
                case 1: LANES_LAUNCH(1); break;
                case 2: LANES_LAUNCH(2); break;
                default: LANES_LAUNCH(3); break;
            }
#undef LANES_LAUNCH
            HIP_TRY(hipGetLastError());
        }
        if (e->do_vad && !e->vf) {
            // After the post passes: the `fea` criterion reads the vector the writer sees (CMS applied), and the energy
            // column is shifted in the finished rows.
            hipLaunchKernelGGL(vad_decide_kernel, dim3(pl->n_utt), dim3(64), 0, s, pl->vad_ci.p, pl->pnr.p, d_rows,
                               pl->d_row_off.p, pl->n_utt, d_vad, e->vp);
            HIP_TRY(hipGetLastError());
        }
        if (e->do_vad && d.o.vad_filter_order > 1) {
            // A file with no more frames than the majority filter delays never gets the filter `ready` (src/vad/vad.h:126-136), so
            // BATCH::flush_vad's loop does not start (src/vad/vad.cc:742-745, src/io/batch.cc:243-249): the reference writes neither a
            // row nor a decision for it.  Its decision bytes become NUL ("nothing written"); ctu_engine_run_host reports 0 rows.
            hipLaunchKernelGGL(vad_short_files_kernel, dim3((unsigned)((pl->n_utt + 255) / 256)), dim3(256), 0, s, d_vad, pl->d_row_off.p, pl->n_utt,
                               (d.o.vad_filter_order - 1) / 2);
            HIP_TRY(hipGetLastError());
        }
        if (e->do_vad && !pl->ring_hidx.empty() && pl->total_frames > 0) {
            // the list behaviour of the reference's majority filter (ctu_plan_set_vad_ring): every column but the energy (which does not
            // go through the ring, src/io/batch.cc:101-120) of row k := the vector of frame ring_src[k], or zeros
            const size_t n = (size_t)pl->total_frames * d.D;
            HIP_TRY(hipMemcpyAsync(pl->ring_tmp.p, d_rows, n * sizeof(float), hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(vad_ring_gather_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535u * 16)), dim3(256), 0, s, pl->ring_tmp.p, d_rows,
                               pl->ring_src.p, (int64_t)pl->total_frames, d.D, d.o.fea_E ? (d.post_order > 0 ? d.D - 1 : d.e_slot) : -1);
            HIP_TRY(hipGetLastError());
        }
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

void *ctu_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void ctu_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

namespace {
bool is_pinned(const void *p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // pageable memory: not an error
        return false;
    }
    return a.type == hipMemoryTypeHost;
}
}  // namespace

// How many utterance ranges a host-buffer run is cut into: CTU_HOST_CHUNKS if set (1 = one range), else 8 for batches of
// at least 16 utterances and 32 MiB of PCM in page-locked buffers.  Modes whose state crosses utterances (the *ss noise seed) stay in one range.
static int host_chunks(const ctu_engine *e, const ctu_plan *pl, bool pinned) {
    if (e->ss) return 1;
    // pageable buffers go through blocking staged copies: cutting those up only adds calls (measured 1.04e8 -> 0.96e8 frames/s)
    int k = (pinned && pl->n_utt >= 16 && pl->total_samples >= (int64_t)16 << 20) ? 8 : 1;
    if (const char *v = getenv("CTU_HOST_CHUNKS")) k = std::max(1, atoi(v));
    return std::min(k, std::max(1, pl->n_utt));
}

int ctu_engine_run_host(ctu_engine *e, const ctu_plan *pl_, const int16_t *h_pcm, float *h_rows, uint8_t *h_vad,
                        int64_t *rows_per_utt) {
    if (!e || !pl_ || pl_->eng != e) return CTU_ERR_INPUT;
    ctu_plan *pl = const_cast<ctu_plan *>(pl_);  // the device copies live in the plan
    const ctu::Design &d = *e->design;
    if (rows_per_utt)
        for (int i = 0; i < pl->n_utt; i++) rows_per_utt[i] = pl->frames[i];
    if (pl->total_frames == 0) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        // Device buffers are allocated once per plan (per part).  Transfers: a caller buffer from ctu_host_alloc (pinned) is
        // DMA-ed asynchronously at the link rate; pageable memory goes through the runtime's own staging (hipMemcpy).
        const bool pin_in = is_pinned(h_pcm), pin_out = is_pinned(h_rows);
        const int nparts = host_chunks(e, pl, pin_in && pin_out);
        if (nparts > 1 && pl->parts_for != nparts) {  // keyed on the count asked for: fewer ranges may come out (a long utterance spans several)
            pl->parts_for = nparts;
            // ranges of about equal PCM; a part's arena is the slice of the caller's arena that starts PCM_HEAD samples
            // ahead of its first utterance (the layout rule of ctu_plan_create is translation invariant)
            pl->parts.clear();
            pl->part_first.assign(1, 0);
            const int64_t per = (pl->sample_off[pl->n_utt] - pl->sample_off[0] + nparts - 1) / nparts;
            for (int k = 1; k < nparts; k++) {
                int u = pl->part_first.back();
                const int64_t goal = pl->sample_off[0] + per * k;
                while (u < pl->n_utt && pl->sample_off[u] < goal) u++;
                if (u > pl->part_first.back() && u < pl->n_utt) pl->part_first.push_back(u);
            }
            pl->part_first.push_back(pl->n_utt);
            for (size_t k = 0; k + 1 < pl->part_first.size(); k++) {
                ctu_plan *sub = nullptr;
                const int u0 = pl->part_first[k], u1 = pl->part_first[k + 1];
                const int rc = ctu_plan_create(e, pl->nsamples.data() + u0, u1 - u0, &sub);
                if (rc != CTU_OK) return rc;
                pl->parts.emplace_back(sub);
            }
            for (hipStream_t &st : pl->part_stream)
                if (!st) HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        }
        std::vector<uint8_t> v(e->do_vad ? (size_t)pl->total_frames : 0);
        if (nparts > 1 && pl->parts.size() > 1) {
            const int np = (int)pl->parts.size();
            std::vector<std::pair<hipEvent_t, hipEvent_t>> part_events;  // (start, stop) of every range's front-end launch
            auto download = [&](int k) {
                ctu_plan *sp = pl->parts[k].get();
                if (sp->total_frames == 0) return;
                hipStream_t st = pl->part_stream[k & 1];
                float *dst = h_rows + pl->row_off[pl->part_first[k]] * d.D;
                if (pin_out) HIP_TRY(hipMemcpyAsync(dst, sp->h_rows.p, (size_t)sp->total_frames * d.D * 4, hipMemcpyDeviceToHost, st));
                else {
                    HIP_TRY(hipStreamSynchronize(st));
                    HIP_TRY(hipMemcpy(dst, sp->h_rows.p, (size_t)sp->total_frames * d.D * 4, hipMemcpyDeviceToHost));
                }
            };
            for (int k = 0; k < np; k++) {
                ctu_plan *sp = pl->parts[k].get();
                hipStream_t st = pl->part_stream[k & 1];
                if (e->do_vad) {  // a range is a plan of its own: it starts its utterances where the caller's plan does
                    const int rc = ctu_plan_set_vad_ring(sp, pl->ring_hidx.empty() ? nullptr : pl->ring_hidx.data() + pl->part_first[k]);
                    if (rc != CTU_OK) return rc;
                }
                if (sp->total_frames) {
                    if (sp->h_pcm.n < (size_t)sp->total_samples) sp->h_pcm.alloc((size_t)sp->total_samples);
                    if (sp->h_rows.n < (size_t)sp->total_frames * d.D) sp->h_rows.alloc((size_t)sp->total_frames * d.D);
                    if (e->do_vad && sp->h_vad.n < (size_t)sp->total_frames) sp->h_vad.alloc((size_t)sp->total_frames);
                    const int16_t *src = h_pcm + (pl->sample_off[pl->part_first[k]] - PCM_HEAD);
                    if (pin_in) HIP_TRY(hipMemcpyAsync(sp->h_pcm.p, src, (size_t)sp->total_samples * 2, hipMemcpyHostToDevice, st));
                    else HIP_TRY(hipMemcpy(sp->h_pcm.p, src, (size_t)sp->total_samples * 2, hipMemcpyHostToDevice));
                    const int rc = ctu_engine_run(e, sp, sp->h_pcm.p, sp->h_rows.p, sp->h_vad.p, st);
                    if (rc != CTU_OK) {  // earlier ranges are still in flight on the two streams: drain them before the caller reuses its buffers
                        (void)hipStreamSynchronize(pl->part_stream[0]);
                        (void)hipStreamSynchronize(pl->part_stream[1]);
                        for (auto &pe : part_events) {
                            (void)hipEventDestroy(pe.first);
                            (void)hipEventDestroy(pe.second);
                        }
                        return rc;
                    }
                    // the engine's event pair is re-recorded by every range: keep the sum (ctu_engine_last_kernel_ms after a host run)
                    part_events.emplace_back();
                    HIP_TRY(hipEventCreate(&part_events.back().first));
                    HIP_TRY(hipEventCreate(&part_events.back().second));
                    std::swap(part_events.back().first, e->ev0);
                    std::swap(part_events.back().second, e->ev1);
                }
                if (k > 0) download(k - 1);  // behind the launch of part k: the copy engines and the kernels overlap
            }
            download(np - 1);
            HIP_TRY(hipStreamSynchronize(pl->part_stream[0]));
            HIP_TRY(hipStreamSynchronize(pl->part_stream[1]));
            e->host_kernel_ms = 0.f;
            for (auto &pe : part_events) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, pe.first, pe.second) == hipSuccess) e->host_kernel_ms += ms;
                (void)hipEventDestroy(pe.first);
                (void)hipEventDestroy(pe.second);
            }
            e->host_timed = true;
            if (e->do_vad)
                for (int k = 0; k < np; k++)
                    if (pl->parts[k]->total_frames)
                        HIP_TRY(hipMemcpy(v.data() + pl->row_off[pl->part_first[k]], pl->parts[k]->h_vad.p, (size_t)pl->parts[k]->total_frames, hipMemcpyDeviceToHost));
        } else {
            if (pl->h_pcm.n < (size_t)pl->total_samples) pl->h_pcm.alloc((size_t)pl->total_samples);
            if (pl->h_rows.n < (size_t)pl->total_frames * d.D) pl->h_rows.alloc((size_t)pl->total_frames * d.D);
            if (e->do_vad && pl->h_vad.n < (size_t)pl->total_frames) pl->h_vad.alloc((size_t)pl->total_frames);
            hipStream_t s = nullptr;
            if (pin_in) HIP_TRY(hipMemcpyAsync(pl->h_pcm.p, h_pcm, (size_t)pl->total_samples * 2, hipMemcpyHostToDevice, s));
            else HIP_TRY(hipMemcpy(pl->h_pcm.p, h_pcm, (size_t)pl->total_samples * 2, hipMemcpyHostToDevice));
            int rc = ctu_engine_run(e, pl, pl->h_pcm.p, pl->h_rows.p, pl->h_vad.p, s);
            if (rc != CTU_OK) return rc;
            if (pin_out) {
                HIP_TRY(hipMemcpyAsync(h_rows, pl->h_rows.p, (size_t)pl->total_frames * d.D * 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
            } else {
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(h_rows, pl->h_rows.p, (size_t)pl->total_frames * d.D * 4, hipMemcpyDeviceToHost));
            }
            if (e->do_vad) HIP_TRY(hipMemcpy(v.data(), pl->h_vad.p, v.size(), hipMemcpyDeviceToHost));
        }
        if (e->do_vad) {
            if (h_vad) std::memcpy(h_vad, v.data(), v.size());
            if (d.o.vad_apply_mode == "drop") {
                // rows of non-speech frames are dropped (src/io/batch.cc:237-238): compact each utterance in place
                for (int i = 0; i < pl->n_utt; i++) {
                    const int64_t r0 = pl->row_off[i], T = pl->frames[i];
                    int64_t keep = 0;
                    for (int64_t t = 0; t < T; t++)
                        if (v[r0 + t] == '1') {
                            if (keep != t) std::memmove(h_rows + (r0 + keep) * d.D, h_rows + (r0 + t) * d.D, (size_t)d.D * 4);
                            keep++;
                        }
                    if (rows_per_utt) rows_per_utt[i] = keep;
                }
            } else if (rows_per_utt) {
                // a file the majority filter never got ready on (no more frames than its delay) writes no row (ctu_engine_run)
                const int64_t delay = (d.o.vad_filter_order - 1) / 2;
                for (int i = 0; i < pl->n_utt; i++)
                    if (pl->frames[i] <= delay) rows_per_utt[i] = 0;
            }
        }
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

static void cmvn_maps(ctu_engine *e) {
    if (!e->col_of_slot.empty()) return;
    const ctu::Design &d = *e->design;
    const int fc = d.o.fea_ncepcoefs + 1, X = fc * (d.post_order + 1);
    e->col_of_slot.assign(X, 0);
    e->slot_of_col.assign(d.D, -1);
    for (int k = 0; k < X; k++) {
        const int i = (k + 1) % X;               // internal vector entry held by slot k (post_impl.cc:56-62)
        const int j = i / fc, ii = i % fc;       // block, entry within the block (0 = c0)
        const int col = fc * j + (ii == 0 ? fc - 1 : ii - 1);  // writer order: c1..cN, c0 (out.cc:188-201)
        e->col_of_slot[k] = col;
        e->slot_of_col[col] = k;
    }
    e->d_col_of_slot.upload(e->col_of_slot);
    e->d_slot_of_col.upload(e->slot_of_col);
}

static int cmvn_check(ctu_engine *e, const ctu_plan *pl, const int32_t *spk_of_utt, int32_t n_spk) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    const ctu::Design &d = *e->design;
    if (!(d.o.stat_cmvn || d.o.apply_cmvn)) {
        set_error(e, "ENGINE: engine was not created with -stat_cmvn / -apply_cmvn");
        return CTU_ERR_INPUT;
    }
    if (n_spk < 1 || (pl->n_utt && !spk_of_utt)) {
        set_error(e, "ENGINE: speaker table missing");
        return CTU_ERR_INPUT;
    }
    for (int i = 0; i < pl->n_utt; i++)
        if (spk_of_utt[i] < 0 || spk_of_utt[i] >= n_spk) {
            set_error(e, "ENGINE: speaker index out of range");
            return CTU_ERR_INPUT;
        }
    return CTU_OK;
}

int ctu_cmvn_cols(const ctu_engine *e) {
    if (!e) return -1;
    const ctu::Design &d = *e->design;
    return (d.o.fea_ncepcoefs + 1) * (d.post_order + 1);
}

int ctu_cmvn_accumulate(ctu_engine *e, const ctu_plan *pl, const float *d_rows, const int32_t *spk_of_utt, int32_t n_spk,
                        const double *mean, double *acc, void *stream) {
    int rc = cmvn_check(e, pl, spk_of_utt, n_spk);
    if (rc != CTU_OK) return rc;
    if (!acc || (pl->total_frames && !d_rows)) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    hipStream_t s = (hipStream_t)stream;
    try {
        HIP_TRY(hipSetDevice(e->device));
        cmvn_maps(e);
        const int cols = ctu_cmvn_cols(e);
        std::vector<int> spk(spk_of_utt, spk_of_utt + pl->n_utt);
        e->d_spk.upload(spk);
        std::vector<double> zero((size_t)n_spk * (cols + 1), 0.0);
        e->d_stat_a.upload(zero);
        if (mean) e->d_stat_b.upload(std::vector<double>(mean, mean + (size_t)n_spk * cols));
        hipLaunchKernelGGL(cmvn_accumulate_kernel, dim3(pl->n_trap_chunks), dim3(256), 0, s, d_rows, pl->utt_info.p,
                           pl->trap_chunks.p, e->d_spk.p, e->d_col_of_slot.p, mean ? e->d_stat_b.p : nullptr, e->d_stat_a.p,
                           cols, e->design->D);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(hipMemcpy(zero.data(), e->d_stat_a.p, zero.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < zero.size(); i++) acc[i] += zero[i];
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

int ctu_cmvn_apply(ctu_engine *e, const ctu_plan *pl, float *d_rows, const int32_t *spk_of_utt, int32_t n_spk,
                   const double *mean, const double *var, void *stream) {
    int rc = cmvn_check(e, pl, spk_of_utt, n_spk);
    if (rc != CTU_OK) return rc;
    if (!mean || !var || (pl->total_frames && !d_rows)) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    hipStream_t s = (hipStream_t)stream;
    try {
        HIP_TRY(hipSetDevice(e->device));
        cmvn_maps(e);
        const int cols = ctu_cmvn_cols(e);
        std::vector<int> spk(spk_of_utt, spk_of_utt + pl->n_utt);
        e->d_spk.upload(spk);
        e->d_stat_a.upload(std::vector<double>(mean, mean + (size_t)n_spk * cols));
        e->d_stat_b.upload(std::vector<double>(var, var + (size_t)n_spk * cols));
        hipLaunchKernelGGL(cmvn_apply_kernel, dim3(pl->n_trap_chunks), dim3(256), 0, s, d_rows, pl->utt_info.p,
                           pl->trap_chunks.p, e->d_spk.p, e->d_slot_of_col.p, e->d_stat_a.p, e->d_stat_b.p, cols, e->design->D);
        HIP_TRY(hipGetLastError());
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

const int64_t *ctu_plan_out_samples(const ctu_plan *p) { return p->out_samples.empty() ? nullptr : p->out_samples.data(); }

int ctu_engine_run_signal(ctu_engine *e, const ctu_plan *pl, const int16_t *d_pcm, int16_t *d_out, void *stream) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    const ctu::Design &d = *e->design;
    if (!d.signal_out) {
        set_error(e, "ENGINE: ctu_engine_run_signal needs -format_out raw|wave");
        return CTU_ERR_INPUT;
    }
    if (pl->n_utt == 0) return CTU_OK;
    if (!d_pcm || !d_out) {
        set_error(e, "ENGINE: null device buffer");
        return CTU_ERR_INPUT;
    }
    hipStream_t s = (hipStream_t)stream;
    e->in_signal_call = true;
    const int rc = ctu_engine_run(e, pl, d_pcm, nullptr, nullptr, stream);  // spectra before / after NR into the plan's scratch
    e->in_signal_call = false;
    if (rc != CTU_OK) return rc;
    try {
        HIP_TRY(hipSetDevice(e->device));
        SynthParams sp;
        sp.K = d.K; sp.wfft = d.wfft; sp.window = d.window; sp.wshift = d.wshift;
        sp.inv_n = 1.0f / (float)d.wfft;
        sp.corr = d.ola_corr;
        if (pl->total_frames > 0 && e->big) {
            // 1024 .. 4096 points: a workgroup per frame, LDS: two buffers of wfft / 2 (+ 4) complex values and the twiddles
            const size_t shm = ((size_t)(d.wfft / 2) * 3 + 8) * sizeof(float2);
            const int g = (int)std::min<int64_t>(pl->total_frames, (int64_t)e->n_cu * 8);
#define BIGSYNTH(NIT_)                                                                                                              \
    do {                                                                                                                            \
        if (shm > 64 * 1024) allow_big_lds(e, &bigsynth_kernel<NIT_>);                                                              \
        hipLaunchKernelGGL((bigsynth_kernel<NIT_>), dim3(g), dim3(256), shm, s, pl->xri.p, pl->pnr.p, pl->ybuf.p, (long long)pl->total_frames, \
                           d.wfft, d.window, sp.inv_n, e->big_tw.p);                                                                \
    } while (0)
            if (d.wfft == 1024) BIGSYNTH(4);
            else if (d.wfft == 2048) BIGSYNTH(8);
            else BIGSYNTH(16);
#undef BIGSYNTH
            HIP_TRY(hipGetLastError());
        }
        else if (pl->total_frames > 0 && !e->sy) {
            const int g = (int)std::min<int64_t>((pl->total_frames + 7) / 8, (int64_t)e->n_cu * 8);
            hipLaunchKernelGGL(synth_kernel, dim3(g), dim3(256), 0, s, pl->xri.p, pl->pnr.p, pl->ybuf.p, (long long)pl->total_frames, sp);
            HIP_TRY(hipGetLastError());
        }
        int64_t longest = 0;
        for (int64_t n : pl->out_samples) longest = std::max(longest, n);
        const int gx = (int)std::max<int64_t>(1, std::min<int64_t>((longest + 255) / 256, 64));
        hipLaunchKernelGGL(ola_kernel, dim3(gx, pl->n_utt), dim3(256), 0, s, pl->ybuf.p, d_out, pl->utt_info.p, pl->d_sample_off.p,
                           pl->n_utt, sp);
        HIP_TRY(hipGetLastError());
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

int ctu_engine_run_signal_host(ctu_engine *e, const ctu_plan *pl, const int16_t *h_pcm, int16_t *h_out) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    try {
        HIP_TRY(hipSetDevice(e->device));
        DevBuf<int16_t> pcm, out;
        pcm.alloc((size_t)pl->total_samples);
        out.alloc((size_t)pl->total_samples);
        HIP_TRY(hipMemcpy(pcm.p, h_pcm, (size_t)pl->total_samples * 2, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(out.p, 0, (size_t)pl->total_samples * 2));
        const int rc = ctu_engine_run_signal(e, pl, pcm.p, out.p, nullptr);
        if (rc != CTU_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h_out, out.p, (size_t)pl->total_samples * 2, hipMemcpyDeviceToHost));
        return CTU_OK;
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
}

int ctu_cmvn_accumulate_host(ctu_engine *e, const ctu_plan *pl, const float *h_rows, const int32_t *spk_of_utt, int32_t n_spk,
                             const double *mean, double *acc) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        DevBuf<float> rows;
        rows.alloc((size_t)pl->total_frames * e->design->D);
        HIP_TRY(hipMemcpy(rows.p, h_rows, rows.n * sizeof(float), hipMemcpyHostToDevice));
        return ctu_cmvn_accumulate(e, pl, rows.p, spk_of_utt, n_spk, mean, acc, nullptr);
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
}

int ctu_cmvn_apply_host(ctu_engine *e, const ctu_plan *pl, float *h_rows, const int32_t *spk_of_utt, int32_t n_spk,
                        const double *mean, const double *var) {
    if (!e || !pl || pl->eng != e) return CTU_ERR_INPUT;
    if (pl->total_frames == 0) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        DevBuf<float> rows;
        rows.alloc((size_t)pl->total_frames * e->design->D);
        HIP_TRY(hipMemcpy(rows.p, h_rows, rows.n * sizeof(float), hipMemcpyHostToDevice));
        const int rc = ctu_cmvn_apply(e, pl, rows.p, spk_of_utt, n_spk, mean, var, nullptr);
        if (rc != CTU_OK) return rc;
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h_rows, rows.p, rows.n * sizeof(float), hipMemcpyDeviceToHost));
        return CTU_OK;
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
}

int ctu_decode_g711(ctu_engine *e, const uint8_t *d_codes, int64_t n, int alaw, int16_t *d_pcm, void *stream) {
    if (!e || n < 0 || (n && (!d_codes || !d_pcm))) return CTU_ERR_INPUT;
    if (n == 0) return CTU_OK;
    try {
        HIP_TRY(hipSetDevice(e->device));
        const int grid = (int)std::min<int64_t>((n / 8 + 255) / 256, (int64_t)e->n_cu * 8);
        hipLaunchKernelGGL(g711_kernel, dim3(std::max(grid, 1)), dim3(256), 0, (hipStream_t)stream, d_codes, d_pcm, n, alaw);
        HIP_TRY(hipGetLastError());
    } catch (const std::exception &ex) {
        set_error(e, std::string("ENGINE: ") + ex.what());
        return CTU_ERR_DEVICE;
    }
    return CTU_OK;
}

int ctu_engine_reset_chain(ctu_engine *e) {
    if (!e) return CTU_ERR_INPUT;
    e->ss_stale.clear();
    e->vad_pos = 0;
    return CTU_OK;
}

int ctu_engine_set_vad_stream(ctu_engine *e, const unsigned char *bytes, int64_t n) {
    if (!e || n < 0 || (n && !bytes)) return CTU_ERR_INPUT;
    if (!e->ss_file) {
        set_error(e, "ENGINE: this configuration does not take its VAD decisions from a stream (-nr_mode hwss|fwss|2fwss -vad file=...)");
        return CTU_ERR_INPUT;
    }
    e->vad_stream.assign(bytes, bytes + n);
    e->vad_pos = 0;
    return CTU_OK;
}

const char *ctu_engine_kernel_name(const ctu_engine *e) {
    if (!e) return "";
    if (e->kname.empty()) {
        const ctu::Design &d = *e->design;
        const ctu::Opts &o = d.o;
        std::string n;
        if (e->wave1k) n = "wave1k_kernel";
        else if (e->big) n = "bigfft_kernel<" + std::to_string(d.wfft / 256) + ">";
        else {
            const char *feat = e->feat == FEAT_DCTC ? "DCTC" : e->feat == FEAT_BANDS ? "BANDS" : e->feat == FEAT_LP ? "LP" : "LPD";
            const bool exten = o.nr_mode == "exten" && !o.nr_when_afterFB;
            const bool plain = plain_cepstral(d) || (!o.fea_E && o.fb_power && o.remove_dc && !o.remove_dc1 && !o.nr_when_afterFB && !d.signal_out);
            n = "frontend_kernel<" + std::to_string(o.remove_dc1 ? 16 : e->nz) + ", " + feat + ", MODE " + std::to_string(e->mode) + ", " +
                (e->sy ? "full" : !plain ? "full" : (e->ss && (!fused_frame_shape(d) || (!(e->md && e->feat == FEAT_DCTC) && !(e->feat == FEAT_BANDS && !o.fb_inld)))) ? "full" :  // launch_vx's *ss branch
                 exten ? "exten" : o.fb_inld ? "inld" : "plain") + (e->md ? ", MD" : "") + (e->vf ? ", VF" : "") +
                (e->ss ? ", SS" : "") + (e->sy ? ", SY" : "") + ">";
        }
        const_cast<ctu_engine *>(e)->kname = n;
    }
    return e->kname.c_str();
}

float ctu_engine_last_kernel_ms(ctu_engine *e) {
    if (!e || !e->timed) return -1.f;
    if (e->host_timed) return e->host_kernel_ms;
    float ms = -1.f;
    if (hipEventSynchronize(e->ev1) != hipSuccess) return -1.f;
    if (hipEventElapsedTime(&ms, e->ev0, e->ev1) != hipSuccess) return -1.f;
    return ms;
}

}  // extern "C"
