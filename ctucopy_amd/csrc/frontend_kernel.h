// frontend_kernel: PCM -> window -> FFT -> |.|^2 -> (exten NR) -> filter bank -> log -> DCT / LPC -> rows.
// Included by engine.hip (one translation unit: the kernels and their host launchers share types).
#pragma once

namespace {

// NZ = number of 32-sample rows that can hold non-zero input (ceil(window/32)); rows >= NZ are
// literal zeros so the compiler prunes the first butterflies.
#ifndef CTU_LB
#define CTU_LB 4        // waves per SIMD the register allocator must leave room for (2 workgroups x 8 waves / 4 SIMDs)
#endif
#ifndef CTU_STAMP
#define CTU_STAMP 0     // diagnostic build: per-wave s_memtime sums per code segment (never in production)
#endif
#if CTU_STAMP
#define STAMP(i)                                                                                   \
    do {                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        unsigned long long now_;                                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        st_acc[i] += now_ - st_prev;                                                               \
        st_prev = now_;                                                                            \
    } while (0)
#else
#define STAMP(i) do { } while (0)
#endif
#ifndef CTU_DUAL
#define CTU_DUAL 1      // the two passes of the 512-point mode side by side in the headline instantiation (see DUAL below)
#endif
#ifndef CTU_PK
#define CTU_PK 1        // 1: DUAL with the two passes in the halves of packed registers (v_pk_*_f32 for all of phase 1's arithmetic); 0: scalar lock step (round 3)
#endif
#ifndef CTU_ABL
#define CTU_ABL 0       // diagnostic builds only (wrong results): bit 0 no lane-0 selects in the untangle, 1 no mean removal, 2 no imaginary
#endif                  // transpose, 3 no mirror fetch, 4 no P stores of the mirror half, 5 no inter-stage twiddles, 6 no phase 2, 7 no second DFT: what a unit of VALU / LDS work costs
#ifndef CTU_CAP_WAVES
#define CTU_CAP_WAVES 1  // 0: the front end's instantiations may run more waves per SIMD than the launch is laid out for (A/B)
#endif
#ifndef CTU_VF8
#define CTU_VF8 1  // 0: the fused VAD criterion of the 256-point mode as two lattices of 16 lanes x 13 samples (A/B)
#endif
#ifndef CTU_BURG_UNROLL2
#define CTU_BURG_UNROLL2 0  // 1: the two lattices of a 16-lane group unrolled into one block (the scheduler may interleave them)
#endif
#ifndef CTU_EXTEN_F64
#define CTU_EXTEN_F64 0 // 1: exten state (Navg, Yavg) and its update in double.  Measured (tools/probes/sweep_err.py, exten_err.py): no accuracy gain - the residual is fp32 FFT noise amplified where a bin is almost fully suppressed - and -30 % throughput
#endif

// MODE 0: 512-point real FFT, one frame per 16-lane group, NZ = rows of 32 samples, two passes of 4 frames.
// MODE 1: 256-point real FFT, TWO frames per 16-lane group packed as re/im of the same 256-point complex FFT
//         (no twiddles in the untangle), NZ = rows of 16 samples, one pass of 8 frames.
// VX:     also export what the VAD kernels need (kept out of the default instantiation: it costs registers).
// NC:     coefficients accumulated per frame in phase 2 (16 or MAXC): a compile-time width keeps eight accumulators
//         and a code path out of the common instantiation.
// GEN:    GEN_PLAIN = the plain chain (DC removal on, power spectrum, no -fea_E, no exten, no intensity-loudness law,
//         no diagnostics): the option flags below become constants, which frees 30 SGPRs and the last spills.
//         GEN_INLD / GEN_EXTEN = the plain chain plus exactly that option (PLP; C4's noise reduction); GEN_DC1 = GEN_FULL
//         with -remove_dc1; GEN_FULL reads
//         every flag at run time.
// LPO:    LP order = number of cepstra when it is fixed at compile time (12: the PLP preset), 0 = run-time orders up to
//         MAX_LP.  The unrolled Levinson / a->c tail then has no guards and no dead orders.
// MD:     the DCT-II tail of phase 2 on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 FMA chains): the band
//         logarithms of a slot are the B operand as they stand (lane = frame + 8 h + 16 kk holds band group kk + 4 h),
//         the folded DCT table is the A operand; FEAT_DCTC with NC = 16, and FEAT_LP with NC = 16 (compressed bands,
//         the cosine iDFT table as A: the lags of the LP analysis are the same kind of contraction over the bands).
//
// Work distribution.  A tile is <= 64 consecutive frames of one utterance.  Stateless chains: the eight waves of a
// workgroup share each tile (wave w takes frame slots 8w..8w+7).  Per-wave chains (p.per_wave; exten): every wave
// walks its own list of tiles, eight frames at a time, and keeps the state that runs along an utterance in its
// registers.  Either way a wave only touches its own eight P rows: no workgroup barrier after the table load.
// VF:     Burg-cepstral VAD criterion fused in (vad_fused.h; 25 ms frames in the 256- and the 512-point mode): the step's eight
//         time-domain frames are rebuilt from the spectra after NR with the original phases and their cepstra stored for
//         vad_lanes_kernel (the detector's recurrences, four lanes per utterance).
// SS:     spectral subtraction with the Burg cepstral detector (hwss / fwss / 2fwss, src/nr/nr.cc:181-442; both modes,
//         25 ms frames): the detector sees the frames rebuilt from the (expanded) spectra, so a step runs phase 1 twice - once to
//         feed the detector, whose transposes and frames use up the P rows, once more for the subtraction itself.
// SY:     speech-enhancement output (row N3, sigOUT src/io/out.cc:405-434): the step's spectra after NR go back to the time
//         domain in registers (the inverse of vad_fused.h with the synthesis conventions) and the frames are written for
//         the overlap-add kernel; no spectra through HBM, no phase 2.
// The fused detector paths of the 512-point mode (VF / SS with MODE 0) keep both passes' transform outputs and a 25-sample
// lattice per lane alive: 256 VGPRs, one workgroup per CU (their staging area takes the LDS of the second one anyway).
#ifndef CTU_SY_LB
#define CTU_SY_LB 2  // register budget of the synthesis instantiations (SY): 256 VGPRs, one workgroup per CU.  Measured (profiles/r03_ab_sy_register_budget.txt): at 128 VGPRs they spill 200 bytes per lane; 2 -> -25 % kernel time, 3 -> -21 %
#endif
#ifndef CTU_VF1_LB
#define CTU_VF1_LB CTU_LB  // experiment: register budget of the fused-detector instantiations of the 256-point mode
#endif
constexpr int fe_waves_per_simd(int mode, bool vf, bool ss, bool sy = false) {
    return ((vf || ss) && mode == 0) ? 2 : (sy ? CTU_SY_LB : ((vf || ss) ? CTU_VF1_LB : CTU_LB));
}

template <int NZ, int FEAT, int MODE, bool VX, int NC, int GEN, int LPO = 0, bool MD = false, bool VF = false, bool SS = false, bool SY = false>
#if CTU_CAP_WAVES
// also the MOST waves per SIMD the register file is laid out for: a workgroup is eight waves and the grid is at most two workgroups per
// CU (engine.hip: max_wg), i.e. four per SIMD when they are dealt evenly.  An instantiation that needs 96 registers or fewer would be
// allowed five or six, and the dispatcher then fills SIMDs unevenly with the same sixteen waves (measured: the exten chains, which last
// as long as their slowest wave, +22 .. +40 % when the kernel dropped from 97 to 96 registers)
__attribute__((amdgpu_waves_per_eu(fe_waves_per_simd(MODE, VF, SS, SY), fe_waves_per_simd(MODE, VF, SS, SY))))
#endif
__global__ __launch_bounds__(WG, fe_waves_per_simd(MODE, VF, SS, SY)) void frontend_kernel(const KParams p) {
    constexpr bool FULL = GEN == GEN_FULL || GEN == GEN_DC1;  // GEN_DC1 = GEN_FULL plus -remove_dc1 (its offsets cost registers the others need)
    static_assert(!MD || ((FEAT == FEAT_DCTC || FEAT == FEAT_LP) && NC == 16), "MD: DCT / cosine-iDFT tail with 16 coefficient rows");
    static_assert(!VF || !VX, "VF: no spectrum export");
    static_assert(!SS || (!VX && !VF && (GEN == GEN_PLAIN || GEN == GEN_FULL)), "SS: the plain chain, or run-time flags (signal output, energy columns, -fb_inld, LP kinds, magnitude spectra)");
    static_assert(!((VF || SS) && MODE == 0) || NZ == 13, "VF / SS in the 512-point mode: 400-sample windows (16 lanes x 25 samples)");
    static_assert(!SY || (!VX && !VF && (GEN == GEN_FULL || GEN == GEN_DC1)), "SY: run-time flags, no export");
    const int o_e_mode = FULL ? p.e_mode : 0, o_dbg = FULL ? p.dbg : 0;
    const bool o_fb_inld = FULL ? p.fb_inld != 0 : GEN == GEN_INLD, o_nr_exten = FULL ? p.nr_exten != 0 : GEN == GEN_EXTEN;
    const bool o_fb_power = FULL ? p.fb_power != 0 : true, o_remove_dc = FULL ? p.remove_dc != 0 : true;
    const bool o_skip_phase2 = SY ? true : (FULL ? p.skip_phase2 != 0 : false);
    constexpr bool o_dc1 = GEN == GEN_DC1;  // -remove_dc1 (decode_kernels.h): an instantiation of its own (generic row count, no export)
    const bool per_wave = (GEN == GEN_EXTEN || FULL || VF || SS) ? p.per_wave != 0 : false;
    extern __shared__ __align__(16) float lds[];
    float *Pt = lds;                       // [TILE][PSTRIDE]
    float *ltab = lds + TILE * PSTRIDE;    // phase-2 tables (layout: KParams)
    float *ltw = ltab + p.tab_floats;      // [16][LTW_STRIDE] per-lane constant records

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15;   // n2 in stage 1, k1 in stage 2
    const int fg = lane >> 4;    // frame slot within the wave pass
    const int partner = ((lane & 48) | ((16 - l16) & 15)) << 2;  // byte address for ds_bpermute
    for (int i = tid; i < p.tab_floats; i += WG) ltab[i] = p.ftab[i];
    for (int i = tid; i < 16 * LANEC; i += WG) ltw[(i / LANEC) * LTW_STRIDE + (i % LANEC)] = p.lanec[i];
    // Phase 2 reads whole 4-bin chunks, so bins a frame never writes (row padding 257..259; everything above bin
    // 128 in the 256-point mode) are read under zero weights: start the tile finite.  Only finite values (spectra,
    // transpose scratch) are ever written afterwards.
    for (int i = tid; i < TILE * PSTRIDE; i += WG) Pt[i] = 0.f;
    __syncthreads();
    const float4 *lc = reinterpret_cast<const float4 *>(ltw + l16 * LTW_STRIDE);  // this lane's constant record
    const float4 *ltw4 = lc + (LC_TW >> 2);                                       // [0,8) stage twiddles, [8,12) untangle
    float *const Pw = Pt + wave * 8 * PSTRIDE;  // this wave's eight P rows
    // transpose scratch: rows 4-7 of the wave (4 x 260 = 16 x 65 dwords).  Pass A's spectra go to rows 0-3; pass B
    // (and the single pass of the 256-point mode) overwrites the scratch with its own spectra after its transposes.
    float *const scratch = Pw + 4 * PSTRIDE;

#if CTU_STAMP
    unsigned long long st_acc[16] = {0}, st_prev;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");
#endif
#ifdef CTU_PRIO  // experiment: static priority for the later-dispatched half of the workgroup (waves 4-7)
    if (wave >= 4) __builtin_amdgcn_s_setprio(CTU_PRIO);
#endif
#ifdef CTU_STAGGER  // experiment: half of the waves start late, so that partners on a SIMD sit in different phases of a step
    if (CTU_STAGGER_SEL) {
        for (int i = 0; i < CTU_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
    }
#endif
    int tile = as_const(p.wg_first)[per_wave ? blockIdx.x * NWAVE + wave : blockIdx.x];
    if (tile < 0) return;
    TileRec rec = load_rec(p.tiles, tile);

    // exten NR state (src/nr/nr.cc:86-93): lane = bin (bin = lane + 64 j), carried along the wave's utterance
    constexpr int NJ = MODE == 1 ? 3 : 5;  // ceil(K / 64): K = 129 / 257
#if CTU_EXTEN_F64
    typedef double xstate_t;
#else
    typedef float xstate_t;
#endif
    xstate_t navg[NJ], yavg[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) {
        navg[j] = (xstate_t)0.95;
        yavg[j] = (xstate_t)0.05;
    }

    // -remove_dc1: ov[0] = o_t, ov[j] = o_{t-j} of one frame; what position i of the frame has had subtracted so far
    constexpr int DCJ = 8;
    auto dc1_load = [&](float (&ov)[DCJ + 1], int64_t frame_row, int t) {
#pragma unroll
        for (int j = 0; j <= DCJ; j++) ov[j] = (j <= p.dc1_J && t - j >= 0) ? p.dc1[frame_row - j] : 0.f;
    };
    auto dc1_cum = [&](const float (&ov)[DCJ + 1], int i, bool with_own) {
        float c = with_own ? ov[0] : 0.f;
#pragma unroll
        for (int j = 1; j <= DCJ; j++) c += (j <= p.dc1_J && i <= p.window - 1 - j * p.wshift) ? ov[j] : 0.f;
        return c;
    };

    CepDetRun sdet;            // SS: the cepstral detector's recurrences along the wave's utterance
    float snavg[NJ], snrav[NJ];  // SS: noise estimate(s), lane = bin
#pragma unroll
    for (int j = 0; j < NJ; j++) snavg[j] = snrav[j] = 0.f;
    if constexpr (SS) cepdet_reset(sdet);

    while (true) {
        const int nvalid = rec.nvalid;
        const int64_t rbase = rec.rbase;
        const int next = rec.next;
        TileRec nrec = rec;
        if (next >= 0) nrec = load_rec(p.tiles, next);
        // SS, second and later passes of the seed iteration (engine.hip): an utterance whose seed did not change keeps its rows
        const bool skip_tile = SS && p.ss_dirty && !reinterpret_cast<const unsigned char *>(p.ss_dirty)[as_const(p.tile_utt)[tile]];
        const int nsub = skip_tile ? 0 : (per_wave ? (nvalid + 7) >> 3 : 1);
        for (int sub = 0; sub < nsub; sub++) {
        // this step's frame slots are [slot0, slot0 + 8) of the tile; their spectra live in the wave's P rows 0..7
        const int slot0 = per_wave ? sub * 8 : wave * 8;
        const int nv = min(max(nvalid - slot0, 0), 8);
        float2 vz[16];   // VF / SS / SY: the forward transform's output Z[l16 + 16 r], kept for the inverse
        float2 vz1[16];  // SY, 512-point mode: the same of the second pass

        // ================= phase 1: frames -> power spectrum rows =================
        // `which`: bit 0 = first pass (slots 0-3; the only pass of the 256-point mode), bit 1 = second pass (slots 4-7)
        auto phase1 = [&](int which) {
        // DUAL: the two passes of the 512-point mode run side by side (slots 0-3 and 4-7 of the step in lock step): each
        // table read serves both, and every stage offers the scheduler two independent instruction streams - the kernel
        // is bound by dependent latency (LDS round trips, transcendental-free but long FMA chains) at four waves per SIMD,
        // not by issue.  Every instantiation of the plain chain and of the plain chain + intensity-loudness law (round 4; rounds 2-3: the
        // headline instantiation only): phase 1 does not depend on the feature tail.
        constexpr bool DUAL = CTU_DUAL && MODE == 0 && (GEN == GEN_PLAIN || GEN == GEN_INLD) && !VX && NZ < 16 && !VF && !SS && !SY;
        if constexpr (DUAL && CTU_PK) {
            // The two passes as the two halves of packed registers: every add / multiply / FMA of phase 1 is a v_pk_*_f32 that
            // serves slots 0-3 and 4-7 together (kernel_common.h: cx2).  Same statements as the scalar DUAL block below.
            if (nv > 0) {
                cx2 v[16];
                const int f0 = slot0 + fg, f1 = f0 + 4;
                const int c0 = f0 < nvalid ? f0 : nvalid - 1, c1 = f1 < nvalid ? f1 : nvalid - 1;  // duplicates are never stored
                const bool st0 = (l16 == 0) && (rec.t0 + c0 == 0), st1 = (l16 == 0) && (rec.t0 + c1 == 0);
                {
                    const int16_t *x0p = p.pcm + rec.sbase + (int64_t)c0 * p.wshift + 2 * l16 - 2;
                    const int16_t *x1p = p.pcm + rec.sbase + (int64_t)c1 * p.wshift + 2 * l16 - 2;
                    pcm4 q0[NZ], q1[NZ];
#pragma unroll
                    for (int j = 0; j < NZ; j++) q0[j] = *reinterpret_cast<const pcm4 *>(x0p + 32 * j);
#pragma unroll
                    for (int j = 0; j < NZ; j++) q1[j] = *reinterpret_cast<const pcm4 *>(x1p + 32 * j);
                    v2f dcp[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};  // four partial sums: a packed result cannot feed the next instruction
#pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const float4 w4 = lc[(LC_WIN + 2 * j) >> 2];  // two rows of window pairs per float4
                        const float w0 = (j & 1) ? w4.z : w4.x, w1 = (j & 1) ? w4.w : w4.y;
                        v2f xm = {(float)(int16_t)(q0[j].lo >> 16), (float)(int16_t)(q1[j].lo >> 16)};
                        const v2f x0 = {(float)(int16_t)(q0[j].hi & 0xffffu), (float)(int16_t)(q1[j].hi & 0xffffu)};
                        const v2f x1 = {(float)(int16_t)(q0[j].hi >> 16), (float)(int16_t)(q1[j].hi >> 16)};
                        if (j == 0) {  // first sample of the file: history is 0
                            xm.x = st0 ? 0.f : xm.x;
                            xm.y = st1 ? 0.f : xm.y;
                        }
                        const v2f y0 = (x0 - xm * p.preem) * w0, y1 = (x1 - x0 * p.preem) * w1;
                        v[j] = cx2{y0, y1};
                        dcp[j & 1] += y0;
                        dcp[2 + (j & 1)] += y1;
                    }
                    const v2f dc = (dcp[0] + dcp[1]) + (dcp[2] + dcp[3]);
#pragma unroll
                    for (int j = NZ; j < 16; j++) v[j] = cx2{v2f{0.f, 0.f}, v2f{0.f, 0.f}};
                    // mean of the windowed frame over `window` samples (src/io/in.cc:375-382); rows < NZ-1 are fully inside
                    const v2f m = v2f{row16_allreduce_add(dc.x), row16_allreduce_add(dc.y)} * p.inv_window;
                    const float4 mk = lc[(LC_MASK + 2 * (NZ - 1)) >> 2];
                    const float mx = ((NZ - 1) & 1) ? mk.z : mk.x, my = ((NZ - 1) & 1) ? mk.w : mk.y;
#pragma unroll
                    for (int j = 0; j < NZ - 1; j++) {
                        v[j].x -= m;
                        v[j].y -= m;
                    }
                    v[NZ - 1].x -= m * mx;
                    v[NZ - 1].y -= m * my;
                }
                dft16(v);
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    const float4 tw = ltw4[h];  // (k1 = 2h+1, k1 = 2h+2)
                    v[2 * h + 1] = cmul_s(v[2 * h + 1], tw.x, tw.y);
                    if (2 * h + 2 < 16) v[2 * h + 2] = cmul_s(v[2 * h + 2], tw.z, tw.w);
                }
                {   // both transposes at once through the wave's eight rows (layout: CTU_ADDTID16_P)
                    const uint32_t sa = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(lvoid_t *)Pw);
                    const uint32_t rd = (uint32_t)(size_t)(lvoid_t *)(Pw + 129 * l16 + 16 * fg);
#define CTU_E_XA(i) v[i].x.x
#define CTU_E_XB(i) v[i].x.y
#define CTU_E_YA(i) v[i].y.x
#define CTU_E_YB(i) v[i].y.y
                    __builtin_amdgcn_wave_barrier();
                    CTU_ADDTID16_P(CTU_E_XA, sa, 0);
                    CTU_ADDTID16_P(CTU_E_XB, sa, 256);
                    v2f re[16], im[16];
                    CTU_READ2_PAIRS16(re, rd);
                    CTU_ADDTID16_P(CTU_E_YA, sa, 0);
                    CTU_ADDTID16_P(CTU_E_YB, sa, 256);
                    CTU_READ2_PAIRS16(im, rd);
#pragma unroll
                    for (int n2 = 0; n2 < 16; n2++) v[n2] = cx2{re[n2], im[n2]};
                    __builtin_amdgcn_wave_barrier();
#undef CTU_E_XA
#undef CTU_E_XB
#undef CTU_E_YA
#undef CTU_E_YB
                }
                dft16(v);
                float *pr0 = Pw + fg * PSTRIDE, *pr1 = Pw + (4 + fg) * PSTRIDE;
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    const float4 u4q = ltw4[8 + (k2 >> 1)];
                    const float wr = (k2 & 1) ? u4q.z : u4q.x, wi = (k2 & 1) ? u4q.w : u4q.y;
                    const int k = l16 + 16 * k2;
                    v2f br = {mirror_fetch(v[15 - k2].x.x, partner), mirror_fetch(v[15 - k2].x.y, partner)};
                    v2f bi = {mirror_fetch(v[15 - k2].y.x, partner), mirror_fetch(v[15 - k2].y.y, partner)};
                    if (l16 == 0) {
                        br = v[(16 - k2) & 15].x;
                        bi = v[(16 - k2) & 15].y;
                    }
                    const v2f ar = v[k2].x, ai = v[k2].y;
                    const v2f sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
                    const v2f tr = di * wr + dr * wi;
                    const v2f ti = di * wi - dr * wr;
                    const v2f ur = sr + tr, ui = si + ti, vr = sr - tr, vi = si - ti;
                    // the untangle's 1/4 is in the window (the engine scales this instantiation's table by 1/2: ctu_engine::half_window)
                    const v2f pk = ur * ur + ui * ui, pm = vr * vr + vi * vi;
                    pr0[k] = pk.x;
                    pr1[k] = pk.y;
                    pr0[256 - k] = pm.x;
                    pr1[256 - k] = pm.y;
                }
                if (l16 == 0) {  // bin 128 is its own mirror (no 1/4 there: the halved window is undone); bin 0 floor (src/io/in.cc:390)
                    const v2f a = v[8].x * 2.f, b = v[8].y * 2.f;
                    const v2f p128 = a * a + b * b;
                    pr0[128] = p128.x;
                    pr1[128] = p128.y;
                    pr0[0] = pr1[0] = 1e-10f;
                }
            }
        } else
        if constexpr (DUAL) {
            if (nv > 0) {
                float2 v0[16], v1[16];
                const int f0 = slot0 + fg, f1 = f0 + 4;
                const int c0 = f0 < nvalid ? f0 : nvalid - 1, c1 = f1 < nvalid ? f1 : nvalid - 1;  // duplicates are never stored
                const bool st0 = (l16 == 0) && (rec.t0 + c0 == 0), st1 = (l16 == 0) && (rec.t0 + c1 == 0);
                {
                    const int16_t *x0p = p.pcm + rec.sbase + (int64_t)c0 * p.wshift + 2 * l16 - 2;
                    const int16_t *x1p = p.pcm + rec.sbase + (int64_t)c1 * p.wshift + 2 * l16 - 2;
                    pcm4 q0[NZ], q1[NZ];
#pragma unroll
                    for (int j = 0; j < NZ; j++) q0[j] = *reinterpret_cast<const pcm4 *>(x0p + 32 * j);
#pragma unroll
                    for (int j = 0; j < NZ; j++) q1[j] = *reinterpret_cast<const pcm4 *>(x1p + 32 * j);
                    float dc0 = 0.f, dc1 = 0.f;
#pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const float4 w4 = lc[(LC_WIN + 2 * j) >> 2];  // two rows of window pairs per float4
                        const float w0 = (j & 1) ? w4.z : w4.x, w1 = (j & 1) ? w4.w : w4.y;
                        float am = (float)(int16_t)(q0[j].lo >> 16), bm = (float)(int16_t)(q1[j].lo >> 16);
                        const float a0 = (float)(int16_t)(q0[j].hi & 0xffffu), a1 = (float)(int16_t)(q0[j].hi >> 16);
                        const float b0 = (float)(int16_t)(q1[j].hi & 0xffffu), b1 = (float)(int16_t)(q1[j].hi >> 16);
                        if (j == 0) {  // first sample of the file: history is 0
                            am = st0 ? 0.f : am;
                            bm = st1 ? 0.f : bm;
                        }
                        const float ya0 = w0 * (a0 - p.preem * am), ya1 = w1 * (a1 - p.preem * a0);
                        const float yb0 = w0 * (b0 - p.preem * bm), yb1 = w1 * (b1 - p.preem * b0);
                        v0[j] = make_float2(ya0, ya1);
                        v1[j] = make_float2(yb0, yb1);
                        dc0 += ya0 + ya1;
                        dc1 += yb0 + yb1;
                    }
#pragma unroll
                    for (int j = NZ; j < 16; j++) v0[j] = v1[j] = make_float2(0.f, 0.f);
                    // mean of the windowed frame over `window` samples (src/io/in.cc:375-382); rows < NZ-1 are fully inside
                    const float m0 = row16_allreduce_add(dc0) * p.inv_window, m1 = row16_allreduce_add(dc1) * p.inv_window;
                    const float4 mk = lc[(LC_MASK + 2 * (NZ - 1)) >> 2];
                    const float mx = ((NZ - 1) & 1) ? mk.z : mk.x, my = ((NZ - 1) & 1) ? mk.w : mk.y;
#pragma unroll
                    for (int j = 0; j < ((CTU_ABL & 2) ? 1 : NZ - 1); j++) {
                        v0[j].x -= m0;
                        v0[j].y -= m0;
                        v1[j].x -= m1;
                        v1[j].y -= m1;
                    }
                    v0[NZ - 1].x -= m0 * mx;
                    v0[NZ - 1].y -= m0 * my;
                    v1[NZ - 1].x -= m1 * mx;
                    v1[NZ - 1].y -= m1 * my;
                }
                dft16(v0);
                dft16(v1);
#pragma unroll
                for (int h = 0; h < ((CTU_ABL & 32) ? 1 : 8); h++) {
                    const float4 tw = ltw4[h];  // (k1 = 2h+1, k1 = 2h+2)
                    v0[2 * h + 1] = cmul(v0[2 * h + 1], make_float2(tw.x, tw.y));
                    v1[2 * h + 1] = cmul(v1[2 * h + 1], make_float2(tw.x, tw.y));
                    if (2 * h + 2 < 16) {
                        v0[2 * h + 2] = cmul(v0[2 * h + 2], make_float2(tw.z, tw.w));
                        v1[2 * h + 2] = cmul(v1[2 * h + 2], make_float2(tw.z, tw.w));
                    }
                }
#ifdef CTU_PAD_SALU  // diagnostic builds: what N more scalar / vector / LDS instructions per step cost (results unchanged)
                {
                    uint32_t pad_ = 0;
                    asm volatile(".rept %c1\n\ts_add_u32 %0, %0, 1\n\t.endr" : "+s"(pad_) : "n"(CTU_PAD_SALU) : "scc");
                }
#endif
#ifdef CTU_PAD_VALU
                {
                    float pad_ = 0.f;
                    asm volatile(".rept %c1\n\tv_add_f32 %0, %0, %0\n\t.endr" : "+v"(pad_) : "n"(CTU_PAD_VALU));
                }
#endif
#ifdef CTU_PAD_LDS
                {
                    f32x4 pad_;
                    asm volatile(".rept %c2\n\tds_read_b128 %0, %1\n\t.endr\n\ts_waitcnt lgkmcnt(0)" : "=&v"(pad_) : "v"((uint32_t)(size_t)(lvoid_t *)(ltw + 4 * lane)), "n"(CTU_PAD_LDS) : "memory");
                }
#endif
                float *pr0 = Pw + fg * PSTRIDE, *pr1 = Pw + (4 + fg) * PSTRIDE;
                auto untangle = [&](const float2 (&v)[16], float *prow, int k2, float wr, float wi) {
                    const int k = l16 + 16 * k2;
                    float br = (CTU_ABL & 8) ? v[15 - k2].x : mirror_fetch(v[15 - k2].x, partner);
                    float bi = (CTU_ABL & 8) ? v[15 - k2].y : mirror_fetch(v[15 - k2].y, partner);
                    if (!(CTU_ABL & 1) && l16 == 0) {
                        br = v[(16 - k2) & 15].x;
                        bi = v[(16 - k2) & 15].y;
                    }
                    const float ar = v[k2].x, ai = v[k2].y;
                    const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
                    const float tr = wr * di + wi * dr;
                    const float ti = wi * di - wr * dr;
                    const float ur = sr + tr, ui = si + ti, vr = sr - tr, vi = si - ti;
                    // the untangle's 1/4 is in the window (the engine scales this instantiation's table by 1/2: ctu_engine::half_window)
                    if (CTU_ABL & 16) prow[k] = (ur * ur + ui * ui) + (vr * vr + vi * vi);
                    else {
                        prow[k] = ur * ur + ui * ui;
                        prow[256 - k] = vr * vr + vi * vi;
                    }
                };
                // both transposes at once: slots 0-3 through the wave's rows 0-3, slots 4-7 through rows 4-7 (their own
                // spectra land there afterwards)
                wave_transpose16_dual(v0, v1, (uint32_t)(size_t)(lvoid_t *)Pw, (uint32_t)(size_t)(lvoid_t *)scratch,
                                      Pw + 65 * l16 + 16 * fg, scratch + 65 * l16 + 16 * fg);
                if (!(CTU_ABL & 128)) {
                    dft16(v0);
                    dft16(v1);
                }
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    const float4 u4q = ltw4[8 + (k2 >> 1)];
                    const float wr = (k2 & 1) ? u4q.z : u4q.x, wi = (k2 & 1) ? u4q.w : u4q.y;
                    untangle(v0, pr0, k2, wr, wi);
                    untangle(v1, pr1, k2, wr, wi);
                }
                if (l16 == 0) {  // bin 128 is its own mirror (no 1/4 there: the halved window is undone); bin 0 floor (src/io/in.cc:390)
                    const float a0 = 2.f * v0[8].x, b0 = 2.f * v0[8].y, a1 = 2.f * v1[8].x, b1 = 2.f * v1[8].y;
                    pr0[128] = a0 * a0 + b0 * b0;
                    pr1[128] = a1 * a1 + b1 * b1;
                    pr0[0] = pr1[0] = 1e-10f;
                }
            }
        } else
        if (o_dbg != 2 && nv > 0) {
            const int npass = (MODE == 0 && nv > 4) ? 2 : 1;
            // the pass body is instantiated twice (it = 0, 1) so that row numbers are compile-time constants
            auto pass = [&](auto IT) {
                constexpr int it = decltype(IT)::value;
                const int f = slot0 + it * 4 + fg;  // frame slot in the tile (MODE 0)
                const int fc = f < nvalid ? f : nvalid - 1;  // clamp: duplicates are computed but never stored
                const bool file_start = (l16 == 0) && (rec.t0 + fc == 0);
                STAMP(0);  // loop overhead / previous tail
                float2 v[16];
                if constexpr (MODE == 1) {
                    // frames A = slot 2*fg, B = A+1 of this step's 8; sample n = 16 j + l16 of each goes to re / im
                    const int fa = slot0 + 2 * fg, fb_ = fa + 1;
                    const int ca = fa < nvalid ? fa : nvalid - 1, cb_ = fb_ < nvalid ? fb_ : nvalid - 1;
                    const int16_t *xa = p.pcm + rec.sbase + (int64_t)ca * p.wshift + l16;
                    const int16_t *xb = p.pcm + rec.sbase + (int64_t)cb_ * p.wshift + l16;
                    const bool start_a = (l16 == 0) && (rec.t0 + ca == 0), start_b = (l16 == 0) && (rec.t0 + cb_ == 0);
                    float dca = 0.f, dcb = 0.f;
                    float ova[DCJ + 1], ovb[DCJ + 1];
                    if (o_dc1) {
                        dc1_load(ova, rbase + ca, rec.t0 + ca);
                        dc1_load(ovb, rbase + cb_, rec.t0 + cb_);
                    }
    #pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const float4 w4 = lc[(LC_WIN + 2 * j) >> 2];
                        const float w = (j & 1) ? w4.z : w4.x;  // 0 beyond the window
                        // x[i-1] and x[i] in one 2-byte-aligned dword load (global memory takes unaligned dwords)
                        const uint32_t wa = reinterpret_cast<const pcm2 *>(xa + 16 * j - 1)->v, wb = reinterpret_cast<const pcm2 *>(xb + 16 * j - 1)->v;
                        float pa = (float)(int16_t)(wa & 0xffffu), pb = (float)(int16_t)(wb & 0xffffu);
                        float a0 = (float)(int16_t)(wa >> 16), b0 = (float)(int16_t)(wb >> 16);
                        if (o_dc1) {
                            const int i = 16 * j + l16;
                            pa -= dc1_cum(ova, i - 1, i >= 1);
                            pb -= dc1_cum(ovb, i - 1, i >= 1);
                            a0 -= dc1_cum(ova, i, true);
                            b0 -= dc1_cum(ovb, i, true);
                        }
                        if (j == 0) {
                            pa = start_a ? 0.f : pa;
                            pb = start_b ? 0.f : pb;
                        }
                        const float ya = w * (a0 - p.preem * pa), yb = w * (b0 - p.preem * pb);
                        v[j] = make_float2(ya, yb);
                        dca += ya;
                        dcb += yb;
                    }
    #pragma unroll
                    for (int j = NZ; j < 16; j++) v[j] = make_float2(0.f, 0.f);
                    STAMP(1);
                    if (o_remove_dc) {
                        const float ma = row16_allreduce_add(dca) * p.inv_window, mb = row16_allreduce_add(dcb) * p.inv_window;
    #pragma unroll
                        for (int j = 0; j < NZ; j++) {
                            const float4 mk = lc[(LC_MASK + 2 * j) >> 2];
                            const float mm = (j & 1) ? mk.z : mk.x;
                            v[j].x -= ma * mm;
                            v[j].y -= mb * mm;
                        }
                    }
                } else {
                    float dc = 0.f;
                    float ov0[DCJ + 1];
                    if (o_dc1) dc1_load(ov0, rbase + fc, rec.t0 + fc);
                    pcm4 q[NZ];
                    // samples x[i0-2 .. i0+1] of row j of this lane's frame: i0 = 32 j + 2 l16
                    const int16_t *x = p.pcm + rec.sbase + (int64_t)fc * p.wshift + 2 * l16 - 2;
    #pragma unroll
                    for (int j = 0; j < NZ; j++) q[j] = *reinterpret_cast<const pcm4 *>(x + 32 * j);
    #pragma unroll
                    for (int j = 0; j < NZ; j++) {
                        const float4 w4 = lc[(LC_WIN + 2 * j) >> 2];  // two rows of window pairs per float4
                        const float w0 = (j & 1) ? w4.z : w4.x, w1 = (j & 1) ? w4.w : w4.y;
                        float xm = (float)(int16_t)(q[j].lo >> 16);
                        float x0 = (float)(int16_t)(q[j].hi & 0xffffu);
                        float x1 = (float)(int16_t)(q[j].hi >> 16);
                        if (o_dc1) {
                            const int i = 32 * j + 2 * l16;
                            xm -= dc1_cum(ov0, i - 1, i >= 1);
                            x0 -= dc1_cum(ov0, i, true);
                            x1 -= dc1_cum(ov0, i + 1, true);
                        }
                        if (j == 0) xm = file_start ? 0.f : xm;  // first sample of the file: history is 0
                        const float y0 = w0 * (x0 - p.preem * xm);
                        const float y1 = w1 * (x1 - p.preem * x0);  // w is 0 beyond the window
                        v[j] = make_float2(y0, y1);
                        dc += y0 + y1;
                    }
    #pragma unroll
                    for (int j = NZ; j < 16; j++) v[j] = make_float2(0.f, 0.f);
                    STAMP(1);  // PCM + window loads, convert, pre-emphasis, window
                    if (o_remove_dc) {
                        // mean of the windowed frame over `window` samples (src/io/in.cc:375-382)
                        const float m = row16_allreduce_add(dc) * p.inv_window;
                        // generic instantiation: any window <= 512, per-sample masks.  The run-time-flag *ss instantiations exist for 13 rows
                        // only and serve every window of at most 13 rows: masks as well (rows beyond the window would take the mean otherwise)
                        if (NZ == 16 || (SS && FULL)) {
    #pragma unroll
                            for (int j = 0; j < NZ; j++) {
                                const float4 mk = lc[(LC_MASK + 2 * j) >> 2];
                                v[j].x -= m * ((j & 1) ? mk.z : mk.x);
                                v[j].y -= m * ((j & 1) ? mk.w : mk.y);
                            }
                        } else {  // exact instantiation: rows < NZ-1 are fully inside the window
                            const float4 mk = lc[(LC_MASK + 2 * (NZ - 1)) >> 2];
    #pragma unroll
                            for (int j = 0; j < NZ - 1; j++) {
                                v[j].x -= m;
                                v[j].y -= m;
                            }
                            v[NZ - 1].x -= m * (((NZ - 1) & 1) ? mk.z : mk.x);
                            v[NZ - 1].y -= m * (((NZ - 1) & 1) ? mk.w : mk.y);
                        }
                    }
                }
                STAMP(2);  // DC removal
                // ---- stage 1: DFT16 over n1 (registers), lane = n2; then twiddle W256^(n2*k1)
                dft16(v);
                __builtin_amdgcn_sched_barrier(0);  // twiddles: fetch them just in time, not across the DFT
    #pragma unroll
                for (int h = 0; h < 8; h++) {
                    const float4 tw = ltw4[h];  // (k1 = 2h+1, k1 = 2h+2)
                    v[2 * h + 1] = cmul(v[2 * h + 1], make_float2(tw.x, tw.y));
                    if (2 * h + 2 < 16) v[2 * h + 2] = cmul(v[2 * h + 2], make_float2(tw.z, tw.w));
                }
                __builtin_amdgcn_sched_barrier(0);
                STAMP(3);  // DFT16 #1 + twiddles
                // ---- transpose [k1][n2] -> lane k1 holds all n2, through the LDS scratch, re then im
                __builtin_amdgcn_wave_barrier();
                wave_transpose16(v, (uint32_t)(size_t)(lvoid_t *)scratch, scratch + 65 * l16 + 16 * fg);
                STAMP(4);  // LDS transpose
                // ---- stage 2: DFT16 over n2, lane = k1: v[k2] = Z[k1 + 16 k2]
                dft16(v);
                STAMP(5);  // DFT16 #2
                if constexpr (VF || SS || SY) {
    #pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (it == 0) vz[r] = v[r];
                        else vz1[r] = v[r];
                    }
                }

                if constexpr (MODE == 1) {
                    // two real frames in one complex FFT: XA[k] = (Z[k] + conj Z[256-k])/2, XB[k] = (Z[k] - conj Z[256-k])/2i;
                    // bins 0..128 of both; the mirror bin comes from lane (16-k1)%16 as in MODE 0
                    const int fa = slot0 + 2 * fg;
                    float *pa = Pw + (2 * fg) * PSTRIDE, *pb = pa + PSTRIDE;
    #pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) {
                        if ((k2 & 3) == 0) __builtin_amdgcn_sched_barrier(0);
                        float br = mirror_fetch(v[15 - k2].x, partner);
                        float bi = mirror_fetch(v[15 - k2].y, partner);
                        if (l16 == 0) {
                            br = v[(16 - k2) & 15].x;
                            bi = v[(16 - k2) & 15].y;
                        }
                        const float ar = v[k2].x, ai = v[k2].y;
                        const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
                        const int k = l16 + 16 * k2;
                        pa[k] = 0.25f * (sr * sr + si * si);
                        pb[k] = 0.25f * (dr * dr + di * di);
                        if (VX && p.vad_export == 1) {  // XA = s/2, XB = (d)/(2i) = (di - i dr)/2
                            if (fa < nvalid) p.xri[(rbase + fa) * 129 + k] = make_float2(0.5f * sr, 0.5f * si);
                            if (fa + 1 < nvalid) p.xri[(rbase + fa + 1) * 129 + k] = make_float2(0.5f * di, -0.5f * dr);
                        }
                    }
                    if (l16 == 0) {
                        pa[128] = v[8].x * v[8].x;
                        pb[128] = v[8].y * v[8].y;
                        if constexpr (SS && SY) {  // X_A[128] = Re Z[128], X_B[128] = Im Z[128]: their signs (out.cc:419, see the SS block)
                            pa[129] = v[8].x;
                            pb[129] = v[8].y;
                        }
                        if (o_remove_dc) pa[0] = pb[0] = 1e-10f;
                        if (VX && p.vad_export == 1) {
                            if (fa < nvalid) p.xri[(rbase + fa) * 129 + 128] = make_float2(v[8].x, 0.f);
                            if (fa + 1 < nvalid) p.xri[(rbase + fa + 1) * 129 + 128] = make_float2(v[8].y, 0.f);
                        }
                    }
                } else {
                    // ---- untangle the packed real FFT and take |.|^2.  Lane k1 handles its bins k2=0..7,
                    //      each together with its mirror bin 256-k held by lane (16-k1)%16 in register 15-k2
                    //      (register (16-k2)%16 for k1 = 0).
                    float *prow = Pw + (it * 4 + fg) * PSTRIDE;
    #pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) {
                        if ((k2 & 3) == 0) __builtin_amdgcn_sched_barrier(0);  // two batches: bounds the registers in flight
                        const float4 u4q = ltw4[8 + (k2 >> 1)];
                        float br = mirror_fetch(v[15 - k2].x, partner);
                        float bi = mirror_fetch(v[15 - k2].y, partner);
                        if (l16 == 0) {
                            br = v[(16 - k2) & 15].x;
                            bi = v[(16 - k2) & 15].y;
                        }
                        const float wr = (k2 & 1) ? u4q.z : u4q.x, wi = (k2 & 1) ? u4q.w : u4q.y;
                        const float ar = v[k2].x, ai = v[k2].y;
                        const float sr = ar + br, si = ai - bi, dr = ar - br, di = ai + bi;
                        const float tr = wr * di + wi * dr;
                        const float ti = wi * di - wr * dr;
                        const float ur = sr + tr, ui = si + ti, vr = sr - tr, vi = si - ti;
                        const float pk = 0.25f * (ur * ur + ui * ui);
                        const float pm = 0.25f * (vr * vr + vi * vi);
                        const int k = l16 + 16 * k2;
                        prow[k] = pk;
                        prow[256 - k] = pm;
                        if constexpr (SS && SY) {  // X[256] = conj(v) / 2 at k = 0: its real part's sign (out.cc:419, see the SS block)
                            if (k2 == 0 && l16 == 0) prow[257] = vr;
                        }
                        if (VX && p.vad_export == 1 && f < nvalid) {  // X[k] = u/2, X[256-k] = conj(v)/2
                            float2 *xo = p.xri + (rbase + f) * 257;
                            xo[k] = make_float2(0.5f * ur, 0.5f * ui);
                            xo[256 - k] = make_float2(0.5f * vr, -0.5f * vi);
                        }
                    }
                    if (l16 == 0) {  // bin 128 is its own mirror: X[128] = conj(Z[128]); bin 0 floor (src/io/in.cc:390)
                        prow[128] = v[8].x * v[8].x + v[8].y * v[8].y;
                        if (o_remove_dc) prow[0] = 1e-10f;
                        if (VX && p.vad_export == 1 && f < nvalid) p.xri[(rbase + f) * 257 + 128] = make_float2(v[8].x, -v[8].y);
                    }
                }
                STAMP(6);  // untangle + P writes
            };
            if (which & 1) pass(std::integral_constant<int, 0>{});
            if (npass == 2 && (which & 2)) pass(std::integral_constant<int, 1>{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        };
        constexpr bool SY_HALVES = SY && MODE == 0 && !SS;  // pass, NR and synthesis per half step: one transform output live at a time
                                                            // (with the *ss modes the detector needs the whole step first: both stay)
        if constexpr (!SY_HALVES) phase1(3);
        // The step's eight time-domain frames rebuilt from the P rows (magnitudes) and vz (directions), then the Burg
        // cepstra of each (vad_fused.h).  Uses up the wave's P rows.  out[x]: lane 16 fg + i holds coefficient i of frame
        // slot 2 fg + x.  HANN: the detector of the *ss modes windows the frame first (src/vdet/CepstralDet.h:131-147).
        auto rebuild_cepstra = [&](auto NCO, auto HANN, auto (&out)[2]) {
            constexpr int nco = decltype(NCO)::value;
            typedef std::remove_reference_t<decltype(out[0])> real_t;  // float, or double for the *ss detector (vad_fused.h)
            // the VAD module's criterion (no Hann window, float): the lattice only - {alpha, k_m} go to the scratch rows and
            // vad_a2c_kernel finishes the cepstra one frame per lane; the *ss detector needs its cepstra here and now
            constexpr bool rc_only = CTU_VF_A2C && !decltype(HANN)::value && sizeof(real_t) == 4;
            constexpr bool VF8 = CTU_VF8 && MODE == 1 && !decltype(HANN)::value && sizeof(real_t) == 4;
            // the *ss detector of the run-time-flag instantiations takes the window at run time too (any window the lanes' 13 / 25 samples
            // cover: up to 208 / 400): the lane and register of the window's last sample are looked up per order instead of being named
            constexpr bool RTW = decltype(HANN)::value && GEN == GEN_FULL;
            // 1 / window: the float lattice's copy is wave-uniform and stays in an SGPR
            auto inv_w_of = [&](auto z) {
                if constexpr (sizeof(z) == 4) return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)p.inv_window_d)));
                else return p.inv_window_d;
            };
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (MODE == 1) {
            float2 vn[16];
            vf_scale_spectra(vz, vn, Pw + (2 * fg) * PSTRIDE, Pw + (2 * fg + 1) * PSTRIDE, l16, partner);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();  // every lane has its gains before the scratch (P rows 4-7) is reused
            vf_inverse_fft(vn, ltw4, (uint32_t)(size_t)(lvoid_t *)scratch, scratch + 65 * l16 + 16 * fg);
            STAMP(11);  // VF / SS: spectra re-scaled, inverse transform
            // time-domain frames into the wave's LDS rows (the spectra are spent): frame slot s at s * VF_FSTRIDE
            float *ta = Pw + (2 * fg) * VF_FSTRIDE + l16, *tb = ta + VF_FSTRIDE;
#pragma unroll
            for (int m = 0; m < VF_SPL; m++) {
                ta[16 * m] = vn[m].x;
                tb[16 * m] = vn[m].y;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (VF8) {
                // the VAD module's criterion: the 200-sample window as 8 lanes x 25 samples, the step's eight frames in ONE lattice
                // instead of two of four frames each (16 lanes x 13): the per-order sums, the division and the edge fix-ups are paid once
                static_assert(VF_WINDOW == 8 * VF8_SPL, "");
                const int f8v = lane >> 3, l8v = lane & 7;
                const float *tx = Pw + f8v * VF_FSTRIDE + VF8_SPL * l8v;  // banks: 216 f + 25 l + j are distinct over a half wave
                float x[VF8_SPL];
                real_t cc[nco];
#pragma unroll
                for (int j = 0; j < VF8_SPL; j++) x[j] = tx[j];
                static_assert(rc_only && nco <= 16, "");
                vf_burg_cepstrum<nco, VF8_SPL - 1, real_t, VF8_SPL, true, 8, true>(x, l8v, 7, VF8_SPL - 1, inv_w_of(real_t{}), cc);
                out[0] = cc[0];  // lane 8 f + i: entries i and i + 8 of frame slot f
                out[1] = cc[1];
            } else
#if CTU_BURG_UNROLL2
#pragma unroll
#endif
            for (int xb = 0; xb < 2; xb++) {  // frame A, then frame B of this 16-lane group
                const float *tx = Pw + (2 * fg + xb) * VF_FSTRIDE + VF_SPL * l16;
                float x[VF_SPL];
                real_t cc[nco];
#pragma unroll
                for (int j = 0; j < VF_SPL; j++) x[j] = (VF_SPL * l16 + j < p.window) ? tx[j] : 0.f;  // the first `window` samples (src/vad/vad.cc:233)
                if constexpr (decltype(HANN)::value) {
                    const float *hw = ltab + p.han_off + VF_SPL * l16;
#pragma unroll
                    for (int j = 0; j < VF_SPL; j++) x[j] *= hw[j];
                }
                vf_burg_cepstrum<nco, RTW ? -1 : VF_JW, real_t, VF_SPL, rc_only, 16, rc_only>(x, l16, RTW ? (p.window - 1) / VF_SPL : VF_LW, RTW ? (p.window - 1) % VF_SPL : VF_JW, inv_w_of(real_t{}), cc, (decltype(HANN)::value && nco != 12) ? p.ss_nc : nco);  // the 12-coefficient instantiations: straight-line code, no order is skipped
                real_t mine = cc[0];
                if constexpr (!rc_only) {
#pragma unroll
                    for (int m = 1; m < nco; m++) mine = l16 == m ? cc[m] : mine;
                }
                out[xb] = mine;
            }
            } else {
            // 512-point mode: one frame per 16-lane group, slots 0-3 (first pass) then 4-7.  Each half re-scales its bins to the
            // magnitudes in its own P rows, runs the packed-real inverse through those rows (as the synthesis of row N3 does)
            // and stages its four frames in the wave's area behind the tables; out[h]: lane 16 fg + i holds coefficient i of
            // frame slot 4 h + fg.
            float *const stage = ltw + LTW_FLOATS + wave * VF0_STAGE;
            auto half_step = [&](const float2 (&vzz)[16], auto HALF) {
                constexpr int half = decltype(HALF)::value;
                float2 vn[16];
                float *rows4 = Pw + 4 * half * PSTRIDE;
                vf_scale_tangle0<false>(vzz, vn, rows4 + fg * PSTRIDE, ltw4, l16, partner, 2.f);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // every lane has its gains before the rows become transpose scratch
                vf_inverse_fft(vn, ltw4, (uint32_t)(size_t)(lvoid_t *)rows4, rows4 + 65 * l16 + 16 * fg);
                STAMP(11);
                // z[n] = x[2n] + i x[2n+1], n = l16 + 16 m: the first 13 rows of 32 samples cover the window
                float2 *ta = reinterpret_cast<float2 *>(stage + fg * VF0_FSTRIDE) + l16;
#pragma unroll
                for (int m = 0; m < 13; m++) ta[16 * m] = vn[m];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const float *tx = stage + fg * VF0_FSTRIDE + VF0_SPL * l16;
                float x[VF0_SPL];
                real_t cc[nco];
#pragma unroll
                for (int j = 0; j < VF0_SPL; j++) x[j] = (VF0_SPL * l16 + j < p.window) ? tx[j] : 0.f;  // the first `window` samples (src/vad/vad.cc:233)
                if constexpr (decltype(HANN)::value) {
                    const float *hw = ltab + p.han_off + VF0_SPL * l16;
#pragma unroll
                    for (int j = 0; j < VF0_SPL; j++) x[j] *= hw[j];
                }
                vf_burg_cepstrum<nco, RTW ? -1 : VF0_JW, real_t, VF0_SPL, rc_only, 16, rc_only>(x, l16, RTW ? (p.window - 1) / VF0_SPL : VF0_LW, RTW ? (p.window - 1) % VF0_SPL : VF0_JW, inv_w_of(real_t{}), cc, (decltype(HANN)::value && nco != 12) ? p.ss_nc : nco);  // the 12-coefficient instantiations: straight-line code, no order is skipped
                real_t mine = cc[0];
                if constexpr (!rc_only) {
#pragma unroll
                    for (int m = 1; m < nco; m++) mine = l16 == m ? cc[m] : mine;
                }
                out[half] = mine;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();  // the staged frames are read before the next half overwrites them
            };
            half_step(vz, std::integral_constant<int, 0>{});
            out[1] = out[0];
            if (nv > 4) half_step(vz1, std::integral_constant<int, 1>{});
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            STAMP(12);  // VF / SS: frames re-laid, Burg lattice, cepstra
        };

        // magnitude instead of power (src/io/in.cc:415-417), rows [r0, r1); off the default path
        auto to_magnitude = [&](int r0, int r1) {
            if (!o_fb_power && r1 > r0) {
                for (int e = lane + r0 * p.K; e < r1 * p.K; e += 64) {
                    const int f = e / p.K, k = e - f * p.K;
                    float *q_ = Pw + f * PSTRIDE + k;
                    *q_ = sqrtf(*q_);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        };

        // ================= hwss / fwss / 2fwss (src/nr/nr.cc:181-442) =================
        if constexpr (SS) {
            if (nv > 0) {
                // signal output: the NR works on magnitudes (-fb_power is forced off, src/io/opts.cc:285-288); all eight rows - a duplicate
                // frame shares a complex transform with a real one and the synthesis reads both
                // ... and so they do on the feature path under -fb_power off (src/io/in.cc:415-417 ahead of the NR, batch.cc:205-213)
                if constexpr (SY) to_magnitude(0, (MODE == 1 || nv > 4) ? 8 : 4);
                else to_magnitude(0, nv);
                const bool two = p.ss_mode == 3;
                const int utt = as_const(p.tile_utt)[tile];
                if (rec.t0 == 0 && slot0 == 0) {  // new_file (nr.cc:212-221, 402-409): seed from the previous file's last vector
                    cepdet_reset(sdet);
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        const float sd = (lane + 64 * j < p.K) ? p.ss_seed[(int64_t)utt * p.K + lane + 64 * j] : 0.f;
                        snavg[j] = two ? sd : ss_pow(sd, p.nr_a);
                        snrav[j] = 0.f;
                    }
                }
                unsigned vbits = 0;
                if (p.ss_cached) {
                    // The detector sees the spectra as they come from the transform (step 1 below), never the subtracted ones: its
                    // decisions do not depend on the noise seed.  The first pass of the seed iteration (engine.hip) stored them;
                    // later passes subtract with the new seed on the spectra the step already holds.
                    for (int f = 0; f < nv; f++) vbits |= (unsigned)(p.ss_vbits[rbase + slot0 + f] & 1) << f;
                } else {
                // (1) what the detector sees: X^a (X itself in 2fwss) with the original phase (nr.cc:278-295)
                if (!two && p.nr_a != 1.0f) {
                    for (int f = 0; f < nv; f++) {
                        float *row = Pw + f * PSTRIDE + lane;
#pragma unroll
                        for (int j = 0; j < NJ; j++)
                            if (lane + 64 * j < p.K) row[64 * j] = ss_pow(row[64 * j], p.nr_a);
                    }
                }
                double mine_ab[2];
                // the lattice is unrolled for SS_NC = 16 coefficients; the plain-chain instantiations also exist unrolled for the presets' 12
                // (LPO = 12 marks them: the LP order it otherwise carries has no meaning for cepstra / band outputs) - fewer registers,
                // and orders that follow each other without a branch in between
                constexpr int SSN = (LPO == 12 && FEAT != FEAT_LP && FEAT != FEAT_LPD) ? 12 : SS_NC;
                rebuild_cepstra(std::integral_constant<int, SSN>{}, std::true_type{}, mine_ab);
                // (2) the detector's recurrences over the step's frames, in order (src/vdet/CepstralDet.h:140-194)
                for (int s_ = 0; s_ < nv; s_++) {
                    // frame slot s_: group s_ / 2, frame s_ % 2 of it (256-point mode); group s_ % 4 of half s_ / 4 (512-point mode)
                    const int src = ((16 * (MODE == 1 ? (s_ >> 1) : (s_ & 3)) + (lane & 15)) << 2);
                    const double sel = (MODE == 1 ? (s_ & 1) : (s_ >> 2)) ? mine_ab[1] : mine_ab[0];
                    const double got = __hiloint2double(__builtin_amdgcn_ds_bpermute(src, __double2hiint(sel)),
                                                        __builtin_amdgcn_ds_bpermute(src, __double2loint(sel)));
                    const double cil = lane < p.ss_nc ? got : 0.0;
                    vbits |= (unsigned)cepdet_frame(sdet, cil, lane, p.ss_nc, p.ss_init, p.nr_p_d, p.ss_q) << s_;
                }
                if (lane < nv) p.ss_vbits[rbase + slot0 + lane] = (unsigned char)((vbits >> lane) & 1u);
                // (3) the spectra again, then the subtraction proper, frames in order, lane = bin
                phase1(3);
                // ... and so they do on the feature path under -fb_power off (src/io/in.cc:415-417 ahead of the NR, batch.cc:205-213)
                if constexpr (SY) to_magnitude(0, (MODE == 1 || nv > 4) ? 8 : 4);
                else to_magnitude(0, nv);
                }
                const float pp = p.nr_p, qq = 1.0f - p.nr_p;
                for (int f = 0; f < nv; f++) {
                    const int t = rec.t0 + slot0 + f;
                    // hwss counts its initial segments down before the test, the others after it (nr.cc:225 vs :367, :440)
                    const bool upd = !((vbits >> f) & 1u) || t < (p.ss_mode == 1 ? p.ss_init - 1 : p.ss_init);
                    float *row = Pw + f * PSTRIDE + lane;
#pragma unroll
                    for (int j = 0; j < NJ; j++) {
                        if (lane + 64 * j < p.K) {
                            float X = row[64 * j];
                            if (two) {
                                if (upd) snavg[j] = pp * snavg[j] + qq * X;
                                X = fabsf(X - snavg[j]);
                                if (upd) snrav[j] = pp * snrav[j] + qq * X;
                                X = fabsf(X - snrav[j]);
                            } else {
                                X = ss_pow(X, p.nr_a);
                                if (upd) snavg[j] = pp * snavg[j] + qq * X;
                                X -= p.nr_b * snavg[j];
                                X = p.ss_mode == 1 ? fmaxf(X, 0.f) : fabsf(X);
                                X = ss_root(X, p.nr_a);
                            }
                            row[64 * j] = X;
                            if (t == rec.T - 1) {
                                // what the next file's noise estimate starts from (nr.cc:212-221).  With signal output sigOUT has flipped the
                                // Nyquist entry's sign by then if that bin's phase was pi (src/io/out.cc:419): phase 1 left the bin's signed
                                // real part behind the row's last bin
                                if (SY && lane + 64 * j == p.K - 1 && row[64 * j + 1] < 0.f) X = -X;
                                p.ss_last[(int64_t)utt * p.K + lane + 64 * j] = X;
                            }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }



        // ================= extended spectral subtraction (src/nr/nr.cc:86-140) =================
        // Sequential in t, independent across bins: lane = bin (bin = lane + 64 j), the step's frames in order, the
        // state (Navg, Yavg) in this wave's registers along its utterance (per-wave chains only).
        const bool o_after_fb = FULL ? p.nr_after_fb != 0 : false;  // -nr_when afterFB: the NR runs on the band energies (phase 2)
        auto apply_exten = [&](int f_lo, int f_hi) {
        if (o_nr_exten && !o_after_fb && f_hi > f_lo) {
            if (rec.t0 == 0 && slot0 == 0 && f_lo == 0) {  // new file: Navg = 0.95, Yavg = 0.05
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    navg[j] = (xstate_t)0.95;
                    yavg[j] = (xstate_t)0.05;
                }
            }
            const xstate_t pp = (xstate_t)p.nr_p_d, qq = (xstate_t)1.0 - pp;
            // the next frame's values are read while this frame's recurrence runs (the chain goes through Navg / Yavg in
            // registers, not through LDS); bins beyond K read the row's padding / the next row: finite, never stored
            float Xn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; j++) Xn[j] = Pw[f_lo * PSTRIDE + lane + 64 * j];
            for (int f = f_lo; f < f_hi; f++) {
                float *row = Pw + f * PSTRIDE + lane;
                float X[NJ];
#pragma unroll
                for (int j = 0; j < NJ; j++) X[j] = (lane + 64 * j < p.K) ? Xn[j] : 1.f;
                {
                    const float *nrow = Pw + (f + 1 < f_hi ? f + 1 : f) * PSTRIDE + lane;
#pragma unroll
                    for (int j = 0; j < NJ; j++) Xn[j] = nrow[64 * j];
                }
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    // H = Navg / (Navg^a + Yavg^a)^(1/a); the output X - H X is formed as X (1 - H) with 1 - H written
                    // without cancellation.  Double state: the reciprocal root / reciprocal start from the float
                    // instructions and take one Newton step in double.
#if CTU_EXTEN_F64
                    const double na = navg[j], ya = yavg[j], Xd = (double)X[j];
                    double H, omH;
                    if (p.nr_a == 1.0f) {
                        const double s_ = na + ya;
                        double ir = (double)__builtin_amdgcn_rcpf((float)s_);
                        ir = ir * (2.0 - s_ * ir);
                        ir = ir * (2.0 - s_ * ir);
                        H = na * ir;
                        omH = ya * ir;
                    } else if (p.nr_a == 2.0f) {
                        const double r2 = na * na + ya * ya;
                        double ir = (double)__builtin_amdgcn_rsqf((float)r2);
                        ir = ir * (1.5 - 0.5 * r2 * ir * ir);
                        ir = ir * (1.5 - 0.5 * r2 * ir * ir);
                        const double r = r2 * ir, dn = r * (r + na);
                        double id = (double)__builtin_amdgcn_rcpf((float)dn);
                        id = id * (2.0 - dn * id);
                        id = id * (2.0 - dn * id);
                        H = na * ir;
                        omH = (ya * ya) * id;
                    } else {
                        H = na / pow(pow(na, (double)p.nr_a) + pow(ya, (double)p.nr_a), 1.0 / (double)p.nr_a);
                        omH = 1.0 - H;
                    }
                    const double N = H * Xd;
                    navg[j] = pp * na + qq * N;
                    yavg[j] = fabs(Xd - navg[j]);
                    X[j] = (float)(Xd * omH);
#else
                    float H, omH;
                    if (p.nr_a == 1.0f) {
                        const float ir = __builtin_amdgcn_rcpf(navg[j] + yavg[j]);
                        H = navg[j] * ir;
                        omH = yavg[j] * ir;
                    } else if (p.nr_a == 2.0f) {
                        const float r2 = navg[j] * navg[j] + yavg[j] * yavg[j];
                        const float ir = __builtin_amdgcn_rsqf(r2);
                        const float r = r2 * ir;
                        H = navg[j] * ir;
                        omH = (yavg[j] * yavg[j]) * __builtin_amdgcn_rcpf(r * (r + navg[j]));
                    } else {
                        H = navg[j] / powf(powf(navg[j], p.nr_a) + powf(yavg[j], p.nr_a), 1.0f / p.nr_a);
                        omH = 1.0f - H;
                    }
                    const float N = H * X[j];
                    navg[j] = pp * navg[j] + qq * N;
                    yavg[j] = fabsf(X[j] - navg[j]);
                    X[j] = X[j] * omH;
#endif
                }
#pragma unroll
                for (int j = 0; j < NJ; j++)
                    if (lane + 64 * j < p.K) row[64 * j] = X[j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        };
        if constexpr (!SY_HALVES && !SS) {
            to_magnitude(0, SY ? 8 : nv);  // SY: all rows - a duplicate frame shares a complex transform with a real one
            apply_exten(0, nv);
        }
        // ================= speech synthesis (row N3): spectra after NR -> time-domain frames =================
        if constexpr (SY) {
            if (nv > 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if constexpr (MODE == 1) {
                    const uint32_t sb = (uint32_t)(size_t)(lvoid_t *)scratch;
                    const float *srd = scratch + 65 * l16 + 16 * fg;
                    float2 vn[16];
                    vf_scale_spectra<true>(vz, vn, Pw + (2 * fg) * PSTRIDE, Pw + (2 * fg + 1) * PSTRIDE, l16, partner, p.syn_scale);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();  // every lane has its magnitudes before the scratch (P rows 4-7) is reused
                    vf_inverse_fft(vn, ltw4, sb, srd);
                    const int fa = slot0 + 2 * fg;
                    float *ya = p.ybuf + (rbase + fa) * p.window + l16, *yb = ya + p.window;
#pragma unroll
                    for (int m = 0; m < 16; m++) {
                        if (l16 + 16 * m < p.window) {
                            if (fa < nvalid) ya[16 * m] = vn[m].x;
                            if (fa + 1 < nvalid) yb[16 * m] = vn[m].y;
                        }
                    }
                } else {
                    // each pass inverts through its own (spent) P rows: slots 0-3 in rows 0-3, slots 4-7 in rows 4-7
                    auto synth = [&](const float2 (&vzz)[16], auto HALF) {
                        constexpr int half = decltype(HALF)::value;
                        float2 vn[16];
                        float *rows4 = Pw + 4 * half * PSTRIDE;
                        vf_scale_tangle0(vzz, vn, rows4 + fg * PSTRIDE, ltw4, l16, partner, 2.f * p.syn_scale);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        vf_inverse_fft(vn, ltw4, (uint32_t)(size_t)(lvoid_t *)rows4, rows4 + 65 * l16 + 16 * fg);
                        // z[n] = x[2n] + i x[2n+1], n = l16 + 16 m: float2 stores (the window is even)
                        const int fr = slot0 + 4 * half + fg;
                        float2 *y = reinterpret_cast<float2 *>(p.ybuf + (rbase + fr) * p.window) + l16;
#pragma unroll
                        for (int m = 0; m < 16; m++)
                            if (2 * (l16 + 16 * m) < p.window && fr < nvalid) y[16 * m] = vn[m];
                    };
                    const int n0 = nv < 4 ? nv : 4;
                    if constexpr (!SS) {
                        phase1(1);
                        to_magnitude(0, 4);
                        apply_exten(0, n0);
                    }
                    synth(vz, std::integral_constant<int, 0>{});
                    if (nv > 4) {
                        if constexpr (!SS) {
                            phase1(2);
                            to_magnitude(4, 8);
                            apply_exten(4, nv);
                        }
                        synth(vz1, std::integral_constant<int, 1>{});
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (VX && p.vad_export && nv > 0) {  // the VAD looks at in->_Xsabs after NR (src/io/batch.cc:230-240, src/vad/vad.cc:96-107,227-230)
            if (p.vad_export == 1) {
                for (int e = lane; e < nv * p.K; e += 64) {
                    const int f = e / p.K, k = e - f * p.K;
                    p.pnr[(rbase + slot0 + f) * p.K + k] = Pw[f * PSTRIDE + k];
                }
            } else {
                const int f8e = lane >> 3, ge = lane & 7;
                float es = 0.f;
                for (int k = ge * p.kstride; k < p.K; k += 8 * p.kstride) {
                    const float x = Pw[f8e * PSTRIDE + k];
                    es += x * x;
                }
                es = lanes8_allreduce_add(es);
                if (ge == 0 && f8e < nv) p.pnr[rbase + slot0 + f8e] = es;
            }
        }
        STAMP(7);  // hand-over to phase 2 (incl. NR)

        // ================= phase 2 (wave-local): lane = (frame, band group) =================
        // The step's 8 frames x 8 band groups.  Bands are dealt to (slot, group) cells by the host so that the
        // 8 bands of a slot have similar widths; every group walks the same number of 4-bin chunks per slot.
        if (o_dbg != 1 && !o_skip_phase2 && nv > 0 && !(CTU_ABL & 64)) {
            // lane -> (frame f8, group g).  MD: lane = f8 + 8 h + 16 kk with g = kk + 4 h, the B-operand layout of the MFMA
            const int f8 = MD ? (lane & 7) : (lane >> 3), g = MD ? (((lane >> 3) & 1) * 4 + (lane >> 4)) : (lane & 7);
            const int fslot = slot0 + f8;
            const bool fvalid = f8 < nv;
            const float *prow2 = Pw + f8 * PSTRIDE;
            // FEAT_LPD: LP analysis on uncompressed band energies (no -fb_inld).  The autocorrelation values of a squared
            // spectrum are nearly equal and the normal equations ill-conditioned: fp32 sums and recursions lose everything
            // (1e-2 .. NaN against a double tail on the same band energies), while the result is insensitive to the 1e-6
            // relative noise of the fp32 band energies themselves.  Accumulation, Levinson-Durbin and a -> c run in double.
            constexpr bool LPD = FEAT == FEAT_LPD;
            constexpr bool IS_LP = FEAT == FEAT_LP || LPD;
            typedef std::conditional_t<LPD, double, float> lp_t;
            lp_t c[NC];
#pragma unroll
            for (int i = 0; i < NC; i++) c[i] = 0;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            double esum = 0.0;  // -fea_E sums squares of values that may already be powers: beyond the float range on silent frames
            if (o_e_mode == 4) {  // raw energy: sum of x[i]^2, i = 1..window-1 (src/io/in.cc:353-361); rare, read from HBM
                const int16_t *xr = p.pcm + rec.sbase + (int64_t)(f8 < nv ? fslot : slot0) * p.wshift;
                for (int i = 1 + g; i < p.window; i += 8) {
                    const float x = (float)xr[i];
                    esum += (double)x * x;
                }
            }
            if (o_e_mode == 5) {  // -nr_when afterFB: E = in->E = log(2 (P0/2 + sum P_i + P_{K-1}/2)) of the power spectrum (src/io/in.cc:404-413, batch.cc:101-104)
                for (int k = g * p.kstride; k < p.K; k += 8 * p.kstride) {
                    const float x = prow2[k];
                    esum += ((k == 0 || k == p.K - 1) ? 0.5 : 1.0) * (o_fb_power ? (double)x : (double)x * x);
                }
            }
            if (o_e_mode == 1) {  // E = log(2 (X0^2/2 + sum X_i^2 + X_{K-1}^2/2)) on the post-NR vector (src/nr/nr.cc:36-45)
                for (int k = g * p.kstride; k < p.K; k += 8 * p.kstride) {
                    const float x = prow2[k];
                    esum += ((k == 0 || k == p.K - 1) ? 0.5 : 1.0) * ((double)x * x);
                }
            }
            // A cell's record {first bin of its chunk run, band index or -1, first chunk of the slot, chunks of the slot} is the only
            // per-lane indirection of a slot.  The next slot's record and this slot's DCT operands are fetched before the chunk walk
            // starts, so no LDS round trip stands between two slots; the base addresses go through an empty asm, so the compiler
            // addresses every chunk as base + immediate instead of re-deriving the table's address (a link-time LDS symbol plus an
            // offset beyond the 16-bit field) with VALU adds per read.
            typedef const __attribute__((address_space(3))) f32x4 lf4_t;
            typedef const __attribute__((address_space(3))) int32x4 li4_t;
            typedef const __attribute__((address_space(3))) float lf_t;
            uint32_t cella = (uint32_t)(size_t)(lvoid_t *)(ltab + p.ck_off + g * 4);
            uint32_t prowa = (uint32_t)(size_t)(lvoid_t *)prow2;
            uint32_t wbasea = (uint32_t)(size_t)(lvoid_t *)(ltab + g * 4);
            uint32_t ama = (uint32_t)(size_t)(lvoid_t *)(ltab + p.am_off + lane);
            asm volatile("" : "+v"(cella), "+v"(prowa), "+v"(wbasea), "+v"(ama));
            li4_t *cellp = (li4_t *)(size_t)cella;
            lf_t *amp = (lf_t *)(size_t)ama;
            int32x4 nxt = cellp[0];
            for (int sl = 0; sl < p.NS; sl++) {
                const int32x4 cur = nxt;
                cellp += 8;
                nxt = cellp[0];  // the table ends on a slot of idle cells
                float a0 = 0.f, a1 = 0.f;
                if constexpr (MD && !LPD) {
                    a0 = amp[0];
                    a1 = amp[64];
                    amp += 128;
                }
                const int kstart = cur.x, bidx = cur.y;
                const int nch = __builtin_amdgcn_readfirstlane(cur.w);
                lf4_t *pq = (lf4_t *)(size_t)(prowa + 4u * (uint32_t)kstart);  // kstart is a multiple of 4
                lf4_t *wq = (lf4_t *)(size_t)(wbasea + 128u * (uint32_t)cur.z);
                float acc = 0.f, accb = 0.f;
                int ch = 0;
                for (; ch + 4 <= nch; ch += 4) {  // 4 chunks per group: 8 LDS reads in flight, then 16 FMAs
                    f32x4 w4[4], p4[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) w4[u] = wq[u * 8];
#pragma unroll
                    for (int u = 0; u < 4; u++) p4[u] = pq[u];
#pragma unroll
                    for (int u = 0; u < 4; u += 2) {
                        acc += w4[u].x * p4[u].x;
                        accb += w4[u + 1].x * p4[u + 1].x;
                        acc += w4[u].y * p4[u].y;
                        accb += w4[u + 1].y * p4[u + 1].y;
                        acc += w4[u].z * p4[u].z;
                        accb += w4[u + 1].z * p4[u + 1].z;
                        acc += w4[u].w * p4[u].w;
                        accb += w4[u + 1].w * p4[u + 1].w;
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // DS reads
                    __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);  // VALU
                    pq += 4;
                    wq += 32;
                }
                if (nch & 2) {
                    const f32x4 wa = wq[0], wb = wq[8];
                    const f32x4 pa = pq[0], pb = pq[1];
                    acc += wa.x * pa.x;
                    accb += wa.y * pa.y;
                    acc += wa.z * pa.z;
                    accb += wa.w * pa.w;
                    acc += wb.x * pb.x;
                    accb += wb.y * pb.y;
                    acc += wb.z * pb.z;
                    accb += wb.w * pb.w;
                    pq += 2;
                    wq += 16;
                }
                if (nch & 1) {
                    const f32x4 w4 = wq[0];
                    const f32x4 p4 = pq[0];
                    acc += w4.x * p4.x;
                    accb += w4.y * p4.y;
                    acc += w4.z * p4.z;
                    accb += w4.w * p4.w;
                }
                acc += accb;
                float y = acc;
                if (o_fb_inld) y = __builtin_amdgcn_exp2f(0.33f * __builtin_amdgcn_logf(y));  // pow(Y, 0.33), src/fea/fb.cc:81-83
                if (FULL && o_after_fb && o_nr_exten) {
                    // exten on the band energies (src/io/batch.cc:207-210; nr.cc:95-140 on fb->_Y): sequential over the step's
                    // frames for every band, state per band in LDS (per wave: [2][64] floats behind the tables), lane (f8, g)
                    // takes its turn when f8 comes up.  GEN_FULL only: lane = 8 f8 + g.
                    float *st = ltw + LTW_FLOATS + wave * 128 + sl * 8 + g;  // Navg at [0], Yavg at [64]
                    if (rec.t0 == 0 && slot0 == 0 && f8 == 0) {
                        st[0] = 0.95f;
                        st[64] = 0.05f;
                    }
                    for (int i = 0; i < nv; i++) {
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (f8 == i && bidx >= 0) {
                            const float na = st[0], ya = st[64];
                            float H, omH;
                            if (p.nr_a == 1.0f) {
                                const float ir = __builtin_amdgcn_rcpf(na + ya);
                                H = na * ir;
                                omH = ya * ir;
                            } else if (p.nr_a == 2.0f) {
                                const float r2 = na * na + ya * ya;
                                const float ir = __builtin_amdgcn_rsqf(r2);
                                const float r = r2 * ir;
                                H = na * ir;
                                omH = (ya * ya) * __builtin_amdgcn_rcpf(r * (r + na));
                            } else {
                                H = na / powf(powf(na, p.nr_a) + powf(ya, p.nr_a), 1.0f / p.nr_a);
                                omH = 1.0f - H;
                            }
                            const float nn = p.nr_p * na + (1.0f - p.nr_p) * (H * y);
                            st[0] = nn;
                            st[64] = fabsf(y - nn);
                            y *= omH;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (o_e_mode == 3 && bidx >= 0)  // band energy of the FB output, i.e. after its ^0.33 (src/fea/fea_impl.cc:44-50,68-74)
                    esum += ((bidx == 0 || bidx == p.B - 1) ? 0.5 : 1.0) * ((double)y * y);
                // v_log_f32 (log2, ~1 ulp) * ln 2: band energies of int16 speech are far from the denormal range
                if (FEAT == FEAT_DCTC || (FEAT == FEAT_BANDS && p.band_log)) y = __builtin_amdgcn_logf(y) * 0.69314718056f;
                if (FEAT == FEAT_BANDS) {
                    float *dst = p.band_to_scratch ? p.logmel : p.rows;
                    const int out_w = p.band_to_scratch ? p.B : p.D;
                    if (bidx >= 0 && fvalid) uniform_ptr(dst + (rbase + slot0) * out_w)[(unsigned)(f8 * out_w + bidx)] = y;
                } else {
                    if (FEAT == FEAT_LP && !o_fb_inld) y *= y;  // src/fea/fea_impl.cc:165-169
                    y = bidx >= 0 ? y : 0.f;  // idle cell: its log(0) must not meet the zero coefficients
                    if constexpr (LPD) {
                        const double yd = o_fb_inld ? (double)y : (double)y * (double)y;
                        cell_accumulate<NC>(c, reinterpret_cast<const double *>(ltab + p.cfd_off) + (sl * 8 + g) * NC, yd);
                    } else if constexpr (MD) {
                        // D[m][n] += sum_kk A[m][kk] B[kk][n]: B = this slot's band logarithms as they stand (n = lane & 15,
                        // kk = lane >> 4); A = DCT rows of the bands in groups kk (columns n < 8) or kk + 4 (columns n >= 8)
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, y, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, y, acc1, 0, 0, 0);
                    } else {
                        if constexpr (!LPD) {
                            const float4 *cf = reinterpret_cast<const float4 *>(ltab + p.cf_off + (sl * 8 + g) * (NC + 4));  // +4: bank spread
                            cell_accumulate<NC>(c, cf, y);
                        }
                    }
                }
            }
            STAMP(8);  // filter bank + per-band accumulation
            if (o_e_mode && !(FEAT == FEAT_BANDS && p.band_to_scratch)) {
                float e = 0.f;
                if (o_e_mode == 1 || o_e_mode == 3 || o_e_mode == 5) e = (float)log(2.0 * lanes8_allreduce_add(esum));
                else if (o_e_mode == 4) e = (float)log(lanes8_allreduce_add(esum));
                if (o_e_mode != 2 && fvalid && g == 0) p.rows[(rbase + fslot) * p.D + p.e_slot] = e;
            }
            if constexpr (MD) {
                // frame f8's coefficients: columns f8 of acc0 (groups 0-3) + f8 + 8 of acc1 (groups 4-7); lane f8 + 16 j
                // then holds output slots 4j .. 4j+3 (norm, lifter and the writer's c1..cN,c0 order are in the table)
                float o4[4];
#pragma unroll
                for (int r = 0; r < 4; r++) o4[r] = acc0[r] + dpp_mov<0x128>(acc1[r]);  // row_ror:8 brings column n + 8
                if constexpr (FEAT == FEAT_LP) {
                    // rows of the table = lags of the cosine iDFT (src/fea/fea_impl.cc:181-198): lags 4j .. 4j+3 go to the frame's
                    // scratch row for lp_tail_kernel (Levinson-Durbin and a -> c, one frame per lane)
                    const int P_ = LPO ? LPO : p.lporder;
                    auto rrow = uniform_ptr(reinterpret_cast<float *>(p.lp_r) + (rbase + slot0) * p.lp_stride);
                    const unsigned ro = (unsigned)(f8 * p.lp_stride + 4 * (lane >> 4));
                    if (fvalid && (lane & 8) == 0) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (4 * (lane >> 4) + r <= P_) rrow[(size_t)ro + r] = o4[r];
                    }
                } else {
                    auto orow = uniform_ptr(p.rows + (rbase + slot0) * p.D);
                    const unsigned ro = (unsigned)(f8 * p.D + 4 * (lane >> 4));
                    if (fvalid && (lane & 8) == 0) {
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            if (4 * (lane >> 4) + r < p.ncoef_out) orow[(size_t)ro + r] = o4[r];
                    }
                }
            } else if (FEAT == FEAT_DCTC || IS_LP) {
                cells_reduce<NC>(c);
                float *orow = p.rows + (rbase + fslot) * p.D;
                if (FEAT == FEAT_DCTC) {
                    // c[r] = value of output slot r = sum_b dct[i(r)][b] * logY[b]  (norm, lifter and the writer's
                    // c1..cN,c0 order are folded into the table on the host); lane g stores slots g, g+8, g+16
#pragma unroll
                    for (int h = 0; h < NC / 8; h++) {
                        if (h * 8 < p.ncoef_out) {
                            float val = c[h * 8];
#pragma unroll
                            for (int j = 1; j < 8; j++) val = (g == j) ? c[h * 8 + j] : val;
                            if (fvalid && h * 8 + g < p.ncoef_out) orow[h * 8 + g] = val;
                        }
                    }
                } else {
                    // c[k] = R[k], the autocorrelation by cosine iDFT (src/fea/fea_impl.cc:181-198), the same in the eight lanes
                    // of a frame.  Levinson-Durbin and a -> c are one short sequential recursion per FRAME: run here, eight lanes
                    // would repeat it for each of the step's eight frames.  The lags go to a scratch row instead (lane g stores
                    // lags g, g + 8, g + 16) and lp_tail_kernel finishes 64 frames per wave, one per lane (lp_tail_kernel.h).
                    constexpr int PM = LPO ? LPO : MAX_LP;
                    const int P_ = LPO ? LPO : p.lporder;
                    lp_t *rrow = reinterpret_cast<lp_t *>(p.lp_r) + (rbase + fslot) * p.lp_stride;
#pragma unroll
                    for (int h = 0; h < (PM + 8) / 8; h++) {
                        if (h * 8 <= P_ && h * 8 < NC) {
                            lp_t val = c[h * 8];
#pragma unroll
                            for (int j = 1; j < 8; j++)
                                if (h * 8 + j < NC) val = (g == j) ? c[h * 8 + j] : val;
                            if (fvalid && h * 8 + g <= P_) rrow[h * 8 + g] = val;
                        }
                    }
                }
            }
            STAMP(10);  // reduction, tail, row store
        }
        // ================= Burg-cepstral VAD criterion of the step's frames (vad_fused.h) =================
        if constexpr (VF) {
            if (nv > 0) {
                float mine_ab[2];
                rebuild_cepstra(std::integral_constant<int, VF_NC>{}, std::false_type{}, mine_ab);
                // the cepstra go to the scratch rows of their frames; vad_lanes_kernel replays the detector's recurrences with one
                // utterance per lane (vad_kernels.h).  256-point mode: mine_ab[x] of lane 16 fg + i = coefficient i of frame slot
                // 2 fg + x; 512-point mode: of frame slot 4 x + fg.
#pragma unroll
                for (int x = 0; x < 2; x++) {
                    if constexpr (CTU_VF8 && MODE == 1) {  // lane 8 f + i: coefficients i and i + 8 of frame slot f
                        int ln = lane;
                        asm volatile("" : "+v"(ln));  // the lane's offset is two instructions: not worth a register across the whole step loop
                        const int sl = ln >> 3, ci = (ln & 7) + 8 * x;
                        if (sl < nv && ci < VF_NC) uniform_ptr(p.vad_cf + (rbase + slot0) * VFC_STRIDE)[(unsigned)(sl * VFC_STRIDE + ci)] = mine_ab[x];
                    } else {
                        const int sl = MODE == 1 ? 2 * fg + x : 4 * x + fg;
                        if (sl < nv && l16 < VF_NC) uniform_ptr(p.vad_cf + (rbase + slot0) * VFC_STRIDE)[(unsigned)(sl * VFC_STRIDE + l16)] = mine_ab[x];
                    }
                }
                STAMP(13);  // VF: cepstra stored
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        }  // steps of 8 frames
        if (next < 0) break;
        rec = nrec;
        tile = next;  // SS looks its utterance up by tile (seed in, last vector out)
    }
#if CTU_STAMP
    if (lane == 0 && p.stamps)
        for (int i = 0; i < 16; i++) p.stamps[(blockIdx.x * NWAVE + wave) * 16 + i] = st_acc[i];
#endif
}

}  // namespace
