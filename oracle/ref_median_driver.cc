// Thin extern "C" entry around the reference's own medianFilter (src/vad/vad.h:79-176, header-only, FFTW-free: the header pulls in
// ../io/opts.h and ../base/types.h only), driven the way VAD::process_frame / BATCH::flush_vad drive it (src/vad/vad.cc:692-699,
// 742-745; src/io/batch.cc:230-249).  Compiled only by oracle/Makefile into oracle/_ref/ when /root/reference exists (the sources
// stay where they lie); used by tests/test_oracle_median_ref.py to pin the oracle's majority filter and its feature delay / flush.
// base/types.h relies on <iostream> and `using namespace std` being in force already (the reference gets them from its own prefix header)
#include <cstdio>
#include <cstring>
#include <iostream>
using namespace std;
#include "vad.h"

extern "C" {

// vad0: T raw decisions; feats: T x nfeat feature vectors as they enter the filter.  Returns the number of (decision, vector) pairs that
// come out (each frame once ready, then the flush), written to dec_out / feat_out (capacity T + order).
int ref_median_run(int order, int nfeat, const double *feats, const unsigned char *vad0, int T, unsigned char *dec_out, double *feat_out) {
    Vec<double> feat(nfeat);
    Vec<double> *out = 0;
    medianFilter f(order, &feat, &out);
    f.cleanFilter();
    int n = 0;
    for (int t = 0; t < T; t++) {
        for (int i = 0; i < nfeat; i++) feat[i] = feats[(long)t * nfeat + i];
        const bool d = f.push(vad0[t] != 0);
        if (f.ready) {  // BATCH::save_frame returns before the writer while !vad_ready
            dec_out[n] = d ? 1 : 0;
            for (int i = 0; i < nfeat; i++) feat_out[(long)n * nfeat + i] = (*out)[i];
            n++;
        }
    }
    for (;;) {  // BATCH::flush_vad: while (vad->flush_frame()) ... with VAD::flush_frame returning the filter's `ready`
        const bool d = f.flush_frame();
        if (!f.ready) break;
        dec_out[n] = d ? 1 : 0;
        for (int i = 0; i < nfeat; i++) feat_out[(long)n * nfeat + i] = (*out)[i];
        n++;
    }
    return n;
}

// A list of files through ONE filter object, as the process keeps it (BATCH owns one VAD for the whole list): per file the pushes, the
// flush loop, then VAD::clean() = cleanFilter() (src/io/batch.cc:292-295, src/vad/vad.cc:705-708) - which resets `start` and the ring but
// neither historyIdx nor historySize.  feats are scalars here (nfeat = 1, the frame's own index + 1, so that 0 = an untouched ring slot);
// out receives, per file, the values the writer would see; n_out[f] their count.
void ref_median_list(int order, const int *frames, int n_files, double *out, int *n_out) {
    Vec<double> feat(1);
    Vec<double> *res = 0;
    medianFilter f(order, &feat, &res);
    f.cleanFilter();
    long w = 0;
    for (int k = 0; k < n_files; k++) {
        int n = 0;
        for (int t = 0; t < frames[k]; t++) {
            feat[0] = t + 1;
            f.push(true);
            if (f.ready) { out[w++] = (*res)[0]; n++; }
        }
        for (;;) {
            f.flush_frame();
            if (!f.ready) break;
            out[w++] = (*res)[0];
            n++;
        }
        n_out[k] = n;
        f.cleanFilter();
    }
}

}
