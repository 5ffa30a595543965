// Thin extern "C" entry around the reference's own CepstralDetector<BurgCepstrumEstimator>
// (src/vdet/CepstralDet.h:92-217, with Burg.h, Fft.h / Fft.cc, Complex.h: all FFTW-free).
// Compiled only by oracle/Makefile into oracle/_ref/ when /root/reference exists (the sources stay where they
// lie); used by tests/test_oracle_cepdet_ref.py to pin the oracle's restatement ctuo_cepdet_*.
#include "CepstralDet.h"

typedef Voice::CepstralDetector<Voice::BurgCepstrumEstimator> CDBurg;

extern "C" {

void *ref_cepdet_new(int npoints, int ninit, int ncoefs, double p, double q) {
    CDBurg::Options opt(ninit, ncoefs, p, q);
    return new CDBurg(npoints, opt);
}
// frames: n_frames x npoints doubles; decisions: one byte per frame (what Process returns)
void ref_cepdet_run(void *h, const double *frames, int npoints, int n_frames, unsigned char *decisions) {
    CDBurg *d = static_cast<CDBurg *>(h);
    for (int t = 0; t < n_frames; t++) decisions[t] = d->Process(frames + (long)t * npoints, frames + (long)(t + 1) * npoints) ? 1 : 0;
}
void ref_cepdet_delete(void *h) { delete static_cast<CDBurg *>(h); }

}
