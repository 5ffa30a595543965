// Thin extern "C" entry around the reference's own header-only Burg estimator
// (src/vdet/Burg.h:28-170, src/vdet/CepstralDet.h is NOT used: it pulls Fft.h).
// Compiled only by oracle/Makefile into oracle/_ref/ when /root/reference exists;
// used by tests/test_oracle_burg_ref.py to pin ctuo_burg_cepstrum().
#include "Burg.h"

extern "C" void ref_burg_cepstrum(const double *x, int npoints, int ncoefs, double *a_out, double *c_out,
                                  double *alpha_out) {
    DSP::Burg burg(npoints, ncoefs);
    DSP::Burg2Cepstrum cep(ncoefs);
    burg.Process(x, x + npoints);
    cep.Process(burg);
    for (int i = 0; i < ncoefs; i++) {
        a_out[i] = burg[i];
        c_out[i] = cep[i];
    }
    *alpha_out = burg.GetAlpha();
}
