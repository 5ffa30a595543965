// Init-time table design on the host, in double precision, for the device kernels.
//
// Everything here is computed once per engine and must match the reference's init-time arithmetic
// bit-for-bit in double before it is rounded to float for the GPU:
//   Hamming window            src/io/in.cc:139-144
//   filter bank               src/fea/fb.cc:100-132 (axes), 134-184 (PLP), 186-253 (grammar),
//                             255-429 (triang/rect), 432-447 (first/last non-zero)
//   DCT-II + lifter           src/fea/fea_impl.cc:81-131
//   cosine iDFT for LPC       src/fea/fea_impl.cc:141-198
//   TRAP Hamming / REDFT10    src/fea/fea_trap.cc:20-107
//   output geometry           src/io/out.cc:95-113,145-171
//   delta / stacking chain    src/io/batch.cc:122-130, src/fea/fea_delta.cc:20-60 (geometry only)
#pragma once

#include <string>
#include <vector>

#include "opts.h"

namespace ctu {

enum class FeaKind { Spec, LogSpec, Dctc, Lpa, Lpc, TrapDct, None };

struct DesignError : std::runtime_error {
    using std::runtime_error::runtime_error;
};

struct Design {
    Opts o;  // options after the filter-bank constructor's overrides (PLP forces bark/eqld/inld)
    int window = 0, wshift = 0, wfft = 0, K = 0, B = 0;
    FeaKind kind = FeaKind::Dctc;
    int nfea = 0;      // internal feature vector length (fvec)
    int D = 0;         // floats per output row
    int Dbase = 0;     // floats per row as the front end writes it (== D unless a delta / stacking chain follows)
    int post_order = 0;        // number of chained deltaFEA stages (0 = none; -fea_trap: 1)
    bool post_stack = false;   // -fea_trap: the single stage stacks 2*d_win+1 frames instead of differentiating
    int post_w[3] = {0, 0, 0}; // window half-width of each stage (d_win, a_win, t_win)
    bool signal_out = false;   // -format_out raw|wave: IN -> NR -> sigOUT, no FB / FEA (src/io/batch.cc:62-65)
    double ola_corr = 1.0;     // largest sum of overlapping Hamming windows (src/io/out.cc:355-377)
    int cms = 0;               // cepstral mean subtraction after the chain: 0 off, 1 exponential, 2 block (src/fea/post_impl.cc:159-240)
    int cms_cols = 0;          // leading row columns it touches (c1..cN and, with -fea_c0, c0)
    int htk_kind = 0;  // HTK parameter kind incl. qualifier bits
    unsigned period = 0;

    std::vector<double> hamming;             // [window]
    std::vector<std::vector<double>> fb;     // [B][K]
    std::vector<int> fb_first, fb_last;      // first / last non-zero bin of each band
    std::vector<double> dct;                 // [(ncep+1)][B]: sqrt(2/B) * cos(...) * lifter, row i = c_i
    std::vector<double> lifter;              // [ncep]
    std::vector<double> idft;                // [(p+1)][B]: R[k] = sum_n idft[k][n] * Y[n]
    std::vector<double> trap;                // [ndct][traplen]: mean-removal + Hamming + REDFT10 folded
    // output row: slot of fvec[i] in the written row (-1 = not written); E slot or -1
    std::vector<int> row_slot;
    int e_slot = -1;

    explicit Design(const Opts &opts);
};

}  // namespace ctu
