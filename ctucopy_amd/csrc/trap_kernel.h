// TRAP-DCT on the matrix cores.
// Included by engine.hip (one translation unit: the kernels and their host launchers share types).
#pragma once

namespace {

// TRAP-DCT (src/fea/fea_trap.cc:53-127): out[t][b*ndct+k] = sum_j G[k][j] * logmel[clamp(t-half+j)][b]
// with mean removal, Hamming and REDFT10 folded into G on the host.  Unlike the banded filter bank this IS a dense
// contraction (ndct x traplen per band and frame), so it runs on the matrix cores: v_mfma_f32_16x16x4_f32 (exact
// fp32 FMA chain), D[k][t] += G[k][4s..4s+3] * X[4s..4s+3][t] with the Toeplitz operand X[j][t] = x[t+j-half][b]
// read straight from an LDS tile of log-mel frames.  Rows of G sum to zero, so each column is offset by its centre
// value first (keeps the fp32 accumulation small).
// One workgroup = 64 output frames of one utterance (4 waves x 16 frames), all bands.

template <int NRB, int NSM>  // row blocks of 16 DCT coefficients (ndct <= 16*NRB); NSM >= ceil(traplen/4) tap groups
__global__ __launch_bounds__(256) void trapdct_mfma_kernel(const float *__restrict__ logmel, float *__restrict__ rows,
                                                           const float *__restrict__ G, const int4 *__restrict__ utt_info,
                                                           const int *__restrict__ chunk_tab, int B, int traplen, int ndct, int D) {
    extern __shared__ float tile[];  // [64 + 4*nsteps][Bs]
    const int u = chunk_tab[blockIdx.x * 2], tc = chunk_tab[blockIdx.x * 2 + 1];
    const int4 ui = utt_info[u];
    const int64_t r0 = ((int64_t)ui.y << 32) | (uint32_t)ui.x;
    const int T = ui.z;
    const int half = (traplen - 1) / 2, nsteps = (traplen + 3) / 4;
    const int Bs = B | 1, nfr = 64 + 4 * (NSM <= 32 ? NSM : nsteps);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // log-mel tile with the first / last frame replicated beyond the utterance (src/fea/fea_trap.cc:64-70,111-127)
    for (int e = tid; e < nfr * B; e += 256) {
        const int f = e / B, b = e - f * B;
        int t = tc - half + f;
        t = t < 0 ? 0 : (t > T - 1 ? T - 1 : t);
        tile[f * Bs + b] = logmel[(r0 + t) * B + b];
    }
    // this lane's slice of G: A[i = lane&15][k = lane>>4] of every 16x4 block
    const int ai = lane & 15, ak = lane >> 4;
    float areg[NRB][NSM];
#pragma unroll
    for (int rb = 0; rb < NRB; rb++)
#pragma unroll
        for (int s_ = 0; s_ < NSM; s_++) {
            const int k = rb * 16 + ai, j = 4 * s_ + ak;
            areg[rb][s_] = (s_ < nsteps && k < ndct && j < traplen) ? G[k * traplen + j] : 0.f;
        }
    __syncthreads();
    const int tl = wave * 16 + (lane & 15);  // local output frame of this lane's column
    const int t_out = tc + tl;
    for (int b = 0; b < B; b++) {
        const float xc = tile[(tl + half) * Bs + b];
        f32x4 acc[NRB];
#pragma unroll
        for (int rb = 0; rb < NRB; rb++) acc[rb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s_ = 0; s_ < NSM; s_++) {
            if (NSM <= 32 || s_ < nsteps) {  // exact instantiations run unguarded (A is zero beyond traplen)
                const float bv = tile[(tl + 4 * s_ + ak) * Bs + b] - xc;
#pragma unroll
                for (int rb = 0; rb < NRB; rb++) acc[rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[rb][s_], bv, acc[rb], 0, 0, 0);
            }
        }
        if (t_out < T && ndct == 16 && NRB == 1 && (D & 3) == 0) {
            // C/D layout: this lane holds coefficients 4*(lane>>4)..+3 of frame column lane&15: one 16-byte store
            *reinterpret_cast<f32x4 *>(rows + (r0 + t_out) * D + b * 16 + ak * 4) = acc[0];
        } else if (t_out < T) {
            float *o = rows + (r0 + t_out) * D + b * ndct;
#pragma unroll
            for (int rb = 0; rb < NRB; rb++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int k = rb * 16 + ak * 4 + r;  // C/D layout: row = (lane>>4)*4 + reg, col = lane&15
                    if (k < ndct) o[k] = acc[rb][r];
                }
        }
    }
}

}  // namespace
