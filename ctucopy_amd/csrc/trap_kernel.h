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

// ---------------------------------------------------------------------------------------------------------------------
// The same contraction on the 16-bit matrix pipe with fp32 accuracy.  Every operand is split into 16-bit terms and the
// significant cross products accumulate in fp32:
//   F16 = true   two fp16 terms  (x = h + l to 2^-22):  hh, hl, lh            3 MFMAs per 32 taps  (ll, 2^-22, dropped)
//   F16 = false  three bf16 terms (x = h + m + l to 2^-24): hh, hm, mh, hl, lh, mm   6 MFMAs     (ml, lm, ll dropped)
// v_mfma_f32_16x16x32_{f16,bf16} do 8x the multiply-adds of the fp32 form per cycle.  The kernel is bound by that pipe
// at the clock the chip holds under it (neither the barrier, nor occupancy, nor the stores, nor the dependency chain of
// the accumulator moved its time), so the fp16 split - half the MFMAs - is the one that runs; the log-mel values are
// centred per band (|x| of a few units) and G is at most 1, far inside fp16's range, and an element's absolute error is
// max(2^-22 |x|, 3e-8) (fp16 subnormal spacing).  The bf16 form (any range) is kept for the record.
// The Toeplitz operand X[j][t] = x[t + j - half] wants 8 consecutive taps per lane (16-byte LDS reads), i.e. a start that
// is a multiple of 8 for every column: the 16 columns of an MFMA are therefore output frames 8 apart (t = tc + 8 n + c),
// and each of the 8 phases c has its own copy of G shifted by c taps (host table, zero-padded to K = 128):
//     D[k][n] = sum_j' Gc[k][j'] X[8 n + j'],   X[tau] = x(tc - OFF + tau) - x0(band),   j' = j + c + OFF - half
// (rows of G sum to zero, so the per-band constant x0 - the tile's centre value - drops out and keeps the split small).
// One workgroup (8 waves, at most 128 VGPRs: two workgroups = four waves per SIMD on a CU) = 128 output frames x all
// bands; wave w takes bands w, w+8, ...; A fragments of a phase (4 k-steps x NT terms) live in registers meanwhile.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int TB_TT = 264;   // tile row in frames: 8 * 15 + 127 < 256 used; 528 bytes per row spreads the bands over the banks and keeps 16-byte alignment
constexpr int TB_OFF = 56;   // tile origin tc - 56: a multiple of 8 not below half = 50

// Workgroup barrier for LDS hand-overs only: __syncthreads() also drains the vector-memory counter, which would put the
// latency of a phase's row stores in front of the next phase.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ unsigned bf16_rn(float v) {  // round to nearest even, finite inputs
    unsigned u = __float_as_uint(v);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ unsigned f16_bits(_Float16 h) { return (unsigned)__builtin_bit_cast(unsigned short, h); }

// the 16-bit terms of a value, most significant first
template <bool F16>
__device__ __forceinline__ void split16(float v, unsigned (&t)[3]) {
    if constexpr (F16) {
        const _Float16 h = (_Float16)v;  // v_cvt_f16_f32, round to nearest even
        const _Float16 l = (_Float16)(v - (float)h);
        t[0] = f16_bits(h);
        t[1] = f16_bits(l);
        t[2] = 0;
    } else {
        t[0] = bf16_rn(v);
        const float r1 = v - __uint_as_float(t[0] << 16);
        t[1] = bf16_rn(r1);
        t[2] = bf16_rn(r1 - __uint_as_float(t[1] << 16));
    }
}

#ifndef TB_WGS
#define TB_WGS 2
#endif
template <bool F16>
__global__ __launch_bounds__(512, TB_WGS) void trapdct_split16_kernel(const float *__restrict__ logmel, float *__restrict__ rows,
                                                                const uint4 *__restrict__ Gtab, const int4 *__restrict__ utt_info,
                                                                const int *__restrict__ chunk_tab, int n_chunks, int B, int ndct, int D) {
    constexpr int NT = F16 ? 2 : 3;       // terms per operand
    constexpr int NA = 4 * NT * 64;       // uint4 fragments of a phase
    extern __shared__ __align__(16) unsigned short xt[];  // [NT][B][TB_TT] terms, then x0[B] floats, then the fragment buffers
    if ((int)blockIdx.x >= n_chunks) return;
    const int u = chunk_tab[blockIdx.x * 2], tc = chunk_tab[blockIdx.x * 2 + 1];  // chunks of 128 frames
    const int4 ui = utt_info[u];
    const int64_t r0 = ((int64_t)ui.y << 32) | (uint32_t)ui.x;
    const int T = ui.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float *x0 = reinterpret_cast<float *>(xt + NT * B * TB_TT);
    if (tid < B) {
        int t = tc + 64;
        t = t > T - 1 ? T - 1 : t;
        x0[tid] = logmel[(r0 + t) * B + tid];
    }
    __syncthreads();
    // log-mel tile with the first / last frame replicated beyond the utterance (src/fea/fea_trap.cc:64-70,111-127)
    unsigned *xt32 = reinterpret_cast<unsigned *>(xt);
    {
        // two frames per lane and step (one dword store per term); every load of the tile is issued before the first use
        constexpr int NF = 6;  // ceil(128 * 24 / 512): B <= 24 on this path
        float v0[NF], v1[NF];
#pragma unroll
        for (int it = 0; it < NF; it++) {
            const int e = min(tid + 512 * it, 128 * B - 1);  // unconditional loads (a branch per step would serialise them)
            const int pr = e / B, b = e - pr * B;
            int t0_ = tc - TB_OFF + 2 * pr, t1_ = t0_ + 1;
            t0_ = t0_ < 0 ? 0 : (t0_ > T - 1 ? T - 1 : t0_);
            t1_ = t1_ < 0 ? 0 : (t1_ > T - 1 ? T - 1 : t1_);
            v0[it] = logmel[(r0 + t0_) * B + b];
            v1[it] = logmel[(r0 + t1_) * B + b];
        }
#pragma unroll
        for (int it = 0; it < NF; it++) {
            const int e = tid + 512 * it;
            if (e < 128 * B) {
                const int pr = e / B, b = e - pr * B;
                unsigned ta[3], tb[3];
                split16<F16>(v0[it] - x0[b], ta);
                split16<F16>(v1[it] - x0[b], tb);
#pragma unroll
                for (int sp = 0; sp < NT; sp++) xt32[((sp * B + b) * TB_TT >> 1) + pr] = ta[sp] | (tb[sp] << 16);
            }
        }
    }
    __syncthreads();
    const int n = lane & 15, q = lane >> 4;
    // A fragments of a phase: 4 x NT x 64 lanes x 16 bytes, fetched once per workgroup into LDS one phase ahead (every wave
    // needs all of them).  One LDS-only barrier per phase: nothing in the loop waits for the rows' stores.
    uint4 *abuf = reinterpret_cast<uint4 *>(x0 + ((B + 3) & ~3));  // [2][NA]
    abuf[tid] = Gtab[tid];
    if (tid + 512 < NA) abuf[tid + 512] = Gtab[tid + 512];
    for (int c = 0; c < 8; c++) {
        lds_barrier();  // phase c's fragments are in abuf[c & 1]; every wave is done with the other half
        // next phase's fragments: named registers, fetched unconditionally (the last phase re-reads its own; lanes beyond the
        // table's end re-read an earlier entry): an array defined under a condition went to scratch
        const uint4 *gn = Gtab + (c + 1 < 8 ? c + 1 : c) * NA;
        const uint4 pre0 = gn[tid], pre1 = gn[(tid + 512 < NA) ? tid + 512 : tid];
        uint4 a[4][NT];
        const uint4 *ab = abuf + (c & 1) * NA;
#pragma unroll
        for (int s_ = 0; s_ < 4; s_++)
#pragma unroll
            for (int sp = 0; sp < NT; sp++) a[s_][sp] = ab[(s_ * NT + sp) * 64 + lane];
        const int t_out = tc + 8 * n + c;
        // a band of the wave: 4 x NT tile fragments from LDS, 12 (fp16) or 24 (bf16) MFMAs, one row store; four waves per
        // SIMD cover each other's LDS round trips
        for (int b = wave; b < B; b += 8) {
            uint4 X[4][NT];
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) {
                const int tau = 8 * n + 32 * s_ + 8 * q;
#pragma unroll
                for (int sp = 0; sp < NT; sp++) X[s_][sp] = *reinterpret_cast<const uint4 *>(xt + (sp * B + b) * TB_TT + tau);
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s_ = 0; s_ < 4; s_++) {
                // smallest terms first
                if constexpr (F16) {
#define MF(A_, X_) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A_), __builtin_bit_cast(f16x8, X_), acc, 0, 0, 0)
                    MF(a[s_][1], X[s_][0]);
                    MF(a[s_][0], X[s_][1]);
                    MF(a[s_][0], X[s_][0]);
#undef MF
                } else {
#define MB(A_, X_) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A_), __builtin_bit_cast(bf16x8, X_), acc, 0, 0, 0)
                    MB(a[s_][1], X[s_][1]);
                    MB(a[s_][NT - 1], X[s_][0]);
                    MB(a[s_][0], X[s_][NT - 1]);
                    MB(a[s_][1], X[s_][0]);
                    MB(a[s_][0], X[s_][1]);
                    MB(a[s_][0], X[s_][0]);
#undef MB
                }
            }
            if (t_out < T) {
                if (ndct == 16 && (D & 3) == 0) {
                    // C/D layout: this lane holds coefficients 4*(lane>>4)..+3 of frame column lane&15: one 16-byte store
                    *reinterpret_cast<f32x4 *>(rows + (r0 + t_out) * D + b * 16 + q * 4) = acc;
                } else {
                    float *o = rows + (r0 + t_out) * D + b * ndct;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (q * 4 + r < ndct) o[q * 4 + r] = acc[r];
                }
            }
        }
        uint4 *an = abuf + ((c + 1) & 1) * NA;
        an[tid] = pre0;
        if (tid + 512 < NA) an[tid + 512] = pre1;
    }
}

}  // namespace
