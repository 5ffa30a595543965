// Skeleton of the matrix-pipe front end that DESIGN.md 5a costs and rejects: the INSTRUCTION MIX and DATA FLOW of one wave working
// on 16 frames (columns of the MFMAs), with arbitrary table contents - it computes nothing meaningful, it measures what the chip
// does with that mix.  Optimistic on purpose: the A fragments are read from a 128 KB LDS table that wraps around (the real tables
// are 256 KB and do not fit), nothing is checked, the PCM loads are coalesced dwordx4 reads.
//
// Per 16 frames and wave (lane = (frame, K-quarter)):
//   16 x dwordx4 PCM loads; byte -> fp16 conversion 6 VALU per PCM dword (64 dwords);
//   stage 1: 16 n2 x 2 (even / odd) x 6 MFMAs (3 matrix orders x 2 n1 halves), one LDS fragment read each;
//   fp32 -> two fp16 terms of the 128 outputs (cvt_pkrtz + fma_mixlo/hi); 4 x 4 quarter transposes (128 permlane swaps);
//   stage 2: 32 blocks x 3 products, 64 fragment reads;
//   |E +- T|^2 with the rank-one mean correction for 32 bin pairs; transpose of the 64 P registers (64 swaps); mel bank 160 FMAs
//   with 40 weight reads; band reduce-scatter; 8 logs; DCT 3 MFMAs; one row store.
// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_skeleton.hip -o /tmp/mfma_skel && /tmp/mfma_skel [blocks per wave]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ h8 as_h8(uint4 v) { return __builtin_bit_cast(h8, v); }

constexpr int WG = 512;
constexpr int TAB_FRAGS = 128;  // 1 KB fragments in LDS (128 KB): the real set is 256

__global__ __launch_bounds__(WG, 2) void skeleton(const uint4 *__restrict__ pcm, const uint4 *__restrict__ gtab, float *__restrict__ out, int blocks_per_wave) {
    extern __shared__ uint4 ltab[];  // [TAB_FRAGS][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < TAB_FRAGS * 64; i += WG) ltab[i] = gtab[i];
    __syncthreads();
    const float *wtab = reinterpret_cast<const float *>(ltab);
    int frag = wave * 7;
    auto next_frag = [&]() {
        frag = (frag + 1) & (TAB_FRAGS - 1);
        return ltab[frag * 64 + lane];
    };
    const int gwave = blockIdx.x * (WG / 64) + wave;
    float sink = 0.f;
    for (int blk = 0; blk < blocks_per_wave; blk++) {
        const uint4 *src = pcm + ((size_t)(gwave * blocks_per_wave + blk) * 16) * 64 + lane;
        f4 s1[32];  // stage-1 outputs: 16 n2 x (E, O)
        uint32_t prev_hi = 0, prev_lo = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {      // four PCM loads of 16 B in flight, each covers four n2
            uint4 q[4];
#pragma unroll
            for (int a = 0; a < 4; a++) q[a] = src[(4 * j + a) * 64];
#pragma unroll
            for (int n = 0; n < 4; n++) {  // one n2: four PCM dwords (the lane's four n1), six VALU each
                uint32_t sreg[8];
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const uint32_t d = n == 0 ? q[a].x : n == 1 ? q[a].y : n == 2 ? q[a].z : q[a].w;
                    uint32_t pe = __builtin_amdgcn_perm(d, d, 0x0c000c01u), po = __builtin_amdgcn_perm(d, d, 0x0c020c03u);
                    pe ^= 0x64006480u;
                    po ^= 0x64006480u;
                    uint32_t ue, uo;
                    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(ue) : "v"(pe), "v"(0xe480e400u));
                    asm volatile("v_pk_add_f16 %0, %1, %2" : "=v"(uo) : "v"(po), "v"(0xe480e400u));
                    sreg[2 * a] = ue;
                    sreg[2 * a + 1] = uo;
                }
                const int n2 = 4 * j + n;
                // E form: (xe, x_prev) samples; O form: (xo, xe): two operands of 8 fp16 each per n1 half
                const uint4 be0 = make_uint4(sreg[0], prev_hi, sreg[2], prev_lo), be1 = make_uint4(sreg[4], sreg[1], sreg[6], sreg[3]);
                const uint4 bo0 = make_uint4(sreg[1], sreg[0], sreg[3], sreg[2]), bo1 = make_uint4(sreg[5], sreg[4], sreg[7], sreg[6]);
                prev_hi = sreg[5];
                prev_lo = sreg[7];
                f4 e = {0.f, 0.f, 0.f, 0.f}, o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ord = 0; ord < 3; ord++) {
                    e = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(next_frag()), as_h8(be0), e, 0, 0, 0);
                    e = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(next_frag()), as_h8(be1), e, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(next_frag()), as_h8(bo0), o, 0, 0, 0);
                    o = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(next_frag()), as_h8(bo1), o, 0, 0, 0);
                }
                s1[2 * n2] = e;
                s1[2 * n2 + 1] = o;
            }
        }
        // ---- fp32 -> two fp16 terms (hi toward zero, lo = value - hi): 128 values -> 64 + 64 registers
        uint32_t hi[64], lo[64];
#pragma unroll
        for (int i = 0; i < 32; i++)
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const float a = s1[i][2 * h], b = s1[i][2 * h + 1];
                const uint32_t hh = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
                uint32_t ll;
                asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                             : "=&v"(ll) : "v"(hh), "v"(a), "v"(b));
                hi[2 * i + h] = hh;
                lo[2 * i + h] = ll;
            }
        // ---- 4 x 4 transposes across the lane quarters: 32-lane swaps, then 16-lane swaps, on pairs of registers
#pragma unroll
        for (int i = 0; i < 64; i += 4) {
            auto r0 = __builtin_amdgcn_permlane32_swap(hi[i], hi[i + 2], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(hi[i + 1], hi[i + 3], false, false);
            auto r2 = __builtin_amdgcn_permlane16_swap(r0[0], r1[0], false, false);
            auto r3 = __builtin_amdgcn_permlane16_swap(r0[1], r1[1], false, false);
            hi[i] = r2[0]; hi[i + 1] = r2[1]; hi[i + 2] = r3[0]; hi[i + 3] = r3[1];
            auto t0 = __builtin_amdgcn_permlane32_swap(lo[i], lo[i + 2], false, false);
            auto t1 = __builtin_amdgcn_permlane32_swap(lo[i + 1], lo[i + 3], false, false);
            auto t2 = __builtin_amdgcn_permlane16_swap(t0[0], t1[0], false, false);
            auto t3 = __builtin_amdgcn_permlane16_swap(t0[1], t1[1], false, false);
            lo[i] = t2[0]; lo[i + 1] = t2[1]; lo[i + 2] = t3[0]; lo[i + 3] = t3[1];
        }
        // ---- stage 2: 16 operand groups (k1 x E/T) x 2 row blocks, three products each; then |E +- T|^2 with the mean correction
        float P[64];
        const float mean = s1[0][0] * 0.0025f;
#pragma unroll
        for (int g = 0; g < 8; g++) {  // a group: E and T of one k1, two row blocks -> 4 accumulators = 8 bin pairs
            const uint4 xh_e = make_uint4(hi[8 * g], hi[8 * g + 1], hi[8 * g + 2], hi[8 * g + 3]), xl_e = make_uint4(lo[8 * g], lo[8 * g + 1], lo[8 * g + 2], lo[8 * g + 3]);
            const uint4 xh_t = make_uint4(hi[8 * g + 4], hi[8 * g + 5], hi[8 * g + 6], hi[8 * g + 7]), xl_t = make_uint4(lo[8 * g + 4], lo[8 * g + 5], lo[8 * g + 6], lo[8 * g + 7]);
            f4 acc[4];
#pragma unroll
            for (int rb = 0; rb < 2; rb++) {
                const uint4 mh = next_frag(), ml = next_frag(), nh = next_frag(), nl = next_frag();
                f4 e = {0.f, 0.f, 0.f, 0.f}, t = {0.f, 0.f, 0.f, 0.f};
                e = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(ml), as_h8(xh_e), e, 0, 0, 0);
                e = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(mh), as_h8(xl_e), e, 0, 0, 0);
                e = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(mh), as_h8(xh_e), e, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(nl), as_h8(xh_t), t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(nh), as_h8(xl_t), t, 0, 0, 0);
                t = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(nh), as_h8(xh_t), t, 0, 0, 0);
                acc[2 * rb] = e;
                acc[2 * rb + 1] = t;
            }
#pragma unroll
            for (int rb = 0; rb < 2; rb++)
#pragma unroll
                for (int c = 0; c < 2; c++) {  // two complex bins per accumulator
                    const float4 rc = reinterpret_cast<const float4 *>(wtab)[(g * 4 + rb * 2 + c) * 64 + lane];  // the rectangle's spectrum at this bin
                    const float er = acc[2 * rb][2 * c] - mean * rc.x, ei = acc[2 * rb][2 * c + 1] - mean * rc.y;
                    const float tr = acc[2 * rb + 1][2 * c] - mean * rc.z, ti = acc[2 * rb + 1][2 * c + 1] - mean * rc.w;
                    const float ur = er + tr, ui = ei + ti, vr = er - tr, vi = ei - ti;
                    P[8 * g + 4 * rb + 2 * c] = ur * ur + ui * ui;
                    P[8 * g + 4 * rb + 2 * c + 1] = vr * vr + vi * vi;
                }
        }
        // ---- the 64 P registers across the quarters (adjacent bins to adjacent quarters), then the mel bank: ~2.5 bands per register
#pragma unroll
        for (int i = 0; i < 64; i += 4) {
            auto r0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[i]), __float_as_uint(P[i + 2]), false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(P[i + 1]), __float_as_uint(P[i + 3]), false, false);
            auto r2 = __builtin_amdgcn_permlane16_swap(r0[0], r1[0], false, false);
            auto r3 = __builtin_amdgcn_permlane16_swap(r0[1], r1[1], false, false);
            P[i] = __uint_as_float(r2[0]); P[i + 1] = __uint_as_float(r2[1]); P[i + 2] = __uint_as_float(r3[0]); P[i + 3] = __uint_as_float(r3[1]);
        }
        float band[26];
#pragma unroll
        for (int b = 0; b < 26; b++) band[b] = 0.f;
#pragma unroll
        for (int i = 0; i < 64; i += 2) {  // a weight read of 16 B serves two registers: 5 FMAs
            const float4 w = reinterpret_cast<const float4 *>(wtab)[(64 + (i >> 1)) * 64 + lane];
            const int b = (i * 26) / 64;
            band[b] += w.x * P[i];
            band[b + 1 < 26 ? b + 1 : b] += w.y * P[i];
            band[b] += w.z * P[i + 1];
            band[b + 1 < 26 ? b + 1 : b] += w.w * P[i + 1];
            if ((i & 2) == 0) band[b + 2 < 26 ? b + 2 : b] += w.x * P[i + 1];
        }
        // ---- reduce-scatter over the quarters: 13 + 7 swap-and-add pairs, each lane ends with 8 bands (one padded)
        float kept[8];
        {
            float half[13];
#pragma unroll
            for (int b = 0; b < 13; b++) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(band[b]), __float_as_uint(band[13 + b]), false, false);
                half[b] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
#pragma unroll
            for (int b = 0; b < 7; b++) {
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(half[b]), __float_as_uint(half[b + 6]), false, false);
                kept[b] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            }
            kept[7] = 1.f;
        }
        uint32_t lh[4], ll4[4];
#pragma unroll
        for (int b = 0; b < 8; b += 2) {
            const float a = __builtin_amdgcn_logf(fabsf(kept[b]) + 1e-10f), c = __builtin_amdgcn_logf(fabsf(kept[b + 1]) + 1e-10f);
            const uint32_t hh = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, c));
            uint32_t l2;
            asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                         : "=&v"(l2) : "v"(hh), "v"(a), "v"(c));
            lh[b >> 1] = hh;
            ll4[b >> 1] = l2;
        }
        f4 cep = {0.f, 0.f, 0.f, 0.f};
        {
            const uint4 dh = next_frag(), dl = next_frag();
            cep = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(dl), as_h8(make_uint4(lh[0], lh[1], lh[2], lh[3])), cep, 0, 0, 0);
            cep = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(dh), as_h8(make_uint4(ll4[0], ll4[1], ll4[2], ll4[3])), cep, 0, 0, 0);
            cep = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(dh), as_h8(make_uint4(lh[0], lh[1], lh[2], lh[3])), cep, 0, 0, 0);
        }
        float *dst = out + ((size_t)(gwave * blocks_per_wave + blk) * 16 + (lane & 15)) * 16 + 4 * (lane >> 4);
        *reinterpret_cast<f4 *>(dst) = cep;
        sink += cep[0];
    }
    if (sink == 12345.678f) out[0] = sink;
}

int main(int argc, char **argv) {
    const int bpw = argc > 1 ? atoi(argv[1]) : 64;
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int grid = prop.multiProcessorCount;  // one 512-thread workgroup per CU: 128 KB of LDS
    const size_t waves = (size_t)grid * (WG / 64), blocks = waves * bpw, frames = blocks * 16;
    std::vector<uint4> hp(blocks * 16 * 64);
    unsigned s = 12345;
    for (auto &v : hp) {
        s = s * 1664525u + 1013904223u; v.x = s; s = s * 1664525u + 1013904223u; v.y = s;
        s = s * 1664525u + 1013904223u; v.z = s; s = s * 1664525u + 1013904223u; v.w = s;
    }
    std::vector<uint4> ht((size_t)TAB_FRAGS * 64);
    for (auto &v : ht) {  // fp16 pairs of moderate size: 0x3xxx = 0.25 .. 2
        s = s * 1664525u + 1013904223u; v.x = (s & 0x0fff0fffu) | 0x30003000u; s = s * 1664525u + 1013904223u; v.y = (s & 0x0fff0fffu) | 0x30003000u;
        s = s * 1664525u + 1013904223u; v.z = (s & 0x0fff0fffu) | 0xb0003000u; s = s * 1664525u + 1013904223u; v.w = (s & 0x0fff0fffu) | 0x3000b000u;
    }
    uint4 *dp, *dt; float *dout;
    hipMalloc(&dp, hp.size() * 16); hipMalloc(&dt, ht.size() * 16); hipMalloc(&dout, frames * 16 * 4);
    hipMemcpy(dp, hp.data(), hp.size() * 16, hipMemcpyHostToDevice); hipMemcpy(dt, ht.data(), ht.size() * 16, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)skeleton, hipFuncAttributeMaxDynamicSharedMemorySize, TAB_FRAGS * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL(skeleton, dim3(grid), dim3(WG), TAB_FRAGS * 1024, 0, dp, dt, dout, bpw);
    hipEventRecord(e0);
    const int reps = 10;
    for (int it = 0; it < reps; it++) hipLaunchKernelGGL(skeleton, dim3(grid), dim3(WG), TAB_FRAGS * 1024, 0, dp, dt, dout, bpw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("skeleton: %zu frames per launch, %.3f ms, %.3g frames/s (one 512-thread workgroup per CU, 2 waves per SIMD); headline kernel today: 2.7e9\n",
           frames, ms, frames / (ms * 1e-3));
    printf("cycles per frame and SIMD at 2.1 GHz: %.0f\n", ms * 1e-3 * 2.1e9 * grid * 4 / frames);
    return 0;
}
