// v_mfma_f32_4x4x1_16b_f32 (sixteen independent 4x4 outer products per instruction):
//  (1) operand and result lane map, found by setting one A lane and one B lane at a time;
//  (2) is D = fma(a, b, c) with one rounding (a*b + c for a = 1 + 2^-12, b = 1 - 2^-12, c = -1: fused gives -2^-24);
//  (3) what a dependent chain of them costs per instruction with LDS operand reads beside it, at 1 and 4 waves per SIMD,
//      and how many plain VALU instructions of another wave issue meanwhile.
// hipcc --offload-arch=gfx950 -O2 tools/probes/mfma4x4_probe.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void one(const float *A, const float *B, const float *C, float *D) {
    const int lane = threadIdx.x;
    f4 c = {C[lane * 4], C[lane * 4 + 1], C[lane * 4 + 2], C[lane * 4 + 3]};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(A[lane], B[lane], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[lane * 4 + i] = c[i];
}

// `steps` dependent MFMAs per accumulator, two accumulators, operands from LDS (one A read, two B reads per step)
__global__ __launch_bounds__(512) void chain(float *out, int steps, int reps, int mode) {
    __shared__ float tab[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) tab[i] = 1.0f / (1 + (i & 255));
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float v = lane;
    const float *w = tab + (lane >> 3) * 4 + (lane & 3), *pb = tab + 2048 + (lane & 7) * 260 + (lane >> 3) * 5;
    for (int r = 0; r < reps; r++) {
        if (mode == 0 || (mode == 2 && (wave & 1) == 0)) {
            // software pipeline: the operands of the next eight steps are in flight while this group's MFMAs issue
            float a[8], b0[8], b1[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { a[u] = w[u * 32]; b0[u] = pb[u]; b1[u] = pb[u + 1040]; }
            for (int s = 0; s < steps; s += 8) {
                float na[8], nb0[8], nb1[8];
                const int sn = (s + 8 < steps) ? s + 8 : 0;
#pragma unroll
                for (int u = 0; u < 8; u++) { na[u] = w[(sn + u) * 32]; nb0[u] = pb[sn + u]; nb1[u] = pb[sn + u + 1040]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    a0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u], b0[u], a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u], b1[u], a1, 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 8; u++) { a[u] = na[u]; b0[u] = nb0[u]; b1[u] = nb1[u]; }
            }
        } else {
#pragma unroll 8
            for (int s = 0; s < steps * 4; s++) v = v * 1.0001f + 0.5f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a0[1] + a0[2] + a0[3] + a1[0] + a1[1] + a1[2] + a1[3] + v;
}


// register operands only: NACC independent accumulators of the 4x4x1 form (KIND 0) or the 16x16x4 form (KIND 1);
// odd waves run a dependent FMA chain instead when mix != 0
template <int KIND, int NACC>
__global__ __launch_bounds__(512) void regchain(float *out, int reps, int mix) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = f4{0, 0, 0, 0};
    float a = 1.0f + lane * 1e-3f, b = 0.5f + lane * 1e-4f, v = lane;
    if (mix && (wave & 1)) {
        for (int r = 0; r < reps; r++)
#pragma unroll
            for (int s = 0; s < 64; s++) v = v * 1.0001f + 0.5f;
    } else {
        for (int r = 0; r < reps; r++)
#pragma unroll
            for (int s = 0; s < 16; s++)
#pragma unroll
                for (int i = 0; i < NACC; i++)
                    acc[i] = KIND == 0 ? __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float t = v;
    for (int i = 0; i < NACC; i++) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
    std::vector<float> A(64, 0.f), B(64, 0.f), C(256, 0.f), D(256);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    auto run = [&]() {
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(one, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    };
    printf("(1) A lane la = 1, B lane lb = 1 -> non-zero D at (lane, reg):\n");
    int la_lb[][2] = {{0, 0}, {1, 0}, {0, 1}, {2, 3}, {3, 2}, {4, 4}, {5, 6}, {0, 4}, {21, 22}, {63, 60}};
    for (auto &t : la_lb) {
        for (auto &x : A) x = 0; for (auto &x : B) x = 0;
        A[t[0]] = 1; B[t[1]] = 1; run();
        printf("  la %2d lb %2d:", t[0], t[1]);
        for (int i = 0; i < 256; i++) if (D[i] != 0) printf(" (lane %d, reg %d)", i / 4, i % 4);
        printf("\n");
    }
    for (auto &x : A) x = 0; for (auto &x : B) x = 0; for (auto &x : C) x = 0;
    A[0] = 1.f + 1.f / 4096; B[0] = 1.f - 1.f / 4096; C[0] = -1.f; run();
    printf("(2) (1+2^-12)(1-2^-12) - 1 = %g   (fused: %g; product rounded first: 0)\n", D[0], -1.0 / 16777216);
    float *dout; hipMalloc(&dout, 4096 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    const double clk = pr.clockRate * 1e3;
    printf("(3) clock %.2f GHz (nominal), %d CUs\n", clk / 1e9, pr.multiProcessorCount);
    for (int mode = 0; mode < 3; mode++)
        for (int wg : {64, 256, 512}) {
            const int steps = 64, reps = 2000, blocks = pr.multiProcessorCount * (wg == 512 ? 2 : 1);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(wg), 0, 0, dout, steps, 10, mode);
            hipEventRecord(e0);
            hipLaunchKernelGGL(chain, dim3(blocks), dim3(wg), 0, 0, dout, steps, reps, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const int waves_simd = wg == 64 ? 1 : (wg == 256 ? 1 : 4);  // 64: one wave on one SIMD; 256: one per SIMD; 512 x 2: four per SIMD
            const double per_step = ms * 1e-3 * clk / ((double)steps * reps);
            printf("  mode %d (%s) wg %3d (%d wave(s)/SIMD): %.3f ms, %.1f cycles per step of one wave (step = 2 MFMA + 3 ds_read_b32, or 4 FMA)\n", mode,
                   mode == 0 ? "all waves MFMA chain" : mode == 1 ? "all waves FMA chain" : "even waves MFMA, odd waves FMA", wg, waves_simd, ms, per_step);
        }
    printf("(4) register operands; cycles (nominal) per MFMA and SIMD:\n");
    auto timeit = [&](auto kern, int wg, int reps, int mix, int per_rep, const char *name) {
        const int blocks = pr.multiProcessorCount * (wg == 512 ? 2 : 1);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(wg), 0, 0, dout, 10, mix);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(wg), 0, 0, dout, reps, mix);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int wps = wg == 512 ? 4 : 1;
        const int mf_waves = mix ? wps / 2 : wps;
        printf("  %-28s wg %3d mix %d: %.3f ms = %.1f cycles per MFMA of one wave, %.1f per MFMA and SIMD\n", name, wg, mix, ms,
               ms * 1e-3 * clk / ((double)reps * per_rep), mf_waves ? ms * 1e-3 * clk / ((double)reps * per_rep * mf_waves) : 0.0);
    };
    for (int wg : {256, 512}) {
        timeit(regchain<0, 1>, wg, 4000, 0, 16, "4x4x1, 1 accumulator");
        timeit(regchain<0, 2>, wg, 4000, 0, 32, "4x4x1, 2 accumulators");
        timeit(regchain<0, 4>, wg, 4000, 0, 64, "4x4x1, 4 accumulators");
        timeit(regchain<1, 1>, wg, 4000, 0, 16, "16x16x4, 1 accumulator");
        timeit(regchain<1, 2>, wg, 4000, 0, 32, "16x16x4, 2 accumulators");
        timeit(regchain<1, 4>, wg, 4000, 0, 64, "16x16x4, 4 accumulators");
    }
    timeit(regchain<0, 4>, 512, 4000, 1, 64, "4x4x1 x4 | FMA on odd waves");
    timeit(regchain<1, 4>, 512, 4000, 1, 64, "16x16x4 x4 | FMA on odd waves");
    {   // the FMA waves alone: 2 waves per SIMD, 64 FMAs per rep
        hipEventRecord(e0);
        hipLaunchKernelGGL((regchain<0, 1>), dim3(pr.multiProcessorCount * 2), dim3(512), 0, 0, dout, 0, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    return 0;
}
