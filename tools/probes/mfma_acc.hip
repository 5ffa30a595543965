// What does v_mfma_f32_16x16x32_f16 do inside one instruction?  (1) fp16 subnormal inputs: kept or flushed; (2) where the
// K = 32 sum is rounded to fp32: a (2^24, 1, -2^24) triple placed at k positions (i, j, l) comes back as 1 only if no fp32
// rounding happens between the three products; (3) whether the C input is added before or after the products.
// hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_acc.hip -o /tmp/mfma_acc && /tmp/mfma_acc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
    // A[16][32] row-major, B[32][16] (k-major), C/D [16][16]
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    h8 a, b;
    for (int j = 0; j < 8; j++) {
        a[j] = A[r * 32 + 8 * q + j];
        b[j] = B[(8 * q + j) * 16 + r];
    }
    f4 c;
    for (int i = 0; i < 4; i++) c[i] = C[(4 * q + i) * 16 + r];
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[(4 * q + i) * 16 + r] = c[i];
}

int main() {
    std::vector<_Float16> A(16 * 32), B(32 * 16);
    std::vector<float> C(256), D(256);
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    auto run = [&]() {
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    };
    auto clear = [&]() { for (auto &v : A) v = 0; for (auto &v : B) v = 0; for (auto &v : C) v = 0; };
    // (1) subnormals: a = 2^-24 (smallest subnormal) times b = 2^14, expect 2^-10
    clear();
    { uint16_t bits = 1; _Float16 sub; memcpy(&sub, &bits, 2); A[0] = sub; B[0] = (_Float16)16384.f; A[32 + 1] = (_Float16)16384.f; uint16_t b2 = 0x00ff; memcpy(&sub, &b2, 2); B[1 * 16 + 1] = sub; }
    run();
    printf("subnormal A: D[0][0] = %g (kept: %g)   subnormal B (255 * 2^-24 * 2^14): D[1][1] = %g (kept: %g)\n", D[0], ldexp(1.0, -10), D[17], 255 * ldexp(1.0, -10));
    // (2) rounding points inside K
    printf("triple (2^24 at i, 1 at j, -2^24 at l): result 1 = exact across the three, 0 = an fp32 rounding in between\n");
    int pos[][3] = {{0, 1, 2}, {0, 1, 3}, {0, 3, 4}, {0, 1, 4}, {0, 4, 7}, {0, 1, 7}, {0, 1, 8}, {0, 7, 8}, {0, 8, 15}, {0, 8, 16}, {0, 15, 16}, {0, 1, 16}, {0, 16, 31}, {0, 1, 31}, {1, 0, 2}, {8, 0, 16}, {16, 0, 24}, {31, 30, 0}, {4, 0, 5}, {2, 3, 1}};
    for (auto &p : pos) {
        clear();
        A[p[0]] = (_Float16)4096.f; B[p[0] * 16] = (_Float16)4096.f;
        A[p[1]] = (_Float16)1.f; B[p[1] * 16] = (_Float16)1.f;
        A[p[2]] = (_Float16)-4096.f; B[p[2] * 16] = (_Float16)4096.f;
        run();
        printf("  k = (%2d, %2d, %2d): %g\n", p[0], p[1], p[2], D[0]);
    }
    // (3) C = 2^24 with products 1 and -2^24 ; C = 1 with 2^24, -2^24
    clear(); C[0] = 16777216.f; A[0] = (_Float16)1.f; B[0] = (_Float16)1.f; A[5] = (_Float16)-4096.f; B[5 * 16] = (_Float16)4096.f; run();
    printf("C = 2^24, products 1 (k=0) and -2^24 (k=5): %g (1 = C joins an exact sum, 0 = rounded)\n", D[0]);
    clear(); C[0] = 1.f; A[0] = (_Float16)4096.f; B[0] = (_Float16)4096.f; A[5] = (_Float16)-4096.f; B[5 * 16] = (_Float16)4096.f; run();
    printf("C = 1, products 2^24 (k=0) and -2^24 (k=5): %g\n", D[0]);
    // (4) many small: 32 products of (1 + 2^-10)^2: exactness of product and sum
    clear(); for (int kk = 0; kk < 32; kk++) { A[kk] = (_Float16)(1.f + 1.f / 1024); B[kk * 16] = (_Float16)(1.f + 1.f / 1024); } run();
    printf("32 x (1+2^-10)^2 = %.10f (exact %.10f)\n", D[0], 32 * (1 + 1.0 / 1024) * (1 + 1.0 / 1024));
    // (5) products 1 + tiny: 2^12 * 2^12 = 2^24 at k=0 and 2^-12 * 2^-12 = 2^-24 at k=1: alignment loss inside
    clear(); A[0] = (_Float16)1.f; B[0] = (_Float16)1.f; A[1] = (_Float16)(1.f / 4096); B[16] = (_Float16)(1.f / 4096); A[2] = (_Float16)-1.f; B[32] = (_Float16)1.f; run();
    printf("1 + 2^-24 - 1 = %g (2^-24 = %g)\n", D[0], ldexp(1.0, -24));
    clear(); A[0] = (_Float16)1.f; B[0] = (_Float16)1.f; A[1] = (_Float16)(1.f / 4096); B[16] = (_Float16)(1.f / 65536.f * 4); A[2] = (_Float16)-1.f; B[32] = (_Float16)1.f; run();
    printf("1 + 2^-26 - 1 = %g (2^-26 = %g)\n", D[0], ldexp(1.0, -26));
    clear(); A[0] = (_Float16)1.f; B[0] = (_Float16)1.f; A[1] = (_Float16)(1.f / 4096); B[16] = (_Float16)(1.f / 65536.f / 64); A[2] = (_Float16)-1.f; B[32] = (_Float16)1.f; run();
    printf("1 + 2^-34 - 1 = %g (2^-34 = %g)\n", D[0], ldexp(1.0, -34));
    return 0;
}
