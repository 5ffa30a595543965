// Round 4 probe: what write rate the chip sustains for the row-store patterns of trapdct_split16_kernel (no arithmetic):
// 128-row windows (a row = 1472 bytes) written per workgroup of 512 threads in different orders.  hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int D = 368, D4 = 92, ROWS = 128;

// MODE 0: grid-stride contiguous fill of the whole buffer.  1: window per workgroup, eight phases of sixteen whole rows eight apart
// (the staged kernel).  2: window per workgroup, rows in order.  3: eight phases, 64-byte pieces: lane (n, q) of wave w writes band
// w, w + 8, w + 16 of row 8 n + c (the unstaged kernel).  NT: nontemporal stores.  PERSIST: workgroups walk windows grid-stride.
template <int MODE, bool NT>
__global__ __launch_bounds__(512) void k(float *out, long n_win) {
    const int tid = threadIdx.x;
    const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
    auto st = [&](float *p) {
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
        else *reinterpret_cast<f32x4 *>(p) = v;
    };
    if (MODE == 0) {
        const long n4 = n_win * ROWS * D4;
        for (long i = (long)blockIdx.x * 512 + tid; i < n4; i += (long)gridDim.x * 512) st(out + 4 * i);
        return;
    }
    for (long w = blockIdx.x; w < n_win; w += gridDim.x) {
        float *base = out + w * ROWS * D;
        if (MODE == 1) {
            for (int c = 0; c < 8; c++)
                for (int e = tid; e < 16 * D4; e += 512) {
                    const int rw = e / D4, c4 = e - rw * D4;
                    st(base + (long)(8 * rw + c) * D + 4 * c4);
                }
        } else if (MODE == 2) {
            for (int e = tid; e < ROWS * D4; e += 512) st(base + 4 * (long)e);
        } else {
            const int lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
            for (int c = 0; c < 8; c++)
                for (int b = wave; b < 23; b += 8) st(base + (long)(8 * n + c) * D + b * 16 + q * 4);
        }
    }
}

// The staged kernel's shape around the same stores: 57 KB of LDS per workgroup (two per CU), CHAIN dependent global loads ahead of a
// window's first store (chunk table -> utterance record -> centre frame -> tile), two LDS-only barriers per phase.
template <int CHAIN, bool BARRIERS>
__global__ __launch_bounds__(512, 2) void kshape(float *out, const int *chain, long n_win) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x;
    const long w = blockIdx.x;
    int idx = (int)(w & 0xfffff);
#pragma unroll
    for (int i = 0; i < CHAIN; i++) idx = chain[((long)idx * 977 + i * 131071 + tid % 16) & 0xfffffff];  // 1 GiB of ints: misses every cache
    lds[tid] = (float)idx;
    __syncthreads();
    const f32x4 v = {lds[(tid + 1) & 511], 2.f, 3.f, (float)tid};
    float *base = out + w * ROWS * D;
    for (int c = 0; c < 8; c++) {
        if (BARRIERS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        for (int e = tid; e < 16 * D4; e += 512) {
            const int rw = e / D4, c4 = e - rw * D4;
            *reinterpret_cast<f32x4 *>(base + (long)(8 * rw + c) * D + 4 * c4) = v;
        }
        if (BARRIERS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}
template <int CHAIN, bool BARRIERS>
void run_shape(const char *name, float *buf, const int *chain, long n_win) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute((const void *)&kshape<CHAIN, BARRIERS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((kshape<CHAIN, BARRIERS>), dim3((int)n_win), dim3(512), 57 * 1024, 0, buf, chain, n_win);
    CK(hipEventRecord(a));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((kshape<CHAIN, BARRIERS>), dim3((int)n_win), dim3(512), 57 * 1024, 0, buf, chain, n_win);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    printf("%-58s grid %6d  %.3f ms  %.2f TB/s\n", name, (int)n_win, ms, (double)n_win * ROWS * D * 4 / ms * 1e-9);
}

template <int MODE, bool NT>
void run(const char *name, float *buf, long n_win, int grid) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int it = 0; it < 2; it++) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(512), 0, 0, buf, n_win);
    CK(hipEventRecord(a));
    for (int it = 0; it < 3; it++) hipLaunchKernelGGL((k<MODE, NT>), dim3(grid), dim3(512), 0, 0, buf, n_win);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 3;
    const double bytes = (double)n_win * ROWS * D * 4 * ((MODE == 3) ? 1.0 : 1.0);
    printf("%-58s grid %6d  %.3f ms  %.2f TB/s\n", name, grid, ms, bytes / ms * 1e-9);
}

int main() {
    const long n_win = 70400;  // 9.0 M rows: 13.3 GB
    float *buf;
    CK(hipMalloc(&buf, (size_t)n_win * ROWS * D * 4));
    CK(hipMemset(buf, 0, (size_t)n_win * ROWS * D * 4));
    run<0, false>("contiguous grid-stride fill", buf, n_win, 2048);
    run<0, true>("contiguous grid-stride fill, nontemporal", buf, n_win, 2048);
    run<2, false>("window per workgroup, rows in order", buf, n_win, (int)n_win);
    run<1, false>("window per workgroup, 8 phases x 16 whole rows 8 apart", buf, n_win, (int)n_win);
    run<1, true>("  the same, nontemporal", buf, n_win, (int)n_win);
    run<3, false>("window per workgroup, 8 phases x 64-byte band pieces", buf, n_win, (int)n_win);
    run<3, true>("  the same, nontemporal", buf, n_win, (int)n_win);
    run<1, false>("whole rows 8 apart, 512 persistent workgroups", buf, n_win, 512);
    run<1, false>("whole rows 8 apart, 1024 persistent workgroups", buf, n_win, 1024);
    run<3, false>("64-byte pieces, 512 persistent workgroups", buf, n_win, 512);
    run<2, false>("rows in order, 512 persistent workgroups", buf, n_win, 512);
    int *chain;
    CK(hipMalloc(&chain, (size_t)1 << 30));
    CK(hipMemset(chain, 1, (size_t)1 << 30));
    run_shape<0, false>("whole rows 8 apart, 57 KB LDS (2 WG/CU), no loads", buf, chain, n_win);
    run_shape<0, true>("  + two barriers per phase", buf, chain, n_win);
    run_shape<1, true>("  + barriers, 1 dependent load ahead of the stores", buf, chain, n_win);
    run_shape<2, true>("  + barriers, 2 dependent loads", buf, chain, n_win);
    run_shape<4, true>("  + barriers, 4 dependent loads", buf, chain, n_win);
    run_shape<4, false>("  4 dependent loads, no barriers", buf, chain, n_win);
    return 0;
}
