// Round 4 probe: LDS cycles per wave-instruction of the access patterns of trapdct_rows_kernel (which of them meets bank conflicts).
// One workgroup of 512 threads per CU, every wave repeats one access 4096 times; time / (waves x repeats) against a conflict-free read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int p0, int p1) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
    for (int i = tid; i < 16384; i += 512) lds[i] = (float)i;
    __syncthreads();
    int off;  // dword offset of this lane's 16 bytes
    if (MODE == 0) off = 4 * lane;                       // contiguous: conflict-free reference
    else if (MODE == 1) off = n * p0 + 4 * q + p1;       // row stride p0 dwords per n, 16 bytes per q (the R reads)
    else off = n * p0 + 4 * q + p1;                      // the same addresses written (stage writes: n rows, q quarters)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4 *rp = reinterpret_cast<const f32x4 *>(lds + off);
    f32x4 *wp = reinterpret_cast<f32x4 *>(lds + off);
    for (int it = 0; it < 4096; it++) {
        if (MODE == 2) {
            *wp = acc;
            asm volatile("" ::: "memory");
        } else {
            const f32x4 v = *rp;
            asm volatile("" : : "v"(v) : "memory");
            acc += v;
        }
    }
    if (acc[0] == 12345.f) out[tid] = acc[1];
}

template <int MODE>
void run(const char *name, float *out, int p0, int p1, float ref) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 65536, 0, out, p0, p1);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 65536, 0, out, p0, p1);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-64s %.3f ms  x%.2f\n", name, ms, ref > 0 ? ms / ref : 1.f);
}

int main() {
    float *out; CK(hipMalloc(&out, 4096));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<0>), dim3(256), dim3(512), 65536, 0, out, 0, 0);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<0>), dim3(256), dim3(512), 65536, 0, out, 0, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ref; CK(hipEventElapsedTime(&ref, a, b));
    printf("%-64s %.3f ms\n", "ds_read_b128, contiguous lanes (reference)", ref);
    for (int s : {64, 68, 72, 76, 80, 84, 88, 92, 100, 116, 132}) {
        char nm[128]; snprintf(nm, sizeof nm, "ds_read_b128, lane (n, q): row stride %d dwords (%d bytes)", s, 4 * s);
        run<1>(nm, out, s, 0, ref);
    }
    hipLaunchKernelGGL((k<2>), dim3(256), dim3(512), 65536, 0, out, 0, 0);
    float wref;
    {
        CK(hipEventRecord(a));
        // contiguous write reference: stride 0 rows is a broadcast conflict; use n * 16 + 4 q (contiguous 16-byte slots)
        hipLaunchKernelGGL((k<2>), dim3(256), dim3(512), 65536, 0, out, 16, 0);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&wref, a, b));
        printf("%-64s %.3f ms\n", "ds_write_b128, contiguous 16-byte slots (reference)", wref);
    }
    for (int s : {368, 372, 376, 380, 388}) {
        char nm[128]; snprintf(nm, sizeof nm, "ds_write_b128, lane (n, q): row stride %d dwords", s);
        run<2>(nm, out, s, 0, wref);
    }
    return 0;
}
