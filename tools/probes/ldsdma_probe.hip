// Probe: semantics of __builtin_amdgcn_global_load_lds (size 16) on gfx950 -- where does lane i's data land?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gvoid;
typedef __attribute__((address_space(3))) void lvoid;
__global__ void k(const int *src, int *out) {
    extern __shared__ __align__(16) int lds[];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) lds[i] = -1;
    __syncthreads();
    // every lane fetches 16 bytes from src + 4*perm(lane) ints (a non-identity pattern), LDS base = lds + 8 ints
    const int *g = src + 4 * ((lane * 7) % 64);
    __builtin_amdgcn_global_load_lds((gvoid *)g, (lvoid *)(lds + 8), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<int> h(256), o(1024);
    for (int i = 0; i < 256; i++) h[i] = i;
    int *ds, *dout;
    hipMalloc(&ds, 1024); hipMalloc(&dout, 4096);
    hipMemcpy(ds, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, ds, dout);
    hipMemcpy(o.data(), dout, 4096, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 4; j++) {
            int expect = 4 * ((lane * 7) % 64) + j;
            if (o[8 + 4 * lane + j] != expect) { ok = 0; printf("lane %d j %d got %d expect %d\n", lane, j, o[8 + 4 * lane + j], expect); break; }
        }
    printf("first ints: "); for (int i = 0; i < 24; i++) printf("%d ", o[i]); printf("\n");
    printf(ok ? "LDSDMA_OK base+16*lane\n" : "LDSDMA_MISMATCH\n");
    return 0;
}
