// VALU issue-rate probe: time per wave64 instruction and SIMD for plain, three-operand, packed, SDWA and DPP forms at 1, 2
// and 4 waves per SIMD (every CU busy).  Reported relative to v_add_f32 at 4 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o tools/probes/valu_rate && tools/probes/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(X) X X X X X X X X
// eight independent destination registers %0..%7; %8, %9 = two more VGPR operands; %10 = an SGPR operand
#define OPS(T, A)                                                                                                        \
    asm volatile(T(0) "\n\t" T(1) "\n\t" T(2) "\n\t" T(3) "\n\t" T(4) "\n\t" T(5) "\n\t" T(6) "\n\t" T(7)                        \
                 : "+v"(A##0), "+v"(A##1), "+v"(A##2), "+v"(A##3), "+v"(A##4), "+v"(A##5), "+v"(A##6), "+v"(A##7)        \
                 : "v"(c), "v"(d), "s"(sc), "v"(c2), "v"(d2), "v"(ic)                                                    \
                 : "vcc", "s20", "s21");

#define T_ADD(n) "v_add_f32 %" #n ", %" #n ", %8"
#define T_ADD64(n) "v_add_f32_e64 %" #n ", %" #n ", %8"
#define T_MUL2(n) "v_mul_f32 %" #n ", %8, %9"
#define T_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %8"
#define T_FMA3(n) "v_fma_f32 %" #n ", %" #n ", %8, %9"
#define T_FMAS(n) "v_fma_f32 %" #n ", %" #n ", %10, %8"
#define T_FMAC(n) "v_fmac_f32 %" #n ", %8, %9"
#define T_FMAMK(n) "v_fmamk_f32 %" #n ", %" #n ", 0x3f800347, %8"
#define T_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %11"
#define T_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %11"
#define T_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %11, %11"
#define T_PKFMA3(n) "v_pk_fma_f32 %" #n ", %" #n ", %11, %12"
#define T_CVT(n) "v_cvt_f32_i32_sdwa %" #n ", sext(%13) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1"
#define T_MOVDPP(n) "v_mov_b32_dpp %" #n ", %8 row_ror:1 row_mask:0xf bank_mask:0xf"
#define T_ADDDPP(n) "v_add_f32_dpp %" #n ", %8, %" #n " row_ror:1 row_mask:0xf bank_mask:0xf"
#define T_CND(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc"
#define T_MOV(n) "v_mov_b32 %" #n ", %8"
#define T_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]"
#define T_BFI(n) "v_bfi_b32 %" #n ", %13, %8, %" #n
#define T_MAX(n) "v_max_f32 %" #n ", %" #n ", %8"
#define T_AND(n) "v_and_b32 %" #n ", %" #n ", %13"
#define T_BFE(n) "v_bfe_i32 %" #n ", %13, 16, 16"
#define T_CVTP(n) "v_cvt_f32_i32 %" #n ", %13"
#define T_CMP(n) "v_cmp_lt_f32 vcc, %" #n ", %8"
#define T_CMP64(n) "v_cmp_lt_f32_e64 s[20:21], %" #n ", %8"
#define T_PERM(n) "v_perm_b32 %" #n ", %13, %" #n ", %13"
#define T_LOG(n) "v_log_f32 %" #n ", %8"
#define T_MIX(n) "v_add_f32 %" #n ", %" #n ", %8\n\tv_add_f32 %" #n ", %" #n ", %9\n\tv_add_f32 %" #n ", %" #n ", %8\n\tv_cndmask_b32 %" #n ", %" #n ", %8, vcc"
#define T_MIX64(n) "v_add_f32 %" #n ", %" #n ", %8\n\tv_add_f32 %" #n ", %" #n ", %9\n\tv_add_f32 %" #n ", %" #n ", %8\n\tv_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]"
#define T_CNDX(n) "v_cndmask_b32 %" #n ", %8, %9, vcc"
#define T_SUBREV(n) "v_subrev_f32 %" #n ", %8, %" #n

template <int KIND>
__global__ void probe(unsigned long long *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a2}, p5 = {a3, a4}, p6 = {a5, a6}, p7 = {a7, a0};
    float c = 1.0001f + out[0], d = 0.9999f + out[0];
    const float sc = __builtin_amdgcn_readfirstlane(__float_as_int(c)) ? 1.0001f : 1.0002f;
    f2 c2 = {c, d}, d2 = {d, c};
    int ic = threadIdx.x * 70001;
    asm volatile("s_mov_b64 s[20:21], 0x5555" ::: "s20", "s21");
    __syncthreads();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0) { REP8(OPS(T_ADD, a)) }
        if (KIND == 1) { REP8(OPS(T_ADD64, a)) }
        if (KIND == 2) { REP8(OPS(T_MUL2, a)) }
        if (KIND == 3) { REP8(OPS(T_FMA, a)) }
        if (KIND == 4) { REP8(OPS(T_FMA3, a)) }
        if (KIND == 5) { REP8(OPS(T_FMAS, a)) }
        if (KIND == 6) { REP8(OPS(T_FMAC, a)) }
        if (KIND == 7) { REP8(OPS(T_FMAMK, a)) }
        if (KIND == 8) { REP8(OPS(T_PKADD, p)) }
        if (KIND == 9) { REP8(OPS(T_PKMUL, p)) }
        if (KIND == 10) { REP8(OPS(T_PKFMA, p)) }
        if (KIND == 11) { REP8(OPS(T_PKFMA3, p)) }
        if (KIND == 12) { REP8(OPS(T_CVT, a)) }
        if (KIND == 13) { REP8(OPS(T_MOVDPP, a)) }
        if (KIND == 14) { REP8(OPS(T_ADDDPP, a)) }
        if (KIND == 15) { REP8(OPS(T_CND, a)) }
        if (KIND == 16) { REP8(OPS(T_MOV, a)) }
        if (KIND == 17) { REP8(OPS(T_SUBREV, a)) }
        if (KIND == 28) { REP8(OPS(T_MIX, a)) }
        if (KIND == 29) { REP8(OPS(T_MIX64, a)) }
        if (KIND == 30) { REP8(OPS(T_CNDX, a)) }
        if (KIND == 18) { REP8(OPS(T_CND64, a)) }
        if (KIND == 19) { REP8(OPS(T_BFI, a)) }
        if (KIND == 20) { REP8(OPS(T_MAX, a)) }
        if (KIND == 21) { REP8(OPS(T_AND, a)) }
        if (KIND == 22) { REP8(OPS(T_BFE, a)) }
        if (KIND == 23) { REP8(OPS(T_CVTP, a)) }
        if (KIND == 24) { REP8(OPS(T_CMP, a)) }
        if (KIND == 25) { REP8(OPS(T_CMP64, a)) }
        if (KIND == 26) { REP8(OPS(T_PERM, a)) }
        if (KIND == 27) { REP8(OPS(T_LOG, a)) }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (s == 12345.678f) out[1] = 1;  // keep the values alive
}

static double base_ns = 0;
template <int KIND>
void run(const char *name, unsigned long long *d, int ncu) {
    const int iters = 4000;
    printf("%-44s", name);
    for (int wps : {1, 2, 4}) {  // waves per SIMD
        const int threads = 256 * wps;
        hipLaunchKernelGGL(probe<KIND>, dim3(ncu), dim3(threads), 0, 0, d, iters);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0);
        (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe<KIND>, dim3(ncu), dim3(threads), 0, 0, d, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns = ms * 1e6 / ((double)iters * 64 * wps);  // per wave-instruction and SIMD
        if (KIND == 0 && wps == 4) base_ns = ns;
        printf("  %dw: %.3f ns", wps, ns);
        if (wps == 4 && base_ns > 0) printf("  (x%.2f of v_add_f32)", ns / base_ns);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    unsigned long long *d;
    (void)hipMalloc(&d, 64);
    (void)hipMemset(d, 0, 64);
    run<0>("v_add_f32 d,d,c (VOP2)", d, ncu);
    run<0>("v_add_f32 d,d,c (VOP2) again", d, ncu);
    run<1>("v_add_f32_e64 d,d,c (VOP3 encoding)", d, ncu);
    run<2>("v_mul_f32 d,c,e", d, ncu);
    run<17>("v_subrev_f32 d,c,d", d, ncu);
    run<3>("v_fma_f32 d,d,c,c", d, ncu);
    run<4>("v_fma_f32 d,d,c,e (3 distinct VGPRs)", d, ncu);
    run<5>("v_fma_f32 d,d,s,c (SGPR multiplier)", d, ncu);
    run<6>("v_fmac_f32 d,c,e (VOP2)", d, ncu);
    run<7>("v_fmamk_f32 d,d,K,c", d, ncu);
    run<8>("v_pk_add_f32", d, ncu);
    run<9>("v_pk_mul_f32", d, ncu);
    run<10>("v_pk_fma_f32 d,d,c,c", d, ncu);
    run<11>("v_pk_fma_f32 d,d,c,e", d, ncu);
    run<12>("v_cvt_f32_i32_sdwa", d, ncu);
    run<13>("v_mov_b32_dpp row_ror", d, ncu);
    run<14>("v_add_f32_dpp row_ror", d, ncu);
    run<15>("v_cndmask_b32 vcc", d, ncu);
    run<16>("v_mov_b32", d, ncu);
    run<28>("3 v_add + 1 v_cndmask vcc (per 4 instr)", d, ncu);
    run<29>("3 v_add + 1 v_cndmask_e64 (per 4 instr)", d, ncu);
    run<30>("v_cndmask_b32 d,c,e,vcc (dst not a source)", d, ncu);
    run<18>("v_cndmask_b32_e64 (SGPR-pair mask)", d, ncu);
    run<19>("v_bfi_b32", d, ncu);
    run<20>("v_max_f32", d, ncu);
    run<21>("v_and_b32", d, ncu);
    run<22>("v_bfe_i32", d, ncu);
    run<23>("v_cvt_f32_i32", d, ncu);
    run<24>("v_cmp_lt_f32 vcc", d, ncu);
    run<25>("v_cmp_lt_f32_e64 sgpr", d, ncu);
    run<26>("v_perm_b32", d, ncu);
    run<27>("v_log_f32", d, ncu);
    return 0;
}
