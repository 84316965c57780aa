// VALU issue rate on gfx950 by instruction kind and waves per SIMD.
// Every wave runs a loop of 32 INDEPENDENT instructions of one kind (inline asm, so the compiler cannot
// fuse, pack or drop them); workgroups of 256 threads = one wave per SIMD, W workgroups per CU = W waves per SIMD.
// Prints SIMD cycles per wave-instruction = (wave cycles) / (instructions per wave x W): what one more VALU
// instruction costs a SIMD that is issue-bound.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

enum { K_FMA, K_MUL, K_ADD, K_PKFMA, K_PKMUL, K_PKADD, K_CNDMASK, K_CMP, K_RCP, K_SQRT, K_EXP, K_MOV, K_ADDU32, K_LSHLADD64,
       K_MINU32, K_DIVSCALE, K_DIVFIXUP, K_FMAC, K_MAX3,
       K_CND64, K_CMPCND, K_MINF, K_MAXF, K_MED3, K_RSQ, K_DIVFMAS, K_AND, K_LSHL, K_RNDNE, K_CVTI, K_LDEXP, K_CMPS, K_CNDLIT, K_FMAAK, K_SUB, K_MULLO, K_COUNT };
static const char *kNames[K_COUNT] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_cndmask_b32",
                                      "v_cmp_lt_f32(vcc)", "v_rcp_f32", "v_sqrt_f32", "v_exp_f32", "v_mov_b32", "v_add_u32", "v_lshl_add_u64",
                                      "v_min_u32", "v_div_scale_f32", "v_div_fixup_f32", "v_fmac_f32", "v_max3_f32",
                                      "v_cndmask_e64(sgpr)", "v_cmp+v_cndmask", "v_min_f32", "v_max_f32", "v_med3_f32", "v_rsq_f32", "v_div_fmas_f32", "v_and_b32", "v_lshlrev_b32", "v_rndne_f32", "v_cvt_i32_f32", "v_ldexp_f32", "v_cmp_e64(sgpr)", "v_cndmask(const,vcc)", "v_fmaak_f32", "v_sub_f32", "v_mul_lo_u32"};

// 32 instructions over 8 rotating destination registers: each depends only on the one 8 instructions earlier
#define ROT8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define ROT32(op) ROT8(op) ROT8(op) ROT8(op) ROT8(op)
#define R8(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
#define OP_FMA(i) "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_MUL(i) "v_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_ADD(i) "v_add_f32 %" #i ", %9, %" #i "\n"
#define OP_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define OP_PKFMA(i) "v_pk_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define OP_PKMUL(i) "v_pk_mul_f32 %" #i ", %8, %" #i "\n"
#define OP_PKADD(i) "v_pk_add_f32 %" #i ", %9, %" #i "\n"
#define OP_CND(i) "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define OP_CMP(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define OP_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define OP_ADDU(i) "v_add_u32 %" #i ", %8, %" #i "\n"
#define OP_LSHLADD(i) "v_lshl_add_u64 %" #i ", %" #i ", 1, %8\n"
#define OP_MINU(i) "v_min_u32 %" #i ", %8, %" #i "\n"
#define OP_DIVSCALE(i) "v_div_scale_f32 %" #i ", vcc, %8, %8, %" #i "\n"
#define OP_DIVFIXUP(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_MAX3(i) "v_max3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_CND64(i) "v_cndmask_b32_e64 %" #i ", %8, %" #i ", s[20:21]\n"
#define OP_CMPCND(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define OP_MINF(i) "v_min_f32 %" #i ", %8, %" #i "\n"
#define OP_MAXF(i) "v_max_f32 %" #i ", %8, %" #i "\n"
#define OP_MED3(i) "v_med3_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_RSQ(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define OP_DIVFMAS(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_AND(i) "v_and_b32 %" #i ", %8, %" #i "\n"
#define OP_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define OP_RNDNE(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define OP_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define OP_LDEXP(i) "v_ldexp_f32 %" #i ", %" #i ", %8\n"
#define OP_CMPS(i) "v_cmp_lt_f32_e64 s[20:21], %" #i ", %8\n"
#define OP_CNDLIT(i) "v_cndmask_b32 %" #i ", 1.0, %" #i ", vcc\n"
#define OP_FMAAK(i) "v_fmaak_f32 %" #i ", %" #i ", %8, 0x3e000000\n"
#define OP_SUB(i) "v_sub_f32 %" #i ", %" #i ", %9\n"
#define OP_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"

template <int KIND>
__global__ __launch_bounds__(256) void bench(float *out, long long *cyc, int iters) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a[8], b = 0.999f, c = 1e-3f;
    f2 pa[8], pb = {b, b}, pc = {c, c};
    unsigned u[8], ub = 7;
    unsigned long long w[8], wb = 5;
    for (int i = 0; i < 8; ++i) { a[i] = 1.0f + 1e-6f * threadIdx.x + i; pa[i] = (f2){a[i], a[i]}; u[i] = threadIdx.x + i; w[i] = threadIdx.x + i; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA) asm volatile(ROT32(OP_FMA) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_MUL) asm volatile(ROT32(OP_MUL) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_ADD) asm volatile(ROT32(OP_ADD) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_FMAC) asm volatile(ROT32(OP_FMAC) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_PKFMA) asm volatile(ROT32(OP_PKFMA) : R8(pa) : "v"(pb), "v"(pc));
        if constexpr (KIND == K_PKMUL) asm volatile(ROT32(OP_PKMUL) : R8(pa) : "v"(pb), "v"(pc));
        if constexpr (KIND == K_PKADD) asm volatile(ROT32(OP_PKADD) : R8(pa) : "v"(pb), "v"(pc));
        if constexpr (KIND == K_CNDMASK) asm volatile(ROT32(OP_CND) : R8(a) : "v"(b), "v"(c) : "vcc");
        if constexpr (KIND == K_CMP) asm volatile(ROT32(OP_CMP) : R8(a) : "v"(b), "v"(c) : "vcc");
        if constexpr (KIND == K_RCP) asm volatile(ROT32(OP_RCP) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_SQRT) asm volatile(ROT32(OP_SQRT) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_EXP) asm volatile(ROT32(OP_EXP) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_MOV) asm volatile(ROT32(OP_MOV) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_ADDU32) asm volatile(ROT32(OP_ADDU) : R8(u) : "v"(ub), "v"(c));
        if constexpr (KIND == K_LSHLADD64) asm volatile(ROT32(OP_LSHLADD) : R8(w) : "v"(wb), "v"(c));
        if constexpr (KIND == K_MINU32) asm volatile(ROT32(OP_MINU) : R8(u) : "v"(ub), "v"(c));
        if constexpr (KIND == K_DIVSCALE) asm volatile(ROT32(OP_DIVSCALE) : R8(a) : "v"(b), "v"(c) : "vcc");
        if constexpr (KIND == K_DIVFIXUP) asm volatile(ROT32(OP_DIVFIXUP) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_MAX3) asm volatile(ROT32(OP_MAX3) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_CND64) asm volatile(ROT32(OP_CND64) : R8(a) : "v"(b), "v"(c) : "s20", "s21");
        if constexpr (KIND == K_CMPCND) asm volatile(ROT8(OP_CMPCND) ROT8(OP_CMPCND) : R8(a) : "v"(b), "v"(c) : "vcc");   // 16 pairs = 32 instructions
        if constexpr (KIND == K_MINF) asm volatile(ROT32(OP_MINF) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_MAXF) asm volatile(ROT32(OP_MAXF) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_MED3) asm volatile(ROT32(OP_MED3) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_RSQ) asm volatile(ROT32(OP_RSQ) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_DIVFMAS) asm volatile(ROT32(OP_DIVFMAS) : R8(a) : "v"(b), "v"(c) : "vcc");
        if constexpr (KIND == K_AND) asm volatile(ROT32(OP_AND) : R8(u) : "v"(ub), "v"(c));
        if constexpr (KIND == K_LSHL) asm volatile(ROT32(OP_LSHL) : R8(u) : "v"(ub), "v"(c));
        if constexpr (KIND == K_RNDNE) asm volatile(ROT32(OP_RNDNE) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_CVTI) asm volatile(ROT32(OP_CVTI) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_LDEXP) asm volatile(ROT32(OP_LDEXP) : R8(a) : "v"(ub), "v"(c));
        if constexpr (KIND == K_CMPS) asm volatile(ROT32(OP_CMPS) : R8(a) : "v"(b), "v"(c) : "s20", "s21");
        if constexpr (KIND == K_CNDLIT) asm volatile(ROT32(OP_CNDLIT) : R8(a) : "v"(b), "v"(c) : "vcc");
        if constexpr (KIND == K_FMAAK) asm volatile(ROT32(OP_FMAAK) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_SUB) asm volatile(ROT32(OP_SUB) : R8(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_MULLO) asm volatile(ROT32(OP_MULLO) : R8(u) : "v"(ub), "v"(c));
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += a[i] + pa[i][0] + pa[i][1] + (float)u[i] + (float)w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// the same with a DEPENDENT chain (every instruction reads the previous result): latency per instruction
template <int KIND>
__global__ __launch_bounds__(256) void bench_dep(float *out, long long *cyc, int iters) {
    float a = 1.0f + 1e-6f * threadIdx.x, b = 0.999f, c = 1e-3f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == K_FMA) asm volatile(REP32("v_fma_f32 %0, %0, %1, %2\n") : "+v"(a) : "v"(b), "v"(c));
        if constexpr (KIND == K_RCP) asm volatile(REP32("v_rcp_f32 %0, %0\n") : "+v"(a));
    }
    long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = a;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(float *out, long long *cyc) {
    const int iters = 400;
    printf("%-20s", kNames[KIND]);
    for (int W : {1, 2, 4, 8}) {
        const int nb = 256 * W;
        bench<KIND><<<nb, 256>>>(out, cyc, iters);
        bench<KIND><<<nb, 256>>>(out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(nb * 4);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[h.size() / 2];
        printf("  W=%d: %.2f", W, med / (iters * 32.0) / W);
    }
    printf("   (SIMD cycles per wave-instruction, median wave)\n");
}

int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 2048 * 256 * sizeof(float));
    hipMalloc(&cyc, 2048 * 4 * sizeof(long long));
    run<K_FMA>(out, cyc); run<K_MUL>(out, cyc); run<K_ADD>(out, cyc); run<K_FMAC>(out, cyc);
    run<K_PKFMA>(out, cyc); run<K_PKMUL>(out, cyc); run<K_PKADD>(out, cyc);
    run<K_CNDMASK>(out, cyc); run<K_CMP>(out, cyc); run<K_MOV>(out, cyc); run<K_ADDU32>(out, cyc); run<K_LSHLADD64>(out, cyc);
    run<K_MINU32>(out, cyc); run<K_MAX3>(out, cyc);
    run<K_CND64>(out, cyc); run<K_CMPCND>(out, cyc); run<K_CNDLIT>(out, cyc); run<K_CMPS>(out, cyc); run<K_MINF>(out, cyc); run<K_MAXF>(out, cyc); run<K_MED3>(out, cyc);
    run<K_AND>(out, cyc); run<K_LSHL>(out, cyc); run<K_RNDNE>(out, cyc); run<K_CVTI>(out, cyc); run<K_LDEXP>(out, cyc); run<K_FMAAK>(out, cyc); run<K_SUB>(out, cyc); run<K_MULLO>(out, cyc);
    run<K_RSQ>(out, cyc); run<K_DIVFMAS>(out, cyc);
    run<K_RCP>(out, cyc); run<K_SQRT>(out, cyc); run<K_EXP>(out, cyc); run<K_DIVSCALE>(out, cyc); run<K_DIVFIXUP>(out, cyc);
    const int iters = 400;
    for (int W : {1, 4}) {
        bench_dep<K_FMA><<<256 * W, 256>>>(out, cyc, iters);
        hipDeviceSynchronize();
        std::vector<long long> h(256 * W * 4);
        hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("dependent v_fma_f32 chain, W=%d: %.2f cycles per instruction per wave\n", W, (double)h[h.size() / 2] / (iters * 32.0));
    }
    return 0;
}
