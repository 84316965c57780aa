// How much VALU work hides under v_mfma_f32_32x32x16_f16 on gfx950, by WHERE the accumulator lives and HOW the MFMAs depend on each other.
// mfma_own_valu.hip (round 3) found ~2 VALU instructions hidden per MFMA with four independent accumulators in VGPRs.  Hypothesis behind this file:
// the 16-register SrcC read / vDst write-back of every MFMA competes with the VALU for the vector register file's ports; an accumulator in AGPRs
// (the other half of the unified file) or a dependent chain (SrcC forwarded inside the matrix unit) might free them.
//   modes: 0 = 4 independent accumulators in VGPRs (the old benchmark)      1 = ONE accumulator in VGPRs, dependent chain (the predictor's regime)
//          2 = 4 independent accumulators in AGPRs                          3 = ONE accumulator in AGPRs, dependent chain
//   fillers per MFMA: K plain v_fma_f32 (kind 0) or v_exp_f32 (kind 1), independent of each other and of the MFMAs.
// One wave per SIMD (256 workgroups of 256 threads) and, second table, two waves per SIMD (512 threads).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_agpr_valu.hip -o mfma_agpr_valu && ./mfma_agpr_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int KIND, int K>
__device__ __forceinline__ void filler(float (&x)[12]) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(1.0001f), "v"(0.5f));
        else asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    }
}

template <int MODE, int KIND, int K, int THREADS>
__global__ __launch_bounds__(THREADS) void bench(float *out, long long *cyc, int iters) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    float x[12];
    for (int i = 0; i < 12; ++i) x[i] = 0.5f + 0.001f * threadIdx.x + i;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            constexpr bool chain = (MODE & 1) != 0;
            f32x16 &c = acc[chain ? 0 : (u & 3)];
            if constexpr (MODE < 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
            filler<KIND, K>(x);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int i = 0; i < 12; ++i) s += x[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int KIND, int K, int THREADS>
double run(float *out, long long *cyc, int nb) {
    const int iters = 1000;
    bench<MODE, KIND, K, THREADS><<<nb, THREADS>>>(out, cyc, iters);
    bench<MODE, KIND, K, THREADS><<<nb, THREADS>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / nb / (iters * 8.0);
}

template <int MODE, int KIND, int THREADS>
void sweep(const char *name, float *out, long long *cyc, int nb) {
    printf("%-44s K=0 %6.1f  2 %6.1f  4 %6.1f  6 %6.1f  8 %6.1f  12 %6.1f\n", name, run<MODE, KIND, 0, THREADS>(out, cyc, nb), run<MODE, KIND, 2, THREADS>(out, cyc, nb),
           run<MODE, KIND, 4, THREADS>(out, cyc, nb), run<MODE, KIND, 6, THREADS>(out, cyc, nb), run<MODE, KIND, 8, THREADS>(out, cyc, nb), run<MODE, KIND, 12, THREADS>(out, cyc, nb));
}

template <int THREADS>
void table(float *out, long long *cyc, int nb) {
    printf("cycles per MFMA OF ONE WAVE with K fillers behind each, %d wave(s) per SIMD\n", THREADS / 256);
    sweep<0, 0, THREADS>("4 accumulators, VGPR   + v_fma_f32", out, cyc, nb);
    sweep<1, 0, THREADS>("dependent chain, VGPR  + v_fma_f32", out, cyc, nb);
    sweep<2, 0, THREADS>("4 accumulators, AGPR   + v_fma_f32", out, cyc, nb);
    sweep<3, 0, THREADS>("dependent chain, AGPR  + v_fma_f32", out, cyc, nb);
    sweep<0, 1, THREADS>("4 accumulators, VGPR   + v_exp_f32", out, cyc, nb);
    sweep<1, 1, THREADS>("dependent chain, VGPR  + v_exp_f32", out, cyc, nb);
    sweep<2, 1, THREADS>("4 accumulators, AGPR   + v_exp_f32", out, cyc, nb);
    sweep<3, 1, THREADS>("dependent chain, AGPR  + v_exp_f32", out, cyc, nb);
}

int main() {
    const int nb = 256;
    float *out; long long *cyc;
    hipMalloc(&out, nb * 512 * sizeof(float));
    hipMalloc(&cyc, nb * sizeof(long long));
    table<256>(out, cyc, nb);
    table<512>(out, cyc, nb);
    return 0;
}
