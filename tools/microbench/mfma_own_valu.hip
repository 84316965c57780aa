// One wave per SIMD: a stream of v_mfma_f32_32x32x16_f16 (4 independent accumulators, operands in
// registers) with K independent VALU instructions of one kind issued behind every MFMA by the SAME wave.
// Prints cycles per MFMA for K = 0..12 and each kind: how much VALU work hides under the matrix pipe.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_own_valu.hip -o mfma_own_valu && ./mfma_own_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int K>
__device__ __forceinline__ void filler(float (&x)[12], f32x2 (&y)[6]) {
#pragma unroll
    for (int i = 0; i < K; ++i) {
        if constexpr (KIND == 0) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
        else if constexpr (KIND == 1) x[i] = __builtin_amdgcn_exp2f(x[i]);
        else if constexpr (KIND == 2) x[i] = __builtin_amdgcn_rcpf(x[i]);
        else y[i % 6] = y[i % 6] * (f32x2){1.0001f, 0.9999f} + (f32x2){0.5f, 0.25f};   // v_pk_fma_f32
    }
}

template <int KIND, int K>
__global__ __launch_bounds__(256) void bench(float *out, long long *cyc, int iters) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * i); }
    float x[12];
    f32x2 y[6];
    for (int i = 0; i < 12; ++i) x[i] = 0.5f + 0.001f * threadIdx.x + i;
    for (int i = 0; i < 6; ++i) y[i] = (f32x2){0.1f * i, 0.2f + threadIdx.x};
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 3], 0, 0, 0);
            filler<KIND, K>(x, y);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    for (int i = 0; i < 12; ++i) s += x[i];
    for (int i = 0; i < 6; ++i) s += y[i][0] + y[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int K>
double run(float *out, long long *cyc, int nb) {
    const int iters = 2000;
    bench<KIND, K><<<nb, 256>>>(out, cyc, iters);
    bench<KIND, K><<<nb, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    return s / nb / (iters * 8.0);
}

template <int KIND>
void sweep(const char *name, float *out, long long *cyc, int nb) {
    printf("%-12s cycles per MFMA with K fillers: K=0 %.1f  2 %.1f  4 %.1f  6 %.1f  8 %.1f  12 %.1f\n", name,
           run<KIND, 0>(out, cyc, nb), run<KIND, 2>(out, cyc, nb), run<KIND, 4>(out, cyc, nb), run<KIND, 6>(out, cyc, nb),
           run<KIND, 8>(out, cyc, nb), run<KIND, 12>(out, cyc, nb));
}

int main() {
    const int nb = 256;                     // one 4-wave workgroup per CU: one wave per SIMD
    float *out; long long *cyc;
    hipMalloc(&out, nb * 256 * sizeof(float));
    hipMalloc(&cyc, nb * sizeof(long long));
    sweep<0>("v_fma_f32", out, cyc, nb);
    sweep<1>("v_exp_f32", out, cyc, nb);
    sweep<2>("v_rcp_f32", out, cyc, nb);
    sweep<3>("v_pk_fma_f32", out, cyc, nb);
    return 0;
}
