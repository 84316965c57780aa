// microbenchmark 2: one LDS-fed MFMA stream per wave with an explicit operand ring of depth D
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
template <int D, int N, int WIDE>
struct R {
    template <int n> static __device__ __forceinline__ void step(f32x16 (&acc)[4], half8 (&a)[D + 4], const uint4 *aw, const half8 &b) {
        acc[n % 4] = MFMA(a[n % (D + 4)], b, acc[n % 4]);
        if constexpr (n + D < N) a[(n + D) % (D + 4)] = *reinterpret_cast<const half8 *>(aw + (n + D) * 64);
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int... Ns> static __device__ __forceinline__ void run(f32x16 (&acc)[4], half8 (&a)[D + 4], const uint4 *aw, const half8 &b, std::integer_sequence<int, Ns...>) { (step<Ns>(acc, a, aw, b), ...); }
};
template <int D>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    extern __shared__ __align__(16) uint4 lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_uint4(i, i * 3, i * 7, i * 11);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
    half8 b;
    for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.002f * (lane - j));
    constexpr int N = 60;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const uint4 *aw = lds + lo;
        half8 a[D + 4];
#pragma unroll
        for (int n = 0; n < D; ++n) a[n] = *reinterpret_cast<const half8 *>(aw + n * 64);
        __builtin_amdgcn_sched_barrier(0);
        R<D, N, 0>::run(acc, a, aw, b, std::make_integer_sequence<int, N>{});
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template <int D> void run(int threads) {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 500;
    (void)hipFuncSetAttribute((const void *)k<D>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((k<D>), dim3(256), dim3(threads), 65536, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2048]; (void)hipMemcpy(h, cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < 256 * (threads / 64); ++i) s += h[i];
    printf("ring depth %d, waves/SIMD %d: %.1f cycles per MFMA per wave\n", D, threads / 256, s / (256.0 * (threads / 64)) / (iters * 60.0));
}
int main() {
    run<1>(256); run<2>(256); run<4>(256); run<8>(256); run<12>(256);
    run<2>(512); run<4>(512);
    return 0;
}
