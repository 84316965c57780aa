// microbenchmark 3: an LDS-fed MFMA stream (waves 0-3) beside a VALU/transcendental stream (waves 4-7) on the same SIMDs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
constexpr int D = 4, N = 60;
template <int n> __device__ __forceinline__ void mstep(f32x16 (&acc)[4], half8 (&a)[D + 4], const uint4 *aw, const half8 &b) {
    acc[n % 4] = MFMA(a[n % (D + 4)], b, acc[n % 4]);
    if constexpr (n + D < N) a[(n + D) % (D + 4)] = *reinterpret_cast<const half8 *>(aw + (n + D) * 64);
    __builtin_amdgcn_sched_barrier(0);
}
template <int... Ns> __device__ __forceinline__ void mrun(f32x16 (&acc)[4], half8 (&a)[D + 4], const uint4 *aw, const half8 &b, std::integer_sequence<int, Ns...>) { (mstep<Ns>(acc, a, aw, b), ...); }

// mode bit 0: group 0 runs the MFMA stream; bit 1: group 1 runs the VALU stream; prio: s_setprio of the VALU group
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters, int mode, int prio) {
    extern __shared__ __align__(16) uint4 lds[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_uint4(i, i * 3, i * 7, i * 11);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, group = wave >> 2;
    float s = 0.f;
    unsigned long long t0 = 0, t1 = 0;
    if (group == 0) {
        if (mode & 1) {
            f32x16 acc[4];
            for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
            half8 b;
            for (int j = 0; j < 8; ++j) b[j] = (_Float16)(0.002f * (lane - j));
            t0 = __builtin_readcyclecounter();
            for (int it = 0; it < iters; ++it) {
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const uint4 *aw = lds + lo;
                half8 a[D + 4];
#pragma unroll
                for (int n = 0; n < D; ++n) a[n] = *reinterpret_cast<const half8 *>(aw + n * 64);
                __builtin_amdgcn_sched_barrier(0);
                mrun(acc, a, aw, b, std::make_integer_sequence<int, N>{});
            }
            t1 = __builtin_readcyclecounter();
            for (int q = 0; q < 4; ++q) for (int i = 0; i < 16; ++i) s += acc[q][i];
        }
    } else if (mode & 2) {
        if (prio) __builtin_amdgcn_s_setprio(3);
        float x[16];
        for (int i = 0; i < 16; ++i) x[i] = 0.01f * (lane + i);
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)          // 4 x 16 units x (exp, add, rcp, fma, mul) = 320 VALU, 128 transcendental
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float e = __builtin_amdgcn_exp2f(x[i]);
                    const float d = __builtin_amdgcn_rcpf(1.0f + e);
                    x[i] = __builtin_fmaf(d, 0.5f, x[i] * 0.25f);
                }
        }
        t1 = __builtin_readcyclecounter();
        for (int i = 0; i < 16; ++i) s += x[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int iters = 500;
    for (int mode = 1; mode <= 3; ++mode)
        for (int prio = 0; prio <= (mode == 3 ? 1 : 0); ++prio) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, 0, out, cyc, iters, mode, prio);
            (void)hipDeviceSynchronize();
            unsigned long long h[2048]; (void)hipMemcpy(h, cyc, 256 * 8 * 8, hipMemcpyDeviceToHost);
            double m = 0, v = 0;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w];
            printf("mode %d prio %d: MFMA stream %.1f cycles/MFMA, VALU stream %.2f cycles/instr\n", mode, prio,
                   m / 1024.0 / (iters * 60.0), v / 1024.0 / (iters * 320.0));
        }
    return 0;
}
