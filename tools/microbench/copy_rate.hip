// copy_rate.hip — what a plain device copy reaches on this box, by kernel shape and buffer size: the yardstick bench.py prices the step
// kernel's "fraction of the achievable rate" against (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 copy_rate.hip -o copy_rate      run: ./copy_rate
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// A: one float4 per thread
__global__ __launch_bounds__(256) void copy_a(float4 *__restrict__ d, const float4 *__restrict__ s, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
// B: P float4 per thread, all loads in flight before the first store, each pass of a block one contiguous 4 KB
template <int P>
__global__ __launch_bounds__(256) void copy_b(float4 *__restrict__ d, const float4 *__restrict__ s, size_t n) {
    const size_t base = (size_t)blockIdx.x * 256 * P + threadIdx.x;
    float4 v[P];
#pragma unroll
    for (int i = 0; i < P; ++i) if (base + i * 256 < n) v[i] = s[base + i * 256];
#pragma unroll
    for (int i = 0; i < P; ++i) if (base + i * 256 < n) d[base + i * 256] = v[i];
}
// C: persistent grid (G blocks), grid-stride, P pieces in flight
template <int P>
__global__ __launch_bounds__(256) void copy_c(float4 *__restrict__ d, const float4 *__restrict__ s, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * P;
    for (size_t base = (size_t)blockIdx.x * 256 * P + threadIdx.x; base < n; base += stride) {
        float4 v[P];
#pragma unroll
        for (int i = 0; i < P; ++i) if (base + i * 256 < n) v[i] = s[base + i * 256];
#pragma unroll
        for (int i = 0; i < P; ++i) if (base + i * 256 < n) d[base + i * 256] = v[i];
    }
}
// D: as B with non-temporal loads and stores
template <int P>
__global__ __launch_bounds__(256) void copy_d(float4 *__restrict__ d, const float4 *__restrict__ s, size_t n) {
    const size_t base = (size_t)blockIdx.x * 256 * P + threadIdx.x;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 *sv = reinterpret_cast<const f4 *>(s);
    f4 *dv = reinterpret_cast<f4 *>(d);
    f4 v[P];
#pragma unroll
    for (int i = 0; i < P; ++i) if (base + i * 256 < n) v[i] = __builtin_nontemporal_load(sv + base + i * 256);
#pragma unroll
    for (int i = 0; i < P; ++i) if (base + i * 256 < n) __builtin_nontemporal_store(v[i], dv + base + i * 256);
}

int main() {
    const size_t sizes_mib[] = {24, 48, 96, 256, 1024};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (size_t mib : sizes_mib) {
        const size_t bytes = mib << 20, n = bytes / 16;
        float4 *s, *d;
        CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
        CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
        struct V { const char *name; void (*run)(float4 *, const float4 *, size_t); };
        const V vs[] = {
            {"A  1 float4/thread", [](float4 *dd, const float4 *ss, size_t nn) { hipLaunchKernelGGL(copy_a, dim3((nn + 255) / 256), dim3(256), 0, 0, dd, ss, nn); }},
            {"B  4/thread", [](float4 *dd, const float4 *ss, size_t nn) { hipLaunchKernelGGL(copy_b<4>, dim3((nn + 1023) / 1024), dim3(256), 0, 0, dd, ss, nn); }},
            {"B  8/thread", [](float4 *dd, const float4 *ss, size_t nn) { hipLaunchKernelGGL(copy_b<8>, dim3((nn + 2047) / 2048), dim3(256), 0, 0, dd, ss, nn); }},
            {"C  persistent 2048 blocks x4", [](float4 *dd, const float4 *ss, size_t nn) { hipLaunchKernelGGL(copy_c<4>, dim3(2048), dim3(256), 0, 0, dd, ss, nn); }},
            {"C  persistent 4096 blocks x2", [](float4 *dd, const float4 *ss, size_t nn) { hipLaunchKernelGGL(copy_c<2>, dim3(4096), dim3(256), 0, 0, dd, ss, nn); }},
            {"D  4/thread non-temporal", [](float4 *dd, const float4 *ss, size_t nn) { hipLaunchKernelGGL(copy_d<4>, dim3((nn + 1023) / 1024), dim3(256), 0, 0, dd, ss, nn); }},
            {"hipMemcpyAsync D2D", [](float4 *dd, const float4 *ss, size_t nn) { (void)hipMemcpyAsync(dd, ss, nn * 16, hipMemcpyDeviceToDevice, 0); }},
        };
        for (const V &v : vs) {
            for (int i = 0; i < 3; ++i) v.run(d, s, n);
            const int reps = mib >= 1024 ? 10 : 40;
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < reps; ++i) v.run(d, s, n);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5zu MiB + %5zu MiB  %-32s %7.1f GB/s (read + written)   %8.2f us per pass\n", mib, mib, v.name, reps * 2.0 * bytes / (ms * 1e-3) / 1e9, ms / reps * 1e3);
        }
        CK(hipFree(s)); CK(hipFree(d));
    }
    return 0;
}
