// Cost of one back-to-back kernel launch on one stream, by launch shape: what a serial launch pays besides its work.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 launch_gap.hip -o launch_gap && ./launch_gap
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { float v[224]; float *out; };     // ~900 B of kernel arguments, like the step kernel's Params
struct Small { float *out; };
template <int N> struct Mid { float v[N]; float *out; };
template <class P, int VG>
__global__ __launch_bounds__(256) void k_empty(const P p) {
    extern __shared__ float smem[];
    if (p.out == nullptr) {                     // never true: keeps LDS and arguments alive
        float acc[VG];
        for (int i = 0; i < VG; ++i) acc[i] = smem[threadIdx.x + i];
        float s = 0; for (int i = 0; i < VG; ++i) s += acc[i] * acc[(i * 7) % VG];
        smem[threadIdx.x] = s;
    }
}
template <class P, int VG>
__global__ __launch_bounds__(256) void k_touch(const P p) {   // every thread stores 16 B: 4 MB dirty per launch
    extern __shared__ float smem[];
    reinterpret_cast<float4 *>(p.out)[blockIdx.x * 256 + threadIdx.x] = make_float4(1, 2, 3, (float)threadIdx.x);
}
template <class K, class P>
double run(K kern, P p, int grid, size_t lds, int n = 4000) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, p);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / n;
}
int main() {
    float *out; hipMalloc(&out, 64 << 20);
    Small s{out}; Big b{}; b.out = out;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Small, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Big, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Big, 100>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Mid<14>, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Mid<30>, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Mid<62>, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Mid<94>, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Mid<126>, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Mid<158>, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty<Mid<190>, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 40960);
    printf("us per back-to-back launch (4000 launches, one stream)\n");
    printf("empty, 256 WG,  no LDS, 8 B args      %.2f\n", run(k_empty<Small, 4>, s, 256, 0));
    printf("empty, 1024 WG, no LDS, 8 B args      %.2f\n", run(k_empty<Small, 4>, s, 1024, 0));
    printf("empty, 1024 WG, 37 KB LDS, 8 B args   %.2f\n", run(k_empty<Small, 4>, s, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 900 B args %.2f\n", run(k_empty<Big, 4>, b, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 900 B args, 100+ VGPRs  %.2f\n", run(k_empty<Big, 100>, b, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS,  64 B args %.2f\n", run(k_empty<Mid<14>, 4>, Mid<14>{{}, out}, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 128 B args %.2f\n", run(k_empty<Mid<30>, 4>, Mid<30>{{}, out}, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 256 B args %.2f\n", run(k_empty<Mid<62>, 4>, Mid<62>{{}, out}, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 384 B args %.2f\n", run(k_empty<Mid<94>, 4>, Mid<94>{{}, out}, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 512 B args %.2f\n", run(k_empty<Mid<126>, 4>, Mid<126>{{}, out}, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 640 B args %.2f\n", run(k_empty<Mid<158>, 4>, Mid<158>{{}, out}, 1024, 37 * 1024));
    printf("empty, 1024 WG, 37 KB LDS, 768 B args %.2f\n", run(k_empty<Mid<190>, 4>, Mid<190>{{}, out}, 1024, 37 * 1024));
    printf("4 MB stored, 1024 WG, no LDS          %.2f\n", run(k_touch<Small, 4>, s, 1024, 0));
    printf("16 MB stored, 4096 WG, no LDS         %.2f\n", run(k_touch<Small, 4>, s, 4096, 0));
    return 0;
}
