// exchange_latency.hip — per-round cost of an all-to-all "publish one granule, wait for everybody's" exchange between
// persistent workgroups, the communication step of hns_fps (csrc/hns_envgen.hip).
//   mode 0: agent-scope write-through store (sc1) + agent-scope loads (sc1): valid across XCDs — what hns_fps does
//   mode 1: L2 atomics (swap to publish, add 0 to poll) without scope bits: only coherent inside ONE XCD's L2
// placement: every workgroup works, or only those with blockIdx % 8 == 0 (round-robin dispatch puts them on one XCD; checked
// with XCC_ID).  build: hipcc -O3 --offload-arch=gfx950 exchange_latency.hip -o exchange_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;

__global__ __launch_bounds__(256) void k(u64 *gran, int groups, int stride, int rounds, int mode, u64 *out, unsigned *xcc) {
    if (blockIdx.x % stride) return;
    const int g = blockIdx.x / stride, tid = threadIdx.x;
    if (tid == 0) xcc[g] = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
    u64 t0 = 0;
    __shared__ int ok;
    for (int r = 1; r <= rounds; ++r) {
        if (r == 2 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
        u64 *slot = gran + (size_t)(r & 1) * groups;
        if (tid == 0) {
            if (mode == 0) __hip_atomic_store(slot + g, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else (void)__hip_atomic_exchange(slot + g, (u64)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // every thread watches one granule (groups <= 256)
        if (tid < groups) {
            unsigned spin = 0;
            for (;;) {
                u64 v;
                if (mode == 0) v = __hip_atomic_load(slot + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else v = __hip_atomic_fetch_add(slot + tid, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (v == (u64)r || ++spin > (1u << 14)) break;       // bounded: a wrong coherence assumption shows up as a slow round, not a hang
            }
        }
        __syncthreads();
    }
    if (tid == 0) out[g] = __builtin_amdgcn_s_memrealtime() - t0;
    (void)ok;
}

int main() {
    u64 *gran, *out;
    unsigned *xcc;
    hipMalloc(&gran, 2 * 256 * sizeof(u64));
    hipMalloc(&out, 256 * sizeof(u64));
    hipMalloc(&xcc, 256 * sizeof(unsigned));
    const int rounds = 1000;
    struct Case { const char *name; int grid, stride, mode; };
    const Case cases[] = {{"256 workgroups, all XCDs, agent-scope store/load", 256, 1, 0},
                          {"32 workgroups on all XCDs, agent-scope store/load", 32, 1, 0},
                          {"32 workgroups on ONE XCD (blockIdx % 8 == 0), agent-scope store/load", 256, 8, 0},
                          {"32 workgroups on ONE XCD, L2 atomics", 256, 8, 1},
                          {"16 workgroups on ONE XCD, L2 atomics", 128, 8, 1}};
    for (const Case &c : cases) {
        const int groups = c.grid / c.stride;
        hipMemset(gran, 0, 2 * 256 * sizeof(u64));
        hipMemset(out, 0, 256 * sizeof(u64));
        hipLaunchKernelGGL(k, dim3(c.grid), dim3(256), 0, 0, gran, groups, c.stride, rounds, c.mode, out, xcc);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", c.name); return 1; }
        std::vector<u64> h(groups);
        std::vector<unsigned> x(groups);
        hipMemcpy(h.data(), out, groups * sizeof(u64), hipMemcpyDeviceToHost);
        hipMemcpy(x.data(), xcc, groups * sizeof(unsigned), hipMemcpyDeviceToHost);
        u64 mx = 0;
        unsigned xmask = 0;
        for (int i = 0; i < groups; ++i) { if (h[i] > mx) mx = h[i]; xmask |= 1u << x[i]; }
        printf("%-90s %7.2f us per round   (XCD mask 0x%02x)\n", c.name, mx * 10.0 / 1000.0 / (rounds - 1), xmask);
    }
    return 0;
}
