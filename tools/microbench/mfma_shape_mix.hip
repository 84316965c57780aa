// mfma_shape_mix.hip — would the predictor's recurrence run faster on v_mfma_f32_16x16x32_f16 than on v_mfma_f32_32x32x16_f16?
// Round 3 found that a 32x32x16 stream starves the VALU of its own SIMD, so the kernel's time is the SUM of its matrix and vector work.
// Here every wave (4 per SIMD, as in hns_tp_lstm_ws_kernel<1>) runs the kernel's own proportion for the same 8 units x 32 envs of work:
//   S32 : 15 dependent 32x32x16 MFMAs into one accumulator, then the cell update of 4 units per lane (88 vector instructions, 32 of them
//         transcendental) — blocked (what the kernel does) or with the vector instructions spread between the MFMAs;
//   S16 : twice { 18 16x16x32 MFMAs (two accumulators of 9: K = 80 padded to 96), cell update of 2 units per lane (44 / 16) } — blocked or spread.
// No LDS, no barriers: an upper bound on what the instruction mix alone allows.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 mfma_shape_mix.hip -o mfma_shape_mix
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// 11 vector instructions on one register set: 4 transcendental (2 exp2, 2 rcp), 7 plain — half a unit's cell update
#define V11(a, b, c) "v_exp_f32 %" #a ", %" #a "\n v_exp_f32 %" #b ", %" #b "\n v_add_f32 %" #c ", 1.0, %" #a "\n v_fma_f32 %" #a ", %" #b ", %" #c ", %" #c "\n" \
                     "v_rcp_f32 %" #a ", %" #a "\n v_sub_f32 %" #b ", 1.0, %" #b "\n v_mul_f32 %" #c ", %" #b ", %" #a "\n v_rcp_f32 %" #b ", %" #c "\n"          \
                     "v_fma_f32 %" #c ", %" #a ", %" #b ", %" #c "\n v_mul_f32 %" #a ", %" #c ", %6\n v_min_f32 %" #b ", %" #a ", %6\n"
#define VREGS "+v"(x0), "+v"(x1), "+v"(x2), "+v"(y0), "+v"(y1), "+v"(y2) : "v"(k)
#define UNIT asm volatile(V11(0, 1, 2) V11(3, 4, 5) : VREGS);          /* 22 instructions = one unit's cell update */
#define HALF_A asm volatile(V11(0, 1, 2) : VREGS);
#define HALF_B asm volatile(V11(3, 4, 5) : VREGS);

template <int MODE>
__global__ __launch_bounds__(1024) void k(float *out, int iters, unsigned long long *cyc) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    float x0 = 0.1f * threadIdx.x, x1 = 0.2f, x2 = 0.3f, y0 = 0.4f, y1 = 0.5f, y2 = 0.6f;
    const float k = 0.999f;
    f32x16 acc = {};
    f32x4 c0 = {}, c1 = {};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {                       // S32 blocked
#pragma unroll
            for (int m = 0; m < 15; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            UNIT UNIT UNIT UNIT
        } else if constexpr (MODE == 1) {                // S32 spread: 88 instructions in 8 blocks of 11 between the MFMAs
#pragma unroll
            for (int m = 0; m < 15; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
                if (m < 8) { if (m & 1) { HALF_B } else { HALF_A } }
            }
        } else if constexpr (MODE == 2) {                // S16 blocked, twice
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
                }
                UNIT UNIT
            }
        } else {                                         // S16 spread: 11 instructions behind every fourth-fifth MFMA
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int m = 0; m < 9; ++m) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
                    if (m == 1 || m == 5) { HALF_A }
                    if (m == 3 || m == 7) { HALF_B }
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = x0 + x1 + x2 + y0 + y1 + y2;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 4; ++i) s += c0[i] + c1[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int iters = 2000, blocks = 256;
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
    const char *names[4] = {"S32 blocked (the kernel's regime)", "S32 spread", "S16 blocked", "S16 spread"};
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(1024), 0, 0, out, iters, cyc); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(1024), 0, 0, out, iters, cyc); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(1024), 0, 0, out, iters, cyc); break;
                default: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(1024), 0, 0, out, iters, cyc); break;
            }
            hipDeviceSynchronize();
        }
        unsigned long long h[256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0;
        for (int i = 0; i < blocks; ++i) s += (double)h[i];
        printf("%-36s %8.0f cycles per (8 units x 32 envs) of one wave, 4 waves per SIMD  -> %6.0f cycles per SIMD\n", names[mode], s / blocks / iters, 4 * s / blocks / iters);
    }
    return 0;
}
