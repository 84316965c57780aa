// fma_mix_check.hip — the fp16 split of a product through v_fma_mixlo / mixhi_f16 (csrc/hns_tp.hip, round 4): hi = RN16(P Q) of the EXACT product,
// lo = RN16(P Q - hi), including the subnormal range of lo (|P Q| < 2^-3) and both register halves.  Prints the mismatches against the host.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 fma_mix_check.hip -o fma_mix_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>

__global__ void k(const float *P, const float *Q, uint32_t *hi, uint32_t *lo, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned h, l;
    const float p0 = P[2 * i], q0 = Q[2 * i], p1 = P[2 * i + 1], q1 = Q[2 * i + 1];
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(p0), "v"(q0));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(l) : "v"(p0), "v"(q0), "v"(h));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(p1), "v"(q1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(p1), "v"(q1), "v"(h));
    hi[i] = h; lo[i] = l;
}

static uint16_t rn16(double x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static double f16d(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (double)h; }

int main() {
    const int n = 1 << 20;
    std::vector<float> P(n), Q(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int i = 0; i < n; ++i) {
        const double mag = std::exp2(-24.0 * rnd());              // products from 1 down to 2^-24: lo deep in the subnormal range
        P[i] = (float)((2.0 * rnd() - 1.0));
        Q[i] = (float)(mag * (0.5 + rnd()));
    }
    float *dP, *dQ; uint32_t *dh, *dl;
    hipMalloc(&dP, n * 4); hipMalloc(&dQ, n * 4); hipMalloc(&dh, n * 2); hipMalloc(&dl, n * 2);
    hipMemcpy(dP, P.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dQ, Q.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 512), dim3(256), 0, 0, dP, dQ, dh, dl, n);
    std::vector<uint32_t> h(n / 2), l(n / 2);
    hipMemcpy(h.data(), dh, n * 2, hipMemcpyDeviceToHost); hipMemcpy(l.data(), dl, n * 2, hipMemcpyDeviceToHost);
    int bad_hi = 0, bad_lo = 0, sub = 0; double worst = 0;
    for (int i = 0; i < n; ++i) {
        const uint16_t gh = (uint16_t)(h[i / 2] >> (16 * (i & 1))), gl = (uint16_t)(l[i / 2] >> (16 * (i & 1)));
        const double prod = (double)P[i] * (double)Q[i];
        const uint16_t rh = rn16(prod), rl = rn16(prod - f16d(rh));
        bad_hi += gh != rh; bad_lo += gl != rl;
        sub += (rl & 0x7c00) == 0 && (rl & 0x3ff) != 0;
        const double err = std::fabs(f16d(gh) + f16d(gl) - prod);
        if (std::fabs(prod) > 1e-30) worst = std::fmax(worst, err / std::fmax(std::fabs(prod), 6e-8));
    }
    printf("fma_mix split: %d products, hi mismatches %d, lo mismatches %d (%d subnormal lo terms), worst |hi + lo - PQ| / max(|PQ|, 2^-24) = %.3g\n", n, bad_hi, bad_lo, sub, worst);
    return (bad_hi || bad_lo) ? 1 : 0;
}
