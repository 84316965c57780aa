// hns_tp.hip — the trajectory predictor inside the observation, for gfx950.
//
// Reference: the `use_TP_net` branch of HideAndSeek._compute_state_and_obs
// (omni_drones/envs/hide_and_seek/hideandseek.py:805-854, 871-880) with TP_net =
// LSTM(I -> 64, 1 layer, zero initial state) + Linear(64 -> 3F) + tanh
// (omni_drones/learning/mappo.py:572-589), I = 7 + 3A, evaluated every step on a T-frame window.
// In the reference this is a cuDNN LSTM over [E,T,I] plus ~25 elementwise launches; here it is
//   hns_tp_lstm_kernel : frame append + window shift + LSTM + FC + tanh + rescale   (MFMA-bound)
//   hns_tp_rows_kernel : the 20+3F-value observation rows                           (HBM-bound)
//
// This is the one dense contraction near the hot path: per env and timestep z[256] = W[256 x (I+64)]·[x;h].
// Mapping onto the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 — bit-for-bit a k-ordered
// fmaf chain, so the CPU oracle reproduces every gate pre-activation exactly):
//   * one wave owns 32 envs for the whole window; gates are the M dimension (8 tiles of 32 rows),
//     envs the N dimension, [x;h] the K dimension;
//   * the weights sit in LDS in A-operand order (lane l: row l&31, k-slot l>>5), staged once per
//     workgroup; x_t and h_{t-1} are B operands held in registers;
//   * D tile layout: lane (env n = l&31, half hb = l>>5), register i  <->  row 8(i>>2)+4hb+(i&3).
//     Gate q of hidden unit u lives in tile 2q + u/32, row u%32: the four gates of a unit land in
//     the SAME lane and register index, so the cell update is lane-local;
//   * the K order of the recurrent product is free, so k-step s is DEFINED to pair the units
//     that the lower and the upper half-wave hold in register s: h_t leaves the cell update in
//     exactly the registers the next timestep's B operand reads — no shuffle, no LDS round trip.
// 8 waves per workgroup (2 per SIMD: one wave's gate nonlinearities overlap the other's MFMAs),
// 256 envs per workgroup => 256 workgroups = one per CU at 65 536 envs.
#include <hip/hip_runtime.h>

#include <new>
#include <string>

#include "hns_device.h"
#include "hns_host.h"

namespace hns {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTpH = HNS_TP_HIDDEN;
constexpr int kTpWaves = 8;
constexpr int kTpThreads = kTpWaves * 64;
constexpr int kTpEnvs = kTpWaves * 32;      // envs per workgroup
constexpr int kTpMaxRows = 32;              // 3F <= 32: one M tile for the output layer

struct TpParams {
    hns_tp_buffers tp;
    const float *drone_state, *target_pos, *target_vel, *progress, *obs_self20;
    const uint8_t *detect;
    int E, A, I, T, F, fill, max_len;
    float mask_value, arena_size, max_height;
};

// LDS image (floats).  W_hh / W_ih / W_fc as [tile][k-step/4][lane][4]: one ds_read_b128 per lane
// yields the A operands of 4 consecutive k-steps; biases as [tile][half][16] (broadcast reads).
struct TpLds {
    int whh, wih, wfc, bias, bfc, total;
};
__host__ __device__ inline TpLds tp_lds_layout(int sxq) {
    TpLds L;
    int o = 0;
    L.whh = o;  o += 8 * 8 * 64 * 4;
    L.wih = o;  o += 8 * sxq * 64 * 4;
    L.wfc = o;  o += 8 * 64 * 4;
    L.bias = o; o += 8 * 2 * 16;
    L.bfc = o;  o += 2 * 16;
    L.total = o;
    return L;
}

// hidden unit that half-wave `hb` holds in register s (s = 16*tile_pair + i): D row 8(i>>2)+4hb+(i&3)
__host__ __device__ inline int tp_unit(int s, int hb) { return 32 * (s >> 4) + 8 * ((s & 15) >> 2) + 4 * hb + (s & 3); }

// gate nonlinearities on the transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each); the oracle
// uses libm, the parity tolerance is the north star's 1e-5
HNS_DEV float tp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
HNS_DEV float tp_tanh(float x) {
    return HNS_FMA(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -2.8853900817779268f)), -1.0f);
}

// component k of the frame [progress, evader pos (masked), evader vel (masked), pursuer positions]
// (hideandseek.py:815-820; the mask is broadcast_detect, :791-803)
HNS_DEV float tp_frame_val(const TpParams &p, int e, int k, bool det) {
    if (k == 0) return p.progress[e];
    if (k < 4) return det ? p.target_pos[(size_t)e * 3 + (k - 1)] : p.mask_value;
    if (k < 7) return det ? p.target_vel[(size_t)e * 3 + (k - 4)] : p.mask_value;
    const int j = k - 7, a = j / 3;
    return p.drone_state[((size_t)e * p.A + a) * 13 + (j - 3 * a)];
}

template <int SXQ>
__global__ __launch_bounds__(kTpThreads) void hns_tp_lstm_kernel(const TpParams p) {
    constexpr int SX = 4 * SXQ;                 // k-steps of the input product; half-wave hb takes x[hb*SX + s]
    extern __shared__ __align__(16) float smem[];
    const TpLds L = tp_lds_layout(SXQ);
    float *sWhh = smem + L.whh, *sWih = smem + L.wih, *sWfc = smem + L.wfc, *sB = smem + L.bias, *sBfc = smem + L.bfc;
    const int tid = threadIdx.x, I = p.I, T = p.T, R = 3 * p.F;

    // ---- stage the parameters in A-operand order (80-100 KB, L2-resident source) -----------------
    for (int idx = tid; idx < 8 * 8 * 64 * 4; idx += kTpThreads) {
        const int r = idx & 3, ln = (idx >> 2) & 63, sq = (idx >> 8) & 7, m = idx >> 11;
        const int row = 32 * m + (ln & 31);
        sWhh[idx] = p.tp.w_hh[row * kTpH + tp_unit(4 * sq + r, ln >> 5)];
    }
    for (int idx = tid; idx < 8 * SXQ * 64 * 4; idx += kTpThreads) {
        const int r = idx & 3, ln = (idx >> 2) & 63, g = idx >> 8, m = g / SXQ, sq = g - m * SXQ;
        const int row = 32 * m + (ln & 31), k = (ln >> 5) * SX + 4 * sq + r;
        sWih[idx] = k < I ? p.tp.w_ih[row * I + k] : 0.0f;
    }
    for (int idx = tid; idx < 8 * 64 * 4; idx += kTpThreads) {
        const int r = idx & 3, ln = (idx >> 2) & 63, sq = idx >> 8;
        const int row = ln & 31;
        sWfc[idx] = row < R ? p.tp.w_fc[row * kTpH + tp_unit(4 * sq + r, ln >> 5)] : 0.0f;
    }
    for (int idx = tid; idx < 8 * 2 * 16; idx += kTpThreads) {
        const int i = idx & 15, hbb = (idx >> 4) & 1, m = idx >> 5;
        const int g = 32 * m + 8 * (i >> 2) + 4 * hbb + (i & 3);
        sB[idx] = p.tp.b_ih[g] + p.tp.b_hh[g];
    }
    for (int idx = tid; idx < 2 * 16; idx += kTpThreads) {
        const int i = idx & 15, hbb = idx >> 4;
        const int row = 8 * (i >> 2) + 4 * hbb + (i & 3);
        sBfc[idx] = row < R ? p.tp.b_fc[row] : 0.0f;
    }
    __syncthreads();

    const int wave = tid >> 6, lane = tid & 63, hb = lane >> 5;
    const int e = blockIdx.x * kTpEnvs + wave * 32 + (lane & 31);
    if (blockIdx.x * kTpEnvs + wave * 32 >= p.E) return;          // whole wave out of range
    const bool valid = e < p.E;
    const int ec = valid ? e : p.E - 1;                           // clamped: loads stay in bounds, stores are guarded
    const int k0 = hb * SX;

    // the new frame, this lane's half
    float xn[SX];
    {
        const bool det = p.detect[ec] != 0;
#pragma unroll
        for (int s = 0; s < SX; ++s) xn[s] = (k0 + s < I) ? tp_frame_val(p, ec, k0 + s, det) : 0.0f;
    }
    float *hist = p.tp.history + (size_t)ec * T * I + k0;
    // x_t = old frame t+1 for t <= T-2, the new frame for t = T-1 (or for every t when filling)
    float xb[SX];
    if (T == 1 || p.fill) {
#pragma unroll
        for (int s = 0; s < SX; ++s) xb[s] = xn[s];
    } else {
#pragma unroll
        for (int s = 0; s < SX; ++s) xb[s] = (k0 + s < I) ? hist[I + s] : 0.0f;
    }

    float h[32], c[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) { h[i] = 0.0f; c[i] = 0.0f; }

    for (int t = 0; t < T; ++t) {
        // prefetch x_{t+1} (slot t+2 of the old window, untouched so far), then shift x_t into slot t
        float xnext[SX];
        const bool from_hist = !p.fill && (t + 1 <= T - 2);
#pragma unroll
        for (int s = 0; s < SX; ++s) xnext[s] = (from_hist && k0 + s < I) ? hist[(t + 2) * I + s] : xn[s];
        if (valid) {
#pragma unroll
            for (int s = 0; s < SX; ++s)
                if (k0 + s < I) hist[t * I + s] = xb[s];
        }
        float hn[32];
        // the weight image is loop-invariant; an opaque lane offset keeps the compiler from hoisting
        // every A operand of the window (hundreds of registers) out of the timestep loop
        int lo = lane * 4;
        asm volatile("" : "+v"(lo));
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {                 // units 32tj..32tj+31: gate tiles m = 2q + tj
            f32x16 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 *b4 = reinterpret_cast<const float4 *>(sB + ((2 * q + tj) * 2 + hb) * 16);
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float4 bv = b4[v];
                    acc[q][4 * v] = bv.x; acc[q][4 * v + 1] = bv.y; acc[q][4 * v + 2] = bv.z; acc[q][4 * v + 3] = bv.w;
                }
            }
#pragma unroll
            for (int sq = 0; sq < SXQ; ++sq) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4 *>(sWih + ((2 * q + tj) * SXQ + sq) * 256 + lo);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, xb[4 * sq], acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, xb[4 * sq + 1], acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, xb[4 * sq + 2], acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, xb[4 * sq + 3], acc[q], 0, 0, 0);
                }
            }
            if (t > 0) {                                   // h_0 = 0: the recurrent product vanishes
#pragma unroll
                for (int sq = 0; sq < 8; ++sq) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 a = *reinterpret_cast<const float4 *>(sWhh + ((2 * q + tj) * 8 + sq) * 256 + lo);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, h[4 * sq], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, h[4 * sq + 1], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, h[4 * sq + 2], acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, h[4 * sq + 3], acc[q], 0, 0, 0);
                    }
                }
            }
            // cell update (torch.nn.LSTM: i, f, g, o), lane-local
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float ig = tp_sigmoid(acc[0][i]), fg = tp_sigmoid(acc[1][i]);
                const float gg = tp_tanh(acc[2][i]), og = tp_sigmoid(acc[3][i]);
                const float cn = HNS_FMA(fg, c[16 * tj + i], ig * gg);
                c[16 * tj + i] = cn;
                hn[16 * tj + i] = og * tp_tanh(cn);
            }
        }
#pragma unroll
        for (int i = 0; i < 32; ++i) h[i] = hn[i];
#pragma unroll
        for (int s = 0; s < SX; ++s) xb[s] = xnext[s];
    }

    // ---- output layer on h_T: tanh(W_fc h + b), rescaled to arena units (hideandseek.py:834-836) ----
    f32x16 o;
    {
        const float4 *b4 = reinterpret_cast<const float4 *>(sBfc + hb * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 bv = b4[v];
            o[4 * v] = bv.x; o[4 * v + 1] = bv.y; o[4 * v + 2] = bv.z; o[4 * v + 3] = bv.w;
        }
    }
#pragma unroll
    for (int sq = 0; sq < 8; ++sq) {
        const float4 a = *reinterpret_cast<const float4 *>(sWfc + (sq * 64 + lane) * 4);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, h[4 * sq], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, h[4 * sq + 1], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, h[4 * sq + 2], o, 0, 0, 0);
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, h[4 * sq + 3], o, 0, 0, 0);
    }
    if (valid) {
        float *pr = p.tp.pred + (size_t)e * R;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 8 * (i >> 2) + 4 * hb + (i & 3);
            if (row < R) {
                const float v = tp_tanh(o[i]);
                const int comp = row % 3;
                pr[row] = (comp < 2) ? (v * 0.5f) * p.arena_size : ((v + 1.0f) * 0.5f) * p.max_height;
            }
        }
    }
}

// ---- observation rows: [rpos_evader(3) | drone - predicted (3F) | quat4 linvel3 heading3 up3 t x4 (17)] -------------
// hideandseek.py:844-854 (state_self, masked rpos from the step kernel's rows) and :873-880
// (state_drones, unmasked rpos); TP_groundtruth / TP_done :838-842.  One thread per pursuer, rows
// assembled in LDS (odd stride) and written as one contiguous slice per workgroup.
constexpr int kRowThreads = 256;
__global__ __launch_bounds__(kRowThreads) void hns_tp_rows_kernel(const TpParams p) {
    extern __shared__ __align__(16) float smem[];
    const int R = 3 * p.F, D = HNS_SELF_DIM + R;
    const int n_agents = p.E * p.A;
    const int first = blockIdx.x * kRowThreads, ia = first + threadIdx.x;
    const int nrows = min(kRowThreads, n_agents - first);
    const bool valid = ia < n_agents;
    float o20[HNS_SELF_DIM];
    float px = 0.f, py = 0.f, pz = 0.f, tx = 0.f, ty = 0.f, tz = 0.f;
    int e = 0;
    if (valid) {
        e = ia / p.A;
        const float4 *s4 = reinterpret_cast<const float4 *>(p.obs_self20 + (size_t)ia * HNS_SELF_DIM);
#pragma unroll
        for (int v = 0; v < HNS_SELF_DIM / 4; ++v) {
            const float4 q = s4[v];
            o20[4 * v] = q.x; o20[4 * v + 1] = q.y; o20[4 * v + 2] = q.z; o20[4 * v + 3] = q.w;
        }
        const float *ds = p.drone_state + (size_t)ia * 13;
        px = ds[0]; py = ds[1]; pz = ds[2];
        tx = p.target_pos[(size_t)e * 3]; ty = p.target_pos[(size_t)e * 3 + 1]; tz = p.target_pos[(size_t)e * 3 + 2];
        if (ia - e * p.A == 0) {
            // CUDA scalar-division form: tensor / python_scalar multiplies by the fp32 reciprocal
            float *gt = p.tp.groundtruth + (size_t)e * 3;
            gt[0] = tx * (1.0f / (0.5f * p.arena_size));
            gt[1] = ty * (1.0f / (0.5f * p.arena_size));
            gt[2] = (tz * (1.0f / p.max_height)) * 2.0f - 1.0f;
            p.tp.tp_done[e] = (uint8_t)(p.progress[e] <= (float)(p.max_len - p.F));
        }
    }
    for (int pass = 0; pass < 2; ++pass) {               // 0: state_self, 1: state_drones
        float *dst = pass == 0 ? p.tp.obs_self : p.tp.state_drones;
        if (!dst) continue;
        if (pass) __syncthreads();
        if (valid) {
            float *row = smem + threadIdx.x * D;
            if (pass == 0) { row[0] = o20[0]; row[1] = o20[1]; row[2] = o20[2]; }
            else { row[0] = px - tx; row[1] = py - ty; row[2] = pz - tz; }
            const float *pr = p.tp.pred + (size_t)e * R;
            for (int f = 0; f < p.F; ++f) {
                row[3 + 3 * f] = px - pr[3 * f];
                row[4 + 3 * f] = py - pr[3 * f + 1];
                row[5 + 3 * f] = pz - pr[3 * f + 2];
            }
#pragma unroll
            for (int j = 3; j < HNS_SELF_DIM; ++j) row[R + j] = o20[j];
        }
        __syncthreads();
        float *g = dst + (size_t)first * D;
        const int n = nrows * D;
        if ((((size_t)first * D) & 3) == 0) {
            const int n4 = n >> 2;
            for (int i = threadIdx.x; i < n4; i += kRowThreads) reinterpret_cast<float4 *>(g)[i] = reinterpret_cast<const float4 *>(smem)[i];
            for (int i = (n4 << 2) + threadIdx.x; i < n; i += kRowThreads) g[i] = smem[i];
        } else {
            for (int i = threadIdx.x; i < n; i += kRowThreads) g[i] = smem[i];
        }
    }
}

}  // namespace hns

// =================================================================================================
// Host side
// =================================================================================================
using hns::TpParams;

static int tp_sxq(int I) { return ((I + 1) / 2 + 3) / 4; }

extern "C" {

int hns_tp_bind(hns_env *env, const hns_tp_buffers *b, int32_t history_step, int32_t future_step) {
    if (!env || !b) { hns_set_error("hns_tp_bind: null argument"); return HNS_ERR_INVALID_ARG; }
    if (!env->bound || !env->buf.detect) {
        hns_set_error("hns_tp_bind: bind the step buffers first, with hns_buffers.detect set (the frame mask is broadcast_detect)");
        return HNS_ERR_NOT_BOUND;
    }
    if (history_step < 1 || history_step > 16 || future_step < 1 || 3 * future_step > hns::kTpMaxRows) {
        hns_set_error("hns_tp_bind: history_step must be in [1,16], future_step in [1,10]");
        return HNS_ERR_INVALID_ARG;
    }
    if (!b->w_ih || !b->w_hh || !b->b_ih || !b->b_hh || !b->w_fc || !b->b_fc || !b->history || !b->pred || !b->obs_self ||
        !b->groundtruth || !b->tp_done) {
        hns_set_error("hns_tp_bind: null buffer (only state_drones may be null)");
        return HNS_ERR_INVALID_ARG;
    }
    const int sxq = tp_sxq(7 + 3 * env->cfg.num_agents);
    if (sxq < 2 || sxq > 4) { hns_set_error("hns_tp_bind: unsupported frame width"); return HNS_ERR_CONFIG; }
    env->tp.buf = *b;
    env->tp.history_step = history_step;
    env->tp.future_step = future_step;
    env->tp.bound = true;
    return HNS_OK;
}

int hns_tp_observe(hns_env *env, int32_t fill_history, void *stream) {
    if (!env) { hns_set_error("hns_tp_observe: null env"); return HNS_ERR_INVALID_ARG; }
    if (!env->tp.bound) { hns_set_error("hns_tp_observe: hns_tp_bind first"); return HNS_ERR_NOT_BOUND; }
    const hns_cfg &c = env->cfg;
    TpParams p;
    p.tp = env->tp.buf;
    p.drone_state = env->buf.drone_state;
    p.target_pos = env->buf.target_pos;
    p.target_vel = env->buf.target_vel;
    p.progress = env->buf.progress;
    p.obs_self20 = env->buf.obs_self;
    p.detect = env->buf.detect;
    p.E = c.num_envs; p.A = c.num_agents; p.I = 7 + 3 * c.num_agents;
    p.T = env->tp.history_step; p.F = env->tp.future_step;
    p.fill = fill_history ? 1 : 0;
    p.max_len = c.max_episode_length;
    p.mask_value = c.mask_value; p.arena_size = c.arena_size; p.max_height = c.max_height;
    const int sxq = tp_sxq(p.I);
    void (*fn)(const TpParams) = sxq == 2 ? hns::hns_tp_lstm_kernel<2> : (sxq == 3 ? hns::hns_tp_lstm_kernel<3> : hns::hns_tp_lstm_kernel<4>);
    const size_t lds = (size_t)hns::tp_lds_layout(sxq).total * sizeof(float);
    static thread_local const void *attr_set[3] = {nullptr, nullptr, nullptr};
    if (attr_set[sxq - 2] != (const void *)fn) {
        HNS_CHECK_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[sxq - 2] = (const void *)fn;
    }
    hipStream_t s = (hipStream_t)stream;
    const int grid = (p.E + hns::kTpEnvs - 1) / hns::kTpEnvs;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(hns::kTpThreads), lds, s, p);
    HNS_CHECK_HIP(hipGetLastError());
    const int D = HNS_SELF_DIM + 3 * p.F;
    const int rgrid = (p.E * p.A + hns::kRowThreads - 1) / hns::kRowThreads;
    hipLaunchKernelGGL(hns::hns_tp_rows_kernel, dim3(rgrid), dim3(hns::kRowThreads), (size_t)hns::kRowThreads * D * sizeof(float), s, p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

}  // extern "C"
