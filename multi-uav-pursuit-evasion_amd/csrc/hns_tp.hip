// hns_tp.hip — the trajectory predictor inside the observation, for gfx950.
//
// Reference: the `use_TP_net` branch of HideAndSeek._compute_state_and_obs
// (omni_drones/envs/hide_and_seek/hideandseek.py:805-854, 871-880) with TP_net =
// LSTM(I -> 64, 1 layer, zero initial state) + Linear(64 -> 3F) + tanh
// (omni_drones/learning/mappo.py:572-589), I = 7 + 3A, evaluated every step on a T-frame window.
// In the reference this is a cuDNN LSTM over [E,T,I] plus ~25 elementwise launches; here it is
//   hns_tp_pack_kernel : parameters -> matrix-core operand image (only when they changed)
//   hns_tp_lstm_kernel : frame append + window shift + LSTM + FC + tanh + rescale, then the 20+3F-value observation rows of
//                        the wave's envs (a second, HBM-bound kernel until round 2)
//
// This is the one dense contraction near the hot path: per env and timestep z[256] = W[256 x (I+64)]·[x;h].
// It runs on the matrix cores at fp32-class accuracy with TWO-TERM fp16 SPLITS:
//     w = w1 + 2^-11 w2,  v = v1 + 2^-11 v2   (w1 = fp16(w), w2 = fp16((w - w1) 2^11), same for v)
//     w·v ~= w1 v1 + 2^-11 (w1 v2 + w2 v1)            (dropped: 2^-22 w2 v2)
// i.e. three v_mfma_f32_32x32x16_f16 per product tile: the two cross terms accumulate first, are
// scaled by 2^-11 (with the bias added) and the leading term accumulates on top, all in fp32.
// Measured against fp64 the split is as accurate as a plain fp32 evaluation (<= 3e-7 on the
// outputs; tools/tp_split_error.py), and it is 5.3x fewer matrix-pipe cycles than
// v_mfma_f32_32x32x2_f32 — which on gfx950 moreover shares the VALU's fp32 lanes: measured here,
// a partner wave's gate nonlinearities took 12.6k cycles beside it instead of 2.3k, so nothing
// overlapped (the fp32-MFMA version of this kernel ran 228 us, this one see DESIGN.md §8).
// The scaling keeps the low parts out of fp16's subnormal range; |x| must stay below 65 504
// (the frame holds `progress` <= max_episode_length, checked at bind time).
//
// Mapping:
//   * one wave owns 32 envs for the whole window; gates are the M dimension (8 tiles of 32 rows),
//     envs the N dimension, [x;h] the K dimension (16 per MFMA);
//   * the operand image sits in LDS in A-operand order (lane l: row l&31, k-slots 8(l>>5)..+7: one
//     ds_read_b128 per lane is one MFMA's A operand); x_t and h_{t-1} are B operands in registers;
//   * D tile layout: lane (env n = l&31, half hb = l>>5), register i  <->  row 8(i>>2)+4hb+(i&3).
//     Gate q of hidden unit u lives in tile 2q + u/32, row u%32: the four gates of a unit land in
//     the SAME lane and register index, so the cell update is lane-local;
//   * the K order of the recurrent product is free, so its k-slots are DEFINED as the units each
//     half-wave holds: h_t leaves the cell update in exactly the lanes whose B operand needs it —
//     no shuffle, no LDS round trip, just the fp16 split.
// 8 waves per workgroup (2 per SIMD), 256 envs per workgroup => 256 workgroups = one per CU at
// 65 536 envs.  The two waves of a SIMD run in phase, so each wave overlaps its own work instead:
// the cell update of units 0..31 is issued between the MFMAs of units 32..63 (TpCell / TpGate SIDE).
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <initializer_list>
#include <new>
#include <string>
#include <type_traits>
#include <utility>

#include "hns_device.h"
#include "hns_host.h"

namespace hns {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kTpH = HNS_TP_HIDDEN;
// waves per workgroup: 8 (2 per SIMD, 256 envs) while the operand image + the parked frames fit the CU's 160 KB of LDS with them;
// three frame chunks (33..48 values) need 123 KB of image, so that shape runs 6 waves (192 envs) per workgroup
__host__ __device__ constexpr int tp_waves(int nxc) { return nxc <= 2 ? 8 : 6; }
constexpr int kTpMaxRows = 32;              // 3F <= 32: one M tile for the output layer
constexpr int kTpMaxChunks = 5;             // 16-value chunks of a frame: 7 + 3 * 7 pursuers + 3 * 16 cylinders = 76 values
constexpr float kTpLoScale = 2048.0f;       // 2^11: the low split term, kept in fp16's normal range
constexpr float kTpLoInv = 1.0f / 2048.0f;

struct TpParams {
    hns_tp_buffers tp;
    const float *drone_state, *target_pos, *target_vel, *progress, *obs_self20, *cylinders;
    const uint8_t *detect;
    int E, A, C, I, T, F, fill, max_len;     // E counts the predictor's UNITS: envs, or (env, evader) pairs with two evaders (unit 2 e + j)
    int NT;                                  // evaders per env (1 or 2)
    float mask_value, arena_size, max_height, cylinder_size;
    unsigned long long *prof;   // diagnostics (hns_set_phase_profile): per-wave stamps of the 100 MHz clock
};

// Operand image (16-byte slots = one lane's A operand of one MFMA; then the fp32 biases).
// Slot index within a matrix: ((term * tiles + tile) * chunks + chunk) * 64 + lane, term 0 = w1, 1 = w2.
struct TpImage {
    int whh, wih, wfc, bias, bfc, slots, bytes;    // slot offsets; bias/bfc in slots too (16 B = 4 floats)
};
__host__ __device__ constexpr TpImage tp_image(int nxc) {
    TpImage L{};
    int o = 0;
    L.whh = o;  o += 2 * 8 * 4 * 64;
    L.wih = o;  o += 2 * 8 * nxc * 64;
    L.wfc = o;  o += 2 * 1 * 4 * 64;
    L.slots = o;                                   // operand slots; biases follow
    L.bias = o; o += 8 * 2 * 16 / 4;
    L.bfc = o;  o += 2 * 16 / 4;
    L.bytes = o * 16;
    return L;
}

// hidden unit that half-wave `hb` holds in register s (s = 16*tile_pair + i): D row 8(i>>2)+4hb+(i&3)
__host__ __device__ inline int tp_unit(int s, int hb) { return 32 * (s >> 4) + 8 * ((s & 15) >> 2) + 4 * hb + (s & 3); }

// The operand image holds the gate rows pre-multiplied by the constant their nonlinearity needs:
// sigmoid(z) = 1 / (1 + 2^(-z log2 e)), tanh(z) = 2 / (1 + 2^(-2 z log2 e)) - 1, so the matrix product
// delivers the exponent directly (one multiply less per gate and unit on the VALU-bound side).
constexpr float kNegLog2e = -1.4426950408889634f;
HNS_DEV float tp_gate_scale(int row) { return (row >= 2 * kTpH && row < 3 * kTpH) ? 2.0f * kNegLog2e : kNegLog2e; }   // i,f,o | g

HNS_DEV void tp_split(float w, _Float16 &hi, _Float16 &lo) {
    hi = (_Float16)w;
    lo = (_Float16)((w - (float)hi) * kTpLoScale);
}
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// ---- parameters -> operand image (run when the parameters changed) ------------------------------
__global__ __launch_bounds__(256) void hns_tp_pack_kernel(const TpParams p, int nxc) {
    const TpImage L = tp_image(nxc);
    const int I = p.I, R = 3 * p.F;
    const int n_hh = 8 * 4 * 64, n_ih = 8 * nxc * 64, n_fc = 4 * 64;
    uint4 *img = reinterpret_cast<uint4 *>(p.tp.packed);
    for (int sidx = blockIdx.x * blockDim.x + threadIdx.x; sidx < n_hh + n_ih + n_fc; sidx += gridDim.x * blockDim.x) {
        float w[8];
        int slot_hi, slot_lo;
        if (sidx < n_hh) {                                  // [tile][chunk][lane]
            const int ln = sidx & 63, c = (sidx >> 6) & 3, m = sidx >> 8;
            const int row = 32 * m + (ln & 31), hb = ln >> 5;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = p.tp.w_hh[row * kTpH + tp_unit(8 * c + j, hb)] * tp_gate_scale(row);
            slot_hi = L.whh + sidx; slot_lo = L.whh + n_hh + sidx;
        } else if (sidx < n_hh + n_ih) {
            const int u = sidx - n_hh, ln = u & 63, g = u >> 6, m = g / nxc, cx = g - m * nxc;
            const int row = 32 * m + (ln & 31), hb = ln >> 5;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * cx + 8 * hb + j;
                w[j] = k < I ? p.tp.w_ih[row * I + k] * tp_gate_scale(row) : 0.0f;
            }
            slot_hi = L.wih + u; slot_lo = L.wih + n_ih + u;
        } else {
            const int u = sidx - n_hh - n_ih, ln = u & 63, c = u >> 6;
            const int row = ln & 31, hb = ln >> 5;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = row < R ? p.tp.w_fc[row * kTpH + tp_unit(8 * c + j, hb)] * (2.0f * kNegLog2e) : 0.0f;
            slot_hi = L.wfc + u; slot_lo = L.wfc + n_fc + u;
        }
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            tp_split(w[j], a, b);
            hi[j] = a; lo[j] = b;
        }
        img[slot_hi] = *reinterpret_cast<uint4 *>(&hi);
        img[slot_lo] = *reinterpret_cast<uint4 *>(&lo);
    }
    float *bias = reinterpret_cast<float *>(img + L.bias), *bfc = reinterpret_cast<float *>(img + L.bfc);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 8 * 2 * 16 + 2 * 16; idx += gridDim.x * blockDim.x) {
        if (idx < 256) {                                    // [tile][half][16]
            const int i = idx & 15, hb = (idx >> 4) & 1, m = idx >> 5;
            const int g = 32 * m + 8 * (i >> 2) + 4 * hb + (i & 3);
            bias[idx] = (p.tp.b_ih[g] + p.tp.b_hh[g]) * tp_gate_scale(g);
        } else {
            const int i = idx & 15, hb = (idx - 256) >> 4;
            const int row = 8 * (i >> 2) + 4 * hb + (i & 3);
            bfc[idx - 256] = row < R ? p.tp.b_fc[row] * (2.0f * kNegLog2e) : 0.0f;
        }
    }
}

// gate nonlinearities on the transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each); the oracle
// uses libm, the parity tolerance is the north star's 1e-5
// arguments come pre-scaled from the matrix product: zs = -z log2 e (sigmoid), zt = -2 z log2 e (tanh)
HNS_DEV float tp_tanh_s(float zt) { return HNS_FMA(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(zt)), -1.0f); }

// component k of the frame [progress, evader pos (masked), evader vel (masked), pursuer positions]
// (hideandseek.py:815-820; the mask is broadcast_detect, :791-803), followed with task.use_obstacles by
// [x, y, cylinder_size] of every cylinder slot (:808-816; I = 7 + 3A + 3C then)
// Two evaders (extension): the SAME network runs once per evader — unit u = 2 e + j sees evader j's position / velocity (target_pos / target_vel
// are [E,2,3] = [units,3]) under evader j's detection bit, beside env e's progress, pursuers and cylinders.
HNS_DEV int tp_env(const TpParams &p, int u) { return p.NT == 2 ? u >> 1 : u; }
HNS_DEV bool tp_det(const TpParams &p, int u) { return p.NT == 2 ? ((p.detect[u >> 1] >> (u & 1)) & 1) != 0 : p.detect[u] != 0; }
HNS_DEV float tp_frame_val(const TpParams &p, int u, int k, bool det) {
    const int e = tp_env(p, u);
    if (k == 0) return p.progress[e];
    if (k < 4) return det ? p.target_pos[(size_t)u * 3 + (k - 1)] : p.mask_value;
    if (k < 7) return det ? p.target_vel[(size_t)u * 3 + (k - 4)] : p.mask_value;
    const int j = k - 7, a = j / 3;
    if (a < p.A) return p.drone_state[((size_t)e * p.A + a) * 13 + (j - 3 * a)];
    const int jc = j - 3 * p.A, cy = jc / 3, comp = jc - 3 * cy;
    return comp == 2 ? p.cylinder_size : p.cylinders[((size_t)e * p.C + cy) * 3 + comp];
}

// The same component as an ADDRESS that is always valid plus what to do with the loaded word: the weight-stationary kernel issues the loads of a thread's
// four frame values and of the detection byte together and selects afterwards — through tp_frame_val every value sat behind its own branch and its own
// wait for memory (five round trips in the prologue of every workgroup; round 6).  kind: 0 = the word, 1 = the word if the evader is detected else
// mask_value, 2 = cylinder_size, 3 = zero (beyond the frame).
HNS_DEV const float *tp_frame_addr(const TpParams &p, int u, int k, int &kind) {
    const int e = tp_env(p, u);
    const float *a = p.progress + e;                       // k = 0, and the stand-in address of the kinds that do not use the word
    kind = k >= p.I ? 3 : 0;
    if (k >= 1 && k < 7) { kind = 1; a = (k < 4 ? p.target_pos + (size_t)u * 3 + (k - 1) : p.target_vel + (size_t)u * 3 + (k - 4)); }
    if (k >= 7 && k < p.I) {
        const int j = k - 7, ag = j / 3;
        if (ag < p.A) a = p.drone_state + ((size_t)e * p.A + ag) * 13 + (j - 3 * ag);
        else {
            const int jc = j - 3 * p.A, cy = jc / 3, comp = jc - 3 * cy;
            if (comp == 2) kind = 2; else a = p.cylinders + ((size_t)e * p.C + cy) * 3 + comp;
        }
    }
    return a;
}

// Two evaders: rows of 24 + 6F values = [the reference's 20 + 3F row for evader 0 | rpos of evader 1 (3) | 0 | drone - predicted evader 1 (3F)];
// groundtruth / tp_done per unit.  Not tuned (plain stores): the shape is an extension.
HNS_DEV void tp_write_row2(const TpParams &p, int ev, int a, const float *pr0, const float *pr1) {
    const int A = p.A, R = 3 * p.F, SD = 24, D = SD + 2 * R;
    const size_t ia = (size_t)ev * A + a;
    const float *o24 = p.obs_self20 + ia * SD, *ds = p.drone_state + ia * 13, *tg = p.target_pos + (size_t)ev * 6;
    const float px = ds[0], py = ds[1], pz = ds[2];
    if (a == 0) {
        for (int j = 0; j < 2; ++j) {
            float *gt = p.tp.groundtruth + ((size_t)ev * 2 + j) * 3;
            gt[0] = tg[3 * j] * (1.0f / (0.5f * p.arena_size));
            gt[1] = tg[3 * j + 1] * (1.0f / (0.5f * p.arena_size));
            gt[2] = (tg[3 * j + 2] * (1.0f / p.max_height)) * 2.0f - 1.0f;
            p.tp.tp_done[(size_t)ev * 2 + j] = (uint8_t)(p.progress[ev] <= (float)(p.max_len - p.F));
        }
    }
    for (int pass = 0; pass < 2; ++pass) {               // 0: state_self (masked rpos from the step kernel's rows), 1: state_drones (unmasked)
        float *dst = pass == 0 ? p.tp.obs_self : p.tp.state_drones;
        if (!dst) continue;
        float *g = dst + ia * D;
        g[0] = pass == 0 ? o24[0] : px - tg[0]; g[1] = pass == 0 ? o24[1] : py - tg[1]; g[2] = pass == 0 ? o24[2] : pz - tg[2];
        for (int f = 0; f < p.F; ++f) { g[3 + 3 * f] = px - pr0[3 * f]; g[4 + 3 * f] = py - pr0[3 * f + 1]; g[5 + 3 * f] = pz - pr0[3 * f + 2]; }
        for (int j = 3; j < HNS_SELF_DIM; ++j) g[R + j] = o24[j];
        g[R + 20] = pass == 0 ? o24[20] : px - tg[3]; g[R + 21] = pass == 0 ? o24[21] : py - tg[4]; g[R + 22] = pass == 0 ? o24[22] : pz - tg[5];
        g[R + 23] = o24[23];
        for (int f = 0; f < p.F; ++f) { g[R + 24 + 3 * f] = px - pr1[3 * f]; g[R + 25 + 3 * f] = py - pr1[3 * f + 1]; g[R + 26 + 3 * f] = pz - pr1[3 * f + 2]; }
    }
}

// ---- one pursuer's observation row: [rpos_evader(3) | drone - predicted (3F) | quat4 linvel3 heading3 up3 t x4 (17)] ----
// hideandseek.py:844-854 (state_self, masked rpos from the step kernel's rows) and :873-880 (state_drones, unmasked rpos);
// TP_groundtruth / TP_done :838-842.  `pr` = the env's 3F predictions (LDS).
HNS_DEV void tp_write_row(const TpParams &p, int er, int a, const float *pr, const float *pr1) {
    if (p.NT == 2) { tp_write_row2(p, er, a, pr, pr1); return; }
    typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));       // rows are 4-byte aligned (D = 20 + 3F floats)
    const int A = p.A, R = 3 * p.F, D = HNS_SELF_DIM + R;
    const size_t ia = (size_t)er * A + a;
    float o20[HNS_SELF_DIM];
    const float4 *s4 = reinterpret_cast<const float4 *>(p.obs_self20 + ia * HNS_SELF_DIM);
#pragma unroll
    for (int v = 0; v < HNS_SELF_DIM / 4; ++v) {
        const float4 q = s4[v];
        o20[4 * v] = q.x; o20[4 * v + 1] = q.y; o20[4 * v + 2] = q.z; o20[4 * v + 3] = q.w;
    }
    const float *ds = p.drone_state + ia * 13, *tg = p.target_pos + (size_t)er * 3;
    const float px = ds[0], py = ds[1], pz = ds[2], tx = tg[0], ty = tg[1], tz = tg[2];
    if (a == 0) {
        // CUDA scalar-division form: tensor / python_scalar multiplies by the fp32 reciprocal
        float *gt = p.tp.groundtruth + (size_t)er * 3;
        gt[0] = tx * (1.0f / (0.5f * p.arena_size));
        gt[1] = ty * (1.0f / (0.5f * p.arena_size));
        gt[2] = (tz * (1.0f / p.max_height)) * 2.0f - 1.0f;
        p.tp.tp_done[er] = (uint8_t)(p.progress[er] <= (float)(p.max_len - p.F));
    }
    for (int pass = 0; pass < 2; ++pass) {               // 0: state_self, 1: state_drones
        float *dst = pass == 0 ? p.tp.obs_self : p.tp.state_drones;
        if (!dst) continue;
        float *g = dst + ia * D;
        const float r0 = pass == 0 ? o20[0] : px - tx, r1 = pass == 0 ? o20[1] : py - ty, r2 = pass == 0 ? o20[2] : pz - tz;
        if (R == 15) {                                    // five predicted points: 35 floats = 8 x 16 B + 3
            float w[36];
            w[0] = r0; w[1] = r1; w[2] = r2;
#pragma unroll
            for (int f = 0; f < 5; ++f) { w[3 + 3 * f] = px - pr[3 * f]; w[4 + 3 * f] = py - pr[3 * f + 1]; w[5 + 3 * f] = pz - pr[3 * f + 2]; }
#pragma unroll
            for (int j = 3; j < HNS_SELF_DIM; ++j) w[15 + j] = o20[j];
#pragma unroll
            for (int v = 0; v < 8; ++v) *reinterpret_cast<f4u *>(g + 4 * v) = (f4u){w[4 * v], w[4 * v + 1], w[4 * v + 2], w[4 * v + 3]};
            g[32] = w[32]; g[33] = w[33]; g[34] = w[34];
        } else {
            g[0] = r0; g[1] = r1; g[2] = r2;
            for (int f = 0; f < p.F; ++f) { g[3 + 3 * f] = px - pr[3 * f]; g[4 + 3 * f] = py - pr[3 * f + 1]; g[5 + 3 * f] = pz - pr[3 * f + 2]; }
#pragma unroll
            for (int j = 3; j < HNS_SELF_DIM; ++j) g[R + j] = o20[j];
        }
    }
}

#define TP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// ---- the gate pre-activations of 32 hidden units (4 gate tiles) for one timestep ------------------
// Op list (n ascending): cross terms  — per k-chunk (frame chunks, then the 4 recurrent chunks):
//                                        4 tiles x {w1·v_lo, w2·v_hi}
//                        scale + bias — acc = 2^-11 acc + b
//                        leading term — per k-chunk: 4 tiles x w1·v_hi
// One ds_read_b128 per MFMA; the reads run kTpDepth ops ahead of the matrix pipe through a register
// ring, and every index below is a template constant (a plain unrolled loop over the op list ended
// up with dynamically indexed register arrays in scratch memory for some shapes: 20x slower).
// The kernel as a whole is bound by the gate nonlinearities (10 transcendental ops per unit and
// timestep), not by this block.
constexpr int kTpDepth = 2;

// A unit pair's values.  0 = 1 keeps them as packed fp32 (v_pk_add/mul/fma_f32); the default issues two plain instructions per
// operation: on gfx950 a packed fp32 instruction occupies a lone wave's issue for 11 cycles (two plain ones: 2 x 5.2-6), saves
// nothing at 4 waves per SIMD either (0.177 against 2 x 0.19-0.27 instructions per cycle), and beside a running MFMA it waits for
// the matrix pipe — one packed instruction per MFMA and wave, where plain and transcendental ones still find 2-3 issue slots
// (tools/microbench/simd_share.hip; round 3).
struct tpv2 {
    float x, y;
    __device__ __forceinline__ float operator[](int i) const { return i ? y : x; }
};
HNS_DEV tpv2 tp_v2(float a, float b) { return tpv2{a, b}; }
HNS_DEV tpv2 tp_add_s(tpv2 a, float s) { return tpv2{a.x + s, a.y + s}; }
HNS_DEV tpv2 tp_mul_s(tpv2 a, float s) { return tpv2{a.x * s, a.y * s}; }
HNS_DEV tpv2 tp_mul_2(tpv2 a, tpv2 b) { return tpv2{a.x * b.x, a.y * b.y}; }
HNS_DEV tpv2 tp_sub_2(tpv2 a, tpv2 b) { return tpv2{a.x - b.x, a.y - b.y}; }
HNS_DEV tpv2 tp_fma_2(tpv2 a, tpv2 b, tpv2 c) { return tpv2{HNS_FMA(a.x, b.x, c.x), HNS_FMA(a.y, b.y, c.y)}; }
HNS_DEV tpv2 tp_fma_s(tpv2 a, float b, float c) { return tpv2{HNS_FMA(a.x, b, c), HNS_FMA(a.y, b, c)}; }
// v_exp_f32 / v_rcp_f32 on both halves of a pair (transcendentals have no packed form)
HNS_DEV tpv2 tp_exp2_2(tpv2 v) { return tp_v2(__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])); }
HNS_DEV tpv2 tp_rcp_2(tpv2 v) { return tp_v2(__builtin_amdgcn_rcpf(v[0]), __builtin_amdgcn_rcpf(v[1])); }
// fp16 split of a pair: hi = fp16(v), lo = fp16((v - hi) 2^11)
HNS_DEV void tp_split_v2(tpv2 w, half2v &hi, half2v &lo) {
    hi = __builtin_convertvector((f32x2){w[0], w[1]}, half2v);
    const tpv2 r = tp_mul_s(tp_sub_2(w, tp_v2((float)hi[0], (float)hi[1])), kTpLoScale);
    lo = __builtin_convertvector((f32x2){r[0], r[1]}, half2v);
}

// ---- cell update of the 16 units of one tile pair (torch.nn.LSTM gate order i, f, g, o), lane-local ----
// Cut into 40 slices (8 unit pairs x 5 stages of <= 8 VALU ops) so that the update of tile pair 0 can be
// issued BETWEEN the MFMAs of tile pair 1 (same wave, independent registers): left to themselves the two
// waves of a SIMD run in phase — both in their MFMA block, then both in their nonlinearities — and the
// matrix pipe and the VALU alternate instead of overlapping (counters: 36 % + 53 % of the kernel).
struct TpCellCtx {
    const f32x16 (&z)[4];     // pre-activations, already scaled to the exponents the nonlinearities need
    float (&c)[32];           // cell state, units 16 TJ + i
    float (&h)[16];           // TJ = 0: h_t of these units, parked while tile pair 1 still reads h_{t-1}
    half8 (&hh)[4];           // TJ = 1: h_{t-1} is dead by then, h_t goes straight into the next B operands
    half8 (&hl)[4];
    tpv2 t[5];                // the unit pair in flight
};
template <int TJ>
struct TpCell {
    static constexpr int N = 40;
    template <int K>
    static __device__ __forceinline__ void slice(TpCellCtx &x) {
        constexpr int S = K % 5, u0 = 2 * (K / 5), u1 = u0 + 1;
        tpv2 *t = x.t;
        if constexpr (S == 0) {            // 1 + e_i, 1 + e_f
            t[0] = tp_add_s(tp_exp2_2(tp_v2(x.z[0][u0], x.z[0][u1])), 1.0f);
            t[1] = tp_add_s(tp_exp2_2(tp_v2(x.z[1][u0], x.z[1][u1])), 1.0f);
        } else if constexpr (S == 1) {     // 1 + e_g; i, f
            t[2] = tp_add_s(tp_exp2_2(tp_v2(x.z[2][u0], x.z[2][u1])), 1.0f);
            t[0] = tp_rcp_2(t[0]);
            t[1] = tp_rcp_2(t[1]);
        } else if constexpr (S == 2) {     // g = tanh; c' = f c + i g; e_c
            const tpv2 g = tp_fma_s(tp_rcp_2(t[2]), 2.0f, -1.0f);
            const tpv2 cn = tp_fma_2(t[1], tp_v2(x.c[16 * TJ + u0], x.c[16 * TJ + u1]), tp_mul_2(t[0], g));
            x.c[16 * TJ + u0] = cn[0]; x.c[16 * TJ + u1] = cn[1];
            t[3] = tp_exp2_2(tp_mul_s(cn, 2.0f * kNegLog2e));
        } else if constexpr (S == 3) {     // o; 1 + e_c
            t[4] = tp_rcp_2(tp_add_s(tp_exp2_2(tp_v2(x.z[3][u0], x.z[3][u1])), 1.0f));
            t[3] = tp_add_s(t[3], 1.0f);
        } else {                           // h' = o tanh(c')
            const tpv2 h = tp_mul_2(t[4], tp_fma_s(tp_rcp_2(t[3]), 2.0f, -1.0f));
            if constexpr (TJ == 0) {
                x.h[u0] = h[0]; x.h[u1] = h[1];
            } else {
                half2v a, b;
                tp_split_v2(h, a, b);
                x.hh[2 + (u0 >> 3)][u0 & 7] = a[0]; x.hl[2 + (u0 >> 3)][u0 & 7] = b[0];
                x.hh[2 + (u1 >> 3)][u1 & 7] = a[1]; x.hl[2 + (u1 >> 3)][u1 & 7] = b[1];
            }
        }
    }
    template <int... Ks>
    static __device__ __forceinline__ void run(TpCellCtx &x, std::integer_sequence<int, Ks...>) { (slice<Ks>(x), ...); }
    static __device__ __forceinline__ void run_all(TpCellCtx &x) { run(x, std::make_integer_sequence<int, N>{}); }
};

template <int NXC, bool WITH_H, bool SIDE = false>
struct TpGate {
    static constexpr int NC = NXC + (WITH_H ? 4 : 0), NLO = 8 * NC, NHI = 4 * NC, N = NLO + NHI;
    static constexpr int D = kTpDepth < N ? kTpDepth : N;
    // A slot is refilled only after a LATER MFMA ON THE SAME ACCUMULATOR has issued (ops n and n+4 share
    // a tile): that one cannot issue before the slot's last reader has completed.  Refilling a source
    // register right behind its MFMA corrupts the operand on gfx950 when the matrix pipe is shared with
    // a partner wave (measured: 2e-5 errors = cross terms picking up the next op's low split term).
    static constexpr int RING = D + 4;
    static constexpr int chunk(int n) { return n < NLO ? n / 8 : (n - NLO) / 4; }
    static constexpr int tile(int n) { return n < NLO ? (n % 8) % 4 : (n - NLO) % 4; }
    static constexpr bool w2(int n) { return n < NLO && (n % 8) / 4 == 1; }       // A = low split term of W
    static constexpr bool vlo(int n) { return n < NLO && (n % 8) / 4 == 0; }      // B = low split term of v
    // slot(n) = base + tj * stride (in 16-byte slots, before the lane offset)
    static constexpr int slot_base(const TpImage &L, int n) {
        const int ci = chunk(n), q = tile(n);
        return ci < NXC ? L.wih + (w2(n) ? 8 * NXC * 64 : 0) + (2 * q * NXC + ci) * 64
                        : L.whh + (w2(n) ? 8 * 4 * 64 : 0) + (2 * q * 4 + (ci - NXC)) * 64;
    }
    static constexpr int slot_stride(int n) { return (chunk(n) < NXC ? NXC : 4) * 64; }

    struct Ctx {
        f32x16 (&acc)[4];
        half8 (&a)[RING];
        const uint4 *aw;          // image + lane
        const float *bias;        // sBias + hb * 16
        int tj;
        const half8 (&xh)[NXC];
        const half8 (&xl)[NXC];
        const half8 (&hh)[4];
        const half8 (&hl)[4];
        TpCellCtx *side;          // SIDE: the other tile pair's cell update, one slice per MFMA
    };

    template <int n>
    static __device__ __forceinline__ half8 load(const Ctx &c) {
        constexpr TpImage L = tp_image(NXC);
        return *reinterpret_cast<const half8 *>(c.aw + slot_base(L, n) + c.tj * slot_stride(n));
    }
    template <int n>
    static __device__ __forceinline__ void prologue(const Ctx &c) {
        c.a[n % RING] = load<n>(c);
        __builtin_amdgcn_sched_barrier(0);
    }

    static __device__ __forceinline__ void scale_bias(const Ctx &c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 *b4 = reinterpret_cast<const float4 *>(c.bias + (2 * q + c.tj) * 32);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 bv = b4[v];
                c.acc[q][4 * v] = HNS_FMA(c.acc[q][4 * v], kTpLoInv, bv.x);
                c.acc[q][4 * v + 1] = HNS_FMA(c.acc[q][4 * v + 1], kTpLoInv, bv.y);
                c.acc[q][4 * v + 2] = HNS_FMA(c.acc[q][4 * v + 2], kTpLoInv, bv.z);
                c.acc[q][4 * v + 3] = HNS_FMA(c.acc[q][4 * v + 3], kTpLoInv, bv.w);
            }
        }
    }

    template <int n>
    static __device__ __forceinline__ void step(const Ctx &c) {
        if constexpr (n == NLO) scale_bias(c);
        constexpr int ci = chunk(n), q = tile(n);
        if constexpr (ci < NXC) {
            if constexpr (vlo(n)) c.acc[q] = TP_MFMA(c.a[n % RING], c.xl[ci], c.acc[q]);
            else c.acc[q] = TP_MFMA(c.a[n % RING], c.xh[ci], c.acc[q]);
        } else {
            if constexpr (vlo(n)) c.acc[q] = TP_MFMA(c.a[n % RING], c.hl[ci - NXC], c.acc[q]);
            else c.acc[q] = TP_MFMA(c.a[n % RING], c.hh[ci - NXC], c.acc[q]);
        }
        if constexpr (n + D < N) c.a[(n + D) % RING] = load<n + D>(c);
        if constexpr (SIDE && n < TpCell<0>::N) TpCell<0>::slice<n>(*c.side);
        __builtin_amdgcn_sched_barrier(0);       // keep the hand-placed MFMA / LDS-read order (a 0x6 mask lets MFMAs move: they count as VALU)
    }
    template <int... Ns>
    static __device__ __forceinline__ void run_prologue(const Ctx &c, std::integer_sequence<int, Ns...>) { (prologue<Ns>(c), ...); }
    template <int... Ns>
    static __device__ __forceinline__ void run_steps(const Ctx &c, std::integer_sequence<int, Ns...>) { (step<Ns>(c), ...); }
};

template <int NXC, bool WITH_H, bool SIDE = false>
HNS_DEV void tp_gate_tiles(f32x16 (&acc)[4], const uint4 *aw, int tj, int hb, const float *sBias,
                           const half8 (&xh)[NXC], const half8 (&xl)[NXC], const half8 (&hh)[4], const half8 (&hl)[4],
                           TpCellCtx *side = nullptr) {
    using G = TpGate<NXC, WITH_H, SIDE>;
    static_assert(!SIDE || G::N >= TpCell<0>::N, "not enough MFMAs to carry the cell update");
    half8 a[G::RING];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.0f;
    const typename G::Ctx c{acc, a, aw, sBias + hb * 16, tj, xh, xl, hh, hl, side};
    G::run_prologue(c, std::make_integer_sequence<int, G::D>{});
    G::run_steps(c, std::make_integer_sequence<int, G::N>{});
}

// NXC = 16-wide k-chunks of the frame (1: I <= 16, i.e. up to 3 pursuers; 2: I <= 32; 3: I <= 48, e.g. 3 pursuers + 8 cylinders
// with task.use_obstacles, hideandseek.py:808-816)
template <int NXC>
__global__ __launch_bounds__(tp_waves(NXC) * 64) void hns_tp_lstm_kernel(const TpParams p) {
    constexpr int kTpWaves = tp_waves(NXC), kTpThreads = kTpWaves * 64, kTpEnvs = kTpWaves * 32;
    extern __shared__ __align__(16) uint4 simg[];
    const TpImage L = tp_image(NXC);
    const int tid = threadIdx.x, I = p.I, T = p.T, R = 3 * p.F;
    // (the wave index through readfirstlane: the stamp pointer is wave-uniform and belongs in scalar registers — as a per-lane value it was the
    //  one thing the two-chunk instantiation spilled across the timestep loop, 3 registers / 16 B of scratch)
    unsigned long long *prof = p.prof ? p.prof + (size_t)(blockIdx.x * kTpWaves + __builtin_amdgcn_readfirstlane(tid >> 6)) * 16 : nullptr;
    if (prof && (tid & 63) == 0) prof[0] = __builtin_amdgcn_s_memrealtime();

    // ---- stage the operand image (88-104 KB, L2-resident, contiguous) ------------------------------
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(p.tp.packed);
        const int n = L.bytes / 16;
        for (int i = tid; i < n; i += kTpThreads) simg[i] = src[i];
    }
    __syncthreads();
    if (prof && (tid & 63) == 0) prof[1] = __builtin_amdgcn_s_memrealtime();

    const int wave = tid >> 6, lane = tid & 63, hb = lane >> 5;
    const int e = blockIdx.x * kTpEnvs + wave * 32 + (lane & 31);
    if ((int)blockIdx.x * kTpEnvs + wave * 32 >= p.E) return;     // whole wave out of range
    const bool valid = e < p.E;
    const int ec = valid ? e : p.E - 1;                           // clamped: loads stay in bounds, stores are guarded
    const float *sBias = reinterpret_cast<const float *>(simg + L.bias), *sBfc = reinterpret_cast<const float *>(simg + L.bfc);
    constexpr int N_FC = 4 * 64;

    // this lane's part of a frame: k = 16 cx + 8 hb + j
    // parked in LDS (read once, as x_{T-1}): [wave][value][lane], conflict-free
    float *sXn = reinterpret_cast<float *>(simg + L.bytes / 16) + (wave * 8 * NXC) * 64 + lane;
    {
        const bool det = tp_det(p, ec);
#pragma unroll
        for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * cx + 8 * hb + j;
                sXn[(8 * cx + j) * 64] = k < I ? tp_frame_val(p, ec, k, det) : 0.0f;
            }
    }
    float *hist = p.tp.history + (size_t)ec * T * I;       // this env's window, [T][I]
    // Window rows are read with unconditional loads (one wait for all of them; per-element predicated loads cost
    // the timestep loop ~4 000 cycles of serialised memory latency per iteration): 16-byte global loads when the
    // frame fills its chunks exactly (I = 16 NXC, e.g. 3 pursuers).  Otherwise rows are only 4-byte aligned and
    // shorter than their chunks: raw BUFFER loads/stores through a descriptor over this workgroup's part of the
    // window — one lane offset + immediates instead of 16 clamped 64-bit addresses (that version spilled 15-39
    // registers), the hardware range check returns 0 past the end of the buffer, and values past the end of
    // the ROW (they belong to the next row) are zeroed / not stored by the k < I selects below.
    const bool vec = (I == 16 * NXC);
    const int e_wg = blockIdx.x * kTpEnvs;
    const long long wg_bytes = (long long)(p.E - e_wg) * T * I * 4;
    const __amdgpu_buffer_rsrc_t hrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.tp.history + (size_t)e_wg * T * I), 0, (int)(wg_bytes < (long long)kTpEnvs * T * I * 4 ? wg_bytes : (long long)kTpEnvs * T * I * 4),
        0x00020000);
    const int lane_off = (((ec - e_wg) * T) * I + 8 * hb) * 4;            // bytes from the descriptor's base
    auto load_row = [&](int slot, float (&dst)[8 * NXC]) {
        if (vec) {
            const float *row = hist + (size_t)slot * I;
#pragma unroll
            for (int cx = 0; cx < NXC; ++cx) {
                const float4 a = *reinterpret_cast<const float4 *>(row + 16 * cx + 8 * hb);
                const float4 b = *reinterpret_cast<const float4 *>(row + 16 * cx + 8 * hb + 4);
                dst[8 * cx] = a.x; dst[8 * cx + 1] = a.y; dst[8 * cx + 2] = a.z; dst[8 * cx + 3] = a.w;
                dst[8 * cx + 4] = b.x; dst[8 * cx + 5] = b.y; dst[8 * cx + 6] = b.z; dst[8 * cx + 7] = b.w;
            }
        } else {
            const int off = lane_off + slot * I * 4;
#pragma unroll
            for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrsrc, off + (16 * cx + j) * 4, 0, 0));
                    dst[8 * cx + j] = (16 * cx + 8 * hb + j) < I ? v : 0.0f;
                }
        }
    };
    auto store_row = [&](int slot, const float (&src)[8 * NXC]) {
        if (vec) {
            float *row = hist + (size_t)slot * I;
#pragma unroll
            for (int cx = 0; cx < NXC; ++cx) {
                *reinterpret_cast<float4 *>(row + 16 * cx + 8 * hb) = make_float4(src[8 * cx], src[8 * cx + 1], src[8 * cx + 2], src[8 * cx + 3]);
                *reinterpret_cast<float4 *>(row + 16 * cx + 8 * hb + 4) = make_float4(src[8 * cx + 4], src[8 * cx + 5], src[8 * cx + 6], src[8 * cx + 7]);
            }
        } else {
            const int off = lane_off + slot * I * 4;
#pragma unroll
            for (int cx = 0; cx < NXC; ++cx) {
                if (16 * cx + 16 <= I) {                  // both half-waves' groups lie inside the row (uniform)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, src[8 * cx + j]), hrsrc, off + (16 * cx + j) * 4, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (16 * cx + 8 * hb + j < I)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, src[8 * cx + j]), hrsrc, off + (16 * cx + j) * 4, 0, 0);
                }
            }
        }
    };
    // x_t = old frame t+1 for t <= T-2, the new frame for t = T-1 (or for every t when filling)
    float xc[8 * NXC];
    if (T == 1 || p.fill) {
#pragma unroll
        for (int i = 0; i < 8 * NXC; ++i) xc[i] = sXn[i * 64];
    } else {
        load_row(1, xc);
    }

    half8 hh[4], hl[4];                     // h_{t-1}: leading and low split terms, k-slot j of chunk c = own register 8c + j
    float c[32];
#pragma unroll
    for (int i = 0; i < 4; ++i) { hh[i] = (half8)(_Float16)0.0f; hl[i] = (half8)(_Float16)0.0f; }
#pragma unroll
    for (int i = 0; i < 32; ++i) c[i] = 0.0f;
    float hn0[16];                          // h_t of tile pair 0's units, parked while tile pair 1 still reads h_{t-1}

    // one timestep; t = 0 (h_0 = 0: no recurrent product) is peeled so that the loop body is straight-line code
    auto timestep = [&](int t, auto with_h) {
        constexpr bool WITH_H = decltype(with_h)::value;
        // shift x_t into slot t and split it; prefetch x_{t+1} (slot t+2 of the old window, untouched so far)
        half8 xh[NXC], xl[NXC];
#pragma unroll
        for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                half2v a, b;
                tp_split_v2(tp_v2(xc[8 * cx + j], xc[8 * cx + j + 1]), a, b);
                xh[cx][j] = a[0]; xl[cx][j] = b[0];
                xh[cx][j + 1] = a[1]; xl[cx][j + 1] = b[1];
            }
        if (valid) store_row(t, xc);
        if (!p.fill && t + 1 <= T - 2) {            // consumed one timestep later: the latency is off the critical path
            load_row(t + 2, xc);
        } else {
#pragma unroll
            for (int i = 0; i < 8 * NXC; ++i) xc[i] = sXn[i * 64];
        }
        // the operand image is loop-invariant; an opaque lane offset keeps the compiler from hoisting
        // the A operands of the whole window (hundreds of registers) out of the timestep loop
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const uint4 *aw = simg + lo;
        // units 0..31 (gate tiles m = 2q) then 32..63 (m = 2q + 1); the cell update of the first half rides between
        // the MFMAs of the second (not at t = 0: too few MFMAs without the recurrent product)
        f32x16 acc0[4], acc1[4];
        TpCellCtx cell0{acc0, c, hn0, hh, hl, {}}, cell1{acc1, c, hn0, hh, hl, {}};
        tp_gate_tiles<NXC, WITH_H>(acc0, aw, 0, hb, sBias, xh, xl, hh, hl);
        constexpr bool SIDE = WITH_H;
        if constexpr (SIDE) {
            tp_gate_tiles<NXC, WITH_H, true>(acc1, aw, 1, hb, sBias, xh, xl, hh, hl, &cell0);
        } else {
            TpCell<0>::run_all(cell0);
            __builtin_amdgcn_sched_barrier(0);            // acc0 is dead before acc1 goes live
            tp_gate_tiles<NXC, WITH_H>(acc1, aw, 1, hb, sBias, xh, xl, hh, hl);
        }
        TpCell<1>::run_all(cell1);
        __builtin_amdgcn_sched_barrier(0);
        // h_{t-1} is dead now: h_t -> B operands of the next timestep
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            half2v a, b;
            tp_split_v2(tp_v2(hn0[i], hn0[i + 1]), a, b);
            hh[i >> 3][i & 7] = a[0]; hl[i >> 3][i & 7] = b[0];
            hh[i >> 3][(i & 7) + 1] = a[1]; hl[i >> 3][(i & 7) + 1] = b[1];
        }
    };
    timestep(0, std::false_type{});
    for (int t = 1; t < T; ++t) timestep(t, std::true_type{});

    if (prof && lane == 0) prof[2] = __builtin_amdgcn_s_memrealtime();
    // ---- output layer on h_T: tanh(W_fc h + b), rescaled to arena units (hideandseek.py:834-836) ----
    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.0f;
    {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const uint4 *aw = simg + lo;
#define TP_A(base, idx) (*reinterpret_cast<const half8 *>(aw + (base) + (idx) * 64))
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            o = TP_MFMA(TP_A(L.wfc, ch), hl[ch], o);
            o = TP_MFMA(TP_A(L.wfc + N_FC, ch), hh[ch], o);
        }
        const float4 *b4 = reinterpret_cast<const float4 *>(sBfc + hb * 16);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 bv = b4[v];
            o[4 * v] = HNS_FMA(o[4 * v], kTpLoInv, bv.x);
            o[4 * v + 1] = HNS_FMA(o[4 * v + 1], kTpLoInv, bv.y);
            o[4 * v + 2] = HNS_FMA(o[4 * v + 2], kTpLoInv, bv.z);
            o[4 * v + 3] = HNS_FMA(o[4 * v + 3], kTpLoInv, bv.w);
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) o = TP_MFMA(TP_A(L.wfc, ch), hh[ch], o);
    }
#undef TP_A
    // the wave's 32 predictions also go to LDS ([env][16], the parked-frame area is free by now): the rows below need all of an env's
    // 3F values in one lane
    // the predictions of this wave's 32 envs, 16 values apart (five predicted points, the reference's horizon) or 32 apart (3F > 16: up to ten points).
    // 32 x 32 values fit the wave's parked-frame slot from two chunks on; with one chunk they take a slot of their own behind all parked frames
    const int ps = R > 16 ? 32 : 16;
    float *sPred = reinterpret_cast<float *>(simg + L.bytes / 16) + ((NXC == 1 && R > 16) ? kTpWaves * 8 * 64 + wave * 1024 : (wave * 8 * NXC) * 64);
    {
        float *pr = p.tp.pred + (size_t)ec * R;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 8 * (i >> 2) + 4 * hb + (i & 3);
            if (row < R) {
                const float v = tp_tanh_s(o[i]);
                const int comp = row % 3;
                const float val = (comp < 2) ? (v * 0.5f) * p.arena_size : ((v + 1.0f) * 0.5f) * p.max_height;
                if (valid) pr[row] = val;
                sPred[(lane & 31) * ps + row] = val;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (prof && lane == 0) prof[3] = __builtin_amdgcn_s_memrealtime();
    // ---- observation rows of this wave's envs (tp_write_row): one lane per pursuer row, 32 A rows per wave, contiguous in every
    // buffer.  Was a second kernel (10 us + a launch); here the stores of one workgroup drain under the recurrences of the others.
    {
        // (two evaders: the wave's 32 units are 16 envs, units 2 el and 2 el + 1 hold an env's two predictions)
        const int A = p.A, NT = p.NT, ev_w0 = (blockIdx.x * kTpEnvs + wave * 32) / NT;
        for (int r = lane; r < (32 / NT) * A; r += 64) {
            const int el = r / A, a = r - el * A, ev = ev_w0 + el;
            if (ev * NT < p.E) tp_write_row(p, ev, a, sPred + (el * NT) * ps, sPred + (el * NT + NT - 1) * ps);
        }
    }
    if (prof && lane == 0) prof[4] = __builtin_amdgcn_s_memrealtime();
}

// =================================================================================================
// Weight-stationary predictor (round 3) — hns_tp_lstm_ws_kernel
// =================================================================================================
// What bounds the tile kernel above (measured, tools/lab/lab_batch60-61, tools/microbench/simd_share.hip): with the
// nonlinearities compiled out it takes 55 us, with the matrix products compiled out 94 us — the full kernel 92-97.  It is
// bound by VALU ISSUE, not by the matrix pipe: a lone wave issues one VALU instruction per 5-6 cycles (9 transcendental),
// two waves per SIMD cannot fill a VALU that takes one wave-instruction per 2 cycles, and its 256-register waves
// (two accumulator sets, cell state and both operand images of h per wave) leave room for two.
// This kernel turns the roles around so that FOUR waves fit a SIMD:
//   * a wave owns 32 gate ROWS (the four gates of 8 hidden units) of the weight matrix for the whole kernel: its A operands
//     (two split terms x (frame chunks + 4 recurrent chunks) x 4 registers = 40 registers for 3 pursuers) are loaded once;
//     8 such waves = one workgroup = all 256 gate rows, serving 128 envs (4 column tiles of 32); two workgroups per CU;
//   * the activations are the moving operand: x_t and h_{t-1} sit in LDS in B-operand order (hi and lo split terms);
//     every wave reads the same 10 operands per column tile, issues 15 MFMAs into ONE 16-register accumulator, runs the
//     cell update of its 8 units x 32 envs on it (lane-local again: local row 8 gate + unit, so a lane holds the four
//     gates of four units) and publishes its slice of h_t; one workgroup barrier pair per timestep;
//   * LDS traffic 640 KB per CU and timestep (960 KB for the tile kernel's weight image reads), 49 KB per workgroup.
// Split terms are UNSCALED here (lo = fp16(v - fp16(v)), subnormal where it must be: v_mfma_f32_32x32x16_f16 keeps fp16
// subnormals, checked in simd_share.hip): hi*hi, hi*lo and lo*hi accumulate into one chain that starts at the bias —
// no scaling pass, no bias pass.  The one operand whose magnitude would make the weights' low term matter, `progress`
// (up to max_episode_length), enters as progress/1024 against its weight column x 1024 (both exact).
constexpr int kWsWaves = 8, kWsThreads = kWsWaves * 64, kWsTiles = 4;
// Column tiles (of 32 units) a workgroup serves — the template parameter TILES of the kernel.  4 is the kernel described above (128 units per workgroup:
// the weights are fetched once per 128 units, two workgroups share a CU).  Below 32 768 units that grid is smaller than the chip (the reference's own
// default, 2 048 envs: 16 workgroups on 256 CUs) and a launch lasts as long as ONE workgroup's recurrence, 8.4 k cycles per timestep of which 5.4 k are its
// four tiles one after the other (profiles/r06_tp_where_the_time_goes.txt, C) — so small batches get 2 or 1 tiles per workgroup (round 6): the same tile
// arithmetic (results bit-identical, tests/test_hip_tp.py), twice / four times the workgroups, half / a quarter of the serial stream.
__host__ __device__ constexpr int ws_envs(int tiles) { return 32 * tiles; }
constexpr float kWsProgScale = 1024.0f, kWsProgInv = 1.0f / 1024.0f;

struct WsImage { int a, wfc, bias, bfc, slots; };     // offsets in 16-byte slots
__host__ __device__ constexpr WsImage ws_image(int nxc) {
    WsImage L{};
    int o = 0;
    L.a = o;    o += 8 * 2 * (nxc + 4) * 64;          // [row slice][term][chunk][lane]
    L.wfc = o;  o += 2 * 4 * 64;                      // [term][chunk][lane]
    L.bias = o; o += 256 / 4;                         // [row slice][half][16] floats
    L.bfc = o;  o += 32 / 4;                          // [half][16] floats
    L.slots = o;
    return L;
}
// hidden unit in k-slot sp (0..7) of lane half hb in recurrent chunk c: the first four come from row slice 2c, the others from 2c+1
__host__ __device__ inline int ws_unit(int c, int hb, int sp) { return 16 * c + 8 * (sp >> 2) + 4 * hb + (sp & 3); }

HNS_DEV void ws_split(float w, _Float16 &hi, _Float16 &lo) {
    // `w` is made opaque first: when it is a product, hipcc folds the multiply into ONE of the two uses of fp16(w) below
    // (v_fma_mixlo_f16: a single rounding of the exact product) and converts the fp32 product for the other (v_cvt_pk_f16_f32);
    // where the fp32 value sits exactly between two fp16 values the two disagree by an fp16 ulp and hi + lo is no longer w
    // (found as 1-3 envs in 512 off by 1e-5, tools/tp_debug3.py)
    asm volatile("" : "+v"(w));
    hi = (_Float16)w;
    lo = (_Float16)(w - (float)hi);
}

__global__ __launch_bounds__(256) void hns_tp_pack_ws_kernel(const TpParams p, int nxc) {
    const WsImage L = ws_image(nxc);
    const int I = p.I, R = 3 * p.F, NC = nxc + 4;
    const int n_a = 8 * NC * 64, n_fc = 4 * 64;
    uint4 *img = reinterpret_cast<uint4 *>(p.tp.packed);
    for (int sidx = blockIdx.x * blockDim.x + threadIdx.x; sidx < n_a + n_fc; sidx += gridDim.x * blockDim.x) {
        float w[8];
        int slot_hi, slot_lo;
        if (sidx < n_a) {
            const int ln = sidx & 63, g = sidx >> 6, ch = g % NC, r = g / NC;
            const int rho = ln & 31, hb = ln >> 5, row = (rho >> 3) * kTpH + 8 * r + (rho & 7);        // torch row: gate * 64 + unit
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (ch < nxc) {
                    const int k = 16 * ch + 8 * hb + j;
                    w[j] = k < I ? p.tp.w_ih[row * I + k] * tp_gate_scale(row) * (k == 0 ? kWsProgScale : 1.0f) : 0.0f;
                } else {
                    w[j] = p.tp.w_hh[row * kTpH + ws_unit(ch - nxc, hb, j)] * tp_gate_scale(row);
                }
            }
            slot_hi = L.a + ((r * 2 + 0) * NC + ch) * 64 + ln;
            slot_lo = L.a + ((r * 2 + 1) * NC + ch) * 64 + ln;
        } else {
            const int u = sidx - n_a, ln = u & 63, c = u >> 6, row = ln & 31, hb = ln >> 5;
#pragma unroll
            for (int j = 0; j < 8; ++j) w[j] = row < R ? p.tp.w_fc[row * kTpH + ws_unit(c, hb, j)] * (2.0f * kNegLog2e) : 0.0f;
            slot_hi = L.wfc + c * 64 + ln;
            slot_lo = L.wfc + (4 + c) * 64 + ln;
        }
        half8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            _Float16 a, b;
            ws_split(w[j], a, b);
            hi[j] = a; lo[j] = b;
        }
        img[slot_hi] = *reinterpret_cast<uint4 *>(&hi);
        img[slot_lo] = *reinterpret_cast<uint4 *>(&lo);
    }
    float *bias = reinterpret_cast<float *>(img + L.bias), *bfc = reinterpret_cast<float *>(img + L.bfc);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 256 + 32; idx += gridDim.x * blockDim.x) {
        if (idx < 256) {                                    // [slice][half][16]: register i <-> gate i>>2 of unit 8 r + 4 hb + (i&3)
            const int i = idx & 15, hb = (idx >> 4) & 1, r = idx >> 5;
            const int g = (i >> 2) * kTpH + 8 * r + 4 * hb + (i & 3);
            bias[idx] = (p.tp.b_ih[g] + p.tp.b_hh[g]) * tp_gate_scale(g);
        } else {
            const int i = idx & 15, hb = (idx - 256) >> 4;
            const int row = 8 * (i >> 2) + 4 * hb + (i & 3);
            bfc[idx - 256] = row < R ? p.tp.b_fc[row] * (2.0f * kNegLog2e) : 0.0f;
        }
    }
}

// One column tile's matrix products in the weight-stationary kernel, all into one accumulator that starts at the bias.  Per 16-wide k-chunk
// (x chunks, then the four h chunks) three ops: w_lo x v_hi, w_hi x v_hi, w_hi x v_lo — the hi operand of the activations is READ ONCE and
// used by two consecutive ops (round 4; until round 3 the cross terms of every chunk came first and the leading terms read the hi operands
// a second time: 15 ds_read_b128 per tile, now 10 — a third of the kernel's LDS operand traffic, which is what it waits for).
// The B operands go through a ring of four registers.  A ring slot is refilled only after the op BEHIND its last user has issued: every op
// depends on its predecessor's accumulator, so that one has retired by then — a ds_read that lands in a register an MFMA in flight still
// reads corrupts the operand (found in round 2 on the A side, and in round 3 on the B side: 1-3 envs in 256 off by 1e-5 while the compiler
// placed the reads).  Load l (chunk l / 2, term l % 2) sits in slot (l + BASE) % 4; loads 0..2 are issued up front, load 3 behind op 0, and then
// 2 c behind op 3 c - 4 and 2 c + 1 behind op 3 c - 3: three to five ops ahead of their first use.  BASE rotates from tile to tile so that the
// up-front reads of a tile, which may be issued right behind the previous tile's last MFMA, never target that MFMA's slot.
template <int NXC, bool WITH_H, int TILES = kWsTiles>
struct WsTile {
    static constexpr int NCH = NXC + (WITH_H ? 4 : 0), N = 3 * NCH, NL = 2 * NCH, RING = 4;
    static constexpr int load_of(int k) { return 2 * (k / 3) + (k % 3 == 2 ? 1 : 0); }
    static constexpr int a_term(int k) { return k % 3 == 0 ? 1 : 0; }
    static constexpr int refill(int k) { return k % 3 == 2 ? 2 * ((k + 4) / 3) : (k % 3 == 0 ? 2 * ((k + 3) / 3) + 1 : -1); }
    static constexpr int base(int te) { return (NL * te) % RING; }
    struct Ctx {
        f32x16 &acc;
        half8 (&b)[RING];
        const half8 (&aw)[2][NXC + 4];
        const uint4 *xb, *hp;     // x buffer of this timestep / h buffer, both + tile * 64 + lane
    };
    template <int l>
    static __device__ __forceinline__ half8 load(const Ctx &c) {
        constexpr int ch = l / 2, term = l % 2;
        constexpr bool is_x = ch < NXC;
        constexpr int off = (((is_x ? ch : ch - NXC) * 2 + term) * TILES) * 64;
#ifdef TP_WS_HALF_B        // (measurement arm: every second operand read skipped — what halving the LDS operand traffic would buy)
        if constexpr (l % 2 == 1) { half8 r = c.b[0]; asm volatile("" : "+v"(r)); return r; }
#endif
        return __builtin_bit_cast(half8, is_x ? c.xb[off] : c.hp[off]);
    }
    template <int BASE, int l>
    static __device__ __forceinline__ void pro(const Ctx &c) {
        c.b[(l + BASE) % RING] = load<l>(c);
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int BASE, int k>
    static __device__ __forceinline__ void step(const Ctx &c) {
#ifndef TP_WS_NO_MFMA      // (measurement arms TP_WS_*: tools/lab/r06_tp_arms.sh, profiles/r06_tp_where_the_time_goes.txt; never defined in the shipped build)
        c.acc = TP_MFMA(c.aw[a_term(k)][k / 3], c.b[(load_of(k) + BASE) % RING], c.acc);
#else
        asm volatile("" : "+v"(c.acc) : "v"(c.aw[a_term(k)][k / 3]), "v"(c.b[(load_of(k) + BASE) % RING]));
#endif
        constexpr int l = refill(k);
        if constexpr (l >= 3 && l < NL) c.b[(l + BASE) % RING] = load<l>(c);
        __builtin_amdgcn_sched_barrier(0);
    }
    template <int BASE, int... Ps, int... Ks>
    static __device__ __forceinline__ void run_seq(const Ctx &c, std::integer_sequence<int, Ps...>, std::integer_sequence<int, Ks...>) {
        (pro<BASE, Ps>(c), ...);
        (step<BASE, Ks>(c), ...);
    }
    template <int BASE>
    static __device__ __forceinline__ void run(const Ctx &c) {
        run_seq<BASE>(c, std::make_integer_sequence<int, (NL < 3 ? NL : 3)>{}, std::make_integer_sequence<int, N>{});
    }
};

// frames of two and more chunks hold 48-72 registers of weights per lane (two chunks under the 128-register cap of four waves per SIMD spill into
// the hot loop: 186 us against 133 us for THREE chunks without the cap) and 66-116 KB of LDS: one workgroup per CU, two waves per SIMD.  The ONE-tile kernel of
// two-chunk frames (a quarter of the cell state and of the published slices, no frame registers in the loop) fits 106 registers: four waves per SIMD again
template <int NXC, int TILES = kWsTiles>
__global__ __launch_bounds__(kWsThreads, (NXC == 1 || (NXC == 2 && TILES == 1)) ? 4 : 2) void hns_tp_lstm_ws_kernel(const TpParams p) {
    static_assert(TILES == 1 || TILES == 2 || TILES == 4, "column tiles per workgroup");
    constexpr int NC = NXC + 4, kWsEnvs = ws_envs(TILES);
    // One-tile workgroups (the smallest batches: a launch is ONE workgroup's serial stream) build every frame of the window in the prologue, all 512 threads at
    // once, instead of frame t + 1 inside timestep t by the 128 threads that own a (unit, quad) pair: with one tile those are waves 0 and 1 only, and their 800
    // cycles per timestep were everybody's wait at the second barrier (tools/tp_phases.py --loop, HNS_TP_TILES=1: 29.1 k cycles per recurrence, 13.5 k in the tiles).
    constexpr bool UPFRONT = TILES == 1;
    constexpr WsImage L = ws_image(NXC);
    extern __shared__ __align__(16) uint4 simg[];
    // LDS (16-byte slots): h_{t-1} [chunk 4][term 2][tile TILES][lane 64] | x [buffer 2][chunk NXC][term 2][tile TILES][lane 64] | bias [64]
    // (UPFRONT: x holds max(T, 2) frames instead of two buffers; h has two buffers — h_t goes to buffer t & 1 while slower waves still read h_{t-1} from
    //  the other one, so the second workgroup barrier of a timestep is not needed: the next timestep's first one orders everything)
    constexpr int kHBuf = 512 * TILES;
    uint4 *sH = simg, *sX = simg + (UPFRONT ? 2 : 1) * kHBuf, *sB = sX + (UPFRONT ? (p.T > 2 ? p.T : 2) : 2) * (128 * TILES * NXC);
    const int tid = threadIdx.x, lane = tid & 63, r = tid >> 6, hb = lane >> 5;
    const int I = p.I, T = p.T, R = 3 * p.F;
    const int e0 = blockIdx.x * kWsEnvs;
    const uint4 *img = reinterpret_cast<const uint4 *>(p.tp.packed);
    // (wave-uniform: through readfirstlane the stamp pointer lives in scalar registers — as a per-lane pair it was the first thing spilled across the loop)
    unsigned long long *prof = p.prof ? p.prof + (size_t)(blockIdx.x * kWsWaves + __builtin_amdgcn_readfirstlane(r)) * 16 : nullptr;
    if (prof && lane == 0) prof[0] = __builtin_amdgcn_s_memrealtime();

    // ---- this wave's rows of the weight matrix: A operands for the whole kernel ----
    half8 aw[2][NC];
#pragma unroll
    for (int term = 0; term < 2; ++term)
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) {
            const uint4 v = img[L.a + ((r * 2 + term) * NC + ch) * 64 + lane];
            aw[term][ch] = __builtin_bit_cast(half8, v);
        }
    if (tid < 64) sB[tid] = img[L.bias + tid];

    // ---- frames: thread (env_l, q) owns values 4 q .. 4 q + 3 of every 16-wide chunk of env e0 + env_l ----
    // (fewer than four tiles: the threads beyond the workgroup's 32 TILES units load the last unit's frame again and store nothing)
    // (UPFRONT: thread = (frame selector tid >> 7, unit, quad): frames tsel, tsel + 4, ... of its pair)
    const int env_l = UPFRONT ? (tid & 127) >> 2 : tid >> 2, q = tid & 3, tsel = tid >> 7;
    const bool mine = UPFRONT || TILES == kWsTiles || env_l < kWsEnvs;
    const int e = e0 + env_l;
    const bool valid = mine && e < p.E;
    const int ec = valid ? e : p.E - 1;
    const bool vec = (I & 3) == 0;                         // rows 16-byte aligned and made of whole quads
    float *hist = p.tp.history + (size_t)ec * T * I;
    float nf[NXC][4];                                      // the new frame
    {
        // every load first (the detection byte and the 4 NXC words: independent, one memory round trip), the selects behind them
        const unsigned dbyte = p.NT == 2 ? p.detect[ec >> 1] : p.detect[ec];
        int kind[NXC][4];
#pragma unroll
        for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
            for (int j = 0; j < 4; ++j) nf[cx][j] = *tp_frame_addr(p, ec, 16 * cx + 4 * q + j, kind[cx][j]);
        const bool det = p.NT == 2 ? ((dbyte >> (ec & 1)) & 1) != 0 : dbyte != 0;
#pragma unroll
        for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kd = kind[cx][j];
                const float v = nf[cx][j];
                nf[cx][j] = kd == 0 ? v : kd == 1 ? (det ? v : p.mask_value) : kd == 2 ? p.cylinder_size : 0.0f;
            }
    }
    auto load_row = [&](int slot, float (&dst)[NXC][4]) {
        const float *row = hist + (size_t)slot * I;
#pragma unroll
        for (int cx = 0; cx < NXC; ++cx) {
            const int k0 = 16 * cx + 4 * q;
            if (vec) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 < I) v = *reinterpret_cast<const float4 *>(row + k0);
                dst[cx][0] = v.x; dst[cx][1] = v.y; dst[cx][2] = v.z; dst[cx][3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[cx][j] = (k0 + j) < I ? row[k0 + j] : 0.0f;
            }
        }
    };
    // frame tt of the new window: stored to the window in HBM (raw) and to LDS as the B operand of timestep tt (split)
    auto store_frame = [&](int tt, const float (&v)[NXC][4]) {
        float *row = hist + (size_t)tt * I;
#pragma unroll
        for (int cx = 0; cx < NXC; ++cx) {
            const int k0 = 16 * cx + 4 * q;
            if (valid) {
                if (vec) { if (k0 < I) *reinterpret_cast<float4 *>(row + k0) = make_float4(v[cx][0], v[cx][1], v[cx][2], v[cx][3]); }
                else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) if (k0 + j < I) row[k0 + j] = v[cx][j];
                }
            }
        }
    };
    auto emit = [&](int tt, const float (&v)[NXC][4]) {
        if constexpr (!UPFRONT) store_frame(tt, v);
#pragma unroll
        for (int cx = 0; cx < NXC; ++cx) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            half4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float w = (cx == 0 && q == 0 && j == 0) ? v[cx][j] * kWsProgInv : v[cx][j];
                _Float16 a, b;
                ws_split(w, a, b);
                hi[j] = a; lo[j] = b;
            }
            // k-slot 4 q + j of the chunk: lane half q >> 1, bytes 8 (q & 1) .. of that lane's 16
            const int slot = (((UPFRONT ? tt : (tt & 1)) * NXC + cx) * 2 * TILES + (env_l >> 5)) * 64 + 32 * (q >> 1) + (env_l & 31);
            if (mine) {
                reinterpret_cast<uint2 *>(sX)[(slot + 0 * TILES * 64) * 2 + (q & 1)] = __builtin_bit_cast(uint2, hi);
                reinterpret_cast<uint2 *>(sX)[(slot + 1 * TILES * 64) * 2 + (q & 1)] = __builtin_bit_cast(uint2, lo);
            }
        }
    };
    const bool fill = (T == 1) || p.fill;
    float xn[NXC][4];
    if constexpr (UPFRONT) {
        // frame tt of the new window = row tt + 1 of the old one (the last: the new frame; a fresh window: the new frame everywhere).  Every load first, then the
        // split operands into LDS; the rows go back to the window BEHIND a workgroup barrier: row tt + 1 is read by the thread of frame tt and written by
        // the thread of frame tt + 1
        constexpr int KF = 4;                              // frames per thread: T <= 16 (hns_tp_bind)
        float fv[KF][NXC][4];
#pragma unroll
        for (int k = 0; k < KF; ++k) {
            const int tt = tsel + 4 * k;
            if (tt < T) {
                if (fill || tt == T - 1) {
#pragma unroll
                    for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
                        for (int j = 0; j < 4; ++j) fv[k][cx][j] = nf[cx][j];
                } else load_row(tt + 1, fv[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < KF; ++k) if (tsel + 4 * k < T) emit(tsel + 4 * k, fv[k]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KF; ++k) if (tsel + 4 * k < T) store_frame(tsel + 4 * k, fv[k]);
    } else {
        if (fill) emit(0, nf);
        else { load_row(1, xn); emit(0, xn); }
        if (T > 1) {
            if (fill || T == 2) {
#pragma unroll
                for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
                    for (int j = 0; j < 4; ++j) xn[cx][j] = nf[cx][j];
            } else load_row(2, xn);
        }
    }

    float c[TILES][4];
#pragma unroll
    for (int te = 0; te < TILES; ++te)
#pragma unroll
        for (int j = 0; j < 4; ++j) c[te][j] = 0.0f;
    uint2 hnew[TILES][2];
    half8 bring[WsTile<NXC, true>::RING];                  // B-operand ring (WsTile)
    const float4 *sBias = reinterpret_cast<const float4 *>(sB) + (r * 2 + hb) * 4;
    if (prof && lane == 0) prof[1] = __builtin_amdgcn_s_memrealtime();
#ifndef TP_WS_NO_PIN
    // The weights were fetched with vector loads at the top of the kernel and are first USED inside the timestep loop, so the compiler put their
    // `s_waitcnt vmcnt(n)` there — at the first matrix instructions of a timestep's first tile (round 6, from the ISA: vmcnt(3) ... vmcnt(0) down the first
    // chain).  The counter is in order: inside the loop that wait also covers the window row this timestep has just requested (needed one timestep LATER)
    // and the row it has just stored.  Claim the registers here, once, where nothing else is in flight.  (Measured: no difference, 81.1 against 82.4 us in
    // one process each — the rows come from the Infinity Cache long before the chain's fourth instruction; kept because the loop's ISA now says what it means.)
#pragma unroll
    for (int term = 0; term < 2; ++term)
#pragma unroll
        for (int ch = 0; ch < NC; ++ch) asm volatile("" : "+v"(aw[term][ch]));
#endif
#ifdef TP_WS_LOOP_PROF      // (measurement arm: shader-clock cycles of a timestep's five segments, summed over the timesteps -> prof[5..9], tools/tp_phases.py --loop)
    unsigned long long lp[5] = {0, 0, 0, 0, 0}, lt = __builtin_readcyclecounter();
#define TP_LP(i) { const unsigned long long n_ = __builtin_readcyclecounter(); lp[i] += n_ - lt; lt = n_; }
#else
#define TP_LP(i)
#endif

    for (int t = 0; t < T; ++t) {
        {   // Two workgroups share a CU and the SIMDs issue oldest-first: left alone the older one runs ahead (its recurrence ends at 65-77 us, the
            // younger one's at ~110 in a stamped launch, tools/tp_phases.py) and the second half of the launch has two waves per SIMD instead of four.
            // Priority by recurrence step — the workgroup that is behind goes first — keeps them together: step + predictor 100.1 -> 95.3 us
            // (alternating blocks in one process, tools/lab/r04_batch65.sh, _66; two levels or other thresholds measured no better).
            const int q = (4 * t) / T;
            if (q == 0) __builtin_amdgcn_s_setprio(3); else if (q == 1) __builtin_amdgcn_s_setprio(2); else if (q == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
        TP_LP(4)
        __syncthreads();                                         // x_t and h_{t-1} are in LDS
        TP_LP(0)
        auto emit_next = [&]() {
            if (!UPFRONT && t + 1 < T) {                   // frame t+1 -> the other x buffer (last read at timestep t-1)
                emit(t + 1, xn);
                if (t + 2 < T) {
                    if (fill || t + 2 == T - 1) {
#pragma unroll
                        for (int cx = 0; cx < NXC; ++cx)
#pragma unroll
                            for (int j = 0; j < 4; ++j) xn[cx][j] = nf[cx][j];
                    } else load_row(t + 3, xn);
                }
            }
        };
        // Wider frames run two waves per SIMD (waves w and w + 4 of this workgroup), in lockstep between the barriers: both in their matrix chains, then both
        // in their cell updates.  The second half of the waves builds frame t + 1 BEHIND its tiles instead of in front of them, so its emission runs beside the
        // other half's matrix chains and vice versa (round 6; 65 536 units, four-tile workgroups: two chunks 152.6 -> 134.0 us, three 130.9 -> 128.1,
        // four 223 -> 200, five 274 -> 235; bit-identical).  Not for one-chunk frames (four waves per SIMD: the other workgroup's waves already fill those gaps): the
        // four-tile kernel spills 35 registers with the late copy at its 128-register cap (111 us), the two-tile one measures +3 %.
        const bool late_emit = NXC >= 2 && !UPFRONT && __builtin_amdgcn_readfirstlane(r) >= kWsWaves / 2;
        if (!late_emit) emit_next();
        TP_LP(1)
        auto tile = [&](auto te_c) {
            constexpr int te = decltype(te_c)::value;
            f32x16 acc;
#ifdef TP_WS_ZERO_BIAS     // (measurement arm: the four bias reads of a tile replaced by register moves)
#pragma unroll
            for (int v = 0; v < 16; ++v) { float z = 0.0f; asm volatile("" : "+v"(z)); acc[v] = z; }
#else
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 b = sBias[v];
                acc[4 * v] = b.x; acc[4 * v + 1] = b.y; acc[4 * v + 2] = b.z; acc[4 * v + 3] = b.w;
            }
#endif
            {
                const uint4 *xb = sX + (UPFRONT ? t : (t & 1)) * (NXC * 2 * TILES * 64) + te * 64 + lane;
                const uint4 *hp = sH + (UPFRONT ? ((t + 1) & 1) * kHBuf : 0) + te * 64 + lane;        // h_{t-1}
                if (t > 0) {
                    using W = WsTile<NXC, true, TILES>;
                    const typename W::Ctx cx{acc, bring, aw, xb, hp};
                    W::template run<W::base(te)>(cx);
                } else {
                    using W = WsTile<NXC, false, TILES>;
                    const typename W::Ctx cx{acc, bring, aw, xb, hp};
                    W::template run<W::base(te)>(cx);
                }
            }
            // cell update of units 8 r + 4 hb + j (torch.nn.LSTM gate order i, f, g, o = accumulator registers j, 4 + j, 8 + j, 12 + j)
            // Round 4: 22 vector instructions per unit, 8 of them transcendental (was 26 / 9).  With e_x = 2^(pre-activation) (the weight
            // image carries -log2 e, -2 log2 e for the g gate):
            //   sigmoid(i) tanh(g) = (2 - E_g) / (E_i E_g) = (1 - e_g) / fma(e_i, E_g, E_g)                 one reciprocal for both gates,
            //   o tanh(c')         = (1 - e_c) / fma(e_o, E_c, E_c)                                         the "1 +" of E_i / E_o folded into the fma,
            //   the fp16 split of h = P Q straight from the exact product: hi = RN16(P Q) (v_fma_mixlo/hi_f16), lo = RN16(P Q - hi) (the same
            //   instruction with hi as its f16 addend): two instructions instead of multiply + convert + convert back + subtract + convert, the
            //   halves land packed, and hi + lo carries P Q to 2^-22 whichever way hi rounded.
            // Saturated gates: e_i or e_o = inf -> the fma is inf, its reciprocal 0, the term 0 (P and 1 - e_g are finite: e_g is clamped to 2^64,
            // e_c <= 2^(2.9 T)); e_f = inf -> forget gate 0.
            unsigned hi_r[2], lo_r[2];
#ifdef TP_WS_NO_CELL
            hi_r[0] = __float_as_uint(acc[0]) ^ __float_as_uint(acc[5]); hi_r[1] = __float_as_uint(acc[2]) ^ __float_as_uint(acc[7]);
            lo_r[0] = __float_as_uint(acc[8]) ^ __float_as_uint(acc[13]); lo_r[1] = __float_as_uint(acc[10]) ^ __float_as_uint(acc[15]);
            hi_r[0] &= 0x03ff03ffu; hi_r[1] &= 0x03ff03ffu; lo_r[0] &= 0x03ff03ffu; lo_r[1] &= 0x03ff03ffu;
#else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float ei = __builtin_amdgcn_exp2f(acc[j]), ef = __builtin_amdgcn_exp2f(acc[4 + j]);
                // (fminf() canonicalises its operand first: a v_max_f32 x, x per unit beside the v_min.  Writing the clamp as ONE v_min_f32 in inline assembly
                //  was tried in round 6 and is WRONG here: the compiler's hazard recogniser does not see an inline-asm read of a matrix instruction's result and
                //  leaves out the wait states in front of it — one prediction in 4 500 off by 1.6e-5, tests/test_hip_tp.py[300-1-0])
                const float eg = __builtin_amdgcn_exp2f(__builtin_fminf(acc[8 + j], 64.0f)), eo = __builtin_amdgcn_exp2f(acc[12 + j]);
                const float Eg = 1.0f + eg;
                const float ig = (1.0f - eg) * __builtin_amdgcn_rcpf(HNS_FMA(ei, Eg, Eg));
                const float cn = HNS_FMA(__builtin_amdgcn_rcpf(1.0f + ef), c[te][j], ig);
                c[te][j] = cn;
                const float ec = __builtin_amdgcn_exp2f(cn * (2.0f * kNegLog2e));
                const float Ec = 1.0f + ec;
                const float P = 1.0f - ec, Q = __builtin_amdgcn_rcpf(HNS_FMA(eo, Ec, Ec));
                if ((j & 1) == 0) {
                    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi_r[j >> 1]) : "v"(P), "v"(Q));
                    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo_r[j >> 1]) : "v"(P), "v"(Q), "v"(hi_r[j >> 1]));
                } else {
                    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi_r[j >> 1]) : "v"(P), "v"(Q));
                    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo_r[j >> 1]) : "v"(P), "v"(Q), "v"(hi_r[j >> 1]));
                }
            }
#endif
            hnew[te][0] = make_uint2(hi_r[0], hi_r[1]);
            hnew[te][1] = make_uint2(lo_r[0], lo_r[1]);
        };
        tile(std::integral_constant<int, 0>{});
        if constexpr (TILES > 1) tile(std::integral_constant<int, 1>{});
        if constexpr (TILES > 2) { tile(std::integral_constant<int, 2>{}); tile(std::integral_constant<int, 3>{}); }
        if (late_emit) emit_next();
        TP_LP(2)
#ifndef TP_WS_NO_BAR2
        if constexpr (!UPFRONT) __syncthreads();                 // every wave has read h_{t-1}
#endif
        TP_LP(3)
        // publish this slice of h_t: chunk r >> 1, k-slots 4 (r & 1) .. + 3 of both lane halves
#pragma unroll
        for (int te = 0; te < TILES; ++te)
#pragma unroll
            for (int term = 0; term < 2; ++term)
                reinterpret_cast<uint2 *>(sH + (UPFRONT ? (t & 1) * kHBuf : 0))[((((r >> 1) * 2 + term) * TILES + te) * 64 + lane) * 2 + (r & 1)] = hnew[te][term];
    }
    // The output layer's weight operands and bias come from the L2 (the packed image): requested HERE, ahead of the barrier that closes the recurrence — the
    // weight registers of the loop are dead, and the round trip (about a microsecond with every workgroup of the launch in its epilogue at once) runs beside the
    // wait for the slowest wave instead of behind it (round 6).
    // (behind a scheduling barrier: hoisted into the last timestep these loads could land in the registers of weight operands that matrix instructions still
    //  in flight are reading — the hazard of WsTile's operand ring, on the A side)
    __builtin_amdgcn_sched_barrier(0);
    half8 f1[4], f2[4];
    float4 fcb[4];
    if (r < TILES) {
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            f1[ch] = __builtin_bit_cast(half8, img[L.wfc + ch * 64 + lane]); f2[ch] = __builtin_bit_cast(half8, img[L.wfc + (4 + ch) * 64 + lane]);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) fcb[v] = (reinterpret_cast<const float4 *>(img + L.bfc) + hb * 4)[v];
    }
    __syncthreads();
    if (prof && lane == 0) prof[2] = __builtin_amdgcn_s_memrealtime();
#ifdef TP_WS_LOOP_PROF
    if (prof && lane == 0) { for (int i = 0; i < 5; ++i) prof[5 + i] = lp[i]; }
#endif

    // ---- output layer on h_T (waves 0..3, one column tile each): tanh(W_fc h + b), rescaled to arena units (hideandseek.py:834-836) ----
    float *sPred = reinterpret_cast<float *>(sX);           // [unit 32 TILES][16], or [unit 32 TILES][32] with more than five predicted points (3F > 16): 4 KB per tile,
    const int ps = R > 16 ? 32 : 16;                        // the size of ONE chunk's x buffers, which are free now
    if (r < TILES) {
        const int te = r;
        f32x16 o;
#pragma unroll
        for (int v = 0; v < 4; ++v) { o[4 * v] = fcb[v].x; o[4 * v + 1] = fcb[v].y; o[4 * v + 2] = fcb[v].z; o[4 * v + 3] = fcb[v].w; }
        half8 fh[4], fl[4];                                 // every operand in its own register, all read before the first MFMA
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const uint4 *bh = sH + (UPFRONT ? ((T - 1) & 1) * kHBuf : 0) + ((ch * 2) * TILES + te) * 64 + lane;      // h_{T-1}
            fh[ch] = __builtin_bit_cast(half8, bh[0]); fl[ch] = __builtin_bit_cast(half8, bh[TILES * 64]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            o = TP_MFMA(f2[ch], fh[ch], o);
            o = TP_MFMA(f1[ch], fl[ch], o);
        }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) o = TP_MFMA(f1[ch], fh[ch], o);
        __builtin_amdgcn_sched_barrier(0);
        const int el = te * 32 + (lane & 31), er = e0 + el;
        float *pr = p.tp.pred + (size_t)(er < p.E ? er : p.E - 1) * R;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int row = 8 * (i >> 2) + 4 * hb + (i & 3);
            if (row < R) {
                const float v = tp_tanh_s(o[i]);
                const int comp = row % 3;
                const float val = (comp < 2) ? (v * 0.5f) * p.arena_size : ((v + 1.0f) * 0.5f) * p.max_height;
                if (er < p.E) pr[row] = val;
                sPred[el * ps + row] = val;
            }
        }
    }
    __syncthreads();
    if (prof && lane == 0) prof[3] = __builtin_amdgcn_s_memrealtime();
    // ---- observation rows of the workgroup's envs ----
    {
        const int A = p.A, NT = p.NT, ev0 = e0 / NT;        // (two evaders: the workgroup's 128 units are 64 envs)
        for (int rr = tid; rr < (kWsEnvs / NT) * A; rr += kWsThreads) {
            const int el = rr / A, a = rr - el * A, ev = ev0 + el;
            if (ev * NT < p.E) tp_write_row(p, ev, a, sPred + (el * NT) * ps, sPred + (el * NT + NT - 1) * ps);
        }
    }
    if (prof && lane == 0) prof[4] = __builtin_amdgcn_s_memrealtime();
}

}  // namespace hns

// =================================================================================================
// Host side
// =================================================================================================
using hns::TpParams;

static int tp_nxc(int I) { return (I + 15) / 16; }
// which kernel serves a frame width: the weight-stationary one for one 16-value chunk (up to 3 pursuers, no cylinders in the
// frame — the reference's default) and for four or five (6-7 pursuers with 12-16 cylinders in the frame: the tile kernel's LDS-resident
// weight image does not fit beyond three chunks); HNS_TP_KERNEL=tile / ws forces one of them where both exist (A/B measurements)
static bool tp_use_ws(int nxc) {
    static const int mode = [] { const char *m = getenv("HNS_TP_KERNEL"); return !m ? 0 : (m[0] == 't' ? 1 : (m[0] == 'w' ? 2 : 0)); }();
    if (nxc > 1) return true;        // three chunks: 137 us against the tile kernel's 245 (tools/tp_widths.py, round 5) — and that instantiation of the tile kernel spilled
                                     // 23 registers at its 256-register cap.  Two chunks: the tile kernel was 8 % ahead (135 against 146 us) with ONE register parked in
                                     // scratch across its timestep loop; a shorter operand ring, a scalar wave index and a late re-derivation of that offset each left
                                     // it at 1-10 spilled registers (round 6), so that instantiation is no longer built either: no shipped kernel touches scratch
                                     // (tests/test_kernel_resources.py)
    return mode != 1;                // one chunk: the weight-stationary kernel unless HNS_TP_KERNEL=tile asks for the other (A/B)
}
static int tp_frame_dim(const hns_cfg &c) { return 7 + 3 * c.num_agents + (c.tp_use_obstacles ? 3 * c.num_cylinders : 0); }

static void tp_fill_params(const hns_env *env, TpParams &p) {
    const hns_cfg &c = env->cfg;
    p.tp = env->tp.buf;
    p.drone_state = env->buf.drone_state;
    p.target_pos = env->buf.target_pos;
    p.target_vel = env->buf.target_vel;
    p.progress = env->buf.progress;
    p.obs_self20 = env->buf.obs_self;
    p.detect = env->buf.detect;
    p.cylinders = env->buf.cylinders;
    p.NT = c.num_targets == 2 ? 2 : 1;
    p.E = c.num_envs * p.NT; p.A = c.num_agents; p.C = c.num_cylinders;
    p.I = tp_frame_dim(c);
    p.cylinder_size = c.cylinder_size;
    p.T = env->tp.history_step; p.F = env->tp.future_step;
    p.fill = 0;
    p.max_len = c.max_episode_length;
    p.prof = env->prof;
    p.mask_value = c.mask_value; p.arena_size = c.arena_size; p.max_height = c.max_height;
}

extern "C" {

size_t hns_tp_packed_bytes(void) {                                                 // the larger of the two kernels' images at their widest frames
    const size_t a = (size_t)hns::tp_image(3).bytes, b = (size_t)hns::ws_image(hns::kTpMaxChunks).slots * 16;
    return a > b ? a : b;
}

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
    if (!b->w_ih || !b->w_hh || !b->b_ih || !b->b_hh || !b->w_fc || !b->b_fc || !b->packed || !b->history || !b->pred ||
        !b->obs_self || !b->groundtruth || !b->tp_done) {
        hns_set_error("hns_tp_bind: null buffer (only state_drones may be null)");
        return HNS_ERR_INVALID_ARG;
    }
    for (const void *ptr : {(const void *)b->w_ih, (const void *)b->w_hh, (const void *)b->b_ih, (const void *)b->b_hh, (const void *)b->w_fc,
                            (const void *)b->b_fc, (const void *)b->packed, (const void *)b->history, (const void *)b->pred,
                            (const void *)b->obs_self, (const void *)b->groundtruth, (const void *)b->tp_done})
        if (!hns_on_env_device(env, ptr)) {
            hns_set_error("hns_tp_bind: every buffer (parameters included) must be device memory of the env's GPU");
            return HNS_ERR_INVALID_ARG;
        }
    if ((reinterpret_cast<uintptr_t>(b->packed) & 15) != 0) { hns_set_error("hns_tp_bind: packed must be 16-byte aligned"); return HNS_ERR_INVALID_ARG; }
    if (tp_nxc(tp_frame_dim(env->cfg)) > hns::kTpMaxChunks) {
        hns_set_error("hns_tp_bind: frame wider than 80 values (7 + 3 num_agents + 3 num_cylinders with tp_use_obstacles)");
        return HNS_ERR_CONFIG;
    }
    if (env->cfg.max_episode_length > 60000) {
        // the frame holds `progress`; the matrix-core operands are fp16 splits (|x| < 65 504)
        hns_set_error("hns_tp_bind: max_episode_length > 60000 does not fit the fp16-split operands of the predictor");
        return HNS_ERR_CONFIG;
    }
    env->tp.buf = *b;
    env->tp.history_step = history_step;
    env->tp.future_step = future_step;
    env->tp.bound = true;
    env->tp.dirty = true;
    return HNS_OK;
}

int hns_tp_refresh(hns_env *env, void *stream) {
    if (!env) { hns_set_error("hns_tp_refresh: null env"); return HNS_ERR_INVALID_ARG; }
    if (!env->tp.bound) { hns_set_error("hns_tp_refresh: hns_tp_bind first"); return HNS_ERR_NOT_BOUND; }
    {   // as for a step launch: the caller's current device must be the env's (the attribute below is per device, and so is the launch)
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != env->device) {
            hns_set_error("hns_tp_observe / hns_tp_refresh: the current HIP device is not the one this env was created on");
            return HNS_ERR_DEVICE;
        }
        env->last_stream = static_cast<hipStream_t>(stream);
    }
    TpParams p;
    tp_fill_params(env, p);
    if (tp_use_ws(tp_nxc(p.I))) hipLaunchKernelGGL(hns::hns_tp_pack_ws_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, p, tp_nxc(p.I));
    else hipLaunchKernelGGL(hns::hns_tp_pack_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, p, tp_nxc(p.I));
    HNS_CHECK_HIP(hipGetLastError());
    env->tp.dirty = false;
    return HNS_OK;
}

int hns_tp_observe(hns_env *env, int32_t fill_history, void *stream) {
    if (!env) { hns_set_error("hns_tp_observe: null env"); return HNS_ERR_INVALID_ARG; }
    if (!env->tp.bound) { hns_set_error("hns_tp_observe: hns_tp_bind first"); return HNS_ERR_NOT_BOUND; }
    {   // as for a step launch: the caller's current device must be the env's (the attribute below is per device, and so is the launch)
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev != env->device) {
            hns_set_error("hns_tp_observe / hns_tp_refresh: the current HIP device is not the one this env was created on");
            return HNS_ERR_DEVICE;
        }
        env->last_stream = static_cast<hipStream_t>(stream);
    }
    if (env->tp.dirty) {
        const int rc = hns_tp_refresh(env, stream);
        if (rc != HNS_OK) return rc;
    }
    TpParams p;
    tp_fill_params(env, p);
    p.fill = fill_history ? 1 : 0;
    const int nxc = tp_nxc(p.I);
    if (tp_use_ws(nxc)) {
        // column tiles per workgroup (ws_envs): four unless that grid leaves most of the chip idle (two-tile workgroups: one-chunk frames, the reference's
        // default shape, only); HNS_TP_TILES=1|2|4 forces a value (A/B measurements); the phase stamps (hns_set_phase_profile) are laid out for four unless a
        // value is forced (tools/tp_phases.py sizes its buffer by the same variable)
        static const int forced = [] { const char *m = getenv("HNS_TP_TILES"); const int v = m ? atoi(m) : 0; return (v == 1 || v == 2 || v == 4) ? v : 0; }();
        int tiles = hns::kWsTiles;
        if (nxc >= 2 && (!p.prof || forced)) {
            // wider frames (tools/tp_tiles.py --agents / --obst / --cyl): two chunks — the one-tile kernel runs four waves per SIMD where the four-tile one has two —
            // 2 048 units 61.6 -> 20.7 us, 65 536 units 146 -> 131, 131 072 units 290 either way; three to five chunks (one workgroup per CU either way)
            // 2 048 units 52-105 -> 19-34 us, 16 384 units 58-112 -> 44-81, 32 768 units 64-138 against 85-156: one tile up to two workgroups per CU
            const int cus = env->cus > 0 ? env->cus : 256, groups1 = (p.E + 31) / 32;
            if (forced == 1 || forced == 4) tiles = forced;
            else tiles = groups1 <= (nxc == 2 ? 8 : 2) * cus ? 1 : 4;
            // one-tile workgroups hold the whole window's operands in LDS: five-chunk frames with windows of more than 14 frames do not fit
            if (tiles == 1 && (2 * 512 + (p.T > 2 ? p.T : 2) * 128 * nxc + 64) * 16 > 160 * 1024) tiles = 4;
        }
        if (nxc == 1 && (!p.prof || forced)) {
            const int cus = env->cus > 0 ? env->cus : 256;
            // measured on one box (tools/tp_tiles.py, profiles/r06_tp_tiles.txt; 256 CUs): one tile up to two workgroups per CU (16 384 units: 26.0 us against
            // 28.5 / 45.7 with two / four), two tiles below three two-tile workgroups per CU (32 768 units: 42.7 against 48.9 / 50.2; 49 152: 67 either way), else four
            if (forced) tiles = forced;
            else if ((p.E + 31) / 32 <= 2 * cus) tiles = 1;
            else if ((p.E + 63) / 64 < 3 * cus) tiles = 2;
            else tiles = 4;
        }
        void (*wfn)(const TpParams) = nxc == 1 ? (tiles == 1 ? hns::hns_tp_lstm_ws_kernel<1, 1> : tiles == 2 ? hns::hns_tp_lstm_ws_kernel<1, 2> : hns::hns_tp_lstm_ws_kernel<1, 4>)
                                      : nxc == 2 ? (tiles == 1 ? hns::hns_tp_lstm_ws_kernel<2, 1> : hns::hns_tp_lstm_ws_kernel<2>)
                                      : nxc == 3 ? (tiles == 1 ? hns::hns_tp_lstm_ws_kernel<3, 1> : hns::hns_tp_lstm_ws_kernel<3>)
                                      : nxc == 4 ? (tiles == 1 ? hns::hns_tp_lstm_ws_kernel<4, 1> : hns::hns_tp_lstm_ws_kernel<4>)
                                      : (tiles == 1 ? hns::hns_tp_lstm_ws_kernel<5, 1> : hns::hns_tp_lstm_ws_kernel<5>);
        const bool upf = tiles == 1;
        const int xframes = upf ? (p.T > 2 ? p.T : 2) : 2;                     // one-tile workgroups hold the whole window's operands (UPFRONT)
        const size_t wlds = (size_t)((upf ? 2 : 1) * 512 * tiles + xframes * 128 * tiles * nxc + 64) * 16;       // (... and two buffers of h)
        static thread_local unsigned long long ws_attr_devs[2 * hns::kTpMaxChunks + 1] = {};
        const unsigned long long bit = 1ull << (env->device & 63);
        const int slot = tiles == hns::kWsTiles ? nxc - 1 : nxc == 1 ? hns::kTpMaxChunks + (tiles - 1) : hns::kTpMaxChunks + nxc;     // (wider frames: one or four tiles)
        if (!(ws_attr_devs[slot] & bit)) {                    // (the largest this instantiation asks for: 16 frames with one tile)
            const int fit = (160 * 1024 / 16 - 2 * 512 - 64) / (128 * nxc);             // frames of the window that fit 160 KB beside h and the bias
            const size_t wcap = upf ? (size_t)(2 * 512 + (fit < 16 ? fit : 16) * 128 * nxc + 64) * 16 : wlds;
            HNS_CHECK_HIP(hipFuncSetAttribute((const void *)wfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wcap));
            ws_attr_devs[slot] |= bit;
        }
        const int per = hns::ws_envs(tiles);
        hipLaunchKernelGGL(wfn, dim3((p.E + per - 1) / per), dim3(hns::kWsThreads), wlds, (hipStream_t)stream, p);
        HNS_CHECK_HIP(hipGetLastError());
        return HNS_OK;
    }
    void (*fn)(const TpParams) = hns::hns_tp_lstm_kernel<1>;          // (nxc == 1 here: tp_use_ws)
    const int waves = hns::tp_waves(nxc);
    const size_t lds = (size_t)hns::tp_image(nxc).bytes + (size_t)waves * 8 * nxc * 64 * sizeof(float)    // image + parked new frame
                       + ((nxc == 1 && 3 * p.F > 16) ? (size_t)waves * 1024 * sizeof(float) : 0);              // + the predictions of more than five points
    // the attribute is per device: remembered per (frame width, device), so envs on two GPUs driven from one thread both get it
    static thread_local unsigned long long attr_devs[2] = {0ull, 0ull};
    const unsigned long long dev_bit = 1ull << (env->device & 63);
    if (!(attr_devs[nxc - 1] & dev_bit)) {
        const size_t lds_cap = (size_t)hns::tp_image(nxc).bytes + (size_t)waves * (8 * nxc * 64 + (nxc == 1 ? 1024 : 0)) * sizeof(float);   // the largest this kernel asks for
        HNS_CHECK_HIP(hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
        attr_devs[nxc - 1] |= dev_bit;
    }
    hipStream_t s = (hipStream_t)stream;
    const int grid = (p.E + waves * 32 - 1) / (waves * 32);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(waves * 64), lds, s, p);
    HNS_CHECK_HIP(hipGetLastError());
    return HNS_OK;
}

}  // extern "C"
