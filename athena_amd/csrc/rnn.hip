// Recurrent layers of the DeepSpeech model: tf.keras.layers.RNN over a GRUCell or an LSTMCell with
// return_sequences=True (reference athena/models/deep_speech.py:75-79, athena/layers/commons.py:486-491).
//
// The input projection of ALL time steps (x W + b) is one large GEMM done by the caller; what is left
// is the serial part.  Per time step: one skinny GEMM  h[t-1] U  ([B, H] x [H, G*H], G = 3 or 4 gates)
// on the shared GEMM core and one element-wise gate kernel.  Batch-major layouts throughout (Keras'
// [B, T, ...]), so a time step is a strided row block and nothing is transposed:
//   xw      [B, T, G*H]      x W + b of every step                                   (compute dtype)
//   hs_ext  [B, T+1, H]      slot 0 = initial state (zeros), slot t+1 = h[t]         (compute dtype)
//   cs_ext  [B, T+1, H]      LSTM cell state, same slots                             (fp32)
//   gates   [B, T, G*H]      activated gates kept for the backward pass              (compute dtype)
//   rh      [B, T, H]        GRU: recurrent part of the candidate, (h[t-1] U + b)_h  (compute dtype)
//   y       [B, T, H]        layer output = h[t] (contiguous copy for the next layer)
// The weight gradients are again large GEMMs over all steps, done by the caller from
//   dxw     [B, T, G*H]      gradient at x W + b
//   dhu_ext [B, T+1, G*H]    gradient at h[t-1] U (+ b); slot T is zero, so that hs_ext^T dhu_ext over all
//                            B*(T+1) rows pairs h[t-1] with step t
//
// GRU (Keras GRUCell, reset_after=True, gate order z, r, h):
//   z = sigmoid(xw_z + hu_z)   r = sigmoid(xw_r + hu_r)   hh = tanh(xw_h + r * hu_h)
//   h[t] = z * h[t-1] + (1 - z) * hh
// LSTM (Keras LSTMCell, gate order i, f, c, o):
//   i, f, o = sigmoid(.)   g = tanh(.)   c[t] = f * c[t-1] + i * g   h[t] = o * tanh(c[t])
#include "common.h"
#include <string.h>
#include <stdlib.h>

#define RNN_GRU 0
#define RNN_LSTM 1

__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

template <typename T>
__global__ __launch_bounds__(256) void gru_step_fwd_kernel(const T* __restrict__ xw, const float* __restrict__ hu,
                                                           T* __restrict__ hs_ext, T* __restrict__ y,
                                                           T* __restrict__ gates, T* __restrict__ rh, int B, int Tn,
                                                           int H, int t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, j = idx - b * H;
    const long xo = ((long)b * Tn + t) * 3 * H;
    const float* hb = hu + (long)b * 3 * H;
    const long ho = ((long)b * (Tn + 1) + t) * H + j;
    const float hp = to_f32<T>(hs_ext[ho]);
    const float z = sigmoid_f(to_f32<T>(xw[xo + j]) + hb[j]);
    const float r = sigmoid_f(to_f32<T>(xw[xo + H + j]) + hb[H + j]);
    const float rhv = hb[2 * H + j];
    const float hh = tanhf(to_f32<T>(xw[xo + 2 * H + j]) + r * rhv);
    const float h = z * hp + (1.f - z) * hh;
    const T hq = from_f32<T>(h);
    hs_ext[ho + H] = hq;
    y[((long)b * Tn + t) * H + j] = hq;
    gates[xo + j] = from_f32<T>(z);
    gates[xo + H + j] = from_f32<T>(r);
    gates[xo + 2 * H + j] = from_f32<T>(hh);
    rh[((long)b * Tn + t) * H + j] = from_f32<T>(rhv);
}

// dh = dy[t] + carry;  carry' = dh * z  (the recurrent product adds dhu U^T onto it afterwards)
template <typename T>
__global__ __launch_bounds__(256) void gru_step_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ hs_ext,
                                                           const T* __restrict__ gates, const T* __restrict__ rh,
                                                           float* __restrict__ carry, T* __restrict__ dxw,
                                                           T* __restrict__ dhu_ext, int B, int Tn, int H, int t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, j = idx - b * H;
    const long go = ((long)b * Tn + t) * 3 * H;
    const long eo = ((long)b * (Tn + 1) + t) * 3 * H;
    const float z = to_f32<T>(gates[go + j]), r = to_f32<T>(gates[go + H + j]), hh = to_f32<T>(gates[go + 2 * H + j]);
    const float rhv = to_f32<T>(rh[((long)b * Tn + t) * H + j]);
    const float hp = to_f32<T>(hs_ext[((long)b * (Tn + 1) + t) * H + j]);
    const float dh = to_f32<T>(dy[((long)b * Tn + t) * H + j]) + carry[idx];
    const float da_h = dh * (1.f - z) * (1.f - hh * hh);
    const float da_z = dh * (hp - hh) * z * (1.f - z);
    const float da_r = da_h * rhv * r * (1.f - r);
    dxw[go + j] = from_f32<T>(da_z);
    dxw[go + H + j] = from_f32<T>(da_r);
    dxw[go + 2 * H + j] = from_f32<T>(da_h);
    dhu_ext[eo + j] = from_f32<T>(da_z);
    dhu_ext[eo + H + j] = from_f32<T>(da_r);
    dhu_ext[eo + 2 * H + j] = from_f32<T>(da_h * r);
    carry[idx] = dh * z;
}

template <typename T>
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const T* __restrict__ xw, const float* __restrict__ hu,
                                                            T* __restrict__ hs_ext, float* __restrict__ cs_ext,
                                                            T* __restrict__ y, T* __restrict__ gates, int B, int Tn,
                                                            int H, int t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, j = idx - b * H;
    const long xo = ((long)b * Tn + t) * 4 * H;
    const float* hb = hu + (long)b * 4 * H;
    const long so = ((long)b * (Tn + 1) + t) * H + j;
    const float i = sigmoid_f(to_f32<T>(xw[xo + j]) + hb[j]);
    const float f = sigmoid_f(to_f32<T>(xw[xo + H + j]) + hb[H + j]);
    const float g = tanhf(to_f32<T>(xw[xo + 2 * H + j]) + hb[2 * H + j]);
    const float o = sigmoid_f(to_f32<T>(xw[xo + 3 * H + j]) + hb[3 * H + j]);
    const float c = f * cs_ext[so] + i * g;
    const T hq = from_f32<T>(o * tanhf(c));
    cs_ext[so + H] = c;
    hs_ext[so + H] = hq;
    y[((long)b * Tn + t) * H + j] = hq;
    gates[xo + j] = from_f32<T>(i);
    gates[xo + H + j] = from_f32<T>(f);
    gates[xo + 2 * H + j] = from_f32<T>(g);
    gates[xo + 3 * H + j] = from_f32<T>(o);
}

// carry_h' = 0 (the recurrent product adds dhu U^T onto it), carry_c' = dc * f
template <typename T>
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const T* __restrict__ dy, const float* __restrict__ cs_ext,
                                                            const T* __restrict__ gates, float* __restrict__ carry_h,
                                                            float* __restrict__ carry_c, T* __restrict__ dxw,
                                                            T* __restrict__ dhu_ext, int B, int Tn, int H, int t) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, j = idx - b * H;
    const long go = ((long)b * Tn + t) * 4 * H;
    const long eo = ((long)b * (Tn + 1) + t) * 4 * H;
    const long so = ((long)b * (Tn + 1) + t) * H + j;
    const float i = to_f32<T>(gates[go + j]), f = to_f32<T>(gates[go + H + j]);
    const float g = to_f32<T>(gates[go + 2 * H + j]), o = to_f32<T>(gates[go + 3 * H + j]);
    const float cp = cs_ext[so], c = cs_ext[so + H];
    const float tc = tanhf(c);
    const float dh = to_f32<T>(dy[((long)b * Tn + t) * H + j]) + carry_h[idx];
    const float dc = carry_c[idx] + dh * o * (1.f - tc * tc);
    const float da[4] = {dc * g * i * (1.f - i), dc * cp * f * (1.f - f), dc * i * (1.f - g * g),
                         dh * tc * o * (1.f - o)};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const T q = from_f32<T>(da[k]);
        dxw[go + k * H + j] = q;
        dhu_ext[eo + k * H + j] = q;
    }
    carry_h[idx] = 0.f;
    carry_c[idx] = dc * f;
}

// ------------------------------------------------------------------ one launch per time step (bf16)
// The skinny product and the gate arithmetic of a step in ONE kernel.  A workgroup owns RS_NU = 8 hidden
// units (all G gates of them), so that a layer of a few thousand units spreads over the whole chip: the
// weights are re-read every step from beyond the L2 (a kernel boundary does not keep them there), and
// that stream is paid per CU.  Its 8 waves split the contraction; every wave keeps up to RS_KB k-steps of
// operand fragments in flight straight from global memory into matrix-core layout; the partial tiles are
// added through LDS and 256 threads finish the 32 x 8 gate elements, whose other operands were requested
// before the contraction started.  A 16-column matrix-core tile holds two gates of the 8 units side by
// side.  Batches beyond 32 rows repeat the pass.
#define RS_WAVES 8
#define RS_NU 8
#define RS_KB 7

template <int NB>
__device__ __forceinline__ void rs_zero(f32x4_t (&acc)[NB][2]) {
#pragma unroll
    for (int g = 0; g < NB; ++g) {
        acc[g][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        acc[g][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }
}

// acc[g][rt] += A[rows of tile rt][k] * W_g[tile column][k] over this wave's share of the K contraction
// elements.  a0 / a1: the lane's row of the two 16-row tiles; wrow[g]: the lane's weight row of tile g
// (contiguous in k)
template <int NB>
__device__ __forceinline__ void rs_contract(const bf16_t* __restrict__ a0, const bf16_t* __restrict__ a1,
                                            const bf16_t* const (&wrow)[NB], int K, int wave, int lane,
                                            f32x4_t (&acc)[NB][2]) {
    const int fk = (lane >> 4) * 8;
    const int nks = (K + 31) / 32;
    const int per = (nks + RS_WAVES - 1) / RS_WAVES;
    const int ks0 = wave * per;
    const int ks1 = min(nks, ks0 + per);
    bf16x8_t zero;
#pragma unroll
    for (int r = 0; r < 8; ++r) zero[r] = (bf16_t)0.f;
    for (int kb = ks0; kb < ks1; kb += RS_KB) {
        bf16x8_t af[RS_KB][2], wf[RS_KB][NB];
#pragma unroll
        for (int i = 0; i < RS_KB; ++i) {
            const int k = (kb + i) * 32 + fk;
            const bool ok = (kb + i) < ks1 && k < K;          // K % 8 == 0: a piece is all in or all out
            const int kk = ok ? k : 0;
            af[i][0] = *(const bf16x8_t*)(a0 + kk);
            af[i][1] = *(const bf16x8_t*)(a1 + kk);
#pragma unroll
            for (int g = 0; g < NB; ++g) wf[i][g] = *(const bf16x8_t*)(wrow[g] + kk);
            if (!ok) { af[i][0] = zero; af[i][1] = zero; }
        }
#pragma unroll
        for (int i = 0; i < RS_KB; ++i)
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][g], af[i][0], acc[g][0], 0, 0, 0);
                acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][g], af[i][1], acc[g][1], 0, 0, 0);
            }
    }
}

// after the waves have stored their tiles: the sum for element (row ml of 32, column c of 16) of tile g
template <int NB>
__device__ __forceinline__ float rs_sum(const float (*red)[NB][2][64][4], int g, int ml, int c) {
    const int rt = ml >> 4, l = (ml & 15) + 16 * (c >> 2), r = c & 3;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < RS_WAVES; ++w) s += red[w][g][rt][l][r];
    return s;
}

template <int CELL>
__global__ __launch_bounds__(512) void rnn_step_fwd_mfma_kernel(const bf16_t* __restrict__ xw,
                                                                const bf16_t* __restrict__ ut, int ldu,
                                                                const float* __restrict__ b_rec,
                                                                bf16_t* __restrict__ hs_ext, float* __restrict__ cs_ext,
                                                                bf16_t* __restrict__ y, bf16_t* __restrict__ gates,
                                                                bf16_t* __restrict__ rh, int B, int Tn, int H, int t) {
    constexpr int G = CELL == RNN_LSTM ? 4 : 3;
    __shared__ float red[RS_WAVES][2][2][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15;
    const int u0 = blockIdx.x * RS_NU;
    const int un = min(u0 + (fr & 7), H - 1);
    // tile 0 = gates 0 | 1, tile 1 = gate 2 | gate 3 (LSTM; a GRU repeats gate 2 there, unused)
    const bf16_t* wrow[2] = {ut + (long)((fr >> 3) * H + un) * ldu,
                             ut + (long)((G == 4 ? 2 + (fr >> 3) : 2) * H + un) * ldu};
    const int ml = tid >> 3, n = tid & 7;
    for (int mb = 0; mb < B; mb += 32) {
        // the element this thread finishes, and its operands (requested now, used after the product)
        const int b = mb + ml, j = u0 + n;
        const bool mine = tid < 256 && b < B && j < H;
        const long xo = ((long)(mine ? b : 0) * Tn + t) * G * H + (mine ? j : 0);
        const long so = ((long)(mine ? b : 0) * (Tn + 1) + t) * H + (mine ? j : 0);
        float xin[G], br[G], prev = 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            xin[g] = (float)xw[xo + g * H];
            br[g] = b_rec ? b_rec[g * H + (mine ? j : 0)] : 0.f;
        }
        if (CELL == RNN_GRU) prev = (float)hs_ext[so];
        else prev = cs_ext[so];
        const int r0 = min(mb + fr, B - 1), r1 = min(mb + 16 + fr, B - 1);
        f32x4_t acc[2][2];
        rs_zero<2>(acc);
        rs_contract<2>(hs_ext + ((long)r0 * (Tn + 1) + t) * H, hs_ext + ((long)r1 * (Tn + 1) + t) * H, wrow, H, wave,
                       lane, acc);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            *(f32x4_t*)&red[wave][g][0][lane][0] = acc[g][0];
            *(f32x4_t*)&red[wave][g][1][lane][0] = acc[g][1];
        }
        __syncthreads();
        if (mine) {
            float hu[G];
#pragma unroll
            for (int g = 0; g < G; ++g) hu[g] = rs_sum<2>(red, g >> 1, ml, (g & 1) * 8 + n) + br[g];
            const long oo = ((long)b * Tn + t) * H + j;
            if (CELL == RNN_GRU) {
                const float z = sigmoid_f(xin[0] + hu[0]);
                const float r = sigmoid_f(xin[1] + hu[1]);
                const float hh = tanhf(xin[2] + r * hu[2]);
                const bf16_t hq = (bf16_t)(z * prev + (1.f - z) * hh);
                hs_ext[so + H] = hq;
                y[oo] = hq;
                gates[xo] = (bf16_t)z;
                gates[xo + H] = (bf16_t)r;
                gates[xo + 2 * H] = (bf16_t)hh;
                rh[oo] = (bf16_t)hu[2];
            } else {
                const float i = sigmoid_f(xin[0] + hu[0]);
                const float f = sigmoid_f(xin[1] + hu[1]);
                const float gg = tanhf(xin[2] + hu[2]);
                const float o = sigmoid_f(xin[G - 1] + hu[G - 1]);
                const float c = f * prev + i * gg;
                const bf16_t hq = (bf16_t)(o * tanhf(c));
                cs_ext[so + H] = c;
                hs_ext[so + H] = hq;
                y[oo] = hq;
                gates[xo] = (bf16_t)i;
                gates[xo + H] = (bf16_t)f;
                gates[xo + 2 * H] = (bf16_t)gg;
                gates[xo + (G - 1) * H] = (bf16_t)o;
            }
        }
        __syncthreads();
    }
}

// Step t of the backward recurrence: dh[t] = dy[t] + carry_h + dhu[t+1] U^T for the workgroup's units
// (dhu_ext slot t+1; slot T is zero), then the gate derivatives of step t for those units: dxw[t],
// dhu_ext[t] and the element-wise part of what flows on to step t-1 (carry_h / carry_c).
template <int CELL>
__global__ __launch_bounds__(512) void rnn_step_bwd_mfma_kernel(const bf16_t* __restrict__ dy,
                                                                const bf16_t* __restrict__ u, int ldu,
                                                                const bf16_t* __restrict__ hs_ext,
                                                                const float* __restrict__ cs_ext,
                                                                const bf16_t* __restrict__ gates,
                                                                const bf16_t* __restrict__ rh,
                                                                float* __restrict__ carry_h, float* __restrict__ carry_c,
                                                                bf16_t* __restrict__ dxw, bf16_t* __restrict__ dhu_ext,
                                                                int B, int Tn, int H, int t) {
    constexpr int G = CELL == RNN_LSTM ? 4 : 3;
    __shared__ float red[RS_WAVES][1][2][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15;
    const int u0 = blockIdx.x * RS_NU;
    // columns 8..15 of the tile repeat the 8 units (same cache lines, results unused)
    const bf16_t* wrow[1] = {u + (long)min(u0 + (fr & 7), H - 1) * ldu};
    const int K = G * H;
    const int ml = tid >> 3, n = tid & 7;
    for (int mb = 0; mb < B; mb += 32) {
        const int b = mb + ml, j = u0 + n;
        const bool mine = tid < 256 && b < B && j < H;
        const int idx = mine ? b * H + j : 0;
        const long go = ((long)(mine ? b : 0) * Tn + t) * K + (mine ? j : 0);
        const long eo = ((long)(mine ? b : 0) * (Tn + 1) + t) * K + (mine ? j : 0);
        const long so = ((long)(mine ? b : 0) * (Tn + 1) + t) * H + (mine ? j : 0);
        const long oo = ((long)(mine ? b : 0) * Tn + t) * H + (mine ? j : 0);
        // operands of the gate derivatives, requested before the product
        float gt[G], e0, e1, e2;
#pragma unroll
        for (int g = 0; g < G; ++g) gt[g] = (float)gates[go + g * H];
        const float dyv = (float)dy[oo];
        const float ch = carry_h[idx];
        if (CELL == RNN_GRU) {
            e0 = (float)rh[oo];
            e1 = (float)hs_ext[so];
            e2 = 0.f;
        } else {
            e0 = cs_ext[so];
            e1 = cs_ext[so + H];
            e2 = carry_c[idx];
        }
        const int r0 = min(mb + fr, B - 1), r1 = min(mb + 16 + fr, B - 1);
        f32x4_t acc[1][2];
        rs_zero<1>(acc);
        rs_contract<1>(dhu_ext + ((long)r0 * (Tn + 1) + t + 1) * K, dhu_ext + ((long)r1 * (Tn + 1) + t + 1) * K, wrow,
                       K, wave, lane, acc);
        *(f32x4_t*)&red[wave][0][0][lane][0] = acc[0][0];
        *(f32x4_t*)&red[wave][0][1][lane][0] = acc[0][1];
        __syncthreads();
        if (mine) {
            const float dh = dyv + ch + rs_sum<1>(red, 0, ml, n);
            if (CELL == RNN_GRU) {
                const float z = gt[0], r = gt[1], hh = gt[2];
                const float da_h = dh * (1.f - z) * (1.f - hh * hh);
                const float da_z = dh * (e1 - hh) * z * (1.f - z);
                const float da_r = da_h * e0 * r * (1.f - r);
                dxw[go] = (bf16_t)da_z;
                dxw[go + H] = (bf16_t)da_r;
                dxw[go + 2 * H] = (bf16_t)da_h;
                dhu_ext[eo] = (bf16_t)da_z;
                dhu_ext[eo + H] = (bf16_t)da_r;
                dhu_ext[eo + 2 * H] = (bf16_t)(da_h * r);
                carry_h[idx] = dh * z;
            } else {
                const float i = gt[0], f = gt[1], gg = gt[2], o = gt[G - 1];
                const float tc = tanhf(e1);
                const float dc = e2 + dh * o * (1.f - tc * tc);
                const float da[4] = {dc * gg * i * (1.f - i), dc * e0 * f * (1.f - f), dc * i * (1.f - gg * gg),
                                     dh * tc * o * (1.f - o)};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k < G) {
                        dxw[go + k * H] = (bf16_t)da[k];
                        dhu_ext[eo + k * H] = (bf16_t)da[k];
                    }
                }
                carry_c[idx] = dc * f;
            }
        }
        __syncthreads();
    }
}

// one launch per step is used for bf16 unless ATHENA_RNN_STEP_GEMM=1 asks for the GEMM + gate kernel pair
static bool rnn_fused_steps() {
    static const bool on = [] {
        const char* e = getenv("ATHENA_RNN_STEP_GEMM");
        return !(e && e[0] == '1');
    }();
    return on;
}

static int rnn_check(const char* name, int cell, int B, int T, int H, int dtype) {
    ATHENA_REQUIRE(cell == RNN_GRU || cell == RNN_LSTM, ATHENA_E_UNSUPPORTED, "%s: cell %d (0 = gru, 1 = lstm)", name, cell);
    ATHENA_REQUIRE(B > 0 && T > 0 && H > 0, ATHENA_E_SHAPE, "%s: bad shape B=%d T=%d H=%d", name, B, T, H);
    ATHENA_REQUIRE(H % 8 == 0, ATHENA_E_ALIGN, "%s: hidden size %d must be a multiple of 8", name, H);
    ATHENA_REQUIRE(dtype == ATHENA_F32 || dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "%s: dtype", name);
    ATHENA_REQUIRE((long)B * (T + 1) * 4 * H < (1L << 31), ATHENA_E_SHAPE, "%s: more than 2^31 gate elements", name);
    return ATHENA_OK;
}

static inline size_t esize(int dtype) { return dtype == ATHENA_BF16 ? 2 : 4; }

// zero slot ``slot`` of every batch row of a [B, slots, width] array
static int zero_slot(void* base, int B, int slots, int width, size_t es, int slot, hipStream_t s, const char* name) {
    hipError_t e = hipMemset2DAsync((char*)base + (size_t)slot * width * es, (size_t)slots * width * es, 0,
                                    (size_t)width * es, (size_t)B, s);
    ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "%s: memset failed: %s", name, hipGetErrorString(e));
    return ATHENA_OK;
}

extern "C" int64_t athena_rnn_workspace_bytes(int cell, int B, int H) {
    if (B <= 0 || H <= 0) return 0;
    const int G = cell == RNN_LSTM ? 4 : 3;
    // forward: h U of one step [B, G*H]; backward: the two carried gradients [B, H] each
    const int64_t fwd = (int64_t)sizeof(float) * B * G * H, bwd = (int64_t)sizeof(float) * 2 * B * H;
    return fwd > bwd ? fwd : bwd;
}

extern "C" int athena_rnn_fwd(int cell, const void* xw, const void* u, const void* u_t, const float* b_rec, void* hs_ext,
                              float* cs_ext, void* y, void* gates, void* rh, int B, int T, int H, int dtype,
                              void* workspace, int64_t workspace_bytes, athena_stream_t stream) {
    int rc = rnn_check("athena_rnn_fwd", cell, B, T, H, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(xw && u && hs_ext && y && gates && workspace, ATHENA_E_SHAPE, "athena_rnn_fwd: null operand");
    ATHENA_REQUIRE(cell == RNN_LSTM ? cs_ext != nullptr : rh != nullptr, ATHENA_E_SHAPE,
                   "athena_rnn_fwd: the cell's state array (cs_ext for lstm, rh for gru) is missing");
    ATHENA_REQUIRE(workspace_bytes >= athena_rnn_workspace_bytes(cell, B, H) && athena_aligned16(workspace),
                   ATHENA_E_SHAPE, "athena_rnn_fwd: workspace too small or misaligned");
    const int G = cell == RNN_LSTM ? 4 : 3;
    const size_t es = esize(dtype);
    hipStream_t s = (hipStream_t)stream;
    rc = zero_slot(hs_ext, B, T + 1, H, es, 0, s, "athena_rnn_fwd");
    if (rc) return rc;
    if (cell == RNN_LSTM) {
        rc = zero_slot(cs_ext, B, T + 1, H, sizeof(float), 0, s, "athena_rnn_fwd");
        if (rc) return rc;
    }
    if (dtype == ATHENA_BF16 && u_t && rnn_fused_steps()) {
        // the transposed copy [G*H, H] has the contraction contiguous in every weight row
        ATHENA_REQUIRE(athena_aligned16(u_t) && athena_aligned16(hs_ext), ATHENA_E_ALIGN, "athena_rnn_fwd: 16-byte alignment required");
        const dim3 grid((H + RS_NU - 1) / RS_NU);
        for (int t = 0; t < T; ++t) {
            if (cell == RNN_GRU)
                ATHENA_LAUNCH(rnn_step_fwd_mfma_kernel<RNN_GRU>, grid, dim3(512), 0, s, (const bf16_t*)xw, (const bf16_t*)u_t, H, b_rec, (bf16_t*)hs_ext, cs_ext, (bf16_t*)y, (bf16_t*)gates, (bf16_t*)rh, B, T, H, t);
            else
                ATHENA_LAUNCH(rnn_step_fwd_mfma_kernel<RNN_LSTM>, grid, dim3(512), 0, s, (const bf16_t*)xw, (const bf16_t*)u_t, H, b_rec, (bf16_t*)hs_ext, cs_ext, (bf16_t*)y, (bf16_t*)gates, (bf16_t*)rh, B, T, H, t);
        }
        ATHENA_CHECK_LAUNCH("athena_rnn_fwd");
        return ATHENA_OK;
    }
    float* hu = (float*)workspace;
    athena_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.B = u;
    d.C = hu;
    d.bias = b_rec;
    d.M = B; d.N = G * H; d.K = H;
    d.lda = (T + 1) * H; d.ldb = G * H; d.ldc = G * H;
    d.in_dtype = dtype;
    d.out_dtype = ATHENA_F32;
    d.split_k = 1;
    d.alpha = 1.f;
    const dim3 grid((unsigned)(((long)B * H + 255) / 256));
    for (int t = 0; t < T; ++t) {
        d.A = (const char*)hs_ext + (size_t)t * H * es;
        rc = athena_gemm(&d, stream);
        if (rc) return rc;
        if (cell == RNN_GRU) {
            if (dtype == ATHENA_BF16)
                ATHENA_LAUNCH(gru_step_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)xw, hu, (bf16_t*)hs_ext, (bf16_t*)y, (bf16_t*)gates, (bf16_t*)rh, B, T, H, t);
            else
                ATHENA_LAUNCH(gru_step_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)xw, hu, (float*)hs_ext, (float*)y, (float*)gates, (float*)rh, B, T, H, t);
        } else {
            if (dtype == ATHENA_BF16)
                ATHENA_LAUNCH(lstm_step_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)xw, hu, (bf16_t*)hs_ext, cs_ext, (bf16_t*)y, (bf16_t*)gates, B, T, H, t);
            else
                ATHENA_LAUNCH(lstm_step_fwd_kernel<float>, grid, dim3(256), 0, s, (const float*)xw, hu, (float*)hs_ext, cs_ext, (float*)y, (float*)gates, B, T, H, t);
        }
    }
    ATHENA_CHECK_LAUNCH("athena_rnn_fwd");
    return ATHENA_OK;
}

extern "C" int athena_rnn_bwd(int cell, const void* dy, const void* u, const void* hs_ext, const float* cs_ext,
                              const void* gates, const void* rh, void* dxw, void* dhu_ext, int B, int T, int H,
                              int dtype, void* workspace, int64_t workspace_bytes, athena_stream_t stream) {
    int rc = rnn_check("athena_rnn_bwd", cell, B, T, H, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(dy && u && hs_ext && gates && dxw && dhu_ext && workspace, ATHENA_E_SHAPE, "athena_rnn_bwd: null operand");
    ATHENA_REQUIRE(cell == RNN_LSTM ? cs_ext != nullptr : rh != nullptr, ATHENA_E_SHAPE,
                   "athena_rnn_bwd: the cell's state array (cs_ext for lstm, rh for gru) is missing");
    ATHENA_REQUIRE(workspace_bytes >= athena_rnn_workspace_bytes(cell, B, H) && athena_aligned16(workspace),
                   ATHENA_E_SHAPE, "athena_rnn_bwd: workspace too small or misaligned");
    const int G = cell == RNN_LSTM ? 4 : 3;
    const size_t es = esize(dtype);
    hipStream_t s = (hipStream_t)stream;
    float* carry_h = (float*)workspace;
    float* carry_c = carry_h + (size_t)B * H;
    hipError_t e = hipMemsetAsync(carry_h, 0, sizeof(float) * 2 * B * H, s);
    ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_rnn_bwd: memset failed");
    rc = zero_slot(dhu_ext, B, T + 1, G * H, es, T, s, "athena_rnn_bwd");
    if (rc) return rc;
    if (dtype == ATHENA_BF16 && rnn_fused_steps()) {
        ATHENA_REQUIRE(athena_aligned16(u) && athena_aligned16(dhu_ext), ATHENA_E_ALIGN, "athena_rnn_bwd: 16-byte alignment required");
        const dim3 grid((H + RS_NU - 1) / RS_NU);
        for (int t = T - 1; t >= 0; --t) {
            if (cell == RNN_GRU)
                ATHENA_LAUNCH(rnn_step_bwd_mfma_kernel<RNN_GRU>, grid, dim3(512), 0, s, (const bf16_t*)dy, (const bf16_t*)u, G * H, (const bf16_t*)hs_ext, cs_ext, (const bf16_t*)gates, (const bf16_t*)rh, carry_h, carry_c, (bf16_t*)dxw, (bf16_t*)dhu_ext, B, T, H, t);
            else
                ATHENA_LAUNCH(rnn_step_bwd_mfma_kernel<RNN_LSTM>, grid, dim3(512), 0, s, (const bf16_t*)dy, (const bf16_t*)u, G * H, (const bf16_t*)hs_ext, cs_ext, (const bf16_t*)gates, (const bf16_t*)rh, carry_h, carry_c, (bf16_t*)dxw, (bf16_t*)dhu_ext, B, T, H, t);
        }
        ATHENA_CHECK_LAUNCH("athena_rnn_bwd");
        return ATHENA_OK;
    }
    athena_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.B = u;
    d.C = carry_h;
    d.M = B; d.N = H; d.K = G * H;
    d.trans_b = 1;                     // B(k, n) = U[n][k]: the product with U^T
    d.lda = (T + 1) * G * H; d.ldb = G * H; d.ldc = H;
    d.in_dtype = dtype;
    d.out_dtype = ATHENA_F32;
    d.accumulate = 1;
    d.split_k = 1;
    d.alpha = 1.f;
    const dim3 grid((unsigned)(((long)B * H + 255) / 256));
    for (int t = T - 1; t >= 0; --t) {
        if (cell == RNN_GRU) {
            if (dtype == ATHENA_BF16)
                ATHENA_LAUNCH(gru_step_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)hs_ext, (const bf16_t*)gates, (const bf16_t*)rh, carry_h, (bf16_t*)dxw, (bf16_t*)dhu_ext, B, T, H, t);
            else
                ATHENA_LAUNCH(gru_step_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, (const float*)hs_ext, (const float*)gates, (const float*)rh, carry_h, (float*)dxw, (float*)dhu_ext, B, T, H, t);
        } else {
            if (dtype == ATHENA_BF16)
                ATHENA_LAUNCH(lstm_step_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dy, cs_ext, (const bf16_t*)gates, carry_h, carry_c, (bf16_t*)dxw, (bf16_t*)dhu_ext, B, T, H, t);
            else
                ATHENA_LAUNCH(lstm_step_bwd_kernel<float>, grid, dim3(256), 0, s, (const float*)dy, cs_ext, (const float*)gates, carry_h, carry_c, (float*)dxw, (float*)dhu_ext, B, T, H, t);
        }
        if (t > 0) {
            d.A = (const char*)dhu_ext + (size_t)t * G * H * es;
            rc = athena_gemm(&d, stream);
            if (rc) return rc;
        }
    }
    ATHENA_CHECK_LAUNCH("athena_rnn_bwd");
    return ATHENA_OK;
}
