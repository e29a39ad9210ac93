// First front-end layer, fused: Conv2D(3x3, stride 2, same, ONE input channel) -> BatchNormalization
// (batch statistics) -> relu6, forward and backward
// (reference athena/models/speech_transformer.py:72-93, the fbank input has one channel).
//
// With one input channel a pre-activation is x[p][c] = sum_t a[p][t] w[t][c] over the 9 taps
// a[p][.] of the input around output pixel p, so everything BatchNormalization needs from the
// [pixels x channels] tensor is a function of the 9 + 45 input moments
//     S[t] = sum_p a[p][t],   R[t][u] = sum_p a[p][t] a[p][u]
// and the 9 x C weights: the pre-activation tensor is never written.
//   forward : moments -> per-channel sum / sum of squares (closed form, fp64) -> mean, rstd;
//             one pass recomputes the 9 multiply-adds per element, applies BN + relu6 and stores y.
//   backward: g = dL/d(BN output) comes in already gated by relu6 (the producer masks it with the
//             stored y); one pass reduces T[t][c] = sum_p a[p][t] g[p][c] and sb[c] = sum_p g[p][c];
//             dbeta = sb, dgamma = rstd (sum_t w T - mean sb), and
//             dw[t][c] = gamma rstd ( T - sb/n S[t] - dgamma/n rstd (sum_u R[t][u] w[u][c] - mean S[t]) ).
#include "common.h"
#include <type_traits>
#include <algorithm>
#include <cstdlib>

#define FC_K 9                      // taps
#define FC_NMOM (FC_K + FC_K * FC_K)
#define FC_NTRI (FC_K + FC_K * (FC_K + 1) / 2)   // tap sums + upper triangle of the tap products: 54 <= 64
#define FC_MOM_BLOCKS 512                         // workgroups (= slabs) of the moments pass

static inline void fc_same_pad(int n, int* out, int* pad_before) {
    int o = (n + 1) / 2;
    int total = (o - 1) * 2 + 3 - n;
    if (total < 0) total = 0;
    *out = o;
    *pad_before = total / 2;
}

// the 9 taps around output pixel (b, ho, wo); zero outside the image
__device__ __forceinline__ void fc_taps(const float* __restrict__ in, int H, int W, int b, int ho, int wo, int ph,
                                        int pw, float (&a)[FC_K]) {
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * ho - ph + kh;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int x = 2 * wo - pw + kw;
            const bool ok = h >= 0 && h < H && x >= 0 && x < W;
            const int o = ok ? (b * H + h) * W + x : 0;        // < 2^31 elements (checked on the host)
            const float v = in[o];
            a[kh * 3 + kw] = ok ? v : 0.f;
        }
    }
}

// ---------------------------------------------------------------- input moments
__global__ __launch_bounds__(256) void fc_moments_kernel(const float* __restrict__ in, int B, int H, int W, int Ho,
                                                         int Wo, int ph, int pw, double* __restrict__ slabs) {
    __shared__ double red[4][FC_NTRI];
    const long npix = (long)B * Ho * Wo;
    // fp64 throughout: the variance is a difference of these sums
    double s[FC_K], r[FC_K * (FC_K + 1) / 2];
#pragma unroll
    for (int t = 0; t < FC_K; ++t) s[t] = 0.0;
#pragma unroll
    for (int t = 0; t < FC_K * (FC_K + 1) / 2; ++t) r[t] = 0.0;
    for (int pix = blockIdx.x * 256 + threadIdx.x; pix < (int)npix; pix += gridDim.x * 256) {
        const int q = pix / Wo, wo = pix - q * Wo;
        const int b = q / Ho, ho = q - b * Ho;
        float a[FC_K];
        fc_taps(in, H, W, b, ho, wo, ph, pw, a);
        int idx = 0;
#pragma unroll
        for (int t = 0; t < FC_K; ++t) {
            s[t] += (double)a[t];
#pragma unroll
            for (int u = t; u < FC_K; ++u) { r[idx] = fma((double)a[t], (double)a[u], r[idx]); ++idx; }
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    auto wave_sum_f64 = [](double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    };
#pragma unroll
    for (int t = 0; t < FC_K; ++t) {
        const double v = wave_sum_f64(s[t]);
        if (lane == 0) red[w][t] = v;
    }
#pragma unroll
    for (int t = 0; t < FC_K * (FC_K + 1) / 2; ++t) {
        const double v = wave_sum_f64(r[t]);
        if (lane == 0) red[w][FC_K + t] = v;
    }
    __syncthreads();
    if (threadIdx.x < FC_NTRI) {
        const int i = threadIdx.x;
        slabs[(long)blockIdx.x * FC_NTRI + i] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
    }
}

// sums the per-workgroup slabs in a fixed order and expands the triangle into the full 9x9 matrix:
// mom[0..9) = tap sums, mom[9 + t*9 + u] = tap products.  One workgroup, 16 waves x 64 entries.
__global__ __launch_bounds__(1024) void fc_moments_fold_kernel(const double* __restrict__ slabs, int nslab,
                                                               double* __restrict__ mom) {
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double a0 = 0.0, a1 = 0.0;
    if (lane < FC_NTRI) {
        int b = w;
        for (; b + 16 < nslab; b += 32) {
            a0 += slabs[(long)b * FC_NTRI + lane];
            a1 += slabs[(long)(b + 16) * FC_NTRI + lane];
        }
        if (b < nslab) a0 += slabs[(long)b * FC_NTRI + lane];
    }
    red[w][lane] = a0 + a1;
    __syncthreads();
    if (w != 0 || lane >= FC_NTRI) return;
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += red[k][lane];
    if (lane < FC_K) {
        mom[lane] = v;
    } else {
        // triangular index -> (t, u), t <= u; the matrix is stored full
        int k = lane - FC_K, t = 0;
        while (k >= FC_K - t) { k -= FC_K - t; ++t; }
        const int u = t + k;
        mom[FC_K + t * FC_K + u] = v;
        mom[FC_K + u * FC_K + t] = v;
    }
}

// per-channel sum and sum of squares of the pre-activations, from the moments
__global__ void fc_colstats_kernel(const double* __restrict__ mom, const float* __restrict__ w, int C,
                                   double* __restrict__ stats) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double wc[FC_K];
#pragma unroll
    for (int t = 0; t < FC_K; ++t) wc[t] = (double)w[t * C + c];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int t = 0; t < FC_K; ++t) {
        s1 += wc[t] * mom[t];
        double row = 0.0;
#pragma unroll
        for (int u = 0; u < FC_K; ++u) row += wc[u] * mom[FC_K + t * FC_K + u];
        s2 += wc[t] * row;
    }
    stats[c] = s1;
    stats[C + c] = s2 < 0.0 ? 0.0 : s2;
}

extern "C" int64_t athena_conv1_moments_len(void) { return FC_NMOM + 6 + (int64_t)FC_MOM_BLOCKS * FC_NTRI; }

extern "C" int athena_conv1_bn_stats(const float* in, const float* w_hwio, int B, int H, int W, int Cout,
                                     double* moments, double* colstats, athena_stream_t stream) {
    ATHENA_REQUIRE(in && w_hwio && moments && colstats && B > 0 && H > 0 && W > 0 && Cout > 0 &&
                       (long)B * H * W < (1L << 31),
                   ATHENA_E_SHAPE, "athena_conv1_bn_stats: bad arguments");
    int Ho, Wo, ph, pw;
    fc_same_pad(H, &Ho, &ph);
    fc_same_pad(W, &Wo, &pw);
    hipStream_t s = (hipStream_t)stream;
    // moments[0, 90): the result; behind it the per-workgroup slabs (no atomics, fixed order)
    double* slabs = moments + FC_NMOM + 6;            // 96 doubles in: keeps the slabs 256-byte aligned with the base
    const long npix = (long)B * Ho * Wo;
    long blocks = (npix + 255) / 256;
    if (blocks > FC_MOM_BLOCKS) blocks = FC_MOM_BLOCKS;
    ATHENA_LAUNCH(fc_moments_kernel, dim3((unsigned)blocks), dim3(256), 0, s, in, B, H, W, Ho, Wo, ph, pw, slabs);
    ATHENA_LAUNCH(fc_moments_fold_kernel, dim3(1), dim3(1024), 0, s, (const double*)slabs, (int)blocks, moments);
    ATHENA_LAUNCH(fc_colstats_kernel, dim3((Cout + 127) / 128), dim3(128), 0, s, (const double*)moments, w_hwio,
                       Cout, colstats);
    ATHENA_CHECK_LAUNCH("athena_conv1_bn_stats");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- forward: conv + BN + relu6 in one pass
// thread = (pixel slot, 8 channels); its 72 weights and the BN scale/shift stay in registers
template <typename T> __device__ __forceinline__ T fc_store_cast(float y);
template <> __device__ __forceinline__ float fc_store_cast<float>(float y) { return y; }
template <> __device__ __forceinline__ bf16_t fc_store_cast<bf16_t>(float y) {
    // a value below 6 must not round up to 6: the backward pass reads the relu6 gate off the
    // stored value, so it is kept at the largest bf16 below 6 instead
    bf16_t r = (bf16_t)y;
    if (y < 6.f && (float)r >= 6.f) r = __builtin_bit_cast(bf16_t, (unsigned short)0x40BF);
    return r;
}

template <typename T>
__global__ __launch_bounds__(256) void fc_fwd_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ s0, const float* __restrict__ s1,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, int infer, int B, int H, int W, int C, int Ho, int Wo,
                                                     int ph, int pw, T* __restrict__ out, int pix_begin) {
    const int cpr = C / 8;                           // 8-channel chunks per pixel; host guarantees 256 % cpr == 0
    const int chunk = threadIdx.x % cpr, slot = threadIdx.x / cpr, slots = 256 / cpr;
    const int c0 = chunk * 8;
    float wr[FC_K][8], sc[8], sh[8];
#pragma unroll
    for (int t = 0; t < FC_K; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) wr[t][i] = w[t * C + c0 + i];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float mu = s0[c0 + i];
        const float rs = infer ? rsqrtf(s1[c0 + i] + eps) : s0[C + c0 + i];
        sc[i] = rs * gamma[c0 + i];
        sh[i] = beta[c0 + i] - mu * sc[i];
    }
    const long npix = (long)B * Ho * Wo;
    for (int pix = pix_begin + blockIdx.x * slots + slot; pix < (int)npix; pix += gridDim.x * slots) {
        const int q = pix / Wo, wo = pix - q * Wo;
        const int b = q / Ho, ho = q - b * Ho;
        float a[FC_K];
        fc_taps(in, H, W, b, ho, wo, ph, pw, a);
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
        for (int t = 0; t < FC_K; ++t)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = fmaf(a[t], wr[t][i], acc[i]);
        if (sizeof(T) == 2) {
            bf16x8_t o;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fc_store_cast<bf16_t>(relu6f(fmaf(acc[i], sc[i], sh[i])));
            *(bf16x8_t*)((bf16_t*)out + (long)pix * C + c0) = o;
        } else {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = relu6f(fmaf(acc[i], sc[i], sh[i]));
            float* op = (float*)out + (long)pix * C + c0;
            *(float4*)op = make_float4(o[0], o[1], o[2], o[3]);
            *(float4*)(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
    }
}

// The same layer on the matrix cores for a bf16 result.  Per output pixel the nine products and the
// BatchNorm shift are one contraction of length 32,
//     y[c] = sum_t a[t] * (w[t][c] * scale[c]) + shift[c],
// with every fp32 factor split into two bf16 halves (x = hi + lo) and the three products that
// matter kept: a_hi w_hi + a_hi w_lo + a_lo w_hi (error ~2^-17 of a term, far below the bf16
// result).  The 32 contraction slots, in groups of eight as the MFMA lanes hold them:
//     group 0: a_hi[0..7] x w_hi[0..7]      group 1: a_hi[0..7] x w_lo[0..7]
//     group 2: a_lo[0..7] x w_hi[0..7]      group 3: a_hi[8] w_hi[8], a_hi[8] w_lo[8], a_lo[8] w_hi[8],
//                                                    1 x shift_hi, 1 x shift_lo, 0, 0, 0
// Channels are the MFMA rows, pixels its columns; the rows of two consecutive products are
// interleaved so that a lane ends up with eight consecutive channels of one pixel (one 16-byte
// store).  A wave takes 16 pixels x 32 * NPAIR channels per trip; the weight fragments stay in
// registers.  The vector ALUs only decode the taps and clamp / convert the result.
__device__ __forceinline__ void fc_split(float x, bf16_t& hi, bf16_t& lo) {
    hi = (bf16_t)x;
    lo = (bf16_t)(x - (float)hi);
}

// Weight fragments of the matrix-core form (see fc_fwd_mfma_kernel): the CB = 32 * NPAIR channels from
// ``cb`` on, scaled by rstd * gamma, with the BatchNorm shift as the last contraction slots.  sW is
// workgroup scratch; all 256 threads take part.
template <int NPAIR, int NTHREADS = 256>
__device__ __forceinline__ void fc_weight_frags(const float* __restrict__ w, const float* __restrict__ s0,
                                                const float* __restrict__ s1, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float eps, int infer, int C, int cb,
                                                float (&sW)[FC_K + 1][32 * NPAIR], bf16x8_t (&wf)[NPAIR][2]) {
    constexpr int CB = 32 * NPAIR;
    const int tid = threadIdx.x, lane = tid & 63, kg = lane >> 4, li = lane & 15;
    for (int i = tid; i < CB; i += NTHREADS) {
        const int c = cb + i;
        const float mu = s0[c];
        const float rs = infer ? rsqrtf(s1[c] + eps) : s0[C + c];
        const float k = rs * gamma[c];
#pragma unroll
        for (int t = 0; t < FC_K; ++t) sW[t][i] = w[t * C + c] * k;
        sW[FC_K][i] = beta[c] - mu * k;
    }
    __syncthreads();
    // A-operand fragments: row m = lane & 15 of product j of pair P is channel 32 P + 8 (m >> 2) + 4 j + (m & 3)
#pragma unroll
    for (int P = 0; P < NPAIR; ++P)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = 32 * P + 8 * (li >> 2) + 4 * j + (li & 3);
            bf16x8_t f;
            if (kg < 3) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    bf16_t hi, lo;
                    fc_split(sW[e][ch], hi, lo);
                    f[e] = kg == 1 ? lo : hi;
                }
            } else {
                bf16_t whi, wlo, shi, slo;
                fc_split(sW[8][ch], whi, wlo);
                fc_split(sW[FC_K][ch], shi, slo);
                const bf16_t z = (bf16_t)0.f;
                f = (bf16x8_t){whi, wlo, whi, shi, slo, z, z, z};
            }
            wf[P][j] = f;
        }
}

// the nine input values around a pixel as loaded (address clamped into the image) and the bit
// mask of those inside it; the mask is applied when the values are used, a trip later
struct FcTaps { float v[FC_K]; uint32_t ok; };
__device__ __forceinline__ void fc_fetch_taps(const float* __restrict__ in, int H, int W, int Ho, int Wo, int ph,
                                              int pw, int pix, FcTaps& t) {
    const int q = pix / Wo, wo = pix - q * Wo;
    const int b = q / Ho, ho = q - b * Ho;
    // addresses first (rows and columns clamped into the image: branch-free and always valid),
    // then the nine loads back to back
    uint32_t okm = 0;
    int idx[FC_K];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * ho - ph + kh;
        const int hc = min(max(h, 0), H - 1);
        const uint32_t hok = (h == hc) ? 1u : 0u;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int x = 2 * wo - pw + kw;
            const int xc = min(max(x, 0), W - 1);
            idx[kh * 3 + kw] = (b * H + hc) * W + xc;
            okm |= (hok & ((x == xc) ? 1u : 0u)) << (kh * 3 + kw);
        }
    }
#pragma unroll
    for (int i = 0; i < FC_K; ++i) t.v[i] = in[idx[i]];
    t.ok = okm;
}
// the same from the pixel's coordinates (a caller that walks the pixels with a fixed stride keeps them
// up to date with additions instead of dividing every time); 32-bit offsets from one base
__device__ __forceinline__ void fc_fetch_taps_at(const float* __restrict__ in, int H, int W, int ph, int pw, int b,
                                                 int ho, int wo, FcTaps& t) {
    uint32_t okm = 0;
    uint32_t idx[FC_K];
    const int rowbase = b * H;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int h = 2 * ho - ph + kh;
        const int hc = min(max(h, 0), H - 1);
        const uint32_t hok = (h == hc) ? 1u : 0u;
        const int rb = (rowbase + hc) * W;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int x = 2 * wo - pw + kw;
            const int xc = min(max(x, 0), W - 1);
            idx[kh * 3 + kw] = (uint32_t)(rb + xc);
            okm |= (hok & ((x == xc) ? 1u : 0u)) << (kh * 3 + kw);
        }
    }
#pragma unroll
    for (int i = 0; i < FC_K; ++i) t.v[i] = in[idx[i]];
    t.ok = okm;
}
// B-operand fragment of a pixel from its nine taps (kg = lane >> 4: the group of contraction slots)
__device__ __forceinline__ bf16x8_t fc_tap_frag(const FcTaps& t, int kg) {
    bf16_t hi[FC_K], lo[FC_K];
#pragma unroll
    for (int i = 0; i < FC_K; ++i) fc_split(((t.ok >> i) & 1u) ? t.v[i] : 0.f, hi[i], lo[i]);
    bf16x8_t f;
    if (kg < 3) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = kg == 2 ? lo[e] : hi[e];
    } else {
        const bf16_t one = (bf16_t)1.f, z = (bf16_t)0.f;
        f = (bf16x8_t){hi[8], hi[8], lo[8], one, one, z, z, z};
    }
    return f;
}

template <int NPAIR>
__global__ __launch_bounds__(256) void fc_fwd_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                          const float* __restrict__ s0, const float* __restrict__ s1,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, int infer, int B,
                                                          int H, int W, int C, int Ho, int Wo, int ph, int pw,
                                                          bf16_t* __restrict__ out) {
    constexpr int CB = 32 * NPAIR;                   // channels per workgroup (blockIdx.y picks the block)
    constexpr int SROW = CB + 8;                     // padded row of the output staging tile (elements)
    constexpr int LPR = CB / 8, RPI = 64 / LPR;      // lanes per tile row, rows per store instruction
    __shared__ float sW[FC_K + 1][CB];               // scaled weights, then the shift
    __shared__ __attribute__((aligned(16))) bf16_t sO[4][16][SROW];      // one 16-pixel tile per wave
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4, li = lane & 15;
    const int cb = blockIdx.y * CB;
    bf16x8_t wf[NPAIR][2];
    fc_weight_frags<NPAIR>(w, s0, s1, gamma, beta, eps, infer, C, cb, sW, wf);
    // the launch covers whole 64-pixel blocks only (the host sends a ragged tail to the vector
    // kernel): no store below is predicated, so the wait for the next trip's taps can leave this
    // trip's stores in flight
    const int nblk = (B * Ho * Wo) / 64;
    auto fetch = [&](int blk, FcTaps& t) { fc_fetch_taps(in, H, W, Ho, Wo, ph, pw, blk * 64 + wave * 16 + li, t); };
    auto tap_frag = [&](const FcTaps& t) { return fc_tap_frag(t, kg); };
    // one trip: the wave's 16 pixels x CB channels from the taps fetched a trip earlier
    auto emit = [&](int blk, const FcTaps& t) {
        const bf16x8_t pf = tap_frag(t);
        bf16_t* so = &sO[wave][0][0];
#pragma unroll
        for (int P = 0; P < NPAIR; ++P) {
            const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[P][0], pf, zero, 0, 0, 0);
            const f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[P][1], pf, zero, 0, 0, 0);
            bf16x8_t ov;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // relu6, with a value below 6 kept from rounding up to 6 (the backward pass reads the
                // relu6 gate off the stored value): 5.96875 is the largest bf16 below 6
                ov[r] = (bf16_t)(d0[r] >= 6.f ? 6.f : __builtin_amdgcn_fmed3f(d0[r], 0.f, 5.96875f));
                ov[4 + r] = (bf16_t)(d1[r] >= 6.f ? 6.f : __builtin_amdgcn_fmed3f(d1[r], 0.f, 5.96875f));
            }
            *(bf16x8_t*)(so + li * SROW + 32 * P + 8 * kg) = ov;
        }
        // the wave's 16 x CB tile goes out in whole rows: a lane's own pieces are 64-byte islands, a
        // row of the tile is CB * 2 contiguous bytes of the output
        __builtin_amdgcn_wave_barrier();
        const int pix0 = blk * 64 + wave * 16;
#pragma unroll
        for (int it = 0; it < 16 / RPI; ++it) {
            const int row = it * RPI + lane / LPR, col = (lane % LPR) * 8;
            const bf16x8_t v = *(const bf16x8_t*)(so + row * SROW + col);
            *(bf16x8_t*)(out + (long)(pix0 + row) * C + cb + col) = v;
        }
        __builtin_amdgcn_wave_barrier();
    };
    // The taps of the next trip are requested before the current one is worked on, into the other
    // of two register sets (no copies: a copy would wait for the loads AND for the stores issued
    // after them).  The fetch is unconditional -- the last trip re-reads its own block -- so the
    // number of outstanding loads is the same on every path.
    FcTaps t0, t1;
    int blk = blockIdx.x;
    if (blk >= nblk) return;
    fetch(blk, t0);
    while (true) {
        int nb = blk + gridDim.x;
        fetch(nb < nblk ? nb : blk, t1);
        emit(blk, t0);
        if (nb >= nblk) break;
        blk = nb;
        nb = blk + gridDim.x;
        fetch(nb < nblk ? nb : blk, t0);
        emit(blk, t1);
        if (nb >= nblk) break;
        blk = nb;
    }
}

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// channels per thread of the backward reduction for a layer of C channels (16 for wide layers: the
// nine taps of a pixel are then fetched and bounds-checked once per 16 channels)
static inline int fc_cpt(int C) { return (C % 16 == 0 && C >= 128 && 256 % (C / 16) == 0) ? 16 : 8; }

static int fc_check(const char* name, int B, int H, int W, int C, int dtype) {
    ATHENA_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0, ATHENA_E_SHAPE, "%s: bad shape", name);
    ATHENA_REQUIRE((long)B * H * W < (1L << 31), ATHENA_E_SHAPE, "%s: input too large", name);
    ATHENA_REQUIRE(C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, ATHENA_E_UNSUPPORTED,
                   "%s: channel count %d must be 8 * a divisor of 256", name, C);
    ATHENA_REQUIRE(dtype == ATHENA_F32 || dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "%s: dtype", name);
    return ATHENA_OK;
}

extern "C" int athena_conv1_bn_relu6_fwd(const float* in, const float* w_hwio, const float* s0, const float* s1,
                                         const float* gamma, const float* beta, float eps, int infer, int B, int H,
                                         int W, int Cout, int dtype, void* out, athena_stream_t stream) {
    int rc = fc_check("athena_conv1_bn_relu6_fwd", B, H, W, Cout, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(in && w_hwio && s0 && gamma && beta && out && (!infer || s1), ATHENA_E_SHAPE,
                   "athena_conv1_bn_relu6_fwd: null operand");
    ATHENA_REQUIRE(athena_aligned16(out), ATHENA_E_ALIGN, "athena_conv1_bn_relu6_fwd: out must be 16-byte aligned");
    int Ho, Wo, ph, pw;
    fc_same_pad(H, &Ho, &ph);
    fc_same_pad(W, &Wo, &pw);
    const long npix = (long)B * Ho * Wo;
    const int slots = 256 / (Cout / 8);
    long blocks = (npix + slots - 1) / slots;
    if (blocks > 2048) blocks = 2048;
    hipStream_t s = (hipStream_t)stream;
    int pix_begin = 0;
    if (dtype == ATHENA_BF16 && Cout % 32 == 0 && (Cout <= 256 ? (256 % Cout == 0) : (Cout % 256 == 0)) &&
        npix >= 64) {
        // matrix-core path for the whole 64-pixel blocks: a workgroup trip is 64 pixels x up to 256 channels
        const int cbw = Cout < 256 ? Cout : 256;
        long nb = npix / 64;
        pix_begin = (int)(nb * 64);
        if (nb > 1024) nb = 1024;
        const dim3 grid((unsigned)nb, (unsigned)(Cout / cbw));
#define FC_MFMA(NP)                                                                                              \
    ATHENA_LAUNCH(fc_fwd_mfma_kernel<NP>, grid, dim3(256), 0, s, in, w_hwio, s0, s1, gamma, beta, eps, infer, B, \
                       H, W, Cout, Ho, Wo, ph, pw, (bf16_t*)out)
        switch (cbw / 32) {
            case 1: FC_MFMA(1); break;
            case 2: FC_MFMA(2); break;
            case 4: FC_MFMA(4); break;
            default: FC_MFMA(8); break;
        }
#undef FC_MFMA
    }
    if (pix_begin < npix) {
        // everything (fp32), or the ragged tail behind the last whole block
        blocks = (npix - pix_begin + slots - 1) / slots;
        if (blocks > 2048) blocks = 2048;
        if (dtype == ATHENA_BF16)
            ATHENA_LAUNCH(fc_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, in, w_hwio, s0, s1, gamma,
                               beta, eps, infer, B, H, W, Cout, Ho, Wo, ph, pw, (bf16_t*)out, pix_begin);
        else
            ATHENA_LAUNCH(fc_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, in, w_hwio, s0, s1, gamma,
                               beta, eps, infer, B, H, W, Cout, Ho, Wo, ph, pw, (float*)out, pix_begin);
    }
    ATHENA_CHECK_LAUNCH("athena_conv1_bn_relu6_fwd");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- backward
#define FC_BWD_BLOCKS 512
// partial[blk][t][c] (t < 9: sum a g; t == 9: sum g), fp32, one slab per workgroup
template <typename T, int CPT>
__global__ __launch_bounds__(256) void fc_bwd_reduce_kernel(const float* __restrict__ in, const T* __restrict__ g,
                                                            const T* __restrict__ act, int B, int H, int W, int C,
                                                            int Ho, int Wo, int ph, int pw,
                                                            float* __restrict__ partial) {
    extern __shared__ float red[];                   // [4 waves][(FC_K + 1)][C] (planes), else [(FC_K + 1)][C]
    constexpr int NP = CPT / 2;
    const int cpr = C / CPT;
    const int chunk = threadIdx.x % cpr, slot = threadIdx.x / cpr, slots = 256 / cpr;
    const int c0 = chunk * CPT;
    f32x2_t acc[FC_K + 1][NP];
#pragma unroll
    for (int t = 0; t <= FC_K; ++t)
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[t][i] = (f32x2_t){0.f, 0.f};
    const bool planes = cpr <= 64;                   // fold by shuffles + one LDS plane per wave
    if (!planes) {
        for (int i = threadIdx.x; i < (FC_K + 1) * C; i += 256) red[i] = 0.f;
        __syncthreads();
    }
    const int npix = B * Ho * Wo;
    // one pixel's operands as they come from memory: nine input values and CPT gradient /
    // activation elements (16-byte pieces)
    constexpr int NV = sizeof(T) == 2 ? CPT / 8 : CPT / 4;
    typedef typename std::conditional<sizeof(T) == 2, bf16x8_t, float4>::type Piece;
    struct Raw { float a[FC_K]; Piece g[NV], y[NV]; };
    auto fetch = [&](int pix, Raw& r) {
        const int q = pix / Wo, wo = pix - q * Wo;
        const int b = q / Ho, ho = q - b * Ho;
        fc_taps(in, H, W, b, ho, wo, ph, pw, r.a);
        const Piece* gp = (const Piece*)(g + (long)pix * C + c0);
#pragma unroll
        for (int j = 0; j < NV; ++j) r.g[j] = gp[j];
        if (act) {
            const Piece* yp = (const Piece*)(act + (long)pix * C + c0);
#pragma unroll
            for (int j = 0; j < NV; ++j) r.y[j] = yp[j];
        }
    };
    auto consume = [&](const Raw& r) {
        float gv[CPT];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int j = 0; j < NV; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = (float)r.g[j][i];
                    // g is the gradient w.r.t. the relu6 output: apply the gate read off the activation
                    if (act) v = ((float)r.y[j][i] > 0.f && (float)r.y[j][i] < 6.f) ? v : 0.f;
                    gv[8 * j + i] = v;
                }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const float h[4] = {r.g[j].x, r.g[j].y, r.g[j].z, r.g[j].w};
                const float y[4] = {r.y[j].x, r.y[j].y, r.y[j].z, r.y[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) gv[4 * j + i] = (!act || (y[i] > 0.f && y[i] < 6.f)) ? h[i] : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const f32x2_t g2 = (f32x2_t){gv[2 * i], gv[2 * i + 1]};
#pragma unroll
            for (int t = 0; t < FC_K; ++t)
                acc[t][i] = __builtin_elementwise_fma((f32x2_t){r.a[t], r.a[t]}, g2, acc[t][i]);
            acc[FC_K][i] += g2;
        }
    };
    // the operands of the next pixel are requested before the current one is accumulated: with
    // two waves per SIMD (the accumulators take most of the register file) a trip would otherwise
    // wait out a full memory latency
    {
        const int stride = gridDim.x * slots;
        int pix = blockIdx.x * slots + slot;
        Raw r0, r1;
        if (pix < npix) {
            fetch(pix, r0);
            while (true) {
                int nx = pix + stride;
                if (nx < npix) fetch(nx, r1);
                consume(r0);
                if (nx >= npix) break;
                pix = nx;
                nx = pix + stride;
                if (nx < npix) fetch(nx, r0);
                consume(r1);
                if (nx >= npix) break;
                pix = nx;
            }
        }
    }
    // fold the pixel slots of the workgroup, then store the slab
    float* dst = partial + (long)blockIdx.x * (FC_K + 1) * C;
    if (planes) {
        // lanes with equal chunk inside a wave by shuffles, then the four waves through LDS
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const int nent = (FC_K + 1) * C;
#pragma unroll
        for (int t = 0; t <= FC_K; ++t)
#pragma unroll
            for (int i = 0; i < CPT; ++i) {
                float v = acc[t][i / 2][i % 2];
                for (int o = cpr; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
                if (lane < cpr) red[w * nent + t * C + c0 + i] = v;
            }
        __syncthreads();
        for (int i = threadIdx.x; i < nent; i += 256)
            dst[i] = (red[i] + red[nent + i]) + (red[2 * nent + i] + red[3 * nent + i]);
        return;
    }
#pragma unroll
    for (int t = 0; t <= FC_K; ++t)
#pragma unroll
        for (int i = 0; i < CPT; ++i) atomicAdd(&red[t * C + c0 + i], acc[t][i / 2][i % 2]);
    __syncthreads();
    for (int i = threadIdx.x; i < (FC_K + 1) * C; i += 256) dst[i] = red[i];
}

// The same reduction on the matrix cores, reading the gradient alone.
//   * The relu6 gate is worked out again instead of read off the stored activation: the forward pass's
//     contraction (fc_fwd_mfma_kernel: channels x pixels from the split weights and taps, bit for bit
//     the value that was clamped and stored) gives the pre-clamp BatchNorm output d, and the gate is
//     [0 < d < 6] -- what [0 < stored < 6] says, since a stored value is never rounded across 0 or 6.
//   * The sums over pixels T[t][c] = sum_p a[p][t] g[p][c] are products too: a wave's 16 pixels are the
//     contraction of one 16x16x16 product per 16 channels, [channels x pixels] [pixels x (9 taps, 1)],
//     with the fp32 taps split into two bf16 halves (two products into one accumulator).  Both operands
//     have the pixel as the slow index where they are made (a lane has 8 channels of ONE pixel, and
//     that pixel's taps), so the wave lays them down in LDS and reads them back transposed.  Sixteen
//     accumulator registers per lane instead of 160, no multiply-adds on the vector ALUs.
// A workgroup takes up to 256 channels, so the taps of a pixel are fetched and split once for all of
// them.  Lane (li, kg) of a wave has pixel li of the wave's 16 and, per pair P, the eight channels
// cb + 32 P + 8 kg .. + 7: the layout of the gate's product and of a 16-byte piece of the gradient.
// Slab layout as fc_bwd_reduce_kernel: partial[blockIdx.x][t][c]; blockIdx.y picks the block of
// 32 * NPAIR channels.
typedef short fc_s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fc_s4 fc_read_tr(const bf16_t* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((fc_s4 __attribute__((address_space(3)))*)p);
}

template <int NPAIR, int NW>
__global__ __launch_bounds__(64 * NW) void fc_bwd_reduce_mfma_kernel(const float* __restrict__ in,
                                                                 const bf16_t* __restrict__ g,
                                                                 const float* __restrict__ w,
                                                                 const float* __restrict__ mean_rstd,
                                                                 const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, int B, int H, int W,
                                                                 int C, int Ho, int Wo, int ph, int pw,
                                                                 float* __restrict__ partial) {
    constexpr int CB = 32 * NPAIR;
    constexpr int GROW = CB + 8;                     // padded rows (elements) of the gradient tile
    constexpr int TROW = 24;                         // and of the two tap tiles (16 columns used)
    constexpr int RED_BYTES = NW * (FC_K + 1) * CB * 4, TILE_BYTES = NW * 16 * GROW * 2;
    __shared__ float sW[FC_K + 1][CB];
    __shared__ __attribute__((aligned(16))) bf16x8_t sF[NPAIR][2][64];       // the weight fragments, by lane
    // the waves' gradient tiles; after the loop the same bytes hold the waves' sums
    __shared__ __attribute__((aligned(16))) char sGR[RED_BYTES > TILE_BYTES ? RED_BYTES : TILE_BYTES];
    __shared__ __attribute__((aligned(16))) bf16_t sT[NW][2][16 * TROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kg = lane >> 4, li = lane & 15;
    const int cb = blockIdx.y * CB;
    {
        bf16x8_t wf[NPAIR][2];
        fc_weight_frags<NPAIR, 64 * NW>(w, mean_rstd, nullptr, gamma, beta, 0.f, 0, C, cb, sW, wf);
        if (wave == 0) {
#pragma unroll
            for (int P = 0; P < NPAIR; ++P) {
                sF[P][0][lane] = wf[P][0];
                sF[P][1][lane] = wf[P][1];
            }
        }
        __syncthreads();
    }
    const int npix = B * Ho * Wo;
    const int nblk = (npix + 16 * NW - 1) / (16 * NW);
    // acc[rb][r]: channel cb + 16 rb + 4 kg + r, column li (taps 0..8, then the plain sum)
    f32x4_t acc[2 * NPAIR];
#pragma unroll
    for (int rb = 0; rb < 2 * NPAIR; ++rb) acc[rb] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // A trip = the wave's 16 pixels x CB channels.  Its taps are requested a trip ahead (two register
    // sets); its NPAIR pieces of the gradient sit in gq, and each is requested again for the next trip
    // at the moment it is used, so a lane always has NPAIR 16-byte loads in flight.  A block beyond the
    // last one and a pixel beyond the last one re-read valid addresses and count for nothing.
    struct Trip { FcTaps t; float one; int pc; };
    // the lane's pixel walks through the image with a fixed stride: its coordinates (b, ho, wo) are
    // advanced by that stride's own coordinates with carries, no division in the loop
    const int step_pix = gridDim.x * 16 * NW;
    const int d_wo = step_pix % Wo, d_q = step_pix / Wo, d_ho = d_q % Ho, d_b = d_q / Ho;
    int c_pix = blockIdx.x * 16 * NW + wave * 16 + li;
    int c_wo = c_pix % Wo, c_ho = (c_pix / Wo) % Ho, c_b = c_pix / Wo / Ho;
    auto fetch_next = [&](Trip& r) {
        // (the walker's current position, then one step on)
        const bool live = c_pix < npix;
        r.pc = live ? c_pix : npix - 1;
        fc_fetch_taps_at(in, H, W, ph, pw, live ? c_b : B - 1, live ? c_ho : Ho - 1, live ? c_wo : Wo - 1, r.t);
        if (!live) r.t.ok = 0;
        r.one = live ? 1.f : 0.f;
        c_pix += step_pix;
        c_wo += d_wo;
        const int cw = c_wo >= Wo ? 1 : 0;
        c_wo -= cw ? Wo : 0;
        c_ho += d_ho + cw;
        const int ch = c_ho >= Ho ? 1 : 0;
        c_ho -= ch ? Ho : 0;
        c_b += d_b + ch;
    };
    bf16x8_t gq[NPAIR];
    bf16_t* myG = (bf16_t*)sGR + wave * 16 * GROW;
    bf16_t* myT0 = sT[wave][0];
    bf16_t* myT1 = sT[wave][1];
    // transposed reads: the 16 lanes of group kg take rows 4 kg .. 4 kg + 3, lane li column (block) li
    const int tr_row = 4 * kg + (li >> 2), tr_col = 4 * (li & 3);
    auto consume = [&](const Trip& r, const Trip& next) {
        const bf16x8_t pf = fc_tap_frag(r.t, kg);
        // the pixel's row of the two tap tiles: hi halves and the 1 of the plain sum | lo halves
        if (kg == 0) {
            bf16x8_t h0, h1, l0, l1;
            const bf16_t z = (bf16_t)0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bf16_t hi, lo;
                fc_split(((r.t.ok >> e) & 1u) ? r.t.v[e] : 0.f, hi, lo);
                h0[e] = hi;
                l0[e] = lo;
                h1[e] = z;
                l1[e] = z;
            }
            bf16_t hi8, lo8;
            fc_split(((r.t.ok >> 8) & 1u) ? r.t.v[8] : 0.f, hi8, lo8);
            h1[0] = hi8;
            l1[0] = lo8;
            h1[1] = (bf16_t)r.one;
            *(bf16x8_t*)(myT0 + li * TROW) = h0;
            *(bf16x8_t*)(myT0 + li * TROW + 8) = h1;
            *(bf16x8_t*)(myT1 + li * TROW) = l0;
            *(bf16x8_t*)(myT1 + li * TROW + 8) = l1;
        }
        const bf16_t* gnext = g + (long)next.pc * C + cb + 8 * kg;
#pragma unroll
        for (int P = 0; P < NPAIR; ++P) {
            const bf16x8_t gc = gq[P];
            gq[P] = *(const bf16x8_t*)(gnext + 32 * P);
            const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4_t d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sF[P][0][lane], pf, zero, 0, 0, 0);
            const f32x4_t d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sF[P][1][lane], pf, zero, 0, 0, 0);
            bf16x8_t gv;
            const bf16_t z = (bf16_t)0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                gv[e] = (d0[e] > 0.f && d0[e] < 6.f) ? gc[e] : z;
                gv[4 + e] = (d1[e] > 0.f && d1[e] < 6.f) ? gc[4 + e] : z;
            }
            *(bf16x8_t*)(myG + li * GROW + 32 * P + 8 * kg) = gv;
        }
        // (one wave: its LDS operations are carried out in order, the reads below see the writes above)
        const fc_s4 bh = fc_read_tr(myT0 + tr_row * TROW + tr_col);
        const fc_s4 bl = fc_read_tr(myT1 + tr_row * TROW + tr_col);
#pragma unroll
        for (int rb = 0; rb < 2 * NPAIR; ++rb) {
            const fc_s4 af = fc_read_tr(myG + tr_row * GROW + 16 * rb + tr_col);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, bh, acc[rb], 0, 0, 0);
            acc[rb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, bl, acc[rb], 0, 0, 0);
        }
    };
    {
        const int stride = gridDim.x;
        int blk = blockIdx.x;
        Trip cur, nxt;
        if (blk < nblk) {
            fetch_next(cur);
            const bf16_t* g0 = g + (long)cur.pc * C + cb + 8 * kg;
#pragma unroll
            for (int P = 0; P < NPAIR; ++P) gq[P] = *(const bf16x8_t*)(g0 + 32 * P);
            for (; blk < nblk; blk += stride) {
                // (the weight fragments are read from LDS every trip: kept in registers across trips
                // they would cost the second wave per SIMD)
                asm volatile("" ::: "memory");
                fetch_next(nxt);
                consume(cur, nxt);
                cur = nxt;
            }
        }
    }
    // the four waves through LDS (the products have summed over the pixels); the tiles are done with
    __syncthreads();
    float (*red)[FC_K + 1][CB] = (float (*)[FC_K + 1][CB])sGR;
    if (li <= FC_K) {
#pragma unroll
        for (int rb = 0; rb < 2 * NPAIR; ++rb)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][li][16 * rb + 4 * kg + e] = acc[rb][e];
    }
    __syncthreads();
    float* dst = partial + (long)blockIdx.x * (FC_K + 1) * C;
    for (int i = tid; i < (FC_K + 1) * CB; i += 64 * NW) {
        const int t = i / CB, c = i - t * CB;
        float v = (red[0][t][c] + red[1][t][c]) + (red[2][t][c] + red[3][t][c]);
        if (NW == 8) v += (red[4][t][c] + red[5][t][c]) + (red[6][t][c] + red[7][t][c]);
        dst[t * C + cb + c] = v;
    }
}

// sums the slabs: workgroup = 64 entries (t, c) x 4 waves, each wave takes a quarter of the slabs
__global__ __launch_bounds__(256) void fc_bwd_fold_kernel(const float* __restrict__ partial, int nblk, int nent,
                                                          double* __restrict__ tot) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    double a0 = 0.0, a1 = 0.0;
    if (e < nent) {
        int b = w;
        for (; b + 4 < nblk; b += 8) {
            a0 += (double)partial[(long)b * nent + e];
            a1 += (double)partial[(long)(b + 4) * nent + e];
        }
        if (b < nblk) a0 += (double)partial[(long)b * nent + e];
    }
    red[w][lane] = a0 + a1;
    __syncthreads();
    if (w == 0 && e < nent) tot[e] = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

__global__ void fc_bwd_final_kernel(const double* __restrict__ tot, const double* __restrict__ mom,
                                    const float* __restrict__ w, const float* __restrict__ mean_rstd,
                                    const float* __restrict__ gamma, long npix, int C, float* __restrict__ dw,
                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double T[FC_K + 1];
#pragma unroll
    for (int t = 0; t <= FC_K; ++t) T[t] = tot[t * C + c];
    const double sb = T[FC_K];
    const double mu = (double)mean_rstd[c], rs = (double)mean_rstd[C + c], ga = (double)gamma[c];
    const double n = (double)npix;
    double wc[FC_K];
#pragma unroll
    for (int t = 0; t < FC_K; ++t) wc[t] = (double)w[t * C + c];
    double gx = 0.0;                                  // sum_p g x
#pragma unroll
    for (int t = 0; t < FC_K; ++t) gx += wc[t] * T[t];
    const double sg = rs * (gx - mu * sb);            // sum_p g xhat
    dbeta[c] = (float)sb;
    dgamma[c] = (float)sg;
#pragma unroll
    for (int t = 0; t < FC_K; ++t) {
        double rw = 0.0;                              // sum_p a[p][t] x[p][c]
#pragma unroll
        for (int u = 0; u < FC_K; ++u) rw += mom[FC_K + t * FC_K + u] * wc[u];
        const double axh = rs * (rw - mu * mom[t]);   // sum_p a[p][t] xhat[p][c]
        dw[t * C + c] = (float)(ga * rs * (T[t] - sb / n * mom[t] - sg / n * axh));
    }
}

extern "C" int64_t athena_conv1_bn_bwd_workspace_bytes(int Cout) {
    const int64_t nent = (int64_t)(FC_K + 1) * (Cout > 0 ? Cout : 0);
    return (int64_t)sizeof(float) * FC_BWD_BLOCKS * nent + (int64_t)sizeof(double) * nent;   // slabs + totals
}

extern "C" int athena_conv1_bn_bwd(const float* in, const void* g, const void* relu6_out, const float* w_hwio,
                                   const double* moments,
                                   const float* mean_rstd, const float* gamma, const float* beta, int B, int H, int W,
                                   int Cout, int dtype, void* workspace, int64_t workspace_bytes, float* dw_hwio,
                                   float* dgamma, float* dbeta, athena_stream_t stream) {
    int rc = fc_check("athena_conv1_bn_bwd", B, H, W, Cout, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(in && g && w_hwio && moments && mean_rstd && gamma && workspace && dw_hwio && dgamma && dbeta,
                   ATHENA_E_SHAPE, "athena_conv1_bn_bwd: null operand");
    ATHENA_REQUIRE(athena_aligned16(g) && athena_aligned16(workspace), ATHENA_E_ALIGN,
                   "athena_conv1_bn_bwd: g and workspace must be 16-byte aligned");
    ATHENA_REQUIRE(workspace_bytes >= athena_conv1_bn_bwd_workspace_bytes(Cout), ATHENA_E_SHAPE,
                   "athena_conv1_bn_bwd: workspace too small (%lld bytes)", (long long)workspace_bytes);
    int Ho, Wo, ph, pw;
    fc_same_pad(H, &Ho, &ph);
    fc_same_pad(W, &Wo, &pw);
    const long npix = (long)B * Ho * Wo;
    const int cpt = fc_cpt(Cout);
    const int slots = 256 / (Cout / cpt);
    long blocks = (npix + slots - 1) / slots;
    if (blocks > FC_BWD_BLOCKS) blocks = FC_BWD_BLOCKS;
    hipStream_t s = (hipStream_t)stream;
    const size_t shm = sizeof(float) * (FC_K + 1) * Cout * (Cout / cpt <= 64 ? 4 : 1);
    ATHENA_REQUIRE(!relu6_out || athena_aligned16(relu6_out), ATHENA_E_ALIGN, "athena_conv1_bn_bwd: relu6_out alignment");
    static const int fc_gate = getenv("ATHENA_FC_GATE") ? atoi(getenv("ATHENA_FC_GATE")) : 8;
    static const int fc_wgs = getenv("ATHENA_FC_WGS") ? atoi(getenv("ATHENA_FC_WGS")) : 256;
    if (relu6_out && beta && dtype == ATHENA_BF16 && Cout % 64 == 0 && (Cout <= 256 || Cout % 256 == 0) && fc_gate > 0) {
        // the gate worked out again instead of read: up to 256 channels per workgroup, eight waves of
        // 16 pixels per trip (one workgroup per CU and slab; four waves for narrow layers)
        const int cbw = Cout < 256 ? Cout : 256, ncb = Cout / cbw;
        const int nw = (cbw == 256 && fc_gate == 8) ? 8 : 4;
        long nb = (npix + 16 * nw - 1) / (16 * nw);
        blocks = std::max(1L, std::min(std::min(nb, (long)FC_BWD_BLOCKS), (long)(fc_wgs * (nw == 8 ? 1 : 2) / ncb)));
        const dim3 grid((unsigned)blocks, (unsigned)ncb);
#define FC_BWD_MFMA(NP, NWV)                                                                                   \
    ATHENA_LAUNCH((fc_bwd_reduce_mfma_kernel<NP, NWV>), grid, dim3(64 * NWV), 0, s, in, (const bf16_t*)g, w_hwio, \
                  mean_rstd, gamma, beta, B, H, W, Cout, Ho, Wo, ph, pw, (float*)workspace)
        switch (cbw / 32) {
            case 2: FC_BWD_MFMA(2, 4); break;
            case 4: FC_BWD_MFMA(4, 4); break;
            case 6: FC_BWD_MFMA(6, 4); break;
            default:
                if (nw == 8) FC_BWD_MFMA(8, 8);
                else FC_BWD_MFMA(8, 4);
                break;
        }
#undef FC_BWD_MFMA
    } else {
#define FC_BWD(TT, CC)                                                                                          \
    ATHENA_LAUNCH((fc_bwd_reduce_kernel<TT, CC>), dim3((unsigned)blocks), dim3(256), shm, s, in, (const TT*)g, \
                       (const TT*)relu6_out, B, H, W, Cout, Ho, Wo, ph, pw, (float*)workspace)
    if (dtype == ATHENA_BF16) {
        if (cpt == 16) FC_BWD(bf16_t, 16);
        else FC_BWD(bf16_t, 8);
    } else {
        if (cpt == 16) FC_BWD(float, 16);
        else FC_BWD(float, 8);
    }
#undef FC_BWD
    }
    const int nent = (FC_K + 1) * Cout;
    double* tot = (double*)((char*)workspace + sizeof(float) * (size_t)FC_BWD_BLOCKS * nent);   // 8-byte aligned
    ATHENA_LAUNCH(fc_bwd_fold_kernel, dim3((nent + 63) / 64), dim3(256), 0, s, (const float*)workspace, (int)blocks,
                       nent, tot);
    ATHENA_LAUNCH(fc_bwd_final_kernel, dim3((Cout + 63) / 64), dim3(64), 0, s, (const double*)tot, moments, w_hwio,
                       mean_rstd, gamma, npix, Cout, dw_hwio, dgamma, dbeta);
    ATHENA_CHECK_LAUNCH("athena_conv1_bn_bwd");
    return ATHENA_OK;
}
