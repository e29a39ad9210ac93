// Everything of a post-LN transformer layer that follows the attention kernel, as ONE launch per
// direction, for d_model = 256 in bf16:
//
//   forward   u  = ctx Wo + bo                      z1 = x + dropout(u)     y1 = LayerNorm1(z1)
//             hp = y1 W1 + b1    h = dropout(gelu(hp))
//             f  = h W2 + b2                        z2 = y1 + dropout(f)    y2 = LayerNorm2(z2)
//   backward  the same chain in reverse: LayerNorm2', W2^T, gelu', W1^T, LayerNorm1', Wo^T
//
// (reference athena/layers/transformer.py:296-310 and :374-388; athena/layers/attention.py:166-169
// for the output projection).  With dff = 0 only the first line runs: an attention sub-layer that
// is not followed by the feed-forward block (the decoder's self-attention).
//
// A workgroup (8 waves) owns 32 complete rows.  The row block stays on the CU from the attention
// context to y2: every intermediate is a [32, 256] image in LDS (the hidden layer goes through in
// 256-wide slices that are consumed by the second product as soon as they exist), LayerNorm
// statistics are reduced across the eight waves through LDS, and only what the backward pass reads
// again is written to memory.  The operands that are NOT row-local are the weights: wave w owns
// output columns 32 w .. 32 w + 31 of every product, so each weight element is needed by exactly one
// wave of the workgroup -- it is loaded from L2 straight into that wave's registers in MFMA fragment
// layout (no LDS staging, no barrier in the contraction loop).  A "unit" is the [32 columns, 256 k]
// weight block of one wave for one slice: 16 fragments = 64 registers.  Two units are live (the two
// products of a hidden-layer slice); as soon as a k-block of a unit has been multiplied, its two
// registers are reloaded with the same k-block of the unit that takes its place one slice later,
// so every fragment is requested a whole slice ahead of its use and 16-32 KB per wave are in flight
// throughout.  Loads return in order, so waiting for ANY later load would wait for all of these:
// the kernels therefore issue no other global load while fragments are in flight except through
// the same one-slice-ahead discipline -- vectors (biases, gamma, beta, row statistics) and row
// blocks needed late are staged in LDS by the prologue.
// Activations never move between registers and global memory directly: in accumulator layout a lane
// owns 4 consecutive columns of 2 x 2 tiles, i.e. 8-byte pieces 16 rows apart, which reach memory as
// 32-byte fragments of cache lines.  Every row block goes through an LDS image instead and is read
// or written with 16 bytes per lane along rows (a wave = two full 512-byte rows).
//
// Bounds: each CU pulls all weights of the layer once (1.15 MB at dff = 1024); all workgroups read
// the same weights, so that is 290 MB out of L2 per launch.  The 72 k-blocks of 4 MFMAs per wave
// take 3.8 us of matrix time.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdlib>
#include "common.h"

#include "layer_tail_frag.h"

// Two sums per row at once: the lane's own values are already added up in a[i], b[i]; the four lanes
// that share a row inside a wave fold by shuffles, the eight waves through LDS (one barrier; the
// caller separates two uses of ``red`` ([2][8][32] floats) by another barrier).
template <int RB>
__device__ __forceinline__ void lt_row_sums2(float (&a)[RB], float (&b)[RB], float* red, int wave, int lane) {
    constexpr int R = 16 * RB;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        a[i] = rows4_sum(a[i]);
        b[i] = rows4_sum(b[i]);
        if (lane < 16) {
            red[wave * R + 16 * i + lane] = a[i];
            red[8 * R + wave * R + 16 * i + lane] = b[i];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float t = 0.f, u = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            t += red[w * R + 16 * i + (lane & 15)];
            u += red[8 * R + w * R + 16 * i + (lane & 15)];
        }
        a[i] = t;
        b[i] = u;
    }
}

// z = res + dropout(acc + bias), y = LayerNorm(z).  v comes in as the accumulators and leaves as y
// (rounded to bf16, which is what the next sub-layer and the backward pass see).  z and y go into
// the LDS images zimg / yimg (the caller copies them out after its next barrier); mean / rstd go
// straight to memory.  bias, gamma, beta: LDS.  red: two arrays of [8][32] floats.  One barrier inside,
// between writing z and writing y.
template <int RB>
__device__ __forceinline__ void lt_add_ln(float (&v)[RB][2][4], const float (&res)[RB][2][4], const float* bias,
                                          const float* gamma, const float* beta, float eps, uint32_t thresh,
                                          float dscale, uint64_t key, int row0, int rows, int wave, int lane,
                                          float* __restrict__ mean, float* __restrict__ rstd, float* red, bf16_t* zimg,
                                          bf16_t* yimg) {
    constexpr int R = 16 * RB;
    const int fr = lane & 15, fq = lane >> 4;
    float s[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) s[i] = 0.f;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = row0 + 16 * i + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 32 * wave + 16 * j + 4 * fq;
            const float4 b4 = *(const float4*)(bias + n);
            const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
            float dm[4] = {1.f, 1.f, 1.f, 1.f};
            if (thresh) lt_dropout4(key, (uint64_t)row * LT_D + n, thresh, dscale, dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[i][j][r] = res[i][j][r] + (v[i][j][r] + bb[r]) * dm[r];
                s[i] += v[i][j][r];
            }
            lt_store4(zimg + lt_img_off<RB>(16 * i + fr, n), v[i][j]);
        }
    }
    // statistics of the wave's own 32 columns (mean, sum of squared deviations from it) by shuffles,
    // then the eight pairs of a row are merged behind ONE barrier: the sum of squares about the row
    // mean is sum_w [M2_w + 32 (mean_w - mean)^2] -- no cancellation, no second pass over the block
    float mu[RB], q[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        s[i] = rows4_sum(s[i]);
        const float mw = s[i] * (1.f / 32.f);
        float m2 = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dv = v[i][j][r] - mw;
                m2 += dv * dv;
            }
        m2 = rows4_sum(m2);
        if (lane < 16) {
            red[wave * R + 16 * i + lane] = mw;
            red[8 * R + wave * R + 16 * i + lane] = m2;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        float mws[8], t = 0.f, u = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            mws[w] = red[w * R + 16 * i + fr];
            t += mws[w];
            u += red[8 * R + w * R + 16 * i + fr];
        }
        mu[i] = t * 0.125f;
#pragma unroll
        for (int w = 0; w < 8; ++w) u += 32.f * (mws[w] - mu[i]) * (mws[w] - mu[i]);
        q[i] = u;
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = row0 + 16 * i + fr;
        const float rs = rsqrtf(q[i] * (1.f / LT_D) + eps);
        if (wave == 0 && fq == 0 && row < rows) {
            mean[row] = mu[i];
            rstd[row] = rs;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 32 * wave + 16 * j + 4 * fq;
            const float4 g4 = *(const float4*)(gamma + n);
            const float4 t4 = *(const float4*)(beta + n);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w}, tt[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i][j][r] = (v[i][j][r] - mu[i]) * rs * gg[r] + tt[r];
            lt_store4(yimg + lt_img_off<RB>(16 * i + fr, n), v[i][j]);
        }
    }
}

#ifdef LT_STAMPS
__device__ unsigned long long lt_stamps[64];
#define LT_STAMP(i) do { if (blockIdx.x == 7 && threadIdx.x == 0) lt_stamps[i] = __builtin_readcyclecounter(); } while (0)
extern "C" int athena_layer_tail_debug_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lt_stamps), sizeof(lt_stamps));
}
#else
#define LT_STAMP(i) do { } while (0)
#endif

#define LT_MAX_DFF 2048                  // the first Dense's bias is staged in LDS

struct LtFwd {
    int rows, dff;
    const bf16_t* ctx;
    const bf16_t* x;
    const bf16_t* wo;       // tiled, rows = 256 outputs
    const float* bo;
    const float* gamma1;
    const float* beta1;
    const bf16_t* w1;       // tiled, rows = dff outputs
    const float* b1;
    const bf16_t* w2;       // tiled, rows = 256 outputs, dff contraction indices
    const float* b2;
    const float* gamma2;
    const float* beta2;
    float eps, dscale;
    uint32_t thresh;
    const uint64_t* rng_state;
    uint32_t site_res1, site_ffn, site_res2;
    bf16_t* z1;
    bf16_t* y1;
    float* mean1;
    float* rstd1;
    bf16_t* hpre;
    bf16_t* h;
    bf16_t* z2;
    bf16_t* y2;
    float* mean2;
    float* rstd2;
    // optional (QKV): three more products on the layer's output, y2 Wn + bn for the q / k / v
    // projections of the NEXT layer's self-attention (tiled copies with rows = outputs), while the rows
    // are still on the CU
    const bf16_t* wn[3];
    const float* bn[3];
    bf16_t* on[3];
    // optional: the residual input once more in fp32 (read instead of x for the residual sum; x stays the
    // operand nothing here multiplies), and an fp32 copy of the launch's final LayerNorm output for
    // the launch that takes it as ITS residual input -- the stream then passes from layer to layer
    // unrounded, and only the copies that products read are bf16
    const float* x32;
    float* y32;
};

template <bool FFN, bool QKV, int RB>
__global__ __launch_bounds__(LT_THREADS) void layer_tail_fwd_kernel(const LtFwd p) {
    constexpr int R = 16 * RB;
    // extra products on the launch's output: three behind a feed-forward part (q, k, v of the next
    // layer's self-attention), one without it (the q of the cross-attention that follows a decoder's
    // self-attention sub-layer)
    // images: [0] ctx, then hidden slices (even)  [1] hidden slices (odd)  [2] y1
    //         [3], [4] staging: x, pre-activation slices (even / odd), z1, z2, y2
    __shared__ __attribute__((aligned(16))) bf16_t imgs[FFN ? 5 : 4][LT_IMG(RB)];
    __shared__ __attribute__((aligned(16))) float vecs[QKV ? 9 : 6][LT_D];   // bo gamma1 beta1 b2 gamma2 beta2 (bq bk bv)
    __shared__ __attribute__((aligned(16))) float b1s[FFN ? LT_MAX_DFF : 4];
    __shared__ float red[2 * 8 * R];
    bf16_t* const img_a = imgs[0];
    bf16_t* const img_h = imgs[1];
    bf16_t* const img_y = imgs[2];
    bf16_t* const st0 = imgs[3];
    bf16_t* const st1 = imgs[FFN ? 4 : 0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int row0 = blockIdx.x * R;
    const int nslice = FFN ? p.dff / LT_D : 0;
    const int n0 = 32 * wave;

    LT_STAMP(0);
    // ---- everything small first: these are the last loads that are waited for as a group
    LtTile<RB> tc, tx;
    lt_tile_load<RB>(tc, p.ctx, LT_D, 0, row0, p.rows, tid);
    // the residual input: its fp32 copy straight into the accumulator layout if there is one, else the
    // bf16 tile through LDS
    const bool wide = p.x32 != nullptr;
    float res[RB][2][4];
    if (wide) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = min(row0 + 16 * i + fr, p.rows - 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float4 t = *(const float4*)(p.x32 + (long)row * LT_D + n0 + 16 * j + 4 * fq);
                res[i][j][0] = t.x; res[i][j][1] = t.y; res[i][j][2] = t.z; res[i][j][3] = t.w;
            }
        }
    } else {
        lt_tile_load<RB>(tx, p.x, LT_D, 0, row0, p.rows, tid);
    }
    const bool hi = tid >= LT_D;
    const int ve = tid & (LT_D - 1);
    float vv[3], bv[LT_MAX_DFF / LT_THREADS];
    vv[0] = (hi ? p.gamma1 : p.bo)[ve];
    vv[1] = (hi && !FFN) ? 0.f : (hi ? p.b2 : p.beta1)[ve];
    vv[2] = FFN ? (hi ? p.beta2 : p.gamma2)[ve] : 0.f;
#pragma unroll
    for (int t = 0; t < LT_MAX_DFF / LT_THREADS; ++t) {
        const int e = tid + t * LT_THREADS;
        bv[t] = (FFN && e < p.dff) ? p.b1[e] : 0.f;
    }
    float nb[2] = {0.f, 0.f};
    if constexpr (QKV && FFN) {
        nb[0] = (hi ? p.bn[1] : p.bn[0])[ve];
        if (!hi) nb[1] = p.bn[2][ve];
    } else if constexpr (QKV) {
        if (!hi) nb[0] = p.bn[0][ve];
    }
    // ---- weights: the projection, and slice 0 of the first Dense
    LtUnit ua, ub;
    lt_load_unit(ua, lt_unit_ptr(p.wo, LT_D, n0, 0, lane), LT_D);
    if (FFN) lt_load_unit(ub, lt_unit_ptr(p.w1, LT_D, n0, 0, lane), LT_D);
    __builtin_amdgcn_sched_barrier(0);
    lt_tile_to_img<RB>(img_a, tc, tid);
    if (!wide) lt_tile_to_img<RB>(st0, tx, tid);
    vecs[hi][ve] = vv[0];
    vecs[2 + hi][ve] = vv[1];
    vecs[4 + hi][ve] = vv[2];
    if (FFN) {
#pragma unroll
        for (int t = 0; t < LT_MAX_DFF / LT_THREADS; ++t) b1s[tid + t * LT_THREADS] = bv[t];
    }
    if constexpr (QKV && FFN) {
        vecs[6 + hi][ve] = nb[0];
        if (!hi) vecs[8][ve] = nb[1];
    } else if constexpr (QKV) {
        if (!hi) vecs[6][ve] = nb[0];
    }
    __syncthreads();
    LT_STAMP(1);

    if (!wide) {                                                 // the residual input, in accumulator layout
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bf16x4_t t = lt_img_get4<RB>(st0, 16 * i + fr, n0 + 16 * j + 4 * fq);
#pragma unroll
                for (int r = 0; r < 4; ++r) res[i][j][r] = (float)t[r];
            }
    }
    f32x4_t acc[RB][2];
    float v[RB][2][4];
    lt_zero<RB>(acc);
    // ctx Wo; the projection's registers are refilled with slice 0 of the second Dense (without a
    // feed-forward part: with the unit of the one extra product, if there is one)
    lt_mul_unit<FFN || QKV, RB>(acc, ua, img_a, lane,
                            FFN ? lt_unit_ptr(p.w2, p.dff, n0, 0, lane) : lt_unit_ptr(p.wn[0], LT_D, n0, 0, lane),
                            FFN ? (long)p.dff : (long)LT_D);
    LT_STAMP(2);
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i][j][r] = acc[i][j][r];
    uint64_t key = 0;
    if (p.thresh) key = rng_key(p.rng_state, p.site_res1);
    // z1 -> st1 (st0 may still be read for x by slower waves), y1 -> img_y
    lt_add_ln<RB>(v, res, vecs[0], vecs[1], vecs[2], p.eps, p.thresh, p.dscale, key, row0, p.rows, wave, lane, p.mean1,
              p.rstd1, red, FFN ? st1 : st0, img_y);
    if (!FFN && p.y32) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = row0 + 16 * i + fr;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (row < p.rows)
                    *(float4*)(p.y32 + (long)row * LT_D + n0 + 16 * j + 4 * fq) =
                        make_float4(v[i][j][0], v[i][j][1], v[i][j][2], v[i][j][3]);
        }
    }
    __syncthreads();                                         // z1 and y1 images complete; img_a free again
    LT_STAMP(3);
    lt_img_to_global<RB>(FFN ? st1 : st0, p.z1, LT_D, 0, row0, p.rows, tid);
    lt_img_to_global<RB>(img_y, p.y1, LT_D, 0, row0, p.rows, tid);
    LT_STAMP(4);
    if constexpr (!FFN && QKV) {
        // y1 Wn + bn into the image the attention context came in (read for the last time before the
        // barrier above)
        lt_zero<RB>(acc);
        lt_mul_unit<false, RB>(acc, ua, img_y, lane, nullptr, 0);
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + 16 * j + 4 * fq;
                const float4 b4 = *(const float4*)(&vecs[6][n]);
                const float o[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
                lt_store4(img_a + lt_img_off<RB>(16 * i + fr, n), o);
            }
        __syncthreads();
        lt_img_to_global<RB>(img_a, p.on[0], LT_D, 0, row0, p.rows, tid);
    }
    if (!FFN) return;

    // v now holds y1 (the residual of the second half)
    f32x4_t facc[RB][2];
    lt_zero<RB>(facc);
    uint64_t hkey = 0;
    if (p.thresh) hkey = rng_key(p.rng_state, p.site_ffn);
    auto slice = [&](int c, auto more_tag) {
        constexpr bool MORE = decltype(more_tag)::value;
        bf16_t* himg = (c & 1) ? img_h : img_a;
        bf16_t* pimg = (c & 1) ? st1 : st0;
        lt_zero<RB>(acc);
        // (the last slice requests the next layer's q and k units in place of a next slice's)
        lt_mul_unit<MORE || QKV, RB>(acc, ub, img_y, lane,
                                 MORE ? lt_unit_ptr(p.w1, LT_D, (c + 1) * LT_D + n0, 0, lane)
                                      : lt_unit_ptr(p.wn[0], LT_D, n0, 0, lane), LT_D);          // y1 W1[slice c]
        LT_STAMP(8 + 5 * c);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = row0 + 16 * i + fr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int nl = n0 + 16 * j + 4 * fq;                              // column inside the slice
                const int n = c * LT_D + nl;
                const float4 b4 = *(const float4*)(b1s + n);
                const float bb[4] = {b4.x, b4.y, b4.z, b4.w};
                float pre[4], hv[4], dm[4] = {1.f, 1.f, 1.f, 1.f};
                if (p.thresh) lt_dropout4(hkey, (uint64_t)row * p.dff + n, p.thresh, p.dscale, dm);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pre[r] = acc[i][j][r] + bb[r];
                    hv[r] = gelu_tanh_f(pre[r]) * dm[r];
                }
                lt_store4(pimg + lt_img_off<RB>(16 * i + fr, nl), pre);
                lt_store4(himg + lt_img_off<RB>(16 * i + fr, nl), hv);
            }
        }
        LT_STAMP(9 + 5 * c);
        __syncthreads();                                     // slice c of the hidden layer is in LDS
        LT_STAMP(10 + 5 * c);
        lt_img_to_global<RB>(pimg, p.hpre, p.dff, c * LT_D, row0, p.rows, tid);
        lt_img_to_global<RB>(himg, p.h, p.dff, c * LT_D, row0, p.rows, tid);
        LT_STAMP(11 + 5 * c);
        lt_mul_unit<MORE || QKV, RB>(facc, ua, himg, lane,
                                 MORE ? lt_unit_ptr(p.w2, p.dff, n0, (c + 1) * LT_D, lane)
                                      : lt_unit_ptr(p.wn[1], LT_D, n0, 0, lane),
                                 MORE ? (long)p.dff : (long)LT_D);                              // += h[slice c] W2[slice c, :]
        LT_STAMP(12 + 5 * c);
    };
    for (int c = 0; c + 1 < nslice; ++c) slice(c, std::true_type{});
    slice(nslice - 1, std::false_type{});
    float f[RB][2][4];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) f[i][j][r] = facc[i][j][r];
    if (p.thresh) key = rng_key(p.rng_state, p.site_res2);
    // the images of the parity the last slice did not use are free: no wave reads them any more
    bf16_t* zimg = (nslice & 1) ? st1 : st0;
    bf16_t* yimg = (nslice & 1) ? img_h : img_a;
    lt_add_ln<RB>(f, v, vecs[3], vecs[4], vecs[5], p.eps, p.thresh, p.dscale, key, row0, p.rows, wave, lane, p.mean2,
              p.rstd2, red, zimg, yimg);
    LT_STAMP(5);
    if (p.y32) {
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = row0 + 16 * i + fr;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                if (row < p.rows)
                    *(float4*)(p.y32 + (long)row * LT_D + n0 + 16 * j + 4 * fq) =
                        make_float4(f[i][j][0], f[i][j][1], f[i][j][2], f[i][j][3]);
        }
    }
    __syncthreads();
    lt_img_to_global<RB>(zimg, p.z2, LT_D, 0, row0, p.rows, tid);
    lt_img_to_global<RB>(yimg, p.y2, LT_D, 0, row0, p.rows, tid);
    LT_STAMP(6);
    if constexpr (QKV && FFN) {
        // ub holds the q unit, ua the k unit; the v unit takes ub's place as that is multiplied.  The
        // results go into the three images nobody reads any more (the hidden-layer images of the
        // other parity and y1's) and leave together behind one barrier.
        bf16_t* const oimg[3] = {(nslice & 1) ? img_a : img_h, (nslice & 1) ? st0 : st1, img_y};
        auto project = [&](int m, LtUnit& u, auto roll_tag) {
            lt_zero<RB>(acc);
            lt_mul_unit<decltype(roll_tag)::value, RB>(acc, u, yimg, lane, lt_unit_ptr(p.wn[2], LT_D, n0, 0, lane), LT_D);
#pragma unroll
            for (int i = 0; i < RB; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int n = n0 + 16 * j + 4 * fq;
                    const float4 b4 = *(const float4*)(&vecs[6 + m][n]);
                    const float o[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
                    lt_store4(oimg[m] + lt_img_off<RB>(16 * i + fr, n), o);
                }
        };
        project(0, ub, std::true_type{});
        LT_STAMP(28);
        project(1, ua, std::false_type{});
        LT_STAMP(29);
        project(2, ub, std::false_type{});
        LT_STAMP(30);
        __syncthreads();
        LT_STAMP(31);
#pragma unroll
        for (int m = 0; m < 3; ++m) lt_img_to_global<RB>(oimg[m], p.on[m], LT_D, 0, row0, p.rows, tid);
        LT_STAMP(32);
    }
}

// ------------------------------------------------------------------------------------- backward
// dz = LayerNorm'(dy) for the rows / columns this lane owns, and this workgroup's partial column
// sums for dgamma / dbeta into ``slab`` ([2][256], one slab per workgroup: every column belongs to
// one wave, so the slab is written without atomics).  dy comes in as g, dz leaves in g; zv holds
// the LayerNorm input, mean / rstd (indexed by the row inside the block) and gamma live in LDS.
template <int RB>
__device__ __forceinline__ void lt_ln_bwd(float (&g)[RB][2][4], const bf16x4_t (&zv)[RB][2], const float* mean,
                                          const float* rstd, const float* gamma, int row0, int rows, int wave, int lane,
                                          float* red, float* __restrict__ slab) {
    const int fr = lane & 15, fq = lane >> 4;
    float xh[RB][2][4], rs[RB], c1[RB], c2[RB];
#pragma unroll
    for (int i = 0; i < RB; ++i) { c1[i] = 0.f; c2[i] = 0.f; }
    float pg[2][4], pb[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { pg[j][r] = 0.f; pb[j][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const bool live = row0 + 16 * i + fr < rows;
        const float mu = mean[16 * i + fr];
        rs[i] = rstd[16 * i + fr];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = 32 * wave + 16 * j + 4 * fq;
            const float4 g4 = *(const float4*)(gamma + n);
            const float gg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = live ? g[i][j][r] : 0.f;
                xh[i][j][r] = live ? ((float)zv[i][j][r] - mu) * rs[i] : 0.f;
                pg[j][r] += d * xh[i][j][r];
                pb[j][r] += d;
                g[i][j][r] = d * gg[r];
                c1[i] += g[i][j][r];
                c2[i] += g[i][j][r] * xh[i][j][r];
            }
        }
    }
    // column sums over the workgroup's rows: lanes with equal (lane >> 4) hold the same columns
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = pg[j][r], b = pb[j][r];
            a = row16_sum(a);
            b = row16_sum(b);
            if (fr == 0) {
                const int n = 32 * wave + 16 * j + 4 * fq + r;
                slab[n] = a;
                slab[LT_D + n] = b;
            }
        }
    lt_row_sums2<RB>(c1, c2, red, wave, lane);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const float m1 = c1[i] * (1.f / LT_D), m2 = c2[i] * (1.f / LT_D);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) g[i][j][r] = rs[i] * (g[i][j][r] - m1 - xh[i][j][r] * m2);
    }
}

struct LtBwd {
    int rows, dff;
    const bf16_t* dy;        // gradient of the layer output (of y1 when dff == 0)
    const bf16_t* z2;
    const float* mean2;
    const float* rstd2;
    const float* gamma2;
    const bf16_t* hpre;
    const bf16_t* w2;        // second Dense, tiled, rows = dff inputs
    const bf16_t* w1;        // first Dense, tiled, rows = 256 inputs, dff contraction indices
    const bf16_t* z1;
    const float* mean1;
    const float* rstd1;
    const float* gamma1;
    const bf16_t* wo;        // output projection, tiled, rows = inputs
    float dscale;
    uint32_t thresh;
    const uint64_t* rng_state;
    uint32_t site_res1, site_ffn, site_res2;
    bf16_t* df;              // gradient of the second Dense's output (after the dropout mask)
    bf16_t* dh;              // gradient of the first Dense's pre-activation
    bf16_t* da;              // gradient of the projection's output (after the dropout mask)
    bf16_t* dx;              // gradient of the residual input
    bf16_t* dctx;            // gradient of the attention context
    float* slab2;            // [workgroups][2][256]
    float* slab1;
    // optional (ADD): dy is short of the product add[rows, 768] (aw[0] | aw[1] | aw[2])^T -- the part of
    // the gradient of this layer's output that comes back through the q / k / v projections of the
    // layer behind it (add = dq | dk | dv side by side; aw: tiled copies with rows = the projections'
    // 256 inputs) -- and the kernel adds it before anything else
    const bf16_t* add;
    const bf16_t* aw[3];
};

template <bool FFN, bool ADD, int RB>
__global__ __launch_bounds__(LT_THREADS) void layer_tail_bwd_kernel(const LtBwd p) {
    constexpr int R = 16 * RB;
    // products added to dy: dq | dk | dv of a self-attention behind a layer with a feed-forward part,
    // the dq of the cross-attention behind a decoder's self-attention sub-layer
    constexpr int NA = FFN ? 3 : 1;
    // images: [0] df, later da  [1] z1  [2], [3] dh slices (even / odd); before the slices dy and z2,
    //         after them dx and dctx  [4], [5] pre-activation slices (even / odd)
    __shared__ __attribute__((aligned(16))) bf16_t imgs[FFN ? 6 : 4][LT_IMG(RB)];
    __shared__ __attribute__((aligned(16))) float gam[2][LT_D];          // gamma2, gamma1
    __shared__ float stat[4][R];                                      // mean2 rstd2 mean1 rstd1
    __shared__ float red[2 * 8 * R];
    __shared__ f32x4_t park[FFN ? 4 : 1][LT_THREADS];                    // dz2 of each thread while the slices run
    bf16_t* const img_a = imgs[0];
    bf16_t* const img_z1 = imgs[1];
    bf16_t* const img_h0 = imgs[2];
    bf16_t* const img_h1 = imgs[3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int row0 = blockIdx.x * R;
    const int nslice = FFN ? p.dff / LT_D : 0;
    const int n0 = 32 * wave;

    // ---- everything small first (see the forward kernel)
    LtTile<RB> tz1, tdy, tz2, thp;
    lt_tile_load<RB>(tz1, p.z1, LT_D, 0, row0, p.rows, tid);
    lt_tile_load<RB>(tdy, p.dy, LT_D, 0, row0, p.rows, tid);
    if (FFN) {
        lt_tile_load<RB>(tz2, p.z2, LT_D, 0, row0, p.rows, tid);
        lt_tile_load<RB>(thp, p.hpre, p.dff, 0, row0, p.rows, tid);
    }
    LtTile<RB> tadd[NA];
    if constexpr (ADD) {
#pragma unroll
        for (int m = 0; m < NA; ++m) lt_tile_load<RB>(tadd[m], p.add, NA * LT_D, m * LT_D, row0, p.rows, tid);
    }
    const bool hi = tid >= LT_D;
    const int ve = tid & (LT_D - 1);
    float gv = 0.f, sv = 0.f;
    if (hi || FFN) gv = (hi ? p.gamma1 : p.gamma2)[ve];
    if (tid < 4 * R) {
        const int which = tid / R;
        const int row = min(row0 + (tid % R), p.rows - 1);
        if (which >= 2 || FFN) sv = (which == 0 ? p.mean2 : which == 1 ? p.rstd2 : which == 2 ? p.mean1 : p.rstd1)[row];
    }
    // ---- weights
    LtUnit ua, ub;
    if (FFN) lt_load_unit(ua, lt_unit_ptr(p.w2, LT_D, n0, 0, lane), LT_D);       // rows of slice 0 of W2, k over d
    else lt_load_unit(ua, lt_unit_ptr(p.wo, LT_D, n0, 0, lane), LT_D);
    if constexpr (ADD) lt_load_unit(ub, lt_unit_ptr(p.aw[0], LT_D, n0, 0, lane), LT_D);
    __builtin_amdgcn_sched_barrier(0);
    lt_tile_to_img<RB>(img_z1, tz1, tid);
    lt_tile_to_img<RB>(img_h0, tdy, tid);
    // dq, dk, dv rows: the two images nothing uses yet and the parking space of dz2 (written only
    // behind the barrier inside the first LayerNorm'); without the feed-forward part the one image
    // that is not used before the end
    bf16_t* const img_add[3] = {imgs[FFN ? 0 : 3], imgs[FFN ? 5 : 0], reinterpret_cast<bf16_t*>(&park[0][0])};
    if constexpr (ADD) {
#pragma unroll
        for (int m = 0; m < NA; ++m) lt_tile_to_img<RB>(img_add[m], tadd[m], tid);
    }
    if (FFN) {
        lt_tile_to_img<RB>(img_h1, tz2, tid);
        lt_tile_to_img<RB>(imgs[4], thp, tid);
        if (nslice > 1) lt_tile_load<RB>(thp, p.hpre, p.dff, LT_D, row0, p.rows, tid);    // slice 1, kept in registers
    }
    gam[hi][ve] = gv;
    if (tid < 4 * R) stat[tid / R][tid % R] = sv;
    __syncthreads();                                         // staged blocks and vectors visible

    float g[RB][2][4];
    bf16x4_t zv[RB][2];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 16 * j + 4 * fq;
            const bf16x4_t t = lt_img_get4<RB>(img_h0, 16 * i + fr, n);
#pragma unroll
            for (int r = 0; r < 4; ++r) g[i][j][r] = (float)t[r];
            if (FFN) zv[i][j] = lt_img_get4<RB>(img_h1, 16 * i + fr, n);
        }
    if constexpr (ADD) {
        f32x4_t aacc[RB][2];
        lt_zero<RB>(aacc);
        if constexpr (NA == 3) {
            lt_mul_unit<true, RB>(aacc, ub, img_add[0], lane, lt_unit_ptr(p.aw[1], LT_D, n0, 0, lane), LT_D);
            lt_mul_unit<true, RB>(aacc, ub, img_add[1], lane, lt_unit_ptr(p.aw[2], LT_D, n0, 0, lane), LT_D);
            lt_mul_unit<false, RB>(aacc, ub, img_add[2], lane, nullptr, 0);
        } else {
            lt_mul_unit<false, RB>(aacc, ub, img_add[0], lane, nullptr, 0);
        }
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) g[i][j][r] += aacc[i][j][r];
    }
    uint64_t key = 0;
    if (FFN) {
        lt_ln_bwd<RB>(g, zv, stat[0], stat[1], gam[0], row0, p.rows, wave, lane, red,
                  p.slab2 + (long)blockIdx.x * 2 * LT_D);
        // g = dz2: the gradient of y1 through the residual, and (masked) the gradient of f
        lt_load_unit(ub, lt_unit_ptr(p.w1, p.dff, n0, 0, lane), p.dff);      // rows n0.. of W1, k over slice 0
        if (p.thresh) key = rng_key(p.rng_state, p.site_res2);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            const int row = row0 + 16 * i + fr;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + 16 * j + 4 * fq;
                float dm[4] = {1.f, 1.f, 1.f, 1.f}, dfv[4];
                if (p.thresh) lt_dropout4(key, (uint64_t)row * LT_D + n, p.thresh, p.dscale, dm);
#pragma unroll
                for (int r = 0; r < 4; ++r) dfv[r] = g[i][j][r] * dm[r];
                lt_store4(img_a + lt_img_off<RB>(16 * i + fr, n), dfv);
                park[FFN ? 2 * i + j : 0][tid] = (f32x4_t){g[i][j][0], g[i][j][1], g[i][j][2], g[i][j][3]};
            }
        }
        __syncthreads();                                     // df image complete
        lt_img_to_global<RB>(img_a, p.df, LT_D, 0, row0, p.rows, tid);
        f32x4_t acc[RB][2], dacc[RB][2];
        lt_zero<RB>(dacc);
        uint64_t hkey = 0;
        if (p.thresh) hkey = rng_key(p.rng_state, p.site_ffn);
        // Pre-activations travel two slices ahead: slice c + 2 is requested (into registers) while
        // slice c is worked on, and slice c + 1 moves from registers into its LDS image before the
        // barrier of slice c, behind which slice c + 1 reads it.
        auto slice = [&](int c, auto more_tag) {
            constexpr bool MORE = decltype(more_tag)::value;
            bf16_t* himg = (c & 1) ? img_h1 : img_h0;
            const bf16_t* pimg = imgs[4 + (c & 1)];
            lt_zero<RB>(acc);
            // df W2[slice c]^T; the registers take slice c + 1, or the projection after the last slice
            lt_mul_unit<true, RB>(acc, ua, img_a, lane,
                              MORE ? lt_unit_ptr(p.w2, LT_D, (c + 1) * LT_D + n0, 0, lane) : lt_unit_ptr(p.wo, LT_D, n0, 0, lane),
                              LT_D);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                const int row = row0 + 16 * i + fr;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int nl = n0 + 16 * j + 4 * fq;
                    const int n = c * LT_D + nl;
                    const bf16x4_t hp = lt_img_get4<RB>(pimg, 16 * i + fr, nl);
                    float dv[4], dm[4] = {1.f, 1.f, 1.f, 1.f};
                    if (p.thresh) lt_dropout4(hkey, (uint64_t)row * p.dff + n, p.thresh, p.dscale, dm);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dv[r] = acc[i][j][r] * dgelu_tanh_f((float)hp[r]) * dm[r];
                    lt_store4(himg + lt_img_off<RB>(16 * i + fr, nl), dv);
                }
            }
            if (MORE) {
                lt_tile_to_img<RB>(imgs[5 - (c & 1)], thp, tid);
                if (c + 2 < nslice) lt_tile_load<RB>(thp, p.hpre, p.dff, (c + 2) * LT_D, row0, p.rows, tid);
            }
            __syncthreads();                                 // slice c of dh is in LDS
            lt_img_to_global<RB>(himg, p.dh, p.dff, c * LT_D, row0, p.rows, tid);
            lt_mul_unit<MORE, RB>(dacc, ub, himg, lane, lt_unit_ptr(p.w1, p.dff, n0, (c + 1) * LT_D, lane), p.dff);  // += dh[slice c] W1[:, slice c]^T
        };
        for (int c = 0; c + 1 < nslice; ++c) slice(c, std::true_type{});
        slice(nslice - 1, std::false_type{});
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f32x4_t t = park[FFN ? 2 * i + j : 0][tid];
#pragma unroll
                for (int r = 0; r < 4; ++r) g[i][j][r] = t[r] + dacc[i][j][r];
            }
    }
    // g = gradient of y1
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) zv[i][j] = lt_img_get4<RB>(img_z1, 16 * i + fr, n0 + 16 * j + 4 * fq);
    lt_ln_bwd<RB>(g, zv, stat[2], stat[3], gam[1], row0, p.rows, wave, lane, red, p.slab1 + (long)blockIdx.x * 2 * LT_D);
    // (a barrier passed: nobody reads the dh images or, without the feed-forward part, dy any more)
    if (p.thresh) key = rng_key(p.rng_state, p.site_res1);
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int row = row0 + 16 * i + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 16 * j + 4 * fq;
            float dm[4] = {1.f, 1.f, 1.f, 1.f}, dav[4];
            if (p.thresh) lt_dropout4(key, (uint64_t)row * LT_D + n, p.thresh, p.dscale, dm);
#pragma unroll
            for (int r = 0; r < 4; ++r) dav[r] = g[i][j][r] * dm[r];
            lt_store4(img_h0 + lt_img_off<RB>(16 * i + fr, n), g[i][j]);
            lt_store4(img_a + lt_img_off<RB>(16 * i + fr, n), dav);
        }
    }
    __syncthreads();                                         // da and dx images complete
    lt_img_to_global<RB>(img_h0, p.dx, LT_D, 0, row0, p.rows, tid);
    lt_img_to_global<RB>(img_a, p.da, LT_D, 0, row0, p.rows, tid);
    f32x4_t cacc[RB][2];
    lt_zero<RB>(cacc);
    lt_mul_unit<false, RB>(cacc, ua, img_a, lane, nullptr, 0);   // da Wo^T
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float o[4] = {cacc[i][j][0], cacc[i][j][1], cacc[i][j][2], cacc[i][j][3]};
            lt_store4(img_h1 + lt_img_off<RB>(16 * i + fr, n0 + 16 * j + 4 * fq), o);
        }
    __syncthreads();
    lt_img_to_global<RB>(img_h1, p.dctx, LT_D, 0, row0, p.rows, tid);
}

// ------------------------------------------------------------------------------------- C ABI
static bool lt_al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int athena_layer_tail_workgroups(int rows) {
    const int r = 16 * lt_row_blocks(rows);
    return rows > 0 ? (rows + r - 1) / r : 0;
}

extern "C" int athena_layer_tail_fwd(const athena_layer_tail_fwd_desc* d, athena_stream_t stream) {
    ATHENA_REQUIRE(d, ATHENA_E_SHAPE, "athena_layer_tail_fwd: null descriptor");
    ATHENA_REQUIRE(d->d_model == LT_D && d->dff >= 0 && d->dff % LT_D == 0 && d->dff <= LT_MAX_DFF, ATHENA_E_UNSUPPORTED,
                   "athena_layer_tail_fwd: d_model %d dff %d (needs d_model 256 and dff a multiple of 256 up to %d)",
                   d->d_model, d->dff, LT_MAX_DFF);
    ATHENA_REQUIRE(d->rows > 0 && (long)d->rows * (d->dff > LT_D ? d->dff : LT_D) < (1L << 31), ATHENA_E_SHAPE,
                   "athena_layer_tail_fwd: rows=%d", d->rows);
    ATHENA_REQUIRE(d->ctx && d->x && d->wo_t && d->bo && d->gamma1 && d->beta1 && d->z1 && d->y1 && d->mean1 && d->rstd1,
                   ATHENA_E_SHAPE, "athena_layer_tail_fwd: null operand");
    ATHENA_REQUIRE(d->dff == 0 || (d->w1_t && d->b1 && d->w2_t && d->b2 && d->gamma2 && d->beta2 && d->hpre && d->h &&
                                   d->z2 && d->y2 && d->mean2 && d->rstd2),
                   ATHENA_E_SHAPE, "athena_layer_tail_fwd: null feed-forward operand");
    ATHENA_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || d->rng_state), ATHENA_E_SHAPE,
                   "athena_layer_tail_fwd: dropout arguments");
    for (const void* q : {d->ctx, d->x, d->wo_t, (const void*)d->bo, (const void*)d->gamma1, (const void*)d->beta1,
                          d->w1_t, (const void*)d->b1, d->w2_t, (const void*)d->b2, (const void*)d->gamma2,
                          (const void*)d->beta2, (const void*)d->z1, (const void*)d->y1, (const void*)d->hpre,
                          (const void*)d->h, (const void*)d->z2, (const void*)d->y2})
        ATHENA_REQUIRE(lt_al16(q), ATHENA_E_SHAPE, "athena_layer_tail_fwd: operands must be 16-byte aligned");
    LtFwd p;
    p.rows = d->rows; p.dff = d->dff;
    p.ctx = (const bf16_t*)d->ctx; p.x = (const bf16_t*)d->x;
    p.wo = (const bf16_t*)d->wo_t; p.bo = d->bo; p.gamma1 = d->gamma1; p.beta1 = d->beta1;
    p.w1 = (const bf16_t*)d->w1_t; p.b1 = d->b1; p.w2 = (const bf16_t*)d->w2_t; p.b2 = d->b2;
    p.gamma2 = d->gamma2; p.beta2 = d->beta2;
    p.eps = d->eps;
    p.thresh = dropout_threshold(d->drop_p);
    p.dscale = p.thresh ? 1.f / (1.f - d->drop_p) : 1.f;
    p.rng_state = d->rng_state;
    p.site_res1 = d->site_res1; p.site_ffn = d->site_ffn; p.site_res2 = d->site_res2;
    p.z1 = (bf16_t*)d->z1; p.y1 = (bf16_t*)d->y1; p.mean1 = d->mean1; p.rstd1 = d->rstd1;
    p.hpre = (bf16_t*)d->hpre; p.h = (bf16_t*)d->h;
    p.z2 = (bf16_t*)d->z2; p.y2 = (bf16_t*)d->y2; p.mean2 = d->mean2; p.rstd2 = d->rstd2;
    ATHENA_REQUIRE(lt_al16(d->x_f32) && lt_al16(d->y_out_f32), ATHENA_E_SHAPE,
                   "athena_layer_tail_fwd: the fp32 residual buffers must be 16-byte aligned");
    p.x32 = d->x_f32;
    p.y32 = d->y_out_f32;
    const bool qkv = d->next_w_t[0] != nullptr;
    for (int m = 0; m < 3; ++m) {
        ATHENA_REQUIRE(!qkv || (m > 0 && d->dff == 0) ||
                           (d->next_w_t[m] && d->next_b[m] && d->next_out[m] && lt_al16(d->next_w_t[m]) &&
                            lt_al16(d->next_out[m])),
                       ATHENA_E_SHAPE, "athena_layer_tail_fwd: the extra projections (three with a feed-forward part, "
                                       "one without) need weights, biases and 16-byte aligned outputs");
        p.wn[m] = (const bf16_t*)d->next_w_t[m];
        p.bn[m] = d->next_b[m];
        p.on[m] = (bf16_t*)d->next_out[m];
    }
    const dim3 grid(athena_layer_tail_workgroups(d->rows));
#define LT_FWD(F, Q)                                                                                              \
    do {                                                                                                          \
        if (lt_row_blocks(d->rows) == 1)                                                                          \
            ATHENA_LAUNCH((layer_tail_fwd_kernel<F, Q, 1>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p); \
        else                                                                                                      \
            ATHENA_LAUNCH((layer_tail_fwd_kernel<F, Q, 2>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p); \
    } while (0)
    if (qkv && d->dff > 0) LT_FWD(true, true);
    else if (qkv) LT_FWD(false, true);
    else if (d->dff > 0) LT_FWD(true, false);
    else LT_FWD(false, false);
#undef LT_FWD
    ATHENA_CHECK_LAUNCH("athena_layer_tail_fwd");
    return ATHENA_OK;
}

extern "C" int athena_layer_tail_bwd(const athena_layer_tail_bwd_desc* d, athena_stream_t stream) {
    ATHENA_REQUIRE(d, ATHENA_E_SHAPE, "athena_layer_tail_bwd: null descriptor");
    ATHENA_REQUIRE(d->d_model == LT_D && d->dff >= 0 && d->dff % LT_D == 0 && d->dff <= LT_MAX_DFF, ATHENA_E_UNSUPPORTED,
                   "athena_layer_tail_bwd: d_model %d dff %d (needs d_model 256 and dff a multiple of 256 up to %d)",
                   d->d_model, d->dff, LT_MAX_DFF);
    ATHENA_REQUIRE(d->rows > 0 && (long)d->rows * (d->dff > LT_D ? d->dff : LT_D) < (1L << 31), ATHENA_E_SHAPE,
                   "athena_layer_tail_bwd: rows=%d", d->rows);
    ATHENA_REQUIRE(d->dy && d->z1 && d->mean1 && d->rstd1 && d->gamma1 && d->wo && d->da && d->dx && d->dctx && d->slab1,
                   ATHENA_E_SHAPE, "athena_layer_tail_bwd: null operand");
    ATHENA_REQUIRE(d->dff == 0 || (d->z2 && d->mean2 && d->rstd2 && d->gamma2 && d->hpre && d->w2 && d->w1 && d->df &&
                                   d->dh && d->slab2),
                   ATHENA_E_SHAPE, "athena_layer_tail_bwd: null feed-forward operand");
    ATHENA_REQUIRE(d->drop_p >= 0.f && d->drop_p < 1.f && (d->drop_p == 0.f || d->rng_state), ATHENA_E_SHAPE,
                   "athena_layer_tail_bwd: dropout arguments");
    for (const void* q : {d->dy, d->z2, (const void*)d->gamma2, d->hpre, d->w2, d->w1, d->z1, (const void*)d->gamma1,
                          d->wo, (const void*)d->df, (const void*)d->dh, (const void*)d->da, (const void*)d->dx,
                          (const void*)d->dctx, (const void*)d->slab1, (const void*)d->slab2})
        ATHENA_REQUIRE(lt_al16(q), ATHENA_E_SHAPE, "athena_layer_tail_bwd: operands must be 16-byte aligned");
    LtBwd p;
    p.rows = d->rows; p.dff = d->dff;
    p.dy = (const bf16_t*)d->dy; p.z2 = (const bf16_t*)d->z2; p.mean2 = d->mean2; p.rstd2 = d->rstd2;
    p.gamma2 = d->gamma2; p.hpre = (const bf16_t*)d->hpre; p.w2 = (const bf16_t*)d->w2; p.w1 = (const bf16_t*)d->w1;
    p.z1 = (const bf16_t*)d->z1; p.mean1 = d->mean1; p.rstd1 = d->rstd1; p.gamma1 = d->gamma1;
    p.wo = (const bf16_t*)d->wo;
    p.thresh = dropout_threshold(d->drop_p);
    p.dscale = p.thresh ? 1.f / (1.f - d->drop_p) : 1.f;
    p.rng_state = d->rng_state;
    p.site_res1 = d->site_res1; p.site_ffn = d->site_ffn; p.site_res2 = d->site_res2;
    p.df = (bf16_t*)d->df; p.dh = (bf16_t*)d->dh; p.da = (bf16_t*)d->da; p.dx = (bf16_t*)d->dx;
    p.dctx = (bf16_t*)d->dctx; p.slab2 = d->slab2; p.slab1 = d->slab1;
    const bool add = d->add_dqkv != nullptr;
    ATHENA_REQUIRE(!add || (d->add_w[0] && lt_al16(d->add_dqkv) && lt_al16(d->add_w[0]) &&
                            (d->dff == 0 || (d->add_w[1] && d->add_w[2] && lt_al16(d->add_w[1]) && lt_al16(d->add_w[2])))),
                   ATHENA_E_SHAPE, "athena_layer_tail_bwd: the extra product needs its weight copies (three with a "
                                   "feed-forward part, one without) and 16-byte aligned operands");
    p.add = (const bf16_t*)d->add_dqkv;
    for (int m = 0; m < 3; ++m) p.aw[m] = (const bf16_t*)d->add_w[m];
    const dim3 grid(athena_layer_tail_workgroups(d->rows));
#define LT_BWD(F, A)                                                                                              \
    do {                                                                                                          \
        if (lt_row_blocks(d->rows) == 1)                                                                          \
            ATHENA_LAUNCH((layer_tail_bwd_kernel<F, A, 1>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p); \
        else                                                                                                      \
            ATHENA_LAUNCH((layer_tail_bwd_kernel<F, A, 2>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p); \
    } while (0)
    if (add && d->dff > 0) LT_BWD(true, true);
    else if (add) LT_BWD(false, true);
    else if (d->dff > 0) LT_BWD(true, false);
    else LT_BWD(false, false);
#undef LT_BWD
    ATHENA_CHECK_LAUNCH("athena_layer_tail_bwd");
    return ATHENA_OK;
}
