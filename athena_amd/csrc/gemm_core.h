// GEMM cores shared by the Dense and Conv2D entry points.
//
//   C[m, n] = epilogue( alpha * sum_k A(m, k) * B(k, n) )
//
// Two cores with the same problem description:
//   * gemm_mfma_kernel  : bf16 operands, fp32 accumulate on the matrix cores
//                         (v_mfma_f32_16x16x32_bf16), tiles staged through a double-buffered LDS,
//                         4 waves of 64 lanes, each owning a quarter of the output tile.
//   * gemm_valu_kernel  : fp32 operands, fp32 FMA on the vector ALUs, 64x64x16 tiles; the
//                         exact-arithmetic path used by the fp32 precision mode.
//
// The contraction dimension is cut into ``nseg`` segments of ``kseg`` elements.  Within a
// segment the A operand is addressed through a row map (plain rows, or the im2col gather of a
// strided convolution), so the same cores serve Dense layers and the implicit-GEMM convolution.
#pragma once
#include "common.h"

// ------------------------------------------------------------------ epilogue
struct Epilogue {
    void* C;
    void* C_pre;
    const float* bias;
    const void* aux;
    const uint64_t* rng_state;
    double* colsum;    // optional [2*N] per-column sum / sum of squares of the stored values
    float* partial;    // split-K slabs [grid.z][M + 1][ldp] (row M: column sums of B) when atomic == 2
    float* bsum;       // optional [N]: sum over k of B(k, n), taken by the first row of workgroups
    int ldp;
    int ldc, ldaux;
    int c_dtype, aux_dtype;
    int act;
    int atomic;        // 0: single writer; 1: fp32 atomics onto C; 2: raw partial sums into ``partial``
    int accum;
    float alpha;
    float drop_scale;
    uint32_t drop_thresh;
    uint32_t rng_site;
};

__device__ __forceinline__ float epilogue_value(const Epilogue& e, long ci, long ai, int n, float acc, uint64_t key,
                                                float* pre_out) {
    float v = acc * e.alpha;
    if (e.bias) v += e.bias[n];
    *pre_out = v;
    switch (e.act) {
        case ATHENA_ACT_RELU6: v = relu6f(v); break;
        case ATHENA_ACT_GELU: v = gelu_tanh_f(v); break;
        case ATHENA_ACT_MUL_DGELU: v *= dgelu_tanh_f(load_as_f32(e.aux, ai, e.aux_dtype)); break;
        case ATHENA_ACT_MUL_DRELU6: {
            float y = load_as_f32(e.aux, ai, e.aux_dtype);
            v = (y > 0.f && y < 6.f) ? v : 0.f;
        } break;
        default:
            if (e.act >= ATHENA_ACT_MUL_DUNARY) v *= dunary_f(e.act - ATHENA_ACT_MUL_DUNARY, load_as_f32(e.aux, ai, e.aux_dtype));
            else if (e.act >= ATHENA_ACT_UNARY) v = unary_f(e.act - ATHENA_ACT_UNARY, v);
            break;
    }
    if (e.drop_thresh) v *= dropout_mult(key, (uint64_t)ci, e.drop_thresh, e.drop_scale);
    return v;
}

__device__ __forceinline__ void epilogue_store(const Epilogue& e, int m, int n, float acc, uint64_t key,
                                               long coff = 0) {
    const long ci = (long)m * e.ldc + n;
    if (e.atomic) {
        atomicAdd((float*)e.C + coff + ci, acc * e.alpha);
        return;
    }
    float pre;
    float v = epilogue_value(e, ci, (long)m * e.ldaux + n, n, acc, key, &pre);
    if (e.C_pre) store_from_f32(e.C_pre, ci, e.c_dtype, pre);
    if (e.accum) v += load_as_f32(e.C, ci, e.c_dtype);
    store_from_f32(e.C, ci, e.c_dtype, v);
}

// four consecutive columns n..n+3 of output row m (caller guarantees n + 3 < N)
__device__ __forceinline__ void epilogue_store4(const Epilogue& e, int m, int n, f32x4_t acc, uint64_t key,
                                                bool vec_ok, long coff = 0) {
    const long ci = (long)m * e.ldc + n;
    if (e.atomic) {
        float* c = (float*)e.C + coff + ci;
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(c + r, acc[r] * e.alpha);
        return;
    }
    if (!vec_ok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) epilogue_store(e, m, n + r, acc[r], key);
        return;
    }
    const long ai = (long)m * e.ldaux + n;
    float v[4], pre[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = epilogue_value(e, ci + r, ai + r, n + r, acc[r], key, &pre[r]);
    if (e.c_dtype == ATHENA_BF16) {
        bf16_t* c = (bf16_t*)e.C + ci;
        if (e.C_pre) {
            bf16x4_t pv = {(bf16_t)pre[0], (bf16_t)pre[1], (bf16_t)pre[2], (bf16_t)pre[3]};
            *(bf16x4_t*)((bf16_t*)e.C_pre + ci) = pv;
        }
        if (e.accum) {
            bf16x4_t old = *(const bf16x4_t*)c;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)old[r];
        }
        bf16x4_t o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *(bf16x4_t*)c = o;
    } else {
        float* c = (float*)e.C + ci;
        if (e.C_pre) *(float4*)((float*)e.C_pre + ci) = make_float4(pre[0], pre[1], pre[2], pre[3]);
        if (e.accum) {
            float4 old = *(const float4*)c;
            v[0] += old.x; v[1] += old.y; v[2] += old.z; v[3] += old.w;
        }
        *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ------------------------------------------------------------------ row maps for the A operand
// prep(r) decodes row r once; off(row, seg) = element offset of that row within segment ``seg``
// (or -1 when the row is all zeros there); step(row, d) moves to row r + d.
struct PlainRows {
    static constexpr bool kCanMiss = false;    // off() never reports an all-zero row
    static constexpr bool kSegmented = false;  // the contraction is one segment
    long ld;
    struct Row { long base; };
    __device__ __forceinline__ Row prep(int r) const { return Row{(long)r * ld}; }
    __device__ __forceinline__ long off(const Row& x, int) const { return x.base; }
    __device__ __forceinline__ void step(Row& x, int d) const { x.base += (long)d * ld; }
};

// Rows of the im2col matrix of a 3x3 / stride-2 / "same" convolution over an NHWC tensor
// [B, H, W, C].  Row r = (b, ho, wo); segment = tap (kh*3+kw); the row is the C-vector at
// input pixel (2*ho - pad_h + kh, 2*wo - pad_w + kw), zero outside the image.
struct ConvFwdRows {
    static constexpr bool kCanMiss = true;
    static constexpr bool kSegmented = true;
    int H, W, C, Ho, Wo, pad_h, pad_w;
    int step_q, step_r;          // GM_BK / Wo and GM_BK % Wo (host side), for step() by one k-tile
    struct Row { int b, ho, wo; };
    __device__ __forceinline__ Row prep(int r) const {
        Row x;
        x.wo = r % Wo;
        int t = r / Wo;
        x.ho = t % Ho;
        x.b = t / Ho;
        return x;
    }
    __device__ __forceinline__ long off(const Row& x, int seg) const {
        const int kh = seg / 3, kw = seg - kh * 3;
        const int h = 2 * x.ho - pad_h + kh, w = 2 * x.wo - pad_w + kw;
        if (h < 0 || h >= H || w < 0 || w >= W) return -1;
        return (long)((x.b * H + h) * W + w) * C;     // pixel index fits 31 bits (checked on the host)
    }
    // advance by one k-tile (d == GM_BK rows): one conditional carry per coordinate instead of a
    // division; the second loop only iterates when an image has fewer rows than the carry
    __device__ __forceinline__ void step(Row& x, int) const {
        x.wo += step_r;
        x.ho += step_q;
        if (x.wo >= Wo) { x.wo -= Wo; ++x.ho; }
        while (x.ho >= Ho) { x.ho -= Ho; ++x.b; }
    }
};

// Rows for the input gradient of the same convolution, one launch per parity class of the
// input pixel.  Row r = (b, ah, aw) <-> one input pixel of the class; segment s indexes the taps
// that reach this class; the row is the Cout-vector of the output gradient at
// (ah + dh(s), aw + dw(s)), both affine in the two bit fields of s.
struct ConvDgradRows {
    static constexpr bool kCanMiss = true;
    static constexpr bool kSegmented = true;
    int Ho, Wo, Co, Ah, Aw;     // Ah x Aw = pixels of this class per image
    int dh0, dw0, shift, mask;  // segment s reads the output pixel (ah + dh0 - (s >> shift), aw + dw0 - (s & mask))
    struct Row { int b, ah, aw; };
    __device__ __forceinline__ Row prep(int r) const {
        Row x;
        x.aw = r % Aw;
        int t = r / Aw;
        x.ah = t % Ah;
        x.b = t / Ah;
        return x;
    }
    __device__ __forceinline__ long off(const Row& x, int seg) const {
        const int ho = x.ah + dh0 - (seg >> shift), wo = x.aw + dw0 - (seg & mask);
        if (ho < 0 || ho >= Ho || wo < 0 || wo >= Wo) return -1;
        return (((long)x.b * Ho + ho) * Wo + wo) * Co;
    }
    __device__ __forceinline__ void step(Row& x, int d) const {
        x.aw += d;
        while (x.aw >= Aw) { x.aw -= Aw; ++x.ah; }
        while (x.ah >= Ah) { x.ah -= Ah; ++x.b; }
    }
};

// Rows of the im2col matrix of a KH x KW convolution with strides (sh, sw) and "same" padding over
// an NHWC tensor [B, H, W, C].  Row r = (b, ho, wo); segment = tap (kh*KW + kw); the row is the
// C-vector at input pixel (sh*ho - pad_h + kh, sw*wo - pad_w + kw), zero outside the image.
struct ConvTapRows {
    static constexpr bool kCanMiss = true;
    static constexpr bool kSegmented = true;
    int H, W, C, Ho, Wo, pad_h, pad_w, KW, sh, sw;
    int step_q, step_r;          // rows per k-step / Wo and % Wo, for step() by one k-tile
    struct Row { int b, ho, wo; };
    __device__ __forceinline__ Row prep(int r) const {
        Row x;
        x.wo = r % Wo;
        int t = r / Wo;
        x.ho = t % Ho;
        x.b = t / Ho;
        return x;
    }
    __device__ __forceinline__ long off(const Row& x, int seg) const {
        const int kh = seg / KW, kw = seg - kh * KW;
        const int h = sh * x.ho - pad_h + kh, w = sw * x.wo - pad_w + kw;
        if (h < 0 || h >= H || w < 0 || w >= W) return -1;
        return (long)((x.b * H + h) * W + w) * C;
    }
    __device__ __forceinline__ void step(Row& x, int) const {
        x.wo += step_r;
        x.ho += step_q;
        if (x.wo >= Wo) { x.wo -= Wo; ++x.ho; }
        while (x.ho >= Ho) { x.ho -= Ho; ++x.b; }
    }
};

// Rows for the input gradient of that convolution.  Row r = one input pixel (b, h, w); segment =
// tap (kh*KW + kw); the row is the Cout-vector of the output gradient at the output pixel this tap
// connects the input pixel to, ((h + pad_h - kh) / sh, (w + pad_w - kw) / sw), and all zeros when
// the division leaves a remainder or the pixel lies outside the output.
struct ConvTapDgradRows {
    static constexpr bool kCanMiss = true;
    static constexpr bool kSegmented = true;
    int H, W, Ho, Wo, Co, pad_h, pad_w, KW, sh, sw;
    struct Row { int b, h, w; };
    __device__ __forceinline__ Row prep(int r) const {
        Row x;
        x.w = r % W;
        int t = r / W;
        x.h = t % H;
        x.b = t / H;
        return x;
    }
    __device__ __forceinline__ long off(const Row& x, int seg) const {
        const int kh = seg / KW, kw = seg - kh * KW;
        const int th = x.h + pad_h - kh, tw = x.w + pad_w - kw;
        if (th < 0 || tw < 0) return -1;
        const int ho = th / sh, wo = tw / sw;
        if (ho * sh != th || wo * sw != tw || ho >= Ho || wo >= Wo) return -1;
        return (((long)x.b * Ho + ho) * Wo + wo) * Co;
    }
    __device__ __forceinline__ void step(Row& x, int d) const {
        x.w += d;
        while (x.w >= W) { x.w -= W; ++x.h; }
        while (x.h >= H) { x.h -= H; ++x.b; }
    }
};

// ------------------------------------------------------------------ problem description
template <class AMap>
struct GemmProblem {
    const void* A;
    const void* B;
    AMap amap;
    int M, N;
    int nseg, kseg;          // K = nseg * kseg
    long ldb;                // leading dimension of B
    // element offset of B for segment s: b_base + (s >> b_shift) * b_stride_hi + (s & b_mask) * b_stride_lo
    long b_base, b_stride_lo, b_stride_hi;
    int b_shift, b_mask;
    int seg_base;            // segment index passed to the row map = seg_base + local segment
    int split_k;             // number of K splits; splits are over k-tiles
    int zbatch;              // independent problems in grid.z (grid.z = zbatch * split_k); batch z uses
    long c_batch_stride;     //   row-map segment seg_base + z and output C + z * c_batch_stride (fp32 elements)
    Epilogue epi;
    // optional output row remap for strided scatter of C rows (ConvDgrad): row m -> c_row(m)
    int crow_mode;           // 0: identity; 1: parity-class scatter
    int cr_Ah, cr_Aw, cr_H, cr_W, cr_h0, cr_w0;   // crow = ((b*H + 2*ah+h0)*W + 2*aw+w0)
    // XCD-aware tile order of a 1-D launch (0 = off, the grid is (tiles_m, tiles_n, z)): workgroup ids
    // go round-robin over the 8 XCDs, so id -> (XCD = id % 8, position j = id / 8 on that XCD); the
    // xcd_tn column tiles of one row tile take consecutive positions of one XCD and find the A rows
    // their neighbour just fetched in that XCD's L2
    int xcd_tn, xcd_tm;
};

template <class AMap>
__device__ __forceinline__ long b_seg_off(const GemmProblem<AMap>& p, int seg) {
    return p.b_base + (long)(seg >> p.b_shift) * p.b_stride_hi + (long)(seg & p.b_mask) * p.b_stride_lo;
}

template <class AMap>
__device__ __forceinline__ int out_row(const GemmProblem<AMap>& p, int m) {
    if (p.crow_mode == 0) return m;
    int aw = m % p.cr_Aw;
    int t = m / p.cr_Aw;
    int ah = t % p.cr_Ah;
    int b = t / p.cr_Ah;
    return (b * p.cr_H + 2 * ah + p.cr_h0) * p.cr_W + 2 * aw + p.cr_w0;
}

// ------------------------------------------------------------------ MFMA core (bf16)
#define GM_BK 64

// Workgroup = 4 waves as 2 (M) x 2 (N); each wave owns WT x WT MFMA tiles of 16x16, so the
// workgroup tile is (32*WT) x (32*WT): WT = 4 -> 128x128 (large outputs), WT = 2 -> 64x64
// (small outputs, more workgroups in flight).  K advances 64 per LDS stage, two stages.
//
// Operand layouts and their LDS images:
//   TA = false: A(m, k) = A[row(m) + k]   contraction contiguous -> image [m][k], row read (ds_read_b128)
//   TA = true : A(m, k) = A[row(k) + m]   output row contiguous  -> image [k][m], transposed read
//   TB = true : B(k, n) = B[seg_off + n*ldb + k]   -> image [n][k], row read
//   TB = false: B(k, n) = B[seg_off + k*ldb + n]   -> image [k][n], transposed read
// A transposed read is two ds_read_b64_tr_b16 per fragment: each hands a lane the four k-values
// of its column from a 4 x 16 block of 16-bit elements.
//
// The MFMA is issued with the operands swapped (B-fragment first), so a lane ends up with four
// CONSECUTIVE columns of one output row: D[n = 4*(lane>>4) + r][m = lane & 15].  The epilogue
// then stores 8 or 16 contiguous bytes per lane.
// LDS images are unpadded and XOR-swizzled at 16-byte (8-element) chunk granularity.
//   row-major image [rows][64 k]: chunk c of row r lives in slot c ^ (r & 7); the 16 lanes of a
//     ds_read_b128 group then touch 16 different 16-byte slots of the 256-byte bank row.
//   transposed image [64 k][NC*8 columns]: chunk c of k-row k lives in slot c ^ tr_swz<NC>(k), so the
//     eight k-rows a 32-lane half of ds_read_b64_tr_b16 touches land on different banks.
__device__ __forceinline__ int row_img_off(int r, int chunk) { return (r * 8 + (chunk ^ (r & 7))) * 8; }
template <int NC>
__device__ __forceinline__ int tr_swz(int k) {
    return NC >= 16 ? (2 * (k & 3) + 8 * ((k >> 3) & 1)) : (2 * ((k >> 1) & 1) + 4 * ((k >> 3) & 1));
}
template <int NC>
__device__ __forceinline__ int tr_img_off(int k, int chunk) { return (k * NC + (chunk ^ tr_swz<NC>(k))) * 8; }

__device__ __forceinline__ bf16x8_t lds_frag_rows(const bf16_t* img, int row, int kofs) {
    return *(const bf16x8_t*)(img + row_img_off(row, kofs >> 3));
}
template <int NC>
__device__ __forceinline__ bf16x8_t lds_frag_tr(const bf16_t* img, int col0, int k0, int lane) {
    // block rows k0 + 8g + q (q = 0..3), then + 4; lane 4q+p of the 16-lane group addresses row q, columns 4p..4p+3
    const int g = lane >> 4, ig = lane & 15, q = ig >> 2, pp = ig & 3;
    const int k = k0 + 8 * g + q;
    const int chunk = (col0 >> 3) + (pp >> 1);
    const bf16_t* a0 = img + tr_img_off<NC>(k, chunk) + (pp & 1) * 4;
    const bf16_t* a1 = img + tr_img_off<NC>(k + 4, chunk) + (pp & 1) * 4;
    typedef short s4 __attribute__((ext_vector_type(4)));
    s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a0);
    s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a1);
    typedef short s8 __attribute__((ext_vector_type(8)));
    s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// One workgroup's share of a problem: output tile (bx, by), K split / batch index bz.  ``p`` lives in
// the kernel arguments and is never written (a modified by-value struct would be spilled to
// scratch memory).
template <class AMap, bool TA, bool TB, int WT>
__device__ __forceinline__ void gemm_mfma_body(const GemmProblem<AMap>& p, const int bx, const int by, const int bz) {
    // independent problems stacked in grid.z
    const int zb = p.zbatch > 1 ? (int)(bz / p.split_k) : 0;
    const int zsplit = p.zbatch > 1 ? (int)(bz % p.split_k) : (int)bz;
    const int seg_base = p.seg_base + zb;
    const long coff = (long)zb * p.c_batch_stride;
    constexpr int BT = 32 * WT;                       // tile edge
    constexpr int NCH = BT * (GM_BK / 8) / 256;       // 16-byte pieces per thread per operand stage
    constexpr int NC = BT / 8;                        // 16-byte chunks per row of a transposed image
    constexpr int A_ELEMS = BT * GM_BK, B_ELEMS = BT * GM_BK;
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (A_ELEMS + B_ELEMS)];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = bx * BT;
    const int bn = by * BT;
    const bf16_t* __restrict__ A = (const bf16_t*)p.A;
    const bf16_t* __restrict__ B = (const bf16_t*)p.B;

    const int tiles_per_seg = (p.kseg + GM_BK - 1) / GM_BK;
    const int ntiles = p.nseg * tiles_per_seg;
    const int per_split = (ntiles + p.split_k - 1) / p.split_k;
    const int t_begin = zsplit * per_split;
    const int t_end = min(ntiles, t_begin + per_split);

    f32x4_t acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // per-thread staging pieces.  Row-major operand: piece c -> (row c/8, k-chunk c%8);
    // transposed operand: piece c -> (k-row c/NC, column chunk c%NC).  Rows and column chunks
    // outside the matrix are clamped to the last valid one rather than zero-filled: they only feed
    // output rows / columns that are never stored, and the loads stay free of branches.
    typename AMap::Row arow[NCH];
    int acol[NCH];                 // element offset inside the A row: k-chunk, or the clamped m-chunk
    const bf16_t* pb[NCH];         // B piece at k = 0 of a segment
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = tid + 256 * i;
        if (!TA) {
            arow[i] = p.amap.prep(min(bm + (c >> 3), p.M - 1));
            acol[i] = (c & 7) * 8;
        } else {
            // rows follow the contraction index: start at the first tile of this split (nseg == 1)
            arow[i] = p.amap.prep(min(t_begin * GM_BK + c / NC, p.kseg - 1));
            acol[i] = min(bm + (c % NC) * 8, ((p.M + 7) & ~7) - 8);
        }
        if (TB) pb[i] = B + (long)min(bn + (c >> 3), p.N - 1) * p.ldb + (c & 7) * 8;
        else pb[i] = B + (long)(c / NC) * p.ldb + min(bn + (c % NC) * 8, ((p.N + 7) & ~7) - 8);
    }

    // two register staging sets: while tile t is multiplied out of LDS, tile t+1 waits in one set
    // for its turn to be written to the other LDS stage and the loads of tile t+2 are in flight
    // into the other set, so a global load has two tile-times to arrive
    uint4 ra0[NCH], rb0[NCH], ra1[NCH], rb1[NCH];
    uint32_t km0 = 0, km1 = 0;
    // column sums of B (bias gradient of a weight-gradient GEMM): in the first row of workgroups
    // the waves with wm == 0 multiply their B fragments with an all-ones A fragment as well, which
    // leaves sum_k B(k, n) in every column of one extra accumulator per 16 columns
    const bool want_bsum = !TB && p.epi.bsum != nullptr && bx == 0 && wm == 0;
    f32x4_t bs[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) bs[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int r = 0; r < 8; ++r) ones[r] = (bf16_t)1.0f;

    // tiles are loaded strictly in order t_begin, t_begin + 1, ...; (segment, k offset) of the next
    // tile to load are carried along in scalar registers instead of being re-derived by a division
    int ld_t = t_begin;
    int ld_seg = AMap::kSegmented ? t_begin / tiles_per_seg : 0;
    int ld_k0 = (t_begin - ld_seg * tiles_per_seg) * GM_BK;
    // A tile is fetched into a register set together with a mask of the pieces that lie inside the
    // matrices (bit i: A piece i, bit 8 + i: B piece i); pieces outside are read from a valid
    // address and zeroed when the set is written to LDS.  The loaded registers are not touched
    // between the load and that write, on any path, so the loads of the following tiles stay in
    // flight behind them (a use right after the load, or on one arm of a branch only, would make
    // the wait for an older tile cover the younger ones as well).
    constexpr uint32_t KM_FULL = ((1u << NCH) - 1u) * 0x101u;
    auto load_tile = [&](uint4 (&ra)[NCH], uint4 (&rb)[NCH], uint32_t& km) {
        const bool live = ld_t < t_end;
        const int seg = live ? ld_seg : 0;
        const int k0 = ld_k0;
        const long bseg = AMap::kSegmented ? b_seg_off(p, seg) : p.b_base;
        ++ld_t;
        ld_k0 += GM_BK;
        if (AMap::kSegmented && ld_k0 >= p.kseg) { ld_k0 = 0; ++ld_seg; }
        uint32_t m = 0;
        if (live && k0 + GM_BK <= p.kseg) {
            // whole tile inside the contraction range (the common case)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                long o = p.amap.off(arow[i], seg_base + seg);
                bool ok = true;
                if (AMap::kCanMiss) {
                    ok = o >= 0;
                    o = ok ? o : 0;
                }
                ra[i] = *(const uint4*)(A + o + (TA ? acol[i] : k0 + acol[i]));
                rb[i] = *(const uint4*)(pb[i] + bseg + (TB ? (long)k0 : (long)k0 * p.ldb));
                m |= ((ok ? 1u : 0u) << i) | (1u << (8 + i));
            }
        } else {
            // the last, partial tile of a segment, or a tile index past the end (all zeros)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + 256 * i;
                const int ka = k0 + (TA ? c / NC : acol[i]);
                const int kb = k0 + (TB ? (c & 7) * 8 : c / NC);
                const long o = p.amap.off(arow[i], seg_base + seg);
                const bool oka = live && ka < p.kseg && o >= 0;
                const bool okb = live && kb < p.kseg;
                const long offa = oka ? o + (TA ? acol[i] : ka) : 0;
                const bf16_t* ptrb = okb ? pb[i] + bseg + (TB ? (long)k0 : (long)k0 * p.ldb) : B;
                ra[i] = *(const uint4*)(A + offa);
                rb[i] = *(const uint4*)ptrb;
                m |= ((oka ? 1u : 0u) << i) | ((okb ? 1u : 0u) << (8 + i));
            }
        }
        km = m;
        if (TA) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) p.amap.step(arow[i], GM_BK);
        }
    };
    auto store_tile = [&](int buf, const uint4 (&ra)[NCH], const uint4 (&rb)[NCH], const uint32_t km) {
        bf16_t* sA = lds + buf * (A_ELEMS + B_ELEMS);
        bf16_t* sB = sA + A_ELEMS;
        if (__builtin_amdgcn_ballot_w64(km != KM_FULL) == 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + 256 * i;
                if (!TA) *(uint4*)(sA + row_img_off(c >> 3, c & 7)) = ra[i];
                else *(uint4*)(sA + tr_img_off<NC>(c / NC, c % NC)) = ra[i];
                if (TB) *(uint4*)(sB + row_img_off(c >> 3, c & 7)) = rb[i];
                else *(uint4*)(sB + tr_img_off<NC>(c / NC, c % NC)) = rb[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = tid + 256 * i;
                const uint32_t ma = ((km >> i) & 1u) ? 0xffffffffu : 0u;
                const uint32_t mb = ((km >> (8 + i)) & 1u) ? 0xffffffffu : 0u;
                const uint4 va = make_uint4(ra[i].x & ma, ra[i].y & ma, ra[i].z & ma, ra[i].w & ma);
                const uint4 vb = make_uint4(rb[i].x & mb, rb[i].y & mb, rb[i].z & mb, rb[i].w & mb);
                if (!TA) *(uint4*)(sA + row_img_off(c >> 3, c & 7)) = va;
                else *(uint4*)(sA + tr_img_off<NC>(c / NC, c % NC)) = va;
                if (TB) *(uint4*)(sB + row_img_off(c >> 3, c & 7)) = vb;
                else *(uint4*)(sB + tr_img_off<NC>(c / NC, c % NC)) = vb;
            }
        }
    };
    auto multiply_tile = [&](int buf) {
        const bf16_t* sA = lds + buf * (A_ELEMS + B_ELEMS);
        const bf16_t* sB = sA + A_ELEMS;
        const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < GM_BK; ks += 32) {
            bf16x8_t af[WT], bfr[WT];
#pragma unroll
            for (int i = 0; i < WT; ++i) {
                const int m0 = wm * 16 * WT + i * 16;
                af[i] = TA ? lds_frag_tr<NC>(sA, m0, ks, lane) : lds_frag_rows(sA, m0 + fr, ks + fk);
            }
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                const int n0 = wn * 16 * WT + j * 16;
                bfr[j] = TB ? lds_frag_rows(sB, n0 + fr, ks + fk) : lds_frag_tr<NC>(sB, n0, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            if (!TB && want_bsum) {
#pragma unroll
                for (int j = 0; j < WT; ++j)
                    bs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], ones, bs[j], 0, 0, 0);
            }
        }
    };

    // Large tiles with a transposed operand need too many address registers to also afford the
    // second staging set (two workgroups per CU leave 256 registers per lane): they prefetch one
    // tile ahead, everything else two.
    constexpr bool PF2 = (WT == 2) || (!TA && TB);
    if (t_begin < t_end) {
        if constexpr (PF2) {
            // tiles are taken in pairs; a tile index past t_end loads as zeros and multiplies to
            // nothing, which keeps the loop body free of exits (one accumulator set, fixed register roles)
            load_tile(ra0, rb0, km0);
            load_tile(ra1, rb1, km1);
            store_tile(0, ra0, rb0, km0);
            __syncthreads();
#pragma nounroll
            for (int t = t_begin; t < t_end; t += 2) {
                // LDS stage 0 holds tile t, set 1 holds tile t+1
                load_tile(ra0, rb0, km0);
                multiply_tile(0);
                store_tile(1, ra1, rb1, km1);
                __syncthreads();
                // LDS stage 1 holds tile t+1, set 0 holds tile t+2
                load_tile(ra1, rb1, km1);
                multiply_tile(1);
                store_tile(0, ra0, rb0, km0);
                __syncthreads();
            }
        } else {
            load_tile(ra0, rb0, km0);
            store_tile(0, ra0, rb0, km0);
            __syncthreads();
#pragma nounroll
            for (int t = t_begin; t < t_end; t += 2) {
                load_tile(ra0, rb0, km0);
                multiply_tile(0);
                store_tile(1, ra0, rb0, km0);
                __syncthreads();
                load_tile(ra0, rb0, km0);
                multiply_tile(1);
                store_tile(0, ra0, rb0, km0);
                __syncthreads();
            }
        }
    }
    if (!TB && want_bsum && (lane & 15) == 0) {
        // lane 16 g holds columns n0 + 4 g + r of each 16-column tile
#pragma unroll
        for (int j = 0; j < WT; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = bn + wn * 16 * WT + j * 16 + 4 * (lane >> 4) + r;
                if (n < p.N) {
                    const float t = bs[j][r];
                    if (p.epi.atomic == 2) p.epi.partial[((long)bz * (p.M + 1) + p.M) * p.epi.ldp + n] = t;
                    else if (p.epi.atomic == 1) atomicAdd(p.epi.bsum + n, t);
                    else p.epi.bsum[n] = t;
                }
            }
        }
    }

    // epilogue.  Element r of acc[i][j] is C[m = m0 + (lane & 15)][n = n0 + 4*(lane >> 4) + r].  The
    // tile is first parked in LDS (fp32, 16-byte chunks XOR-swizzled by row); then every thread
    // walks one 8-column piece down the rows of the tile, so bias and column statistics stay in
    // registers, C is written with 16-byte accesses along rows, and everything that does not
    // depend on the element (activation kind, dropout on/off, output type) is decided once.
    float* sC = (float*)lds;                      // BT*BT floats fit exactly in the two stages
    {
        const int fq = lane >> 4, fr = lane & 15;
#pragma unroll
        for (int i = 0; i < WT; ++i) {
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                const int row = wm * 16 * WT + i * 16 + fr;
                const int chunk = (wn * 16 * WT + j * 16) / 4 + fq;
                *(f32x4_t*)(sC + row * BT + ((chunk ^ (row & 7)) * 4)) = acc[i][j];
            }
        }
    }
    __syncthreads();
    constexpr int CPR = BT / 8;                   // 8-column pieces per tile row
    constexpr int RSTEP = 256 / CPR;              // rows between two pieces of a thread
    const int c8 = tid % CPR, row0 = tid / CPR;
    const int n = bn + c8 * 8;
    const int nv = min(8, p.N - n);               // <= 0: this thread's columns are outside the matrix
    const Epilogue& e = p.epi;
    auto read_piece = [&](int row, float (&v)[8]) {
        const f32x4_t x0 = *(const f32x4_t*)(sC + row * BT + (((2 * c8) ^ (row & 7)) * 4));
        const f32x4_t x1 = *(const f32x4_t*)(sC + row * BT + (((2 * c8 + 1) ^ (row & 7)) * 4));
        v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3];
        v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
    };
    if (e.atomic == 2) {
        // raw partial sums; ldp is a multiple of 8 and columns past N hold zeros
        if (nv > 0) {
            for (int row = row0; row < BT && bm + row < p.M; row += RSTEP) {
                float v[8];
                read_piece(row, v);
                float* c = e.partial + ((long)bz * (p.M + 1) + out_row(p, bm + row)) * e.ldp + n;
                *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
            }
        }
        return;
    }
    if (e.atomic == 1) {
        if (nv > 0) {
            for (int row = row0; row < BT && bm + row < p.M; row += RSTEP) {
                float v[8];
                read_piece(row, v);
                float* c = (float*)e.C + coff + (long)out_row(p, bm + row) * e.ldc + n;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    if (r < nv) atomicAdd(c + r, v[r] * e.alpha);
            }
        }
        return;
    }
    float csum[8], csq[8], b8[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        csum[r] = 0.f;
        csq[r] = 0.f;
        b8[r] = (e.bias && r < nv) ? e.bias[n + r] : 0.f;
    }
    const bool bf_out = e.c_dtype == ATHENA_BF16;
    const bool vec8 = nv == 8 && (e.ldc % 8 == 0) && ((((uintptr_t)e.C) & 15) == 0) &&
                      (!e.C_pre || (((uintptr_t)e.C_pre) & 15) == 0);
    const bool aux_vec = nv == 8 && e.aux && (e.ldaux % 8 == 0) && ((((uintptr_t)e.aux) & 15) == 0);
    uint64_t key = 0;
    if (e.drop_thresh) key = rng_key(e.rng_state, e.rng_site);
    if (nv > 0) {
        for (int row = row0; row < BT && bm + row < p.M; row += RSTEP) {
            float v[8], pre[8];
            read_piece(row, v);
            const int mo = out_row(p, bm + row);
            const long ci = (long)mo * e.ldc + n;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                v[r] = v[r] * e.alpha + b8[r];
                pre[r] = v[r];
            }
            if (e.act == ATHENA_ACT_RELU6) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = relu6f(v[r]);
            } else if (e.act == ATHENA_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = gelu_tanh_f(v[r]);
            } else if (e.act >= ATHENA_ACT_UNARY && e.act < ATHENA_ACT_MUL_DUNARY) {
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] = unary_f(e.act - ATHENA_ACT_UNARY, v[r]);
            } else if (e.act == ATHENA_ACT_MUL_DGELU || e.act == ATHENA_ACT_MUL_DRELU6 || e.act >= ATHENA_ACT_MUL_DUNARY) {
                float a[8];
                const long ai = (long)mo * e.ldaux + n;
                if (aux_vec && e.aux_dtype == ATHENA_BF16) {
                    const bf16x8_t h = *(const bf16x8_t*)((const bf16_t*)e.aux + ai);
#pragma unroll
                    for (int r = 0; r < 8; ++r) a[r] = (float)h[r];
                } else if (aux_vec) {
                    const float4 h0 = *(const float4*)((const float*)e.aux + ai);
                    const float4 h1 = *(const float4*)((const float*)e.aux + ai + 4);
                    a[0] = h0.x; a[1] = h0.y; a[2] = h0.z; a[3] = h0.w;
                    a[4] = h1.x; a[5] = h1.y; a[6] = h1.z; a[7] = h1.w;
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) a[r] = r < nv ? load_as_f32(e.aux, ai + r, e.aux_dtype) : 0.f;
                }
                if (e.act == ATHENA_ACT_MUL_DGELU) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] *= dgelu_tanh_f(a[r]);
                } else if (e.act >= ATHENA_ACT_MUL_DUNARY) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] *= dunary_f(e.act - ATHENA_ACT_MUL_DUNARY, a[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = (a[r] > 0.f && a[r] < 6.f) ? v[r] : 0.f;
                }
            }
            if (e.drop_thresh) {
                float dm[8];
                dropout_mult8(key, (uint64_t)ci, e.drop_thresh, e.drop_scale, dm);
#pragma unroll
                for (int r = 0; r < 8; ++r) v[r] *= dm[r];
            }
            if (e.colsum) {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    float sv = v[r];
                    if (bf_out) sv = (float)(bf16_t)sv;
                    if (r < nv) { csum[r] += sv; csq[r] += sv * sv; }
                }
            }
            if (vec8) {
                if (bf_out) {
                    bf16_t* c = (bf16_t*)e.C + ci;
                    if (e.C_pre) {
                        bf16x8_t pv;
#pragma unroll
                        for (int r = 0; r < 8; ++r) pv[r] = (bf16_t)pre[r];
                        *(bf16x8_t*)((bf16_t*)e.C_pre + ci) = pv;
                    }
                    if (e.accum) {
                        const bf16x8_t old = *(const bf16x8_t*)c;
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] += (float)old[r];
                    }
                    bf16x8_t o;
#pragma unroll
                    for (int r = 0; r < 8; ++r) o[r] = (bf16_t)v[r];
                    *(bf16x8_t*)c = o;
                } else {
                    float* c = (float*)e.C + ci;
                    if (e.C_pre) {
                        float* cp = (float*)e.C_pre + ci;
                        *(float4*)cp = make_float4(pre[0], pre[1], pre[2], pre[3]);
                        *(float4*)(cp + 4) = make_float4(pre[4], pre[5], pre[6], pre[7]);
                    }
                    if (e.accum) {
                        const float4 o0 = *(const float4*)c, o1 = *(const float4*)(c + 4);
                        v[0] += o0.x; v[1] += o0.y; v[2] += o0.z; v[3] += o0.w;
                        v[4] += o1.x; v[5] += o1.y; v[6] += o1.z; v[7] += o1.w;
                    }
                    *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                    *(float4*)(c + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r < nv) {
                        if (e.C_pre) store_from_f32(e.C_pre, ci + r, e.c_dtype, pre[r]);
                        float o = v[r];
                        if (e.accum) o += load_as_f32(e.C, ci + r, e.c_dtype);
                        store_from_f32(e.C, ci + r, e.c_dtype, o);
                    }
                }
            }
        }
    }
    if (e.colsum) {
        // threads with equal (tid % CPR) own the same 8 columns: fold inside the wave, then across
        // the four waves through LDS, so a block issues one fp64 atomic per column and statistic
        __syncthreads();                              // every thread is done reading sC
        float* red = (float*)lds;                     // [4 waves][BT columns][2]
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float sm = csum[r], q = csq[r];
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) { sm += __shfl_xor(sm, o, 64); q += __shfl_xor(q, o, 64); }
            if (lane < CPR) {
                red[(wave * BT + lane * 8 + r) * 2] = sm;
                red[(wave * BT + lane * 8 + r) * 2 + 1] = q;
            }
        }
        __syncthreads();
        if (tid < 2 * BT) {
            const int col = tid >> 1, which = tid & 1;
            const float t = red[(0 * BT + col) * 2 + which] + red[(1 * BT + col) * 2 + which] +
                            red[(2 * BT + col) * 2 + which] + red[(3 * BT + col) * 2 + which];
            if (bn + col < p.N) atomicAdd(e.colsum + which * p.N + bn + col, (double)t);
        }
    }
}

template <class AMap, bool TA, bool TB, int WT>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(GemmProblem<AMap> p) {
    int bx = blockIdx.x, by = blockIdx.y;
    if (p.xcd_tn > 0) {
        const int j = blockIdx.x >> 3;
        by = j % p.xcd_tn;
        bx = (j / p.xcd_tn) * 8 + (blockIdx.x & 7);
        if (bx >= p.xcd_tm) return;                 // whole workgroup: the grid is rounded up to 8 row tiles
    }
    gemm_mfma_body<AMap, TA, TB, WT>(p, bx, by, blockIdx.z);
}

// Several independent problems (same operand layouts) in one launch: the workgroups of all
// problems are numbered consecutively along grid.x; first[i] is the first workgroup of problem i.
#define GEMM_GROUP_MAX 12
struct GemmGroup {
    int count;
    int first[GEMM_GROUP_MAX + 1];
    int tiles_m[GEMM_GROUP_MAX], tiles_n[GEMM_GROUP_MAX];
    GemmProblem<PlainRows> p[GEMM_GROUP_MAX];
};

template <bool TA, bool TB, int WT>
__global__ __launch_bounds__(256, 2) void gemm_mfma_grouped_kernel(GemmGroup g) {
    int pi = 0;
    while (pi + 1 < g.count && (int)blockIdx.x >= g.first[pi + 1]) ++pi;
    const int local = blockIdx.x - g.first[pi];
    const int tm = g.tiles_m[pi], tn = g.tiles_n[pi];
    const int bx = local % tm, r = local / tm;
    gemm_mfma_body<PlainRows, TA, TB, WT>(g.p[pi], bx, r % tn, r / tn);
}

// ------------------------------------------------------------------ VALU core (fp32)
#define GV_BM 64
#define GV_BN 64
#define GV_BK 16

template <class AMap, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_valu_kernel(GemmProblem<AMap> p) {
    const int zb = p.zbatch > 1 ? (int)(blockIdx.z / p.split_k) : 0;
    const int zsplit = p.zbatch > 1 ? (int)(blockIdx.z % p.split_k) : (int)blockIdx.z;
    const int seg_base = p.seg_base + zb;
    const long coff = (long)zb * p.c_batch_stride;
    __shared__ float sA[GV_BK][GV_BM + 4];
    __shared__ float sB[GV_BK][GV_BN + 4];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int bm = blockIdx.x * GV_BM, bn = blockIdx.y * GV_BN;
    const float* __restrict__ A = (const float*)p.A;
    const float* __restrict__ B = (const float*)p.B;

    const int tiles_per_seg = (p.kseg + GV_BK - 1) / GV_BK;
    const int ntiles = p.nseg * tiles_per_seg;
    const int per_split = (ntiles + p.split_k - 1) / p.split_k;
    const int t_begin = zsplit * per_split;
    const int t_end = min(ntiles, t_begin + per_split);

    // fp32 running sums are emptied into fp64 totals every GV_FLUSH k-tiles: the exact-arithmetic
    // path also serves contractions over 10^5 rows (weight gradients), where one long fp32 chain
    // would lose four digits against the fp64 oracle
    constexpr int GV_FLUSH = 8;
    float acc[4][4];
    double tot[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[i][j] = 0.f; tot[i][j] = 0.0; }

    for (int t = t_begin; t < t_end; ++t) {
        if (((t - t_begin) % GV_FLUSH) == GV_FLUSH - 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { tot[i][j] += (double)acc[i][j]; acc[i][j] = 0.f; }
        }
        const int seg = t / tiles_per_seg;
        const int k0 = (t - seg * tiles_per_seg) * GV_BK;
        // 64x16 elements per operand, 4 per thread
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + 256 * i;
            {
                int mm, kk;
                if (!TA) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }
                const int m = bm + mm, k = k0 + kk;
                float v = 0.f;
                if (m < p.M && k < p.kseg) {
                    if (!TA) {
                        long o = p.amap.off(p.amap.prep(m), seg_base + seg);
                        if (o >= 0) v = A[o + k];
                    } else {
                        long o = p.amap.off(p.amap.prep(k), seg_base + seg);
                        if (o >= 0) v = A[o + m];
                    }
                }
                sA[kk][mm] = v;
            }
            {
                int nn, kk;
                if (TB) { kk = e & 15; nn = e >> 4; } else { nn = e & 63; kk = e >> 6; }
                const int n = bn + nn, k = k0 + kk;
                float v = 0.f;
                if (n < p.N && k < p.kseg) {
                    const long base = b_seg_off(p, seg);
                    v = TB ? B[base + (long)n * p.ldb + k] : B[base + (long)k * p.ldb + n];
                }
                sB[kk][nn] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GV_BK; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = sA[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = sB[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float)(tot[i][j] + (double)acc[i][j]);
    uint64_t key = 0;
    if (p.epi.drop_thresh) key = rng_key(p.epi.rng_state, p.epi.rng_site);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = bm + ty * 4 + i;
        if (m >= p.M) continue;
        const int mo = out_row(p, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = bn + tx * 4 + j;
            if (n < p.N) {
                if (p.epi.atomic == 2) {
                    p.epi.partial[((long)blockIdx.z * (p.M + 1) + mo) * p.epi.ldp + n] = acc[i][j];
                    continue;
                }
                epilogue_store(p.epi, mo, n, acc[i][j], key, coff);
                if (p.epi.colsum) {
                    float v = acc[i][j] * p.epi.alpha;
                    atomicAdd(p.epi.colsum + n, (double)v);
                    atomicAdd(p.epi.colsum + p.N + n, (double)v * (double)v);
                }
            }
        }
    }
}

// Second pass of a split-K product whose workgroups wrote raw partial sums: adds the ``split``
// slabs of each batch in a fixed order (bit-reproducible, no atomics, no zero-fill of C) and
// writes C = alpha * sum (fp32, or bf16 with ``c_bf16``).  Row M of a slab carries the column sums of
// B and goes to ``bsum``.
static __global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int split, int M, int N,
                                                            int ldp, float* __restrict__ C, int ldc,
                                                            long c_batch_stride, float alpha,
                                                            float* __restrict__ bsum, int c_bf16) {
    // workgroup = 64 column quads x 4 waves; wave w adds slabs w, w + 4, ... and the four partial
    // sums meet in LDS (fixed order: the result does not depend on scheduling)
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = ldp >> 2;
    const long idx = (long)blockIdx.x * 64 + lane;
    const int row = (int)(idx / q), n = (int)(idx % q) * 4;
    const bool live = row <= M && n < N && (row < M || bsum);
    const long slab = (long)(M + 1) * ldp;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (live) {
        const float* src = partial + (long)blockIdx.y * split * slab + (long)row * ldp + n;
        int i = w;
        for (; i + 4 < split; i += 8) {
            const float4 a = *(const float4*)(src + (long)i * slab);
            const float4 b = *(const float4*)(src + (long)(i + 4) * slab);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (i < split) {
            const float4 a = *(const float4*)(src + (long)i * slab);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[w][lane] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (w != 0 || !live) return;
    const float4 r0 = red[0][lane], r1 = red[1][lane], r2 = red[2][lane], r3 = red[3][lane];
    const float v[4] = {(r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z),
                        (r0.w + r1.w) + (r2.w + r3.w)};
    if (row == M) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < N) bsum[n + r] = v[r];
        return;
    }
    if (c_bf16) {
        bf16_t* c = (bf16_t*)C + (long)blockIdx.y * c_batch_stride + (long)row * ldc + n;
        if (n + 3 < N && (ldc & 3) == 0 && ((((uintptr_t)C) & 7) == 0) && (c_batch_stride & 3) == 0) {
            const bf16x4_t t = {(bf16_t)(v[0] * alpha), (bf16_t)(v[1] * alpha), (bf16_t)(v[2] * alpha), (bf16_t)(v[3] * alpha)};
            *(bf16x4_t*)c = t;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < N) c[r] = (bf16_t)(v[r] * alpha);
        }
        return;
    }
    float* c = C + (long)blockIdx.y * c_batch_stride + (long)row * ldc + n;
    if (n + 3 < N && (ldc & 3) == 0 && ((((uintptr_t)C) & 15) == 0) && (c_batch_stride & 3) == 0) {
        *(float4*)c = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < N) c[r] = v[r] * alpha;
    }
}

static inline size_t splitk_workspace_bytes(int M, int N, int split, int zbatch) {
    const size_t ldp = (size_t)(N + 7) / 8 * 8;
    return sizeof(float) * (size_t)(zbatch > 1 ? zbatch : 1) * split * (size_t)(M + 1) * ldp;
}

struct ReduceGroup {
    int count;
    int first[GEMM_GROUP_MAX + 1];
    const float* partial[GEMM_GROUP_MAX];
    float* C[GEMM_GROUP_MAX];
    float* bsum[GEMM_GROUP_MAX];
    int split[GEMM_GROUP_MAX], M[GEMM_GROUP_MAX], N[GEMM_GROUP_MAX], ldp[GEMM_GROUP_MAX], ldc[GEMM_GROUP_MAX];
    float alpha[GEMM_GROUP_MAX];
};

// splitk_reduce_kernel for the problems of a grouped launch
static __global__ __launch_bounds__(256) void splitk_reduce_grouped_kernel(ReduceGroup g) {
    __shared__ float4 red[4][64];
    int pi = 0;
    while (pi + 1 < g.count && (int)blockIdx.x >= g.first[pi + 1]) ++pi;
    const float* __restrict__ partial = g.partial[pi];
    const int split = g.split[pi], M = g.M[pi], N = g.N[pi], ldp = g.ldp[pi], ldc = g.ldc[pi];
    float* __restrict__ C = g.C[pi];
    float* __restrict__ bsum = g.bsum[pi];
    const float alpha = g.alpha[pi];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int q = ldp >> 2;
    const long idx = (long)(blockIdx.x - g.first[pi]) * 64 + lane;
    const int row = (int)(idx / q), n = (int)(idx % q) * 4;
    const bool live = row <= M && n < N && (row < M || bsum);
    const long slab = (long)(M + 1) * ldp;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (live) {
        const float* src = partial + (long)row * ldp + n;
        int i = w;
        for (; i + 4 < split; i += 8) {
            const float4 a = *(const float4*)(src + (long)i * slab);
            const float4 b = *(const float4*)(src + (long)(i + 4) * slab);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (i < split) {
            const float4 a = *(const float4*)(src + (long)i * slab);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[w][lane] = make_float4(s0.x + s1.x, s0.y + s1.y, s0.z + s1.z, s0.w + s1.w);
    __syncthreads();
    if (w != 0 || !live) return;
    const float4 r0 = red[0][lane], r1 = red[1][lane], r2 = red[2][lane], r3 = red[3][lane];
    const float v[4] = {(r0.x + r1.x) + (r2.x + r3.x), (r0.y + r1.y) + (r2.y + r3.y), (r0.z + r1.z) + (r2.z + r3.z),
                        (r0.w + r1.w) + (r2.w + r3.w)};
    if (row == M) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < N) bsum[n + r] = v[r];
        return;
    }
    float* c = C + (long)row * ldc + n;
    if (n + 3 < N && (ldc & 3) == 0 && ((((uintptr_t)C) & 15) == 0)) {
        *(float4*)c = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < N) c[r] = v[r] * alpha;
    }
}

// number of workgroups a launch would use with tile edge ``bt``
static inline long gemm_tiles(int M, int N, int bt) { return (long)((M + bt - 1) / bt) * ((N + bt - 1) / bt); }

template <class AMap, int WT>
static void launch_mfma(const GemmProblem<AMap>& p, bool ta, bool tb, hipStream_t stream) {
    constexpr int BT = 32 * WT;
    const int tm = (p.M + BT - 1) / BT, tn = (p.N + BT - 1) / BT;
    const int gz = p.split_k * (p.zbatch > 1 ? p.zbatch : 1);
    dim3 grid(tm, tn, gz);
    GemmProblem<AMap> q = p;
    q.xcd_tn = 0;
    q.xcd_tm = tm;
    if (gz == 1 && tn >= 2 && tn <= 16 && tm >= 32) {
        // tall output with a few column tiles: the A operand dominates the traffic
        q.xcd_tn = tn;
        grid = dim3((unsigned)((tm + 7) / 8 * 8 * tn), 1, 1);
    }
    if (!ta && tb) ATHENA_LAUNCH((gemm_mfma_kernel<AMap, false, true, WT>), grid, dim3(256), 0, stream, q);
    else if (!ta && !tb) ATHENA_LAUNCH((gemm_mfma_kernel<AMap, false, false, WT>), grid, dim3(256), 0, stream, q);
    else if (ta && !tb) ATHENA_LAUNCH((gemm_mfma_kernel<AMap, true, false, WT>), grid, dim3(256), 0, stream, q);
    else ATHENA_LAUNCH((gemm_mfma_kernel<AMap, true, true, WT>), grid, dim3(256), 0, stream, q);
}

// Tile edge and split-K factor for an M x N output with ``k_elems`` contraction elements and
// ``zbatch`` independent problems.  128x128 tiles do twice the MFMA work per LDS byte of 64x64
// tiles, so they are used whenever they still fill the chip (~256 CUs, two workgroups each) --
// on their own, or together with a K split that leaves each workgroup at least 8 k-tiles.
struct GemmPlan { int bt, split; };
#define GM_WG_SLOTS 512      // 256 CUs x 2 resident workgroups of the 128x128-tile kernels

static inline GemmPlan gemm_plan(int M, int N, long k_elems, int in_dtype, int zbatch, bool may_split) {
    GemmPlan pl;
    if (zbatch < 1) zbatch = 1;
    if (in_dtype != ATHENA_BF16) {
        const long tiles = gemm_tiles(M, N, GV_BM) * zbatch;
        const long ktiles = (k_elems + GV_BK - 1) / GV_BK;
        long split = may_split ? 512 / (tiles > 0 ? tiles : 1) : 1;
        if (split > ktiles / 4) split = ktiles / 4;
        if (split < 1) split = 1;
        if (split > 256) split = 256;
        pl.bt = GV_BM;
        pl.split = (int)split;
        return pl;
    }
    const long ktiles = (k_elems + GM_BK - 1) / GM_BK;
    const long t128 = gemm_tiles(M, N, 128) * zbatch;
    // 128x128 tiles run two workgroups per CU: tiles x splits must not exceed those 512 places, or
    // the few workgroups beyond them run as a second round behind the first (conv2 weight gradient:
    // 540 workgroups 563 us, 504 workgroups 350 us)
    long s128 = may_split ? GM_WG_SLOTS / t128 : 1;
    if (s128 > ktiles / 8) s128 = ktiles / 8;
    if (s128 < 1) s128 = 1;
    if (M > 64 && N > 64 && (t128 >= 192 || t128 * s128 >= 256)) {
        pl.bt = 128;
        pl.split = t128 >= 384 ? 1 : (int)s128;
        return pl;
    }
    const long t64 = gemm_tiles(M, N, 64) * zbatch;
    long s64 = may_split ? GM_WG_SLOTS / (t64 > 0 ? t64 : 1) : 1;
    if (s64 > ktiles / 4) s64 = ktiles / 4;
    if (s64 < 1) s64 = 1;
    if (s64 > 256) s64 = 256;
    pl.bt = 64;
    pl.split = (int)s64;
    return pl;
}

// ``bt``: tile edge chosen by gemm_plan (bf16 path)
template <class AMap>
static int launch_gemm(const GemmProblem<AMap>& p, int in_dtype, bool ta, bool tb, int bt, hipStream_t stream) {
    if (in_dtype == ATHENA_BF16) {
        if (bt == 128) launch_mfma<AMap, 4>(p, ta, tb, stream);
        else launch_mfma<AMap, 2>(p, ta, tb, stream);
    } else {
        dim3 grid((p.M + GV_BM - 1) / GV_BM, (p.N + GV_BN - 1) / GV_BN, p.split_k * (p.zbatch > 1 ? p.zbatch : 1));
        if (!ta && tb) ATHENA_LAUNCH((gemm_valu_kernel<AMap, false, true>), grid, dim3(256), 0, stream, p);
        else if (!ta && !tb) ATHENA_LAUNCH((gemm_valu_kernel<AMap, false, false>), grid, dim3(256), 0, stream, p);
        else if (ta && !tb) ATHENA_LAUNCH((gemm_valu_kernel<AMap, true, false>), grid, dim3(256), 0, stream, p);
        else ATHENA_LAUNCH((gemm_valu_kernel<AMap, true, true>), grid, dim3(256), 0, stream, p);
    }
    ATHENA_CHECK_LAUNCH("gemm");
    return ATHENA_OK;
}
