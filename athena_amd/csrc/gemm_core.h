// GEMM cores shared by the Dense and Conv2D entry points.
//
//   C[m, n] = epilogue( alpha * sum_k A(m, k) * B(k, n) )
//
// Two cores with the same problem description:
//   * gemm_mfma_kernel  : bf16 operands, fp32 accumulate on the matrix cores
//                         (v_mfma_f32_16x16x32_bf16), 128x128x32 tiles staged through LDS,
//                         4 waves of 64 lanes, each owning a 64x64 sub-tile (4x4 MFMA tiles).
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
    int ldc, ldaux;
    int c_dtype, aux_dtype;
    int act;
    int atomic;
    int accum;
    float alpha;
    float drop_scale;
    uint32_t drop_thresh;
    uint32_t rng_site;
};

__device__ __forceinline__ void epilogue_store(const Epilogue& e, int m, int n, float acc, uint64_t key) {
    float v = acc * e.alpha;
    const long ci = (long)m * e.ldc + n;
    if (e.atomic) {
        atomicAdd((float*)e.C + ci, v);
        return;
    }
    if (e.bias) v += e.bias[n];
    if (e.C_pre) store_from_f32(e.C_pre, ci, e.c_dtype, v);
    switch (e.act) {
        case ATHENA_ACT_RELU6: v = relu6f(v); break;
        case ATHENA_ACT_GELU: v = gelu_tanh_f(v); break;
        case ATHENA_ACT_MUL_DGELU: {
            float h = load_as_f32(e.aux, (long)m * e.ldaux + n, e.aux_dtype);
            v *= dgelu_tanh_f(h);
        } break;
        case ATHENA_ACT_MUL_DRELU6: {
            float y = load_as_f32(e.aux, (long)m * e.ldaux + n, e.aux_dtype);
            v = (y > 0.f && y < 6.f) ? v : 0.f;
        } break;
        default: break;
    }
    if (e.drop_thresh) v *= dropout_mult(key, (uint64_t)ci, e.drop_thresh, e.drop_scale);
    if (e.accum) v += load_as_f32(e.C, ci, e.c_dtype);
    store_from_f32(e.C, ci, e.c_dtype, v);
}

// ------------------------------------------------------------------ row maps for the A operand
// off(r, seg) = element offset of row r of segment seg, or -1 when the row is all zeros.
struct PlainRows {
    long ld;
    long seg_stride;
    __device__ __forceinline__ long off(int r, int seg) const { return (long)r * ld + (long)seg * seg_stride; }
};

// Rows of the im2col matrix of a 3x3 / stride-2 / "same" convolution over an NHWC tensor
// [B, H, W, C].  Row r = (b, ho, wo); segment = tap (kh*3+kw); the row is the C-vector at
// input pixel (2*ho - pad_h + kh, 2*wo - pad_w + kw), zero outside the image.
struct ConvFwdRows {
    int H, W, C, Ho, Wo, pad_h, pad_w;
    __device__ __forceinline__ long off(int r, int seg) const {
        int wo = r % Wo;
        int t = r / Wo;
        int ho = t % Ho;
        int b = t / Ho;
        int kh = seg / 3, kw = seg - kh * 3;
        int h = 2 * ho - pad_h + kh, w = 2 * wo - pad_w + kw;
        if (h < 0 || h >= H || w < 0 || w >= W) return -1;
        return (((long)b * H + h) * W + w) * C;
    }
};

// Rows for the input gradient of the same convolution, one launch per parity class of the
// input pixel.  Row r = (b, ah, aw) <-> input pixel (2*ah + ph - pad_h, 2*aw + pw - pad_w)
// where (ph, pw) is the class; segment s indexes the taps that reach this class; the row is
// the Cout-vector of the output gradient at (ah + dh[s], aw + dw[s]).
struct ConvDgradRows {
    int Ho, Wo, Co, Ah, Aw;     // Ah x Aw = pixels of this class per image
    int dh[4], dw[4];
    __device__ __forceinline__ long off(int r, int seg) const {
        int aw = r % Aw;
        int t = r / Aw;
        int ah = t % Ah;
        int b = t / Ah;
        int ho = ah + dh[seg], wo = aw + dw[seg];
        if (ho < 0 || ho >= Ho || wo < 0 || wo >= Wo) return -1;
        return (((long)b * Ho + ho) * Wo + wo) * Co;
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
    long b_seg_off[9];       // element offset of B for each segment
    int seg_base;            // segment index passed to the row map = seg_base + local segment
    int split_k;             // number of K splits (grid.z); splits are over k-tiles
    // when TA: the output rows m index the contiguous dimension of A and the contraction index
    // runs over A's (mapped) rows; ``rows_a`` = number of contraction rows per segment.
    Epilogue epi;
    // optional output row remap for strided scatter of C rows (ConvDgrad): row m -> c_row(m)
    int crow_mode;           // 0: identity; 1: parity-class scatter
    int cr_Ah, cr_Aw, cr_H, cr_W, cr_h0, cr_w0;   // crow = ((b*H + 2*ah+h0)*W + 2*aw+w0)
};

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
#define GM_BK 32
#define GM_LDS_STRIDE 40   // bf16 elements per LDS row: 32 + 8 pad (80 B, 16-B aligned)

// Workgroup = 4 waves as 2 (M) x 2 (N); each wave owns WT x WT MFMA tiles of 16x16, so the
// workgroup tile is (32*WT) x (32*WT): WT = 4 -> 128x128 (large outputs), WT = 2 -> 64x64
// (small outputs, more workgroups in flight).
// TA = false: A(m, k) = A[amap.off(m, seg) + kk]                (contraction contiguous)
// TA = true : A(m, k) = A[amap.off(kk_row, seg) + m]            (output row contiguous)
// TB = true : B(k, n) = B[b_seg_off[seg] + n*ldb + kk]          (contraction contiguous, "[N][K]")
// TB = false: B(k, n) = B[b_seg_off[seg] + kk*ldb + n]          ("[K][N]")
template <class AMap, bool TA, bool TB, int WT>
__global__ __launch_bounds__(256) void gemm_mfma_kernel(GemmProblem<AMap> p) {
    constexpr int BT = 32 * WT;              // tile edge
    constexpr int NCH = BT * 4 / 256;        // 16-byte pieces per thread per operand tile
    __shared__ __attribute__((aligned(16))) bf16_t sA[BT * GM_LDS_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16_t sB[BT * GM_LDS_STRIDE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.x * BT;
    const int bn = blockIdx.y * BT;
    const bf16_t* __restrict__ A = (const bf16_t*)p.A;
    const bf16_t* __restrict__ B = (const bf16_t*)p.B;

    const int tiles_per_seg = (p.kseg + GM_BK - 1) / GM_BK;
    const int ntiles = p.nseg * tiles_per_seg;
    const int per_split = (ntiles + p.split_k - 1) / p.split_k;
    const int t_begin = blockIdx.z * per_split;
    const int t_end = min(ntiles, t_begin + per_split);

    f32x4_t acc[WT][WT];
#pragma unroll
    for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    uint4 ra[NCH], rb[NCH];
    const uint4 zero4 = make_uint4(0, 0, 0, 0);

    auto load_tile = [&](int t) {
        const int seg = t / tiles_per_seg;
        const int k0 = (t - seg * tiles_per_seg) * GM_BK;
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i;
            if (!TA) {
                const int row = c >> 2, kc = (c & 3) * 8;
                const int m = bm + row, k = k0 + kc;
                uint4 v = zero4;
                if (m < p.M && k < p.kseg) {
                    long o = p.amap.off(m, p.seg_base + seg);
                    if (o >= 0) v = *(const uint4*)(A + o + k);
                }
                ra[i] = v;
            } else {
                const int kk = c / (BT / 8), mc = (c % (BT / 8)) * 8;
                const int m = bm + mc, k = k0 + kk;
                uint4 v = zero4;
                if (m < p.M && k < p.kseg) {
                    long o = p.amap.off(k, p.seg_base + seg);
                    if (o >= 0) v = *(const uint4*)(A + o + m);
                }
                ra[i] = v;
            }
            if (TB) {
                const int row = c >> 2, kc = (c & 3) * 8;
                const int n = bn + row, k = k0 + kc;
                uint4 v = zero4;
                if (n < p.N && k < p.kseg) v = *(const uint4*)(B + p.b_seg_off[seg] + (long)n * p.ldb + k);
                rb[i] = v;
            } else {
                const int kk = c / (BT / 8), nc = (c % (BT / 8)) * 8;
                const int n = bn + nc, k = k0 + kk;
                uint4 v = zero4;
                if (n < p.N && k < p.kseg) v = *(const uint4*)(B + p.b_seg_off[seg] + (long)k * p.ldb + n);
                rb[i] = v;
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int c = tid + 256 * i;
            if (!TA) {
                const int row = c >> 2, kc = (c & 3) * 8;
                *(uint4*)(sA + row * GM_LDS_STRIDE + kc) = ra[i];
            } else {
                const int kk = c / (BT / 8), mc = (c % (BT / 8)) * 8;
                const bf16_t* e = (const bf16_t*)&ra[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) sA[(mc + j) * GM_LDS_STRIDE + kk] = e[j];
            }
            if (TB) {
                const int row = c >> 2, kc = (c & 3) * 8;
                *(uint4*)(sB + row * GM_LDS_STRIDE + kc) = rb[i];
            } else {
                const int kk = c / (BT / 8), nc = (c % (BT / 8)) * 8;
                const bf16_t* e = (const bf16_t*)&rb[i];
#pragma unroll
                for (int j = 0; j < 8; ++j) sB[(nc + j) * GM_LDS_STRIDE + kk] = e[j];
            }
        }
    };

    if (t_begin < t_end) {
        load_tile(t_begin);
        store_tile();
        __syncthreads();
        for (int t = t_begin; t < t_end; ++t) {
            const bool more = (t + 1 < t_end);
            if (more) load_tile(t + 1);
            bf16x8_t af[WT], bfr[WT];
            const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
            for (int i = 0; i < WT; ++i)
                af[i] = *(const bf16x8_t*)(sA + (wm * 16 * WT + i * 16 + fr) * GM_LDS_STRIDE + fk);
#pragma unroll
            for (int j = 0; j < WT; ++j)
                bfr[j] = *(const bf16x8_t*)(sB + (wn * 16 * WT + j * 16 + fr) * GM_LDS_STRIDE + fk);
#pragma unroll
            for (int i = 0; i < WT; ++i)
#pragma unroll
                for (int j = 0; j < WT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            __syncthreads();
            if (more) {
                store_tile();
                __syncthreads();
            }
        }
    }

    // epilogue: accumulator element r of tile (i, j) is C[row = (lane>>4)*4 + r][col = lane&15]
    uint64_t key = 0;
    if (p.epi.drop_thresh) key = rng_key(p.epi.rng_state, p.epi.rng_site);
    const int fq = lane >> 4, fr = lane & 15;
    float csum[WT], csq[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j) { csum[j] = 0.f; csq[j] = 0.f; }
#pragma unroll
    for (int i = 0; i < WT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = bm + wm * 16 * WT + i * 16 + fq * 4 + r;
            if (m >= p.M) continue;
            const int mo = out_row(p, m);
#pragma unroll
            for (int j = 0; j < WT; ++j) {
                const int n = bn + wn * 16 * WT + j * 16 + fr;
                if (n < p.N) {
                    epilogue_store(p.epi, mo, n, acc[i][j][r], key);
                    if (p.epi.colsum) {
                        float v = acc[i][j][r] * p.epi.alpha;
                        if (p.epi.c_dtype == ATHENA_BF16) v = (float)(bf16_t)v;
                        csum[j] += v;
                        csq[j] += v * v;
                    }
                }
            }
        }
    }
    if (p.epi.colsum) {
        // lanes with equal (lane & 15) hold the same column: fold the 4 row groups, then one atomic per column
#pragma unroll
        for (int j = 0; j < WT; ++j) {
            float s = csum[j], q = csq[j];
            s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
            const int n = bn + wn * 16 * WT + j * 16 + fr;
            if (fq == 0 && n < p.N) {
                atomicAdd(p.epi.colsum + n, (double)s);
                atomicAdd(p.epi.colsum + p.N + n, (double)q);
            }
        }
    }
}

// ------------------------------------------------------------------ VALU core (fp32)
#define GV_BM 64
#define GV_BN 64
#define GV_BK 16

template <class AMap, bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_valu_kernel(GemmProblem<AMap> p) {
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
    const int t_begin = blockIdx.z * per_split;
    const int t_end = min(ntiles, t_begin + per_split);

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int t = t_begin; t < t_end; ++t) {
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
                        long o = p.amap.off(m, p.seg_base + seg);
                        if (o >= 0) v = A[o + k];
                    } else {
                        long o = p.amap.off(k, p.seg_base + seg);
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
                    const long base = p.b_seg_off[seg];
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
                epilogue_store(p.epi, mo, n, acc[i][j], key);
                if (p.epi.colsum) {
                    float v = acc[i][j] * p.epi.alpha;
                    atomicAdd(p.epi.colsum + n, (double)v);
                    atomicAdd(p.epi.colsum + p.N + n, (double)v * (double)v);
                }
            }
        }
    }
}

// number of workgroups a launch would use with the large (128) / small (64) tile
static inline long gemm_tiles(int M, int N, int bt) { return (long)((M + bt - 1) / bt) * ((N + bt - 1) / bt); }

template <class AMap, int WT>
static void launch_mfma(const GemmProblem<AMap>& p, bool ta, bool tb, hipStream_t stream) {
    constexpr int BT = 32 * WT;
    dim3 grid((p.M + BT - 1) / BT, (p.N + BT - 1) / BT, p.split_k);
    if (!ta && tb) hipLaunchKernelGGL((gemm_mfma_kernel<AMap, false, true, WT>), grid, dim3(256), 0, stream, p);
    else if (!ta && !tb) hipLaunchKernelGGL((gemm_mfma_kernel<AMap, false, false, WT>), grid, dim3(256), 0, stream, p);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_mfma_kernel<AMap, true, false, WT>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((gemm_mfma_kernel<AMap, true, true, WT>), grid, dim3(256), 0, stream, p);
}

// tile edge the bf16 path will use for an M x N output
static inline int gemm_mfma_tile(int M, int N) { return gemm_tiles(M, N, 128) >= 192 ? 128 : 64; }

template <class AMap>
static int launch_gemm(const GemmProblem<AMap>& p, int in_dtype, bool ta, bool tb, hipStream_t stream) {
    if (in_dtype == ATHENA_BF16) {
        if (gemm_mfma_tile(p.M, p.N) == 128) launch_mfma<AMap, 4>(p, ta, tb, stream);
        else launch_mfma<AMap, 2>(p, ta, tb, stream);
    } else {
        dim3 grid((p.M + GV_BM - 1) / GV_BM, (p.N + GV_BN - 1) / GV_BN, p.split_k);
        if (!ta && tb) hipLaunchKernelGGL((gemm_valu_kernel<AMap, false, true>), grid, dim3(256), 0, stream, p);
        else if (!ta && !tb) hipLaunchKernelGGL((gemm_valu_kernel<AMap, false, false>), grid, dim3(256), 0, stream, p);
        else if (ta && !tb) hipLaunchKernelGGL((gemm_valu_kernel<AMap, true, false>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((gemm_valu_kernel<AMap, true, true>), grid, dim3(256), 0, stream, p);
    }
    ATHENA_CHECK_LAUNCH("gemm");
    return ATHENA_OK;
}

// split-K factor for an output of M x N with ``ktiles`` contraction tiles: enough workgroups to
// fill the chip (~512) without multiplying the atomic traffic more than needed
static inline int gemm_auto_split(int M, int N, long k_elems, int in_dtype) {
    const int bt = in_dtype == ATHENA_BF16 ? gemm_mfma_tile(M, N) : GV_BM;
    const int bk = in_dtype == ATHENA_BF16 ? GM_BK : GV_BK;
    const long tiles = gemm_tiles(M, N, bt);
    const long ktiles = (k_elems + bk - 1) / bk;
    long split = 512 / (tiles > 0 ? tiles : 1);
    if (split > ktiles / 4) split = ktiles / 4;     // at least 4 k-tiles per split
    if (split < 1) split = 1;
    if (split > 256) split = 256;
    return (int)split;
}
