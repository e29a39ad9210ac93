// Products with a short contraction and many output columns: y[rows, 256 c .. 256 c + 255] =
// x[rows, 256] W_c (+ bias_c) for a list of [256 x 256] weight blocks W_c -- the key / value projections of
// the encoder output for all decoder layers (attention.py:150-152 for every layer of transformer.py:346-388,
// twelve blocks), the q / k / v projections of an attention's input (three), the input gradient of a
// Dense layer with 256 outputs (its in_features / 256 blocks).
//
// A tile kernel starts over for every output tile: with a contraction of 256 a 128 x 128 tile is four
// k-steps between a pipeline fill and an epilogue, and a launch of 1500 such tiles runs at a tenth of
// the matrix rate.  Here a workgroup of eight waves owns 16 or 32 complete rows: it lays them into an LDS
// operand image ONCE and then walks over all weight blocks; wave w owns output columns 32 w .. 32 w + 31
// of every block and streams its part of the weights from L2 straight into registers in fragment
// layout (the fragment-tiled copies athena_refresh_shadows writes), each k-block requested for the
// weight block AFTER THE NEXT as soon as it has been multiplied (two register units alternate); the
// rows' own fragments are read from the image once and stay in registers.  Results go through one of two
// LDS images and leave with 16 bytes per lane along rows; one barrier per block.  Biases are staged in LDS by the prologue: loads return in order, so a global load issued
// between the fragment requests would make its consumer wait for every fragment requested before it.
//
// Bounds: every workgroup pulls all weight blocks through its CU's L2 port (128 KB per block: about
// 1 us per block), so a block costs every CU about as long as its 16 matrix instructions per wave take
// four times over; rows / 32 workgroups want to be at most one round (8000 rows: 250 workgroups).
#include <hip/hip_runtime.h>
#include "common.h"
#include "layer_tail_frag.h"

#define PG_MAX_BLOCKS ATHENA_PANEL_MAX_BLOCKS

struct PanelArgs {
    const bf16_t* x;
    long ldx;
    int rows, nblock;
    const bf16_t* w[PG_MAX_BLOCKS];
    const float* bias[PG_MAX_BLOCKS];
    bf16_t* out[PG_MAX_BLOCKS];
    long ldo[PG_MAX_BLOCKS];
};

template <int RB>
__global__ __launch_bounds__(LT_THREADS) void panel_gemm_kernel(const PanelArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t ximg[LT_IMG(RB)];
    __shared__ __attribute__((aligned(16))) bf16_t oimg[2][LT_IMG(RB)];
    __shared__ __attribute__((aligned(16))) float bias_s[PG_MAX_BLOCKS][LT_D];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4, fk = fq * 8;
    const int row0 = blockIdx.x * 16 * RB;
    const int n0 = wave * 32;

    // The weights do not depend on the rows: the first two blocks' units are requested in front of
    // everything else.  Two units alternate, each refilled for the block after the next one as it is
    // multiplied: a block's 128 KB take a CU's L2 port about a microsecond, its matrix instructions a
    // quarter of that, so the requests have to run well ahead.
    LtUnit u0, u1;
    lt_load_unit(u0, lt_unit_ptr(p.w[0], LT_D, n0, 0, lane), LT_D);
    lt_load_unit(u1, lt_unit_ptr(p.w[p.nblock > 1 ? 1 : 0], LT_D, n0, 0, lane), LT_D);
    LtTile<RB> t;
    lt_tile_load<RB>(t, p.x, p.ldx, 0, row0, p.rows, tid);
    for (int i = tid; i < p.nblock * LT_D; i += LT_THREADS) {
        const int c = i >> 8, n = i & 255;
        bias_s[c][n] = p.bias[c] ? p.bias[c][n] : 0.f;
    }
    lt_tile_to_img<RB>(ximg, t, tid);
    __syncthreads();
    // every wave multiplies the SAME rows by its columns of every block: the row fragments are read from
    // the image once and stay in registers (64 at 32 rows), so a block is 16 RB matrix instructions
    // back to back with nothing to wait for but its weights
    bf16x8_t a[LT_KB][RB];
#pragma unroll
    for (int kb = 0; kb < LT_KB; ++kb)
#pragma unroll
        for (int i = 0; i < RB; ++i) a[kb][i] = *(const bf16x8_t*)(ximg + lt_img_off<RB>(16 * i + fr, kb * 32 + fk));

    auto block = [&](int c, LtUnit& u) {
        f32x4_t acc[RB][2];
        lt_zero<RB>(acc);
        // (the last two blocks request the last one once more: 16 KB per wave that nobody waits for)
        const int cn = c + 2 < p.nblock ? c + 2 : p.nblock - 1;
        const bf16_t* const next = lt_unit_ptr(p.w[cn], LT_D, n0, 0, lane);
#pragma unroll
        for (int kb = 0; kb < LT_KB; ++kb) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int i = 0; i < RB; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.f[kb][j], a[kb][i], acc[i][j], 0, 0, 0);
            lt_load_slot(u, kb, next, LT_D);
        }
        bf16_t* const o = oimg[c & 1];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + 16 * j + 4 * fq;
                const float4 b4 = *(const float4*)(&bias_s[c][n]);
                const float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
                lt_store4(o + lt_img_off<RB>(16 * i + fr, n), v);
            }
        __syncthreads();
        // rows past the end were loaded as copies of the last row and hold its results: they are
        // stored onto it again, so that no store sits behind a branch (a conditional store would make
        // the compiler's count of outstanding memory operations, which the waits for the weight
        // fragments are derived from, depend on the path)
        bf16_t* const dst = p.out[c];
        const long ld = p.ldo[c];
#pragma unroll
        for (int s = 0; s < RB; ++s) {
            const int q = tid + s * LT_THREADS;
            const int r = q >> 5;
            const int row = min(row0 + r, p.rows - 1);
            lt_store16_through(dst, (long)row * ld + (q & 31) * 8, *(const bf16x8_t*)(o + lt_img_off<RB>(r, (q & 31) * 8)));
        }
    };
    for (int c = 0; c < p.nblock; c += 2) {
        block(c, u0);
        if (c + 1 < p.nblock) block(c + 1, u1);
    }
}

static bool pg_al16(const void* q) { return (((uintptr_t)q) & 15) == 0; }

extern "C" int athena_panel_gemm(const void* x, long ldx, int rows, int nblock, const void* const* w_tiled,
                                 const float* const* bias, void* const* out, const long* ldo,
                                 athena_stream_t stream) {
    ATHENA_REQUIRE(x && w_tiled && out && ldo, ATHENA_E_SHAPE, "athena_panel_gemm: null operand");
    ATHENA_REQUIRE(rows > 0 && nblock > 0 && nblock <= PG_MAX_BLOCKS, ATHENA_E_SHAPE,
                   "athena_panel_gemm: rows=%d blocks=%d (at most %d blocks)", rows, nblock, PG_MAX_BLOCKS);
    ATHENA_REQUIRE(ldx >= LT_D && ldx % 8 == 0 && pg_al16(x), ATHENA_E_ALIGN,
                   "athena_panel_gemm: x needs 16-byte aligned rows of at least 256 elements");
    PanelArgs p = {};
    p.x = (const bf16_t*)x;
    p.ldx = ldx;
    p.rows = rows;
    p.nblock = nblock;
    for (int c = 0; c < nblock; ++c) {
        ATHENA_REQUIRE(w_tiled[c] && out[c], ATHENA_E_SHAPE, "athena_panel_gemm: block %d without weights or output", c);
        ATHENA_REQUIRE(pg_al16(w_tiled[c]) && pg_al16(out[c]) && ldo[c] >= LT_D && ldo[c] % 8 == 0 &&
                           (bias == nullptr || bias[c] == nullptr || pg_al16(bias[c])),
                       ATHENA_E_ALIGN, "athena_panel_gemm: block %d: 16-byte aligned operands and output rows expected", c);
        ATHENA_REQUIRE((long)rows * ldo[c] < (1L << 31), ATHENA_E_SHAPE, "athena_panel_gemm: output %d too large", c);
        p.w[c] = (const bf16_t*)w_tiled[c];
        p.bias[c] = bias ? bias[c] : nullptr;
        p.out[c] = (bf16_t*)out[c];
        p.ldo[c] = ldo[c];
    }
    const int rb = lt_row_blocks(rows);
    const dim3 grid((unsigned)((rows + 16 * rb - 1) / (16 * rb)));
    if (rb == 1) ATHENA_LAUNCH((panel_gemm_kernel<1>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p);
    else ATHENA_LAUNCH((panel_gemm_kernel<2>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p);
    ATHENA_CHECK_LAUNCH("athena_panel_gemm");
    return ATHENA_OK;
}

// ------------------------------------------------------------------------------------------------
// The other way round: a LONG contraction and 256 output columns,
//   out[rows, 256] (+)= act( sum_c a[rows, 256 c .. 256 c + 255] W_c + bias ),
// W_c the [256 x 256] pieces of the kernel along its contraction.  A tile kernel gets 63 x 2 tiles of
// 128 x 128 out of 8000 x 256 -- half the chip -- or splits the contraction and adds the parts up in a
// second launch.  Here a workgroup owns 32 (16) rows of the OUTPUT: the accumulators stay in registers
// over the whole contraction, the rows' 256-wide pieces of a pass through two LDS images (the next
// piece is requested before the current one is multiplied), the weights stream as above.  One barrier
// per piece.
struct PanelKArgs {
    const bf16_t* a;
    long lda;
    int rows, nchunk, kcols;           // kcols: valid columns of a (a multiple of 8; the last piece may be short)
    int act, accumulate;
    const float* bias;
    bf16_t* out;
    long ldo;
    const bf16_t* w[PG_MAX_BLOCKS];
    long ldw[PG_MAX_BLOCKS];
};

// columns col0 .. col0 + 255 of rows row0 ..; 16-byte pieces at or past ``kcols`` are taken from the last
// valid piece of the row instead (finite values that meet zero weights), rows past the end from the last row
template <int RB>
__device__ __forceinline__ void pk_tile_load(LtTile<RB>& t, const bf16_t* __restrict__ src, long ld, int col0, int kcols,
                                             int row0, int rows, int tid) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        const int q = tid + u * LT_THREADS;
        const int col = min(col0 + (q & 31) * 8, kcols - 8);
        t.v[u] = *(const bf16x8_t*)(src + (long)min(row0 + (q >> 5), rows - 1) * ld + col);
    }
}

template <int RB>
__global__ __launch_bounds__(LT_THREADS) void panel_k_kernel(const PanelKArgs p) {
    __shared__ __attribute__((aligned(16))) bf16_t img[2][LT_IMG(RB)];
    __shared__ __attribute__((aligned(16))) float bias_s[LT_D];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int row0 = blockIdx.x * 16 * RB;
    const int n0 = wave * 32;

    LtUnit u;
    lt_load_unit(u, lt_unit_ptr(p.w[0], p.ldw[0], n0, 0, lane), p.ldw[0]);
    LtTile<RB> t;
    pk_tile_load<RB>(t, p.a, p.lda, 0, p.kcols, row0, p.rows, tid);
    if (tid < LT_D) bias_s[tid] = p.bias ? p.bias[tid] : 0.f;
    lt_tile_to_img<RB>(img[0], t, tid);
    __syncthreads();
    f32x4_t acc[RB][2];
    lt_zero<RB>(acc);
    for (int c = 0; c < p.nchunk; ++c) {
        // (past the last piece both requests repeat it: nobody waits for them, and no load sits behind a
        // branch, which would make the compiler's count of outstanding loads depend on the path)
        const int cn = c + 1 < p.nchunk ? c + 1 : c;
        pk_tile_load<RB>(t, p.a, p.lda, 256 * cn, p.kcols, row0, p.rows, tid);
        lt_mul_unit<true, RB>(acc, u, img[c & 1], lane, lt_unit_ptr(p.w[cn], p.ldw[cn], n0, 0, lane), p.ldw[cn]);
        lt_tile_to_img<RB>(img[(c + 1) & 1], t, tid);
        __syncthreads();
    }
    // results through an image nobody reads any more, then 16 bytes per lane along rows
    bf16_t* const o = img[0];       // (every wave is past the last pass' barrier: both images are free)
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + 16 * j + 4 * fq;
            const float4 b4 = *(const float4*)(&bias_s[n]);
            float v[4] = {acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w};
            if (p.act == ATHENA_ACT_RELU6)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r], 0.f), 6.f);
            lt_store4(o + lt_img_off<RB>(16 * i + fr, n), v);
        }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RB; ++s) {
        const int q = tid + s * LT_THREADS;
        const int r = q >> 5;
        if (row0 + r >= p.rows) continue;
        bf16x8_t v = *(const bf16x8_t*)(o + lt_img_off<RB>(r, (q & 31) * 8));
        bf16_t* const dst = p.out + (long)(row0 + r) * p.ldo + (q & 31) * 8;
        if (p.accumulate) {
            const bf16x8_t old = *(const bf16x8_t*)dst;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (bf16_t)((float)v[e] + (float)old[e]);
        }
        *(bf16x8_t*)dst = v;
    }
}

extern "C" int athena_panel_gemm_k(const void* a, long lda, int rows, int kcols, int nchunk, const void* const* w_tiled,
                                   const long* ldw, const float* bias, int act, int accumulate, void* out, long ldo,
                                   athena_stream_t stream) {
    ATHENA_REQUIRE(a && w_tiled && ldw && out, ATHENA_E_SHAPE, "athena_panel_gemm_k: null operand");
    ATHENA_REQUIRE(rows > 0 && nchunk > 0 && nchunk <= PG_MAX_BLOCKS && kcols >= 8 && kcols % 8 == 0 &&
                       kcols > 256 * (nchunk - 1) && kcols <= 256 * nchunk,
                   ATHENA_E_SHAPE, "athena_panel_gemm_k: rows=%d columns=%d pieces=%d (at most %d pieces, the last one not empty)",
                   rows, kcols, nchunk, PG_MAX_BLOCKS);
    ATHENA_REQUIRE(act == ATHENA_ACT_NONE || act == ATHENA_ACT_RELU6, ATHENA_E_UNSUPPORTED, "athena_panel_gemm_k: act %d", act);
    ATHENA_REQUIRE(lda >= kcols && lda % 8 == 0 && ldo >= LT_D && ldo % 8 == 0 && pg_al16(a) && pg_al16(out) &&
                       (bias == nullptr || pg_al16(bias)),
                   ATHENA_E_ALIGN, "athena_panel_gemm_k: 16-byte aligned operands and rows expected");
    ATHENA_REQUIRE((long)rows * lda < (1L << 31) && (long)rows * ldo < (1L << 31), ATHENA_E_SHAPE, "athena_panel_gemm_k: too large");
    PanelKArgs p = {};
    p.a = (const bf16_t*)a;
    p.lda = lda;
    p.rows = rows;
    p.nchunk = nchunk;
    p.kcols = kcols;
    p.act = act;
    p.accumulate = accumulate;
    p.bias = bias;
    p.out = (bf16_t*)out;
    p.ldo = ldo;
    for (int c = 0; c < nchunk; ++c) {
        ATHENA_REQUIRE(w_tiled[c] && pg_al16(w_tiled[c]) && ldw[c] >= LT_D && ldw[c] % 32 == 0, ATHENA_E_ALIGN,
                       "athena_panel_gemm_k: piece %d: 16-byte aligned tiled weights with a contraction length that is a "
                       "multiple of 32 expected", c);
        p.w[c] = (const bf16_t*)w_tiled[c];
        p.ldw[c] = ldw[c];
    }
    const int rb = lt_row_blocks(rows);
    const dim3 grid((unsigned)((rows + 16 * rb - 1) / (16 * rb)));
    if (rb == 1) ATHENA_LAUNCH((panel_k_kernel<1>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p);
    else ATHENA_LAUNCH((panel_k_kernel<2>), grid, dim3(LT_THREADS), 0, (hipStream_t)stream, p);
    ATHENA_CHECK_LAUNCH("athena_panel_gemm_k");
    return ATHENA_OK;
}
