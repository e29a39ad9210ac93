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
