// Shared pieces of the row-block kernels (layer_tail.hip, panel.hip): a workgroup of eight waves owns
// 16 or 32 complete rows of 256-wide activations as LDS operand images, and every wave multiplies them
// by the 32 output columns it owns, with the weights streamed from L2 straight into registers in
// matrix-instruction fragment layout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "common.h"

#define LT_D 256
#define LT_THREADS 512
#ifndef LT_AHEAD2
#define LT_AHEAD2 0                      // k-blocks of image fragments the 32-row build reads ahead
#endif
#define LT_KB 8                          // 32-wide contraction blocks of a 256-wide slice
// A workgroup owns R = 16 RB rows, RB = 2 (the tall launches: one workgroup per CU at 8000 rows) or 1
// (row counts that leave most CUs without a workgroup -- a decoder's 1056 rows: the launch takes as
// long as ONE workgroup, and a workgroup with half the rows is done sooner).  RB is a template
// parameter of everything below.
#define LT_IMG(RB) (16 * (RB) * LT_D)    // elements of a [R rows][256 k] operand image (16 KB at RB = 2)
// row blocks per workgroup for a launch of ``rows`` rows
// (ATHENA_TAIL_SMALL_ROWS: the largest row count that gets the 16-row workgroups; default 2048 = half
// the CUs busy with 32-row ones; 0 = never.  A forward and its backward launch must see the same value:
// the slabs of the LayerNorm sums are one per workgroup.)
static inline int lt_row_blocks(int rows) {
    const char* v = getenv("ATHENA_TAIL_SMALL_ROWS");        // read on every call so that a test can switch it
    return rows <= (v ? atoi(v) : 2048) ? 1 : 2;
}

// Operand image: eight [32 rows][32 k] blocks, 64-byte rows; the 16-byte chunk c of row r sits at
// position c ^ 2*((r >> 2) & 1), which spreads the 16 rows of a fragment read over all banks.
template <int RB>
__device__ __forceinline__ int lt_img_off(int row, int k) {
    const int kb = k >> 5, c = (k >> 3) & 3;
    return kb * (16 * RB * 32) + (row * 4 + (c ^ (((row >> 2) & 1) << 1))) * 8 + (k & 7);
}

struct LtUnit {
    bf16x8_t f[LT_KB][2];
};

// Weights come fragment-tiled (see athena_refresh_shadows): blocks of 512 elements = 16 rows x 32
// contraction indices in lane order, nb-major.  With row-major weights a fragment load touches 16
// rows a power-of-two stride apart -- 16 half-used cache lines on a quarter (or one) of the L2
// channels -- and the kernel ran 1.5x longer.
// Address of this lane's part of a unit: rows n0 .. n0 + 31 of W (``kdim`` contraction indices per
// row), contraction indices from k0.
__device__ __forceinline__ const bf16_t* lt_unit_ptr(const bf16_t* W, long kdim, int n0, int k0, int lane) {
    return W + (long)n0 * kdim + (k0 >> 5) * 512 + lane * 8;
}
// Request k-block kb of a unit (both 16-row halves; ld = contraction indices per row).
__device__ __forceinline__ void lt_load_slot(LtUnit& u, int kb, const bf16_t* __restrict__ p, long ld) {
    u.f[kb][0] = *(const bf16x8_t*)(p + kb * 512);
    u.f[kb][1] = *(const bf16x8_t*)(p + 16 * ld + kb * 512);
}
__device__ __forceinline__ void lt_load_unit(LtUnit& u, const bf16_t* __restrict__ p, long ld) {
#pragma unroll
    for (int kb = 0; kb < LT_KB; ++kb) lt_load_slot(u, kb, p, ld);
}

// acc[i][j] += rows 16 i .. of the image times the unit's columns 16 j ..; element r of acc[i][j]
// is the output at row 16 i + (lane & 15), column n0 + 16 j + 4 (lane >> 4) + r.  With ROLL each
// k-block of the unit is requested again from ``next`` as soon as it has been multiplied (a
// compile-time choice: behind a run-time branch the compiler has to place its waits for the path
// that issued no loads, and every use of a fragment would then wait for the newest requests too).
template <bool ROLL, int RB>
__device__ __forceinline__ void lt_mul_unit(f32x4_t (&acc)[RB][2], LtUnit& u, const bf16_t* img, int lane,
                                            const bf16_t* __restrict__ next, long next_ld) {
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    // The image fragments are requested AHEAD k-blocks before the products that use them: read one
    // k-block at a time and waited for at once, a multiply phase is a chain of eight LDS latencies
    // with two or four matrix instructions each.  The 16-row build has the registers for all eight
    // at once (step -16 us); the 32-row build sits at 256 registers, and one or two blocks ahead cost it
    // more in spills and scheduling fences than they saved (+21 / +49 us per step): it reads as it goes.
    constexpr int AHEAD = RB == 1 ? LT_KB : LT_AHEAD2;
    bf16x8_t a[LT_KB][RB];
#pragma unroll
    for (int kb = 0; kb < AHEAD; ++kb)
#pragma unroll
        for (int i = 0; i < RB; ++i) a[kb][i] = *(const bf16x8_t*)(img + lt_img_off<RB>(16 * i + fr, kb * 32 + fk));
    if (AHEAD > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kb = 0; kb < LT_KB; ++kb) {
        if (kb + AHEAD < LT_KB || AHEAD == 0) {
            const int kn = kb + AHEAD;
#pragma unroll
            for (int i = 0; i < RB; ++i)
                a[kn][i] = *(const bf16x8_t*)(img + lt_img_off<RB>(16 * i + fr, kn * 32 + fk));
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < RB; ++i)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(u.f[kb][j], a[kb][i], acc[i][j], 0, 0, 0);
        if (ROLL) lt_load_slot(u, kb, next, next_ld);
        if (AHEAD > 0 && AHEAD < LT_KB) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int RB>
__device__ __forceinline__ void lt_zero(f32x4_t (&acc)[RB][2]) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
}

// dropout multipliers of the four consecutive elements base .. base + 3, base a multiple of 4 (same
// decisions as dropout_mult)
__device__ __forceinline__ void lt_dropout4(uint64_t key, uint64_t base, uint32_t thresh, float scale, float (&m)[4]) {
    dropout_mult4(key, base, thresh, scale, m);
}

__device__ __forceinline__ void lt_store4(bf16_t* p, const float (&v)[4]) {
    const bf16x4_t t = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *(bf16x4_t*)p = t;
}
__device__ __forceinline__ float lt_round(float v) { return (float)(bf16_t)v; }

// ---- row blocks between global memory and LDS images: 1024 pieces of 16 bytes, two per thread;
// piece q = row q / 32, columns 8 (q % 32) .. + 7 of the block.
template <int RB>
struct LtTile {
    bf16x8_t v[RB];
};
// rows row0 .. row0 + 31 (clamped to the last row), columns col0 .. col0 + 255 of src (row stride ld)
template <int RB>
__device__ __forceinline__ void lt_tile_load(LtTile<RB>& t, const bf16_t* __restrict__ src, long ld, int col0, int row0,
                                             int rows, int tid) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        const int q = tid + u * LT_THREADS;
        t.v[u] = *(const bf16x8_t*)(src + (long)min(row0 + (q >> 5), rows - 1) * ld + col0 + (q & 31) * 8);
    }
}
template <int RB>
__device__ __forceinline__ void lt_tile_to_img(bf16_t* img, const LtTile<RB>& t, int tid) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        const int q = tid + u * LT_THREADS;
        *(bf16x8_t*)(img + lt_img_off<RB>(q >> 5, (q & 31) * 8)) = t.v[u];
    }
}
// A 16-byte store that writes through L2 without keeping the line there (sc1).  The row blocks a
// launch writes (7.5 KB per row of the layer, 60 MB in all) are read by LATER launches; kept in L2
// they push out the layer's weights, which every workgroup of this launch streams from there
// (forward tail 42.6 -> 40.5 us, backward 44.0 -> 41-43 us at 8000 rows).
// ``base`` is the (wave-uniform) start of the tensor, ``elem`` the element the 16 bytes start at.
// Issued as a buffer store through the compiler's builtin, not as inline assembly: a 16-byte store
// followed too closely by a write to its data registers needs wait states that only the compiler's
// hazard recognizer inserts (as inline assembly this stored garbage now and then).
typedef unsigned int lt_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lt_store16_through(bf16_t* base, long elem, bf16x8_t v) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xffffffff, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(lt_u32x4, v), rsrc, (unsigned)(elem * 2), 0, 16 /* sc1 */);
}

template <int RB>
__device__ __forceinline__ void lt_img_to_global(const bf16_t* img, bf16_t* __restrict__ dst, long ld, int col0,
                                                 int row0, int rows, int tid) {
#pragma unroll
    for (int u = 0; u < RB; ++u) {
        const int q = tid + u * LT_THREADS;
        const int r = q >> 5;
#ifdef LT_X_NOSTORE
        if (row0 + r < rows && ld == 12345)
#else
        if (row0 + r < rows)
#endif
            lt_store16_through(dst, (long)(row0 + r) * ld + col0 + (q & 31) * 8, *(const bf16x8_t*)(img + lt_img_off<RB>(r, (q & 31) * 8)));
    }
}
// the four values of an image a lane owns in accumulator layout (row r, columns n .. n + 3, n % 4 == 0)
template <int RB>
__device__ __forceinline__ bf16x4_t lt_img_get4(const bf16_t* img, int r, int n) {
    return *(const bf16x4_t*)(img + lt_img_off<RB>(r, n));
}
