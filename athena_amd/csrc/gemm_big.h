// Large-tile bf16 GEMM core for the long contractions of the step: the implicit-GEMM 3x3 / stride-2
// convolution (forward, input gradient, weight gradient) and the weight gradients of the Dense
// layers, whose contraction runs over all B*T' rows of a batch.
//
//   workgroup = 8 waves (2 along M x 4 along N), output tile 256 x 256, K step 32
//   each wave: 128 x 64 outputs = 8 x 4 accumulators of v_mfma_f32_16x16x32_bf16 (128 registers)
//   operands go from global memory straight into LDS (global_load_lds, 16 bytes per lane, no
//   staging registers) through a ring of four 32 KB stages: while one k-step is multiplied the
//   loads of the next three are in flight (96 KB per CU), and one barrier per k-step hands a
//   landed stage to the readers and a consumed one back to the loads
//
// At 256 x 256 a workgroup does 128 FLOP per operand byte it pulls from L2 (64 at 128 x 128), and a
// conv2-shaped product reads its large operand once.  The depth of the ring is what the memory
// system asks for: with one 64 KB tile in flight per CU (two stages of K step 64) every CU drew
// 29.6 GB/s whatever the product -- 64 KB per 2.2 us of latency -- and the MFMAs waited.
//
// LDS images: row image [rows][32 k] (64-byte rows; 16-byte chunk c of row r sits at position
// c ^ 2*((r >> 2) & 1), which keeps the 16 lanes of every ds_read_b128 group on 16 different bank
// groups) and the transposed image [32 k][256 columns] of gemm_core.h, swizzled by k-row.  A
// direct-to-LDS load writes lane l at "wave base + 16 l", so the swizzle is applied to the SOURCE:
// lane l fetches the chunk that belongs at its position.  Rows that do not exist (a convolution
// tap outside the image, a contraction index past the end) are fetched from a block of zeros.
#pragma once
#include "gemm_core.h"

#define GB_BM 256
#define GB_BN 256
#define GB_BK 32
#define GB_NS 4                                       // stages of the LDS ring
#define GB_THREADS 512
#define GB_A_ELEMS (GB_BM * GB_BK)
#define GB_B_ELEMS (GB_BN * GB_BK)
#define GB_STAGE_ELEMS (GB_A_ELEMS + GB_B_ELEMS)      // 16384 bf16 = 32 KB
#define GB_LDS_ELEMS (GB_NS * GB_STAGE_ELEMS)         // 128 KB

__device__ __attribute__((aligned(64))) const uint32_t gb_zero_chunk[16] = {0};

// One direct-to-LDS load: 64 lanes x 16 bytes land at LDS byte address ``lds_addr`` + 16 * lane
// (``lds_addr`` wave-uniform, in a scalar register; it travels in M0).  Written as inline assembly
// on purpose: the compiler then does not know of the pending LDS write and puts no wait in front of
// the LDS reads of the OTHER stage (it cannot tell two stages of one array apart and would make
// every read wait for all loads in flight, which serialises load and multiply).  The waits are
// placed by hand instead: vmcnt(0) before the barrier that ends a tile.
__device__ __forceinline__ void gb_glds16(const bf16_t* src, uint32_t lds_addr) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(src), "s"(lds_addr)
                 : "memory", "m0");
}
__device__ __forceinline__ void gb_wait_loads() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// all but the 8 youngest loads of this wave (two stages of four) have landed
// (a wave that fills one piece of the A image per stage instead of two has two stages of three)
template <int LOADS>
__device__ __forceinline__ void gb_wait_oldest_stage() {
    static_assert(LOADS == 3 || LOADS == 4, "loads per wave and stage");
    if constexpr (LOADS == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}

// row image with 64-byte rows
__device__ __forceinline__ int row32_img_off(int r, int chunk) { return (r * 4 + (chunk ^ (((r >> 2) & 1) << 1))) * 8; }
__device__ __forceinline__ bf16x8_t lds_frag_rows32(const bf16_t* img, int row, int kofs) {
    return *(const bf16x8_t*)(img + row32_img_off(row, kofs >> 3));
}

// What the epilogue of the large-tile core supports: a plain store of alpha * acc (fp32 or bf16,
// optional output-row scatter), per-column sum / sum of squares of the stored values (BatchNorm
// statistics), raw split-K partial slabs, and the column sums of B taken on the matrix cores.
//
// MH: 16-row sets per wave, 8 for the 256-row tile; 4 makes a 128-row tile out of the same eight waves
// (row images only: each wave then fills one piece of the A image per stage and the upper half of the
// image's space stays unused).  ``bm``: first output row of the tile.
template <class AMap, bool TA, bool TB, int MH = 8>
__device__ __forceinline__ void gemm_big_body(const GemmProblem<AMap>& p, const int bm, const int by, const int bz,
                                              bf16_t* lds) {
    static_assert(MH == 8 || (MH == 4 && !TA), "half tiles are built for row images of A only");
    constexpr int NA = MH / 4;                        // pieces of the A image a wave fills per stage
    const int zb = p.zbatch > 1 ? (int)(bz / p.split_k) : 0;
    const int zsplit = p.zbatch > 1 ? (int)(bz % p.split_k) : (int)bz;
    const int seg_base = p.seg_base + zb;
    const long coff = (long)zb * p.c_batch_stride;
    constexpr int NC = GB_BM / 8;                     // 16-byte chunks per k-row of a transposed image (BM == BN)
    static_assert(GB_BM == GB_BN, "one chunk count for both transposed images");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int bn = by * GB_BN;
    const bf16_t* __restrict__ A = (const bf16_t*)p.A;
    const bf16_t* __restrict__ B = (const bf16_t*)p.B;
    const bf16_t* zero = (const bf16_t*)gb_zero_chunk;
    // LDS byte address of this wave's first piece of the A image of stage 0
    // LDS byte address of this wave's first piece of the A image of stage 0 (2 pieces of 1 KB per
    // wave, operand and stage)
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) bf16_t*)lds;
    const uint32_t lds_wave_a = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)wave * (1024u * NA));
    const uint32_t lds_wave_b = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)wave * 2048u);

    const int tiles_per_seg = (p.kseg + GB_BK - 1) / GB_BK;
    const int ntiles = p.nseg * tiles_per_seg;
    const int per_split = (ntiles + p.split_k - 1) / p.split_k;
    const int t_begin = zsplit * per_split;
    const int t_end = min(ntiles, t_begin + per_split);

    f32x4_t acc[MH][4];
#pragma unroll
    for (int i = 0; i < MH; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // Each wave fills two 1 KB pieces of the A image and two of the B image per stage.
    //   row image:        piece q = rows 16q .. 16q+15; lane l -> row 16q + (l >> 2), position l & 3,
    //                     which holds chunk (l & 3) ^ 2*((l >> 4) & 1) of that row
    //   transposed image: piece q = k-rows 2q, 2q+1; lane l -> k-row 2q + (l >> 5), position l & 31,
    //                     which holds chunk (l & 31) ^ tr_swz(k)
    typename AMap::Row arow[2];
    int acol[2];
    const bf16_t* pb[2];
    int bk[2];                     // k-row of a transposed B piece inside the stage
    int ak[2];                     // k-row of a transposed A piece inside the stage
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = wave * 2 + i;
        const int rchunk = ((lane & 3) ^ (((lane >> 4) & 1) << 1)) * 8;
        if (!TA) {
            const int r = 16 * (wave * NA + (i < NA ? i : 0)) + (lane >> 2);
            arow[i] = p.amap.prep(min(bm + r, p.M - 1));
            acol[i] = rchunk;
            ak[i] = 0;
        } else {
            const int k = 2 * q + (lane >> 5);
            ak[i] = k;
            arow[i] = p.amap.prep(min(t_begin * GB_BK + k, p.kseg - 1));
            const int chunk = (lane & 31) ^ tr_swz<NC>(k);
            acol[i] = min(bm + chunk * 8, ((p.M + 7) & ~7) - 8);
        }
        if (TB) {
            const int r = 16 * q + (lane >> 2);
            pb[i] = B + (long)min(bn + r, p.N - 1) * p.ldb + rchunk;
            bk[i] = 0;
        } else {
            const int k = 2 * q + (lane >> 5);
            bk[i] = k;
            const int chunk = (lane & 31) ^ tr_swz<NC>(k);
            pb[i] = B + (long)k * p.ldb + min(bn + chunk * 8, ((p.N + 7) & ~7) - 8);
        }
    }

    const bool want_bsum = !TB && p.epi.bsum != nullptr && bm == 0 && wm == 0;
    f32x4_t bs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bs[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t ones;
#pragma unroll
    for (int r = 0; r < 8; ++r) ones[r] = (bf16_t)1.0f;

    int ld_t = t_begin;
    int ld_seg = AMap::kSegmented ? t_begin / tiles_per_seg : 0;
    int ld_k0 = (t_begin - ld_seg * tiles_per_seg) * GB_BK;
    // row-image A: where a row starts inside the current segment changes only with the segment
    long aoff[2] = {0, 0};
    bool aok[2] = {true, true};
    int aoff_seg = -1;

    // Issue the four direct-to-LDS loads of the next k-step into ring slot ``slot``.  Past the last
    // k-step the loads still go out (from the block of zeros, into a slot nobody reads any more):
    // every wave then has the same number of loads in flight at every step, and one counted wait
    // serves the whole loop.
    auto stage_tile = [&](int slot) {
        const uint32_t dA = lds_wave_a + (uint32_t)slot * (2u * GB_STAGE_ELEMS);     // bytes
        const uint32_t dB = lds_wave_b + (uint32_t)slot * (2u * GB_STAGE_ELEMS) + 2u * GB_A_ELEMS;
        const bool live = ld_t < t_end;
        const int seg = live ? ld_seg : 0;
        const int k0 = ld_k0;
        const long bseg = AMap::kSegmented ? b_seg_off(p, seg) : p.b_base;
        ++ld_t;
        ld_k0 += GB_BK;
        if (AMap::kSegmented && ld_k0 >= p.kseg) { ld_k0 = 0; ++ld_seg; }
        if (!TA && seg != aoff_seg) {            // uniform: a new segment (convolution tap)
            aoff_seg = seg;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const long o = p.amap.off(arow[i], seg_base + seg);
                aok[i] = !AMap::kCanMiss || o >= 0;
                aoff[i] = o + acol[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const bf16_t* src;
            if (!TA) {
                src = (live && aok[i]) ? A + aoff[i] + k0 : zero;
            } else {
                const long o = p.amap.off(arow[i], seg_base + seg);
                bool ok = live && (k0 + ak[i] < p.kseg);
                if (AMap::kCanMiss) ok = ok && o >= 0;
                src = ok ? A + o + acol[i] : zero;
            }
            gb_glds16(src, dA + 1024u * i);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* src;
            if (TB) src = live ? pb[i] + bseg + k0 : zero;
            else src = (live && k0 + bk[i] < p.kseg) ? pb[i] + bseg + (long)k0 * p.ldb : zero;
            gb_glds16(src, dB + 1024u * i);
        }
        if (TA) {
#pragma unroll
            for (int i = 0; i < 2; ++i) p.amap.step(arow[i], GB_BK);
        }
    };
    // A k-step is "request the twelve fragments from LDS" and "32 MFMAs on them".
    bf16x8_t fa[MH], fb[4];
    auto read_frags = [&](int slot) {
        const bf16_t* sA = lds + slot * GB_STAGE_ELEMS;
        const bf16_t* sB = sA + GB_A_ELEMS;
        const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n0 = wn * 64 + j * 16;
            fb[j] = TB ? lds_frag_rows32(sB, n0 + fr, fk) : lds_frag_tr<NC>(sB, n0, 0, lane);
        }
#pragma unroll
        for (int i = 0; i < MH; ++i) {
            const int m0 = wm * (16 * MH) + i * 16;
            fa[i] = TA ? lds_frag_tr<NC>(sA, m0, 0, lane) : lds_frag_rows32(sA, m0 + fr, fk);
        }
    };
    auto multiply = [&]() {
#pragma unroll
        for (int i = 0; i < MH; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        if (!TB && want_bsum) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], ones, bs[j], 0, 0, 0);
        }
    };

    // The two waves that share a SIMD (wave w and w + 4) would run in lockstep behind the barrier:
    // both compute addresses, both wait for LDS, both want the matrix pipe.  The upper four waves
    // therefore run half a step late: they hold the fragments of a k-step across the barrier and
    // multiply them while the lower four issue loads and read fragments, and the other way round.
    const bool late = __builtin_amdgcn_readfirstlane(wave) >= 4;
    if (t_begin < t_end) {
#pragma unroll
        for (int s_ = 0; s_ < GB_NS - 1; ++s_) stage_tile(s_);
        int slot = 0;
        for (int t = t_begin; t < t_end; ++t) {
            gb_wait_oldest_stage<NA + 2>();  // this wave's loads of k-step t have landed ...
            __syncthreads();                 // ... and everybody's; everybody is done reading k-step t-1
            if (late && t > t_begin) multiply();                // k-step t-1, fragments already in registers
            __builtin_amdgcn_sched_barrier(0);
            stage_tile((slot + GB_NS - 1) & (GB_NS - 1));      // k-step t+3 into the slot of k-step t-1
            read_frags(slot);
            __builtin_amdgcn_sched_barrier(0);
            if (!late) multiply();
            slot = (slot + 1) & (GB_NS - 1);
        }
        if (late) multiply();
        gb_wait_loads();                     // the epilogue reuses LDS: no load may still be landing
        __syncthreads();
    }

    const Epilogue& e = p.epi;
    const int fq = lane >> 4, fr = lane & 15;
    if (!TB && want_bsum && fr == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = bn + wn * 64 + j * 16 + 4 * fq + r;
                if (n < p.N) {
                    if (e.atomic == 2) e.partial[((long)bz * (p.M + 1) + p.M) * e.ldp + n] = bs[j][r];
                    else e.bsum[n] = bs[j][r];
                }
            }
        }
    }
    // element r of acc[i][j] is C[m = bm + wm*16*MH + i*16 + (lane & 15)][n = bn + wn*64 + j*16 + 4*(lane >> 4) + r]
    float cs[4][4], cq[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[j][r] = 0.f; cq[j][r] = 0.f; }
    const bool bf_out = e.c_dtype == ATHENA_BF16;
    const bool vec_ok = (e.ldc % 4 == 0) && ((((uintptr_t)e.C) & 15) == 0) && (coff % 4 == 0);
    // the bias of the lane's sixteen columns (zeros without one)
    float bias_v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = bn + wn * 64 + j * 16 + 4 * fq + r;
            bias_v[j][r] = (e.bias && e.atomic != 2 && n < p.N) ? e.bias[n] : 0.f;
        }
#pragma unroll
    for (int i = 0; i < MH; ++i) {
        const int m = bm + wm * (16 * MH) + i * 16 + fr;
        if (m >= p.M) continue;
        const int mo = out_row(p, m);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = bn + wn * 64 + j * 16 + 4 * fq;
            if (n >= p.N) continue;
            if (e.atomic == 2) {
                // raw partial sums; ldp is a multiple of 8 and the slab is private to this split
                float* c = e.partial + ((long)bz * (p.M + 1) + mo) * e.ldp + n;
                *(f32x4_t*)c = acc[i][j];
                continue;
            }
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r] * e.alpha + bias_v[j][r];
            if (e.colsum) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sv = bf_out ? (float)(bf16_t)v[r] : v[r];
                    if (n + r < p.N) { cs[j][r] += sv; cq[j][r] += sv * sv; }
                }
            }
            if (bf_out) {
                bf16_t* c = (bf16_t*)e.C + coff + (long)mo * e.ldc + n;
                if (n + 3 < p.N && vec_ok) {
                    bf16x4_t o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
                    *(bf16x4_t*)c = o;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) c[r] = (bf16_t)v[r];
                }
            } else {
                float* c = (float*)e.C + coff + (long)mo * e.ldc + n;
                if (n + 3 < p.N && vec_ok) {
                    *(float4*)c = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) c[r] = v[r];
                }
            }
        }
    }
    if (e.colsum && e.atomic != 2) {
        // lanes with equal (lane >> 4) hold the same columns: fold the 16 rows of a lane group by
        // DPP steps, then the two wave rows through LDS; one fp64 atomic per column and statistic
        float* red = (float*)lds;                     // [2 wave rows][256 columns][2]
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float s = row16_sum(cs[j][r]), q = row16_sum(cq[j][r]);
                if (fr == 0) {
                    const int col = wn * 64 + j * 16 + 4 * fq + r;
                    red[(wm * GB_BN + col) * 2] = s;
                    red[(wm * GB_BN + col) * 2 + 1] = q;
                }
            }
        __syncthreads();
        {
            const int col = tid >> 1, which = tid & 1;
            const float t = red[(0 * GB_BN + col) * 2 + which] + red[(1 * GB_BN + col) * 2 + which];
            if (bn + col < p.N) atomicAdd(e.colsum + which * p.N + bn + col, (double)t);
        }
    }
}

template <class AMap, bool TA, bool TB>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_big_kernel(GemmProblem<AMap> p) {
    __shared__ __attribute__((aligned(1024))) bf16_t lds[GB_LDS_ELEMS];
    gemm_big_body<AMap, TA, TB>(p, blockIdx.x * GB_BM, blockIdx.y, blockIdx.z, lds);
}

// The same product with the last, partly filled round of workgroups cut into 128-row tiles: with T
// tiles on 256 CUs the launch takes ceil(T / 256) tile times although the last round may hold few
// tiles; its r tiles as 2 r half tiles (still one round when r <= 128) take little more than half a
// tile time.  Workgroups 0 .. full - 1 are whole tiles in the order of the plain launch (x fastest),
// the rest are the halves of tiles full, full + 1, ...
template <class AMap>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_big_tail_kernel(GemmProblem<AMap> p, int tiles_m, int full) {
    __shared__ __attribute__((aligned(1024))) bf16_t lds[GB_LDS_ELEMS];
    const int id = blockIdx.x;
    if (id < full) {
        gemm_big_body<AMap, false, true>(p, (id % tiles_m) * GB_BM, id / tiles_m, 0, lds);
    } else {
        const int t = full + ((id - full) >> 1), half = (id - full) & 1;
        const int bm = (t % tiles_m) * GB_BM + half * (GB_BM / 2);
        if (bm >= p.M) return;
        gemm_big_body<AMap, false, true, 4>(p, bm, t / tiles_m, 0, lds);
    }
}

// ------------------------------------------------------------------ a table of weight gradients
// C_i[M_i, N_i] = A_i^T B_i for many (A_i [K_i, M_i], B_i [K_i, N_i]) in one launch, no K split:
// every 256 x 256 output tile is one workgroup that runs down the whole contraction.  The table
// lives in device memory (mirror of athena_wgrad_item); workgroup ids are dealt to the problems
// through first_tile.
struct GbItem {
    const void* A;
    const void* B;
    float* C;
    float* bsum;
    int32_t M, N, K;
    int32_t lda, ldb, ldc;
    int32_t first_tile, tiles_m;
};
static_assert(sizeof(GbItem) == sizeof(athena_wgrad_item), "GbItem mirrors athena_wgrad_item");

__global__ __launch_bounds__(GB_THREADS, 2) void gemm_big_table_kernel(const GbItem* __restrict__ table, int count) {
    __shared__ __attribute__((aligned(1024))) bf16_t lds[GB_LDS_ELEMS];
    // workgroup -> tile: the planner's eight ranges, one per XCD under round-robin placement
    const int per = gridDim.x >> 3;
    const int tile = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    // the problem this tile belongs to (uniform; the table is a few KB and stays in cache)
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_tile <= tile) lo = mid;
        else hi = mid - 1;
    }
    const GbItem it = table[lo];
    const int local = tile - it.first_tile;
    if (local < 0 || local >= it.tiles_m * ((it.N + GB_BN - 1) / GB_BN)) return;      // padding of a range
    GemmProblem<PlainRows> p = {};
    p.A = it.A;
    p.B = it.B;
    p.amap.ld = it.lda;
    p.M = it.M;
    p.N = it.N;
    p.nseg = 1;
    p.kseg = it.K;
    p.ldb = it.ldb;
    p.b_base = 0; p.b_stride_lo = 0; p.b_stride_hi = 0; p.b_shift = 0; p.b_mask = 0;
    p.seg_base = 0;
    p.split_k = 1;
    p.zbatch = 1;
    p.c_batch_stride = 0;
    p.crow_mode = 0;
    p.epi.C = it.C;
    p.epi.colsum = nullptr;
    p.epi.partial = nullptr;
    p.epi.bsum = it.bsum;
    p.epi.ldp = 0;
    p.epi.ldc = it.ldc;
    p.epi.c_dtype = ATHENA_F32;
    p.epi.atomic = 0;
    p.epi.alpha = 1.f;
    gemm_big_body<PlainRows, true, false>(p, (local % it.tiles_m) * GB_BM, local / it.tiles_m, 0, lds);
}

// does the large-tile core cover this problem?
template <class AMap>
static inline bool gemm_big_ok(const GemmProblem<AMap>& p, int in_dtype, bool ta) {
    const Epilogue& e = p.epi;
    if (in_dtype != ATHENA_BF16) return false;
    if (e.C_pre || e.aux || e.act != ATHENA_ACT_NONE || e.drop_thresh || e.accum || e.atomic == 1) return false;
    if (e.bias && e.atomic == 2) return false;           // (a split product adds its bias in the reduce pass)
    if (!ta && (p.kseg % GB_BK) != 0) return false;     // a row image has no zero fill for a partial k-step
    return true;
}

// ------------------------------------------------------------------ several problems, one launch
// Up to four problems of one row map (the four parity classes of the stride-2 convolution's input
// gradient: 1, 2, 2 and 4 taps, each a fraction of a chip's worth of tiles) as ONE launch: the tiles
// of the problem with the longest contraction come first, so that the short ones fill the CUs behind
// them.  first_tile[i] .. first_tile[i + 1] - 1 are the workgroups of problem i.
#define GB_MULTI_MAX 4
template <class AMap>
struct GbMulti {
    GemmProblem<AMap> p[GB_MULTI_MAX];
    int first_tile[GB_MULTI_MAX + 1];
    int count;
};

template <class AMap, bool TA, bool TB>
__global__ __launch_bounds__(GB_THREADS, 2) void gemm_big_multi_kernel(const GbMulti<AMap> m) {
    __shared__ __attribute__((aligned(1024))) bf16_t lds[GB_LDS_ELEMS];
    int i = 0;
#pragma unroll
    for (int q = 1; q < GB_MULTI_MAX; ++q)
        if (q < m.count && (int)blockIdx.x >= m.first_tile[q]) i = q;
    const int local = blockIdx.x - m.first_tile[i];
    // a uniform switch instead of m.p[i]: every problem is addressed with constant offsets
    switch (i) {
        case 0: { const int tm = (m.p[0].M + GB_BM - 1) / GB_BM; gemm_big_body<AMap, TA, TB>(m.p[0], (local % tm) * GB_BM, local / tm, 0, lds); } break;
        case 1: { const int tm = (m.p[1].M + GB_BM - 1) / GB_BM; gemm_big_body<AMap, TA, TB>(m.p[1], (local % tm) * GB_BM, local / tm, 0, lds); } break;
        case 2: { const int tm = (m.p[2].M + GB_BM - 1) / GB_BM; gemm_big_body<AMap, TA, TB>(m.p[2], (local % tm) * GB_BM, local / tm, 0, lds); } break;
        default: { const int tm = (m.p[3].M + GB_BM - 1) / GB_BM; gemm_big_body<AMap, TA, TB>(m.p[3], (local % tm) * GB_BM, local / tm, 0, lds); } break;
    }
}

// problems with split_k == 1, no z batch, (!ta && tb); sorted here by contraction length
template <class AMap>
static int launch_gemm_big_multi(const GemmProblem<AMap>* probs, int count, hipStream_t stream) {
    if (count < 1 || count > GB_MULTI_MAX) return ATHENA_E_UNSUPPORTED;
    int order[GB_MULTI_MAX];
    for (int i = 0; i < count; ++i) order[i] = i;
    for (int i = 0; i < count; ++i)
        for (int j = i + 1; j < count; ++j)
            if ((long)probs[order[j]].nseg * probs[order[j]].kseg > (long)probs[order[i]].nseg * probs[order[i]].kseg) {
                const int t = order[i]; order[i] = order[j]; order[j] = t;
            }
    GbMulti<AMap> m;
    memset(&m, 0, sizeof(m));
    m.count = count;
    int tiles = 0;
    for (int i = 0; i < count; ++i) {
        m.p[i] = probs[order[i]];
        if (m.p[i].split_k != 1 || m.p[i].zbatch > 1) return ATHENA_E_UNSUPPORTED;
        m.first_tile[i] = tiles;
        tiles += ((m.p[i].M + GB_BM - 1) / GB_BM) * ((m.p[i].N + GB_BN - 1) / GB_BN);
    }
    for (int i = count; i <= GB_MULTI_MAX; ++i) m.first_tile[i] = tiles;
    ATHENA_LAUNCH((gemm_big_multi_kernel<AMap, false, true>), dim3(tiles), dim3(GB_THREADS), 0, stream, m);
    ATHENA_CHECK_LAUNCH("gemm(large tile, several problems)");
    return ATHENA_OK;
}

#define GB_CHIP_TILES 256     // workgroups of this core the chip runs at a time (one per CU)
static inline bool gb_half_tiles_enabled() {
    static const bool on = !(getenv("ATHENA_GEMM_HALF_TILES") && atoi(getenv("ATHENA_GEMM_HALF_TILES")) == 0);
    return on;
}
template <class AMap>
static int launch_gemm_big(const GemmProblem<AMap>& p, bool ta, bool tb, hipStream_t stream) {
    const int tm = (p.M + GB_BM - 1) / GB_BM, tn = (p.N + GB_BN - 1) / GB_BN;
    const int gz = p.split_k * (p.zbatch > 1 ? p.zbatch : 1);
    const dim3 grid(tm, tn, gz);
    // a last round of at most 128 tiles runs as half tiles (see gemm_big_tail_kernel)
    const long tiles = (long)tm * tn;
    const int last = (int)(tiles % GB_CHIP_TILES);
    if (!ta && tb && gz == 1 && tiles > GB_CHIP_TILES && last > 0 && last <= GB_CHIP_TILES / 2 && gb_half_tiles_enabled()) {
        const int full = (int)(tiles - last);
        ATHENA_LAUNCH((gemm_big_tail_kernel<AMap>), dim3((unsigned)(full + 2 * last)), dim3(GB_THREADS), 0, stream, p, tm, full);
        ATHENA_CHECK_LAUNCH("gemm(large tile, halved last round)");
        return ATHENA_OK;
    }
    if (!ta && tb) ATHENA_LAUNCH((gemm_big_kernel<AMap, false, true>), grid, dim3(GB_THREADS), 0, stream, p);
    else if (ta && !tb) ATHENA_LAUNCH((gemm_big_kernel<AMap, true, false>), grid, dim3(GB_THREADS), 0, stream, p);
    else return ATHENA_E_UNSUPPORTED;
    ATHENA_CHECK_LAUNCH("gemm(large tile)");
    return ATHENA_OK;
}
