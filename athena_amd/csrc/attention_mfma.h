// Fused multi-head attention on the matrix cores (bf16 operands, fp32 softmax and accumulation).
//
// Workgroup = 4 waves; a wave owns 16 rows of the "resident" side (queries in the forward and
// dQ kernels, keys in the dK/dV kernel) and walks the other side in blocks of 64 rows that are
// staged through a double-buffered LDS image.  Every product is issued so that the 16 resident
// rows are the COLUMNS of the MFMA result: a lane then holds, for resident row (lane & 15), four
// consecutive rows 4*(lane >> 4) + r of each 16-row tile of the streamed side.  With that layout
//   * the softmax statistics of a query live in the lane(s) that also hold its output columns,
//   * the probabilities a lane holds are exactly the B-operand fragment of the next product
//     (the contraction index is a permutation of the streamed rows, applied identically to the
//     A operand through the transposing LDS read), so P and dS never travel through LDS.
// Scores never leave the chip.  The additive mask is the reference's fp32 arithmetic,
// s * scale + (-1e9), so fully masked rows come out as the same uniform average
// (athena/layers/attention.py:47-70).
#pragma once
#include "common.h"

#define AM_ROWS 64      // rows per workgroup on the resident side and per LDS block on the streamed side

template <int DK>
struct AttImg {
    static constexpr int NC = DK / 8;     // 16-byte chunks per row
    // chunk swizzle: conflict-free for 8 consecutive rows reading one chunk (ds_read_b128) and for
    // 4 consecutive rows reading a pair of chunks (ds_read_b64_tr_b16)
    __device__ static __forceinline__ int swz(int r) {
        return DK == 32 ? ((((r >> 1) & 1) << 1) | ((r >> 2) & 1)) : (((r & 3) << 1) | ((r >> 2) & 1));
    }
    __device__ static __forceinline__ int off(int r, int chunk) { return (r * NC + (chunk ^ swz(r))) * 8; }
};

// register-staged copy of a [64][DK] block of rows (row stride ld) into an LDS image
template <int DK>
struct AttStage {
    static constexpr int NC = DK / 8;
    static constexpr int NP = AM_ROWS * NC / 256;
    uint4 r[NP];
    __device__ __forceinline__ void load(const bf16_t* __restrict__ base, int ld, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = tid + 256 * i;
            const int row = row0 + c / NC, ch = c % NC;
            const int rc = row < nrows ? row : nrows - 1;
            uint4 v = *(const uint4*)(base + (long)rc * ld + ch * 8);
            if (row >= nrows) v = make_uint4(0, 0, 0, 0);
            r[i] = v;
        }
    }
    __device__ __forceinline__ void store(bf16_t* img, int tid) const {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = tid + 256 * i;
            *(uint4*)(img + AttImg<DK>::off(c / NC, c % NC)) = r[i];
        }
    }
};

// A-operand fragment from image rows: row (row0 + lane & 15), elements 32*ks + 8*(lane >> 4) .. +7
template <int DK>
__device__ __forceinline__ bf16x8_t att_frag_rows(const bf16_t* img, int row0, int ks, int lane) {
    return *(const bf16x8_t*)(img + AttImg<DK>::off(row0 + (lane & 15), 4 * ks + (lane >> 4)));
}

// A-operand fragment of the TRANSPOSED image: for output row (col0 + lane & 15) the eight
// contraction slots of lane group g = lane >> 4 are image rows ra + 4g + {0..3}, rb + 4g + {0..3}
template <int DK>
__device__ __forceinline__ bf16x8_t att_frag_tr(const bf16_t* img, int ra, int rb, int col0, int lane) {
    typedef short s4 __attribute__((ext_vector_type(4)));
    typedef short s8 __attribute__((ext_vector_type(8)));
    const int g = lane >> 4, ig = lane & 15, q = ig >> 2, pp = ig & 3;
    const int chunk = (col0 >> 3) + (pp >> 1);
    const bf16_t* a0 = img + AttImg<DK>::off(ra + 4 * g + q, chunk) + (pp & 1) * 4;
    const bf16_t* a1 = img + AttImg<DK>::off(rb + 4 * g + q, chunk) + (pp & 1) * 4;
    const s4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a0);
    const s4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)a1);
    const s8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t att_pack(const f32x4_t a, const f32x4_t b) {
    bf16x8_t o;
#pragma unroll
    for (int r = 0; r < 4; ++r) { o[r] = (bf16_t)a[r]; o[4 + r] = (bf16_t)b[r]; }
    return o;
}

// eight elements of row min(row, nrows - 1) of a matrix, zeroed when row >= nrows (the load itself is
// always in bounds, so the select is on the data and nothing goes through scratch memory)
__device__ __forceinline__ bf16x8_t att_load8(const bf16_t* base, int row, int nrows, int ld, int col) {
    const int rc = row < nrows ? row : nrows - 1;
    uint4 v = *(const uint4*)(base + (long)rc * ld + col);
    if (row >= nrows) v = make_uint4(0, 0, 0, 0);
    return __builtin_bit_cast(bf16x8_t, v);
}

// additive score bias of a streamed key: 0 = attend, -1e9 = masked (the reference's additive mask),
// -inf = past the end.  Kept as a float per key so that the score loop is one multiply-add per
// element with no branches (per-element flag tests compiled into a chain of exec-mask branches
// that took longer than the matrix products of the block).
__device__ __forceinline__ float att_key_bias(const uint8_t* __restrict__ key_mask, long b, int Tk, int kj) {
    if (kj >= Tk) return -INFINITY;
    return (key_mask && key_mask[b * Tk + kj]) ? -1e9f : 0.f;
}

// Workgroup -> (block w of nW, head): the launch is one-dimensional and consecutive workgroup ids go to
// the eight XCDs in turn, each with an L2 of its own, so the nW workgroups of a head are given ids
// that differ by multiples of eight: they run on one XCD at about the same time and the head's
// operands come through that one L2 once.  The grid is nW * roundup(heads, 8); false = padding.
__device__ __forceinline__ bool att_xcd_map(int nW, int heads, int& w, int& head) {
    if (heads < 0) {        // (plain order, for measurements: consecutive workgroups = the blocks of a head)
        head = blockIdx.x / nW;
        w = blockIdx.x - head * nW;
        return head < -heads;
    }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    head = (slot / nW) * 8 + xcd;
    w = slot - (slot / nW) * nW;
    return head < heads;
}

// ---------------------------------------------------------------- forward
// KH = 2: eight waves, the upper four take the second half of the key blocks of the same 64 queries,
// and the two partial results (running maximum, sum and unnormalised output per query) are merged
// through LDS at the end.  A launch takes as long as one workgroup's chain of key blocks (there are
// about as many workgroups as the chip holds at once), so halving the chain shortens the launch.
template <int DK, bool CAUSAL, int KH>
__global__ __launch_bounds__(256 * KH) void mha_fwd_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                           const bf16_t* __restrict__ v, int ldq, int ldk, int ldv,
                                                           const uint8_t* __restrict__ key_mask, int causal, int h,
                                                           int Tq, int Tk, float scale, bf16_t* __restrict__ out,
                                                           int ldo, float* __restrict__ lse, int heads) {
    constexpr int KS = DK / 32, NT = DK / 16;
    __shared__ __attribute__((aligned(16))) bf16_t sK[KH][2][AM_ROWS * DK];
    __shared__ __attribute__((aligned(16))) bf16_t sV[KH][2][AM_ROWS * DK];
    __shared__ __attribute__((aligned(16))) float sM[KH][2][AM_ROWS];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int kh = KH == 2 ? (int)(threadIdx.x >> 8) : 0;
    int qblk, head;
    if (!att_xcd_map((Tq + AM_ROWS - 1) / AM_ROWS, heads, qblk, head)) return;
    const int b = head / h, hh = head % h;
    const bf16_t* qb = q + (long)b * Tq * ldq + hh * DK;
    const bf16_t* kb = k + (long)b * Tk * ldk + hh * DK;
    const bf16_t* vb = v + (long)b * Tk * ldv + hh * DK;
    const int qi = qblk * AM_ROWS + wave * 16 + li;

    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = att_load8(qb, qi, Tq, ldq, 32 * ks + 8 * g);
    f32x4_t o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) o[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lp = 0.f;

    AttStage<DK> stK, stV;
    float mflag = 0.f;
    const int nblk_all = (Tk + AM_ROWS - 1) / AM_ROWS;
    // this half's key blocks: [blk0, blk0 + nblk); both halves run the same number of trips (the
    // barriers are the workgroup's), a half without a block in a trip works on keys past the end
    const int per = (nblk_all + KH - 1) / KH;
    const int blk0 = kh * per;
    const int nblk = per;
    stK.load(kb, ldk, blk0 * AM_ROWS, Tk, tid);
    stV.load(vb, ldv, blk0 * AM_ROWS, Tk, tid);
    if (tid < AM_ROWS) mflag = att_key_bias(key_mask, b, Tk, blk0 * AM_ROWS + tid);
    stK.store(sK[kh][0], tid);
    stV.store(sV[kh][0], tid);
    if (tid < AM_ROWS) sM[kh][0][tid] = mflag;
    __syncthreads();

    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, j0 = (blk0 + blk) * AM_ROWS;
        const bool more = blk + 1 < nblk;
        if (more) {
            stK.load(kb, ldk, j0 + AM_ROWS, Tk, tid);
            stV.load(vb, ldv, j0 + AM_ROWS, Tk, tid);
            if (tid < AM_ROWS) mflag = att_key_bias(key_mask, b, Tk, j0 + AM_ROWS + tid);
        }
        // scores, transposed: s[jt][r] = q(qi) . k(j0 + 16 jt + 4 g + r)
        f32x4_t s[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            s[jt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                s[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_frag_rows<DK>(sK[kh][cur], 16 * jt, ks, lane), qf[ks],
                                                                s[jt], 0, 0, 0);
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const float4 b4 = *(const float4*)&sM[kh][cur][16 * jt + 4 * g];
            const float kb4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float bias = kb4[r];
                if constexpr (CAUSAL) {
                    // a key that is both padded and ahead of the query gets the additive mask once
                    const int kj = j0 + 16 * jt + 4 * g + r;
                    bias = (kj > qi + causal - 1 && bias == 0.f) ? -1e9f : bias;
                }
                const float sv = s[jt][r] * scale + bias;
                s[jt][r] = sv;
                tmax = fmaxf(tmax, sv);
            }
        }
        tmax = rows4_max(tmax);
        const float mn = fmaxf(m, tmax);
        // (a block of keys past the end leaves the maximum at -inf: nothing to rescale, nothing to add)
        const float alpha = mn == -INFINITY ? 1.f : __expf(m - mn);          // first block: exp(-inf) = 0
        m = mn;
        float psum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = mn == -INFINITY ? 0.f : __expf(s[jt][r] - mn);   // keys past the end: exp(-inf) = 0
                s[jt][r] = p;
                psum += p;
            }
        lp = lp * alpha + psum;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) o[nt] *= alpha;
        // out^T[d][query] += sum_key v[key][d] p[query][key]
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const bf16x8_t pf = att_pack(s[2 * kk], s[2 * kk + 1]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    att_frag_tr<DK>(sV[kh][cur], 32 * kk, 32 * kk + 16, 16 * nt, lane), pf, o[nt], 0, 0, 0);
        }
        if (more) {
            stK.store(sK[kh][cur ^ 1], tid);
            stV.store(sV[kh][cur ^ 1], tid);
            if (tid < AM_ROWS) sM[kh][cur ^ 1][tid] = mflag;
        }
        __syncthreads();
    }
    float l = lp;
    l = rows4_sum(l);
    if constexpr (KH == 2) {
        // the upper half parks (m, l, o) of its 64 queries in its own first key image (free behind the
        // last barrier of the loop); the lower half merges: o = o0 e^(m0 - M) + o1 e^(m1 - M)
        float* park_o = reinterpret_cast<float*>(&sK[1][0][0]);      // [256 threads][NT * 4]: exactly the two images
        float* park_ml = reinterpret_cast<float*>(&sV[1][0][0]);     // [256 threads][2]
        static_assert(256 * NT * 4 * 4 <= 2 * AM_ROWS * DK * 2, "the parking space is the half's two key images");
        if (kh == 1) {
            float* mine = park_o + tid * (NT * 4);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) *(f32x4_t*)(mine + 4 * nt) = o[nt];
            park_ml[2 * tid] = m;
            park_ml[2 * tid + 1] = l;
        }
        __syncthreads();
        if (kh == 1) return;
        const float* other = park_o + tid * (NT * 4);
        const float m1 = park_ml[2 * tid], l1 = park_ml[2 * tid + 1];
        const float M = fmaxf(m, m1);
        const float a0 = m == -INFINITY ? 0.f : __expf(m - M), a1 = m1 == -INFINITY ? 0.f : __expf(m1 - M);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[nt][r] = o[nt][r] * a0 + other[4 * nt + r] * a1;
        l = l * a0 + l1 * a1;
        m = M;
    }
    if (qi < Tq) {
        const float inv = 1.f / l;
        bf16_t* ob = out + ((long)b * Tq + qi) * ldo + hh * DK + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x4_t w = {(bf16_t)(o[nt][0] * inv), (bf16_t)(o[nt][1] * inv), (bf16_t)(o[nt][2] * inv),
                          (bf16_t)(o[nt][3] * inv)};
            *(bf16x4_t*)(ob + 16 * nt) = w;
        }
        if (g == 0) lse[((long)b * h + hh) * Tq + qi] = m + __logf(l);
    }
}

// ---------------------------------------------------------------- backward
// dQ: a wave owns 16 queries, keys/values streamed.  delta[b,h,i] = sum_c dout[b,i,h,c] * out[b,i,h,c]
// is worked out here from the dout fragments the kernel holds anyway (the four lanes of a query
// cover its DK columns between them) and written out for the dK/dV kernel that follows.
// LDS of a backward workgroup, either role: two stages of two [64][DK] images and of two float rows
template <int DK>
struct AttBwdLds {
    bf16_t a[2][AM_ROWS * DK];      // K (dQ role) or Q (dK/dV role)
    bf16_t b[2][AM_ROWS * DK];      // V or dout
    float x[2][AM_ROWS];            // key bias or lse
    float y[2][AM_ROWS];            // -, or delta
};

// dQ: a wave owns RS sets of 16 queries (a workgroup 64 * RS), keys / values streamed.  With two sets a
// fragment of the streamed side read from LDS serves two products, and a head has half as many of
// these workgroups: together with its dK/dV workgroups they then fit the chip in one go at three
// waves per SIMD.
template <int DK, bool CAUSAL, int RS>
__device__ __forceinline__ void mha_bwd_dq_body(AttBwdLds<DK>& sm, int qblk, int head, const bf16_t* __restrict__ q,
                                                const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ldq,
                                                int ldk, int ldv, const uint8_t* __restrict__ key_mask, int causal,
                                                int h, int Tq, int Tk, float scale, const bf16_t* __restrict__ out,
                                                const bf16_t* __restrict__ dout, int ldo,
                                                const float* __restrict__ lse, float* __restrict__ delta,
                                                bf16_t* __restrict__ dq, int lddq) {
    constexpr int KS = DK / 32, NT = DK / 16;
    bf16_t (&sK)[2][AM_ROWS * DK] = sm.a;
    bf16_t (&sV)[2][AM_ROWS * DK] = sm.b;
    float (&sM)[2][AM_ROWS] = sm.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int b = head / h, hh = head % h;
    const bf16_t* kb = k + (long)b * Tk * ldk + hh * DK;
    const bf16_t* vb = v + (long)b * Tk * ldv + hh * DK;
    const int qi0 = qblk * (AM_ROWS * RS) + wave * (16 * RS) + li;        // set rs: qi0 + 16 rs

    bf16x8_t qf[RS][KS], dof[RS][KS];
    float del_i[RS], lse_i[RS];
    f32x4_t acc[RS][NT];
#pragma unroll
    for (int rs = 0; rs < RS; ++rs) {
        const int qi = qi0 + 16 * rs;
        float d = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            qf[rs][ks] = att_load8(q + (long)b * Tq * ldq + hh * DK, qi, Tq, ldq, 32 * ks + 8 * g);
            dof[rs][ks] = att_load8(dout + (long)b * Tq * ldo + hh * DK, qi, Tq, ldo, 32 * ks + 8 * g);
            const bf16x8_t of = att_load8(out + (long)b * Tq * ldo + hh * DK, qi, Tq, ldo, 32 * ks + 8 * g);
#pragma unroll
            for (int r = 0; r < 8; ++r) d += (float)dof[rs][ks][r] * (float)of[r];
        }
        d = rows4_sum(d);
        del_i[rs] = d;
        lse_i[rs] = qi < Tq ? lse[((long)b * h + hh) * Tq + qi] : INFINITY;
        if (qi < Tq && g == 0) delta[((long)b * h + hh) * Tq + qi] = d;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[rs][nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    AttStage<DK> stK, stV;
    float mflag = 0.f;
    const int nblk = (Tk + AM_ROWS - 1) / AM_ROWS;
    stK.load(kb, ldk, 0, Tk, tid);
    stV.load(vb, ldv, 0, Tk, tid);
    if (tid < AM_ROWS) mflag = att_key_bias(key_mask, b, Tk, tid);
    stK.store(sK[0], tid);
    stV.store(sV[0], tid);
    if (tid < AM_ROWS) sM[0][tid] = mflag;
    __syncthreads();

    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, j0 = blk * AM_ROWS;
        const bool more = blk + 1 < nblk;
        if (more) {
            stK.load(kb, ldk, j0 + AM_ROWS, Tk, tid);
            stV.load(vb, ldv, j0 + AM_ROWS, Tk, tid);
            if (tid < AM_ROWS) mflag = att_key_bias(key_mask, b, Tk, j0 + AM_ROWS + tid);
        }
        // 32 keys at a time (two 16-key tiles: the contraction of one step of the dq product)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4_t s[RS][2], dp[RS][2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int jt = 2 * kk + h2;
#pragma unroll
                for (int rs = 0; rs < RS; ++rs) {
                    s[rs][h2] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                    dp[rs][h2] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8_t kfr = att_frag_rows<DK>(sK[cur], 16 * jt, ks, lane);
                    const bf16x8_t vfr = att_frag_rows<DK>(sV[cur], 16 * jt, ks, lane);
#pragma unroll
                    for (int rs = 0; rs < RS; ++rs) {
                        s[rs][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, qf[rs][ks], s[rs][h2], 0, 0, 0);
                        dp[rs][h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr, dof[rs][ks], dp[rs][h2], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int jt = 2 * kk + h2;
                const float4 b4 = *(const float4*)&sM[cur][16 * jt + 4 * g];
                const float kb4[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int rs = 0; rs < RS; ++rs) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float bias = kb4[r];
                        if constexpr (CAUSAL) {
                            const int kj = j0 + 16 * jt + 4 * g + r;
                            bias = (kj > qi0 + 16 * rs + causal - 1 && bias == 0.f) ? -1e9f : bias;
                        }
                        const float sv = s[rs][h2][r] * scale + bias;
                        const float p = __expf(sv - lse_i[rs]);
                        s[rs][h2][r] = p * (dp[rs][h2][r] - del_i[rs]) * scale;     // dS
                    }
                }
            }
            // dq^T[d][query] += sum_key k[key][d] dS[query][key]
            bf16x8_t dsf[RS];
#pragma unroll
            for (int rs = 0; rs < RS; ++rs) dsf[rs] = att_pack(s[rs][0], s[rs][1]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const bf16x8_t ktr = att_frag_tr<DK>(sK[cur], 32 * kk, 32 * kk + 16, 16 * nt, lane);
#pragma unroll
                for (int rs = 0; rs < RS; ++rs)
                    acc[rs][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktr, dsf[rs], acc[rs][nt], 0, 0, 0);
            }
        }
        if (more) {
            stK.store(sK[cur ^ 1], tid);
            stV.store(sV[cur ^ 1], tid);
            if (tid < AM_ROWS) sM[cur ^ 1][tid] = mflag;
        }
        __syncthreads();
    }
#pragma unroll
    for (int rs = 0; rs < RS; ++rs) {
        const int qi = qi0 + 16 * rs;
        if (qi < Tq) {
            bf16_t* ob = dq + ((long)b * Tq + qi) * lddq + hh * DK + 4 * g;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bf16x4_t w = {(bf16_t)acc[rs][nt][0], (bf16_t)acc[rs][nt][1], (bf16_t)acc[rs][nt][2],
                              (bf16_t)acc[rs][nt][3]};
                *(bf16x4_t*)(ob + 16 * nt) = w;
            }
        }
    }
}

// dK, dV: a wave owns 16 keys, queries / dout streamed.  delta[i] = sum_c dout[i][c] out[i][c] of the
// streamed queries is worked out here as well, from the pieces of dout the staging loads hold and the
// same pieces of out (the eight lanes that share a row add up their parts), so this role does not
// wait for the other one.
template <int DK>
__device__ __forceinline__ float att_stage_rowdot(const AttStage<DK>& a, const bf16_t* __restrict__ other, int ld,
                                                  int row0, int nrows, int tid, int i) {
    constexpr int NC = DK / 8;
    const int c = tid + 256 * i;
    const int row = row0 + c / NC, ch = c % NC;
    float s = 0.f;
    if (row < nrows) {
        const bf16x8_t o = __builtin_bit_cast(bf16x8_t, *(const uint4*)(other + (long)row * ld + ch * 8));
        const bf16x8_t d = __builtin_bit_cast(bf16x8_t, a.r[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += (float)d[e] * (float)o[e];
    }
    // the NC lanes of a row are consecutive
#pragma unroll
    for (int o = 1; o < NC; o <<= 1) s += __shfl_xor(s, o, 64);
    return s;
}

template <int DK, bool CAUSAL>
__device__ __forceinline__ void mha_bwd_dkv_body(AttBwdLds<DK>& sm, int kblk, int head, const bf16_t* __restrict__ q,
                                                 const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ldq,
                                                 int ldk, int ldv, const uint8_t* __restrict__ key_mask, int causal,
                                                 int h, int Tq, int Tk, float scale, const bf16_t* __restrict__ out,
                                                 const bf16_t* __restrict__ dout, int ldo,
                                                 const float* __restrict__ lse, bf16_t* __restrict__ dk_,
                                                 bf16_t* __restrict__ dv, int lddk, int lddv) {
    constexpr int KS = DK / 32, NT = DK / 16, NC = DK / 8, NP = AM_ROWS * NC / 256;
    bf16_t (&sQ)[2][AM_ROWS * DK] = sm.a;
    bf16_t (&sO)[2][AM_ROWS * DK] = sm.b;
    float (&sL)[2][AM_ROWS] = sm.x;
    float (&sD)[2][AM_ROWS] = sm.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
    const int b = head / h, hh = head % h;
    const bf16_t* qb = q + (long)b * Tq * ldq + hh * DK;
    const bf16_t* ob = dout + (long)b * Tq * ldo + hh * DK;
    const bf16_t* yb = out + (long)b * Tq * ldo + hh * DK;
    const float* lb = lse + ((long)b * h + hh) * Tq;
    const int kj = kblk * AM_ROWS + wave * 16 + li;
    const bool kok = kj < Tk;

    bf16x8_t kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kf[ks] = att_load8(k + (long)b * Tk * ldk + hh * DK, kj, Tk, ldk, 32 * ks + 8 * g);
        vf[ks] = att_load8(v + (long)b * Tk * ldv + hh * DK, kj, Tk, ldv, 32 * ks + 8 * g);
    }
    const bool kmask = kok && key_mask && key_mask[(long)b * Tk + kj];
    f32x4_t ak[NT], av[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ak[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        av[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    }

    AttStage<DK> stQ, stO;
    float rl = 0.f, rd[NP];
    const int nblk = (Tq + AM_ROWS - 1) / AM_ROWS;
    stQ.load(qb, ldq, 0, Tq, tid);
    stO.load(ob, ldo, 0, Tq, tid);
    if (tid < AM_ROWS) rl = tid < Tq ? lb[tid] : INFINITY;      // queries past the end: p = exp(-inf) = 0
#pragma unroll
    for (int i = 0; i < NP; ++i) rd[i] = att_stage_rowdot<DK>(stO, yb, ldo, 0, Tq, tid, i);
    stQ.store(sQ[0], tid);
    stO.store(sO[0], tid);
    if (tid < AM_ROWS) sL[0][tid] = rl;
#pragma unroll
    for (int i = 0; i < NP; ++i)
        if ((tid + 256 * i) % NC == 0) sD[0][(tid + 256 * i) / NC] = rd[i];
    __syncthreads();

    for (int blk = 0; blk < nblk; ++blk) {
        const int cur = blk & 1, i0 = blk * AM_ROWS;
        const bool more = blk + 1 < nblk;
        if (more) {
            stQ.load(qb, ldq, i0 + AM_ROWS, Tq, tid);
            stO.load(ob, ldo, i0 + AM_ROWS, Tq, tid);
            if (tid < AM_ROWS) {
                const int i = i0 + AM_ROWS + tid;
                rl = i < Tq ? lb[i] : INFINITY;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) rd[i] = att_stage_rowdot<DK>(stO, yb, ldo, i0 + AM_ROWS, Tq, tid, i);
        }
        // s[it][r] = q(i0 + 16 it + 4 g + r) . k(kj); dp likewise with dout . v -- 32 streamed rows at a
        // time (two 16-row tiles: the contraction of one step of the second pair of products)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f32x4_t s[2], dp[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int it = 2 * kk + h2;
                s[h2] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                dp[h2] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_frag_rows<DK>(sQ[cur], 16 * it, ks, lane), kf[ks],
                                                                    s[h2], 0, 0, 0);
                    dp[h2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(att_frag_rows<DK>(sO[cur], 16 * it, ks, lane), vf[ks],
                                                                     dp[h2], 0, 0, 0);
                }
            }
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int it = 2 * kk + h2;
                const float4 l4 = *(const float4*)&sL[cur][16 * it + 4 * g];
                const float4 d4 = *(const float4*)&sD[cur][16 * it + 4 * g];
                const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qi = i0 + 16 * it + 4 * g + r;
                    const float sv = s[h2][r] * scale + ((kmask || (CAUSAL && kj > qi + causal - 1)) ? -1e9f : 0.f);
                    const float p = __expf(sv - lv[r]);
                    s[h2][r] = p;
                    dp[h2][r] = p * (dp[h2][r] - dl[r]) * scale;    // dS
                }
            }
            // dv^T[d][key] += sum_query dout[query][d] p[query][key]; dk^T likewise with q and dS
            const bf16x8_t pf = att_pack(s[0], s[1]);
            const bf16x8_t dsf = att_pack(dp[0], dp[1]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                av[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    att_frag_tr<DK>(sO[cur], 32 * kk, 32 * kk + 16, 16 * nt, lane), pf, av[nt], 0, 0, 0);
                ak[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    att_frag_tr<DK>(sQ[cur], 32 * kk, 32 * kk + 16, 16 * nt, lane), dsf, ak[nt], 0, 0, 0);
            }
        }
        if (more) {
            stQ.store(sQ[cur ^ 1], tid);
            stO.store(sO[cur ^ 1], tid);
            if (tid < AM_ROWS) sL[cur ^ 1][tid] = rl;
#pragma unroll
            for (int i = 0; i < NP; ++i)
                if ((tid + 256 * i) % NC == 0) sD[cur ^ 1][(tid + 256 * i) / NC] = rd[i];
        }
        __syncthreads();
    }
    if (kok) {
        bf16_t* o1 = dk_ + ((long)b * Tk + kj) * lddk + hh * DK + 4 * g;
        bf16_t* o2 = dv + ((long)b * Tk + kj) * lddv + hh * DK + 4 * g;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bf16x4_t w1 = {(bf16_t)ak[nt][0], (bf16_t)ak[nt][1], (bf16_t)ak[nt][2], (bf16_t)ak[nt][3]};
            bf16x4_t w2 = {(bf16_t)av[nt][0], (bf16_t)av[nt][1], (bf16_t)av[nt][2], (bf16_t)av[nt][3]};
            *(bf16x4_t*)(o1 + 16 * nt) = w1;
            *(bf16x4_t*)(o2 + 16 * nt) = w2;
        }
    }
}

// One launch for the whole backward pass of the attention: the workgroups of a head are its dK/dV
// blocks (first: the longer role) and its dQ blocks (64 * RS queries each), side by side on one XCD.
// Three waves per SIMD for head sizes up to 64: at T = 250 the 4 + 2 workgroups of each of 128 heads
// are then all on the chip at once (at two waves per SIMD the launch ran as two rounds).
#ifndef AM_BWD_MIN_WGS
#define AM_BWD_MIN_WGS 3
#endif
template <int DK, bool CAUSAL, int RS>
__global__ __launch_bounds__(256, DK <= 64 ? AM_BWD_MIN_WGS : 1) void mha_bwd_mfma_kernel(
    const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ldq, int ldk, int ldv,
    const uint8_t* __restrict__ key_mask, int causal, int h, int Tq, int Tk, float scale,
    const bf16_t* __restrict__ out, const bf16_t* __restrict__ dout, int ldo, const float* __restrict__ lse,
    float* __restrict__ delta, bf16_t* __restrict__ dq, bf16_t* __restrict__ dk_, bf16_t* __restrict__ dv, int lddq,
    int lddk, int lddv, int heads, int only) {
    __shared__ __attribute__((aligned(16))) AttBwdLds<DK> sm;
    // (``only``: 1 = the dK/dV blocks alone, 2 = the dQ blocks alone -- for measurements)
    const int nkb = only == 2 ? 0 : (Tk + AM_ROWS - 1) / AM_ROWS;
    const int nqb = only == 1 ? 0 : (Tq + AM_ROWS * RS - 1) / (AM_ROWS * RS);
    int w, head;
    if (!att_xcd_map(nkb + nqb, heads, w, head)) return;
    if (w < nkb)
        mha_bwd_dkv_body<DK, CAUSAL>(sm, w, head, q, k, v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, out, dout,
                                     ldo, lse, dk_, dv, lddk, lddv);
    else
        mha_bwd_dq_body<DK, CAUSAL, RS>(sm, w - nkb, head, q, k, v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale,
                                        out, dout, ldo, lse, delta, dq, lddq);
}
