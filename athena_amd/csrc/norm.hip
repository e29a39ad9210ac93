// LayerNormalization (with fused residual add + dropout), BatchNormalization + relu6,
// column sums.  All arithmetic in fp32 (fp64 for the batch statistics).
#include "common.h"
#include <initializer_list>

// ---------------------------------------------------------------- 8-wide vector access
template <typename T> struct Vec8;
template <> struct Vec8<float> {
    static __device__ __forceinline__ void load(const float* p, float* v) {
        float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
        *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct Vec8<bf16_t> {
    static __device__ __forceinline__ void load(const bf16_t* p, float* v) {
        bf16x8_t a = *(const bf16x8_t*)p;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
    }
    static __device__ __forceinline__ void store(bf16_t* p, const float* v) {
        bf16x8_t a;
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = (bf16_t)v[i];
        *(bf16x8_t*)p = a;
    }
};

// ================================================================ LayerNorm
// one wave per row, 4 rows per 256-thread block
template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, int rows,
                                                                int d, uint32_t thresh, float dscale,
                                                                const uint64_t* rng_state, uint32_t site,
                                                                T* __restrict__ z, T* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    const long base = (long)row * d;
    float s = 0.f;
    for (int c = lane; c < d; c += 64) {
        float v = to_f32<T>(a[base + c]);
        if (b) {
            float bv = to_f32<T>(b[base + c]);
            if (thresh) bv *= dropout_mult(key, (uint64_t)(base + c), thresh, dscale);
            v += bv;
        }
        // the stored z is what the backward pass reads; round-trip it so both passes see one value
        T zt = from_f32<T>(v);
        z[base + c] = zt;
        s += to_f32<T>(zt);
    }
    s = wave_sum(s);
    const float mu = s / d;
    float q = 0.f;
    for (int c = lane; c < d; c += 64) {
        float dv = to_f32<T>(z[base + c]) - mu;
        q += dv * dv;
    }
    q = wave_sum(q);
    const float rs = rsqrtf(q / d + eps);
    for (int c = lane; c < d; c += 64) {
        float xh = (to_f32<T>(z[base + c]) - mu) * rs;
        y[base + c] = from_f32<T>(xh * gamma[c] + beta[c]);
    }
    if (lane == 0) {
        mean[row] = mu;
        rstd[row] = rs;
    }
}

// block = 256 threads = 4 waves; each wave walks rows r = blockIdx*ROWS_PER_BLOCK + w, +4, ...
// per-lane register partials of dgamma/dbeta for columns lane + 64*v, v < VPL
#define LN_ROWS_PER_BLOCK 32
template <typename T, int VPL>
__global__ __launch_bounds__(256) void add_layernorm_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, int rows, int d,
                                                                uint32_t thresh, float dscale,
                                                                const uint64_t* rng_state, uint32_t site,
                                                                int accumulate_da, T* __restrict__ da,
                                                                T* __restrict__ db, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
    __shared__ float red[4][64 * VPL];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    float g[VPL], pg[VPL], pb[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
        int c = lane + 64 * v;
        g[v] = c < d ? gamma[c] : 0.f;
        pg[v] = 0.f;
        pb[v] = 0.f;
    }
    const int r0 = blockIdx.x * LN_ROWS_PER_BLOCK;
    const int r1 = min(rows, r0 + LN_ROWS_PER_BLOCK);
    for (int row = r0 + w; row < r1; row += 4) {
        const long base = (long)row * d;
        const float mu = mean[row], rs = rstd[row];
        float xh[VPL], gy[VPL];
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            int c = lane + 64 * v;
            float dyv = 0.f, x = 0.f;
            if (c < d) {
                dyv = to_f32<T>(dy[base + c]);
                x = (to_f32<T>(z[base + c]) - mu) * rs;
            }
            xh[v] = x;
            gy[v] = dyv * g[v];
            pg[v] += dyv * x;
            pb[v] += dyv;
            c1 += gy[v];
            c2 += gy[v] * x;
        }
        c1 = wave_sum(c1) / d;
        c2 = wave_sum(c2) / d;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            int c = lane + 64 * v;
            if (c < d) {
                float dz = rs * (gy[v] - c1 - xh[v] * c2);
                float dav = dz;
                if (accumulate_da) dav += to_f32<T>(da[base + c]);
                da[base + c] = from_f32<T>(dav);
                if (db) {
                    float m = thresh ? dropout_mult(key, (uint64_t)(base + c), thresh, dscale) : 1.f;
                    db[base + c] = from_f32<T>(dz * m);
                }
            }
        }
    }
    // fold the 4 waves, one atomic per column per block
#pragma unroll
    for (int v = 0; v < VPL; ++v) red[w][lane + 64 * v] = pg[v];
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dgamma + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
    __syncthreads();
#pragma unroll
    for (int v = 0; v < VPL; ++v) red[w][lane + 64 * v] = pb[v];
    __syncthreads();
    for (int c = threadIdx.x; c < d; c += 256) atomicAdd(dbeta + c, red[0][c] + red[1][c] + red[2][c] + red[3][c]);
}


// ---------------------------------------------------------------- vector kernels (d % 8 == 0, d <= 512)
// A row is spread over a group of G lanes (G = power of two >= d / 8), one 16-byte piece of 8
// elements per lane, so a row is read and written exactly once with full-width accesses and all
// statistics are reduced with lane shuffles inside the group.
__device__ __forceinline__ float group_sum(float v, int G) {
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void add_layernorm_fwd_vec_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float eps,
                                                                    int rows, int d, int G, uint32_t thresh,
                                                                    float dscale, const uint64_t* rng_state,
                                                                    uint32_t site, T* __restrict__ z,
                                                                    T* __restrict__ y, float* __restrict__ mean,
                                                                    float* __restrict__ rstd,
                                                                    const float* __restrict__ a32,
                                                                    float* __restrict__ y32) {
    const int rpb = 256 / G;                         // rows per block
    const int gl = threadIdx.x % G;
    const int row = blockIdx.x * rpb + threadIdx.x / G;
    const bool live = row < rows && gl * 8 < d;
    const long base = (long)(row < rows ? row : rows - 1) * d + (gl * 8 < d ? gl * 8 : 0);
    // wide: the residual stream passes in fp32 -- a comes from its fp32 copy, statistics and y are
    // worked out from the unrounded sum, and only the stored z and y are rounded
    const bool wide = a32 != nullptr || y32 != nullptr;
    float v[8];
    if (a32) Vec8<float>::load(a32 + base, v);
    else Vec8<T>::load(a + base, v);
    if (b) {
        float bv[8];
        Vec8<T>::load(b + base, bv);
        if (thresh) {
            const uint64_t key = rng_key(rng_state, site);
            float dm[8];
            dropout_mult8(key, (uint64_t)base, thresh, dscale, dm);
#pragma unroll
            for (int i = 0; i < 8; ++i) bv[i] *= dm[i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += bv[i];
    }
    // the stored z is what the backward pass reads; round-trip it so both passes see one value
    float s = 0.f;
    if (live) Vec8<T>::store(z + base, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = live ? (wide ? v[i] : to_f32<T>(from_f32<T>(v[i]))) : 0.f;
        s += v[i];
    }
    const float mu = group_sum(s, G) / d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float dv = live ? v[i] - mu : 0.f;
        q += dv * dv;
    }
    const float rs = rsqrtf(group_sum(q, G) / d + eps);
    if (live) {
        float gm[8], bt[8], o[8];
        Vec8<float>::load(gamma + gl * 8, gm);
        Vec8<float>::load(beta + gl * 8, bt);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (v[i] - mu) * rs * gm[i] + bt[i];
        Vec8<T>::store(y + base, o);
        if (y32) Vec8<float>::store(y32 + base, o);
        if (gl == 0) {
            mean[row] = mu;
            rstd[row] = rs;
        }
    }
}

#define LNV_ROWS_PER_BLOCK 32
// sum over the G lanes of a group, G a compile-time power of two (the shuffles unroll)
template <int G>
__device__ __forceinline__ float group_sum_c(float v) {
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int G>
__global__ __launch_bounds__(256) void add_layernorm_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                                    const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd,
                                                                    const float* __restrict__ gamma, int rows, int d,
                                                                    uint32_t thresh, float dscale,
                                                                    const uint64_t* rng_state, uint32_t site,
                                                                    int accumulate_da, T* __restrict__ da,
                                                                    T* __restrict__ db, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta,
                                                                    float* __restrict__ slabs) {
    __shared__ float red[4][2][512];                 // [wave][dgamma | dbeta][column]
    constexpr int rpp = 256 / G;                     // rows per pass of the block
    const int gl = threadIdx.x % G, gi = threadIdx.x / G;
    const bool col_ok = gl * 8 < d;
    const int c0 = col_ok ? gl * 8 : 0;
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    float g[8], pg[8], pb[8];
    Vec8<float>::load(gamma + c0, g);
#pragma unroll
    for (int i = 0; i < 8; ++i) { pg[i] = 0.f; pb[i] = 0.f; }
    const int r0 = blockIdx.x * LNV_ROWS_PER_BLOCK;
    const int r1 = min(rows, r0 + LNV_ROWS_PER_BLOCK);
    // four row passes per trip, all loads issued before any arithmetic (a workgroup per CU has
    // to keep enough bytes in flight on its own)
    constexpr int NU = 4;
    for (int rb = r0; rb < r1; rb += NU * rpp) {
        int rowv[NU];
        bool livev[NU];
        long basev[NU];
        float dyv[NU][8], x[NU][8], muv[NU], rsv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            rowv[u] = rb + u * rpp + gi;
            livev[u] = rowv[u] < r1 && col_ok;
            const int rc = rowv[u] < r1 ? rowv[u] : r1 - 1;
            basev[u] = (long)rc * d + c0;
            Vec8<T>::load(dy + basev[u], dyv[u]);
            Vec8<T>::load(z + basev[u], x[u]);
            muv[u] = mean[rc];
            rsv[u] = rstd[rc];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bool live = livev[u];
            const long base = basev[u];
            const float mu = muv[u], rs = rsv[u];
            float gy[8];
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                dyv[u][i] = live ? dyv[u][i] : 0.f;
                x[u][i] = live ? (x[u][i] - mu) * rs : 0.f;
                gy[i] = dyv[u][i] * g[i];
                pg[i] += dyv[u][i] * x[u][i];
                pb[i] += dyv[u][i];
                c1 += gy[i];
                c2 += gy[i] * x[u][i];
            }
            c1 = group_sum_c<G>(c1) / d;
            c2 = group_sum_c<G>(c2) / d;
            if (live) {
                float dz[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) dz[i] = rs * (gy[i] - c1 - x[u][i] * c2);
                if (accumulate_da) {
                    float old[8];
                    Vec8<T>::load(da + base, old);
#pragma unroll
                    for (int i = 0; i < 8; ++i) old[i] += dz[i];
                    Vec8<T>::store(da + base, old);
                } else {
                    Vec8<T>::store(da + base, dz);
                }
                if (db) {
                    if (thresh) {
                        float dm[8];
                        dropout_mult8(key, (uint64_t)base, thresh, dscale, dm);
#pragma unroll
                        for (int i = 0; i < 8; ++i) dz[i] *= dm[i];
                    }
                    Vec8<T>::store(db + base, dz);
                }
            }
        }
    }
    // fold the row groups: inside a wave by shuffles (lanes with equal gl), across the four waves in LDS
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float a = pg[i], b = pb[i];
#pragma unroll
        for (int o = G; o < 64; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
        if (lane < G && col_ok) {
            red[w][0][c0 + i] = a;
            red[w][1][c0 + i] = b;
        }
    }
    __syncthreads();
    // G > 64 never happens (d <= 512); for G == 64 each wave holds one row group
    if (slabs) {
        // one slab [2][d] per workgroup, summed by ln_fold_kernel: no same-address atomic chains
        float* dst = slabs + (long)blockIdx.x * 2 * d;
        for (int c = threadIdx.x; c < 2 * d; c += 256) {
            const int which = c >= d, col = c - which * d;
            dst[c] = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
        }
        return;
    }
    for (int c = threadIdx.x; c < 2 * d; c += 256) {
        const int which = c >= d, col = c - which * d;
        const float t = (red[0][which][col] + red[1][which][col]) + (red[2][which][col] + red[3][which][col]);
        atomicAdd((which ? dbeta : dgamma) + col, t);
    }
}

// dgamma / dbeta = column sums of the slabs.  Workgroup = 16 entries x 16 parts: part p adds slabs
// p, p + 16, ... (four loads in flight), the 16 partial sums meet in LDS in a fixed order.
__global__ __launch_bounds__(256) void ln_fold_kernel(const float* __restrict__ slabs, int nblk, int d,
                                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[16][17];
    const int part = threadIdx.x >> 4, le = threadIdx.x & 15;
    const int e = blockIdx.x * 16 + le;              // entry of the [2][d] slab
    const long stride = 2L * d;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < 2 * d) {
        int b = part;
        for (; b + 48 < nblk; b += 64) {
            a0 += slabs[(long)b * stride + e];
            a1 += slabs[(long)(b + 16) * stride + e];
            a2 += slabs[(long)(b + 32) * stride + e];
            a3 += slabs[(long)(b + 48) * stride + e];
        }
        for (; b < nblk; b += 16) a0 += slabs[(long)b * stride + e];
    }
    red[part][le] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && e < 2 * d) {
        float t = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < 16; ++p2) t += red[p2][le];
        if (e < d) dgamma[e] = t;
        else dbeta[e - d] = t;
    }
}

// the same fold for several LayerNorms in one launch (blockIdx.y picks the item)
struct LnFoldItem {
    const float* slabs;
    float* dgamma;
    float* dbeta;
    int32_t nblk;
    int32_t d;
};
static_assert(sizeof(LnFoldItem) == sizeof(athena_ln_fold_item), "LnFoldItem mirrors athena_ln_fold_item");

__global__ __launch_bounds__(256) void ln_fold_batched_kernel(const LnFoldItem* __restrict__ items) {
    __shared__ float red[16][17];
    const LnFoldItem it = items[blockIdx.y];
    const int part = threadIdx.x >> 4, le = threadIdx.x & 15;
    const int e = blockIdx.x * 16 + le;
    const int d = it.d, nblk = it.nblk;
    if (blockIdx.x * 16 >= 2 * d) return;            // uniform per workgroup
    const long stride = 2L * d;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (e < 2 * d) {
        int b = part;
        for (; b + 48 < nblk; b += 64) {
            a0 += it.slabs[(long)b * stride + e];
            a1 += it.slabs[(long)(b + 16) * stride + e];
            a2 += it.slabs[(long)(b + 32) * stride + e];
            a3 += it.slabs[(long)(b + 48) * stride + e];
        }
        for (; b < nblk; b += 16) a0 += it.slabs[(long)b * stride + e];
    }
    red[part][le] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (part == 0 && e < 2 * d) {
        float t = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < 16; ++p2) t += red[p2][le];
        if (e < d) it.dgamma[e] = t;
        else it.dbeta[e - d] = t;
    }
}

extern "C" int athena_ln_fold_batched(const athena_ln_fold_item* items, int count, int max_d,
                                      athena_stream_t stream) {
    ATHENA_REQUIRE(items && count > 0 && count <= 65535 && max_d > 0, ATHENA_E_SHAPE, "athena_ln_fold_batched: bad arguments");
    ATHENA_LAUNCH(ln_fold_batched_kernel, dim3((2 * max_d + 15) / 16, count), dim3(256), 0, (hipStream_t)stream,
                       (const LnFoldItem*)items);
    ATHENA_CHECK_LAUNCH("athena_ln_fold_batched");
    return ATHENA_OK;
}

static inline int ln_group(int d) {
    int G = 1;
    while (G * 8 < d) G <<= 1;
    return G;
}
static inline bool ln_vec_ok(int d, std::initializer_list<const void*> ptrs) {
    if (d % 8 || d > 512) return false;
    for (const void* p : ptrs)
        if (p && !athena_aligned16(p)) return false;
    return true;
}

static int add_layernorm_fwd_any(const void* a, const void* b, const float* gamma, const float* beta, float eps, int rows,
                                 int d, int dtype, float drop_p, const uint64_t* rng_state, uint32_t rng_site, void* z,
                                 void* y, float* mean, float* rstd, const float* a32, float* y32,
                                 athena_stream_t stream);

extern "C" int athena_add_layernorm_fwd(const void* a, const void* b, const float* gamma, const float* beta,
                                        float eps, int rows, int d, int dtype, float drop_p,
                                        const uint64_t* rng_state, uint32_t rng_site, void* z, void* y,
                                        float* mean, float* rstd, athena_stream_t stream) {
    return add_layernorm_fwd_any(a, b, gamma, beta, eps, rows, d, dtype, drop_p, rng_state, rng_site, z, y, mean, rstd,
                                 nullptr, nullptr, stream);
}

extern "C" int athena_add_layernorm_fwd_wide(const void* a, const float* a_f32, const void* b, const float* gamma,
                                             const float* beta, float eps, int rows, int d, int dtype, float drop_p,
                                             const uint64_t* rng_state, uint32_t rng_site, void* z, void* y,
                                             float* y_f32, float* mean, float* rstd, athena_stream_t stream) {
    ATHENA_REQUIRE(a_f32 || y_f32, ATHENA_E_SHAPE, "add_layernorm_fwd_wide: neither fp32 buffer given");
    ATHENA_REQUIRE((dtype == ATHENA_BF16 || dtype == ATHENA_F32) && ln_vec_ok(d, {a, b, gamma, beta, z, y, a_f32, y_f32}),
                   ATHENA_E_UNSUPPORTED, "add_layernorm_fwd_wide: needs d %% 8 == 0, d <= 512 and 16-byte aligned operands");
    return add_layernorm_fwd_any(a, b, gamma, beta, eps, rows, d, dtype, drop_p, rng_state, rng_site, z, y, mean, rstd,
                                 a_f32, y_f32, stream);
}

static int add_layernorm_fwd_any(const void* a, const void* b, const float* gamma, const float* beta, float eps, int rows,
                                 int d, int dtype, float drop_p, const uint64_t* rng_state, uint32_t rng_site, void* z,
                                 void* y, float* mean, float* rstd, const float* a32, float* y32,
                                 athena_stream_t stream) {
    ATHENA_REQUIRE(a && gamma && beta && z && y && mean && rstd, ATHENA_E_SHAPE, "add_layernorm_fwd: null operand");
    ATHENA_REQUIRE(rows > 0 && d > 0, ATHENA_E_SHAPE, "add_layernorm_fwd: bad shape rows=%d d=%d", rows, d);
    ATHENA_REQUIRE(drop_p == 0.f || (rng_state && drop_p < 1.f && b), ATHENA_E_SHAPE, "add_layernorm_fwd: dropout args");
    uint32_t th = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    if ((dtype == ATHENA_BF16 || dtype == ATHENA_F32) && ln_vec_ok(d, {a, b, gamma, beta, z, y})) {
        const int G = ln_group(d);
        dim3 vgrid((rows + 256 / G - 1) / (256 / G));
        if (dtype == ATHENA_BF16)
            ATHENA_LAUNCH(add_layernorm_fwd_vec_kernel<bf16_t>, vgrid, dim3(256), 0, (hipStream_t)stream,
                               (const bf16_t*)a, (const bf16_t*)b, gamma, beta, eps, rows, d, G, th, ds, rng_state,
                               rng_site, (bf16_t*)z, (bf16_t*)y, mean, rstd, a32, y32);
        else
            ATHENA_LAUNCH(add_layernorm_fwd_vec_kernel<float>, vgrid, dim3(256), 0, (hipStream_t)stream,
                               (const float*)a, (const float*)b, gamma, beta, eps, rows, d, G, th, ds, rng_state,
                               rng_site, (float*)z, (float*)y, mean, rstd, a32, y32);
        ATHENA_CHECK_LAUNCH("add_layernorm_fwd");
        return ATHENA_OK;
    }
    ATHENA_REQUIRE(!a32 && !y32, ATHENA_E_UNSUPPORTED, "add_layernorm_fwd_wide: shape not covered");
    dim3 grid((rows + 3) / 4);
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(add_layernorm_fwd_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)a,
                           (const bf16_t*)b, gamma, beta, eps, rows, d, th, ds, rng_state, rng_site, (bf16_t*)z,
                           (bf16_t*)y, mean, rstd);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(add_layernorm_fwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)a,
                           (const float*)b, gamma, beta, eps, rows, d, th, ds, rng_state, rng_site, (float*)z,
                           (float*)y, mean, rstd);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "add_layernorm_fwd: dtype");
    ATHENA_CHECK_LAUNCH("add_layernorm_fwd");
    return ATHENA_OK;
}

template <typename T>
static int launch_ln_bwd(const void* dy, const void* z, const float* mean, const float* rstd, const float* gamma,
                         int rows, int d, uint32_t th, float ds, const uint64_t* rng_state, uint32_t site,
                         int acc, void* da, void* db, float* dgamma, float* dbeta, float* slabs,
                         hipStream_t stream) {
    if (!dgamma) {
        // partial sums only: the caller folds the slabs later (athena_ln_fold_batched)
        ATHENA_REQUIRE(slabs && ln_vec_ok(d, {dy, z, gamma, da, db}), ATHENA_E_UNSUPPORTED,
                       "add_layernorm_bwd: deferred column sums need the workspace and d %% 8 == 0, d <= 512");
    }
    if (ln_vec_ok(d, {dy, z, gamma, da, db})) {
        const int nblk = (rows + LNV_ROWS_PER_BLOCK - 1) / LNV_ROWS_PER_BLOCK;
        if (!slabs) {
            ATHENA_REQUIRE(hipMemsetAsync(dgamma, 0, sizeof(float) * d, stream) == hipSuccess &&
                               hipMemsetAsync(dbeta, 0, sizeof(float) * d, stream) == hipSuccess,
                           ATHENA_E_LAUNCH, "add_layernorm_bwd: memset failed");
        }
#define LNV_BWD(GG)                                                                                                  \
    ATHENA_LAUNCH((add_layernorm_bwd_vec_kernel<T, GG>), dim3(nblk), dim3(256), 0, stream, (const T*)dy,      \
                       (const T*)z, mean, rstd, gamma, rows, d, th, ds, rng_state, site, acc, (T*)da, (T*)db, dgamma, \
                       dbeta, slabs)
        switch (ln_group(d)) {
            case 1: LNV_BWD(1); break;
            case 2: LNV_BWD(2); break;
            case 4: LNV_BWD(4); break;
            case 8: LNV_BWD(8); break;
            case 16: LNV_BWD(16); break;
            case 32: LNV_BWD(32); break;
            default: LNV_BWD(64); break;
        }
#undef LNV_BWD
        if (slabs && dgamma)
            ATHENA_LAUNCH(ln_fold_kernel, dim3((2 * d + 15) / 16), dim3(256), 0, stream, (const float*)slabs, nblk, d,
                               dgamma, dbeta);
        ATHENA_CHECK_LAUNCH("add_layernorm_bwd");
        return ATHENA_OK;
    }
    if (dbeta == dgamma + d) {       // adjacent (the usual parameter layout): one fill
        ATHENA_REQUIRE(hipMemsetAsync(dgamma, 0, sizeof(float) * 2 * d, stream) == hipSuccess, ATHENA_E_LAUNCH,
                       "add_layernorm_bwd: memset failed");
    } else {
        ATHENA_REQUIRE(hipMemsetAsync(dgamma, 0, sizeof(float) * d, stream) == hipSuccess &&
                           hipMemsetAsync(dbeta, 0, sizeof(float) * d, stream) == hipSuccess,
                       ATHENA_E_LAUNCH, "add_layernorm_bwd: memset failed");
    }
    dim3 grid((rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK);
#define LN_BWD_CASE(V)                                                                                          \
    ATHENA_LAUNCH((add_layernorm_bwd_kernel<T, V>), grid, dim3(256), 0, stream, (const T*)dy, (const T*)z, \
                       mean, rstd, gamma, rows, d, th, ds, rng_state, site, acc, (T*)da, (T*)db, dgamma, dbeta)
    if (d <= 64 * 4) LN_BWD_CASE(4);
    else if (d <= 64 * 8) LN_BWD_CASE(8);
    else if (d <= 64 * 16) LN_BWD_CASE(16);
    else LN_BWD_CASE(32);
#undef LN_BWD_CASE
    ATHENA_CHECK_LAUNCH("add_layernorm_bwd");
    return ATHENA_OK;
}

extern "C" int athena_add_layernorm_bwd_slabs(int rows, int d) {
    if (rows <= 0 || d <= 0 || d % 8 != 0 || d > 512) return 0;
    return (rows + LNV_ROWS_PER_BLOCK - 1) / LNV_ROWS_PER_BLOCK;
}

extern "C" int64_t athena_add_layernorm_bwd_workspace_bytes(int rows, int d) {
    if (rows <= 0 || d <= 0) return 0;
    return (int64_t)sizeof(float) * ((rows + LNV_ROWS_PER_BLOCK - 1) / LNV_ROWS_PER_BLOCK) * 2 * d;
}

extern "C" int athena_add_layernorm_bwd(const void* dy, const void* z, const float* mean, const float* rstd,
                                        const float* gamma, int rows, int d, int dtype, float drop_p,
                                        const uint64_t* rng_state, uint32_t rng_site, int accumulate_da, void* da,
                                        void* db, float* dgamma, float* dbeta, void* workspace,
                                        int64_t workspace_bytes, athena_stream_t stream) {
    ATHENA_REQUIRE(dy && z && mean && rstd && gamma && da && ((dgamma && dbeta) || (!dgamma && !dbeta)), ATHENA_E_SHAPE, "add_layernorm_bwd: null operand");
    ATHENA_REQUIRE(rows > 0 && d > 0 && d <= 2048, ATHENA_E_SHAPE, "add_layernorm_bwd: bad shape rows=%d d=%d (d <= 2048)", rows, d);
    ATHENA_REQUIRE(drop_p == 0.f || (rng_state && drop_p < 1.f && db), ATHENA_E_SHAPE, "add_layernorm_bwd: dropout args");
    uint32_t th = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    hipStream_t s = (hipStream_t)stream;
    float* slabs = nullptr;
    if (workspace && athena_aligned16(workspace) &&
        workspace_bytes >= athena_add_layernorm_bwd_workspace_bytes(rows, d))
        slabs = (float*)workspace;
    if (dtype == ATHENA_BF16)
        return launch_ln_bwd<bf16_t>(dy, z, mean, rstd, gamma, rows, d, th, ds, rng_state, rng_site, accumulate_da, da, db, dgamma, dbeta, slabs, s);
    if (dtype == ATHENA_F32)
        return launch_ln_bwd<float>(dy, z, mean, rstd, gamma, rows, d, th, ds, rng_state, rng_site, accumulate_da, da, db, dgamma, dbeta, slabs, s);
    ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "add_layernorm_bwd: dtype");
}

// ================================================================ column sums (bias gradients)
// threads are laid out as (row slot, 8-column chunk); a thread always owns the same chunk.
#define CS_ROWS_PER_BLOCK 256
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ X, int M, int N, int ld,
                                                     float* __restrict__ out) {
    __shared__ float red[256][9];
    const int nchunk = (N + 7) / 8;            // chunks per row
    const int cpb = min(nchunk, 256);          // chunks handled per block column group
    const int slots = 256 / cpb;               // row slots
    const int chunk = blockIdx.y * cpb + (threadIdx.x % cpb);
    const int slot = threadIdx.x / cpb;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const int r0 = blockIdx.x * CS_ROWS_PER_BLOCK;
    const int r1 = min(M, r0 + CS_ROWS_PER_BLOCK);
    if (chunk < nchunk && slot < slots) {
        const int c0 = chunk * 8;
        for (int r = r0 + slot; r < r1; r += slots) {
            const T* p = X + (long)r * ld + c0;
            if (c0 + 8 <= N) {
                float v[8];
                Vec8<T>::load(p, v);
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] += v[i];
            } else {
                for (int i = 0; i < 8 && c0 + i < N; ++i) acc[i] += to_f32<T>(p[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[i];
    __syncthreads();
    if (slot == 0 && chunk < nchunk) {
        const int c0 = chunk * 8;
        for (int i = 0; i < 8 && c0 + i < N; ++i) {
            float s = 0.f;
            for (int k = 0; k < slots; ++k) s += red[k * cpb + (threadIdx.x % cpb)][i];
            atomicAdd(out + c0 + i, s);
        }
    }
}

extern "C" int athena_colsum(const void* X, int dtype, int M, int N, int ld, float* out, athena_stream_t stream) {
    ATHENA_REQUIRE(X && out && M > 0 && N > 0 && ld >= N, ATHENA_E_SHAPE, "athena_colsum: bad arguments");
    ATHENA_REQUIRE(ld % 8 == 0 && athena_aligned16(X), ATHENA_E_ALIGN, "athena_colsum: ld %% 8 and 16-byte alignment required");
    hipStream_t s = (hipStream_t)stream;
    ATHENA_REQUIRE(hipMemsetAsync(out, 0, sizeof(float) * N, s) == hipSuccess, ATHENA_E_LAUNCH, "athena_colsum: memset failed");
    const int nchunk = (N + 7) / 8;
    const int cpb = nchunk < 256 ? nchunk : 256;
    // 256 % cpb may be non-zero: the surplus threads idle (slot >= slots)
    dim3 grid((M + CS_ROWS_PER_BLOCK - 1) / CS_ROWS_PER_BLOCK, (nchunk + cpb - 1) / cpb);
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)X, M, N, ld, out);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)X, M, N, ld, out);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_colsum: dtype");
    ATHENA_CHECK_LAUNCH("athena_colsum");
    return ATHENA_OK;
}

// ================================================================ BatchNorm (+relu6)
__global__ void bn_finalize_kernel(const double* __restrict__ stats, long rows, int C, float eps, float momentum,
                                   float* __restrict__ mean_rstd, float* __restrict__ moving_mean,
                                   float* __restrict__ moving_var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double n = (double)rows;
    const double mu = stats[c] / n;
    double var = stats[C + c] / n - mu * mu;
    if (var < 0) var = 0;
    mean_rstd[c] = (float)mu;
    mean_rstd[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (moving_mean) moving_mean[c] = moving_mean[c] * momentum + (float)mu * (1.f - momentum);
    if (moving_var) {
        const double unb = rows > 1 ? var * n / (n - 1.0) : var;
        moving_var[c] = moving_var[c] * momentum + (float)unb * (1.f - momentum);
    }
}

extern "C" int athena_bn_finalize(const double* colstats, long rows, int C, float eps, float momentum,
                                  float* mean_rstd, float* moving_mean, float* moving_var, athena_stream_t stream) {
    ATHENA_REQUIRE(colstats && mean_rstd && rows > 0 && C > 0, ATHENA_E_SHAPE, "athena_bn_finalize: bad arguments");
    ATHENA_LAUNCH(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, colstats, rows, C,
                       eps, momentum, mean_rstd, moving_mean, moving_var);
    ATHENA_CHECK_LAUNCH("athena_bn_finalize");
    return ATHENA_OK;
}

template <bool RELU6>
__device__ __forceinline__ float bn_act(float v) { return RELU6 ? relu6f(v) : v; }

// y = relu6(x * sc[c] + sh[c]) (or without the relu6: the BatchNormalization that follows each recurrent
// layer, reference athena/models/deep_speech.py:80) with sc = gamma*rstd, sh = beta - mean*gamma*rstd (training) or the
// moving-statistics equivalents (inference)
template <typename T, bool INFER, bool RELU6>
__global__ __launch_bounds__(256) void bn_relu6_apply_kernel(const T* __restrict__ x, const float* __restrict__ s0,
                                                             const float* __restrict__ s1,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, long nchunks,
                                                             int C, T* __restrict__ y) {
    const int cpr = C / 8;
    const long stride = (long)gridDim.x * blockDim.x;
    const long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    auto coeffs = [&](int c0, float (&sc)[8], float (&sh)[8], float (&be)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i;
            const float mu = s0[c];
            const float rs = INFER ? rsqrtf(s1[c] + eps) : s0[C + c];
            sc[i] = rs * gamma[c];
            sh[i] = mu;                     // (x - mean) first: no cancellation between large terms
            be[i] = beta[c];
        }
    };
    if (stride % cpr == 0) {
        // every piece this thread visits holds the same eight channels: their scale and shift stay
        // in registers and two pieces are in flight per trip
        float sc[8], sh[8], be[8];
        coeffs((int)(e0 % cpr) * 8, sc, sh, be);
        long e = e0;
        for (; e + stride < nchunks; e += 2 * stride) {
            float v0[8], v1[8];
            Vec8<T>::load(x + e * 8, v0);
            Vec8<T>::load(x + (e + stride) * 8, v1);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                v0[i] = bn_act<RELU6>(fmaf(v0[i] - sh[i], sc[i], be[i]));
                v1[i] = bn_act<RELU6>(fmaf(v1[i] - sh[i], sc[i], be[i]));
            }
            Vec8<T>::store(y + e * 8, v0);
            Vec8<T>::store(y + (e + stride) * 8, v1);
        }
        if (e < nchunks) {
            float v0[8];
            Vec8<T>::load(x + e * 8, v0);
#pragma unroll
            for (int i = 0; i < 8; ++i) v0[i] = bn_act<RELU6>(fmaf(v0[i] - sh[i], sc[i], be[i]));
            Vec8<T>::store(y + e * 8, v0);
        }
        return;
    }
    for (long e = e0; e < nchunks; e += stride) {
        float sc[8], sh[8], be[8], v[8];
        coeffs((int)(e % cpr) * 8, sc, sh, be);
        Vec8<T>::load(x + e * 8, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = bn_act<RELU6>(fmaf(v[i] - sh[i], sc[i], be[i]));
        Vec8<T>::store(y + e * 8, v);
    }
}

static int bn_apply_common(const void* x, const float* s0, const float* s1, const float* gamma, const float* beta,
                           float eps, long rows, int C, int dtype, void* y, bool infer, hipStream_t stream,
                           bool relu6 = true) {
    ATHENA_REQUIRE(x && s0 && gamma && beta && y && rows > 0 && C > 0, ATHENA_E_SHAPE, "athena_bn_relu6: bad arguments");
    ATHENA_REQUIRE(C % 8 == 0 && athena_aligned16(x) && athena_aligned16(y), ATHENA_E_ALIGN, "athena_bn_relu6: C %% 8 and 16-byte alignment required");
    const long nchunks = rows * (C / 8);
    long blocks = (nchunks + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks);
#define BN_APPLY(T, I, R) ATHENA_LAUNCH((bn_relu6_apply_kernel<T, I, R>), grid, dim3(256), 0, stream, (const T*)x, s0, s1, gamma, beta, eps, nchunks, C, (T*)y)
#define BN_APPLY_R(T, I) do { if (relu6) BN_APPLY(T, I, true); else BN_APPLY(T, I, false); } while (0)
    if (dtype == ATHENA_BF16) { if (infer) BN_APPLY_R(bf16_t, true); else BN_APPLY_R(bf16_t, false); }
    else if (dtype == ATHENA_F32) { if (infer) BN_APPLY_R(float, true); else BN_APPLY_R(float, false); }
    else ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_bn_relu6: dtype");
#undef BN_APPLY_R
#undef BN_APPLY
    ATHENA_CHECK_LAUNCH("athena_bn_relu6_apply");
    return ATHENA_OK;
}

extern "C" int athena_bn_relu6_apply(const void* x, const float* mean_rstd, const float* gamma, const float* beta,
                                     long rows, int C, int dtype, void* y, athena_stream_t stream) {
    return bn_apply_common(x, mean_rstd, nullptr, gamma, beta, 0.f, rows, C, dtype, y, false, (hipStream_t)stream);
}
extern "C" int athena_bn_relu6_infer(const void* x, const float* moving_mean, const float* moving_var,
                                     const float* gamma, const float* beta, float eps, long rows, int C, int dtype,
                                     void* y, athena_stream_t stream) {
    ATHENA_REQUIRE(moving_var, ATHENA_E_SHAPE, "athena_bn_relu6_infer: null moving_var");
    return bn_apply_common(x, moving_mean, moving_var, gamma, beta, eps, rows, C, dtype, y, true, (hipStream_t)stream);
}
extern "C" int athena_bn_apply(const void* x, const float* mean_rstd, const float* gamma, const float* beta, long rows,
                               int C, int dtype, void* y, athena_stream_t stream) {
    return bn_apply_common(x, mean_rstd, nullptr, gamma, beta, 0.f, rows, C, dtype, y, false, (hipStream_t)stream,
                           false);
}
extern "C" int athena_bn_infer(const void* x, const float* moving_mean, const float* moving_var, const float* gamma,
                               const float* beta, float eps, long rows, int C, int dtype, void* y,
                               athena_stream_t stream) {
    ATHENA_REQUIRE(moving_var, ATHENA_E_SHAPE, "athena_bn_infer: null moving_var");
    return bn_apply_common(x, moving_mean, moving_var, gamma, beta, eps, rows, C, dtype, y, true, (hipStream_t)stream,
                           false);
}

// reduce: slab[blockIdx.x][c] = sum dy', slab[blockIdx.x][C+c] = sum dy'*xhat over the block's rows
// (dy' = dy masked by the relu6 gate).  One fp32 slab per workgroup instead of atomics on the 2C
// totals: a few hundred workgroups adding to the same 2C addresses serialise there.
#define BN_ROWS_PER_BLOCK 512
template <typename T, bool RELU6>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta,
                                                            const float* __restrict__ mean_rstd, long rows, int C,
                                                            float* __restrict__ slabs) {
    __shared__ float red[256][17];
    const int cpr = C / 8;
    const int cpb = min(cpr, 256);
    const int slots = 256 / cpb;
    const int chunk = blockIdx.y * cpb + (threadIdx.x % cpb);
    const int slot = threadIdx.x / cpb;
    float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long r0 = (long)blockIdx.x * BN_ROWS_PER_BLOCK;
    const long r1 = min(rows, r0 + BN_ROWS_PER_BLOCK);
    if (chunk < cpr && slot < slots) {
        const int c0 = chunk * 8;
        float mu[8], rs[8], ga[8], be[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mu[i] = mean_rstd[c0 + i]; rs[i] = mean_rstd[C + c0 + i];
            ga[i] = gamma[c0 + i]; be[i] = beta[c0 + i];
        }
        auto add_row = [&](const float (&dv)[8], const float (&xv)[8]) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float xh = (xv[i] - mu[i]) * rs[i];
                const float yb = xh * ga[i] + be[i];        // the relu6 gate is recomputed, not read
                const float g = (!RELU6 || (yb > 0.f && yb < 6.f)) ? dv[i] : 0.f;
                a1[i] += g;
                a2[i] += g * xh;
            }
        };
        // four rows in flight per thread: a row is one 16-byte piece of each tensor, and the loop
        // would otherwise wait out a full memory latency per row
        long r = r0 + slot;
        for (; r + 3L * slots < r1; r += 4L * slots) {
            float dv[4][8], xv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long o = (r + (long)u * slots) * C + c0;
                Vec8<T>::load(dy + o, dv[u]);
                Vec8<T>::load(x + o, xv[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add_row(dv[u], xv[u]);
        }
        for (; r < r1; r += slots) {
            const long o = r * C + c0;
            float dv[8], xv[8];
            Vec8<T>::load(dy + o, dv);
            Vec8<T>::load(x + o, xv);
            add_row(dv, xv);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = a1[i]; red[threadIdx.x][8 + i] = a2[i]; }
    __syncthreads();
    if (slot == 0 && chunk < cpr) {
        const int c0 = chunk * 8;
        float* dst = slabs + (long)blockIdx.x * 2 * C;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < slots; ++k) {
                s1 += red[k * cpb + (threadIdx.x % cpb)][i];
                s2 += red[k * cpb + (threadIdx.x % cpb)][8 + i];
            }
            dst[c0 + i] = s1;
            dst[C + c0 + i] = s2;
        }
    }
}

// totals[e] = sum over the slabs, in fp64 and in a fixed order; workgroup = 64 entries x 16 waves
__global__ __launch_bounds__(1024) void bn_bwd_fold_kernel(const float* __restrict__ slabs, int nslab, int nent,
                                                           double* __restrict__ totals) {
    __shared__ double red[16][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    double a0 = 0.0, a1 = 0.0;
    if (e < nent) {
        int b = w;
        for (; b + 16 < nslab; b += 32) {
            a0 += (double)slabs[(long)b * nent + e];
            a1 += (double)slabs[(long)(b + 16) * nent + e];
        }
        if (b < nslab) a0 += (double)slabs[(long)b * nent + e];
    }
    red[w][lane] = a0 + a1;
    __syncthreads();
    if (w == 0 && e < nent) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][lane];
        totals[e] = t;
    }
}

template <typename T, bool RELU6>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ gamma,
                                                           const double* __restrict__ scratch, long rows, int C,
                                                           T* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta) {
    const int cpr = C / 8;
    const long nchunks = rows * cpr;
    const float inv_n = 1.f / (float)rows;
    const long stride = (long)gridDim.x * blockDim.x;
    const long e0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    // dx = k1 * (g - k2 - xhat * k3) with the relu6 gate recomputed from xhat
    struct Co { float mu[8], rs[8], ga[8], be[8], k1[8], k2[8], k3[8]; };
    auto coeffs = [&](int c0, Co& k) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = c0 + i;
            k.mu[i] = mean_rstd[c]; k.rs[i] = mean_rstd[C + c];
            k.ga[i] = gamma[c]; k.be[i] = beta[c];
            k.k1[i] = k.ga[i] * k.rs[i];
            k.k2[i] = (float)scratch[c] * inv_n;
            k.k3[i] = (float)scratch[C + c] * inv_n;
        }
    };
    auto piece = [&](const Co& k, const float (&dv)[8], const float (&xv)[8], float (&o)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float xh = (xv[i] - k.mu[i]) * k.rs[i];
            const float yb = xh * k.ga[i] + k.be[i];
            const float g = (!RELU6 || (yb > 0.f && yb < 6.f)) ? dv[i] : 0.f;
            o[i] = k.k1[i] * (g - k.k2[i] - xh * k.k3[i]);
        }
    };
    if (stride % cpr == 0) {
        // the same eight channels on every trip: coefficients in registers, two pieces in flight
        Co k;
        coeffs((int)(e0 % cpr) * 8, k);
        long e = e0;
        for (; e + stride < nchunks; e += 2 * stride) {
            float d0[8], x0[8], d1[8], x1[8], o0[8], o1[8];
            Vec8<T>::load(dy + e * 8, d0);
            Vec8<T>::load(x + e * 8, x0);
            Vec8<T>::load(dy + (e + stride) * 8, d1);
            Vec8<T>::load(x + (e + stride) * 8, x1);
            piece(k, d0, x0, o0);
            piece(k, d1, x1, o1);
            Vec8<T>::store(dx + e * 8, o0);
            Vec8<T>::store(dx + (e + stride) * 8, o1);
        }
        if (e < nchunks) {
            float d0[8], x0[8], o0[8];
            Vec8<T>::load(dy + e * 8, d0);
            Vec8<T>::load(x + e * 8, x0);
            piece(k, d0, x0, o0);
            Vec8<T>::store(dx + e * 8, o0);
        }
    } else {
        for (long e = e0; e < nchunks; e += stride) {
            Co k;
            coeffs((int)(e % cpr) * 8, k);
            float dv[8], xv[8], o[8];
            Vec8<T>::load(dy + e * 8, dv);
            Vec8<T>::load(x + e * 8, xv);
            piece(k, dv, xv, o);
            Vec8<T>::store(dx + e * 8, o);
        }
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            dbeta[c] = (float)scratch[c];
            dgamma[c] = (float)scratch[C + c];
        }
    }
}

extern "C" int64_t athena_bn_relu6_bwd_scratch_bytes(long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    const int64_t nslab = (rows + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK;
    return (int64_t)sizeof(double) * 2 * C + (int64_t)sizeof(float) * nslab * 2 * C;     // totals + slabs
}

static int bn_bwd_common(const char* name, const void* dy, const void* x, const float* mean_rstd, const float* gamma,
                         const float* beta, long rows, int C, int dtype, void* scratch, int64_t scratch_bytes,
                         void* dx, float* dgamma, float* dbeta, bool relu6, hipStream_t s) {
    ATHENA_REQUIRE(dy && x && beta && mean_rstd && gamma && scratch && dx && dgamma && dbeta && rows > 0 && C > 0,
                   ATHENA_E_SHAPE, "%s: bad arguments", name);
    ATHENA_REQUIRE(C % 8 == 0, ATHENA_E_ALIGN, "%s: C %% 8 required", name);
    ATHENA_REQUIRE(scratch_bytes >= athena_bn_relu6_bwd_scratch_bytes(rows, C) && athena_aligned16(scratch),
                   ATHENA_E_SHAPE, "%s: scratch too small (%lld bytes) or misaligned", name, (long long)scratch_bytes);
    ATHENA_REQUIRE(dtype == ATHENA_BF16 || dtype == ATHENA_F32, ATHENA_E_UNSUPPORTED, "%s: dtype", name);
    const int cpr = C / 8;
    double* totals = (double*)scratch;
    float* slabs = (float*)(totals + 2 * C);
    const int cpb = cpr < 256 ? cpr : 256;
    const int nslab = (int)((rows + BN_ROWS_PER_BLOCK - 1) / BN_ROWS_PER_BLOCK);
    dim3 g1((unsigned)nslab, (cpr + cpb - 1) / cpb);
    const long nchunks = rows * cpr;
    long blocks = (nchunks + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    const int nent = 2 * C;
#define BN_BWD(T, R)                                                                                                    \
    do {                                                                                                                \
        ATHENA_LAUNCH((bn_bwd_reduce_kernel<T, R>), g1, dim3(256), 0, s, (const T*)dy, (const T*)x, gamma, beta,  \
                           mean_rstd, rows, C, slabs);                                                                  \
        ATHENA_LAUNCH(bn_bwd_fold_kernel, dim3((nent + 63) / 64), dim3(1024), 0, s, (const float*)slabs, nslab,    \
                           nent, totals);                                                                               \
        ATHENA_LAUNCH((bn_bwd_apply_kernel<T, R>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)dy,          \
                           (const T*)x, beta, mean_rstd, gamma, (const double*)totals, rows, C, (T*)dx, dgamma, dbeta); \
    } while (0)
    if (dtype == ATHENA_BF16) { if (relu6) BN_BWD(bf16_t, true); else BN_BWD(bf16_t, false); }
    else { if (relu6) BN_BWD(float, true); else BN_BWD(float, false); }
#undef BN_BWD
    ATHENA_CHECK_LAUNCH(name);
    return ATHENA_OK;
}

extern "C" int athena_bn_relu6_bwd(const void* dy, const void* x, const float* mean_rstd, const float* gamma,
                                   const float* beta, long rows, int C, int dtype, void* scratch,
                                   int64_t scratch_bytes, void* dx, float* dgamma, float* dbeta,
                                   athena_stream_t stream) {
    return bn_bwd_common("athena_bn_relu6_bwd", dy, x, mean_rstd, gamma, beta, rows, C, dtype, scratch, scratch_bytes,
                         dx, dgamma, dbeta, true, (hipStream_t)stream);
}

extern "C" int athena_bn_bwd(const void* dy, const void* x, const float* mean_rstd, const float* gamma,
                             const float* beta, long rows, int C, int dtype, void* scratch, int64_t scratch_bytes,
                             void* dx, float* dgamma, float* dbeta, athena_stream_t stream) {
    return bn_bwd_common("athena_bn_bwd", dy, x, mean_rstd, gamma, beta, rows, C, dtype, scratch, scratch_bytes, dx,
                         dgamma, dbeta, false, (hipStream_t)stream);
}

// ---------------------------------------------------------------- column statistics for BatchNorm
// stats[c] = sum over rows of x[r][c], stats[C + c] = sum of squares, in fp64 (the same statistics the
// convolution epilogue collects).  Workgroup = 64 column pieces x 4 row lanes; per-workgroup fp64
// partial sums are added with atomics onto the zero-filled totals.
template <typename T>
__global__ __launch_bounds__(256) void bn_colstats_kernel(const T* __restrict__ x, long rows, int C,
                                                          double* __restrict__ stats, int rows_per_block) {
    __shared__ double red[2][4][64 * 8];
    const int cpr = C / 8;
    const int piece = blockIdx.y * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    double s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(rows, r0 + rows_per_block);
    if (piece < cpr) {
        for (long r = r0 + rl; r < r1; r += 4) {
            float v[8];
            Vec8<T>::load(x + r * C + piece * 8, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                s1[i] += (double)v[i];
                s2[i] += (double)v[i] * (double)v[i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        red[0][rl][(threadIdx.x & 63) * 8 + i] = s1[i];
        red[1][rl][(threadIdx.x & 63) * 8 + i] = s2[i];
    }
    __syncthreads();
    if (rl == 0 && piece < cpr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = (threadIdx.x & 63) * 8 + i;
            const double a = red[0][0][o] + red[0][1][o] + red[0][2][o] + red[0][3][o];
            const double b = red[1][0][o] + red[1][1][o] + red[1][2][o] + red[1][3][o];
            atomicAdd(stats + piece * 8 + i, a);
            atomicAdd(stats + C + piece * 8 + i, b);
        }
    }
}

extern "C" int athena_bn_colstats(const void* x, long rows, int C, int dtype, double* colstats,
                                  athena_stream_t stream) {
    ATHENA_REQUIRE(x && colstats && rows > 0 && C > 0, ATHENA_E_SHAPE, "athena_bn_colstats: bad arguments");
    ATHENA_REQUIRE(C % 8 == 0 && athena_aligned16(x), ATHENA_E_ALIGN, "athena_bn_colstats: C %% 8 and 16-byte alignment required");
    ATHENA_REQUIRE(dtype == ATHENA_BF16 || dtype == ATHENA_F32, ATHENA_E_UNSUPPORTED, "athena_bn_colstats: dtype");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(colstats, 0, sizeof(double) * 2 * C, s);
    ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_bn_colstats: memset failed");
    const int cpr = C / 8;
    const int rpb = 256;
    dim3 grid((unsigned)((rows + rpb - 1) / rpb), (cpr + 63) / 64);
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(bn_colstats_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, rows, C, colstats, rpb);
    else
        ATHENA_LAUNCH(bn_colstats_kernel<float>, grid, dim3(256), 0, s, (const float*)x, rows, C, colstats, rpb);
    ATHENA_CHECK_LAUNCH("athena_bn_colstats");
    return ATHENA_OK;
}
