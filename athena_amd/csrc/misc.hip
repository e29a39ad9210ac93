// Front-end helpers: the few-input-channel 3x3/stride-2 convolution (the fbank input has one
// channel), positional encoding + dropout, embedding gather/scatter, casts.
#include "common.h"

static inline void same_pad2(int n, int* out, int* pad_before) {
    int o = (n + 1) / 2;
    int total = (o - 1) * 2 + 3 - n;
    if (total < 0) total = 0;
    *out = o;
    *pad_before = total / 2;
}

// ---------------------------------------------------------------- conv, small Cin, forward
// thread = (pixel slot, 8 output channels); weights for the thread's channels live in registers
// (9*Cin*8 floats, Cin <= 4 is looped outside to bound registers); per-channel statistics of the
// stored values are reduced per workgroup and added with fp64 atomics.
#define SC_PIX_PER_BLOCK 256
template <typename T>
__global__ __launch_bounds__(256) void conv_small_cin_fwd_kernel(const float* __restrict__ in,
                                                                 const float* __restrict__ w, T* __restrict__ out,
                                                                 int B, int H, int W, int Cin, int Cout, int Ho,
                                                                 int Wo, int ph, int pw, double* __restrict__ stats) {
    __shared__ float red[256][17];
    const int cpr = Cout / 8;
    const int cpb = min(cpr, 256);
    const int slots = 256 / cpb;
    const int chunk = blockIdx.y * cpb + (threadIdx.x % cpb);
    const int slot = threadIdx.x / cpb;
    const long npix = (long)B * Ho * Wo;
    const long p0 = (long)blockIdx.x * SC_PIX_PER_BLOCK;
    const long p1 = min(npix, p0 + SC_PIX_PER_BLOCK);
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (chunk < cpr && slot < slots) {
        const int c0 = chunk * 8;
        for (long pix = p0 + slot; pix < p1; pix += slots) {
            const int wo = (int)(pix % Wo);
            const long t = pix / Wo;
            const int ho = (int)(t % Ho);
            const int b = (int)(t / Ho);
            float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int kh = 0; kh < 3; ++kh) {
                const int h = 2 * ho - ph + kh;
                if (h < 0 || h >= H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int x = 2 * wo - pw + kw;
                    if (x < 0 || x >= W) continue;
                    const float* ip = in + (((long)b * H + h) * W + x) * Cin;
                    const float* wp = w + ((long)(kh * 3 + kw) * Cin) * Cout + c0;
                    for (int ci = 0; ci < Cin; ++ci) {
                        const float iv = ip[ci];
                        const float4 w0 = *(const float4*)(wp + (long)ci * Cout);
                        const float4 w1 = *(const float4*)(wp + (long)ci * Cout + 4);
                        acc[0] = fmaf(iv, w0.x, acc[0]); acc[1] = fmaf(iv, w0.y, acc[1]);
                        acc[2] = fmaf(iv, w0.z, acc[2]); acc[3] = fmaf(iv, w0.w, acc[3]);
                        acc[4] = fmaf(iv, w1.x, acc[4]); acc[5] = fmaf(iv, w1.y, acc[5]);
                        acc[6] = fmaf(iv, w1.z, acc[6]); acc[7] = fmaf(iv, w1.w, acc[7]);
                    }
                }
            }
            T* op = out + pix * Cout + c0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                T st = from_f32<T>(acc[i]);
                op[i] = st;
                float sv = to_f32<T>(st);
                s1[i] += sv;
                s2[i] += sv * sv;
            }
        }
    }
    if (!stats) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) { red[threadIdx.x][i] = s1[i]; red[threadIdx.x][8 + i] = s2[i]; }
    __syncthreads();
    if (slot == 0 && chunk < cpr) {
        const int c0 = chunk * 8;
        for (int i = 0; i < 8; ++i) {
            float a = 0.f, q = 0.f;
            for (int k = 0; k < slots; ++k) {
                a += red[k * cpb + (threadIdx.x % cpb)][i];
                q += red[k * cpb + (threadIdx.x % cpb)][8 + i];
            }
            atomicAdd(stats + c0 + i, (double)a);
            atomicAdd(stats + Cout + c0 + i, (double)q);
        }
    }
}

extern "C" int athena_conv3x3s2_small_cin_fwd(const float* in, const float* w_hwio, void* out, int B, int H, int W,
                                              int Cin, int Cout, int dtype, double* colstats,
                                              athena_stream_t stream) {
    ATHENA_REQUIRE(in && w_hwio && out, ATHENA_E_SHAPE, "athena_conv3x3s2_small_cin_fwd: null operand");
    ATHENA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 4 && Cout > 0 && Cout % 8 == 0, ATHENA_E_SHAPE,
                   "athena_conv3x3s2_small_cin_fwd: needs Cin <= 4 and Cout %% 8 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    int Ho, Wo, ph, pw;
    same_pad2(H, &Ho, &ph);
    same_pad2(W, &Wo, &pw);
    hipStream_t s = (hipStream_t)stream;
    if (colstats)
        ATHENA_REQUIRE(hipMemsetAsync(colstats, 0, sizeof(double) * 2 * Cout, s) == hipSuccess, ATHENA_E_LAUNCH, "athena_conv3x3s2_small_cin_fwd: memset failed");
    const int cpr = Cout / 8, cpb = cpr < 256 ? cpr : 256;
    const long npix = (long)B * Ho * Wo;
    dim3 grid((unsigned)((npix + SC_PIX_PER_BLOCK - 1) / SC_PIX_PER_BLOCK), (cpr + cpb - 1) / cpb);
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(conv_small_cin_fwd_kernel<bf16_t>, grid, dim3(256), 0, s, in, w_hwio, (bf16_t*)out, B, H, W, Cin, Cout, Ho, Wo, ph, pw, colstats);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(conv_small_cin_fwd_kernel<float>, grid, dim3(256), 0, s, in, w_hwio, (float*)out, B, H, W, Cin, Cout, Ho, Wo, ph, pw, colstats);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_conv3x3s2_small_cin_fwd: dtype");
    ATHENA_CHECK_LAUNCH("athena_conv3x3s2_small_cin_fwd");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- conv, small Cin, weight gradient
// dw[tap][ci][co] = sum_pix in(pix, tap)[ci] * dout[pix][co]; one pass per ci (grid.z)
#define SW_PIX_PER_BLOCK 1024
template <typename T>
__global__ __launch_bounds__(256) void conv_small_cin_wgrad_kernel(const float* __restrict__ in,
                                                                   const T* __restrict__ dout,
                                                                   float* __restrict__ dw, int B, int H, int W,
                                                                   int Cin, int Cout, int Ho, int Wo, int ph, int pw) {
    __shared__ float red[256][9];
    const int ci = blockIdx.z;
    const int cpr = Cout / 8;
    const int cpb = min(cpr, 256);
    const int slots = 256 / cpb;
    const int chunk = blockIdx.y * cpb + (threadIdx.x % cpb);
    const int slot = threadIdx.x / cpb;
    const long npix = (long)B * Ho * Wo;
    const long p0 = (long)blockIdx.x * SW_PIX_PER_BLOCK;
    const long p1 = min(npix, p0 + SW_PIX_PER_BLOCK);
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[t][i] = 0.f;
    if (chunk < cpr && slot < slots) {
        const int c0 = chunk * 8;
        for (long pix = p0 + slot; pix < p1; pix += slots) {
            const int wo = (int)(pix % Wo);
            const long t = pix / Wo;
            const int ho = (int)(t % Ho);
            const int b = (int)(t / Ho);
            float g[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) g[i] = to_f32<T>(dout[pix * Cout + c0 + i]);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int h = 2 * ho - ph + kh;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int x = 2 * wo - pw + kw;
                    float iv = 0.f;
                    if (h >= 0 && h < H && x >= 0 && x < W) iv = in[(((long)b * H + h) * W + x) * Cin + ci];
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[kh * 3 + kw][i] = fmaf(iv, g[i], acc[kh * 3 + kw][i]);
                }
            }
        }
    }
    // reduce the row slots tap by tap
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) red[threadIdx.x][i] = acc[t][i];
        __syncthreads();
        if (slot == 0 && chunk < cpr) {
            const int c0 = chunk * 8;
            for (int i = 0; i < 8; ++i) {
                float a = 0.f;
                for (int k = 0; k < slots; ++k) a += red[k * cpb + (threadIdx.x % cpb)][i];
                atomicAdd(dw + ((long)t * Cin + ci) * Cout + c0 + i, a);
            }
        }
    }
}

extern "C" int athena_conv3x3s2_small_cin_wgrad(const float* in, const void* dout, float* dw_hwio, int B, int H,
                                                int W, int Cin, int Cout, int dtype, athena_stream_t stream) {
    ATHENA_REQUIRE(in && dout && dw_hwio, ATHENA_E_SHAPE, "athena_conv3x3s2_small_cin_wgrad: null operand");
    ATHENA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cin <= 4 && Cout > 0 && Cout % 8 == 0, ATHENA_E_SHAPE,
                   "athena_conv3x3s2_small_cin_wgrad: needs Cin <= 4 and Cout %% 8 == 0");
    int Ho, Wo, ph, pw;
    same_pad2(H, &Ho, &ph);
    same_pad2(W, &Wo, &pw);
    hipStream_t s = (hipStream_t)stream;
    ATHENA_REQUIRE(hipMemsetAsync(dw_hwio, 0, sizeof(float) * 9L * Cin * Cout, s) == hipSuccess, ATHENA_E_LAUNCH, "athena_conv3x3s2_small_cin_wgrad: memset failed");
    const int cpr = Cout / 8, cpb = cpr < 256 ? cpr : 256;
    const long npix = (long)B * Ho * Wo;
    dim3 grid((unsigned)((npix + SW_PIX_PER_BLOCK - 1) / SW_PIX_PER_BLOCK), (cpr + cpb - 1) / cpb, Cin);
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(conv_small_cin_wgrad_kernel<bf16_t>, grid, dim3(256), 0, s, in, (const bf16_t*)dout, dw_hwio, B, H, W, Cin, Cout, Ho, Wo, ph, pw);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(conv_small_cin_wgrad_kernel<float>, grid, dim3(256), 0, s, in, (const float*)dout, dw_hwio, B, H, W, Cin, Cout, Ho, Wo, ph, pw);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_conv3x3s2_small_cin_wgrad: dtype");
    ATHENA_CHECK_LAUNCH("athena_conv3x3s2_small_cin_wgrad");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- positional encoding / dropout
template <typename T>
__global__ void add_pe_dropout_kernel(const T* __restrict__ x, const float* __restrict__ pe, long n, int Tn, int d,
                                      float xscale, uint32_t thresh, float dscale, const uint64_t* rng_state,
                                      uint32_t site, T* __restrict__ y) {
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const int t = (int)((i / d) % Tn);
        float v = to_f32<T>(x[i]) * xscale + pe[(long)t * d + c];
        if (thresh) v *= dropout_mult(key, (uint64_t)i, thresh, dscale);
        y[i] = from_f32<T>(v);
    }
}

extern "C" int athena_add_pe_dropout(const void* x, const float* pe, int B, int T, int d, int dtype, float xscale,
                                     float drop_p, const uint64_t* rng_state, uint32_t rng_site, void* y,
                                     athena_stream_t stream) {
    ATHENA_REQUIRE(x && pe && y && B > 0 && T > 0 && d > 0, ATHENA_E_SHAPE, "athena_add_pe_dropout: bad arguments");
    ATHENA_REQUIRE(drop_p == 0.f || (rng_state && drop_p < 1.f), ATHENA_E_SHAPE, "athena_add_pe_dropout: dropout args");
    const long n = (long)B * T * d;
    uint32_t th = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(add_pe_dropout_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)x, pe, n, T, d, xscale, th, ds, rng_state, rng_site, (bf16_t*)y);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(add_pe_dropout_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)x, pe, n, T, d, xscale, th, ds, rng_state, rng_site, (float*)y);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_add_pe_dropout: dtype");
    ATHENA_CHECK_LAUNCH("athena_add_pe_dropout");
    return ATHENA_OK;
}

template <typename T>
__global__ void dropout_bwd_kernel(const T* __restrict__ dy, long n, float xscale, uint32_t thresh, float dscale,
                                   const uint64_t* rng_state, uint32_t site, const T* __restrict__ relu6_out,
                                   T* __restrict__ dx) {
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = to_f32<T>(dy[i]) * xscale;
        if (thresh) v *= dropout_mult(key, (uint64_t)i, thresh, dscale);
        if (relu6_out) {
            const float y = to_f32<T>(relu6_out[i]);
            if (!(y > 0.f && y < 6.f)) v = 0.f;
        }
        dx[i] = from_f32<T>(v);
    }
}

extern "C" int athena_dropout_bwd(const void* dy, long n, int dtype, float xscale, float drop_p,
                                  const uint64_t* rng_state, uint32_t rng_site, const void* relu6_out, void* dx,
                                  athena_stream_t stream) {
    ATHENA_REQUIRE(dy && dx && n > 0, ATHENA_E_SHAPE, "athena_dropout_bwd: bad arguments");
    ATHENA_REQUIRE(drop_p == 0.f || (rng_state && drop_p < 1.f), ATHENA_E_SHAPE, "athena_dropout_bwd: dropout args");
    uint32_t th = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(dropout_bwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16_t*)dy, n, xscale, th, ds, rng_state, rng_site, (const bf16_t*)relu6_out, (bf16_t*)dx);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(dropout_bwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)dy, n, xscale, th, ds, rng_state, rng_site, (const float*)relu6_out, (float*)dx);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_dropout_bwd: dtype");
    ATHENA_CHECK_LAUNCH("athena_dropout_bwd");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- embedding
template <typename T>
__global__ void embed_fwd_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                 const float* __restrict__ pe, long n, int L, int d, int V, float xscale,
                                 uint32_t thresh, float dscale, const uint64_t* rng_state, uint32_t site,
                                 T* __restrict__ y) {
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % d);
        const long tok = i / d;
        const int l = (int)(tok % L);
        int id = ids[tok];
        id = id < 0 ? 0 : (id >= V ? V - 1 : id);
        float v = table[(long)id * d + c] * xscale + pe[(long)l * d + c];
        if (thresh) v *= dropout_mult(key, (uint64_t)i, thresh, dscale);
        y[i] = from_f32<T>(v);
    }
}

// One WAVE per row v of the table (four rows per workgroup): the wave walks the tokens 256 at a time --
// four coalesced loads of ids in flight, a compare and a ballot each -- and where a chunk holds
// matches (a wave-uniform branch) adds those tokens' rows of dy into the columns its lanes own, 64
// apart.  Matches are taken in token order, so the sum has a fixed order, but their rows are fetched
// four at a time with every load in bounds and unconditional (a token or column that does not exist
// reads a neighbour and counts with factor zero): a row that many tokens point at -- the padding id,
// sos -- costs an issue slot per match and element, not a memory round trip.  Every row of dE is
// written (zero where no token has the id); no LDS, no barriers, no atomics.
// EMB_COLS: columns per lane and pass over the tokens (4: tables up to 256 wide in one pass; 8: 512 per pass)
template <typename T, int EMB_COLS>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dy, const int32_t* __restrict__ ids,
                                                        int ntok, int d, int V, float xscale, uint32_t thresh,
                                                        float dscale, const uint64_t* rng_state, uint32_t site,
                                                        float* __restrict__ dtable) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= V) return;
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    for (int c0 = 0; c0 < d; c0 += 64 * EMB_COLS) {
        float acc[EMB_COLS];
        int col[EMB_COLS];
        float cok[EMB_COLS];
#pragma unroll
        for (int k = 0; k < EMB_COLS; ++k) {
            acc[k] = 0.f;
            const int c = c0 + 64 * k + lane;
            col[k] = c < d ? c : d - 1;
            cok[k] = c < d ? xscale : 0.f;
        }
        for (int t0 = 0; t0 < ntok; t0 += 256) {
            int idv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = t0 + 64 * q + lane;
                idv[q] = ids[t < ntok ? t : ntok - 1];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int t = t0 + 64 * q + lane;
                int id = idv[q];
                id = id < 0 ? 0 : (id >= V ? V - 1 : id);
                uint64_t b = __ballot(t < ntok && id == v);
                while (b) {
                    // up to four matches, rows fetched together, added in token order
                    int tok[4];
                    float w[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool any = b != 0;
                        tok[j] = any ? t0 + 64 * q + __builtin_ctzll(b) : (j ? tok[j - 1] : 0);
                        w[j] = any ? 1.f : 0.f;
                        b &= b - 1;
                    }
                    float g[4][EMB_COLS];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < EMB_COLS; ++k) g[j][k] = to_f32<T>(dy[(long)tok[j] * d + col[k]]);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int k = 0; k < EMB_COLS; ++k) {
                            float x = g[j][k] * (cok[k] * w[j]);
                            if (thresh) x *= dropout_mult(key, (uint64_t)((long)tok[j] * d + col[k]), thresh, dscale);
                            acc[k] += x;
                        }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < EMB_COLS; ++k) {
            const int c = c0 + 64 * k + lane;
            if (c < d) dtable[(long)v * d + c] = acc[k];
        }
    }
}

// The same walk for tables whose width is a multiple of four (with dy aligned to four elements): a lane
// owns FOUR CONSECUTIVE columns, so a token's row of 256 columns is one load per lane, one hash decides
// its four dropout multipliers, and eight matches are in flight at a time.  A pass over the tokens
// covers 256 columns.
template <typename T>
__device__ __forceinline__ void emb_load4(const T* p, float (&x)[4]);
template <>
__device__ __forceinline__ void emb_load4<float>(const float* p, float (&x)[4]) {
    const float4 v = *(const float4*)p;
    x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
}
template <>
__device__ __forceinline__ void emb_load4<bf16_t>(const bf16_t* p, float (&x)[4]) {
    const uint2 v = *(const uint2*)p;
    x[0] = __uint_as_float(v.x << 16); x[1] = __uint_as_float(v.x & 0xffff0000u);
    x[2] = __uint_as_float(v.y << 16); x[3] = __uint_as_float(v.y & 0xffff0000u);
}
#define EMB_BATCH 8
// adds the rows of dy of the tokens in [ta, tb) whose id is v into acc (columns cc .. cc + 3, factor cf)
// -- unless there are more than ``limit`` of them: then false comes back as soon as that shows
template <typename T>
__device__ __forceinline__ bool emb_walk(const T* __restrict__ dy, const int32_t* __restrict__ ids, int ta, int tb, int d,
                                         int V, int v, int cc, float cf, uint64_t key, uint32_t thresh, float dscale,
                                         int lane, int limit, float (&acc)[4]) {
    int count = 0;
    // the ids of the next 256 tokens are requested before the current ones are looked at
    int nxt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int t = ta + 64 * q + lane;
        nxt[q] = ids[t < tb ? t : tb - 1];
    }
    for (int t0 = ta; t0 < tb; t0 += 256) {
        int idv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            idv[q] = nxt[q];
            const int t = t0 + 256 + 64 * q + lane;
            nxt[q] = ids[t < tb ? t : tb - 1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = t0 + 64 * q + lane;
            int id = idv[q];
            id = id < 0 ? 0 : (id >= V ? V - 1 : id);
            uint64_t b = __ballot(t < tb && id == v);
            count += __popcll(b);
            if (count > limit) return false;
            while (b) {
                int tok[EMB_BATCH];
                float w[EMB_BATCH];
#pragma unroll
                for (int j = 0; j < EMB_BATCH; ++j) {
                    const bool any = b != 0;
                    tok[j] = any ? t0 + 64 * q + __builtin_ctzll(b) : (j ? tok[j - 1] : ta);
                    w[j] = any ? cf : 0.f;
                    b &= b - 1;
                }
                float g[EMB_BATCH][4];
#pragma unroll
                for (int j = 0; j < EMB_BATCH; ++j) emb_load4<T>(dy + (long)tok[j] * d + cc, g[j]);
#pragma unroll
                for (int j = 0; j < EMB_BATCH; ++j) {
                    float m[4] = {1.f, 1.f, 1.f, 1.f};
                    if (thresh) dropout_mult4(key, (uint64_t)((long)tok[j] * d + cc), thresh, dscale, m);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] += g[j][r] * (w[j] * m[r]);
                }
            }
        }
    }
    return true;
}

// NW rows of the table per workgroup, one per wave.  A wave sums the rows of dy of the tokens that
// point at its row by itself, in token order -- up to EMB_HOT of them.  A row more tokens point at (the
// padding id, sos) would keep one wave busy for a microsecond per eight tokens while the rest of the
// chip has finished, so the wave gives up when the count passes EMB_HOT and the NW waves take such
// a row together afterwards: each sums an NW-th of the tokens (contiguous, in order) and the
// partial sums are added in wave order.  Either way the order of the additions depends on
// the ids alone.
#define EMB_HOT 48
template <typename T, int NW>
__global__ __launch_bounds__(64 * NW) void embed_bwd_vec_kernel(const T* __restrict__ dy, const int32_t* __restrict__ ids,
                                                             int ntok, int d, int V, float xscale, uint32_t thresh,
                                                             float dscale, const uint64_t* rng_state, uint32_t site,
                                                             float* __restrict__ dtable) {
    __shared__ float part[NW][256];
    __shared__ int hot[NW];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int v = blockIdx.x * NW + wave;
    uint64_t key = 0;
    if (thresh) key = rng_key(rng_state, site);
    int slice = (ntok + NW - 1) / NW;
    slice = (slice + 63) & ~63;
    for (int c0 = 0; c0 < d; c0 += 256) {
        const int c = c0 + 4 * lane;
        const int cc = c < d ? c : d - 4;               // a lane beyond the table reads its last columns, factor 0
        const float cf = c < d ? xscale : 0.f;
        bool done = true;
        if (v < V) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            done = emb_walk<T>(dy, ids, 0, ntok, d, V, v, cc, cf, key, thresh, dscale, lane, EMB_HOT, acc);
            if (done && c < d) *(float4*)(dtable + (long)v * d + c) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        if (lane == 0) hot[wave] = done ? 0 : 1;
        __syncthreads();
        for (int r = 0; r < NW; ++r) {
            if (!hot[r]) continue;                       // uniform over the workgroup
            const int vr = blockIdx.x * NW + r;
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const int ta = wave * slice, tb = min(ntok, ta + slice);
            if (ta < tb) emb_walk<T>(dy, ids, ta, tb, d, V, vr, cc, cf, key, thresh, dscale, lane, 0x7fffffff, acc);
            *(float4*)&part[wave][4 * lane] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            __syncthreads();
            if (wave == 0 && c < d) {
                float4 t = *(const float4*)&part[0][4 * lane];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const float4 u = *(const float4*)&part[w][4 * lane];
                    t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
                }
                *(float4*)(dtable + (long)vr * d + c) = t;
            }
            __syncthreads();
        }
        __syncthreads();                                 // hot[] is rewritten by the next pass
    }
}

extern "C" int athena_embed_fwd(const float* table, const int32_t* ids, const float* pe, int B, int L, int d, int V,
                                int dtype, float xscale, float drop_p, const uint64_t* rng_state, uint32_t rng_site,
                                void* y, athena_stream_t stream) {
    ATHENA_REQUIRE(table && ids && pe && y && B > 0 && L > 0 && d > 0 && V > 0, ATHENA_E_SHAPE, "athena_embed_fwd: bad arguments");
    ATHENA_REQUIRE(drop_p == 0.f || (rng_state && drop_p < 1.f), ATHENA_E_SHAPE, "athena_embed_fwd: dropout args");
    const long n = (long)B * L * d;
    uint32_t th = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(embed_fwd_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, table, ids, pe, n, L, d, V, xscale, th, ds, rng_state, rng_site, (bf16_t*)y);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(embed_fwd_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, table, ids, pe, n, L, d, V, xscale, th, ds, rng_state, rng_site, (float*)y);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_embed_fwd: dtype");
    ATHENA_CHECK_LAUNCH("athena_embed_fwd");
    return ATHENA_OK;
}

#define EMB_WAVES 4      // measured: 10 / 15 / 14 us at 4 / 8 / 16 waves on the benchmark shape, 32 / 28 / 26 us with half of the tokens on one id
#define EMB_VEC(T)                                                                                                     \
    do {                                                                                                               \
        if (nw == 4) ATHENA_LAUNCH((embed_bwd_vec_kernel<T, 4>), gridw, dim3(256), 0, s, (const T*)dy, ids, ntok, d, V, xscale, th, ds, rng_state, rng_site, dtable);        \
        else if (nw == 8) ATHENA_LAUNCH((embed_bwd_vec_kernel<T, 8>), gridw, dim3(512), 0, s, (const T*)dy, ids, ntok, d, V, xscale, th, ds, rng_state, rng_site, dtable);   \
        else ATHENA_LAUNCH((embed_bwd_vec_kernel<T, 16>), gridw, dim3(1024), 0, s, (const T*)dy, ids, ntok, d, V, xscale, th, ds, rng_state, rng_site, dtable);              \
    } while (0)
extern "C" int athena_embed_bwd(const void* dy, const int32_t* ids, int B, int L, int d, int V, int dtype,
                                float xscale, float drop_p, const uint64_t* rng_state, uint32_t rng_site,
                                float* dtable, athena_stream_t stream) {
    ATHENA_REQUIRE(dy && ids && dtable && B > 0 && L > 0 && d > 0 && V > 0, ATHENA_E_SHAPE, "athena_embed_bwd: bad arguments");
    ATHENA_REQUIRE(drop_p == 0.f || (rng_state && drop_p < 1.f), ATHENA_E_SHAPE, "athena_embed_bwd: dropout args");
    ATHENA_REQUIRE((long)B * L < (1L << 31) - 64, ATHENA_E_SHAPE, "athena_embed_bwd: too many tokens");
    const int ntok = B * L;
    uint32_t th = drop_p > 0.f ? dropout_threshold(drop_p) : 0u;
    float ds = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    hipStream_t s = (hipStream_t)stream;
    static const int nw_env = getenv("ATHENA_EMBED_WAVES") ? atoi(getenv("ATHENA_EMBED_WAVES")) : 0;
    const int nw = (nw_env == 4 || nw_env == 8 || nw_env == 16) ? nw_env : EMB_WAVES;
    const dim3 grid((unsigned)((V + 3) / 4)), gridw((unsigned)((V + nw - 1) / nw));
    const bool vec = d % 4 == 0 && ((uintptr_t)dy & 15) == 0 && ((uintptr_t)dtable & 15) == 0;
    if (dtype == ATHENA_BF16) {
        if (vec) EMB_VEC(bf16_t);
        else ATHENA_LAUNCH((embed_bwd_kernel<bf16_t, 8>), grid, dim3(256), 0, s, (const bf16_t*)dy, ids, ntok, d, V, xscale, th, ds, rng_state, rng_site, dtable);
    } else if (dtype == ATHENA_F32) {
        if (vec) EMB_VEC(float);
        else ATHENA_LAUNCH((embed_bwd_kernel<float, 8>), grid, dim3(256), 0, s, (const float*)dy, ids, ntok, d, V, xscale, th, ds, rng_state, rng_site, dtable);
    } else {
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_embed_bwd: dtype");
    }
    ATHENA_CHECK_LAUNCH("athena_embed_bwd");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- several copies, one launch
// A training step starts by copying a handful of small tensors (lengths, labels, the feature batch)
// into the buffers its recorded graph reads, and ends by copying a handful of small tensors (the
// arg-max ids of the metrics) to pinned host memory; as separate copies each is a launch of 4 us
// or more with a host round in between.  blockIdx.y picks the copy, the x blocks stride over it.
struct CopyMany {
    const void* src[ATHENA_COPY_MANY_MAX];
    void* dst[ATHENA_COPY_MANY_MAX];
    long bytes[ATHENA_COPY_MANY_MAX];
};
__global__ __launch_bounds__(256) void copy_many_kernel(CopyMany c) {
    const int k = blockIdx.y;
    const char* src = (const char*)c.src[k];
    char* dst = (char*)c.dst[k];
    const long n = c.bytes[k];
    const long tid = (long)blockIdx.x * 256 + threadIdx.x, nthr = (long)gridDim.x * 256;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        const long n16 = n >> 4;
        for (long i = tid; i < n16; i += nthr) ((uint4*)dst)[i] = ((const uint4*)src)[i];
        for (long i = (n16 << 4) + tid; i < n; i += nthr) dst[i] = src[i];
    } else {
        for (long i = tid; i < n; i += nthr) dst[i] = src[i];
    }
}

extern "C" int athena_copy_many(const void* const* src, void* const* dst, const long* bytes, int n,
                                athena_stream_t stream) {
    ATHENA_REQUIRE(src && dst && bytes && n > 0 && n <= ATHENA_COPY_MANY_MAX, ATHENA_E_SHAPE,
                   "athena_copy_many: 1 to %d copies", ATHENA_COPY_MANY_MAX);
    CopyMany c;
    long most = 0;
    for (int i = 0; i < ATHENA_COPY_MANY_MAX; ++i) {
        const int k = i < n ? i : 0;
        ATHENA_REQUIRE(src[k] && dst[k] && bytes[k] >= 0, ATHENA_E_SHAPE, "athena_copy_many: bad copy %d", k);
        c.src[i] = src[k];
        c.dst[i] = dst[k];
        c.bytes[i] = i < n ? bytes[k] : 0;
        if (c.bytes[i] > most) most = c.bytes[i];
    }
    long blocks = (most / 16 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    ATHENA_LAUNCH(copy_many_kernel, dim3((unsigned)blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, c);
    ATHENA_CHECK_LAUNCH("athena_copy_many");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- casts / axpby
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = from_f32<TD>(to_f32<TS>(src[i]));
}

extern "C" int athena_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, athena_stream_t stream) {
    ATHENA_REQUIRE(src && dst && n > 0, ATHENA_E_SHAPE, "athena_cast: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    dim3 g((unsigned)blocks), b(256);
    if (src_dtype == ATHENA_F32 && dst_dtype == ATHENA_BF16) ATHENA_LAUNCH((cast_kernel<float, bf16_t>), g, b, 0, s, (const float*)src, (bf16_t*)dst, n);
    else if (src_dtype == ATHENA_BF16 && dst_dtype == ATHENA_F32) ATHENA_LAUNCH((cast_kernel<bf16_t, float>), g, b, 0, s, (const bf16_t*)src, (float*)dst, n);
    else if (src_dtype == ATHENA_F32 && dst_dtype == ATHENA_F32) ATHENA_LAUNCH((cast_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, n);
    else if (src_dtype == ATHENA_BF16 && dst_dtype == ATHENA_BF16) ATHENA_LAUNCH((cast_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)src, (bf16_t*)dst, n);
    else ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_cast: dtype");
    ATHENA_CHECK_LAUNCH("athena_cast");
    return ATHENA_OK;
}

template <typename T>
__global__ void axpby_kernel(float a, const T* __restrict__ x, float b, T* __restrict__ y, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float v = a * to_f32<T>(x[i]);
        if (b != 0.f) v += b * to_f32<T>(y[i]);
        y[i] = from_f32<T>(v);
    }
}

extern "C" int athena_axpby(float a, const void* x, float b, void* y, long n, int dtype, athena_stream_t stream) {
    ATHENA_REQUIRE(x && y && n > 0, ATHENA_E_SHAPE, "athena_axpby: bad arguments");
    long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ATHENA_BF16) ATHENA_LAUNCH(axpby_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, a, (const bf16_t*)x, b, (bf16_t*)y, n);
    else if (dtype == ATHENA_F32) ATHENA_LAUNCH(axpby_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, a, (const float*)x, b, (float*)y, n);
    else ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_axpby: dtype");
    ATHENA_CHECK_LAUNCH("athena_axpby");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- labels and key masks of a step
// one workgroup per utterance
__global__ __launch_bounds__(256) void prepare_labels_kernel(const int32_t* __restrict__ labels,
                                                             const int32_t* __restrict__ label_len,
                                                             const int32_t* __restrict__ input_len, int U, int Tp,
                                                             int sos, int eos, int32_t* __restrict__ y0,
                                                             int32_t* __restrict__ tgt, uint8_t* __restrict__ y_mask,
                                                             int32_t* __restrict__ logit_len,
                                                             uint8_t* __restrict__ x_mask) {
    const int b = blockIdx.x;
    const int L = U + 1;
    const int len = label_len[b];
    const int n = input_len[b];
    const int tl = (((n + 1) >> 1) + 1) >> 1;        // ceil(ceil(n / 2) / 2)
    if (threadIdx.x == 0) logit_len[b] = tl;
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int yv = i == 0 ? sos : labels[(long)b * U + i - 1];
        y0[(long)b * L + i] = yv;
        y_mask[(long)b * L + i] = yv == 0 ? 1 : 0;
        int tv = i < U ? labels[(long)b * U + i] : 0;
        if (i == len) tv += eos;
        tgt[(long)b * L + i] = tv;
    }
    for (int t = threadIdx.x; t < Tp; t += blockDim.x) x_mask[(long)b * Tp + t] = t >= tl ? 1 : 0;
}

extern "C" int athena_prepare_labels(const int32_t* labels, const int32_t* label_len, const int32_t* input_len, int B,
                                     int U, int Tp, int sos, int eos, int32_t* y0, int32_t* tgt, uint8_t* y_mask,
                                     int32_t* logit_len, uint8_t* x_mask, athena_stream_t stream) {
    ATHENA_REQUIRE(labels && label_len && input_len && y0 && tgt && y_mask && logit_len && x_mask, ATHENA_E_SHAPE,
                   "athena_prepare_labels: null operand");
    ATHENA_REQUIRE(B > 0 && U > 0 && Tp > 0, ATHENA_E_SHAPE, "athena_prepare_labels: bad shape B=%d U=%d T'=%d", B, U, Tp);
    ATHENA_LAUNCH(prepare_labels_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, labels, label_len, input_len, U,
                       Tp, sos, eos, y0, tgt, y_mask, logit_len, x_mask);
    ATHENA_CHECK_LAUNCH("athena_prepare_labels");
    return ATHENA_OK;
}
