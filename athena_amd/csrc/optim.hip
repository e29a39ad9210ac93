// Per-tensor clip_by_norm + Adam with the warm-up schedule on flat fp32 buffers, and the
// refresh of the bf16 weight shadows.  Pure streaming kernels (28 B/param for the update).
#include "common.h"

#define ADAM_CHUNK 4096   // elements per workgroup

// binary search: largest i with seg_offsets[i] <= idx
__device__ __forceinline__ int find_segment(const int64_t* __restrict__ off, int nseg, long idx) {
    int lo = 0, hi = nseg - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (off[mid] <= idx) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// sum of squares per tensor: each workgroup reduces one chunk; a chunk may straddle tensors
#define SUMSQ_CHUNK 16384  // elements per workgroup: 16 four-element loads in flight per thread
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* __restrict__ g, long n,
                                                         const int64_t* __restrict__ seg_offsets, int nseg,
                                                         float gscale, float* __restrict__ norms) {
    __shared__ float red[32];
    const long c0 = (long)blockIdx.x * SUMSQ_CHUNK;
    const long c1 = min(n, c0 + SUMSQ_CHUNK);
    int seg = find_segment(seg_offsets, nseg, c0);
    long pos = c0;
    while (pos < c1) {
        const long send = min(c1, (long)seg_offsets[seg + 1]);
        float s = 0.f;
        if (((pos | send) & 3) == 0 && ((((uintptr_t)g) & 15) == 0)) {
            float s1 = 0.f, s2 = 0.f, s3 = 0.f;
            for (long i = pos + 4L * threadIdx.x; i < send; i += 4L * blockDim.x) {
                const float4 v = *(const float4*)(g + i);
                s = fmaf(v.x * gscale, v.x * gscale, s);
                s1 = fmaf(v.y * gscale, v.y * gscale, s1);
                s2 = fmaf(v.z * gscale, v.z * gscale, s2);
                s3 = fmaf(v.w * gscale, v.w * gscale, s3);
            }
            s = (s + s1) + (s2 + s3);
        } else {
            for (long i = pos + threadIdx.x; i < send; i += blockDim.x) {
                float v = g[i] * gscale;
                s += v * v;
            }
        }
        s = block_sum(s, red);
        if (threadIdx.x == 0 && send > pos) atomicAdd(norms + seg, s);
        pos = send;
        ++seg;
    }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const int64_t* __restrict__ seg_offsets, int nseg,
                                                        const float* __restrict__ norms,
                                                        const int64_t* __restrict__ step_ptr, athena_adam_cfg cfg,
                                                        float* __restrict__ lr_out) {
    const long step = step_ptr[0];
    // learning rate of the warm-up schedule evaluated at ``step`` (0 on the very first update)
    const float fs = (float)step;
    const float arg1 = step > 0 ? rsqrtf(fs) : INFINITY;
    const float arg2 = fs * powf(cfg.warmup_steps, -1.5f);
    const float kk = cfg.k * powf(cfg.decay_rate, floorf(fs / cfg.decay_steps));
    float lr = kk * rsqrtf(cfg.d_model) * fminf(arg1, arg2);
    if (cfg.schedule == 1.f) {
        // exponential decay in whole multiples of decay_steps, from start_decay_steps on
        const float factor = fs < cfg.start_decay_steps
                                 ? 1.f
                                 : powf(cfg.decay_rate, floorf((fs - cfg.start_decay_steps) / cfg.decay_steps));
        lr = fminf(fmaxf(cfg.initial_lr * factor, cfg.final_lr), cfg.initial_lr);
    }
    const float t = fs + 1.f;
    const float lr_t = lr * sqrtf(1.f - powf(cfg.beta2, t)) / (1.f - powf(cfg.beta1, t));
    if (lr_out && blockIdx.x == 0 && threadIdx.x == 0) lr_out[0] = lr;

    const long c0 = (long)blockIdx.x * ADAM_CHUNK;
    const long c1 = min(n, c0 + ADAM_CHUNK);
    int seg = find_segment(seg_offsets, nseg, c0);
    long pos = c0;
    while (pos < c1) {
        const long send = min(c1, (long)seg_offsets[seg + 1]);
        float scale = cfg.grad_scale;
        if (cfg.clip_norm > 0.f) {
            const float nrm = sqrtf(norms[seg]);
            scale *= cfg.clip_norm / fmaxf(nrm, cfg.clip_norm);
        }
        for (long i = pos + threadIdx.x; i < send; i += blockDim.x) {
            const float gi = g[i] * scale;
            const float mi = cfg.beta1 * m[i] + (1.f - cfg.beta1) * gi;
            const float vi = cfg.beta2 * v[i] + (1.f - cfg.beta2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            p[i] -= lr_t * mi / (sqrtf(vi) + cfg.eps);
        }
        pos = send;
        ++seg;
    }
}

__global__ void step_increment_kernel(int64_t* step) { step[0] += 1; }

extern "C" int athena_clip_adam_step(float* params, const float* grads, float* m, float* v, long n,
                                     const int64_t* seg_offsets, int nseg, float* norms, int64_t* step,
                                     const athena_adam_cfg* cfg, float* lr_out, athena_stream_t stream) {
    ATHENA_REQUIRE(params && grads && m && v && seg_offsets && norms && step && cfg, ATHENA_E_SHAPE, "athena_clip_adam_step: null operand");
    ATHENA_REQUIRE(n > 0 && nseg > 0, ATHENA_E_SHAPE, "athena_clip_adam_step: bad sizes");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((n + ADAM_CHUNK - 1) / ADAM_CHUNK);
    if (cfg->clip_norm > 0.f) {
        ATHENA_REQUIRE(hipMemsetAsync(norms, 0, sizeof(float) * nseg, s) == hipSuccess, ATHENA_E_LAUNCH, "athena_clip_adam_step: memset failed");
        const unsigned sblocks = (unsigned)((n + SUMSQ_CHUNK - 1) / SUMSQ_CHUNK);
        ATHENA_LAUNCH(grad_sumsq_kernel, dim3(sblocks), dim3(256), 0, s, grads, n, seg_offsets, nseg, cfg->grad_scale, norms);
    }
    ATHENA_LAUNCH(clip_adam_kernel, dim3(blocks), dim3(256), 0, s, params, grads, m, v, n, seg_offsets, nseg, norms, step, *cfg, lr_out);
    ATHENA_LAUNCH(step_increment_kernel, dim3(1), dim3(1), 0, s, step);
    ATHENA_CHECK_LAUNCH("athena_clip_adam_step");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- bf16 shadows
// one workgroup per 32x32 tile of one matrix; writes the straight copy and the transposed copy, and
// for matrices that ask for them (offsets >= 0; dimensions multiples of 32) the two fragment-tiled
// copies: blocks of 512 elements, block (nb, kb) = 16 rows x 32 contraction indices, element
// 8 * lane + e of a block = row 16 nb + (lane & 15), contraction index 32 kb + 8 (lane >> 4) + e,
// blocks ordered nb-major.  "tiled": rows = the matrix' rows, contraction over its columns;
// "tiled_t": rows = its columns, contraction over its rows.  A wave reads a block with one
// 16-bytes-per-lane load of 1 KB of consecutive memory and has an MFMA B fragment in registers.
__global__ __launch_bounds__(256) void refresh_shadows_kernel(const float* __restrict__ params,
                                                              bf16_t* __restrict__ shadow,
                                                              const int64_t* __restrict__ table, int nmat,
                                                              const int64_t* __restrict__ tile_prefix) {
    __shared__ float tile[32][33];
    const long tid = blockIdx.x;
    int mi = find_segment(tile_prefix, nmat, tid);
    const int64_t* d = table + 9L * mi;
    const long src = d[0];
    const int rows = (int)d[1], cols = (int)d[2];
    const long dst = d[3], ld_s = d[4], dst_t = d[5], ld_t = d[6];
    const long til = d[7], til_t = d[8];
    const int tcols = (cols + 31) / 32;
    const int local = (int)(tid - tile_prefix[mi]);
    const int tr = local / tcols, tc = local - tr * tcols;
    // matrices whose dimensions, leading dimensions and offsets are all multiples of four move in
    // four-element pieces: one 16-byte load, an 8-byte store of the straight copy and an 8-byte
    // store of the transposed copy per thread
    const bool vec = ((rows | cols | (int)(ld_s & 3) | (int)(ld_t & 3)) & 3) == 0 && ((src | dst | dst_t) & 3) == 0 &&
                     ((((uintptr_t)params) & 15) | (((uintptr_t)shadow) & 7)) == 0;
    if (vec) {
        const int y = threadIdx.x >> 3, x4 = (threadIdx.x & 7) * 4;
        {
            const int r = tr * 32 + y, c = tc * 32 + x4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < rows && c < cols) {
                v = *(const float4*)(params + src + (long)r * cols + c);
                union { bf16_t h[4]; uint2 u; } pk;
                pk.h[0] = (bf16_t)v.x; pk.h[1] = (bf16_t)v.y; pk.h[2] = (bf16_t)v.z; pk.h[3] = (bf16_t)v.w;
                *(uint2*)(shadow + dst + (long)r * ld_s + c) = pk.u;
            }
            tile[y][x4] = v.x; tile[y][x4 + 1] = v.y; tile[y][x4 + 2] = v.z; tile[y][x4 + 3] = v.w;
        }
        __syncthreads();
        // transposed copy: row index c of the original columns, four consecutive original rows
        const int c = tc * 32 + y, r = tr * 32 + x4;
        if (r < rows && c < cols) {
            union { bf16_t h[4]; uint2 u; } pk;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk.h[k] = (bf16_t)tile[x4 + k][y];
            *(uint2*)(shadow + dst_t + (long)c * ld_t + r) = pk.u;
        }
        if (til >= 0 && til_t >= 0) {
            // two blocks of each tiled copy lie inside this 32x32 tile; a thread writes 4 elements of each
            const int b = threadIdx.x >> 7, lane = (threadIdx.x >> 1) & 63, e0 = (threadIdx.x & 1) * 4;
            const int fr = lane & 15, k0 = (lane >> 4) * 8 + e0;
            union { bf16_t h[4]; uint2 u; } pk;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk.h[k] = (bf16_t)tile[b * 16 + fr][k0 + k];
            *(uint2*)(shadow + til + ((long)(2 * tr + b) * (cols / 32) + tc) * 512 + lane * 8 + e0) = pk.u;
#pragma unroll
            for (int k = 0; k < 4; ++k) pk.h[k] = (bf16_t)tile[k0 + k][b * 16 + fr];
            *(uint2*)(shadow + til_t + ((long)(2 * tc + b) * (rows / 32) + tr) * 512 + lane * 8 + e0) = pk.u;
        }
        return;
    }
    const int x = threadIdx.x & 31, y0 = threadIdx.x >> 5;    // 32 x 8
    for (int y = y0; y < 32; y += 8) {
        const int r = tr * 32 + y, c = tc * 32 + x;
        float v = 0.f;
        if (r < rows && c < cols) {
            v = params[src + (long)r * cols + c];
            shadow[dst + (long)r * ld_s + c] = (bf16_t)v;
        }
        tile[y][x] = v;
    }
    __syncthreads();
    bf16_t* tdst = shadow + dst_t;
    for (int y = y0; y < 32; y += 8) {
        const int c = tc * 32 + y, r = tr * 32 + x;        // transposed: row index c, column index r
        if (r < rows && c < cols) tdst[(long)c * ld_t + r] = (bf16_t)tile[x][y];
    }
}

extern "C" int athena_refresh_shadows(const float* params, void* shadow, const int64_t* table, int nmat,
                                      const int64_t* tile_prefix, long total_tiles, athena_stream_t stream) {
    ATHENA_REQUIRE(params && shadow && table && tile_prefix && nmat > 0 && total_tiles > 0, ATHENA_E_SHAPE, "athena_refresh_shadows: bad arguments");
    ATHENA_REQUIRE(total_tiles < (1L << 31), ATHENA_E_SHAPE, "athena_refresh_shadows: too many tiles");
    ATHENA_LAUNCH(refresh_shadows_kernel, dim3((unsigned)total_tiles), dim3(256), 0, (hipStream_t)stream, params, (bf16_t*)shadow, table, nmat, tile_prefix);
    ATHENA_CHECK_LAUNCH("athena_refresh_shadows");
    return ATHENA_OK;
}

__global__ void advance_rng_kernel(uint64_t* st) { st[1] += 1; }
extern "C" int athena_advance_rng(uint64_t* rng_state, athena_stream_t stream) {
    ATHENA_REQUIRE(rng_state, ATHENA_E_SHAPE, "athena_advance_rng: null");
    ATHENA_LAUNCH(advance_rng_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state);
    ATHENA_CHECK_LAUNCH("athena_advance_rng");
    return ATHENA_OK;
}
