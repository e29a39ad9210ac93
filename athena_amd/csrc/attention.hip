// Fused multi-head attention, forward and backward: entry points, and the vector-ALU kernels
// used for fp32 operands and for head sizes the matrix-core kernels (attention_mfma.h) do not take.
//
// One thread group of P lanes owns one query row (forward, dQ pass) or one key row (dK/dV pass)
// and keeps its DK/P-wide slices of the row vectors in registers; the rows of the other side
// are streamed through LDS in tiles and read as broadcasts.  Scores never leave the chip.
// The additive mask is the reference's fp32 arithmetic, s + (-1e9), so fully masked rows
// come out as the same uniform average.
#include "common.h"
#include "attention_mfma.h"
#include <initializer_list>
#include <cstdlib>

#define ATT_TILE 32        // rows of the streamed side per LDS tile
#define ATT_BLOCK 128      // threads per block

template <typename T>
__device__ __forceinline__ void stage_rows(const T* __restrict__ src, int ld, int row0, int nrows_total, int dk,
                                           float* __restrict__ dst /*[ATT_TILE][dk]*/) {
    // cooperative copy of ATT_TILE rows x dk columns into LDS as fp32, zero past the end
    for (int e = threadIdx.x; e < ATT_TILE * dk; e += ATT_BLOCK) {
        int r = e / dk, c = e - r * dk;
        int row = row0 + r;
        dst[e] = row < nrows_total ? to_f32<T>(src[(long)row * ld + c]) : 0.f;
    }
}

// ---------------------------------------------------------------- forward
template <typename T, int DK, int P>
__global__ __launch_bounds__(ATT_BLOCK) void mha_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                            const T* __restrict__ v, int ldq, int ldk, int ldv,
                                                            const uint8_t* __restrict__ key_mask, int causal, int h,
                                                            int Tq, int Tk, float scale, T* __restrict__ out, int ldo,
                                                            float* __restrict__ lse) {
    constexpr int D = DK / P;
    constexpr int ROWS = ATT_BLOCK / P;
    __shared__ float sK[ATT_TILE * DK];
    __shared__ float sV[ATT_TILE * DK];
    __shared__ float sM[ATT_TILE];
    const int b = blockIdx.z, hh = blockIdx.y;
    const int sub = threadIdx.x % P;
    const int i = blockIdx.x * ROWS + threadIdx.x / P;
    const bool active = i < Tq;
    const T* qb = q + ((long)b * Tq) * ldq + hh * DK;
    const T* kb = k + ((long)b * Tk) * ldk + hh * DK;
    const T* vb = v + ((long)b * Tk) * ldv + hh * DK;
    float qr[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        qr[c] = active ? to_f32<T>(qb[(long)i * ldq + sub * D + c]) : 0.f;
        o[c] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int j0 = 0; j0 < Tk; j0 += ATT_TILE) {
        __syncthreads();
        stage_rows<T>(kb, ldk, j0, Tk, DK, sK);
        stage_rows<T>(vb, ldv, j0, Tk, DK, sV);
        if (threadIdx.x < ATT_TILE) {
            int j = j0 + threadIdx.x;
            sM[threadIdx.x] = (j < Tk && key_mask && key_mask[(long)b * Tk + j]) ? 1.f : 0.f;
        }
        __syncthreads();
        const int nj = min(ATT_TILE, Tk - j0);
        float s[ATT_TILE];
        float tmax = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < ATT_TILE; ++jj) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) acc = fmaf(qr[c], sK[jj * DK + sub * D + c], acc);
#pragma unroll
            for (int w = 1; w < P; w <<= 1) acc += __shfl_xor(acc, w, 64);
            float sv = acc * scale;
            const bool masked = sM[jj] != 0.f || (causal && (j0 + jj) > i + causal - 1);
            if (masked) sv += -1e9f;
            if (jj >= nj) sv = -INFINITY;
            s[jj] = sv;
            tmax = fmaxf(tmax, sv);
        }
        const float mn = fmaxf(m, tmax);
        const float alpha = __expf(m - mn);     // first tile: exp(-inf) = 0
        l *= alpha;
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] *= alpha;
#pragma unroll
        for (int jj = 0; jj < ATT_TILE; ++jj) {
            const float p = __expf(s[jj] - mn);  // padded keys: exp(-inf) = 0
            l += p;
#pragma unroll
            for (int c = 0; c < D; ++c) o[c] = fmaf(p, sV[jj * DK + sub * D + c], o[c]);
        }
        m = mn;
    }
    if (active) {
        const float inv = 1.f / l;
        T* ob = out + ((long)b * Tq + i) * ldo + hh * DK + sub * D;
#pragma unroll
        for (int c = 0; c < D; ++c) ob[c] = from_f32<T>(o[c] * inv);
        if (sub == 0) lse[((long)b * h + hh) * Tq + i] = m + __logf(l);
    }
}

// ---------------------------------------------------------------- backward
// delta[b,h,i] = sum_c dout[b,i,h,c] * out[b,i,h,c]
template <typename T>
__global__ void mha_delta_kernel(const T* __restrict__ out, const T* __restrict__ dout, int ldo, int h, int Tq,
                                 int dk, long total /*B*h*Tq*/, float* __restrict__ delta) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int i = (int)(idx % Tq);
    const long t = idx / Tq;
    const int hh = (int)(t % h);
    const long b = t / h;
    const long o = (b * Tq + i) * ldo + hh * dk;
    float s = 0.f;
    for (int c = 0; c < dk; ++c) s += to_f32<T>(out[o + c]) * to_f32<T>(dout[o + c]);
    delta[idx] = s;
}

// dQ pass: a thread group per query row, keys/values streamed
template <typename T, int DK, int P>
__global__ __launch_bounds__(ATT_BLOCK) void mha_bwd_dq_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                               const T* __restrict__ v, int ldq, int ldk, int ldv,
                                                               const uint8_t* __restrict__ key_mask, int causal,
                                                               int h, int Tq, int Tk, float scale,
                                                               const T* __restrict__ dout, int ldo,
                                                               const float* __restrict__ lse,
                                                               const float* __restrict__ delta, T* __restrict__ dq,
                                                               int lddq) {
    constexpr int D = DK / P;
    constexpr int ROWS = ATT_BLOCK / P;
    __shared__ float sK[ATT_TILE * DK];
    __shared__ float sV[ATT_TILE * DK];
    __shared__ float sM[ATT_TILE];
    const int b = blockIdx.z, hh = blockIdx.y;
    const int sub = threadIdx.x % P;
    const int i = blockIdx.x * ROWS + threadIdx.x / P;
    const bool active = i < Tq;
    const T* kb = k + ((long)b * Tk) * ldk + hh * DK;
    const T* vb = v + ((long)b * Tk) * ldv + hh * DK;
    float qr[D], dor[D], acc[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        qr[c] = active ? to_f32<T>(q[((long)b * Tq + i) * ldq + hh * DK + sub * D + c]) : 0.f;
        dor[c] = active ? to_f32<T>(dout[((long)b * Tq + i) * ldo + hh * DK + sub * D + c]) : 0.f;
        acc[c] = 0.f;
    }
    const float li = active ? lse[((long)b * h + hh) * Tq + i] : 0.f;
    const float di = active ? delta[((long)b * h + hh) * Tq + i] : 0.f;
    for (int j0 = 0; j0 < Tk; j0 += ATT_TILE) {
        __syncthreads();
        stage_rows<T>(kb, ldk, j0, Tk, DK, sK);
        stage_rows<T>(vb, ldv, j0, Tk, DK, sV);
        if (threadIdx.x < ATT_TILE) {
            int j = j0 + threadIdx.x;
            sM[threadIdx.x] = (j < Tk && key_mask && key_mask[(long)b * Tk + j]) ? 1.f : 0.f;
        }
        __syncthreads();
        const int nj = min(ATT_TILE, Tk - j0);
        for (int jj = 0; jj < nj; ++jj) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                s = fmaf(qr[c], sK[jj * DK + sub * D + c], s);
                dp = fmaf(dor[c], sV[jj * DK + sub * D + c], dp);
            }
#pragma unroll
            for (int w = 1; w < P; w <<= 1) {
                s += __shfl_xor(s, w, 64);
                dp += __shfl_xor(dp, w, 64);
            }
            float sv = s * scale;
            if (sM[jj] != 0.f || (causal && (j0 + jj) > i + causal - 1)) sv += -1e9f;
            const float p = __expf(sv - li);
            const float ds = p * (dp - di) * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) acc[c] = fmaf(ds, sK[jj * DK + sub * D + c], acc[c]);
        }
    }
    if (active) {
        T* o = dq + ((long)b * Tq + i) * lddq + hh * DK + sub * D;
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] = from_f32<T>(acc[c]);
    }
}

// dK/dV pass: a thread group per key row, queries / dout streamed
template <typename T, int DK, int P>
__global__ __launch_bounds__(ATT_BLOCK) void mha_bwd_dkv_kernel(const T* __restrict__ q, const T* __restrict__ k,
                                                                const T* __restrict__ v, int ldq, int ldk, int ldv,
                                                                const uint8_t* __restrict__ key_mask, int causal,
                                                                int h, int Tq, int Tk, float scale,
                                                                const T* __restrict__ dout, int ldo,
                                                                const float* __restrict__ lse,
                                                                const float* __restrict__ delta, T* __restrict__ dk_,
                                                                T* __restrict__ dv, int lddk, int lddv) {
    constexpr int D = DK / P;
    constexpr int ROWS = ATT_BLOCK / P;
    __shared__ float sQ[ATT_TILE * DK];
    __shared__ float sO[ATT_TILE * DK];
    __shared__ float sL[ATT_TILE];
    __shared__ float sD[ATT_TILE];
    const int b = blockIdx.z, hh = blockIdx.y;
    const int sub = threadIdx.x % P;
    const int j = blockIdx.x * ROWS + threadIdx.x / P;
    const bool active = j < Tk;
    const T* qb = q + ((long)b * Tq) * ldq + hh * DK;
    const T* ob = dout + ((long)b * Tq) * ldo + hh * DK;
    float kr[D], vr[D], ak[D], av[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
        kr[c] = active ? to_f32<T>(k[((long)b * Tk + j) * ldk + hh * DK + sub * D + c]) : 0.f;
        vr[c] = active ? to_f32<T>(v[((long)b * Tk + j) * ldv + hh * DK + sub * D + c]) : 0.f;
        ak[c] = 0.f;
        av[c] = 0.f;
    }
    const bool kmask = active && key_mask && key_mask[(long)b * Tk + j];
    for (int i0 = 0; i0 < Tq; i0 += ATT_TILE) {
        __syncthreads();
        stage_rows<T>(qb, ldq, i0, Tq, DK, sQ);
        stage_rows<T>(ob, ldo, i0, Tq, DK, sO);
        if (threadIdx.x < ATT_TILE) {
            int i = i0 + threadIdx.x;
            sL[threadIdx.x] = i < Tq ? lse[((long)b * h + hh) * Tq + i] : 0.f;
            sD[threadIdx.x] = i < Tq ? delta[((long)b * h + hh) * Tq + i] : 0.f;
        }
        __syncthreads();
        const int ni = min(ATT_TILE, Tq - i0);
        for (int ii = 0; ii < ni; ++ii) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                s = fmaf(kr[c], sQ[ii * DK + sub * D + c], s);
                dp = fmaf(vr[c], sO[ii * DK + sub * D + c], dp);
            }
#pragma unroll
            for (int w = 1; w < P; w <<= 1) {
                s += __shfl_xor(s, w, 64);
                dp += __shfl_xor(dp, w, 64);
            }
            float sv = s * scale;
            if (kmask || (causal && j > (i0 + ii) + causal - 1)) sv += -1e9f;
            const float p = __expf(sv - sL[ii]);
            const float ds = p * (dp - sD[ii]) * scale;
#pragma unroll
            for (int c = 0; c < D; ++c) {
                av[c] = fmaf(p, sO[ii * DK + sub * D + c], av[c]);
                ak[c] = fmaf(ds, sQ[ii * DK + sub * D + c], ak[c]);
            }
        }
    }
    if (active) {
        T* o1 = dk_ + ((long)b * Tk + j) * lddk + hh * DK + sub * D;
        T* o2 = dv + ((long)b * Tk + j) * lddv + hh * DK + sub * D;
#pragma unroll
        for (int c = 0; c < D; ++c) {
            o1[c] = from_f32<T>(ak[c]);
            o2[c] = from_f32<T>(av[c]);
        }
    }
}

// ---------------------------------------------------------------- dispatch
static int mha_check(const char* name, int B, int h, int Tq, int Tk, int dk, int dtype) {
    ATHENA_REQUIRE(B > 0 && h > 0 && Tq > 0 && Tk > 0, ATHENA_E_SHAPE, "%s: bad shape", name);
    ATHENA_REQUIRE(dk == 8 || dk == 16 || dk == 32 || dk == 64 || dk == 128, ATHENA_E_UNSUPPORTED,
                   "%s: head size %d not in {8,16,32,64,128}", name, dk);
    ATHENA_REQUIRE(dtype == ATHENA_F32 || dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "%s: dtype", name);
    ATHENA_REQUIRE(B <= 65535 && h <= 65535, ATHENA_E_SHAPE, "%s: grid too large", name);
    return ATHENA_OK;
}

// the matrix-core kernels take bf16, head sizes 32/64/128 and 16-byte aligned rows
static bool mha_mfma_ok(int dtype, int dk, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (dtype != ATHENA_BF16 || !(dk == 32 || dk == 64 || dk == 128)) return false;
    for (int ld : lds)
        if (ld % 8) return false;
    for (const void* p : ptrs)
        if (!athena_aligned16(p)) return false;
    return true;
}

// the causal test of the scores is compiled in only where it is asked for
static bool mha_plain_order() {
    static const bool v = getenv("ATHENA_MHA_PLAIN_ORDER") && atoi(getenv("ATHENA_MHA_PLAIN_ORDER")) != 0;
    return v;
}

static bool mha_fwd_halves() {
    static const bool v = !(getenv("ATHENA_MHA_FWD_HALVES") && atoi(getenv("ATHENA_MHA_FWD_HALVES")) == 0);
    return v;
}

#define MHA_MFMA_DISPATCH(KERNEL, GRID, ...)                                                              \
    do {                                                                                                  \
        if (causal) {                                                                                     \
            if (dk == 32) ATHENA_LAUNCH((KERNEL<32, true>), GRID, dim3(256), 0, s, __VA_ARGS__);      \
            else if (dk == 64) ATHENA_LAUNCH((KERNEL<64, true>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else ATHENA_LAUNCH((KERNEL<128, true>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
        } else {                                                                                          \
            if (dk == 32) ATHENA_LAUNCH((KERNEL<32, false>), GRID, dim3(256), 0, s, __VA_ARGS__);     \
            else if (dk == 64) ATHENA_LAUNCH((KERNEL<64, false>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else ATHENA_LAUNCH((KERNEL<128, false>), GRID, dim3(256), 0, s, __VA_ARGS__);             \
        }                                                                                                 \
    } while (0)

// backward launch: RS sets of 16 queries per wave of a dQ workgroup (two only for head sizes up to 64
// without a causal band: with one the two-set build spills)
#define MHA_BWD_DISPATCH(RS, GRID, ...)                                                                          \
    do {                                                                                                         \
        if (causal) {                                                                                            \
            if (dk == 32) ATHENA_LAUNCH((mha_bwd_mfma_kernel<32, true, 1>), GRID, dim3(256), 0, s, __VA_ARGS__);       \
            else if (dk == 64) ATHENA_LAUNCH((mha_bwd_mfma_kernel<64, true, 1>), GRID, dim3(256), 0, s, __VA_ARGS__);  \
            else ATHENA_LAUNCH((mha_bwd_mfma_kernel<128, true, 1>), GRID, dim3(256), 0, s, __VA_ARGS__);               \
        } else {                                                                                                 \
            if (dk == 32) ATHENA_LAUNCH((mha_bwd_mfma_kernel<32, false, RS>), GRID, dim3(256), 0, s, __VA_ARGS__);     \
            else if (dk == 64) ATHENA_LAUNCH((mha_bwd_mfma_kernel<64, false, RS>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else ATHENA_LAUNCH((mha_bwd_mfma_kernel<128, false, 1>), GRID, dim3(256), 0, s, __VA_ARGS__);              \
        }                                                                                                        \
    } while (0)

#define MHA_DISPATCH(KERNEL, T, ...)                                                                          \
    do {                                                                                                      \
        switch (dk) {                                                                                         \
            case 8: { constexpr int P = 1; dim3 g((rows + ATT_BLOCK / P - 1) / (ATT_BLOCK / P), h, B);        \
                ATHENA_LAUNCH((KERNEL<T, 8, P>), g, dim3(ATT_BLOCK), 0, s, __VA_ARGS__); } break;        \
            case 16: { constexpr int P = 1; dim3 g((rows + ATT_BLOCK / P - 1) / (ATT_BLOCK / P), h, B);       \
                ATHENA_LAUNCH((KERNEL<T, 16, P>), g, dim3(ATT_BLOCK), 0, s, __VA_ARGS__); } break;       \
            case 32: { constexpr int P = 1; dim3 g((rows + ATT_BLOCK / P - 1) / (ATT_BLOCK / P), h, B);       \
                ATHENA_LAUNCH((KERNEL<T, 32, P>), g, dim3(ATT_BLOCK), 0, s, __VA_ARGS__); } break;       \
            case 64: { constexpr int P = 2; dim3 g((rows + ATT_BLOCK / P - 1) / (ATT_BLOCK / P), h, B);       \
                ATHENA_LAUNCH((KERNEL<T, 64, P>), g, dim3(ATT_BLOCK), 0, s, __VA_ARGS__); } break;       \
            default: { constexpr int P = 4; dim3 g((rows + ATT_BLOCK / P - 1) / (ATT_BLOCK / P), h, B);       \
                ATHENA_LAUNCH((KERNEL<T, 128, P>), g, dim3(ATT_BLOCK), 0, s, __VA_ARGS__); } break;      \
        }                                                                                                     \
    } while (0)

extern "C" int athena_mha_fwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                              const uint8_t* key_mask, int causal, int B, int h, int Tq, int Tk, int dk, int dtype,
                              float scale, void* out, int ldo, float* lse, athena_stream_t stream) {
    int rc = mha_check("athena_mha_fwd", B, h, Tq, Tk, dk, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(q && k && v && out && lse, ATHENA_E_SHAPE, "athena_mha_fwd: null operand");
    ATHENA_REQUIRE(ldq >= h * dk && ldk >= h * dk && ldv >= h * dk && ldo >= h * dk, ATHENA_E_SHAPE, "athena_mha_fwd: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    const int rows = Tq;
    if (mha_mfma_ok(dtype, dk, {ldq, ldk, ldv, ldo}, {q, k, v, out})) {
        const long heads8 = ((long)B * h + 7) / 8 * 8;
        ATHENA_REQUIRE(heads8 * ((Tq + AM_ROWS - 1) / AM_ROWS) < (1L << 31), ATHENA_E_SHAPE, "athena_mha_fwd: too many blocks");
        const dim3 grid((unsigned)(heads8 * ((Tq + AM_ROWS - 1) / AM_ROWS)));
        // two halves of the key blocks side by side in one workgroup where there are at least three
        // blocks (and the two sets of images fit: head size up to 64)
        const bool halves = dk <= 64 && (Tk + AM_ROWS - 1) / AM_ROWS >= 3 && mha_fwd_halves();
#define MHA_FWD(DKV, CA)                                                                                          \
    do {                                                                                                          \
        if (halves)                                                                                               \
            ATHENA_LAUNCH((mha_fwd_mfma_kernel<DKV, CA, 2>), grid, dim3(512), 0, s, (const bf16_t*)q, (const bf16_t*)k, \
                          (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (bf16_t*)out, ldo, lse,  \
                          mha_plain_order() ? -(B * h) : B * h);                                                  \
        else                                                                                                      \
            ATHENA_LAUNCH((mha_fwd_mfma_kernel<DKV, CA, 1>), grid, dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k, \
                          (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (bf16_t*)out, ldo, lse,  \
                          mha_plain_order() ? -(B * h) : B * h);                                                  \
    } while (0)
        if (causal) {
            if (dk == 32) MHA_FWD(32, true);
            else if (dk == 64) MHA_FWD(64, true);
            else ATHENA_LAUNCH((mha_fwd_mfma_kernel<128, true, 1>), grid, dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (bf16_t*)out, ldo, lse,
                               mha_plain_order() ? -(B * h) : B * h);
        } else {
            if (dk == 32) MHA_FWD(32, false);
            else if (dk == 64) MHA_FWD(64, false);
            else ATHENA_LAUNCH((mha_fwd_mfma_kernel<128, false, 1>), grid, dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                               (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (bf16_t*)out, ldo, lse,
                               mha_plain_order() ? -(B * h) : B * h);
        }
#undef MHA_FWD
        ATHENA_CHECK_LAUNCH("athena_mha_fwd");
        return ATHENA_OK;
    }
    if (dtype == ATHENA_BF16)
        MHA_DISPATCH(mha_fwd_kernel, bf16_t, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (bf16_t*)out, ldo, lse);
    else
        MHA_DISPATCH(mha_fwd_kernel, float, (const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (float*)out, ldo, lse);
    ATHENA_CHECK_LAUNCH("athena_mha_fwd");
    return ATHENA_OK;
}

extern "C" int athena_mha_bwd(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                              const uint8_t* key_mask, int causal, int B, int h, int Tq, int Tk, int dk, int dtype,
                              float scale, const void* out, const void* dout, int ldo, const float* lse, float* delta,
                              void* dq, void* dk_, void* dv, int lddq, int lddk, int lddv, athena_stream_t stream) {
    int rc = mha_check("athena_mha_bwd", B, h, Tq, Tk, dk, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(q && k && v && out && dout && lse && delta && dq && dk_ && dv, ATHENA_E_SHAPE, "athena_mha_bwd: null operand");
    ATHENA_REQUIRE(ldq >= h * dk && ldk >= h * dk && ldv >= h * dk && ldo >= h * dk && lddq >= h * dk && lddk >= h * dk && lddv >= h * dk,
                   ATHENA_E_SHAPE, "athena_mha_bwd: leading dimension too small");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * h * Tq;
    dim3 gd((unsigned)((total + 255) / 256));
    if (mha_mfma_ok(dtype, dk, {ldq, ldk, ldv, ldo, lddq, lddk, lddv}, {q, k, v, out, dout, dq, dk_, dv})) {
        const long heads8 = ((long)B * h + 7) / 8 * 8;
        // dQ workgroups of 128 queries (two sets of 16 per wave) once there are more than 96 of them
        static const int rs_env = getenv("ATHENA_MHA_BWD_RS") ? atoi(getenv("ATHENA_MHA_BWD_RS")) : 0;
        const int rs = (dk > 64 || causal) ? 1 : (rs_env == 1 || rs_env == 2) ? rs_env : (Tq > 96 ? 2 : 1);
        const long nqb = (Tq + AM_ROWS * rs - 1) / (AM_ROWS * rs), nkb = (Tk + AM_ROWS - 1) / AM_ROWS;
        ATHENA_REQUIRE(heads8 * (nqb + nkb) < (1L << 31), ATHENA_E_SHAPE, "athena_mha_bwd: too many blocks");
        static const bool split = getenv("ATHENA_MHA_BWD_SPLIT") && atoi(getenv("ATHENA_MHA_BWD_SPLIT")) != 0;
#define MHA_BWD_ARGS(HEADS, ONLY)                                                                                   \
    (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale,        \
        (const bf16_t*)out, (const bf16_t*)dout, ldo, lse, delta, (bf16_t*)dq, (bf16_t*)dk_, (bf16_t*)dv, lddq, lddk, \
        lddv, HEADS, ONLY
        if (split) {
            const dim3 gk((unsigned)(heads8 * nkb)), gq((unsigned)(heads8 * nqb));
            if (rs == 2) {
                MHA_BWD_DISPATCH(2, gk, MHA_BWD_ARGS(B * h, 1));
                MHA_BWD_DISPATCH(2, gq, MHA_BWD_ARGS(B * h, 2));
            } else {
                MHA_BWD_DISPATCH(1, gk, MHA_BWD_ARGS(B * h, 1));
                MHA_BWD_DISPATCH(1, gq, MHA_BWD_ARGS(B * h, 2));
            }
            ATHENA_CHECK_LAUNCH("athena_mha_bwd");
            return ATHENA_OK;
        }
        const dim3 grid((unsigned)(heads8 * (nqb + nkb)));
        const int heads = mha_plain_order() ? -(B * h) : B * h;
        if (rs == 2) MHA_BWD_DISPATCH(2, grid, MHA_BWD_ARGS(heads, 0));
        else MHA_BWD_DISPATCH(1, grid, MHA_BWD_ARGS(heads, 0));
#undef MHA_BWD_ARGS
        ATHENA_CHECK_LAUNCH("athena_mha_bwd");
        return ATHENA_OK;
    }
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(mha_delta_kernel<bf16_t>, gd, dim3(256), 0, s, (const bf16_t*)out, (const bf16_t*)dout, ldo, h, Tq, dk, total, delta);
    else
        ATHENA_LAUNCH(mha_delta_kernel<float>, gd, dim3(256), 0, s, (const float*)out, (const float*)dout, ldo, h, Tq, dk, total, delta);
    {
        const int rows = Tq;
        if (dtype == ATHENA_BF16)
            MHA_DISPATCH(mha_bwd_dq_kernel, bf16_t, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (const bf16_t*)dout, ldo, lse, delta, (bf16_t*)dq, lddq);
        else
            MHA_DISPATCH(mha_bwd_dq_kernel, float, (const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (const float*)dout, ldo, lse, delta, (float*)dq, lddq);
    }
    {
        const int rows = Tk;
        if (dtype == ATHENA_BF16)
            MHA_DISPATCH(mha_bwd_dkv_kernel, bf16_t, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (const bf16_t*)dout, ldo, lse, delta, (bf16_t*)dk_, (bf16_t*)dv, lddk, lddv);
        else
            MHA_DISPATCH(mha_bwd_dkv_kernel, float, (const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, key_mask, causal, h, Tq, Tk, scale, (const float*)dout, ldo, lse, delta, (float*)dk_, (float*)dv, lddk, lddv);
    }
    ATHENA_CHECK_LAUNCH("athena_mha_bwd");
    return ATHENA_OK;
}

// ------------------------------------------------------------------ attention weights, on request
// weights[b, h, i, :] = softmax_j(q_i . k_j * scale + mask) as a tensor in memory: what
// MultiHeadAttention.call returns beside its output (athena/layers/attention.py:161,171).  The training
// step never asks for it (the fused kernels above keep the probabilities on chip); this is the
// inspection path, one workgroup per (query, head), fp32 arithmetic.
template <typename T>
__global__ __launch_bounds__(256) void mha_weights_kernel(const T* __restrict__ q, const T* __restrict__ k, int ldq,
                                                          int ldk, const uint8_t* __restrict__ key_mask, int causal, int h,
                                                          int Tq, int Tk, int dk, float scale, float* __restrict__ w) {
    __shared__ float red[32];
    __shared__ float qs[256];
    const int i = blockIdx.x, bh = blockIdx.y, b = bh / h, hd = bh % h;
    const T* qr = q + ((long)b * Tq + i) * ldq + hd * dk;
    for (int c = threadIdx.x; c < dk; c += 256) qs[c] = to_f32<T>(qr[c]);
    __syncthreads();
    float* wr = w + ((long)bh * Tq + i) * Tk;
    float mx = -INFINITY;
    for (int j = threadIdx.x; j < Tk; j += 256) {
        const T* kr = k + ((long)b * Tk + j) * ldk + hd * dk;
        float s = 0.f;
        for (int c = 0; c < dk; ++c) s += qs[c] * to_f32<T>(kr[c]);
        s *= scale;
        if (causal > 0 && j > i + causal - 1) s += -1e9f;
        if (key_mask && key_mask[(long)b * Tk + j]) s += -1e9f;
        wr[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int j = threadIdx.x; j < Tk; j += 256) {
        const float e = __expf(wr[j] - mx);
        wr[j] = e;
        sum += e;
    }
    sum = block_sum(sum, red);
    const float inv = 1.f / sum;
    for (int j = threadIdx.x; j < Tk; j += 256) wr[j] *= inv;
}

extern "C" int athena_mha_weights(const void* q, const void* k, int ldq, int ldk, const uint8_t* key_mask, int causal,
                                  int B, int h, int Tq, int Tk, int dk, int dtype, float scale, float* weights,
                                  athena_stream_t stream) {
    ATHENA_REQUIRE(q && k && weights, ATHENA_E_SHAPE, "athena_mha_weights: null operand");
    ATHENA_REQUIRE(B > 0 && h > 0 && Tq > 0 && Tk > 0 && dk > 0 && dk <= 256 && (long)B * h <= 65535 && causal >= 0 &&
                       ldq >= h * dk && ldk >= h * dk,
                   ATHENA_E_SHAPE, "athena_mha_weights: bad shape B=%d h=%d Tq=%d Tk=%d dk=%d", B, h, Tq, Tk, dk);
    const dim3 grid(Tq, B * h);
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(mha_weights_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k,
                      ldq, ldk, key_mask, causal, h, Tq, Tk, dk, scale, weights);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(mha_weights_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)q, (const float*)k,
                      ldq, ldk, key_mask, causal, h, Tq, Tk, dk, scale, weights);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_mha_weights: dtype");
    ATHENA_CHECK_LAUNCH("athena_mha_weights");
    return ATHENA_OK;
}
