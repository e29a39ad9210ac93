// CTC loss (log-softmax + alpha/beta over the blank-augmented lattice + gradient), the
// label-smoothed cross-entropy, and row argmax.  Bandwidth-bound kernels: each logits row is
// streamed with 16-byte loads and reduced inside the wavefront by DPP steps.
//
// CTC is three launches (loss alone, or loss and gradient from one pass over the logits):
//   ctc_row     : one workgroup per logits row, the row in registers: logsumexp, first arg-max, the
//                 emissions of the row's label states (a table of doubles, relative to blank) and,
//                 on the one-pass path, scale * softmax written as the dense part of the gradient
//   ctc_lattice : one workgroup per sample: alpha and beta recursions in two wavefronts (states in
//                 registers, probabilities in double precision with power-of-two rescaling), two
//                 more wavefronts copying the state vectors from an LDS ring to memory; a sample
//                 without a path, with a state beyond the double range, or on which the finite
//                 log-zeros of TensorFlow's dense branch would show in the result is redone on
//                 logarithms the way that branch does it
//   ctc_post    : one wave per logits row: the state posterior alpha*beta, normalised, as a short
//                 list (blank, label 1, label 2, ...), the loss of each sample, and on the one-pass
//                 path the gradient columns that have an occupancy
//   ctc_grad    : the separate backward pass over the posterior list: one workgroup per row, the
//                 list scattered by class into an LDS accumulator, grad = scale * (softmax -
//                 occupancy) streamed out
#include "common.h"

#include <mutex>

#define CTC_LOGZERO (-1e30f)
#define CTC_MAX_R 16            // lattice states per lane: supports 2U+1 <= 1024
#define CTC_MAX_U (64 * CTC_MAX_R / 2)   // label positions
#define CTC_EMIS_PAD 16         // rows of padding in front of and behind the emission table
// the beta rows of the R = 2 lattice are stored shifted by one state (see ctc_lattice_kernel)
#define CTC_BETA_SHIFT(R) ((R) == 2)
#define CTC_LOG2E 1.4426950408889634f
#define CTC_LN2 0.6931471805599453

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// states per lane the lattice kernel is instantiated for
static int ctc_launch_r(int S) {
    const int R = (S + 63) / 64;
    static const int sizes[] = {2, 4, 6, 8, 12, CTC_MAX_R};    // even: a lane then holds whole (blank, label) pairs
    for (int v : sizes)
        if (R <= v) return v;
    return CTC_MAX_R;
}

struct CtcWorkspace {
    float* lse;      // [B*T]     natural-log normaliser of each row (softmax of the gradient)
    float* nrm;      // [B*T]     lse - logit[blank]: the normaliser relative to the blank logit
    double* alpha;   // [B*T*Sp]  alpha up to a per-(b,t) power of two (emission at t included)
    double* beta;    // [B*T*Sp]  beta  up to a per-(b,t) power of two (emission at t excluded)
    double* emis;    // [B*T*Sp/2] emission of every label state relative to blank (0 beyond the last label)
    float* occ;      // [B*T*Up]  state posterior of a row: [0] all blank states, [1+u] label u
    int32_t* first;  // [B*U]     first label position with the same class (-1: the class is the blank)
    double* ll;      // [B]       log2 of the path sum over blank-relative emissions (-inf: no path)
    int32_t* flags;  // [B]       1: no alignment fits; occ then holds TensorFlow's leaky lattice
    int Sp;          // row pitch of alpha / beta: 64 * states per lane
    int Up;          // row pitch of occ: U + 1 rounded up to 4
};

static size_t ctc_ws_layout(int B, int T, int U, void* base, CtcWorkspace* ws) {
    const int Sp = 64 * ctc_launch_r(2 * U + 1);
    const int Up = (U + 1 + 3) / 4 * 4;
    size_t off = 0;
    char* p = (char*)base;
    size_t n_row = align_up(sizeof(float) * (size_t)B * T, 256);
    size_t n_lat = align_up(sizeof(double) * (size_t)B * T * Sp, 256);
    if (ws) ws->lse = (float*)(p + off);
    off += n_row;
    if (ws) ws->nrm = (float*)(p + off);
    off += n_row;
    if (ws) ws->alpha = (double*)(p + off);
    off += n_lat;
    if (ws) ws->beta = (double*)(p + off);
    off += n_lat;
    // (the lattice prefetches a few rows past either end of a sample's part)
    if (ws) ws->emis = (double*)(p + off) + (size_t)CTC_EMIS_PAD * (Sp / 2);
    off += align_up(sizeof(double) * ((size_t)B * T + 2 * CTC_EMIS_PAD) * (Sp / 2), 256);
    if (ws) ws->occ = (float*)(p + off);
    off += align_up(sizeof(float) * (size_t)B * T * Up, 256);
    if (ws) ws->first = (int32_t*)(p + off);
    off += align_up(sizeof(int32_t) * (size_t)B * (U > 0 ? U : 1), 256);
    if (ws) ws->ll = (double*)(p + off);
    off += align_up(sizeof(double) * (size_t)B, 256);
    if (ws) ws->flags = (int32_t*)(p + off);
    off += align_up(sizeof(int32_t) * (size_t)B, 256);
    if (ws) {
        ws->Sp = Sp;
        ws->Up = Up;
    }
    return off;
}

extern "C" size_t athena_ctc_workspace_bytes(int B, int T, int U) {
    if (B <= 0 || T <= 0 || U < 0) return 0;
    return ctc_ws_layout(B, T, U, nullptr, nullptr);
}

// ---------------------------------------------------------------- row logsumexp helper
// online (max, sum) over a row of V logits, 256 threads, 4-wide loads where aligned
template <typename T>
__device__ __forceinline__ void row_max_sum(const T* __restrict__ row, int V, float& m_out, float& s_out,
                                            float* red) {
    float m = -INFINITY, s = 0.f;
    for (int c = threadIdx.x; c < V; c += blockDim.x) {
        float x = to_f32<T>(row[c]);
        if (x > m) {
            s = s * __expf(m - x) + 1.f;
            m = x;
        } else {
            s += __expf(x - m);
        }
    }
    const float gm = block_max(m, red);
    float contrib = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
    s_out = block_sum(contrib, red);
    m_out = gm;
}

template <>
__device__ __forceinline__ void row_max_sum<float>(const float* __restrict__ row, int V, float& m_out, float& s_out,
                                                   float* red) {
    float m = -INFINITY, s = 0.f;
    const int V4 = V & ~3;
    // rows start 16-byte aligned (ld % 4 == 0 is checked by the host)
    for (int c = threadIdx.x * 4; c < V4; c += blockDim.x * 4) {
        float4 x = *(const float4*)(row + c);
        float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
        if (mx > m) {
            s *= __expf(m - mx);
            m = mx;
        }
        s += __expf(x.x - m) + __expf(x.y - m) + __expf(x.z - m) + __expf(x.w - m);
    }
    for (int c = V4 + threadIdx.x; c < V; c += blockDim.x) {
        float x = row[c];
        if (x > m) {
            s = s * __expf(m - x) + 1.f;
            m = x;
        } else {
            s += __expf(x - m);
        }
    }
    const float gm = block_max(m, red);
    float contrib = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
    s_out = block_sum(contrib, red);
    m_out = gm;
}

// the same online (max, sum) that also tracks the first column holding the row maximum (the greedy
// CTC path of the accuracy metric, reference athena/metrics.py:121-132, for no extra pass over the row)
template <typename T>
__device__ __forceinline__ void row_max_sum_arg(const T* __restrict__ row, int V, float& m_out, float& s_out,
                                                int& arg_out, float* red) {
    float m = -INFINITY, s = 0.f;
    int idx = 0x7fffffff;
    if constexpr (sizeof(T) == 4) {
        const int V4 = V & ~3;
        for (int c = threadIdx.x * 4; c < V4; c += blockDim.x * 4) {
            const float4 x = *(const float4*)((const float*)row + c);
            const float mx = fmaxf(fmaxf(x.x, x.y), fmaxf(x.z, x.w));
            if (mx > m) {
                s *= __expf(m - mx);
                m = mx;
                idx = c + (x.x == mx ? 0 : x.y == mx ? 1 : x.z == mx ? 2 : 3);
            }
            s += __expf(x.x - m) + __expf(x.y - m) + __expf(x.z - m) + __expf(x.w - m);
        }
        for (int c = V4 + threadIdx.x; c < V; c += blockDim.x) {
            const float x = ((const float*)row)[c];
            if (x > m) {
                s = s * __expf(m - x) + 1.f;
                m = x;
                idx = c;
            } else {
                s += __expf(x - m);
            }
        }
    } else {
        for (int c = threadIdx.x; c < V; c += blockDim.x) {
            const float x = to_f32<T>(row[c]);
            if (x > m) {
                s = s * __expf(m - x) + 1.f;
                m = x;
                idx = c;
            } else {
                s += __expf(x - m);
            }
        }
    }
    const float gm = block_max(m, red);
    const float contrib = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
    s_out = block_sum(contrib, red);
    // smallest column among the threads that saw the maximum (column indices are exact in fp32)
    const float cand = (m == gm && idx != 0x7fffffff) ? -(float)idx : -INFINITY;
    const float best = block_max(cand, red);
    arg_out = best == -INFINITY ? 0 : (int)(-best);
    m_out = gm;
}

// sum / maximum over the wave: four DPP steps inside each row of 16 lanes (quad swaps, half mirror,
// mirror), then the four row results through scalar registers -- no trip through the LDS crossbar
__device__ __forceinline__ float wave_fsum_dpp(float v) {
    v = dpp_fadd<0xB1>(v);
    v = dpp_fadd<0x4E>(v);
    v = dpp_fadd<0x141>(v);
    v = dpp_fadd<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
template <int CTRL>
__device__ __forceinline__ float dpp_fmax(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)));
}
__device__ __forceinline__ float wave_fmax_dpp(float v) {
    v = dpp_fmax<0xB1>(v);
    v = dpp_fmax<0x4E>(v);
    v = dpp_fmax<0x141>(v);
    v = dpp_fmax<0x140>(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
// fp32 rows of at most CTC_ROW_REGS * 1024 columns with ld % 4 == 0: the whole row is fetched into
// registers before anything depends on it (all loads of a thread in flight at once instead of one
// per trip of an online loop), then maximum, first arg-max and the sum of exponentials relative to
// the workgroup's maximum are taken from the registers, which stay with the caller (columns
// V .. ld-1 are padding and come back as -inf).  Two workgroup reductions: the maximum, then the
// sum and the arg-max candidates together.
#define CTC_ROW_REGS 8
template <int NR>
__device__ __forceinline__ void row_regs_load(const float* __restrict__ row, int ld, float4 (&x)[NR]) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        x[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (c < ld) x[i] = *(const float4*)(row + c);
    }
}
// Statistics of a row held in registers behind ONE workgroup barrier: every wave reduces its own part
// by lane shuffles (maximum, first column holding it, sum of exponentials relative to the wave's
// maximum), lane 0 of each wave leaves the three numbers in LDS, and behind the barrier every thread
// combines the four waves' numbers for itself.  ``slot``: 12 floats of LDS.
// kKeep: the registers come back holding exp(x - the wave's maximum), the exponentials the sum was
// made of, and ``w_out`` the factor exp(wave maximum - row maximum) that turns them into
// exp(x - row maximum): a caller that goes on to write the softmax needs no second exponential.
template <bool kKeep, int NR>
__device__ __forceinline__ void row_regs_stats(float4 (&x)[NR], int V, float& m_out, float& s_out,
                                               int& arg_out, float* slot, float& w_out) {
    float m = -INFINITY;
    int idx = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        if (c + 3 >= V) {
            if (c + 0 >= V) x[i].x = -INFINITY;
            if (c + 1 >= V) x[i].y = -INFINITY;
            if (c + 2 >= V) x[i].z = -INFINITY;
            x[i].w = -INFINITY;
        }
        const float mx = fmaxf(fmaxf(x[i].x, x[i].y), fmaxf(x[i].z, x[i].w));
        if (mx > m) {
            m = mx;
            idx = c + (x[i].x == mx ? 0 : x[i].y == mx ? 1 : x[i].z == mx ? 2 : 3);
        }
    }
    const float wm = wave_fmax_dpp(m);
    float s = 0.f;
    if (wm != -INFINITY) {
        const float nwm = -wm * CTC_LOG2E;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            // exp(x - wm) = 2^(x log2(e) - wm log2(e)): one multiply-add and the exponential unit
            const float4 e = make_float4(__builtin_amdgcn_exp2f(fmaf(x[i].x, CTC_LOG2E, nwm)),
                                         __builtin_amdgcn_exp2f(fmaf(x[i].y, CTC_LOG2E, nwm)),
                                         __builtin_amdgcn_exp2f(fmaf(x[i].z, CTC_LOG2E, nwm)),
                                         __builtin_amdgcn_exp2f(fmaf(x[i].w, CTC_LOG2E, nwm)));
            s += (e.x + e.y) + (e.z + e.w);
            if constexpr (kKeep) x[i] = e;
        }
    } else if constexpr (kKeep) {
#pragma unroll
        for (int i = 0; i < NR; ++i) x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    s = wave_fsum_dpp(s);
    // smallest column among the lanes that saw the wave's maximum (column indices are exact in fp32)
    const float cand = wave_fmax_dpp(m == wm && idx != 0x7fffffff ? -(float)idx : -INFINITY);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) {
        slot[w] = wm;
        slot[4 + w] = s;
        slot[8 + w] = cand;
    }
    __syncthreads();
    const float gm = fmaxf(fmaxf(slot[0], slot[1]), fmaxf(slot[2], slot[3]));
    float tot = 0.f, best = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float wk = slot[k];
        if (wk != -INFINITY) tot += slot[4 + k] * __expf(wk - gm);
        if (wk == gm) best = fmaxf(best, slot[8 + k]);
    }
    m_out = gm;
    s_out = tot;
    arg_out = best == -INFINITY ? 0 : (int)(-best);
    w_out = wm != -INFINITY ? __expf(wm - gm) : 0.f;
}

// ---------------------------------------------------------------- lattice arithmetic
// The recursions run on probabilities (not their logarithms) in double precision: a step is two
// additions and one multiplication per state instead of a log-sum-exp of three terms, and the only
// transcendental left is the exponential of the emission, which does not sit on the dependency
// chain.  Every fourth step the state vector is scaled by a power of two (exact) so that its
// largest element stays near 1; an emission lies in [2^-126, 2^128], so nothing can overflow
// between two scalings, and what underflows is below 2^-500 of the row's largest state.  A sample
// whose path sum still comes out as zero or not finite is handed to the slow path of ctc_post, which
// works on logarithms.

// value held by the previous / next lane of the wave (0 where there is none): DPP moves
// (wave_shr:1 / wave_shl:1, lanes without a source read 0) on the two halves instead of a trip
// through the LDS crossbar
__device__ __forceinline__ double lane_prev_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next_f64(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_imax(int v) {
    return max(v, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false));
}
// maximum over the wave: four DPP steps inside each row of 16 lanes (quad swaps, half mirror,
// mirror), then the four row results through scalar registers
__device__ __forceinline__ int wave_imax_dpp(int v) {
    v = dpp_imax<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_imax<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_imax<0x141>(v);     // row_half_mirror
    v = dpp_imax<0x140>(v);     // row_mirror
    const int r0 = __builtin_amdgcn_readlane(v, 0);
    const int r1 = __builtin_amdgcn_readlane(v, 16);
    const int r2 = __builtin_amdgcn_readlane(v, 32);
    const int r3 = __builtin_amdgcn_readlane(v, 48);
    return max(max(r0, r1), max(r2, r3));
}
// The high words of non-negative doubles order like the doubles.  From the largest high word of a
// set: the power of two that brings its owner into [1, 2) (0 if the set is all zero or not finite).
__device__ __forceinline__ int scale_exponent(int max_hi) {
    const int e = (max_hi >> 20) & 0x7ff;
    return (e == 0 || e == 0x7ff) ? 0 : e - 1023;
}

// R consecutive doubles per lane, as the widest stores the count allows
template <int R>
__device__ __forceinline__ void store_states(double* __restrict__ dst, const double (&a)[R]) {
    if constexpr (R % 2 == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 2) *(double2*)(dst + r) = make_double2(a[r], a[r + 1]);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) dst[r] = a[r];
    }
}

// ---------------------------------------------------------------- ctc rows
template <typename TL>
struct CtcRowVec;
template <>
struct CtcRowVec<float> {
    float4 v;
    __device__ __forceinline__ void load(const float* p) { v = *(const float4*)p; }
    __device__ __forceinline__ float get(int i) const { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
};
template <>
struct CtcRowVec<bf16_t> {
    uint2 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *(const uint2*)p; }
    __device__ __forceinline__ float get(int i) const {
        const uint32_t w = i < 2 ? v.x : v.y;
        return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
    }
};
template <typename TG>
__device__ __forceinline__ void ctc_store4(TG* p, float a, float b, float c, float d);
template <>
__device__ __forceinline__ void ctc_store4<float>(float* p, float a, float b, float c, float d) {
    *(float4*)p = make_float4(a, b, c, d);
}
template <>
__device__ __forceinline__ void ctc_store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
    union { bf16_t h[4]; uint2 u; } pk;
    pk.h[0] = (bf16_t)a; pk.h[1] = (bf16_t)b; pk.h[2] = (bf16_t)c; pk.h[3] = (bf16_t)d;
    *(uint2*)p = pk.u;
}

template <typename TG>
__device__ __forceinline__ void ctc_store1(TG* p, float a) { *p = (TG)a; }

// First CTC launch: one workgroup per logits row.
//   * logsumexp of the row (lse, and nrm = lse - logit[blank]) and its first arg-max (the greedy path
//     of the accuracy metric, reference athena/metrics.py:121-132)
//   * the emissions the lattice will need, gathered while the row is at hand: one double per label,
//     p[u] = exp(logit[label u] - logit[blank]) (0 beyond the sample's last label; a blank state has
//     factor 1 by construction), so that the recursion reads R/2 consecutive doubles per lane
//     instead of gathering 2U+1 cache lines per step
//   * kGrad: the dense part of the gradient, gscale * softmax(row), written while the row is in
//     registers (the padding columns and the rows beyond logit_length as zeros); ctc_post subtracts
//     the occupancy at the few columns that have one
//   NR: fp32 rows of up to NR * 1024 columns are held in registers (four columns per thread and
//       register set); the launch picks the smallest of the sizes built that fits the row, since the
//       statistics and the softmax are worked out for every register set whether the row reaches it or not
template <typename T, typename TG, bool kGrad, int NR>
__global__ __launch_bounds__(256) void ctc_row_kernel(const T* __restrict__ logits, int ldl,
                                                      const int32_t* __restrict__ labels,
                                                      const int32_t* __restrict__ label_len,
                                                      const int32_t* __restrict__ logit_len, int T_, int V, int U,
                                                      int blank, int Sp, float* __restrict__ lse,
                                                      float* __restrict__ nrm, double* __restrict__ emis,
                                                      int32_t* __restrict__ argmax,
                                                      int32_t* __restrict__ first, float gscale,
                                                      TG* __restrict__ grad, int lds_gather) {
    // lds_gather (with a row held in registers): the row is also laid down in LDS (ld floats of dynamic
    // shared memory) and the logits of the label columns are picked up there.  With a few dozen labels
    // the gather from memory hides behind the reductions; with 200 of them on each of 16 000 rows it is
    // three million scattered requests, which the memory system takes longer over than over the rows.
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];
    __shared__ float red[32];
    __shared__ int32_t labs[CTC_MAX_U];
    const int row = blockIdx.x;
    const int b = row / T_, t = row - b * T_;
    if (row == 0 && threadIdx.x == 0) emis[-1] = 0.0;      // read (and multiplied by zero) by the shifted beta recursion
    if (t == 0) {
        // labels that repeat share one logits column: ctc_post sums their occupancies at the first of them
        for (int u = threadIdx.x; u < U; u += 256) labs[u] = labels[(long)b * U + u];
        __syncthreads();
        for (int u = threadIdx.x; u < U; u += 256) {
            const int c = labs[u];
            int f = u;
            for (int u2 = u - 1; u2 >= 0; --u2) f = labs[u2] == c ? u2 : f;
            first[(long)b * U + u] = c == blank ? -1 : f;
        }
    }
    if (t >= logit_len[b]) {
        if (argmax && threadIdx.x == 0) argmax[row] = 0;
        if constexpr (kGrad) {
            TG* gr = grad + (long)row * ldl;
            for (int c = threadIdx.x * 4; c < ldl; c += 1024) ctc_store4<TG>(gr + c, 0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const T* lr = logits + (long)row * ldl;
    bool in_regs = false;
    if constexpr (sizeof(T) == 4) in_regs = ldl <= NR * 1024;
    float4 x[NR];
    if constexpr (sizeof(T) == 4) {
        if (in_regs) row_regs_load<NR>((const float*)lr, ldl, x);
    }
    const bool from_lds = in_regs && lds_gather != 0;        // (uniform)
    // the emissions are requested behind the row, so they arrive during its reductions
    const int L = min(max(label_len[b], 0), U);
    const int Lp = Sp / 2;
    constexpr int EP = CTC_MAX_U / 256;
    float ev[EP];
    int ec[EP];
#pragma unroll
    for (int i = 0; i < EP; ++i) {
        const int u = threadIdx.x + 256 * i;
        ev[i] = 0.f;
        ec[i] = blank;
        if (u < L) {
            int c = labels[(long)b * U + u];
            if (c < 0 || c >= V) c = blank;
            ec[i] = c;
            if (!from_lds) ev[i] = to_f32<T>(lr[c]);
        }
    }
    float xb = from_lds ? 0.f : to_f32<T>(lr[blank]);
    if constexpr (sizeof(T) == 4) {
        if (from_lds) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int c = (i * 256 + threadIdx.x) * 4;
                if (c < ldl) *(float4*)(rowbuf + c) = x[i];
            }
        }
    }
    float m, sum, wfac = 0.f;
    int am = 0;
    if (in_regs) {
        if constexpr (sizeof(T) == 4) row_regs_stats<kGrad, NR>(x, V, m, sum, am, red, wfac);
    } else if (argmax) {
        row_max_sum_arg<T>(lr, V, m, sum, am, red);
    } else {
        row_max_sum<T>(lr, V, m, sum, red);
    }
    if (from_lds) {
        // (the barrier inside the statistics stands between the stores above and these reads)
        xb = rowbuf[blank];
#pragma unroll
        for (int i = 0; i < EP; ++i)
            if (threadIdx.x + 256 * i < L) ev[i] = rowbuf[ec[i]];
    }
    const float l = m + __logf(sum);
    if (threadIdx.x == 0) {
        lse[row] = l;
        nrm[row] = l - xb;
        if (argmax) argmax[row] = am;
    }
#pragma unroll
    for (int i = 0; i < EP; ++i) {
        const int u = threadIdx.x + 256 * i;
        if (u < Lp) {
            // 2^e over the whole double range: the fraction on the fp32 unit, the integer part as exponent
            const float e = fminf(fmaxf((ev[i] - xb) * CTC_LOG2E, -2000.f), 2000.f);
            const float ei = floorf(e);
            emis[(long)row * Lp + u] = u < L ? ldexp((double)__builtin_amdgcn_exp2f(e - ei), (int)ei) : 0.0;
        }
    }
    if constexpr (kGrad) {
        TG* gr = grad + (long)row * ldl;
        if (in_regs) {
            if constexpr (sizeof(T) == 4) {
                const float gfac = sum > 0.f ? gscale * wfac / sum : 0.f;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int c = (i * 256 + threadIdx.x) * 4;
                    // the registers hold exp(x - wave maximum) (masked columns the zero the padding wants)
                    if (c < ldl) ctc_store4<TG>(gr + c, gfac * x[i].x, gfac * x[i].y, gfac * x[i].z, gfac * x[i].w);
                }
            }
        } else {
            for (int c = threadIdx.x * 4; c < ldl; c += 1024) {
                CtcRowVec<T> v;
                v.load(lr + c);
                float g[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = (c + k < V) ? gscale * __expf(v.get(k) - l) : 0.f;
                ctc_store4<TG>(gr + c, g[0], g[1], g[2], g[3]);
            }
        }
    }
}

// TensorFlow's dense CTC starts its forward vector at log(1e-307) (cast to fp32) instead of log 0
// on every state but the start state, and ends its backward vector the same way on every state but
// the two final ones.  Where an alignment exists these leaks change nothing (1e-307 relative);
// where none exists they ARE the result: a finite loss of about 7e2 and a gradient that follows
// the leaked paths.
#define CTC_TF_LOG0 (-706.8936157226562)

__device__ __forceinline__ double lse_pair(double a, double b) {
    const double m = fmax(a, b);
    if (m == -(double)INFINITY) return m;
    return m + log(exp(a - m) + exp(b - m));
}

// The slow path of the lattice launch, for a sample the recursion found no path for (path sum zero:
// the labels do not fit the frames, or an emission fell outside what the probability-domain
// recursion holds) or whose fitting alignments are so much less likely than those through a wrong
// start or end state that TensorFlow's finite log-zeros show, on the four waves of the sample's workgroup: tf.nn.ctc_loss's dense
// recursion restated literally (reference call site athena/loss.py:38-45) -- every one of the U label
// states and U+1 blank states of the padded label row plus the silent start state, the leaky initial
// and final vectors, the loss read from the backward recursion -- in double precision on logarithms,
// states in LDS.  It keeps its forward lattice (log2 units, shifted by the row maximum) in the
// sample's part of the alpha buffer, writes the posterior list of all U labels (and, with ``grad``,
// the gradient columns that have an occupancy) from the backward loop and raises flags[b], which
// tells ctc_grad to scatter all U + 1 entries.
// reductions over the first four waves of a workgroup (the slow path runs on those, whatever the
// size of the workgroup it is called from); ``red``: 4 floats of LDS; every thread gets the result
__device__ __forceinline__ float slow_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float slow_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
struct CtcSlowLds {
    double st[2][64 * CTC_MAX_R];
    float red[32];
    float tot[1][CTC_MAX_U];          // occupancy by label position
    int16_t first[CTC_MAX_U];         // first position with the same class (-1: the blank class)
};
template <typename T, typename TG>
__device__ __forceinline__ void ctc_slow_path(CtcSlowLds& lds, int b, const T* __restrict__ logits, int ldl,
                                              const int32_t* __restrict__ labels,
                                              const int32_t* __restrict__ label_len,
                                              const int32_t* __restrict__ logit_len, int T_, int V, int U,
                                              int blank, int Sp, int Up, const float* __restrict__ lse,
                                              double* __restrict__ alpha, float* __restrict__ occ,
                                              const int32_t* __restrict__ first_g, int32_t* __restrict__ flags,
                                              float* __restrict__ loss, float gscale, TG* __restrict__ grad) {
    double (&st)[2][64 * CTC_MAX_R] = lds.st;
    float (&red)[32] = lds.red;
    float (&tot)[1][CTC_MAX_U] = lds.tot;
    int16_t (&first)[CTC_MAX_U] = lds.first;
    const int tid = threadIdx.x;
    int len = logit_len[b];
    if (len > T_) len = T_;
    const int L = min(max(label_len[b], 0), U);
    const int32_t* lab = labels + (long)b * U;
    if (grad) {
        for (int u = tid; u < U; u += 256) first[u] = (int16_t)first_g[(long)b * U + u];
        __syncthreads();
    }
    const int S2 = 2 * U + 1;
    const T* lb = logits + (long)b * T_ * ldl;
    constexpr int PER = (64 * CTC_MAX_R) / 256;
    int cls[PER];
    bool skip_in[PER], skip_out[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int s = tid + 256 * i;
        int c = blank;
        bool si = false, so = false;
        if (s < S2 && (s & 1)) {
            const int u = s >> 1;
            c = lab[u];
            if (u >= 1) si = c != lab[u - 1];
            if (u + 1 < U) so = lab[u + 1] != c;
        }
        cls[i] = (c >= 0 && c < V) ? c : blank;
        skip_in[i] = si;
        skip_out[i] = so;
    }
    auto emission = [&](int t, int i) -> double {
        return (double)to_f32<T>(lb[(long)t * ldl + cls[i]]) - (double)lse[(long)b * T_ + t];
    };
    float* ab = (float*)(alpha + (long)b * T_ * Sp);     // forward lattice in floats, row pitch Sp
    // forward: fwd[t+1][s] = lse over predecessors of fwd[t] + emission at t
    double start = 0.0;
#pragma unroll
    for (int i = 0; i < PER; ++i) st[0][tid + 256 * i] = CTC_TF_LOG0;
    __syncthreads();
    int cur = 0;
    for (int t = 0; t < len; ++t) {
        double nv[PER];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + 256 * i;
            nv[i] = -(double)INFINITY;
            if (s < S2) {
                double v = st[cur][s];
                if (s >= 1) v = lse_pair(v, st[cur][s - 1]);
                if (skip_in[i]) v = lse_pair(v, st[cur][s - 2]);
                if (s <= 1) v = lse_pair(v, start);          // start -> blank 0, start -> label 1
                nv[i] = v + emission(t, i);
                mx = fmaxf(mx, (float)nv[i]);
            }
        }
        mx = slow_max(mx, red);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + 256 * i;
            if (s < S2) {
                const double v = nv[i] - (double)mx;
                st[cur ^ 1][s] = v;
                ab[(long)t * Sp + s] = fmaxf((float)(v * (double)CTC_LOG2E), CTC_LOGZERO);
            }
        }
        start = -(double)INFINITY;                           // the start state emits nothing
        cur ^= 1;
        __syncthreads();
    }
    // backward: bwd[len] = final vector; bwd[t] = lse over successors of (bwd[t+1] + emission at t)
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int s = tid + 256 * i;
        const bool fin = s == 2 * L || (L >= 1 && s == 2 * L - 1);
        st[0][s] = fin ? 0.0 : CTC_TF_LOG0;
    }
    double bstart = L == 0 ? 0.0 : CTC_TF_LOG0;
    double cum = 0.0;
    cur = 0;
    __syncthreads();
    for (int t = len - 1; t >= 0; --t) {
        // beta[t] = bwd[t+1]: the posterior of row t is alpha[t] * beta[t], normalised
        {
            float pv[PER];
            float pm = -INFINITY;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int s = tid + 256 * i;
                pv[i] = -INFINITY;
                if (s < S2) {
                    pv[i] = ab[(long)t * Sp + s] + fmaxf((float)(st[cur][s] * (double)CTC_LOG2E), CTC_LOGZERO);
                    pm = fmaxf(pm, pv[i]);
                }
            }
            pm = slow_max(pm, red);
            float ps = 0.f;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                pv[i] = (tid + 256 * i < S2) ? __builtin_amdgcn_exp2f(pv[i] - pm) : 0.f;
                ps += pv[i];
            }
            ps = slow_sum(ps, red);
            const float inv = ps > 0.f ? 1.f / ps : 0.f;
            float* orow = occ + ((long)b * T_ + t) * Up;
            if (grad) {
                for (int u = tid; u < U; u += 256) tot[0][u] = 0.f;
                __syncthreads();
            }
            float bl = 0.f;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int s = tid + 256 * i;
                if (s < S2) {
                    if (s & 1) {
                        orow[1 + (s >> 1)] = pv[i] * inv;
                        if (grad) {
                            const int fu = first[s >> 1];
                            if (fu < 0) bl += pv[i] * inv;
                            else atomicAdd(&tot[0][fu], pv[i] * inv);
                        }
                    } else {
                        bl += pv[i] * inv;
                    }
                }
            }
            bl = slow_sum(bl, red);
            if (tid == 0) orow[0] = bl;
            if (grad) {
                const T* lr = lb + (long)t * ldl;
                TG* gr = grad + ((long)b * T_ + t) * ldl;
                const float l = lse[(long)b * T_ + t];
                for (int u = tid; u < U; u += 256) {
                    const int c = lab[u];
                    if (first[u] == u && c >= 0 && c < V)
                        ctc_store1<TG>(gr + c, gscale * (__expf(to_f32<T>(lr[c]) - l) - tot[0][u]));
                }
                if (tid == 0) ctc_store1<TG>(gr + blank, gscale * (__expf(to_f32<T>(lr[blank]) - l) - bl));
                __syncthreads();
            }
        }
        // w[s] = bwd[t+1][s] + emission at t, parked in the other buffer
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + 256 * i;
            if (s < S2) st[cur ^ 1][s] = st[cur][s] + emission(t, i);
        }
        __syncthreads();
        double nv[PER];
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + 256 * i;
            nv[i] = -(double)INFINITY;
            if (s < S2) {
                double v = st[cur ^ 1][s];
                if (s + 1 < S2) v = lse_pair(v, st[cur ^ 1][s + 1]);
                if (skip_out[i]) v = lse_pair(v, st[cur ^ 1][s + 2]);
                nv[i] = v;
                mx = fmaxf(mx, (float)v);
            }
        }
        double ns = st[cur ^ 1][0];                          // start -> blank 0
        if (S2 > 1) ns = lse_pair(ns, st[cur ^ 1][1]);       // start -> label 1
        mx = fmaxf(mx, (float)ns);
        mx = slow_max(mx, red);                             // (its barriers also end the reads of w)
        cum += (double)mx;
        bstart = ns - (double)mx;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + 256 * i;
            if (s < S2) st[cur][s] = nv[i] - (double)mx;
        }
        __syncthreads();
    }
    if (tid == 0) {
        loss[b] = (float)(-(bstart + cum));
        flags[b] = 1;
    }
}

// ---------------------------------------------------------------- ctc lattice
// Second CTC launch: one workgroup of two waves per sample.  Wave 0 runs the alpha recursion forward
// in time, wave 1 the beta recursion backward; each lane owns R consecutive states in registers and
// gets its neighbours by DPP.  The emissions come from the table ctc_row_kernel left, relative to the
// blank logit of their row, so a blank state costs no multiplication; the per-row constant
// lse - logit[blank] is summed back by ctc_post.
// One LDS word read or written by an instruction of its own, with the wait inside the statement:
// the counters two waves talk through.  (A volatile access would do, but the compiler follows each
// with a wait for every outstanding memory operation of the wave, emission prefetches included.)
__device__ __forceinline__ int lds_peek(const int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v)
                 : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const int*)p)
                 : "memory");
    return v;
}
__device__ __forceinline__ void lds_poke(int* p, int v) {
    asm volatile("ds_write_b32 %0, %1"
                 :
                 : "v"((uint32_t)(uintptr_t)(__attribute__((address_space(3))) int*)p), "v"(v)
                 : "memory");
}

// R consecutive doubles per lane, as the widest loads the count allows
template <int R>
__device__ __forceinline__ void load_states(const double* __restrict__ src, double (&e)[R]) {
    if constexpr (R % 2 == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 2) {
            const double2 v = *(const double2*)(src + r);
            e[r] = v.x;
            e[r + 1] = v.y;
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) e[r] = src[r];
    }
}

// The posterior of one logits row from its two lattice rows, by one wave: lane l holds the states
// l, l + 64, ... (a state and its lane have the same parity: even lanes hold blank states, odd lanes
// labels; states beyond the sample's last one hold zeros).  alpha * beta normalised (the power-of-two
// scales of the two factors are constants of the row and cancel) leaves as a short list --
// occ[row][0] = all blank states together, occ[row][1 + u] = the state of label u -- and, with
// ``grad``, the gradient columns of the row that have an occupancy are written again, complete.
// ``totw``: Lp floats of LDS that belong to the wave.
template <typename TG, int R>
__device__ __forceinline__ void ctc_post_row(long row, int lane, int Sb, int V, int blank, int ldl, int Up, bool fix,
                                             const double (&av)[R], const double (&bv)[R], const double (&pe)[R],
                                             const int (&cl)[R], const int (&fu)[R], float nr, float* totw,
                                             float* __restrict__ occ, float gscale, TG* __restrict__ grad) {
    int mh = 0;
    double w[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        w[i] = lane + 64 * i < Sb ? av[i] * bv[i] : 0.0;
        mh = max(mh, __double2hiint(w[i]));
    }
    const int k = scale_exponent(wave_imax_dpp(mh));
    float f[R];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        f[i] = (float)ldexp(w[i], -k);
        sm += f[i];
    }
    float* orow = occ + row * Up;
    // Labels that repeat (their occupancies are summed at the first of them) or that name the blank
    // class need the sums in LDS below.  Without any -- the usual sample -- the blank column gets what
    // the even lanes hold and every label state its own column: one reduction that keeps the two
    // parities of the lanes apart (lanes l ^ 2, then the row of 16 lanes rotated by 4 and by 8) gives
    // both the normaliser and the blank sum.
    bool plain = true;
    if (fix) {
        bool odd_one = false;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int s = lane + 64 * i;
            odd_one = odd_one || ((s & 1) && s < Sb && fu[i] != (s >> 1));
        }
        plain = !__any(odd_one);
    }
    if (plain) {
        float e = sm;
        e = dpp_fadd<0x4E>(e);      // quad_perm [2,3,0,1]
        e = dpp_fadd<0x124>(e);     // row_ror:4
        e = dpp_fadd<0x128>(e);     // row_ror:8
        float part[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            part[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 16 * (q >> 1) + (q & 1)));
        const float even = (part[0] + part[2]) + (part[4] + part[6]);
        const float odd = (part[1] + part[3]) + (part[5] + part[7]);
        const float tot = even + odd;
        const float inv = tot > 0.f ? __builtin_amdgcn_rcpf(tot) : 0.f;
        const float bl = even * inv;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int s = lane + 64 * i;
            f[i] *= inv;
            if ((s & 1) && s < Sb) orow[1 + (s >> 1)] = f[i];
        }
        if (lane == 0) orow[0] = bl;
        if (grad) {
            TG* gr = grad + row * ldl;
            const double pb = (double)__expf(-nr);
            if (fix) {
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int s = lane + 64 * i;
                    if ((s & 1) && s < Sb && cl[i] >= 0 && cl[i] < V)
                        ctc_store1<TG>(gr + cl[i], gscale * ((float)(pe[i] * pb) - f[i]));
                }
            }
            if (lane == 0) ctc_store1<TG>(gr + blank, gscale * ((float)pb - bl));
        }
        return;
    }
    sm = wave_fsum_dpp(sm);
    const float inv = sm > 0.f ? __builtin_amdgcn_rcpf(sm) : 0.f;
    float bl = 0.f;
    if (fix) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int s = lane + 64 * i;
            if ((s & 1) && s < Sb) totw[s >> 1] = 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int s = lane + 64 * i;
        f[i] *= inv;
        if (s < Sb) {
            if (s & 1) {
                orow[1 + (s >> 1)] = f[i];
                if (fix) {
                    if (fu[i] < 0) bl += f[i];
                    else atomicAdd(&totw[fu[i]], f[i]);
                }
            } else {
                bl += f[i];
            }
        }
    }
    bl = wave_fsum_dpp(bl);
    if (lane == 0) orow[0] = bl;
    if (grad) {
        // The row launch left gscale * softmax in the row.  softmax of a label's column = its emission
        // (relative to blank) times the softmax of the blank column, exp(-(lse - logit[blank])).
        // The sums in LDS were made by this wave: LDS operations are carried out in the order they
        // were issued, so the wave only has to wait for its own (a fence would also wait for the
        // stores above to be acknowledged).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        TG* gr = grad + row * ldl;
        const double pb = (double)__expf(-nr);
        if (fix) {
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int s = lane + 64 * i;
                if ((s & 1) && s < Sb && fu[i] == (s >> 1) && cl[i] >= 0 && cl[i] < V)
                    ctc_store1<TG>(gr + cl[i], gscale * ((float)(pe[i] * pb) - totw[s >> 1]));
            }
        }
        if (lane == 0) ctc_store1<TG>(gr + blank, gscale * ((float)pb - bl));
        asm volatile("" ::: "memory");
    }
}

// A single wave gets an issue slot every 2.5 ns or so, whatever the instruction (vector, scalar, wait:
// measured), and the recursion is one wave per direction, so a step costs what its instruction
// count says.  The loops below are written to keep that count down: no bounds test and no address
// arithmetic per step (the emission table is padded, so a prefetch past either end of a sample
// reads memory that is never used), no conversion, zero fill by the DPP itself, LDS slots at
// compile-time offsets, the scaling test two compares every fourth step, the hand-over counters
// touched once per group of steps, the stores left to other waves.
template <typename T, typename TG, int R>
__global__ __launch_bounds__(256) void ctc_lattice_kernel(const double* __restrict__ emis,
                                                          const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ label_len,
                                                          const int32_t* __restrict__ logit_len, int T_, int U,
                                                          double* __restrict__ alpha, double* __restrict__ beta,
                                                          double* __restrict__ ll, const T* __restrict__ logits,
                                                          int ldl, int V, int blank, int Up,
                                                          const float* __restrict__ lse, float* __restrict__ occ,
                                                          const int32_t* __restrict__ first_g,
                                                          int32_t* __restrict__ flags, float* __restrict__ loss,
                                                          float gscale, TG* __restrict__ grad) {
    static_assert(R % 2 == 0, "a lane holds whole (blank, label) pairs: even r is a blank state, odd r a label");
    constexpr int Sp = 64 * R;
    constexpr int HR = R / 2;                   // labels per lane = doubles per lane in an emission row
    constexpr int Lp = Sp / 2;
    constexpr bool kParity = true;
    constexpr bool kBetaShift = CTC_BETA_SHIFT(R);
    // emission rows fetched ahead of the recursion through a ring of registers; one trip of the
    // unrolled loop is one turn of that ring
    constexpr int PF = R <= 4 ? 16 : (R <= 12 ? 8 : 4);
    static_assert(PF <= CTC_EMIS_PAD, "prefetch runs past the padding");
    // The state vectors leave through LDS.  A wave that both loads and stores waits for its stores
    // whenever it waits for a load (one counter serves both), and a store takes a microsecond or
    // more to be acknowledged, so waves 2 and 3 do the storing: the recursion waves write each
    // vector into a ring of DS slots and publish a count every G vectors, the flush waves copy
    // published slots to memory and publish how far they got.  LDS operations of a CU are carried
    // out in the order they were issued, so a count written after the data is seen after the data.
    // Vector i (i >= 1) sits in slot (i - 1) % DS, vector 0 in slot DS - 1.
    constexpr int DS = R <= 2 ? 32 : (R <= 4 ? 16 : (R <= 8 ? 8 : 4));
    constexpr int G = DS / 2;
    static_assert(2 * DS * Sp * 8 <= 65536 && (PF % G == 0 || G % PF == 0), "LDS ring");
    static_assert(sizeof(CtcSlowLds) <= 2 * DS * Sp * 8, "the slow path borrows the ring");
    __shared__ __attribute__((aligned(16))) double ring_s[2][DS][Sp];
    __shared__ int prod[2], cons[2];
    __shared__ int no_path, beta_broke;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int dir = (threadIdx.x >> 6) & 1;
    const bool flusher = threadIdx.x >= 128;
    // label of each of the lane's label states, its predecessor and its successor
    int lab_c[HR], lab_p[HR], lab_n[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int u = lane * HR + i;
        lab_c[i] = lab_p[i] = lab_n[i] = 0;
        if (U > 0) {
            lab_c[i] = labels[(long)b * U + min(u, U - 1)];
            lab_p[i] = labels[(long)b * U + min(max(u - 1, 0), U - 1)];
            lab_n[i] = labels[(long)b * U + min(u + 1, U - 1)];
        }
    }
    const int L = min(max(label_len[b], 0), U);
    const int Sb = 2 * L + 1;
    int len = logit_len[b];
    if (len > T_) len = T_;
    if (len <= 0) {
        if (threadIdx.x == 0) ll[b] = 0.0;
        return;
    }
    if (threadIdx.x < 2) {
        prod[threadIdx.x] = 0;
        cons[threadIdx.x] = 0;
    }
    __syncthreads();
    if (flusher) {
        // vector number i of a direction belongs to time i (alpha) or len-1-i (beta)
        double* dst = (dir == 0 ? alpha : beta) + (long)b * T_ * Sp + lane * R;
        int done = 0;
        while (done < len) {
            int avail;
            while ((avail = lds_peek(&prod[dir])) <= done) __builtin_amdgcn_s_sleep(2);
            // published vectors come in groups of G (the last group may be short): all LDS reads of
            // a group are issued before the first store waits for them
            for (; done + G <= avail; done += G) {
                double v[G][R];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const double* src = &ring_s[dir][(done + g + DS - 1) % DS][lane * R];
#pragma unroll
                    for (int r = 0; r < R; ++r) v[g][r] = src[r];
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int t = dir == 0 ? done + g : len - 1 - (done + g);
                    store_states<R>(dst + (long)t * Sp, v[g]);
                }
            }
            for (; done < avail; ++done) {
                const int t = dir == 0 ? done : len - 1 - done;
                double v[R];
                const double* src = &ring_s[dir][(done + DS - 1) % DS][lane * R];
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = src[r];
                store_states<R>(dst + (long)t * Sp, v);
            }
            lds_poke(&cons[dir], done);
        }
    } else {
    __builtin_amdgcn_s_setprio(3);
    // the skip transitions of each state as factors 0 / 1 (the labels were requested together with the
    // lengths, through clamped indices, so that they cost no memory latency of their own)
    double skip_in[R], skip_out[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = lane * R + r;
        const int u = s >> 1;
        const bool si = (s & 1) && s < Sb && u >= 1 && lab_c[r >> 1] != lab_p[r >> 1];
        const bool so = (s & 1) && s + 2 < Sb && lab_n[r >> 1] != lab_c[r >> 1];
        skip_in[r] = si ? 1.0 : 0.0;
        skip_out[r] = so ? 1.0 : 0.0;
    }
    double a[R];
    long cum = 0;
    // Scaling.  Every fourth step two compares tell whether the largest state of the wave has left
    // [2^-256, 2^256]; only then is its exponent found (a reduction over the wave) and taken out of
    // every state.  An emission within 2^(+-130) per step cannot carry a state from that band to the
    // end of the double range in four steps; one beyond that leaves a zero path sum or a state that
    // is not finite, and the sample goes to the slow path.
    auto rescale = [&]() {
        int mh = __double2hiint(a[0]);
#pragma unroll
        for (int r = 1; r < R; ++r) mh = max(mh, __double2hiint(a[r]));
        const bool big = __ballot(mh > ((1023 + 256) << 20)) != 0;
        const bool small = __ballot(mh >= ((1023 - 256) << 20)) == 0;
        if (big || small) {
            const int k = scale_exponent(wave_imax_dpp(mh));
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = ldexp(a[r], -k);
            cum += k;
        }
    };
    // An infinity or a NaN in a state stays one to the end of the recursion (every state is a term of
    // its own successor), so one look at the last vector tells whether any state ever left the
    // double range -- also one that no fitting alignment passes through and the path sum does not
    // see, whose product with a zero of the other lattice would be a NaN in the posterior.
    auto states_finite = [&]() {
        bool bad = false;
#pragma unroll
        for (int r = 0; r < R; ++r) bad |= !(a[r] < (double)INFINITY);
        return __ballot(bad) == 0;
    };
    // TensorFlow's recursion starts and ends on finite log-zeros (CTC_TF_LOG0 below) where this one has
    // true zeros.  Its result shows that only when alignments that start or end in a wrong state are
    // about e^707 = 2^1020 times likelier than the fitting ones: the last alpha vector (for the end)
    // and the last beta vector times the first frame's emissions (for the start) tell.  From the
    // largest high word of all states and of the legal ones: true when the two are more than 2^960
    // apart -- from there on the leak may reach the fp32 result, and the sample is the slow path's,
    // which restates TensorFlow's recursion literally.
    auto leak_shows = [&](int hi_all, int hi_legal) {
        const int e_all = (wave_imax_dpp(hi_all) >> 20) & 0x7ff;
        const int e_legal = (wave_imax_dpp(hi_legal) >> 20) & 0x7ff;
        return e_all - e_legal > 960;
    };
    // vector i = 1 + k * PF + j of trip k goes into slot (k * PF + j) % DS: ``base`` is the trip's
    // part of that, j the compile-time part
    double* myring = &ring_s[dir][0][lane * R];
    auto put = [&](int slot, const double (&v)[R]) {
        double* dstp = myring + slot * Sp;
#pragma unroll
        for (int r = 0; r < R; ++r) dstp[r] = v[r];
    };
    // before vector i (i - 1 a multiple of G) is written, the G slots from its own on must be empty
    auto wait_room = [&](int i) {
        while (i + G - lds_peek(&cons[dir]) > DS) __builtin_amdgcn_s_sleep(1);
    };
    // emission rows are read through a pointer that moves one row per step
    const double* ep;
    double ring[PF][HR];
    auto trips = [&](auto step, int row_stride) {
        // ring slot j of trip k holds the emissions for vector 1 + k * PF + j
#pragma unroll
        for (int j = 0; j < PF; ++j) load_states<HR>(ep + (long)j * row_stride, ring[j]);
        ep += (long)PF * row_stride;
        int i = 1, base = 0;
        for (; i + PF <= len; i += PF) {
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                double p[HR];
#pragma unroll
                for (int r = 0; r < HR; ++r) p[r] = ring[j][r];
                load_states<HR>(ep + (long)j * row_stride, ring[j]);
                if (j % G == 0) wait_room(i + j);
                step(p, (1 + j) & 3);
                put(DS >= PF ? base + j : j % DS, a);
                if (j % G == G - 1) lds_poke(&prod[dir], i + j + 1);
            }
            ep += (long)PF * row_stride;
            base = (base + PF) & (DS - 1);
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            if (i + j < len) {
                if (j % G == 0) wait_room(i + j);
                step(ring[j], (1 + j) & 3);
                put(DS >= PF ? base + j : j % DS, a);
                if (j % G == G - 1) lds_poke(&prod[dir], i + j + 1);
            }
        }
        lds_poke(&prod[dir], len);
    };
    const double* eb = emis + (long)b * T_ * Lp + lane * HR;
    if (dir == 0) {
        // ---------------- alpha: vector i is time i; step i reads the emissions of time i
        auto step = [&](const double (&p)[HR], int ph) {
            // neighbours: state s-1 and s-2; across the lane boundary they live in lane-1.  The
            // even r are blank states: factor 1, or a state beyond the sample's last one, which
            // only zeros ever reach
            const double p1 = lane_prev_f64(a[R - 1]);
            const double p2 = 0.0;
            double na[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double x1 = r >= 1 ? a[r >= 1 ? r - 1 : 0] : p1;
                if (kParity && (r % 2) == 0) {
                    na[r] = a[r] + x1;
                } else {
                    const double x2 = r >= 2 ? a[r >= 2 ? r - 2 : 0] : (r == 1 ? p1 : p2);
                    na[r] = fma(x2, skip_in[r], a[r] + x1) * p[r >> 1];
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = na[r];
            if (ph == 0) rescale();
        };
        {
            double p0[HR];
            load_states<HR>(eb, p0);
#pragma unroll
            for (int r = 0; r < R; ++r) a[r] = 0.0;
            if (lane == 0) {
                a[0] = 1.0;
                a[1] = p0[0];
            }
            put(DS - 1, a);
            if (len == 1) lds_poke(&prod[dir], 1);
        }
        ep = eb + Lp;
        trips(step, Lp);
        // path sum from the two final states at t = len-1
        double fin = 0.0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            if (s == Sb - 1 || (s == Sb - 2 && Sb >= 2)) fin += a[r];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) fin += __shfl_xor(fin, o, 64);
        int hi_all = 0, hi_end = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            hi_all = max(hi_all, __double2hiint(a[r]));
            if (s == Sb - 1 || (s == Sb - 2 && Sb >= 2)) hi_end = max(hi_end, __double2hiint(a[r]));
        }
        const bool whole = states_finite() && !leak_shows(hi_all, hi_end);
        if (lane == 0) {
            const bool ok = fin > 0.0 && fin < (double)INFINITY && whole;
            ll[b] = ok ? (double)cum + log2(fin) : -(double)INFINITY;
            no_path = ok ? 0 : 1;
        }
    } else {
        bool start_leak;
        // ---------------- beta: vector i is time len-1-i.  beta[len-1][s] = 1 for final states;
        // beta[t][s] = sum over successors s' in {s, s+1, s+2(skip)} of beta[t+1][s'] * p[t+1][s'],
        // so step i reads the emissions of time len-i
        if constexpr (kBetaShift) {
            // R = 2: the lane holds (label 2l-1, blank 2l) instead of (blank 2l, label 2l+1), the mirror
            // image of what the alpha recursion has: both successors of a state that are not in its own
            // lane are then the label state of the next lane -- one neighbour exchange per step, not two.
            // Vector entry j of a stored beta row therefore belongs to state j - 1 (ctc_post knows).
            // Lane 0's label slot is the state "-1": nothing reads it, and its emission is the last
            // entry of the table row in front (always 0 with at most 63 labels in 64 entries; in front
            // of the first row the padding entry that ctc_row_kernel zeroes).
            const double skip_l = (lane >= 1 && 2 * lane + 1 < Sb && lab_c[0] != lab_p[0]) ? 1.0 : 0.0;
            auto step = [&](const double (&p)[HR], int ph) {
                const double w0 = a[0] * p[0];
                const double n0 = lane_next_f64(w0);
                a[0] = fma(n0, skip_l, w0 + a[1]);
                a[1] = a[1] + n0;
                if (ph == 0) rescale();
            };
            // final states: blank 2L = slot 1 of lane L, label 2L-1 = slot 0 of lane L
            a[0] = (lane == L && L >= 1) ? 1.0 : 0.0;
            a[1] = lane == L ? 1.0 : 0.0;
            put(DS - 1, a);
            if (len == 1) lds_poke(&prod[dir], 1);
            ep = eb - 1 + (long)(len - 1) * Lp;
            trips(step, -Lp);
            // the start states 0 and 1: slot 1 of lane 0 and slot 0 of lane 1
            double p0[HR];
            load_states<HR>(eb - 1, p0);
            const int h0 = __double2hiint(a[0] * p0[0]), h1 = __double2hiint(a[1]);
            start_leak = leak_shows(max(h0, h1), max(lane == 1 ? h0 : 0, lane == 0 ? h1 : 0));
        } else {
        auto step = [&](const double (&p)[HR], int ph) {
            double w[R];
#pragma unroll
            for (int r = 0; r < R; ++r) w[r] = (r % 2) == 0 ? a[r] : a[r] * p[r >> 1];
            const double n1 = lane_next_f64(w[0]);
            const double n2 = (R >= 2) ? lane_next_f64(w[R >= 2 ? 1 : 0]) : lane_next_f64(n1);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double x1 = (r + 1 < R) ? w[r + 1 < R ? r + 1 : 0] : n1;
                if (kParity && (r % 2) == 0) {
                    a[r] = w[r] + x1;
                } else {
                    const double x2 = (r + 2 < R) ? w[r + 2 < R ? r + 2 : 0] : (r + 2 == R ? n1 : n2);
                    a[r] = fma(x2, skip_out[r], w[r] + x1);
                }
            }
            if (ph == 0) rescale();
        };
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            const bool fin = (s == Sb - 1) || (s == Sb - 2 && Sb >= 2);
            a[r] = fin ? 1.0 : 0.0;
        }
        put(DS - 1, a);
        if (len == 1) lds_poke(&prod[dir], 1);
        ep = eb + (long)(len - 1) * Lp;
        trips(step, -Lp);
        double p0[HR];
        load_states<HR>(eb, p0);
        int hi_all = 0, hi_start = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int h = __double2hiint((r % 2) == 0 ? a[r] : a[r] * p0[r >> 1]);
            hi_all = max(hi_all, h);
            if (lane * R + r < 2) hi_start = max(hi_start, h);
        }
        start_leak = leak_shows(hi_all, hi_start);
        }
        const bool whole = states_finite() && !start_leak;
        if (lane == 0) beta_broke = whole ? 0 : 1;
    }
    __builtin_amdgcn_s_setprio(0);
    }
    // all four waves meet here; the ring is no longer in use
    __syncthreads();
    if (beta_broke && !no_path) {
        __syncthreads();
        if (threadIdx.x == 0) {
            ll[b] = -(double)INFINITY;          // (thread 0 wrote the path sum there itself)
            no_path = 1;
        }
        __syncthreads();
    }
    if (no_path)
        ctc_slow_path<T, TG>(*reinterpret_cast<CtcSlowLds*>(&ring_s[0][0][0]), b, logits, ldl, labels, label_len,
                             logit_len, T_, V, U, blank, Sp, Up, lse, alpha, occ, first_g, flags, loss, gscale, grad);
}

// Third CTC launch.  CTC_POST_ROWS logits rows of one sample per workgroup, one wave per row at a
// time: the posterior over the lattice states of the row, alpha * beta normalised (the power-of-two
// scales of the two factors are constants of the row and cancel), leaves as a short list --
// occ[row][0] = all blank states together, occ[row][1 + u] = the state of label u -- so that
// ctc_grad neither reads the lattices nor reduces anything.  Wave 0 of a sample's first workgroup
// also writes loss[b] = -(ln of the blank-relative path sum - sum over t of (lse - logit[blank])).
//
// With ``grad`` the wave also writes the gradient columns of its row that have an occupancy.  Samples
// the lattice launch found no path for are its slow path's business and skipped here.
// (16 rows = 16 waves where the kernel fits 128 registers per lane, 8 otherwise)
#define CTC_POST_ROWS(R) ((R) <= 4 ? 16 : 8)
template <typename TG, int R>
__global__ __launch_bounds__(64 * CTC_POST_ROWS(R)) void ctc_post_kernel(const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ logit_len, int T_, int V,
                                                       int U, int blank, int ldl, int Up, int chunks,
                                                       const float* __restrict__ nrm,
                                                       const double* __restrict__ emis,
                                                       const double* __restrict__ alpha,
                                                       const double* __restrict__ beta, float* __restrict__ occ,
                                                       const int32_t* __restrict__ first_g,
                                                       const double* __restrict__ ll,
                                                       int32_t* __restrict__ flags, float* __restrict__ loss,
                                                       float gscale, TG* __restrict__ grad) {
    constexpr int Sp = 64 * R, Lp = Sp / 2;
    __shared__ float tot[CTC_POST_ROWS(R)][Lp];     // occupancy by label position, one set per wave
    const int b = blockIdx.x / chunks, chunk = blockIdx.x - b * chunks;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int t = chunk * CTC_POST_ROWS(R) + wv;
    const long row = (long)b * T_ + min(t, T_ - 1);
    // Everything the row needs is requested at once, before anything is tested (a load that waited
    // for the lengths, and the lengths for each other, would cost a memory latency each): lane l
    // holds the states l, l + 64, ...; a state and its lane have the same parity: even lanes hold
    // blank states, odd lanes labels.  States beyond the sample's last one hold zeros.
    int len = logit_len[b];
    const int Lraw = label_len[b];
    const double llb = ll[b];
    const double* ar = alpha + row * Sp;
    const double* br = beta + row * Sp;
    double av[R], bv[R], pe[R];
    int cl[R], fu[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        av[i] = ar[lane + 64 * i];
        bv[i] = br[lane + 64 * i + (CTC_BETA_SHIFT(R) ? 1 : 0)];
    }
    if (grad != nullptr && U > 0) {
        const int32_t* lab = labels + (long)b * U;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int u = (lane + 64 * i) >> 1;
            cl[i] = lab[min(u, U - 1)];
            fu[i] = first_g[(long)b * U + min(u, U - 1)];
            pe[i] = emis[row * Lp + u];
        }
    }
    const float nr = grad ? nrm[row] : 0.f;
    if (len > T_) len = T_;
    const int L = min(max(Lraw, 0), U);
    if (len <= 0) {
        // no frames: TensorFlow reads the loss off its final vector at the start state
        if (chunk == 0 && tid == 0) {
            loss[b] = L == 0 ? 0.f : (float)(-CTC_TF_LOG0);
            flags[b] = 0;
        }
        return;
    }
    if (llb == -(double)INFINITY) return;        // the slow path's sample
    if (chunk == 0 && wv == 0) {
        double acc = 0.0;
        for (int tt = lane; tt < len; tt += 64) acc += (double)nrm[(long)b * T_ + tt];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            loss[b] = (float)(acc - llb * CTC_LN2);
            flags[b] = 0;
        }
    }
    if (t >= len) return;
    const int Sb = 2 * L + 1;
    const bool fix = grad != nullptr && L > 0;   // (wave-uniform)
    ctc_post_row<TG, R>(row, lane, Sb, V, blank, ldl, Up, fix, av, bv, pe, cl, fu, nr, tot[wv], occ, gscale, grad);
}

// ---------------------------------------------------------------- ctc_grad
// One workgroup per logits row at a time.  The row's posterior list (at most U + 1 entries) and then
// the row itself are requested; the list is scattered by class into an LDS accumulator (labels that
// repeat add up there), and grad = scale * (softmax - occupancy) is streamed out, each thread
// clearing the accumulator entries it consumed.
#define CTC_OCC_CHUNK 8192
#define CTC_GRAD_VEC 8          // 4-wide column groups per thread and chunk: 256 * 8 * 4 = CTC_OCC_CHUNK
#define CTC_LIST_PER ((64 * CTC_MAX_R / 2 + 255) / 256)      // list entries per thread: U + 1 <= 512
template <typename TL, typename TG>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const TL* __restrict__ logits, int ldl,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ logit_len, int T_, int V, int U,
                                                       int blank, int Up, const float* __restrict__ lse,
                                                       const float* __restrict__ occ,
                                                       const int32_t* __restrict__ flags,
                                                       const float* __restrict__ grad_scale, float host_scale,
                                                       TG* __restrict__ grad, int rows_total, int chunk) {
    extern __shared__ float acc[];       // [chunk]
    for (int c = threadIdx.x; c < chunk; c += blockDim.x) acc[c] = 0.f;
    __syncthreads();
    const float scale = host_scale * (grad_scale ? grad_scale[0] : 1.f);
    // columns [0, ldl) are written: logits columns get the gradient, the padding zeros.  ldl % 4 == 0.
    for (int row = blockIdx.x; row < rows_total; row += gridDim.x) {
        const int b = row / T_, t = row - b * T_;
        TG* gr = grad + (long)row * ldl;
        if (t >= logit_len[b]) {
            for (int c = threadIdx.x * 4; c < ldl; c += blockDim.x * 4) ctc_store4<TG>(gr + c, 0.f, 0.f, 0.f, 0.f);
            continue;
        }
        // a sample without a fitting alignment carries TensorFlow's lattice over the whole padded label row
        const int nst = flags[b] ? U : min(max(label_len[b], 0), U);
        const float* orow = occ + (long)row * Up;
        const int32_t* lab = labels + (long)b * U;
        const TL* lr = logits + (long)row * ldl;
        const float l = lse[row];
        for (int v0 = 0; v0 < ldl; v0 += chunk) {
            // the list first (loads come back in the order they were asked for), then the row
            // (this chunk of it) into registers
            int cj[CTC_LIST_PER];
            float oj[CTC_LIST_PER];
#pragma unroll
            for (int q = 0; q < CTC_LIST_PER; ++q) {
                const int j = threadIdx.x + 256 * q;
                cj[q] = -1;
                oj[q] = 0.f;
                if (j <= nst) {
                    cj[q] = j == 0 ? blank : lab[j - 1];
                    oj[q] = orow[j];
                }
            }
            CtcRowVec<TL> x[CTC_GRAD_VEC];
#pragma unroll
            for (int i = 0; i < CTC_GRAD_VEC; ++i) {
                const int c = v0 + (i * 256 + threadIdx.x) * 4;
                if (i * 1024 < chunk && c < ldl) x[i].load(lr + c);
            }
#pragma unroll
            for (int q = 0; q < CTC_LIST_PER; ++q)
                if (cj[q] >= v0 && cj[q] < v0 + chunk && cj[q] < V) atomicAdd(&acc[cj[q] - v0], oj[q]);
            __syncthreads();
#pragma unroll
            for (int i = 0; i < CTC_GRAD_VEC; ++i) {
                const int cl = (i * 256 + threadIdx.x) * 4;     // column within the chunk
                const int c = v0 + cl;
                if (i * 1024 < chunk && c < ldl) {
                    const float4 o = *(const float4*)(acc + cl);
                    if (o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) *(float4*)(acc + cl) = make_float4(0.f, 0.f, 0.f, 0.f);
                    float g[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float ok = k == 0 ? o.x : k == 1 ? o.y : k == 2 ? o.z : o.w;
                        g[k] = (c + k < V) ? scale * (__expf(x[i].get(k) - l) - ok) : 0.f;
                    }
                    ctc_store4<TG>(gr + c, g[0], g[1], g[2], g[3]);
                }
            }
            __syncthreads();
        }
    }
}

static int ctc_check(const char* name, const void* logits, int dtype, int ldl, int B, int T, int V, int U, int blank,
                     void* workspace, size_t workspace_bytes) {
    ATHENA_REQUIRE(logits && workspace, ATHENA_E_SHAPE, "%s: null operand", name);
    ATHENA_REQUIRE(B > 0 && T > 0 && V > 1 && U >= 0 && ldl >= V, ATHENA_E_SHAPE, "%s: bad shape B=%d T=%d V=%d U=%d ldl=%d", name, B, T, V, U, ldl);
    ATHENA_REQUIRE(blank >= 0 && blank < V, ATHENA_E_SHAPE, "%s: blank %d outside [0,%d)", name, blank, V);
    ATHENA_REQUIRE(2 * U + 1 <= 64 * CTC_MAX_R, ATHENA_E_UNSUPPORTED, "%s: label length %d too long (2U+1 <= %d)", name, U, 64 * CTC_MAX_R);
    ATHENA_REQUIRE(dtype == ATHENA_F32 || dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "%s: dtype", name);
    ATHENA_REQUIRE(ldl % 4 == 0 && athena_aligned16(logits), ATHENA_E_ALIGN, "%s: ldl %% 4 and 16-byte alignment required", name);
    ATHENA_REQUIRE(workspace_bytes >= athena_ctc_workspace_bytes(B, T, U) && athena_aligned16(workspace), ATHENA_E_SHAPE, "%s: workspace too small or misaligned", name);
    return ATHENA_OK;
}

// the three launches of the forward pass; with ``grad`` they also leave the gradient
// gscale * (softmax - occupancy) (rows beyond logit_length and padding columns zero)
template <typename T, typename TG>
static void launch_ctc_forward(const T* logits, int ldl, const int32_t* labels, const int32_t* label_len,
                               const int32_t* logit_len, int B, int Tn, int V, int U, int blank, const CtcWorkspace& ws,
                               float* loss, int32_t* argmax, float gscale, TG* grad, hipStream_t s) {
    // (label logits from an LDS copy of the row: fp32 rows that fit the registers, from 64 labels on)
    static const int lg_env = getenv("ATHENA_CTC_LDS_GATHER") ? atoi(getenv("ATHENA_CTC_LDS_GATHER")) : -1;
    const int lds_gather = (sizeof(T) == 4 && ldl <= CTC_ROW_REGS * 1024 && (lg_env < 0 ? U >= 64 : lg_env != 0)) ? 1 : 0;
    const size_t row_lds = lds_gather ? sizeof(float) * (size_t)ldl : 0;
#define CTC_ROW(G, NR)                                                                                         \
    ATHENA_LAUNCH((ctc_row_kernel<T, TG, G, NR>), dim3(B * Tn), dim3(256), row_lds, s, logits, ldl, labels, label_len, \
                       logit_len, Tn, V, U, blank, ws.Sp, ws.lse, ws.nrm, ws.emis, argmax, ws.first, gscale, grad, lds_gather)
    if (grad) {
        if (ldl <= 5 * 1024) CTC_ROW(true, 5);
        else CTC_ROW(true, CTC_ROW_REGS);
    } else {
        if (ldl <= 5 * 1024) CTC_ROW(false, 5);
        else CTC_ROW(false, CTC_ROW_REGS);
    }
#undef CTC_ROW
#define CTC_LAT_POST(N)                                                                                          \
    ATHENA_LAUNCH((ctc_lattice_kernel<T, TG, N>), dim3(B), dim3(256), 0, s, ws.emis, labels, label_len,     \
                       logit_len, Tn, U, ws.alpha, ws.beta, ws.ll, logits, ldl, V, blank, ws.Up, ws.lse, ws.occ, \
                       ws.first, ws.flags, loss, gscale, grad);                                                  \
    ATHENA_LAUNCH((ctc_post_kernel<TG, N>), dim3(B * ((Tn + CTC_POST_ROWS(N) - 1) / CTC_POST_ROWS(N))), dim3(64 * CTC_POST_ROWS(N)), 0, s, labels, label_len, logit_len, \
                       Tn, V, U, blank, ldl, ws.Up, (Tn + CTC_POST_ROWS(N) - 1) / CTC_POST_ROWS(N), ws.nrm, ws.emis, ws.alpha, ws.beta, ws.occ, ws.first, \
                       ws.ll, ws.flags, loss, gscale, grad)
    switch (ctc_launch_r(2 * U + 1)) {
        case 2: CTC_LAT_POST(2); break;
        case 4: CTC_LAT_POST(4); break;
        case 6: CTC_LAT_POST(6); break;
        case 8: CTC_LAT_POST(8); break;
        case 12: CTC_LAT_POST(12); break;
        default: CTC_LAT_POST(CTC_MAX_R); break;
    }
#undef CTC_LAT_POST
}

extern "C" int athena_ctc_loss_fwd(const void* logits, int logits_dtype, int ldl, const int32_t* labels,
                                   const int32_t* label_len, const int32_t* logit_len, int B, int T, int V, int U,
                                   int blank, float* loss, int32_t* argmax, void* workspace, size_t workspace_bytes,
                                   athena_stream_t stream) {
    int rc = ctc_check("athena_ctc_loss_fwd", logits, logits_dtype, ldl, B, T, V, U, blank, workspace, workspace_bytes);
    if (rc) return rc;
    ATHENA_REQUIRE(labels && label_len && logit_len && loss, ATHENA_E_SHAPE, "athena_ctc_loss_fwd: null operand");
    CtcWorkspace ws;
    ctc_ws_layout(B, T, U, workspace, &ws);
    hipStream_t s = (hipStream_t)stream;
    if (logits_dtype == ATHENA_BF16)
        launch_ctc_forward<bf16_t, float>((const bf16_t*)logits, ldl, labels, label_len, logit_len, B, T, V, U, blank, ws,
                                          loss, argmax, 0.f, nullptr, s);
    else
        launch_ctc_forward<float, float>((const float*)logits, ldl, labels, label_len, logit_len, B, T, V, U, blank, ws,
                                         loss, argmax, 0.f, nullptr, s);
    ATHENA_CHECK_LAUNCH("athena_ctc_loss_fwd");
    return ATHENA_OK;
}

extern "C" int athena_ctc_loss_fwd_bwd(const void* logits, int logits_dtype, int ldl, const int32_t* labels,
                                       const int32_t* label_len, const int32_t* logit_len, int B, int T, int V, int U,
                                       int blank, float* loss, int32_t* argmax, float grad_scale, void* grad,
                                       int grad_dtype, void* workspace, size_t workspace_bytes,
                                       athena_stream_t stream) {
    int rc = ctc_check("athena_ctc_loss_fwd_bwd", logits, logits_dtype, ldl, B, T, V, U, blank, workspace, workspace_bytes);
    if (rc) return rc;
    ATHENA_REQUIRE(labels && label_len && logit_len && loss && grad, ATHENA_E_SHAPE, "athena_ctc_loss_fwd_bwd: null operand");
    ATHENA_REQUIRE(grad_dtype == ATHENA_F32 || grad_dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "athena_ctc_loss_fwd_bwd: grad dtype");
    ATHENA_REQUIRE(athena_aligned16(grad), ATHENA_E_ALIGN, "athena_ctc_loss_fwd_bwd: grad must be 16-byte aligned");
    CtcWorkspace ws;
    ctc_ws_layout(B, T, U, workspace, &ws);
    hipStream_t s = (hipStream_t)stream;
#define CTC_FB(TL, TG)                                                                                            \
    launch_ctc_forward<TL, TG>((const TL*)logits, ldl, labels, label_len, logit_len, B, T, V, U, blank, ws, loss, \
                               argmax, grad_scale, (TG*)grad, s)
    if (logits_dtype == ATHENA_F32 && grad_dtype == ATHENA_F32) CTC_FB(float, float);
    else if (logits_dtype == ATHENA_F32 && grad_dtype == ATHENA_BF16) CTC_FB(float, bf16_t);
    else if (logits_dtype == ATHENA_BF16 && grad_dtype == ATHENA_BF16) CTC_FB(bf16_t, bf16_t);
    else CTC_FB(bf16_t, float);
#undef CTC_FB
    ATHENA_CHECK_LAUNCH("athena_ctc_loss_fwd_bwd");
    return ATHENA_OK;
}

// x *= upstream[0] / expected unless the two are equal: every workgroup reads the two scalars and
// leaves at once in the expected case
template <typename TG>
__global__ __launch_bounds__(256) void rescale_unless_kernel(TG* __restrict__ x, long n4,
                                                             const float* __restrict__ upstream, float expected) {
    const float up = upstream[0];
    if (up == expected) return;
    const float f = up / expected;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        CtcRowVec<TG> v;
        v.load(x + 4 * i);
        ctc_store4<TG>(x + 4 * i, v.get(0) * f, v.get(1) * f, v.get(2) * f, v.get(3) * f);
    }
}

extern "C" int athena_rescale_unless(void* x, long n, int dtype, const float* upstream, float expected,
                                     athena_stream_t stream) {
    ATHENA_REQUIRE(x && upstream && n >= 0 && n % 4 == 0 && athena_aligned16(x), ATHENA_E_SHAPE,
                   "athena_rescale_unless: null operand, n %% 4 != 0 or misaligned");
    ATHENA_REQUIRE(dtype == ATHENA_F32 || dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "athena_rescale_unless: dtype");
    ATHENA_REQUIRE(expected != 0.f, ATHENA_E_SHAPE, "athena_rescale_unless: expected factor must not be zero");
    if (n == 0) return ATHENA_OK;
    const long n4 = n / 4;
    // few workgroups: in the expected case the launch does nothing, and how long nothing takes grows
    // with the grid
    const int grid = (int)((n4 + 255) / 256 < 128 ? (n4 + 255) / 256 : 128);
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(rescale_unless_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n4,
                           upstream, expected);
    else
        ATHENA_LAUNCH(rescale_unless_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float*)x, n4,
                           upstream, expected);
    ATHENA_CHECK_LAUNCH("athena_rescale_unless");
    return ATHENA_OK;
}

// ---------------------------------------------------------------- allocation-free forward
// athena_ctc_loss_fwd with its workspace in a device array of this library instead of a caller
// buffer, for callers that cannot allocate device memory (a stream under graph capture, a serving
// loop).  ``scratch`` names the array for the device compiler only: in host code that name is a
// stand-in, not a device address, so the address of each GPU's copy is asked of the runtime once
// per device and that pointer is what the kernels receive.
#define CTC_STATIC_SCRATCH_BYTES ((size_t)16 << 20)
#define CTC_STATIC_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[CTC_STATIC_SCRATCH_BYTES];
static void* g_scratch_ptr[CTC_STATIC_MAX_DEVICES];
static std::mutex g_scratch_mutex;

// address of the current device's copy of ``scratch`` (resolved at the first call on that device)
static int ctc_static_scratch(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    ATHENA_REQUIRE(e == hipSuccess && dev >= 0 && dev < CTC_STATIC_MAX_DEVICES, ATHENA_E_LAUNCH,
                   "athena_ctc_loss_fwd_static: hipGetDevice failed or device index %d out of range", dev);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        ATHENA_REQUIRE(e == hipSuccess && ptr, ATHENA_E_LAUNCH,
                       "athena_ctc_loss_fwd_static: hipGetSymbolAddress failed: %s", hipGetErrorString(e));
        g_scratch_ptr[dev] = ptr;
    }
    *out = g_scratch_ptr[dev];
    return ATHENA_OK;
}

extern "C" int athena_ctc_loss_fwd_static(const void* logits, int logits_dtype, int ldl, const int32_t* labels,
                                          const int32_t* label_len, const int32_t* logit_len, int B, int T, int V,
                                          int U, int blank, float* loss, int32_t* argmax, athena_stream_t stream) {
    ATHENA_REQUIRE(B > 0 && T > 0 && U >= 0, ATHENA_E_SHAPE, "athena_ctc_loss_fwd_static: bad shape B=%d T=%d U=%d", B, T, U);
    const size_t need = athena_ctc_workspace_bytes(B, T, U);
    ATHENA_REQUIRE(need <= CTC_STATIC_SCRATCH_BYTES, ATHENA_E_UNSUPPORTED,
                   "athena_ctc_loss_fwd_static: B=%d T=%d U=%d needs %zu bytes of scratch, the built-in buffer holds %zu; "
                   "use athena_ctc_loss_fwd with a caller workspace", B, T, U, need, (size_t)CTC_STATIC_SCRATCH_BYTES);
    void* ws = nullptr;
    int rc = ctc_static_scratch(&ws);
    if (rc) return rc;
    // every word of the workspace that the kernels read they have written earlier in the same
    // call, so nothing of a previous call has to be cleared
    return athena_ctc_loss_fwd(logits, logits_dtype, ldl, labels, label_len, logit_len, B, T, V, U, blank, loss, argmax,
                               ws, CTC_STATIC_SCRATCH_BYTES, stream);
}

extern "C" int athena_ctc_loss_bwd(const void* logits, int logits_dtype, int ldl, const int32_t* labels,
                                   const int32_t* label_len, const int32_t* logit_len, int B, int T, int V, int U,
                                   int blank, const float* grad_scale, float host_scale, void* grad, int grad_dtype,
                                   void* workspace, size_t workspace_bytes, athena_stream_t stream) {
    int rc = ctc_check("athena_ctc_loss_bwd", logits, logits_dtype, ldl, B, T, V, U, blank, workspace, workspace_bytes);
    if (rc) return rc;
    ATHENA_REQUIRE(labels && label_len && logit_len && grad, ATHENA_E_SHAPE, "athena_ctc_loss_bwd: null operand");
    ATHENA_REQUIRE(grad_dtype == ATHENA_F32 || grad_dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "athena_ctc_loss_bwd: grad dtype");
    ATHENA_REQUIRE(athena_aligned16(grad), ATHENA_E_ALIGN, "athena_ctc_loss_bwd: grad must be 16-byte aligned");
    CtcWorkspace ws;
    ctc_ws_layout(B, T, U, workspace, &ws);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * T;
    const int grid = rows < 2048 ? rows : 2048;
    // the occupancy accumulator covers the padded row in one piece when it fits
    int chunk = (ldl + 1023) / 1024 * 1024;
    if (chunk > CTC_OCC_CHUNK) chunk = CTC_OCC_CHUNK;
#define CTC_GRAD(TL, TG)                                                                                          \
    ATHENA_LAUNCH((ctc_grad_kernel<TL, TG>), dim3(grid), dim3(256), chunk * sizeof(float), s,                \
                       (const TL*)logits, ldl, labels, label_len, logit_len, T, V, U, blank, ws.Up, ws.lse,       \
                       ws.occ, ws.flags, grad_scale, host_scale, (TG*)grad, rows, chunk)
    if (logits_dtype == ATHENA_F32 && grad_dtype == ATHENA_F32) CTC_GRAD(float, float);
    else if (logits_dtype == ATHENA_F32 && grad_dtype == ATHENA_BF16) CTC_GRAD(float, bf16_t);
    else if (logits_dtype == ATHENA_BF16 && grad_dtype == ATHENA_BF16) CTC_GRAD(bf16_t, bf16_t);
    else CTC_GRAD(bf16_t, float);
#undef CTC_GRAD
    ATHENA_CHECK_LAUNCH("athena_ctc_loss_bwd");
    return ATHENA_OK;
}

// ================================================================ label-smoothed cross-entropy
// one workgroup per (b,l) row.  fwd: lse and the row loss; bwd: gradient.
template <typename T>
__global__ __launch_bounds__(256) void ce_fwd_kernel(const T* __restrict__ logits, int ldl,
                                                     const int32_t* __restrict__ targets, int V, float ls,
                                                     float* __restrict__ row_loss, float* __restrict__ lse) {
    __shared__ float red[32];
    const int row = blockIdx.x;
    const int tgt = targets[row];
    const T* lr = logits + (long)row * ldl;
    float m, s;
    row_max_sum<T>(lr, V, m, s, red);
    const float l = m + __logf(s);
    // sum_v log_softmax = sum_v x_v - V*l, needed only when smoothing
    float sx = 0.f;
    if (ls != 0.f) {
        for (int c = threadIdx.x; c < V; c += blockDim.x) sx += to_f32<T>(lr[c]);
        sx = block_sum(sx, red);
    }
    if (threadIdx.x == 0) {
        lse[row] = l;
        float loss = 0.f;
        if (tgt != 0) {
            // a target outside [0, V) has an all-zero one-hot row (tf.one_hot), it is never dereferenced
            const float lp_t = (tgt > 0 && tgt < V) ? to_f32<T>(lr[tgt]) - l : 0.f;
            loss = -(1.f - ls) * lp_t;
            if (ls != 0.f) loss -= (ls / V) * (sx - (float)V * l);
        }
        row_loss[row] = loss;
    }
}

template <typename TL, typename TG>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const TL* __restrict__ logits, int ldl,
                                                     const int32_t* __restrict__ targets, int V, float ls,
                                                     const float* __restrict__ lse,
                                                     const float* __restrict__ grad_scale, float host_scale,
                                                     TG* __restrict__ grad) {
    const int row = blockIdx.x;
    const int tgt = targets[row];
    TG* gr = grad + (long)row * ldl;
    if (tgt == 0) {
        for (int c = threadIdx.x; c < ldl; c += blockDim.x) gr[c] = from_f32<TG>(0.f);
        return;
    }
    const float scale = host_scale * (grad_scale ? grad_scale[0] : 1.f);
    const TL* lr = logits + (long)row * ldl;
    const float l = lse[row];
    const float uni = ls / V;
    // sum_v y_v = (1-ls) + V*ls/V = 1 (only ls for a target outside [0, V), whose one-hot row is empty)
    const float sum_y = (tgt > 0 && tgt < V) ? 1.f : ls;
    for (int c = threadIdx.x; c < V; c += blockDim.x) {
        const float p = __expf(to_f32<TL>(lr[c]) - l);
        const float y = uni + (c == tgt ? 1.f - ls : 0.f);
        gr[c] = from_f32<TG>(scale * (p * sum_y - y));
    }
    for (int c = V + threadIdx.x; c < ldl; c += blockDim.x) gr[c] = from_f32<TG>(0.f);
}

extern "C" int athena_ce_loss_fwd(const void* logits, int logits_dtype, int ldl, const int32_t* targets, int rows,
                                  int V, float label_smoothing, float* row_loss, float* lse, athena_stream_t stream) {
    ATHENA_REQUIRE(logits && targets && row_loss && lse, ATHENA_E_SHAPE, "athena_ce_loss_fwd: null operand");
    ATHENA_REQUIRE(rows > 0 && V > 1 && ldl >= V, ATHENA_E_SHAPE, "athena_ce_loss_fwd: bad shape");
    ATHENA_REQUIRE(ldl % 4 == 0 && athena_aligned16(logits), ATHENA_E_ALIGN, "athena_ce_loss_fwd: ldl %% 4 and 16-byte alignment required");
    hipStream_t s = (hipStream_t)stream;
    if (logits_dtype == ATHENA_BF16)
        ATHENA_LAUNCH(ce_fwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, ldl, targets, V, label_smoothing, row_loss, lse);
    else if (logits_dtype == ATHENA_F32)
        ATHENA_LAUNCH(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)logits, ldl, targets, V, label_smoothing, row_loss, lse);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_ce_loss_fwd: dtype");
    ATHENA_CHECK_LAUNCH("athena_ce_loss_fwd");
    return ATHENA_OK;
}

extern "C" int athena_ce_loss_bwd(const void* logits, int logits_dtype, int ldl, const int32_t* targets, int rows,
                                  int V, float label_smoothing, const float* lse, const float* grad_scale,
                                  float host_scale, void* grad, int grad_dtype, athena_stream_t stream) {
    ATHENA_REQUIRE(logits && targets && lse && grad, ATHENA_E_SHAPE, "athena_ce_loss_bwd: null operand");
    ATHENA_REQUIRE(rows > 0 && V > 1 && ldl >= V, ATHENA_E_SHAPE, "athena_ce_loss_bwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
#define CE_BWD(TL, TG) ATHENA_LAUNCH((ce_bwd_kernel<TL, TG>), dim3(rows), dim3(256), 0, s, (const TL*)logits, ldl, targets, V, label_smoothing, lse, grad_scale, host_scale, (TG*)grad)
    if (logits_dtype == ATHENA_F32 && grad_dtype == ATHENA_F32) CE_BWD(float, float);
    else if (logits_dtype == ATHENA_F32 && grad_dtype == ATHENA_BF16) CE_BWD(float, bf16_t);
    else if (logits_dtype == ATHENA_BF16 && grad_dtype == ATHENA_BF16) CE_BWD(bf16_t, bf16_t);
    else if (logits_dtype == ATHENA_BF16 && grad_dtype == ATHENA_F32) CE_BWD(bf16_t, float);
    else ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_ce_loss_bwd: dtype");
#undef CE_BWD
    ATHENA_CHECK_LAUNCH("athena_ce_loss_bwd");
    return ATHENA_OK;
}

// ================================================================ argmax over rows, scaled row sums
template <typename T>
__global__ __launch_bounds__(256) void argmax_rows_kernel(const T* __restrict__ logits, int ldl, int V,
                                                          int32_t* __restrict__ out) {
    __shared__ float sv[256];
    __shared__ int si[256];
    const int row = blockIdx.x;
    const T* lr = logits + (long)row * ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += blockDim.x) {
        float x = to_f32<T>(lr[c]);
        if (x > best) { best = x; bi = c; }
    }
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            float x = sv[threadIdx.x + o];
            int i = si[threadIdx.x + o];
            if (x > sv[threadIdx.x] || (x == sv[threadIdx.x] && i < si[threadIdx.x])) {
                sv[threadIdx.x] = x;
                si[threadIdx.x] = i;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row] = si[0] == 0x7fffffff ? 0 : si[0];
}

extern "C" int athena_argmax_rows(const void* logits, int dtype, int ldl, int rows, int V, int32_t* out,
                                  athena_stream_t stream) {
    ATHENA_REQUIRE(logits && out && rows > 0 && V > 0 && ldl >= V, ATHENA_E_SHAPE, "athena_argmax_rows: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == ATHENA_BF16)
        ATHENA_LAUNCH(argmax_rows_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, ldl, V, out);
    else if (dtype == ATHENA_F32)
        ATHENA_LAUNCH(argmax_rows_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)logits, ldl, V, out);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_argmax_rows: dtype");
    ATHENA_CHECK_LAUNCH("athena_argmax_rows");
    return ATHENA_OK;
}

// out[g] = scale * sum_{i < group} x[g*group + i]   (deterministic: one workgroup per output)
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ x, int group, float scale,
                                                        float* __restrict__ out) {
    __shared__ float red[32];
    const float* p = x + (long)blockIdx.x * group;
    float s = 0.f;
    for (int i = threadIdx.x; i < group; i += blockDim.x) s += p[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[blockIdx.x] = s * scale;
}

// ---------------------------------------------------------------- MPC loss (masked L1)
// row_loss[b, t] = sum_c |target - logit| over the elements whose (masked) input is exactly 0, for
// frames t < logit_len[b]; one workgroup per row.
__global__ __launch_bounds__(256) void mpc_loss_fwd_kernel(const float* __restrict__ logits, int ldl,
                                                          const float* __restrict__ target,
                                                          const float* __restrict__ masked,
                                                          const int32_t* __restrict__ logit_len, int Tp, int N,
                                                          float* __restrict__ row_loss) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const int b = (int)(row / Tp), t = (int)(row % Tp);
    float s = 0.f;
    if (t < logit_len[b]) {
        for (int c = threadIdx.x; c < N; c += 256)
            if (masked[row * N + c] == 0.f) s += fabsf(target[row * N + c] - logits[row * ldl + c]);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) row_loss[row] = s;
}

// grad = d(sum of row losses)/dlogits * gscale[0] * host_scale: -sign(target - logit) where the
// element counts, 0 elsewhere (also in the padding columns N .. ldl - 1)
template <typename TG>
__global__ __launch_bounds__(256) void mpc_loss_bwd_kernel(const float* __restrict__ logits, int ldl,
                                                          const float* __restrict__ target,
                                                          const float* __restrict__ masked,
                                                          const int32_t* __restrict__ logit_len, int Tp, int N,
                                                          const float* __restrict__ gscale, float host_scale,
                                                          TG* __restrict__ grad) {
    const long row = blockIdx.x;
    const int b = (int)(row / Tp), t = (int)(row % Tp);
    const bool live = t < logit_len[b];
    const float g = gscale[0] * host_scale;
    for (int c = threadIdx.x; c < ldl; c += 256) {
        float v = 0.f;
        if (live && c < N && masked[row * N + c] == 0.f) {
            const float d = target[row * N + c] - logits[row * ldl + c];
            v = d > 0.f ? -g : (d < 0.f ? g : 0.f);
        }
        grad[row * ldl + c] = from_f32<TG>(v);
    }
}

extern "C" int athena_mpc_loss_fwd(const float* logits, int ldl, const float* target, const float* masked_input,
                                   const int32_t* logit_len, int B, int Tp, int N, float* row_loss,
                                   athena_stream_t stream) {
    ATHENA_REQUIRE(logits && target && masked_input && logit_len && row_loss, ATHENA_E_SHAPE, "athena_mpc_loss_fwd: null operand");
    ATHENA_REQUIRE(B > 0 && Tp > 0 && N > 0 && ldl >= N && (long)B * Tp < (1L << 31), ATHENA_E_SHAPE,
                   "athena_mpc_loss_fwd: bad shape B=%d Tp=%d N=%d ldl=%d", B, Tp, N, ldl);
    ATHENA_LAUNCH(mpc_loss_fwd_kernel, dim3(B * Tp), dim3(256), 0, (hipStream_t)stream, logits, ldl, target,
                       masked_input, logit_len, Tp, N, row_loss);
    ATHENA_CHECK_LAUNCH("athena_mpc_loss_fwd");
    return ATHENA_OK;
}

extern "C" int athena_mpc_loss_bwd(const float* logits, int ldl, const float* target, const float* masked_input,
                                   const int32_t* logit_len, int B, int Tp, int N, const float* grad_scale,
                                   float host_scale, void* grad, int grad_dtype, athena_stream_t stream) {
    ATHENA_REQUIRE(logits && target && masked_input && logit_len && grad_scale && grad, ATHENA_E_SHAPE, "athena_mpc_loss_bwd: null operand");
    ATHENA_REQUIRE(B > 0 && Tp > 0 && N > 0 && ldl >= N && (long)B * Tp < (1L << 31), ATHENA_E_SHAPE,
                   "athena_mpc_loss_bwd: bad shape B=%d Tp=%d N=%d ldl=%d", B, Tp, N, ldl);
    if (grad_dtype == ATHENA_BF16)
        ATHENA_LAUNCH(mpc_loss_bwd_kernel<bf16_t>, dim3(B * Tp), dim3(256), 0, (hipStream_t)stream, logits, ldl, target,
                           masked_input, logit_len, Tp, N, grad_scale, host_scale, (bf16_t*)grad);
    else if (grad_dtype == ATHENA_F32)
        ATHENA_LAUNCH(mpc_loss_bwd_kernel<float>, dim3(B * Tp), dim3(256), 0, (hipStream_t)stream, logits, ldl, target,
                           masked_input, logit_len, Tp, N, grad_scale, host_scale, (float*)grad);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_mpc_loss_bwd: grad dtype");
    ATHENA_CHECK_LAUNCH("athena_mpc_loss_bwd");
    return ATHENA_OK;
}

extern "C" int athena_group_sum(const float* x, int ngroups, int group, float scale, float* out,
                                athena_stream_t stream) {
    ATHENA_REQUIRE(x && out && ngroups > 0 && group > 0, ATHENA_E_SHAPE, "athena_group_sum: bad arguments");
    ATHENA_LAUNCH(group_sum_kernel, dim3(ngroups), dim3(256), 0, (hipStream_t)stream, x, group, scale, out);
    ATHENA_CHECK_LAUNCH("athena_group_sum");
    return ATHENA_OK;
}

// out[0] = sum_i weights[i] * terms[i][0]: the total of a few scalar loss terms as one launch
#define WSUM_MAX 8
struct WeightedTerms {
    const float* term[WSUM_MAX];
    float weight[WSUM_MAX];
    int n;
};
__global__ void weighted_sum_kernel(WeightedTerms w, float* __restrict__ out) {
    float s = 0.f;
    for (int i = 0; i < w.n; ++i) s = fmaf(w.weight[i], w.term[i][0], s);
    out[0] = s;
}

extern "C" int athena_weighted_sum(const float* const* terms, const float* weights, int n, float* out,
                                   athena_stream_t stream) {
    ATHENA_REQUIRE(terms && weights && out && n > 0 && n <= WSUM_MAX, ATHENA_E_SHAPE,
                   "athena_weighted_sum: 1..%d terms expected, got %d", WSUM_MAX, n);
    WeightedTerms w = {};
    w.n = n;
    for (int i = 0; i < n; ++i) {
        ATHENA_REQUIRE(terms[i], ATHENA_E_SHAPE, "athena_weighted_sum: term %d is null", i);
        w.term[i] = terms[i];
        w.weight[i] = weights[i];
    }
    ATHENA_LAUNCH(weighted_sum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, w, out);
    ATHENA_CHECK_LAUNCH("athena_weighted_sum");
    return ATHENA_OK;
}
