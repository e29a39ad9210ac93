// CTC loss (log-softmax + alpha/beta over the blank-augmented lattice + gradient), the
// label-smoothed cross-entropy, and row argmax.  Bandwidth-bound kernels: each logits row is
// streamed with 16-byte loads and reduced with wavefront shuffles.
//
// CTC is three kernels:
//   ctc_fwd    : the lattice workgroups (one per sample: alpha and beta recursions in two
//                wavefronts, states in registers) and the row workgroups (logsumexp of one logits
//                row each) in one launch, so the serial recursion overlaps the streaming pass
//   ctc_finish : per-sample loss from the lattice path sum and the row normalisers
//   ctc_grad   : one workgroup per row: posterior = normalise(alpha+beta) scattered by class
//                into an LDS accumulator, then grad = scale * (softmax - occupancy) streamed out
#include "common.h"

#include <mutex>

#define CTC_LOGZERO (-1e30f)
#define CTC_MAX_R 16            // lattice states per lane: supports 2U+1 <= 1024
#define CTC_LOG2E 1.4426950408889634f
#define CTC_LN2 0.6931471805599453

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// states per lane the lattice kernel is instantiated for
static int ctc_launch_r(int S) {
    const int R = (S + 63) / 64;
    static const int sizes[] = {1, 2, 3, 4, 5, 7, 10, CTC_MAX_R};
    for (int v : sizes)
        if (R <= v) return v;
    return CTC_MAX_R;
}

struct CtcWorkspace {
    float* lse;      // [B*T]     natural-log normaliser of each row (softmax of the gradient)
    float* nrm;      // [B*T]     lse - logit[blank]: the normaliser relative to the blank logit
    float* alpha;    // [B*T*Sp]  log2 alpha up to a per-(b,t) constant (emission at t included)
    float* beta;     // [B*T*Sp]  log2 beta  up to a per-(b,t) constant (emission at t excluded)
    double* ll;      // [B]       log2 of the path sum over blank-relative emissions (-inf: no path)
    int32_t* flags;  // [B]       1: no alignment fits; alpha / beta then hold TensorFlow's leaky lattice
    int Sp;          // row pitch of alpha / beta: 64 * states per lane
};

static size_t ctc_ws_layout(int B, int T, int U, void* base, CtcWorkspace* ws) {
    const int Sp = 64 * ctc_launch_r(2 * U + 1);
    size_t off = 0;
    char* p = (char*)base;
    size_t n_row = align_up(sizeof(float) * (size_t)B * T, 256);
    size_t n_lat = align_up(sizeof(float) * (size_t)B * T * Sp, 256);
    if (ws) ws->lse = (float*)(p + off);
    off += n_row;
    if (ws) ws->nrm = (float*)(p + off);
    off += n_row;
    if (ws) ws->alpha = (float*)(p + off);
    off += n_lat;
    if (ws) ws->beta = (float*)(p + off);
    off += n_lat;
    if (ws) ws->ll = (double*)(p + off);
    off += align_up(sizeof(double) * (size_t)B, 256);
    if (ws) ws->flags = (int32_t*)(p + off);
    off += align_up(sizeof(int32_t) * (size_t)B, 256);
    if (ws) ws->Sp = Sp;
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

// fp32 rows of at most CTC_ROW_REGS * 1024 columns with ld % 4 == 0: the whole row is fetched into
// registers before anything depends on it (all loads of a thread in flight at once instead of one
// per trip of an online loop), then maximum, first arg-max and the sum of exponentials relative to
// the workgroup's maximum are taken from the registers.  Columns V .. ld-1 are padding and masked.
#define CTC_ROW_REGS 8
__device__ __forceinline__ void row_max_sum_arg_regs(const float* __restrict__ row, int V, int ld, float& m_out,
                                                     float& s_out, int& arg_out, float* red) {
    float4 x[CTC_ROW_REGS];
#pragma unroll
    for (int i = 0; i < CTC_ROW_REGS; ++i) {
        const int c = (i * 256 + threadIdx.x) * 4;
        x[i] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        if (c < ld) x[i] = *(const float4*)(row + c);
    }
    float m = -INFINITY;
    int idx = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < CTC_ROW_REGS; ++i) {
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
    const float gm = block_max(m, red);
    float s = 0.f;
    if (gm != -INFINITY) {
#pragma unroll
        for (int i = 0; i < CTC_ROW_REGS; ++i)
            s += __expf(x[i].x - gm) + __expf(x[i].y - gm) + __expf(x[i].z - gm) + __expf(x[i].w - gm);
    }
    s_out = block_sum(s, red);
    const float cand = (m == gm && idx != 0x7fffffff) ? -(float)idx : -INFINITY;
    const float best = block_max(cand, red);
    arg_out = best == -INFINITY ? 0 : (int)(-best);
    m_out = gm;
}

// ---------------------------------------------------------------- lattice arithmetic (log2 domain)
// log2(2^a + 2^b + 2^c) and log2(2^a + 2^b) on the raw exp2 / log2 units: the largest term
// contributes exactly 1, so one exponential fewer than terms
__device__ __forceinline__ float lse3_2(float a, float b, float c) {
    const float hi = fmaxf(a, fmaxf(b, c));
    const float lo = fminf(a, fminf(b, c));
    const float md = __builtin_amdgcn_fmed3f(a, b, c);
    return hi + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(md - hi) + __builtin_amdgcn_exp2f(lo - hi));
}
__device__ __forceinline__ float lse2_2(float a, float b) {
    return fmaxf(a, b) + __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-fabsf(a - b)));
}

// value held by the previous / next lane of the wave (``fill`` where there is none): a single DPP
// move (wave_shr:1 / wave_shl:1) instead of a trip through the LDS crossbar
__device__ __forceinline__ float lane_prev(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false)));
}
// maximum over the wave: four DPP steps inside each row of 16 lanes (quad swaps, half mirror,
// mirror), then the four row results through scalar registers
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = dpp_max<0xB1>(v);       // quad_perm [1,0,3,2]
    v = dpp_max<0x4E>(v);       // quad_perm [2,3,0,1]
    v = dpp_max<0x141>(v);      // row_half_mirror
    v = dpp_max<0x140>(v);      // row_mirror
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
    const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
    const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// R consecutive floats per lane, as the widest stores the count allows
template <int R>
__device__ __forceinline__ void store_states(float* __restrict__ dst, const float (&a)[R]) {
    if constexpr (R % 4 == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 4) *(float4*)(dst + r) = make_float4(a[r], a[r + 1], a[r + 2], a[r + 3]);
    } else if constexpr (R % 2 == 0) {
#pragma unroll
        for (int r = 0; r < R; r += 2) *(float2*)(dst + r) = make_float2(a[r], a[r + 1]);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) dst[r] = a[r];
    }
}

// ---------------------------------------------------------------- ctc forward
// One launch, two kinds of workgroup.
//   blockIdx <  B : the lattice of sample b.  Wave 0 runs the alpha recursion forward in time,
//                   wave 1 the beta recursion backward; each lane owns R consecutive states in
//                   registers and gets its neighbours by DPP.  Emissions are read straight from
//                   the logits relative to the blank logit of the row,
//                       e[t][s] = (logit[t][ext[s]] - logit[t][blank]) * log2(e),
//                   so the lattice does not wait for the row normalisers and its values stay
//                   small wherever blank or the label dominates; the per-row constant
//                   lse - logit[blank] is summed back by ctc_finish.  Every fourth step the state
//                   vector is shifted by its maximum (kept in a double) to hold fp32 resolution.
//   blockIdx >= B : one logits row each: logsumexp over V.
// The lattice workgroups are dispatched first and run for the length of the recursion while the
// rest of the chip streams the rows.
template <typename T, int R>
__global__ __launch_bounds__(256) void ctc_fwd_kernel(const T* __restrict__ logits, int ldl,
                                                      const int32_t* __restrict__ labels,
                                                      const int32_t* __restrict__ label_len,
                                                      const int32_t* __restrict__ logit_len, int B, int T_, int V,
                                                      int U, int blank, float* __restrict__ lse,
                                                      float* __restrict__ nrm, float* __restrict__ alpha,
                                                      float* __restrict__ beta, double* __restrict__ ll,
                                                      int32_t* __restrict__ argmax) {
    __shared__ float red[32];
    if ((int)blockIdx.x >= B) {
        const int row = blockIdx.x - B;
        const int b = row / T_, t = row - b * T_;
        if (t >= logit_len[b]) return;
        const T* lr = logits + (long)row * ldl;
        float m, s;
        int am = 0;
        bool in_regs = false;
        if constexpr (sizeof(T) == 4) in_regs = ldl <= CTC_ROW_REGS * 1024 && (ldl & 3) == 0;
        if (in_regs) row_max_sum_arg_regs((const float*)lr, V, ldl, m, s, am, red);     // uniform branches
        else if (argmax) row_max_sum_arg<T>(lr, V, m, s, am, red);
        else row_max_sum<T>(lr, V, m, s, red);
        if (threadIdx.x == 0) {
            const float ls = __logf(s);
            lse[row] = m + ls;
            nrm[row] = (m - to_f32<T>(lr[blank])) + ls;
            if (argmax) argmax[row] = am;
        }
        return;
    }
    if (threadIdx.x >= 128) return;
    // the recursion is one long dependency chain: let it issue ahead of the row workgroups that
    // share its SIMD
    __builtin_amdgcn_s_setprio(3);
    constexpr int Sp = 64 * R;
    constexpr bool kParity = (R % 2) == 0;      // state parity is then a property of r alone
    // emission rows fetched ahead of the recursion (a step is some 300 cycles, a load under the
    // row workgroups' traffic a few microseconds).  Measured on gfx950: 12 and 16 steps ahead run
    // the recursion at its arithmetic pace; 20 and more are twice as slow.
    constexpr int CTC_PF = R <= 2 ? 16 : 8;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int dir = threadIdx.x >> 6;
    const int L = min(max(label_len[b], 0), U);
    const int Sb = 2 * L + 1;
    int len = logit_len[b];
    if (len > T_) len = T_;
    if (len <= 0) {
        if (dir == 0 && lane == 0) ll[b] = 0.0;
        return;
    }
    const T* lb = logits + (long)b * T_ * ldl;
    // per-state static data: class, validity (as an additive penalty) and the skip transitions
    int cls[R];
    float pen[R];
    bool skip_in[R], skip_out[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = lane * R + r;
        int c = blank;
        bool si = false, so = false;
        if ((s & 1) && s < Sb) {
            const int u = s >> 1;
            c = labels[(long)b * U + u];
            if (u >= 1) si = c != labels[(long)b * U + u - 1];
            if (s + 2 < Sb) so = labels[(long)b * U + u + 1] != c;
        }
        cls[r] = (c >= 0 && c < V) ? c : blank;
        pen[r] = s < Sb ? 0.f : CTC_LOGZERO;
        skip_in[r] = si;
        skip_out[r] = so;
    }
    // emissions are fetched CTC_PF steps ahead through a ring of registers: a step is far shorter
    // than the latency of a load, so a one-step prefetch would leave every step waiting for memory
    T ring[CTC_PF][R];
    T ringb[CTC_PF];
    // the blank logit goes through a vector register like the others (a uniform address read as a
    // scalar would put the whole ring of them into scalar registers, which do not hold that many)
    int vblank = blank;
    asm volatile("" : "+v"(vblank));
    auto fetch = [&](int j, int t) {
        const T* rowp = lb + (long)t * ldl;
        ringb[j] = rowp[vblank];
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (!(kParity && (r % 2) == 0)) ring[j][r] = rowp[cls[r]];
    };
    auto emis = [&](int j, float (&e)[R]) {
        const float xb = to_f32<T>(ringb[j]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (kParity && (r % 2) == 0) e[r] = pen[r];      // a blank state: logit - logit[blank] = 0
            else e[r] = fmaf(to_f32<T>(ring[j][r]) - xb, CTC_LOG2E, pen[r]);
        }
    };
    float a[R];
    if (dir == 0) {
        // ---------------- alpha
        double cum = 0.0;
        float* ab = alpha + (long)b * T_ * Sp + lane * R;
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j) fetch(j, j < len ? j : len - 1);
        for (int tb = 0; tb < len; tb += CTC_PF) {
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) {
                const int t = tb + j;
                if (t >= len) continue;
                float e[R];
                emis(j, e);
                fetch(j, (t + CTC_PF < len) ? t + CTC_PF : len - 1);
                if (j == 0 && t == 0) {
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r] = (lane * R + r < 2) ? e[r] : CTC_LOGZERO;
                } else {
                    // neighbours: state s-1 and s-2; across the lane boundary they live in lane-1
                    const float p1 = lane_prev(a[R - 1], CTC_LOGZERO);
                    const float p2 = (R >= 2) ? lane_prev(a[R >= 2 ? R - 2 : 0], CTC_LOGZERO)
                                              : lane_prev(p1, CTC_LOGZERO);
                    float na[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const float x1 = r >= 1 ? a[r >= 1 ? r - 1 : 0] : p1;
                        float v;
                        if (kParity && (r % 2) == 0) {
                            v = lse2_2(a[r], x1);
                        } else {
                            float x2 = r >= 2 ? a[r >= 2 ? r - 2 : 0] : (r == 1 ? p1 : p2);
                            if (!skip_in[r]) x2 = CTC_LOGZERO;
                            v = lse3_2(a[r], x1, x2);
                        }
                        na[r] = v + e[r];
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r] = na[r];
                }
                if ((j & 3) == 0) {
                    // the values are logarithms: shifting them by the maximum only keeps them small
                    // for the sake of fp32 resolution, so every fourth step is enough
                    float mx = a[0];
#pragma unroll
                    for (int r = 1; r < R; ++r) mx = fmaxf(mx, a[r]);
                    mx = wave_max_dpp(mx);
                    cum += (double)mx;
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r] = fmaxf(a[r] - mx, CTC_LOGZERO);
                }
                store_states<R>(ab + (long)t * Sp, a);
            }
        }
        // path sum from the two final states at t = len-1
        float fin = CTC_LOGZERO;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            if (s == Sb - 1 || (s == Sb - 2 && Sb >= 2)) fin = (fin == CTC_LOGZERO) ? a[r] : lse2_2(fin, a[r]);
        }
        const bool live = fin > 0.5f * CTC_LOGZERO;
        const float fm = wave_max(live ? fin : CTC_LOGZERO);
        const float fs = wave_sum(live ? __builtin_amdgcn_exp2f(fin - fm) : 0.f);
        if (lane == 0) ll[b] = fs > 0.f ? cum + (double)fm + (double)__builtin_amdgcn_logf(fs) : -(double)INFINITY;
    } else {
        // ---------------- beta: beta[len-1][s] = 0 for final states; beta[t][s] = lse over successors
        // s' in {s, s+1, s+2(skip)} of beta[t+1][s'] + e[t+1][s']
        float* bb = beta + (long)b * T_ * Sp + lane * R;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            const bool fin = (s == Sb - 1) || (s == Sb - 2 && Sb >= 2);
            a[r] = fin ? 0.f : CTC_LOGZERO;
        }
        store_states<R>(bb + (long)(len - 1) * Sp, a);
        // step t (going down from len-2) consumes the emissions of t+1; ring slot j of a trip holds
        // the emissions of time t+1 for t = top - j
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j) fetch(j, (len - 1 - j) >= 0 ? len - 1 - j : 0);
        for (int top = len - 2; top >= 0; top -= CTC_PF) {
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) {
                const int t = top - j;
                if (t < 0) continue;
                float e[R];
                emis(j, e);
                fetch(j, (t + 1 - CTC_PF) >= 0 ? t + 1 - CTC_PF : 0);
                // w[s] = beta[t+1][s] + e[t+1][s]
                float w[R];
#pragma unroll
                for (int r = 0; r < R; ++r) w[r] = a[r] + e[r];
                const float n1 = lane_next(w[0], CTC_LOGZERO);
                const float n2 = (R >= 2) ? lane_next(w[R >= 2 ? 1 : 0], CTC_LOGZERO) : lane_next(n1, CTC_LOGZERO);
                float nb[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float x1 = (r + 1 < R) ? w[r + 1 < R ? r + 1 : 0] : n1;
                    if (kParity && (r % 2) == 0) {
                        nb[r] = lse2_2(w[r], x1);
                    } else {
                        float x2 = (r + 2 < R) ? w[r + 2 < R ? r + 2 : 0] : (r + 2 == R ? n1 : n2);
                        if (!skip_out[r]) x2 = CTC_LOGZERO;
                        nb[r] = lse3_2(w[r], x1, x2);
                    }
                }
                if ((j & 3) == 0) {
                    float mx = nb[0];
#pragma unroll
                    for (int r = 1; r < R; ++r) mx = fmaxf(mx, nb[r]);
                    mx = wave_max_dpp(mx);
#pragma unroll
                    for (int r = 0; r < R; ++r) nb[r] = fmaxf(nb[r] - mx, CTC_LOGZERO);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) a[r] = nb[r];
                store_states<R>(bb + (long)t * Sp, a);
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

// loss[b] = -(ln of the blank-relative path sum - sum over t of (lse - logit[blank])); one workgroup
// per sample, of which one wave does this sum.
//
// A sample whose labels do not fit its frames (path sum zero) takes the slow path instead, on all
// four waves: tf.nn.ctc_loss's dense recursion restated literally for that sample (reference call
// site athena/loss.py:38-45) -- every one of the U label states and U+1 blank states of the padded
// label row plus the silent start state, the leaky initial and final vectors, the loss read from the
// backward recursion -- in double precision, states in LDS.  It overwrites the sample's alpha / beta
// rows with that lattice (same state order: even = blank, odd = label; log2 units; shifted by the
// row maximum) and raises flags[b], which tells ctc_grad to walk all 2U+1 states.
template <typename T>
__global__ __launch_bounds__(256) void ctc_finish_kernel(const T* __restrict__ logits, int ldl,
                                                         const int32_t* __restrict__ labels,
                                                         const int32_t* __restrict__ label_len,
                                                         const int32_t* __restrict__ logit_len, int T_, int V,
                                                         int U, int blank, int Sp, const float* __restrict__ lse,
                                                         const float* __restrict__ nrm, float* __restrict__ alpha,
                                                         float* __restrict__ beta, const double* __restrict__ ll,
                                                         int32_t* __restrict__ flags, float* __restrict__ loss) {
    __shared__ double st[2][64 * CTC_MAX_R];
    __shared__ float red[32];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    int len = logit_len[b];
    if (len > T_) len = T_;
    const int L = min(max(label_len[b], 0), U);
    if (len <= 0) {
        // no frames: TensorFlow reads the loss off its final vector at the start state
        if (tid == 0) {
            loss[b] = L == 0 ? 0.f : (float)(-CTC_TF_LOG0);
            flags[b] = 0;
        }
        return;
    }
    const double llb = ll[b];
    if (llb != -(double)INFINITY) {
        if (tid >= 64) return;
        double acc = 0.0;
        for (int t = lane; t < len; t += 64) acc += (double)nrm[(long)b * T_ + t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        if (lane == 0) {
            loss[b] = (float)(acc - llb * CTC_LN2);
            flags[b] = 0;
        }
        return;
    }
    // ---------------- slow path
    const int S2 = 2 * U + 1;
    const T* lb = logits + (long)b * T_ * ldl;
    const int32_t* lab = labels + (long)b * U;
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
    float* ab = alpha + (long)b * T_ * Sp;
    float* bb = beta + (long)b * T_ * Sp;
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
        mx = block_max(mx, red);
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
        // beta[t] = bwd[t+1]
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int s = tid + 256 * i;
            if (s < S2) bb[(long)t * Sp + s] = fmaxf((float)(st[cur][s] * (double)CTC_LOG2E), CTC_LOGZERO);
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
        mx = block_max(mx, red);                             // (its barriers also end the reads of w)
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

// ---------------------------------------------------------------- ctc_grad
// One workgroup per logits row at a time.  The row is fetched into registers first; while it is in
// flight every wave works out the normaliser of the state posterior alpha+beta for itself (no
// workgroup barrier), the states are scattered by class into an LDS accumulator, and then
// grad = scale * (softmax - occupancy) is streamed out, each thread clearing the accumulator
// entries it consumed.
#define CTC_OCC_CHUNK 8192
#define CTC_GRAD_VEC 8          // 4-wide column groups per thread and chunk: 256 * 8 * 4 = CTC_OCC_CHUNK
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

template <typename TL, typename TG>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const TL* __restrict__ logits, int ldl,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ logit_len, int T_, int V, int U,
                                                       int blank, int Sp, const float* __restrict__ lse,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ beta,
                                                       const int32_t* __restrict__ flags,
                                                       const float* __restrict__ grad_scale, float host_scale,
                                                       TG* __restrict__ grad, int rows_total, int chunk) {
    extern __shared__ float occ[];       // [chunk]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < chunk; c += blockDim.x) occ[c] = 0.f;
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
        const int Sb = flags[b] ? 2 * U + 1 : 2 * min(max(label_len[b], 0), U) + 1;
        const float* ar = alpha + (long)row * Sp;
        const float* br = beta + (long)row * Sp;
        const TL* lr = logits + (long)row * ldl;
        const float l = lse[row];
        for (int v0 = 0; v0 < ldl; v0 += chunk) {
            // the row (this chunk of it) into registers
            CtcRowVec<TL> x[CTC_GRAD_VEC];
#pragma unroll
            for (int i = 0; i < CTC_GRAD_VEC; ++i) {
                const int c = v0 + (i * 256 + threadIdx.x) * 4;
                if (i * 1024 < chunk && c < ldl) x[i].load(lr + c);
            }
            // posterior over the lattice states of this row: every wave normalises for itself
            float mx = -INFINITY;
            for (int s = lane; s < Sb; s += 64) mx = fmaxf(mx, ar[s] + br[s]);
            mx = wave_max(mx);
            float sm = 0.f;
            for (int s = lane; s < Sb; s += 64) sm += __builtin_amdgcn_exp2f(ar[s] + br[s] - mx);
            sm = wave_sum(sm);
            const float inv = 1.f / sm;
            for (int s = threadIdx.x; s < Sb; s += 256) {
                const int cls = (s & 1) ? labels[(long)b * U + (s >> 1)] : blank;
                if (cls >= v0 && cls < v0 + chunk && cls < V)
                    atomicAdd(&occ[cls - v0], __builtin_amdgcn_exp2f(ar[s] + br[s] - mx) * inv);
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < CTC_GRAD_VEC; ++i) {
                const int cl = (i * 256 + threadIdx.x) * 4;     // column within the chunk
                const int c = v0 + cl;
                if (i * 1024 < chunk && c < ldl) {
                    const float4 o = *(const float4*)(occ + cl);
                    if (o.x != 0.f || o.y != 0.f || o.z != 0.f || o.w != 0.f) *(float4*)(occ + cl) = make_float4(0.f, 0.f, 0.f, 0.f);
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

template <typename T, int R>
static void launch_ctc_fwd(const T* logits, int ldl, const int32_t* labels, const int32_t* label_len,
                           const int32_t* logit_len, int B, int Tn, int V, int U, int blank, const CtcWorkspace& ws,
                           int32_t* argmax, hipStream_t s) {
    hipLaunchKernelGGL((ctc_fwd_kernel<T, R>), dim3(B + B * Tn), dim3(256), 0, s, logits, ldl, labels, label_len,
                       logit_len, B, Tn, V, U, blank, ws.lse, ws.nrm, ws.alpha, ws.beta, ws.ll, argmax);
}

template <typename T>
static void dispatch_ctc_fwd(int R, const T* logits, int ldl, const int32_t* labels, const int32_t* label_len,
                             const int32_t* logit_len, int B, int Tn, int V, int U, int blank, const CtcWorkspace& ws,
                             int32_t* argmax, hipStream_t s) {
#define CTC_FWD(N) launch_ctc_fwd<T, N>(logits, ldl, labels, label_len, logit_len, B, Tn, V, U, blank, ws, argmax, s)
    switch (R) {
        case 1: CTC_FWD(1); break;
        case 2: CTC_FWD(2); break;
        case 3: CTC_FWD(3); break;
        case 4: CTC_FWD(4); break;
        case 5: CTC_FWD(5); break;
        case 7: CTC_FWD(7); break;
        case 10: CTC_FWD(10); break;
        default: CTC_FWD(CTC_MAX_R); break;
    }
#undef CTC_FWD
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
    const int R = ctc_launch_r(2 * U + 1);
    if (logits_dtype == ATHENA_BF16)
        dispatch_ctc_fwd<bf16_t>(R, (const bf16_t*)logits, ldl, labels, label_len, logit_len, B, T, V, U, blank, ws, argmax, s);
    else
        dispatch_ctc_fwd<float>(R, (const float*)logits, ldl, labels, label_len, logit_len, B, T, V, U, blank, ws, argmax, s);
    if (logits_dtype == ATHENA_BF16)
        hipLaunchKernelGGL(ctc_finish_kernel<bf16_t>, dim3(B), dim3(256), 0, s, (const bf16_t*)logits, ldl, labels,
                           label_len, logit_len, T, V, U, blank, ws.Sp, ws.lse, ws.nrm, ws.alpha, ws.beta, ws.ll,
                           ws.flags, loss);
    else
        hipLaunchKernelGGL(ctc_finish_kernel<float>, dim3(B), dim3(256), 0, s, (const float*)logits, ldl, labels,
                           label_len, logit_len, T, V, U, blank, ws.Sp, ws.lse, ws.nrm, ws.alpha, ws.beta, ws.ll,
                           ws.flags, loss);
    ATHENA_CHECK_LAUNCH("athena_ctc_loss_fwd");
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
    hipLaunchKernelGGL((ctc_grad_kernel<TL, TG>), dim3(grid), dim3(256), chunk * sizeof(float), s,                \
                       (const TL*)logits, ldl, labels, label_len, logit_len, T, V, U, blank, ws.Sp, ws.lse,       \
                       ws.alpha, ws.beta, ws.flags, grad_scale, host_scale, (TG*)grad, rows, chunk)
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
        hipLaunchKernelGGL(ce_fwd_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, ldl, targets, V, label_smoothing, row_loss, lse);
    else if (logits_dtype == ATHENA_F32)
        hipLaunchKernelGGL(ce_fwd_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)logits, ldl, targets, V, label_smoothing, row_loss, lse);
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
#define CE_BWD(TL, TG) hipLaunchKernelGGL((ce_bwd_kernel<TL, TG>), dim3(rows), dim3(256), 0, s, (const TL*)logits, ldl, targets, V, label_smoothing, lse, grad_scale, host_scale, (TG*)grad)
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
        hipLaunchKernelGGL(argmax_rows_kernel<bf16_t>, dim3(rows), dim3(256), 0, s, (const bf16_t*)logits, ldl, V, out);
    else if (dtype == ATHENA_F32)
        hipLaunchKernelGGL(argmax_rows_kernel<float>, dim3(rows), dim3(256), 0, s, (const float*)logits, ldl, V, out);
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
    hipLaunchKernelGGL(mpc_loss_fwd_kernel, dim3(B * Tp), dim3(256), 0, (hipStream_t)stream, logits, ldl, target,
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
        hipLaunchKernelGGL(mpc_loss_bwd_kernel<bf16_t>, dim3(B * Tp), dim3(256), 0, (hipStream_t)stream, logits, ldl, target,
                           masked_input, logit_len, Tp, N, grad_scale, host_scale, (bf16_t*)grad);
    else if (grad_dtype == ATHENA_F32)
        hipLaunchKernelGGL(mpc_loss_bwd_kernel<float>, dim3(B * Tp), dim3(256), 0, (hipStream_t)stream, logits, ldl, target,
                           masked_input, logit_len, Tp, N, grad_scale, host_scale, (float*)grad);
    else
        ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_mpc_loss_bwd: grad dtype");
    ATHENA_CHECK_LAUNCH("athena_mpc_loss_bwd");
    return ATHENA_OK;
}

extern "C" int athena_group_sum(const float* x, int ngroups, int group, float scale, float* out,
                                athena_stream_t stream) {
    ATHENA_REQUIRE(x && out && ngroups > 0 && group > 0, ATHENA_E_SHAPE, "athena_group_sum: bad arguments");
    hipLaunchKernelGGL(group_sum_kernel, dim3(ngroups), dim3(256), 0, (hipStream_t)stream, x, group, scale, out);
    ATHENA_CHECK_LAUNCH("athena_group_sum");
    return ATHENA_OK;
}
