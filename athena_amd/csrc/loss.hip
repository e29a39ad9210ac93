// CTC loss (log-softmax + alpha/beta over the blank-augmented lattice + gradient), the
// label-smoothed cross-entropy, and row argmax.  Bandwidth-bound kernels: each logits row is
// streamed with 16-byte loads and reduced with wavefront shuffles.
//
// CTC is three kernels:
//   ctc_prep   : one workgroup per (b,t) row: logsumexp over V and the gather of the 2U+1
//                lattice emissions  em[b,t,s] = logit[b,t,ext_b[s]] - lse[b,t]
//   ctc_lattice: one workgroup of two wavefronts per sample: wavefront 0 runs the alpha
//                recursion forward in time, wavefront 1 the beta recursion backward; each lane
//                owns R consecutive lattice states in registers, neighbours come from lane
//                shuffles, every step is renormalised by the wavefront maximum (the normalisers
//                are summed separately) so fp32 keeps full precision over hundreds of steps
//   ctc_grad   : one workgroup per row: posterior = normalise(alpha+beta) scattered by class
//                into an LDS accumulator, then grad = scale * (softmax - occupancy) streamed out
#include "common.h"

#define CTC_LOGZERO (-1e30f)
#define CTC_PF 8                // emission rows fetched ahead of the lattice recursion
#define CTC_MAX_R 16            // lattice states per lane: supports 2U+1 <= 1024

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct CtcWorkspace {
    float* lse;      // [B*T]
    float* em;       // [B*T*S]
    float* alpha;    // [B*T*S] normalised log alpha (emission at t included)
    float* beta;     // [B*T*S] normalised log beta  (emission at t excluded)
};

static size_t ctc_ws_layout(int B, int T, int U, void* base, CtcWorkspace* ws) {
    const size_t S = 2 * (size_t)U + 1;
    size_t off = 0;
    char* p = (char*)base;
    size_t n_lse = align_up(sizeof(float) * (size_t)B * T, 256);
    size_t n_lat = align_up(sizeof(float) * (size_t)B * T * S, 256);
    if (ws) ws->lse = (float*)(p + off);
    off += n_lse;
    if (ws) ws->em = (float*)(p + off);
    off += n_lat;
    if (ws) ws->alpha = (float*)(p + off);
    off += n_lat;
    if (ws) ws->beta = (float*)(p + off);
    off += n_lat;
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

// ---------------------------------------------------------------- ctc_prep
template <typename T>
__global__ __launch_bounds__(256) void ctc_prep_kernel(const T* __restrict__ logits, int ldl,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ logit_len, int T_, int V, int U,
                                                       int blank, float* __restrict__ lse, float* __restrict__ em) {
    __shared__ float red[32];
    const int row = blockIdx.x;
    const int b = row / T_, t = row - b * T_;
    if (t >= logit_len[b]) return;
    const T* lr = logits + (long)row * ldl;
    float m, s;
    row_max_sum<T>(lr, V, m, s, red);
    const float l = m + __logf(s);
    if (threadIdx.x == 0) lse[row] = l;
    const int S = 2 * U + 1;
    const int Sb = 2 * label_len[b] + 1;
    for (int st = threadIdx.x; st < S; st += blockDim.x) {
        float e = CTC_LOGZERO;
        if (st < Sb) {
            int cls = (st & 1) ? labels[(long)b * U + (st >> 1)] : blank;
            e = to_f32<T>(lr[cls]) - l;
        }
        em[(long)row * S + st] = e;
    }
}

// ---------------------------------------------------------------- ctc_lattice
__device__ __forceinline__ float lse3(float a, float b, float c) {
    float m = fmaxf(a, fmaxf(b, c));
    return m + __logf(__expf(a - m) + __expf(b - m) + __expf(c - m));
}
__device__ __forceinline__ float lse2(float a, float b) {
    float m = fmaxf(a, b);
    return m + __logf(__expf(a - m) + __expf(b - m));
}

// value held by the previous / next lane of the wave (``fill`` where there is none): a single DPP
// move (wave_shr:1 / wave_shl:1) instead of a trip through the LDS crossbar
__device__ __forceinline__ float lane_prev(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v, float fill) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}

// block = 128 threads: wave 0 = alpha (forward), wave 1 = beta (backward)
template <int R>
__global__ __launch_bounds__(128) void ctc_lattice_kernel(const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ label_len,
                                                          const int32_t* __restrict__ logit_len, int T_, int U,
                                                          const float* __restrict__ em, float* __restrict__ alpha,
                                                          float* __restrict__ beta, float* __restrict__ loss) {
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int dir = threadIdx.x >> 6;
    const int S = 2 * U + 1;
    const int L = label_len[b];
    const int Sb = 2 * L + 1;
    int len = logit_len[b];
    if (len > T_) len = T_;
    const float* emb = em + (long)b * T_ * S;
    if (len <= 0) {
        if (dir == 0 && lane == 0) loss[b] = L == 0 ? 0.f : -CTC_LOGZERO;
        return;
    }
    // per-state static data: validity and whether the skip transition (s-2 -> s) is allowed
    bool valid[R], skip_in[R], skip_out[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = lane * R + r;
        valid[r] = s < Sb;
        bool si = false, so = false;
        if ((s & 1) && s < Sb) {
            const int u = s >> 1;
            if (u >= 1) si = labels[(long)b * U + u] != labels[(long)b * U + u - 1];
            if (s + 2 < Sb) so = labels[(long)b * U + u + 1] != labels[(long)b * U + u];
        }
        skip_in[r] = si;
        skip_out[r] = so;
    }
    float a[R];
    if (dir == 0) {
        // ---------------- alpha
        float cum = 0.f;
        // emissions are fetched CTC_PF steps ahead through a ring of registers: a step is far shorter
        // than the latency of a load, so a one-step prefetch would leave every step waiting for memory
        float ring[CTC_PF][R];
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j) {
            const int tj = j < len ? j : len - 1;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int s = lane * R + r;
                ring[j][r] = s < S ? emb[(long)tj * S + s] : CTC_LOGZERO;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            a[r] = (valid[r] && s < 2) ? ring[0][r] : CTC_LOGZERO;
        }
        for (int tb = 0; tb < len; tb += CTC_PF) {
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) {
                const int t = tb + j;
                if (t >= len) break;
                float e[R];
                const int tn = (t + CTC_PF < len) ? t + CTC_PF : len - 1;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int s = lane * R + r;
                    e[r] = ring[j][r];
                    ring[j][r] = s < S ? emb[(long)tn * S + s] : CTC_LOGZERO;
                }
                if (t > 0) {
                    // neighbours: state s-1 and s-2; across the lane boundary they live in lane-1
                    const float p1 = lane_prev(a[R - 1], CTC_LOGZERO);
                    const float p2 = lane_prev(R >= 2 ? a[R - 2] : a[R - 1], CTC_LOGZERO);
                    const float q2 = (R == 1) ? lane_prev(p1, CTC_LOGZERO) : 0.f;
                    float na[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        float x1, x2;
                        if (r >= 1) x1 = a[r - 1];
                        else x1 = p1;
                        if (r >= 2) x2 = a[r - 2];
                        else if (R == 1) x2 = q2;
                        else if (r == 1) x2 = p1;
                        else x2 = p2;
                        if (!skip_in[r]) x2 = CTC_LOGZERO;
                        float v = lse3(a[r], x1, x2) + e[r];
                        na[r] = valid[r] ? v : CTC_LOGZERO;
                    }
#pragma unroll
                    for (int r = 0; r < R; ++r) a[r] = na[r];
                }
                // renormalise by the maximum over states.  The values are logarithms, so this only
                // keeps them small for the sake of fp32 resolution (the gradient kernel subtracts
                // alpha + beta from the log-likelihood): every fourth step is enough, and the
                // cross-lane maximum is the longest dependency chain of a step.
                float mx = 0.f;
                if ((t & 3) == 0 || t == len - 1) {
                    mx = a[0];
#pragma unroll
                    for (int r = 1; r < R; ++r) mx = fmaxf(mx, a[r]);
                    mx = wave_max(mx);
                }
                cum += mx;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    a[r] = valid[r] ? fmaxf(a[r] - mx, CTC_LOGZERO) : CTC_LOGZERO;
                    const int s = lane * R + r;
                    if (s < S) alpha[((long)b * T_ + t) * S + s] = a[r];
                }
            }
        }
        // log-likelihood from the two final states at t = len-1
        float fin = CTC_LOGZERO;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            if (s == Sb - 1 || (s == Sb - 2 && Sb >= 2)) fin = (fin == CTC_LOGZERO) ? a[r] : lse2(fin, a[r]);
        }
        // combine across lanes (at most two lanes hold a final state)
        float fm = wave_max(fin);
        float fs = wave_sum(fin <= CTC_LOGZERO ? 0.f : __expf(fin - fm));
        // no alignment fits: report the finite log-zero instead of -log(0)
        if (lane == 0) loss[b] = fs > 0.f ? -(cum + fm + __logf(fs)) : -CTC_LOGZERO;
    } else {
        // ---------------- beta: beta[len-1][s] = 0 for final states; beta[t][s] = lse over successors
        // s' in {s, s+1, s+2(skip)} of beta[t+1][s'] + em[t+1][s']
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int s = lane * R + r;
            const bool fin = (s == Sb - 1) || (s == Sb - 2 && Sb >= 2);
            a[r] = fin ? 0.f : CTC_LOGZERO;
            if (s < S) beta[((long)b * T_ + (len - 1)) * S + s] = a[r];
        }
        // step t (going down from len-2) consumes the emissions of t+1; ring slot j of a trip holds
        // the emissions of time t+1 for t = top - j
        float ring[CTC_PF][R];
#pragma unroll
        for (int j = 0; j < CTC_PF; ++j) {
            const int tj = (len - 1 - j) >= 0 ? len - 1 - j : 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int s = lane * R + r;
                ring[j][r] = s < S ? emb[(long)tj * S + s] : CTC_LOGZERO;
            }
        }
        for (int top = len - 2; top >= 0; top -= CTC_PF) {
#pragma unroll
            for (int j = 0; j < CTC_PF; ++j) {
                const int t = top - j;
                if (t < 0) break;
                float e[R];
                const int tn = (t + 1 - CTC_PF) >= 0 ? t + 1 - CTC_PF : 0;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int s = lane * R + r;
                    e[r] = ring[j][r];
                    ring[j][r] = s < S ? emb[(long)tn * S + s] : CTC_LOGZERO;
                }
                // w[s] = beta[t+1][s] + em[t+1][s]
                float w[R];
#pragma unroll
                for (int r = 0; r < R; ++r) w[r] = valid[r] ? a[r] + e[r] : CTC_LOGZERO;
                const float n1 = lane_next(w[0], CTC_LOGZERO);
                const float n2 = lane_next(R >= 2 ? w[1] : w[0], CTC_LOGZERO);
                const float m2 = (R == 1) ? lane_next(n1, CTC_LOGZERO) : 0.f;
                float nb[R];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float x1, x2;
                    if (r + 1 < R) x1 = w[r + 1];
                    else x1 = n1;
                    if (r + 2 < R) x2 = w[r + 2];
                    else if (R == 1) x2 = m2;
                    else if (r + 2 == R) x2 = n1;
                    else x2 = n2;
                    if (!skip_out[r]) x2 = CTC_LOGZERO;
                    float v = lse3(w[r], x1, x2);
                    nb[r] = valid[r] ? v : CTC_LOGZERO;
                }
                float mx = 0.f;
                if ((t & 3) == 0) {
                    mx = nb[0];
#pragma unroll
                    for (int r = 1; r < R; ++r) mx = fmaxf(mx, nb[r]);
                    mx = wave_max(mx);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    a[r] = valid[r] ? fmaxf(nb[r] - mx, CTC_LOGZERO) : CTC_LOGZERO;
                    const int s = lane * R + r;
                    if (s < S) beta[((long)b * T_ + t) * S + s] = a[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------- ctc_grad
#define CTC_OCC_CHUNK 8192
template <typename TL, typename TG>
__global__ __launch_bounds__(256) void ctc_grad_kernel(const TL* __restrict__ logits, int ldl,
                                                       const int32_t* __restrict__ labels,
                                                       const int32_t* __restrict__ label_len,
                                                       const int32_t* __restrict__ logit_len, int T_, int V, int U,
                                                       int blank, const float* __restrict__ lse,
                                                       const float* __restrict__ alpha,
                                                       const float* __restrict__ beta,
                                                       const float* __restrict__ grad_scale, float host_scale,
                                                       TG* __restrict__ grad, int rows_total) {
    __shared__ float occ[CTC_OCC_CHUNK];
    __shared__ float red[32];
    const int S = 2 * U + 1;
    for (int c = threadIdx.x; c < CTC_OCC_CHUNK; c += blockDim.x) occ[c] = 0.f;
    __syncthreads();
    const float scale = host_scale * (grad_scale ? grad_scale[0] : 1.f);
    for (int row = blockIdx.x; row < rows_total; row += gridDim.x) {
        const int b = row / T_, t = row - b * T_;
        TG* gr = grad + (long)row * ldl;
        if (t >= logit_len[b]) {
            for (int c = threadIdx.x; c < ldl; c += blockDim.x) gr[c] = from_f32<TG>(0.f);
            continue;
        }
        const int Sb = 2 * label_len[b] + 1;
        const float* ar = alpha + (long)row * S;
        const float* br = beta + (long)row * S;
        // posterior over lattice states, normalised per time step
        float mx = -INFINITY;
        for (int s = threadIdx.x; s < Sb; s += blockDim.x) mx = fmaxf(mx, ar[s] + br[s]);
        mx = block_max(mx, red);
        float sm = 0.f;
        for (int s = threadIdx.x; s < Sb; s += blockDim.x) sm += __expf(ar[s] + br[s] - mx);
        sm = block_sum(sm, red);
        const float inv = 1.f / sm;
        const TL* lr = logits + (long)row * ldl;
        const float l = lse[row];
        for (int v0 = 0; v0 < V; v0 += CTC_OCC_CHUNK) {
            const int v1 = min(V, v0 + CTC_OCC_CHUNK);
            for (int s = threadIdx.x; s < Sb; s += blockDim.x) {
                const int cls = (s & 1) ? labels[(long)b * U + (s >> 1)] : blank;
                if (cls >= v0 && cls < v1) atomicAdd(&occ[cls - v0], __expf(ar[s] + br[s] - mx) * inv);
            }
            __syncthreads();
            for (int c = v0 + threadIdx.x; c < v1; c += blockDim.x) {
                const float p = __expf(to_f32<TL>(lr[c]) - l);
                gr[c] = from_f32<TG>(scale * (p - occ[c - v0]));
            }
            __syncthreads();
            for (int s = threadIdx.x; s < Sb; s += blockDim.x) {
                const int cls = (s & 1) ? labels[(long)b * U + (s >> 1)] : blank;
                if (cls >= v0 && cls < v1) occ[cls - v0] = 0.f;
            }
            __syncthreads();
        }
        for (int c = V + threadIdx.x; c < ldl; c += blockDim.x) gr[c] = from_f32<TG>(0.f);
    }
}

template <int R>
static void launch_lattice(int B, hipStream_t s, const int32_t* labels, const int32_t* label_len,
                           const int32_t* logit_len, int T, int U, const CtcWorkspace& ws, float* loss) {
    hipLaunchKernelGGL(ctc_lattice_kernel<R>, dim3(B), dim3(128), 0, s, labels, label_len, logit_len, T, U, ws.em,
                       ws.alpha, ws.beta, loss);
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

extern "C" int athena_ctc_loss_fwd(const void* logits, int logits_dtype, int ldl, const int32_t* labels,
                                   const int32_t* label_len, const int32_t* logit_len, int B, int T, int V, int U,
                                   int blank, float* loss, void* workspace, size_t workspace_bytes,
                                   athena_stream_t stream) {
    int rc = ctc_check("athena_ctc_loss_fwd", logits, logits_dtype, ldl, B, T, V, U, blank, workspace, workspace_bytes);
    if (rc) return rc;
    ATHENA_REQUIRE(labels && label_len && logit_len && loss, ATHENA_E_SHAPE, "athena_ctc_loss_fwd: null operand");
    CtcWorkspace ws;
    ctc_ws_layout(B, T, U, workspace, &ws);
    hipStream_t s = (hipStream_t)stream;
    if (logits_dtype == ATHENA_BF16)
        hipLaunchKernelGGL(ctc_prep_kernel<bf16_t>, dim3(B * T), dim3(256), 0, s, (const bf16_t*)logits, ldl, labels, label_len, logit_len, T, V, U, blank, ws.lse, ws.em);
    else
        hipLaunchKernelGGL(ctc_prep_kernel<float>, dim3(B * T), dim3(256), 0, s, (const float*)logits, ldl, labels, label_len, logit_len, T, V, U, blank, ws.lse, ws.em);
    const int S = 2 * U + 1;
    const int R = (S + 63) / 64;
    if (R <= 1) launch_lattice<1>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    else if (R <= 2) launch_lattice<2>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    else if (R <= 3) launch_lattice<3>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    else if (R <= 4) launch_lattice<4>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    else if (R <= 5) launch_lattice<5>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    else if (R <= 7) launch_lattice<7>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    else if (R <= 10) launch_lattice<10>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    else launch_lattice<CTC_MAX_R>(B, s, labels, label_len, logit_len, T, U, ws, loss);
    ATHENA_CHECK_LAUNCH("athena_ctc_loss_fwd");
    return ATHENA_OK;
}

extern "C" int athena_ctc_loss_bwd(const void* logits, int logits_dtype, int ldl, const int32_t* labels,
                                   const int32_t* label_len, const int32_t* logit_len, int B, int T, int V, int U,
                                   int blank, const float* grad_scale, float host_scale, void* grad, int grad_dtype,
                                   void* workspace, size_t workspace_bytes, athena_stream_t stream) {
    int rc = ctc_check("athena_ctc_loss_bwd", logits, logits_dtype, ldl, B, T, V, U, blank, workspace, workspace_bytes);
    if (rc) return rc;
    ATHENA_REQUIRE(labels && label_len && logit_len && grad, ATHENA_E_SHAPE, "athena_ctc_loss_bwd: null operand");
    ATHENA_REQUIRE(grad_dtype == ATHENA_F32 || grad_dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "athena_ctc_loss_bwd: grad dtype");
    CtcWorkspace ws;
    ctc_ws_layout(B, T, U, workspace, &ws);
    hipStream_t s = (hipStream_t)stream;
    const int rows = B * T;
    const int grid = rows < 2048 ? rows : 2048;
#define CTC_GRAD(TL, TG)                                                                                          \
    hipLaunchKernelGGL((ctc_grad_kernel<TL, TG>), dim3(grid), dim3(256), 0, s, (const TL*)logits, ldl, labels,    \
                       label_len, logit_len, T, V, U, blank, ws.lse, ws.alpha, ws.beta, grad_scale, host_scale,   \
                       (TG*)grad, rows)
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
            const float lp_t = to_f32<T>(lr[tgt]) - l;
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
    // sum_v y_v = (1-ls) + V*ls/V = 1
    for (int c = threadIdx.x; c < V; c += blockDim.x) {
        const float p = __expf(to_f32<TL>(lr[c]) - l);
        const float y = uni + (c == tgt ? 1.f - ls : 0.f);
        gr[c] = from_f32<TG>(scale * (p - y));
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

extern "C" int athena_group_sum(const float* x, int ngroups, int group, float scale, float* out,
                                athena_stream_t stream) {
    ATHENA_REQUIRE(x && out && ngroups > 0 && group > 0, ATHENA_E_SHAPE, "athena_group_sum: bad arguments");
    hipLaunchKernelGGL(group_sum_kernel, dim3(ngroups), dim3(256), 0, (hipStream_t)stream, x, group, scale, out);
    ATHENA_CHECK_LAUNCH("athena_group_sum");
    return ATHENA_OK;
}
