// Dense-layer and Conv2D entry points on top of the shared GEMM cores.
#include "gemm_big.h"
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void athena_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* athena_last_error(void) { return g_err; }

// ATHENA_GEMM_BIG: 0 keeps every product on the 128x128 / 64x64 cores (for A/B measurements), 2 sends
// every product the 256x256 core can do to it, whatever its size (for tests); default: by size.
// Read on every call so that a test can switch it.
static int big_tiles_mode() {
    const char* v = getenv("ATHENA_GEMM_BIG");
    return (v && v[0] >= '0' && v[0] <= '2') ? v[0] - '0' : 1;
}
static int big_min_k() {
    static const int k = getenv("ATHENA_GEMM_BIG_MIN_K") ? atoi(getenv("ATHENA_GEMM_BIG_MIN_K")) : 256;
    return k;
}
static bool big_tiles_enabled() { return big_tiles_mode() != 0; }
static bool big_tiles_forced() { return big_tiles_mode() == 2; }

extern "C" int athena_abi_version(void) { return 15; }

// ---------------------------------------------------------------- per-kernel timing sessions
// Between athena_timing_begin and athena_timing_end every launch of the library carries two events
// (common.h, ATHENA_LAUNCH); athena_timing_read returns the kernels' own durations in launch order.
int g_athena_timing_on = 0;
namespace {
std::mutex g_timing_mu;
std::vector<hipEvent_t> g_timing_events;      // start, stop, start, stop, ...
size_t g_timing_used = 0;
}  // namespace

void athena_timing_next(hipEvent_t* start, hipEvent_t* stop) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    if (g_timing_used + 2 > g_timing_events.size()) {
        for (int i = 0; i < 2; ++i) {
            hipEvent_t e = nullptr;
            (void)hipEventCreate(&e);
            g_timing_events.push_back(e);
        }
    }
    *start = g_timing_events[g_timing_used];
    *stop = g_timing_events[g_timing_used + 1];
    g_timing_used += 2;
}

extern "C" int athena_timing_begin(void) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_timing_used = 0;
    g_athena_timing_on = 1;
    return ATHENA_OK;
}

extern "C" int athena_timing_count(void) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    return (int)(g_timing_used / 2);
}

extern "C" int athena_timing_end(void) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    g_athena_timing_on = 0;
    return (int)(g_timing_used / 2);
}

extern "C" int athena_timing_read(int first, int count, float* ms) {
    std::lock_guard<std::mutex> lock(g_timing_mu);
    ATHENA_REQUIRE(ms && first >= 0 && count >= 0 && (size_t)(first + count) * 2 <= g_timing_used, ATHENA_E_SHAPE,
                   "athena_timing_read: launches %d..%d of %zu", first, first + count, g_timing_used / 2);
    for (int i = 0; i < count; ++i) {
        hipEvent_t a = g_timing_events[2 * (size_t)(first + i)], b = g_timing_events[2 * (size_t)(first + i) + 1];
        hipError_t e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[i], a, b);
        ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_timing_read: launch %d: %s", first + i, hipGetErrorString(e));
    }
    return ATHENA_OK;
}

static void fill_epilogue(Epilogue& e, const athena_gemm_desc* d) {
    memset(&e, 0, sizeof(e));
    e.C = d->C;
    e.C_pre = d->C_pre;
    e.bias = d->bias;
    e.aux = d->aux;
    e.rng_state = d->rng_state;
    e.colsum = nullptr;
    e.ldc = d->ldc;
    e.ldaux = d->ldaux;
    e.c_dtype = d->out_dtype;
    e.aux_dtype = d->aux_dtype;
    e.act = d->act;
    e.atomic = d->split_k > 1;
    e.bsum = d->b_colsum;
    e.accum = d->accumulate;
    e.alpha = d->alpha;
    e.drop_thresh = d->drop_p > 0.f ? dropout_threshold(d->drop_p) : 0u;
    e.drop_scale = d->drop_p > 0.f ? 1.f / (1.f - d->drop_p) : 1.f;
    e.rng_site = d->rng_site;
}

static int gemm_validate(const athena_gemm_desc* d) {
    ATHENA_REQUIRE(d && d->A && d->B && d->C, ATHENA_E_SHAPE, "athena_gemm: null operand");
    ATHENA_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, ATHENA_E_SHAPE, "athena_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    ATHENA_REQUIRE(d->split_k >= 0 && d->split_k <= 1024, ATHENA_E_SHAPE, "athena_gemm: bad split_k %d", d->split_k);
    ATHENA_REQUIRE(d->in_dtype == ATHENA_F32 || d->in_dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "athena_gemm: in_dtype");
    ATHENA_REQUIRE(d->out_dtype == ATHENA_F32 || d->out_dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "athena_gemm: out_dtype");
    const int a_cols = d->trans_a ? d->M : d->K;
    const int b_cols = d->trans_b ? d->K : d->N;
    ATHENA_REQUIRE(d->lda >= a_cols && d->ldb >= b_cols && d->ldc >= d->N, ATHENA_E_SHAPE,
                   "athena_gemm: leading dimension too small (lda=%d ldb=%d ldc=%d)", d->lda, d->ldb, d->ldc);
    // a bf16 output can only be split through the workspace (the second pass converts): automatic split only
    if (d->split_k != 1)
        ATHENA_REQUIRE((d->out_dtype == ATHENA_F32 || d->split_k == 0) && d->act == ATHENA_ACT_NONE && !d->bias &&
                           !d->C_pre && d->drop_p == 0.f && !d->accumulate,
                       ATHENA_E_UNSUPPORTED, "athena_gemm: split_k needs a plain output (fp32, or bf16 with split_k = 0)");
    ATHENA_REQUIRE((d->act >= 0 && d->act <= ATHENA_ACT_MUL_DRELU6) ||
                       (d->act >= ATHENA_ACT_UNARY && d->act < ATHENA_ACT_UNARY + ATHENA_UNARY_COUNT) ||
                       (d->act >= ATHENA_ACT_MUL_DUNARY && d->act < ATHENA_ACT_MUL_DUNARY + ATHENA_UNARY_COUNT),
                   ATHENA_E_UNSUPPORTED, "athena_gemm: activation %d", d->act);
    if (d->act == ATHENA_ACT_MUL_DGELU || d->act == ATHENA_ACT_MUL_DRELU6 || d->act >= ATHENA_ACT_MUL_DUNARY)
        ATHENA_REQUIRE(d->aux && d->ldaux >= d->N, ATHENA_E_SHAPE, "athena_gemm: backward epilogue needs aux");
    if (d->drop_p > 0.f)
        ATHENA_REQUIRE(d->rng_state && d->drop_p < 1.f, ATHENA_E_SHAPE, "athena_gemm: dropout needs rng_state and p < 1");
    if (d->in_dtype == ATHENA_BF16) {
        ATHENA_REQUIRE(d->lda % 8 == 0 && d->ldb % 8 == 0, ATHENA_E_ALIGN,
                       "athena_gemm: bf16 needs lda, ldb multiples of 8 (lda=%d ldb=%d)", d->lda, d->ldb);
        ATHENA_REQUIRE((d->trans_a && !d->trans_b) || d->K % 8 == 0, ATHENA_E_ALIGN,
                       "athena_gemm: bf16 needs K %% 8 == 0 unless trans_a && !trans_b (K=%d)", d->K);
        ATHENA_REQUIRE(!d->trans_a || d->lda >= (d->M + 7) / 8 * 8, ATHENA_E_ALIGN, "athena_gemm: trans_a needs lda >= roundup8(M) (M=%d lda=%d)", d->M, d->lda);
        ATHENA_REQUIRE(d->trans_b || d->ldb >= (d->N + 7) / 8 * 8, ATHENA_E_ALIGN, "athena_gemm: !trans_b needs ldb >= roundup8(N) (N=%d ldb=%d)", d->N, d->ldb);
        ATHENA_REQUIRE(athena_aligned16(d->A) && athena_aligned16(d->B), ATHENA_E_ALIGN, "athena_gemm: operands must be 16-byte aligned");
    }
    if (d->b_colsum) {
        ATHENA_REQUIRE(d->trans_a && !d->trans_b, ATHENA_E_UNSUPPORTED, "athena_gemm: b_colsum needs trans_a && !trans_b");
        ATHENA_REQUIRE(d->split_k <= 1, ATHENA_E_UNSUPPORTED, "athena_gemm: b_colsum with an explicit split_k > 1 is not supported");
    }
    return ATHENA_OK;
}

// problem description for ``split`` K splits; partial != null: raw partial sums go there
static void gemm_fill(const athena_gemm_desc* d, int split, float* partial, GemmProblem<PlainRows>& p) {
    memset(&p, 0, sizeof(p));
    p.A = d->A;
    p.B = d->B;
    p.amap.ld = d->lda;
    p.M = d->M;
    p.N = d->N;
    p.nseg = 1;
    p.kseg = d->K;
    p.ldb = d->ldb;
    p.split_k = split;
    fill_epilogue(p.epi, d);
    if (d->in_dtype != ATHENA_BF16) p.epi.bsum = nullptr;     // fp32 operands: stand-alone column sums
    p.epi.atomic = split > 1 ? 1 : 0;
    if (split > 1 && partial) {
        p.epi.atomic = 2;
        p.epi.partial = partial;
        p.epi.ldp = (d->N + 7) / 8 * 8;
    }
}

static int gemm_zero_fill_for_atomics(const athena_gemm_desc* d, float* bsum, hipStream_t stream) {
    hipError_t e = (d->ldc == d->N)
                       ? hipMemsetAsync(d->C, 0, sizeof(float) * (size_t)d->M * d->N, stream)
                       : hipMemset2DAsync(d->C, sizeof(float) * (size_t)d->ldc, 0, sizeof(float) * (size_t)d->N, d->M, stream);
    ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_gemm: memset failed");
    if (bsum) {
        e = hipMemsetAsync(bsum, 0, sizeof(float) * (size_t)d->N, stream);
        ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_gemm: memset failed");
    }
    return ATHENA_OK;
}

extern "C" int athena_gemm(const athena_gemm_desc* d, athena_stream_t stream) {
    int rc = gemm_validate(d);
    if (rc) return rc;
    const GemmPlan plan = gemm_plan(d->M, d->N, d->K, d->in_dtype, 1, d->split_k == 0);
    int split = d->split_k == 0 ? plan.split : d->split_k;
    if (d->b_colsum && d->in_dtype != ATHENA_BF16) {
        rc = athena_colsum(d->B, d->in_dtype, d->K, d->N, d->ldb, d->b_colsum, stream);
        if (rc) return rc;
    }
    bool use_ws = d->split_k == 0 && split > 1 && d->workspace && athena_aligned16(d->workspace) &&
                  (size_t)d->workspace_bytes >= splitk_workspace_bytes(d->M, d->N, split, 1);
    if (d->out_dtype != ATHENA_F32 && split > 1 && !use_ws) split = 1;      // no atomics onto a bf16 output
    GemmProblem<PlainRows> p;
    gemm_fill(d, split, use_ws ? (float*)d->workspace : nullptr, p);
    if (d->split_k == 0 && split > 1 && !use_ws) {
        // no workspace: fp32 atomics onto a zero-filled output
        rc = gemm_zero_fill_for_atomics(d, p.epi.bsum, (hipStream_t)stream);
        if (rc) return rc;
    }
    // large plain products ([M,K] x [N,K]^T or [K,M]^T x [K,N], no split) run on the 256x256 core
    const bool big_layout = (!d->trans_a && d->trans_b) || (d->trans_a && !d->trans_b);
    if (big_tiles_enabled() && big_layout && split == 1 &&
        (big_tiles_forced() || (d->M >= 1024 && d->N >= 1024 && d->K >= big_min_k() && gemm_tiles(d->M, d->N, 256) >= 200)) &&
        gemm_big_ok(p, d->in_dtype, d->trans_a != 0))
        return launch_gemm_big(p, d->trans_a != 0, d->trans_b != 0, (hipStream_t)stream);
    rc = launch_gemm(p, d->in_dtype, d->trans_a != 0, d->trans_b != 0, plan.bt, (hipStream_t)stream);
    if (rc) return rc;
    if (use_ws) {
        // the partial sums are added up by a second kernel in a fixed order
        const long work = (long)(d->M + 1) * (p.epi.ldp / 4);
        ATHENA_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((work + 63) / 64), 1), dim3(256), 0, (hipStream_t)stream,
                           (const float*)p.epi.partial, split, d->M, d->N, p.epi.ldp, (float*)d->C, d->ldc, 0L, d->alpha,
                           p.epi.bsum, d->out_dtype == ATHENA_BF16 ? 1 : 0);
        ATHENA_CHECK_LAUNCH("athena_gemm(split-k reduce)");
    }
    return ATHENA_OK;
}

template <int WT>
static void launch_grouped(const GemmGroup& g, bool ta, bool tb, int blocks, hipStream_t stream) {
    const dim3 grid((unsigned)blocks);
    if (!ta && tb) ATHENA_LAUNCH((gemm_mfma_grouped_kernel<false, true, WT>), grid, dim3(256), 0, stream, g);
    else if (!ta && !tb) ATHENA_LAUNCH((gemm_mfma_grouped_kernel<false, false, WT>), grid, dim3(256), 0, stream, g);
    else if (ta && !tb) ATHENA_LAUNCH((gemm_mfma_grouped_kernel<true, false, WT>), grid, dim3(256), 0, stream, g);
    else ATHENA_LAUNCH((gemm_mfma_grouped_kernel<true, true, WT>), grid, dim3(256), 0, stream, g);
}

extern "C" int athena_gemm_grouped(const athena_gemm_desc* descs, int count, athena_stream_t stream) {
    ATHENA_REQUIRE(descs && count > 0, ATHENA_E_SHAPE, "athena_gemm_grouped: empty group");
    bool uniform = true, all_auto = true, any_auto = false;
    for (int i = 0; i < count; ++i) {
        int rc = gemm_validate(&descs[i]);
        if (rc) return rc;
        uniform = uniform && descs[i].trans_a == descs[0].trans_a && descs[i].trans_b == descs[0].trans_b &&
                  descs[i].in_dtype == descs[0].in_dtype;
        all_auto = all_auto && descs[i].split_k == 0;
        any_auto = any_auto || descs[i].split_k == 0;
        ATHENA_REQUIRE(descs[i].split_k <= 1, ATHENA_E_UNSUPPORTED, "athena_gemm_grouped: split_k must be 0 or 1");
    }
    // one launch needs bf16 operands, one operand layout and either all or no automatic splits
    if (count == 1 || !uniform || descs[0].in_dtype != ATHENA_BF16 || (any_auto && !all_auto)) {
        for (int i = 0; i < count; ++i) {
            int rc = athena_gemm(&descs[i], stream);
            if (rc) return rc;
        }
        return ATHENA_OK;
    }
    hipStream_t s = (hipStream_t)stream;
    for (int g0 = 0; g0 < count; g0 += GEMM_GROUP_MAX) {
        const athena_gemm_desc* d = descs + g0;
        const int n = count - g0 < GEMM_GROUP_MAX ? count - g0 : GEMM_GROUP_MAX;
        // tile edge for the whole group, then one split factor per problem
        long t128 = 0, t64 = 0, min_k8 = 1L << 40;
        bool big = true;
        for (int i = 0; i < n; ++i) {
            t128 += gemm_tiles(d[i].M, d[i].N, 128);
            t64 += gemm_tiles(d[i].M, d[i].N, 64);
            big = big && d[i].M > 64 && d[i].N > 64;
            const long k8 = ((d[i].K + GM_BK - 1) / GM_BK) / 8;
            min_k8 = k8 < min_k8 ? k8 : min_k8;
        }
        if (min_k8 < 1) min_k8 = 1;
        const bool bt128 = big && (t128 >= 192 || (all_auto && t128 * min_k8 >= 256));
        const int bt = bt128 ? 128 : 64;
        const long base = bt128 ? t128 : t64;
        // the same split for every problem of the group, chosen so that all workgroups of the launch
        // find a place in one round (a little below the 512 places: problems whose tile counts do not
        // divide evenly add a few)
        // (the 64x64-tile kernel keeps more workgroups resident: its launches aim at 640 of them)
        long want = !all_auto ? 1 : bt128 ? (GM_WG_SLOTS - 32) / base : (640 + base - 1) / base;
        if (want < 1) want = 1;
        GemmGroup grp;
        ReduceGroup red;
        memset(&grp, 0, sizeof(grp));
        memset(&red, 0, sizeof(red));
        size_t ws_need = 0;
        int splits[GEMM_GROUP_MAX];
        for (int i = 0; i < n; ++i) {
            const long ktiles = (d[i].K + GM_BK - 1) / GM_BK;
            long si = want, cap = ktiles / (bt128 ? 8 : 4);
            if (si > cap) si = cap;
            if (si < 1) si = 1;
            splits[i] = (int)si;
            if (si > 1) ws_need += splitk_workspace_bytes(d[i].M, d[i].N, (int)si, 1);
        }
        char* ws = (char*)d[0].workspace;
        if (ws_need > 0 && !(ws && athena_aligned16(ws) && (size_t)d[0].workspace_bytes >= ws_need)) {
            // no room for the partial sums: run the problems one by one
            for (int i = 0; i < n; ++i) {
                int rc = athena_gemm(&d[i], stream);
                if (rc) return rc;
            }
            continue;
        }
        int blocks = 0, rblocks = 0;
        grp.count = n;
        for (int i = 0; i < n; ++i) {
            float* partial = nullptr;
            if (splits[i] > 1) {
                partial = (float*)ws;
                ws += splitk_workspace_bytes(d[i].M, d[i].N, splits[i], 1);
            }
            gemm_fill(&d[i], splits[i], partial, grp.p[i]);
            grp.first[i] = blocks;
            grp.tiles_m[i] = (d[i].M + bt - 1) / bt;
            grp.tiles_n[i] = (d[i].N + bt - 1) / bt;
            blocks += grp.tiles_m[i] * grp.tiles_n[i] * splits[i];
            if (splits[i] > 1) {
                const int j = red.count++;
                red.first[j] = rblocks;
                red.partial[j] = partial;
                red.C[j] = (float*)d[i].C;
                red.bsum[j] = grp.p[i].epi.bsum;
                red.split[j] = splits[i];
                red.M[j] = d[i].M;
                red.N[j] = d[i].N;
                red.ldp[j] = grp.p[i].epi.ldp;
                red.ldc[j] = d[i].ldc;
                red.alpha[j] = d[i].alpha;
                const long work = (long)(d[i].M + 1) * (grp.p[i].epi.ldp / 4);
                rblocks += (int)((work + 63) / 64);
            }
        }
        grp.first[n] = blocks;
        red.first[red.count] = rblocks;
        if (bt128) launch_grouped<4>(grp, d[0].trans_a != 0, d[0].trans_b != 0, blocks, s);
        else launch_grouped<2>(grp, d[0].trans_a != 0, d[0].trans_b != 0, blocks, s);
        ATHENA_CHECK_LAUNCH("athena_gemm_grouped");
        if (red.count > 0) {
            ATHENA_LAUNCH(splitk_reduce_grouped_kernel, dim3((unsigned)rblocks), dim3(256), 0, s, red);
            ATHENA_CHECK_LAUNCH("athena_gemm_grouped(split-k reduce)");
        }
    }
    return ATHENA_OK;
}

extern "C" int64_t athena_gemm_workspace_bytes(int M, int N, int K, int in_dtype) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int split = gemm_plan(M, N, K, in_dtype, 1, true).split;
    return split > 1 ? (int64_t)splitk_workspace_bytes(M, N, split, 1) : 0;
}

// ------------------------------------------------------------------ Conv2D 3x3 / stride 2 / same
static inline void same_pad(int n, int* out, int* pad_before) {
    int o = (n + 1) / 2;
    int total = (o - 1) * 2 + 3 - n;
    if (total < 0) total = 0;
    *out = o;
    *pad_before = total / 2;
}

static int conv_check(const char* name, int B, int H, int W, int Cin, int Cout, int dtype) {
    ATHENA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, ATHENA_E_SHAPE, "%s: bad shape", name);
    ATHENA_REQUIRE(dtype == ATHENA_F32 || dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "%s: dtype", name);
    if (dtype == ATHENA_BF16)
        ATHENA_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, ATHENA_E_ALIGN, "%s: bf16 needs Cin, Cout multiples of 8", name);
    ATHENA_REQUIRE((long)B * H * W < (1L << 31) / 4, ATHENA_E_SHAPE, "%s: too many pixels", name);
    return ATHENA_OK;
}

extern "C" int athena_conv3x3s2_fwd(const void* in, const void* w, int w_layout, void* out, int B, int H, int W,
                                    int Cin, int Cout, int dtype, double* colstats, athena_stream_t stream) {
    int rc = conv_check("athena_conv3x3s2_fwd", B, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(in && w && out, ATHENA_E_SHAPE, "athena_conv3x3s2_fwd: null operand");
    int Ho, Wo, ph, pw;
    same_pad(H, &Ho, &ph);
    same_pad(W, &Wo, &pw);
    GemmProblem<ConvFwdRows> p;
    memset(&p, 0, sizeof(p));
    p.A = in;
    p.B = w;
    p.amap = ConvFwdRows{H, W, Cin, Ho, Wo, ph, pw, GM_BK / Wo, GM_BK % Wo};
    p.M = B * Ho * Wo;
    p.N = Cout;
    p.nseg = 9;
    p.kseg = Cin;
    p.b_mask = 15;           // segment s = tap s, offsets linear in s
    p.seg_base = 0;
    p.split_k = 1;
    if (w_layout == 1) {   // [Cout][9][Cin]
        p.ldb = 9L * Cin;
        p.b_stride_lo = Cin;
    } else {               // HWIO [9][Cin][Cout]
        p.ldb = Cout;
        p.b_stride_lo = (long)Cin * Cout;
    }
    p.epi.C = out;
    p.epi.ldc = Cout;
    p.epi.c_dtype = dtype;
    p.epi.alpha = 1.f;
    p.epi.colsum = colstats;
    if (colstats) {
        hipError_t e = hipMemsetAsync(colstats, 0, sizeof(double) * 2 * Cout, (hipStream_t)stream);
        ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_conv3x3s2_fwd: memset failed");
    }
    // tall outputs with at least 129 columns go to the 256x256 core (the weights are then read
    // through their [Cout][9][Cin] copy as a row image)
    if (big_tiles_enabled() && w_layout == 1 && ((Cout > 128 && p.M >= 64 * GB_BM) || big_tiles_forced()) &&
        gemm_big_ok(p, dtype, false))
        return launch_gemm_big(p, false, true, (hipStream_t)stream);
    return launch_gemm(p, dtype, false, w_layout == 1, gemm_plan(p.M, p.N, 9L * Cin, dtype, 1, false).bt,
                       (hipStream_t)stream);
}

extern "C" int athena_conv3x3s2_dgrad(const void* dout, const void* w_hwio, void* din, int B, int H, int W, int Cin,
                                      int Cout, int dtype, const void* relu6_out, athena_stream_t stream) {
    int rc = conv_check("athena_conv3x3s2_dgrad", B, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(dout && w_hwio && din, ATHENA_E_SHAPE, "athena_conv3x3s2_dgrad: null operand");
    int Ho, Wo, ph, pw;
    same_pad(H, &Ho, &ph);
    same_pad(W, &Wo, &pw);
    GemmProblem<ConvDgradRows> merged[4];
    int n_merged = 0;
    // padded coordinate hp = h + pad: even hp = 2a is reached by taps kh=0 (ho=a) and kh=2 (ho=a-1),
    // odd hp = 2a+1 by tap kh=1 (ho=a)
    for (int ch = 0; ch < 2; ++ch) {
        for (int cw = 0; cw < 2; ++cw) {
            // first pixel of the class and its 'a'
            int h0 = ((ch - ph) % 2 + 2) % 2;            // smallest h >= 0 with (h + ph) % 2 == ch
            int w0 = ((cw - pw) % 2 + 2) % 2;
            if (h0 >= H || w0 >= W) continue;
            int a0h = (h0 + ph) / 2, a0w = (w0 + pw) / 2;
            int Ah = (H - h0 + 1) / 2, Aw = (W - w0 + 1) / 2;
            int khs[2], dhs[2], nh = 0, kws[2], dws[2], nw = 0;
            if (ch == 0) { khs[0] = 0; dhs[0] = a0h; khs[1] = 2; dhs[1] = a0h - 1; nh = 2; }
            else { khs[0] = 1; dhs[0] = a0h; nh = 1; }
            if (cw == 0) { kws[0] = 0; dws[0] = a0w; kws[1] = 2; dws[1] = a0w - 1; nw = 2; }
            else { kws[0] = 1; dws[0] = a0w; nw = 1; }
            GemmProblem<ConvDgradRows> p;
            memset(&p, 0, sizeof(p));
            p.A = dout;
            p.B = w_hwio;
            p.amap.Ho = Ho; p.amap.Wo = Wo; p.amap.Co = Cout; p.amap.Ah = Ah; p.amap.Aw = Aw;
            // segment s = i * nw + j walks the taps (kh = khs[i], kw = kws[j]) of this class:
            // i = s >> shift, j = s & mask; dh = a0h - i, dw = a0w - j; kh = 2i (or 1), kw = 2j (or 1)
            const int s = nh * nw;
            const int shift = nw == 2 ? 1 : 0, mask = nw == 2 ? 1 : 0;
            p.amap.dh0 = a0h; p.amap.dw0 = a0w; p.amap.shift = shift; p.amap.mask = mask;
            p.b_shift = shift;
            p.b_mask = mask;
            p.b_base = (long)(khs[0] * 3 + kws[0]) * Cin * Cout;
            p.b_stride_hi = nh == 2 ? 6L * Cin * Cout : 0;
            p.b_stride_lo = nw == 2 ? 2L * Cin * Cout : 0;
            (void)dhs; (void)dws;
            p.M = B * Ah * Aw;
            p.N = Cin;
            p.nseg = s;
            p.kseg = Cout;
            p.ldb = Cout;          // B(k = co, n = ci) = w[tap][ci][co]: "[N][K]" with ld = Cout
            p.seg_base = 0;
            p.split_k = 1;
            p.epi.C = din;
            p.epi.ldc = Cin;
            p.epi.c_dtype = dtype;
            p.epi.alpha = 1.f;
            if (relu6_out) {
                // the input of this convolution is a relu6 output: mask the gradient with its gate
                p.epi.act = ATHENA_ACT_MUL_DRELU6;
                p.epi.aux = relu6_out;
                p.epi.ldaux = Cin;
                p.epi.aux_dtype = dtype;
            }
            p.crow_mode = 1;
            p.cr_Ah = Ah; p.cr_Aw = Aw; p.cr_H = H; p.cr_W = W; p.cr_h0 = h0; p.cr_w0 = w0;
            if (big_tiles_enabled() && ((Cin > 128 && p.M >= 64 * GB_BM) || big_tiles_forced()) &&
                gemm_big_ok(p, dtype, false)) {
                // the parity classes that qualify for the large tile go out together below
                merged[n_merged++] = p;
                continue;
            }
            rc = launch_gemm(p, dtype, false, true, gemm_plan(p.M, p.N, (long)p.nseg * Cout, dtype, 1, false).bt,
                             (hipStream_t)stream);
            if (rc) return rc;
        }
    }
    if (n_merged > 0) return launch_gemm_big_multi(merged, n_merged, (hipStream_t)stream);
    return ATHENA_OK;
}

extern "C" int64_t athena_conv3x3s2_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const int rows = B * ((H + 1) / 2) * ((W + 1) / 2);
    return (int64_t)splitk_workspace_bytes(Cin, Cout, gemm_plan(Cin, Cout, rows, dtype, 9, true).split, 9);
}

extern "C" int athena_conv3x3s2_wgrad(const void* in, const void* dout, float* dw_hwio, int B, int H, int W, int Cin,
                                      int Cout, int dtype, void* workspace, int64_t workspace_bytes,
                                      athena_stream_t stream) {
    int rc = conv_check("athena_conv3x3s2_wgrad", B, H, W, Cin, Cout, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(in && dout && dw_hwio, ATHENA_E_SHAPE, "athena_conv3x3s2_wgrad: null operand");
    int Ho, Wo, ph, pw;
    same_pad(H, &Ho, &ph);
    same_pad(W, &Wo, &pw);
    const int rows = B * Ho * Wo;
    // dw[tap][ci][co] = sum_r in(r, tap)[ci] * dout[r][co]: A transposed through the gather, B = dout [K][N]
    // the nine taps are nine independent GEMMs stacked in grid.z; split the long contraction so
    // that taps x splits x output tiles fill the chip
    GemmPlan plan = gemm_plan(Cin, Cout, rows, dtype, 9, true);
    // 256x256 tiles: one workgroup per CU, taps x tiles x splits of about the CU count, every split at
    // least 8 k-tiles long, and the partial sums must fit the workspace
    bool big = big_tiles_enabled() && dtype == ATHENA_BF16 &&
               ((Cin > 128 && Cout > 128 && rows >= 64 * GM_BK) || big_tiles_forced()) && workspace &&
               athena_aligned16(workspace);
    if (big) {
        const long tiles = 9L * ((Cin + GB_BM - 1) / GB_BM) * ((Cout + GB_BN - 1) / GB_BN);
        long sp = 256 / tiles;
        const long ktiles = (rows + GM_BK - 1) / GM_BK;
        if (sp > ktiles / 8) sp = ktiles / 8;
        if (sp < 2 && big_tiles_forced() && ktiles >= 2) sp = 2;
        while (sp > 1 && splitk_workspace_bytes(Cin, Cout, (int)sp, 9) > (size_t)workspace_bytes) --sp;
        if (sp < 2 || splitk_workspace_bytes(Cin, Cout, (int)sp, 9) > (size_t)workspace_bytes) big = false;
        else plan.split = (int)sp;
    }
    const int split = plan.split;
    const int kstep = big ? GB_BK : GM_BK;      // rows of the contraction a core advances per step
    GemmProblem<ConvFwdRows> p;
    memset(&p, 0, sizeof(p));
    p.A = in;
    p.B = dout;
    p.amap = ConvFwdRows{H, W, Cin, Ho, Wo, ph, pw, kstep / Wo, kstep % Wo};
    p.M = Cin;
    p.N = Cout;
    p.nseg = 1;
    p.kseg = rows;
    p.ldb = Cout;
    p.seg_base = 0;
    p.split_k = split;
    p.zbatch = 9;
    p.c_batch_stride = (long)Cin * Cout;
    p.epi.C = dw_hwio;
    p.epi.ldc = Cout;
    p.epi.c_dtype = ATHENA_F32;
    p.epi.alpha = 1.f;
    const bool partial = workspace && athena_aligned16(workspace) &&
                         (size_t)workspace_bytes >= splitk_workspace_bytes(Cin, Cout, split, 9);
    if (partial) {
        p.epi.atomic = 2;
        p.epi.partial = (float*)workspace;
        p.epi.ldp = (Cout + 7) / 8 * 8;
    } else {
        p.epi.atomic = 1;
        hipError_t e = hipMemsetAsync(dw_hwio, 0, sizeof(float) * 9L * Cin * Cout, (hipStream_t)stream);
        ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_conv3x3s2_wgrad: memset failed");
    }
    if (big && partial) rc = launch_gemm_big(p, true, false, (hipStream_t)stream);
    else rc = launch_gemm(p, dtype, true, false, plan.bt, (hipStream_t)stream);
    if (rc) return rc;
    if (partial) {
        const long work = (long)(Cin + 1) * (p.epi.ldp / 4);
        ATHENA_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((work + 63) / 64), 9), dim3(256), 0, (hipStream_t)stream,
                           (const float*)p.epi.partial, split, Cin, Cout, p.epi.ldp, dw_hwio, Cout, (long)Cin * Cout, 1.f,
                           (float*)nullptr, 0);
        ATHENA_CHECK_LAUNCH("athena_conv3x3s2_wgrad(split-k reduce)");
    }
    return ATHENA_OK;
}

// ------------------------------------------------------------------ many weight gradients, one launch
extern "C" int athena_gemm_wgrad_plan(athena_wgrad_item* items, int count, int* total_tiles) {
    ATHENA_REQUIRE(items && count > 0 && total_tiles, ATHENA_E_SHAPE, "athena_gemm_wgrad_plan: empty table");
    for (int i = 0; i < count; ++i) {
        athena_wgrad_item& it = items[i];
        ATHENA_REQUIRE(it.A && it.B && it.C && it.M > 0 && it.N > 0 && it.K > 0, ATHENA_E_SHAPE,
                       "athena_gemm_wgrad_plan: item %d: null operand or empty shape", i);
        ATHENA_REQUIRE(it.lda % 8 == 0 && it.ldb % 8 == 0 && it.lda >= (it.M + 7) / 8 * 8 && it.ldb >= (it.N + 7) / 8 * 8 &&
                           it.ldc >= it.N,
                       ATHENA_E_ALIGN, "athena_gemm_wgrad_plan: item %d: lda=%d ldb=%d ldc=%d for M=%d N=%d", i, it.lda,
                       it.ldb, it.ldc, it.M, it.N);
        ATHENA_REQUIRE(athena_aligned16(it.A) && athena_aligned16(it.B), ATHENA_E_ALIGN,
                       "athena_gemm_wgrad_plan: item %d: operands must be 16-byte aligned", i);
        it.tiles_m = (it.M + GB_BM - 1) / GB_BM;
    }
    // Placement.  A tile streams its two operand panels from memory (2 x K x 256 bf16), and the tiles
    // of one product -- and of neighbouring products with a common operand: the q / k / v projections
    // of a layer share their input -- read the same panel.  Workgroups are dealt to the eight XCDs
    // round-robin (blockIdx % 8), each XCD with an L2 of its own, so the launch maps workgroup b to
    // tile (b % 8) * per + b / 8: a contiguous range of ``per`` tiles per XCD, dispatched in step.
    // The planner fills the eight ranges with whole groups of tiles that share an operand (they run
    // on one XCD at about the same time and the shared panel comes out of its L2 once), longest
    // contraction first and always into the range with the least work so far, so that the ranges
    // finish together.  Ranges are padded to equal length; a padding tile exits at once.  Which XCD a
    // workgroup really runs on is the hardware's business: the placement is for speed only.
    const auto tiles_of = [](const athena_wgrad_item& it) { return (long)it.tiles_m * ((it.N + GB_BN - 1) / GB_BN); };
    struct Group { int first, n; long cost, tiles; };
    std::vector<Group> groups;
    for (int i = 0; i < count; ++i) {
        const athena_wgrad_item& it = items[i];
        const long t = tiles_of(it);
        if (!groups.empty()) {
            const athena_wgrad_item& pv = items[i - 1];
            if (pv.K == it.K && (pv.A == it.A || pv.B == it.B)) {
                Group& g = groups.back();
                g.n += 1; g.cost += t * it.K; g.tiles += t;
                continue;
            }
        }
        groups.push_back({i, 1, t * it.K, t});
    }
    std::vector<int> order(groups.size());
    for (size_t g = 0; g < groups.size(); ++g) order[g] = (int)g;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return groups[a].cost > groups[b].cost; });
    long load[8] = {0}, ntile[8] = {0};
    std::vector<int> bins[8];
    for (int g : order) {
        int k = 0;
        for (int x = 1; x < 8; ++x)
            if (load[x] < load[k]) k = x;
        bins[k].push_back(g);
        load[k] += groups[g].cost;
        ntile[k] += groups[g].tiles;
    }
    long per = 1;
    for (int k = 0; k < 8; ++k) per = std::max(per, ntile[k]);
    ATHENA_REQUIRE(8 * per < (1L << 30), ATHENA_E_SHAPE, "athena_gemm_wgrad_plan: too many tiles");
    std::vector<athena_wgrad_item> placed;
    placed.reserve(count);
    for (int k = 0; k < 8; ++k) {
        long cursor = k * per;
        for (int g : bins[k])
            for (int i = groups[g].first; i < groups[g].first + groups[g].n; ++i) {
                athena_wgrad_item it = items[i];
                it.first_tile = (int32_t)cursor;
                cursor += tiles_of(it);
                placed.push_back(it);
            }
    }
    for (int i = 0; i < count; ++i) items[i] = placed[i];      // ascending first_tile: the kernel searches it
    *total_tiles = (int)(8 * per);
    return ATHENA_OK;
}

extern "C" int athena_gemm_wgrad_table(const athena_wgrad_item* table_dev, int count, int total_tiles,
                                       athena_stream_t stream) {
    ATHENA_REQUIRE(table_dev && count > 0 && total_tiles > 0 && total_tiles % 8 == 0, ATHENA_E_SHAPE,
                   "athena_gemm_wgrad_table: empty table, or a tile count that athena_gemm_wgrad_plan did not return");
    ATHENA_LAUNCH(gemm_big_table_kernel, dim3((unsigned)total_tiles), dim3(GB_THREADS), 0, (hipStream_t)stream,
                       (const GbItem*)table_dev, count);
    ATHENA_CHECK_LAUNCH("athena_gemm_wgrad_table");
    return ATHENA_OK;
}

// ------------------------------------------------------------------ contraction split of a small table
#define WG_FOLD_CHUNK 1024          // floats of one fold a workgroup sums

extern "C" int athena_gemm_wgrad_split(const athena_wgrad_item* items, int count, int chip_tiles, float* workspace,
                                       int64_t workspace_floats, int64_t* need_floats, athena_wgrad_item* out_items,
                                       int out_capacity, int* out_count, athena_wgrad_fold* folds, int fold_capacity,
                                       int* fold_count, int* fold_blocks) {
    ATHENA_REQUIRE(items && count > 0 && need_floats && out_count && fold_count, ATHENA_E_SHAPE,
                   "athena_gemm_wgrad_split: empty table or missing result pointers");
    if (chip_tiles <= 0) chip_tiles = 256;
    // The table's work in units of its longest contraction: that many workgroups are busy for the
    // whole launch, the rest of the chip for less or not at all.
    long kmax = 1, cost = 0;
    for (int i = 0; i < count; ++i) {
        const athena_wgrad_item& it = items[i];
        ATHENA_REQUIRE(it.M > 0 && it.N > 0 && it.K > 0, ATHENA_E_SHAPE, "athena_gemm_wgrad_split: item %d: empty shape", i);
        const long t = (long)((it.M + GB_BM - 1) / GB_BM) * ((it.N + GB_BN - 1) / GB_BN);
        kmax = std::max(kmax, (long)it.K);
        cost += t * it.K;
    }
    const long busy = (cost + kmax - 1) / kmax;
    // parts of the longest contraction: as many as bring the table to about one workgroup per CU, of
    // at least 1024 rows each (a part pays a prologue, an epilogue and a partial result of its own)
    long s = std::min(std::min((long)chip_tiles / std::max(busy, 1L), 8L), kmax / 1024);
    if (s < 2) s = 1;
    const long part_rows = (kmax + s - 1) / s;
    std::vector<int> parts(count, 1);
    long need = 0, n_out = 0, n_fold = 0;
    for (int i = 0; i < count; ++i) {
        const athena_wgrad_item& it = items[i];
        int pi = s > 1 ? (int)((it.K + part_rows / 2) / part_rows) : 1;      // rounded: no sliver of a part
        if (pi < 2 || it.ldc != it.N) pi = 1;
        parts[i] = pi;
        n_out += pi;
        if (pi > 1) {
            need += (long)pi * (((long)it.M * it.N + 3) / 4 * 4 + (it.bsum ? ((long)it.N + 3) / 4 * 4 : 0));
            n_fold += it.bsum ? 2 : 1;
        }
    }
    *need_floats = need;
    *out_count = (int)n_out;
    *fold_count = (int)n_fold;
    if (out_items == nullptr) return ATHENA_OK;       // the sizing call
    ATHENA_REQUIRE(out_items && out_capacity >= n_out && (n_fold == 0 || (folds && fold_capacity >= n_fold && fold_blocks)),
                   ATHENA_E_SHAPE, "athena_gemm_wgrad_split: room for %ld items and %ld folds needed", n_out, n_fold);
    ATHENA_REQUIRE(need == 0 || workspace_floats >= need, ATHENA_E_SHAPE,
                   "athena_gemm_wgrad_split: workspace of %ld floats needed", need);
    ATHENA_REQUIRE(need == 0 || athena_aligned16(workspace), ATHENA_E_ALIGN,
                   "athena_gemm_wgrad_split: workspace must be 16-byte aligned");
    // Part p of every item, then part p + 1 of every item that has one: neighbours that share an
    // operand (the q / k / v projections of a layer) stay neighbours part by part, which is what the
    // planner's placement looks for.
    std::vector<float*> c_part(count, nullptr), b_part(count, nullptr);
    std::vector<long> c_stride(count, 0);
    float* cursor = workspace;
    int f = 0, blocks = 0;
    for (int i = 0; i < count; ++i) {
        if (parts[i] < 2) continue;
        const athena_wgrad_item& it = items[i];
        const long nc = ((long)it.M * it.N + 3) / 4 * 4, nb = it.bsum ? ((long)it.N + 3) / 4 * 4 : 0;
        c_part[i] = cursor;
        b_part[i] = it.bsum ? cursor + nc : nullptr;
        c_stride[i] = nc + nb;
        cursor += (long)parts[i] * (nc + nb);
        athena_wgrad_fold& fc = folds[f++];
        fc.dst = it.C; fc.src = c_part[i]; fc.n = (long)it.M * it.N; fc.part_stride = nc + nb; fc.parts = parts[i];
        fc.first_block = blocks;
        blocks += (int)((fc.n + WG_FOLD_CHUNK - 1) / WG_FOLD_CHUNK);
        if (it.bsum) {
            athena_wgrad_fold& fb = folds[f++];
            fb.dst = it.bsum; fb.src = b_part[i]; fb.n = it.N; fb.part_stride = nc + nb; fb.parts = parts[i];
            fb.first_block = blocks;
            blocks += (int)((fb.n + WG_FOLD_CHUNK - 1) / WG_FOLD_CHUNK);
        }
    }
    int o = 0;
    for (int p = 0; p < 8; ++p)
        for (int i = 0; i < count; ++i) {
            if (p >= parts[i]) continue;
            athena_wgrad_item it = items[i];
            if (parts[i] > 1) {
                // equal row ranges, multiples of 8 rows (the last one takes what is left)
                const long rows = ((long)items[i].K + parts[i] - 1) / parts[i];
                const long step = (rows + 7) / 8 * 8;
                const long r0 = std::min((long)p * step, (long)items[i].K);
                const long r1 = p + 1 == parts[i] ? (long)items[i].K : std::min(r0 + step, (long)items[i].K);
                if (r1 <= r0) {
                    // (cannot happen with parts of >= 512 rows; a part without rows would leave its
                    // partial result unwritten)
                    ATHENA_REQUIRE(false, ATHENA_E_SHAPE, "athena_gemm_wgrad_split: item %d: empty part", i);
                }
                it.A = (const char*)items[i].A + r0 * items[i].lda * 2;
                it.B = (const char*)items[i].B + r0 * items[i].ldb * 2;
                it.K = (int32_t)(r1 - r0);
                it.C = c_part[i] + (long)p * c_stride[i];
                it.bsum = b_part[i] ? b_part[i] + (long)p * c_stride[i] : nullptr;
            }
            out_items[o++] = it;
        }
    if (fold_blocks) *fold_blocks = blocks;
    return ATHENA_OK;
}

// dst[j] = sum over parts of src[part * stride + j]: one workgroup per WG_FOLD_CHUNK floats of a fold
__global__ __launch_bounds__(256) void wgrad_fold_kernel(const athena_wgrad_fold* __restrict__ folds, int count) {
    int lo = 0, hi = count - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (folds[mid].first_block <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const athena_wgrad_fold f = folds[lo];
    const long base = (long)((int)blockIdx.x - f.first_block) * WG_FOLD_CHUNK;
    const bool vec = (((uintptr_t)f.dst | (uintptr_t)f.src) & 15) == 0 && (f.part_stride & 3) == 0;
    if (vec) {
#pragma unroll
        for (int i = 0; i < WG_FOLD_CHUNK / 1024; ++i) {
            const long j = base + (i * 256 + threadIdx.x) * 4;
            if (j + 4 <= f.n) {
                float4 acc = *(const float4*)(f.src + j);
                for (int p = 1; p < f.parts; ++p) {
                    const float4 v = *(const float4*)(f.src + p * f.part_stride + j);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
                *(float4*)(f.dst + j) = acc;
            } else {
                for (long jj = j; jj < f.n; ++jj) {
                    float acc = f.src[jj];
                    for (int p = 1; p < f.parts; ++p) acc += f.src[p * f.part_stride + jj];
                    f.dst[jj] = acc;
                }
            }
        }
    } else {
        for (long j = base + threadIdx.x; j < min(base + (long)WG_FOLD_CHUNK, (long)f.n); j += 256) {
            float acc = f.src[j];
            for (int p = 1; p < f.parts; ++p) acc += f.src[p * f.part_stride + j];
            f.dst[j] = acc;
        }
    }
}

extern "C" int athena_gemm_wgrad_fold(const athena_wgrad_fold* folds_dev, int count, int blocks, athena_stream_t stream) {
    ATHENA_REQUIRE(folds_dev && count > 0 && blocks > 0, ATHENA_E_SHAPE, "athena_gemm_wgrad_fold: empty list");
    ATHENA_LAUNCH(wgrad_fold_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, folds_dev, count);
    ATHENA_CHECK_LAUNCH("athena_gemm_wgrad_fold");
    return ATHENA_OK;
}
