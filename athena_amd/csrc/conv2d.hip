// Conv2D with any kernel size and strides, "same" padding, NHWC, no bias: the two convolutions of the
// DeepSpeech front end (reference athena/models/deep_speech.py:49-66: 41x11 stride (2,2), 21x11
// stride (2,1)).  Forward, input gradient and weight gradient are implicit GEMMs on the shared cores
// through the tap row maps; a layer whose input has fewer than 8 channels goes through an explicit
// im2col matrix instead (its rows are too short for the 16-byte gathers).
#include "gemm_core.h"
#include <string.h>

// tf.nn's "SAME": out = ceil(n / s), total padding max((out-1)*s + k - n, 0), the smaller half first
static inline void same_geometry(int n, int k, int s, int* out, int* pad_before) {
    const int o = (n + s - 1) / s;
    int total = (o - 1) * s + k - n;
    if (total < 0) total = 0;
    *out = o;
    *pad_before = total / 2;
}

extern "C" int athena_conv2d_out_shape(int H, int W, int KH, int KW, int sh, int sw, int* Ho, int* Wo) {
    ATHENA_REQUIRE(H > 0 && W > 0 && KH > 0 && KW > 0 && sh > 0 && sw > 0 && Ho && Wo, ATHENA_E_SHAPE,
                   "athena_conv2d_out_shape: bad arguments");
    int p;
    same_geometry(H, KH, sh, Ho, &p);
    same_geometry(W, KW, sw, Wo, &p);
    return ATHENA_OK;
}

static int conv2d_check(const char* name, int B, int H, int W, int Cin, int Cout, int KH, int KW, int sh, int sw,
                        int dtype) {
    ATHENA_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && KH > 0 && KW > 0 && sh > 0 && sw > 0,
                   ATHENA_E_SHAPE, "%s: bad shape", name);
    ATHENA_REQUIRE(KH * KW <= 1024, ATHENA_E_UNSUPPORTED, "%s: more than 1024 taps", name);
    ATHENA_REQUIRE(dtype == ATHENA_F32 || dtype == ATHENA_BF16, ATHENA_E_UNSUPPORTED, "%s: dtype", name);
    ATHENA_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, ATHENA_E_ALIGN,
                   "%s: Cin and Cout must be multiples of 8 (fewer input channels: athena_im2col + athena_gemm)", name);
    ATHENA_REQUIRE((long)B * H * W < (1L << 31) / 4, ATHENA_E_SHAPE, "%s: too many pixels", name);
    return ATHENA_OK;
}

// tap s reads the weights at w + s * Cin * Cout: linear in s for every s < 1024
template <class P>
static void linear_taps(P& p, long stride) {
    p.b_shift = 10;
    p.b_mask = 1023;
    p.b_stride_lo = stride;
    p.b_stride_hi = 1024 * stride;
}

extern "C" int athena_conv2d_fwd(const void* in, const void* w_hwio, void* out, int B, int H, int W, int Cin, int Cout,
                                 int KH, int KW, int sh, int sw, int dtype, double* colstats, athena_stream_t stream) {
    int rc = conv2d_check("athena_conv2d_fwd", B, H, W, Cin, Cout, KH, KW, sh, sw, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(in && w_hwio && out, ATHENA_E_SHAPE, "athena_conv2d_fwd: null operand");
    int Ho, Wo, ph, pw;
    same_geometry(H, KH, sh, &Ho, &ph);
    same_geometry(W, KW, sw, &Wo, &pw);
    GemmProblem<ConvTapRows> p;
    memset(&p, 0, sizeof(p));
    p.A = in;
    p.B = w_hwio;
    p.amap = ConvTapRows{H, W, Cin, Ho, Wo, ph, pw, KW, sh, sw, GM_BK / Wo, GM_BK % Wo};
    p.M = B * Ho * Wo;
    p.N = Cout;
    p.nseg = KH * KW;
    p.kseg = Cin;
    p.ldb = Cout;                        // HWIO: B(k = ci, n = co) = w[tap][ci][co]
    linear_taps(p, (long)Cin * Cout);
    p.split_k = 1;
    p.epi.C = out;
    p.epi.ldc = Cout;
    p.epi.c_dtype = dtype;
    p.epi.alpha = 1.f;
    p.epi.colsum = colstats;
    if (colstats) {
        hipError_t e = hipMemsetAsync(colstats, 0, sizeof(double) * 2 * Cout, (hipStream_t)stream);
        ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_conv2d_fwd: memset failed");
    }
    return launch_gemm(p, dtype, false, false, gemm_plan(p.M, p.N, (long)p.nseg * Cin, dtype, 1, false).bt,
                       (hipStream_t)stream);
}

extern "C" int athena_conv2d_dgrad(const void* dout, const void* w_hwio, void* din, int B, int H, int W, int Cin,
                                   int Cout, int KH, int KW, int sh, int sw, int dtype, const void* relu6_out,
                                   athena_stream_t stream) {
    int rc = conv2d_check("athena_conv2d_dgrad", B, H, W, Cin, Cout, KH, KW, sh, sw, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(dout && w_hwio && din, ATHENA_E_SHAPE, "athena_conv2d_dgrad: null operand");
    int Ho, Wo, ph, pw;
    same_geometry(H, KH, sh, &Ho, &ph);
    same_geometry(W, KW, sw, &Wo, &pw);
    // every tap is a segment; the taps whose stride phase does not match an input pixel contribute a
    // zero row there (1 - 1/(sh*sw) of the matrix-core work is spent on zeros)
    GemmProblem<ConvTapDgradRows> p;
    memset(&p, 0, sizeof(p));
    p.A = dout;
    p.B = w_hwio;
    p.amap = ConvTapDgradRows{H, W, Ho, Wo, Cout, ph, pw, KW, sh, sw};
    p.M = B * H * W;
    p.N = Cin;
    p.nseg = KH * KW;
    p.kseg = Cout;
    p.ldb = Cout;                        // B(k = co, n = ci) = w[tap][ci][co]: "[N][K]" with ld = Cout
    linear_taps(p, (long)Cin * Cout);
    p.split_k = 1;
    p.epi.C = din;
    p.epi.ldc = Cin;
    p.epi.c_dtype = dtype;
    p.epi.alpha = 1.f;
    if (relu6_out) {
        p.epi.act = ATHENA_ACT_MUL_DRELU6;
        p.epi.aux = relu6_out;
        p.epi.ldaux = Cin;
        p.epi.aux_dtype = dtype;
    }
    return launch_gemm(p, dtype, false, true, gemm_plan(p.M, p.N, (long)p.nseg * Cout, dtype, 1, false).bt,
                       (hipStream_t)stream);
}

extern "C" int64_t athena_conv2d_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int sh,
                                                       int sw, int dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || sh <= 0 || sw <= 0) return 0;
    int Ho, Wo, p;
    same_geometry(H, KH, sh, &Ho, &p);
    same_geometry(W, KW, sw, &Wo, &p);
    const int rows = B * Ho * Wo;
    return (int64_t)splitk_workspace_bytes(Cin, Cout, gemm_plan(Cin, Cout, rows, dtype, KH * KW, true).split, KH * KW);
}

extern "C" int athena_conv2d_wgrad(const void* in, const void* dout, float* dw_hwio, int B, int H, int W, int Cin,
                                   int Cout, int KH, int KW, int sh, int sw, int dtype, void* workspace,
                                   int64_t workspace_bytes, athena_stream_t stream) {
    int rc = conv2d_check("athena_conv2d_wgrad", B, H, W, Cin, Cout, KH, KW, sh, sw, dtype);
    if (rc) return rc;
    ATHENA_REQUIRE(in && dout && dw_hwio, ATHENA_E_SHAPE, "athena_conv2d_wgrad: null operand");
    int Ho, Wo, ph, pw;
    same_geometry(H, KH, sh, &Ho, &ph);
    same_geometry(W, KW, sw, &Wo, &pw);
    const int rows = B * Ho * Wo;
    const int taps = KH * KW;
    // dw[tap][ci][co] = sum_r in(r, tap)[ci] * dout[r][co]: the taps are independent products stacked in
    // grid.z, each with the contraction over the output pixels split so that the chip is filled
    const GemmPlan plan = gemm_plan(Cin, Cout, rows, dtype, taps, true);
    const int kstep = GM_BK;             // rows of the contraction the matrix-core kernel advances per step
    GemmProblem<ConvTapRows> p;
    memset(&p, 0, sizeof(p));
    p.A = in;
    p.B = dout;
    p.amap = ConvTapRows{H, W, Cin, Ho, Wo, ph, pw, KW, sh, sw, kstep / Wo, kstep % Wo};
    p.M = Cin;
    p.N = Cout;
    p.nseg = 1;
    p.kseg = rows;
    p.ldb = Cout;
    p.split_k = plan.split;
    p.zbatch = taps;
    p.c_batch_stride = (long)Cin * Cout;
    p.epi.C = dw_hwio;
    p.epi.ldc = Cout;
    p.epi.c_dtype = ATHENA_F32;
    p.epi.alpha = 1.f;
    const bool partial = workspace && athena_aligned16(workspace) &&
                         (size_t)workspace_bytes >= splitk_workspace_bytes(Cin, Cout, plan.split, taps);
    if (partial) {
        p.epi.atomic = 2;
        p.epi.partial = (float*)workspace;
        p.epi.ldp = (Cout + 7) / 8 * 8;
    } else {
        p.epi.atomic = 1;
        hipError_t e = hipMemsetAsync(dw_hwio, 0, sizeof(float) * (size_t)taps * Cin * Cout, (hipStream_t)stream);
        ATHENA_REQUIRE(e == hipSuccess, ATHENA_E_LAUNCH, "athena_conv2d_wgrad: memset failed");
    }
    rc = launch_gemm(p, dtype, true, false, plan.bt, (hipStream_t)stream);
    if (rc) return rc;
    if (partial) {
        const long work = (long)(Cin + 1) * (p.epi.ldp / 4);
        ATHENA_LAUNCH(splitk_reduce_kernel, dim3((unsigned)((work + 63) / 64), taps), dim3(256), 0,
                           (hipStream_t)stream, (const float*)p.epi.partial, plan.split, Cin, Cout, p.epi.ldp, dw_hwio,
                           Cout, (long)Cin * Cout, 1.f, (float*)nullptr, 0);
        ATHENA_CHECK_LAUNCH("athena_conv2d_wgrad(split-k reduce)");
    }
    return ATHENA_OK;
}

// ------------------------------------------------------------------ explicit im2col
// col[r][tap*Cin + ci] = in[b, sh*ho - pad_h + kh, sw*wo - pad_w + kw, ci] (zero outside the image and in
// the padding columns [KH*KW*Cin, ldcol)).  One thread per element of the matrix.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void im2col_kernel(const TI* __restrict__ in, TO* __restrict__ col, long rows,
                                                     int H, int W, int Cin, int Ho, int Wo, int KW, int K, int ldcol,
                                                     int ph, int pw, int sh, int sw) {
    const long total = rows * ldcol;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const long r = e / ldcol;
        const int k = (int)(e - r * ldcol);
        float v = 0.f;
        if (k < K) {
            const int ci = k % Cin, tap = k / Cin;
            const int kh = tap / KW, kw = tap - kh * KW;
            const int wo = (int)(r % Wo);
            const long t = r / Wo;
            const int ho = (int)(t % Ho);
            const long b = t / Ho;
            const int h = sh * ho - ph + kh, w = sw * wo - pw + kw;
            if (h >= 0 && h < H && w >= 0 && w < W) v = to_f32<TI>(in[((b * H + h) * W + w) * Cin + ci]);
        }
        col[e] = from_f32<TO>(v);
    }
}

extern "C" int athena_im2col(const void* in, int in_dtype, void* col, int col_dtype, int ldcol, int B, int H, int W,
                             int Cin, int KH, int KW, int sh, int sw, athena_stream_t stream) {
    ATHENA_REQUIRE(in && col && B > 0 && H > 0 && W > 0 && Cin > 0 && KH > 0 && KW > 0 && sh > 0 && sw > 0,
                   ATHENA_E_SHAPE, "athena_im2col: bad arguments");
    const int K = KH * KW * Cin;
    ATHENA_REQUIRE(ldcol >= K, ATHENA_E_SHAPE, "athena_im2col: ldcol %d below KH*KW*Cin = %d", ldcol, K);
    int Ho, Wo, ph, pw;
    same_geometry(H, KH, sh, &Ho, &ph);
    same_geometry(W, KW, sw, &Wo, &pw);
    const long rows = (long)B * Ho * Wo;
    long blocks = (rows * ldcol + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipStream_t s = (hipStream_t)stream;
#define IM2COL(TI, TO)                                                                                              \
    ATHENA_LAUNCH((im2col_kernel<TI, TO>), dim3((unsigned)blocks), dim3(256), 0, s, (const TI*)in, (TO*)col, rows, \
                       H, W, Cin, Ho, Wo, KW, K, ldcol, ph, pw, sh, sw)
    if (in_dtype == ATHENA_F32 && col_dtype == ATHENA_F32) IM2COL(float, float);
    else if (in_dtype == ATHENA_F32 && col_dtype == ATHENA_BF16) IM2COL(float, bf16_t);
    else if (in_dtype == ATHENA_BF16 && col_dtype == ATHENA_BF16) IM2COL(bf16_t, bf16_t);
    else ATHENA_REQUIRE(false, ATHENA_E_UNSUPPORTED, "athena_im2col: dtype pair");
#undef IM2COL
    ATHENA_CHECK_LAUNCH("athena_im2col");
    return ATHENA_OK;
}
