// Accuracy scoring on the device: hypothesis extraction from arg-max ids (greedy CTC collapse, or
// cut at eos) and the Levenshtein distance against the non-zero labels, one workgroup per utterance.
// Integer work on a few hundred ids per utterance: nothing here is bandwidth- or matrix-bound, the
// point is that a training step scores its metrics without a device-to-host crossing.
#include "common.h"

namespace {

constexpr int kMaxFrames = 4095;          // ids per utterance: columns 0 .. L of the table, 64 per lane at most

// wave_shr:1 -- lane l receives the value of lane l - 1, lane 0 receives ``first``
__device__ __forceinline__ int wave_shift_up(int v, int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}

// ids that survive into the hypothesis are appended to ``hyp`` (LDS) in order; returns their number.
// MODE 0 (greedy CTC decoding, merge_repeated): id[p] for p < n with id[p] != special (the blank)
// and id[p] != id[p - 1].  MODE 1 (sequence cut at eos, zeros dropped): id[p] != 0 for p < n.
template <int MODE>
__device__ int compact_ids(const int32_t* __restrict__ ids, int n, int special, int* hyp, int* wsum) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int total = 0;
    for (int base = 0; base < n; base += blockDim.x) {
        const int p = base + threadIdx.x;
        bool keep = false;
        int c = 0;
        if (p < n) {
            c = ids[p];
            if (MODE == 0) keep = c != special && (p == 0 || c != ids[p - 1]);
            else keep = c != 0;
        }
        const unsigned long long m = __ballot(keep);
        const int pre = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = total, all = 0;
        for (int i = 0; i < nw; ++i) {
            const int x = wsum[i];
            if (i < w) off += x;
            all += x;
        }
        if (keep) hyp[off + pre] = c;
        total += all;
        __syncthreads();
    }
    return total;
}

constexpr int kInf = 0x3fffffff;

// v (op) the value of the lane a DPP control names; lanes without a source (and rows the mask leaves
// out) see ``kInf``, the identity of min
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_min(int v) {
    return min(v, __builtin_amdgcn_update_dpp(kInf, v, CTRL, ROW_MASK, 0xf, false));
}
// exclusive prefix minimum over the 64 lanes: lane l receives min over lanes < l (kInf in lane 0)
__device__ __forceinline__ int wave_exclusive_min(int v) {
    v = dpp_min<0x111, 0xf>(v);            // row_shr:1
    v = dpp_min<0x112, 0xf>(v);            // row_shr:2
    v = dpp_min<0x114, 0xf>(v);            // row_shr:4
    v = dpp_min<0x118, 0xf>(v);            // row_shr:8   -> inclusive scan inside each row of 16 lanes
    v = dpp_min<0x142, 0xa>(v);            // row_bcast:15 into rows 1 and 3
    v = dpp_min<0x143, 0xc>(v);            // row_bcast:31 into rows 2 and 3 -> inclusive scan of the wave
    return wave_shift_up(v, kInf);
}

// One wave per utterance.  The edit-distance table is swept row by row (one reference label per row),
// a lane holding E consecutive columns (hypothesis prefixes lane E .. lane E + E - 1) in registers:
//   t[j]    = min(D[i-1][j] + 1, D[i-1][j-1] + [ref_i != hyp_j])          -- no dependence inside the row
//   D[i][j] = min_k<=j (t[k] + j - k) = j + prefixmin_j (t[k] - k)         -- insertions as a prefix minimum
// so a row costs E element updates per lane and one wave-wide prefix minimum (six DPP steps): the
// sweep takes as many steps as the utterance has labels, each a few dozen instructions, where an
// anti-diagonal sweep needs a step per label AND per frame.  Nothing in a step touches LDS or memory
// except the (uniform) read of the row's label.
template <int MODE, int E>
__global__ __launch_bounds__(64) void seq_errors_kernel(const int32_t* __restrict__ ids, int ld_ids,
                                                        const int32_t* __restrict__ hyp_len,
                                                        const int32_t* __restrict__ labels, int L, int U, int special,
                                                        int32_t* __restrict__ per_utt, unsigned long long* __restrict__ counts) {
    extern __shared__ int lds[];
    int* hyp = lds;                        // [L]
    int* ref = hyp + L;                    // [U]
    __shared__ int wsum[16];
    __shared__ int cut;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t* row = ids + (long)b * ld_ids;

    int n = L;
    if (MODE == 0) {
        n = min(max(hyp_len[b], 0), L);
    } else if (special != 0) {             // tokens before the first eos
        if (lane == 0) cut = L;
        __syncthreads();
        int first = L;
        for (int p = lane; p < L; p += 64)
            if (row[p] == special) { first = p; break; }
        if (first < L) atomicMin(&cut, first);
        __syncthreads();
        n = cut;
    }
    const int la = compact_ids<MODE>(row, n, special, hyp, wsum);
    const int lb = compact_ids<1>(labels + (long)b * U, U, 0, ref, wsum);

    int dist = la + lb;
    if (la > 0 && lb > 0) {
        int prev[E], hy[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = lane * E + e;
            prev[e] = j;                                               // D[0][j]
            hy[e] = (j >= 1 && j <= la) ? hyp[j - 1] : -2;             // the hypothesis element column j ends with
        }
        int r = ref[0];
        for (int i = 1; i <= lb; ++i) {
            const int rn = ref[min(i, lb - 1)];                        // the next row's label, requested a row ahead
            const int left = wave_shift_up(prev[E - 1], 0);            // D[i-1][lane E - 1]
            int u[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int j = lane * E + e;
                const int diag = e == 0 ? left : prev[e - 1];
                int t = min(prev[e] + 1, diag + (hy[e] != r ? 1 : 0));
                if (j == 0) t = i;                                     // D[i][0]
                u[e] = t - j;
            }
#pragma unroll
            for (int e = 1; e < E; ++e) u[e] = min(u[e], u[e - 1]);
            const int before = wave_exclusive_min(u[E - 1]);
#pragma unroll
            for (int e = 0; e < E; ++e) prev[e] = lane * E + e + min(u[e], before);
            r = rn;
        }
        // D[lb][la] sits in lane la / E, element la % E
        int mine = 0;
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (e == la % E) mine = prev[e];
        dist = __shfl(mine, la / E, 64);
    }
    if (lane == 0) {
        if (per_utt) {
            per_utt[2 * b] = dist;
            per_utt[2 * b + 1] = lb;
        }
        if (counts) {
            atomicAdd(&counts[0], (unsigned long long)dist);
            atomicAdd(&counts[1], (unsigned long long)lb);
        }
    }
}

}  // namespace

extern "C" int athena_seq_errors(const int32_t* ids, int ld_ids, const int32_t* hyp_len, const int32_t* labels, int B,
                                 int L, int U, int mode, int special, int32_t* per_utt, uint64_t* counts,
                                 athena_stream_t stream) {
    ATHENA_REQUIRE(ids && labels && (per_utt || counts), ATHENA_E_SHAPE, "athena_seq_errors: null operand");
    ATHENA_REQUIRE(mode == 0 || mode == 1, ATHENA_E_UNSUPPORTED, "athena_seq_errors: mode %d", mode);
    ATHENA_REQUIRE(mode == 1 || hyp_len, ATHENA_E_SHAPE, "athena_seq_errors: greedy CTC scoring needs the logit lengths");
    ATHENA_REQUIRE(B > 0 && L > 0 && U > 0 && ld_ids >= L, ATHENA_E_SHAPE,
                   "athena_seq_errors: bad shape B=%d L=%d U=%d ld=%d", B, L, U, ld_ids);
    ATHENA_REQUIRE(L <= kMaxFrames, ATHENA_E_UNSUPPORTED, "athena_seq_errors: %d ids per utterance (at most %d)", L,
                   kMaxFrames);
    const size_t lds = sizeof(int) * ((size_t)L + U);
    ATHENA_REQUIRE(lds <= 60 * 1024, ATHENA_E_UNSUPPORTED, "athena_seq_errors: L=%d U=%d need %zu bytes of LDS", L, U, lds);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(MODE, E)                                                                                              \
    ATHENA_LAUNCH((seq_errors_kernel<MODE, E>), dim3(B), dim3(64), lds, s, ids, ld_ids, hyp_len, labels, L, U, special, \
                  per_utt, (unsigned long long*)counts)
#define PICK(MODE)                                  \
    do {                                            \
        if (L < 64 * 4) LAUNCH(MODE, 4);            \
        else if (L < 64 * 8) LAUNCH(MODE, 8);       \
        else if (L < 64 * 16) LAUNCH(MODE, 16);     \
        else if (L < 64 * 32) LAUNCH(MODE, 32);     \
        else LAUNCH(MODE, 64);                      \
    } while (0)
    if (mode == 0) PICK(0);
    else PICK(1);
#undef PICK
#undef LAUNCH
    ATHENA_CHECK_LAUNCH("athena_seq_errors");
    return ATHENA_OK;
}
