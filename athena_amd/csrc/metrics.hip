// Accuracy scoring on the device: hypothesis extraction from arg-max ids (greedy CTC collapse, or
// cut at eos) and the Levenshtein distance against the non-zero labels, one workgroup per utterance.
// Integer work on a few hundred ids per utterance: nothing here is bandwidth- or matrix-bound, the
// point is that a training step scores its metrics without a device-to-host crossing.
#include "common.h"

namespace {

constexpr int kMaxLabels = 1024;          // label columns of one utterance (one per thread)

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

// One workgroup per utterance, thread j - 1 owns column j of the edit-distance table (reference
// label j); the table is swept by anti-diagonals, so a step needs ONE value from the neighbouring
// column (its entry of the previous diagonal; the entry of the diagonal before that is what the
// thread saw one step earlier).  A single wave passes it by a DPP shift, several waves through LDS.
template <int MODE, bool ONE_WAVE>
__global__ void seq_errors_kernel(const int32_t* __restrict__ ids, int ld_ids, const int32_t* __restrict__ hyp_len,
                                  const int32_t* __restrict__ labels, int L, int U, int special,
                                  int32_t* __restrict__ per_utt, unsigned long long* __restrict__ counts) {
    extern __shared__ int lds[];
    int* hyp = lds;                        // [L]
    int* ref = hyp + L;                    // [U]
    int* xch = ref + U;                    // [2][blockDim.x + 1]
    __shared__ int wsum[16];
    __shared__ int cut;
    const int b = blockIdx.x;
    const int32_t* row = ids + (long)b * ld_ids;

    int n = L;
    if (MODE == 0) {
        n = min(max(hyp_len[b], 0), L);
    } else if (special != 0) {             // tokens before the first eos
        if (threadIdx.x == 0) cut = L;
        __syncthreads();
        int first = L;
        for (int p = threadIdx.x; p < L; p += blockDim.x)
            if (row[p] == special) { first = p; break; }
        if (first < L) atomicMin(&cut, first);
        __syncthreads();
        n = cut;
    }
    const int la = compact_ids<MODE>(row, n, special, hyp, wsum);
    const int lb = compact_ids<1>(labels + (long)b * U, U, 0, ref, wsum);

    int dist;
    if (la == 0 || lb == 0) {
        dist = la + lb;
    } else {
        const int j = threadIdx.x + 1;                     // column
        const int rj = j <= lb ? ref[j - 1] : -1;
        int own = j;                                       // D[0][j]
        int diag = j - 1;                                  // D[0][j-1]
        const int steps = la + lb;
        if (ONE_WAVE) {
            // Column j meets hypothesis element s - j - 1 at step s: the element column j - 1 met one
            // step earlier.  The hypothesis therefore travels through the lanes by the same shift as
            // the table entries, entering at lane 0 from a register copy of 64 elements at a time
            // (v_readlane with a uniform index); no LDS access and nothing to wait for inside a step.
            int chunk = hyp[min((int)threadIdx.x, la - 1)];            // elements 0 .. 63
            int hv = 0;
            for (int s = 2; s <= steps; ++s) {
                const int e = s - 2;                                   // the element lane 0 meets at this step
                if (e >= 64 && (e & 63) == 0) chunk = hyp[min(e + (int)threadIdx.x, la - 1)];
                const int head = __builtin_amdgcn_readlane(chunk, e & 63);
                hv = wave_shift_up(hv, head);
                int nb = wave_shift_up(own, 0);
                if (j == 1) nb = s - 1;                                // D[i][0] = i
                const int i = s - j;
                if (j <= lb && i >= 1 && i <= la) {
                    const int sub = diag + (hv != rj ? 1 : 0);
                    own = min(min(own, nb) + 1, sub);
                }
                diag = nb;
            }
            dist = __shfl(own, lb - 1, 64);                // the owner of column lb holds D[la][lb]
        } else {
            for (int s = 2; s <= steps; ++s) {
                int* buf = xch + (s & 1) * (blockDim.x + 1);
                buf[j] = own;
                const int i = s - j;
                const int he = hyp[min(max(i - 1, 0), la - 1)];        // requested before the barrier it can wait behind
                __syncthreads();
                int nb = buf[j - 1];
                if (j == 1) nb = s - 1;                                // D[i][0] = i
                if (j <= lb && i >= 1 && i <= la) {
                    const int sub = diag + (he != rj ? 1 : 0);
                    own = min(min(own, nb) + 1, sub);
                }
                diag = nb;
            }
            __syncthreads();
            if (j == lb) xch[0] = own;
            __syncthreads();
            dist = xch[0];
        }
    }
    if (threadIdx.x == 0) {
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
    ATHENA_REQUIRE(U <= kMaxLabels, ATHENA_E_UNSUPPORTED, "athena_seq_errors: %d labels per utterance (at most %d)", U,
                   kMaxLabels);
    const int threads = (U + 63) / 64 * 64;
    const size_t lds = sizeof(int) * ((size_t)L + U + 2 * (threads + 1));
    ATHENA_REQUIRE(lds <= 60 * 1024, ATHENA_E_UNSUPPORTED, "athena_seq_errors: L=%d U=%d need %zu bytes of LDS", L, U, lds);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH(MODE, ONE)                                                                                          \
    ATHENA_LAUNCH((seq_errors_kernel<MODE, ONE>), dim3(B), dim3(threads), lds, s, ids, ld_ids, hyp_len, labels, L, \
                       U, special, per_utt, (unsigned long long*)counts)
    if (mode == 0) {
        if (threads == 64) LAUNCH(0, true); else LAUNCH(0, false);
    } else {
        if (threads == 64) LAUNCH(1, true); else LAUNCH(1, false);
    }
#undef LAUNCH
    ATHENA_CHECK_LAUNCH("athena_seq_errors");
    return ATHENA_OK;
}
