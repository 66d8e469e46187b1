// Native batch builder (see include/sbr_rnn.h): the reference's SequenceGenerator +
// _gen_mini_batch + _prepare_input (data_handling.py:126-174, rnn_base.py:373-420, rnn_one_hot.py:83-106,
// rnn_sampling.py:159-194) with the training set resident in HBM as CSR and every per-batch array produced by
// two kernels on the engine's stream.  The host only walks the user list once per pass (integer bookkeeping).
#include "sbr_common.h"
#include <algorithm>
#include <unordered_map>

struct sbr_dataset {
    int64_t n_users, nnz; int n_items;
    hipStream_t stream;
    int* d_items; long long* d_off;
    float* d_popdb; double* d_cdf;
    int* d_rate;                                    // rating one-hot index (0..9) of every interaction (--rf), or NULL
    int shuffle_targets;                            // --shuffle_targets: the targets are drawn from the whole remaining sequence
    // sequence noise (sbr_dataset_noise_pass): this pass's noised copy of every user's sequence, stored at the user's own
    // offset (dropout only shortens it) + its length; `noised` = the batches of the pass read the copy
    int *d_items_n, *d_rate_n, *d_len_n; int noised; int64_t max_len;
    std::vector<int> len_n32;
    std::vector<int64_t> len_cur;                   // lengths the current plan was made for (noised or not)
    // --target_bias (sbr_dataset_set_target_bias): whether a row has a target is a draw, so the rows of a pass -- user, split
    // point, target positions -- are planned on the HOST (sbr_plan_rows_host) and uploaded; the device only packs them
    int host_rows, hr_targets; uint64_t hr_seed, hr_pass;
    std::vector<float> keep_prob;
    std::vector<int> h_items, h_items_cur;          // host copy of the sequences (as uploaded / this pass's noised copy)
    std::vector<int64_t> h_off;
    std::vector<int> hr_user, hr_split, hr_tgt, hr_pu, hr_ps, hr_pt;      // rows of the pass (complete batches) | carried rows
    int *d_hr_user, *d_hr_split, *d_hr_tgt; size_t cap_hru, cap_hrs, cap_hrt;
    std::vector<int64_t> len;                       // host copy of the sequence lengths
    std::vector<int> pend_user, pend_k;             // trailing partial batch carried into the next pass
    std::vector<int> seg_user, seg_k, seg_row0, seg_batch, batch_begin;   // host plan (batch_begin: n_batches+1)
    int *d_seg_user, *d_seg_k, *d_seg_row0, *d_batch_begin; size_t cap_su, cap_sk, cap_sr, cap_bb;
    int *d_split, *d_rowuser; size_t cap_rows;      // per-row scratch of the batch being built
    int64_t n_batches; int batch_size;
};


// ---------------------------------------------------------------------------------------
// host planner (rnn_base.py:394-415): users in order; k = min(B - j, len - 2); j += k; a full batch closes
// ---------------------------------------------------------------------------------------
extern "C" int sbr_plan_pass_host(const int64_t* lengths, const int32_t* order, int64_t n_users, int32_t B,
                                  int32_t* pend_user, int32_t* pend_k, int32_t* n_pend, int32_t* seg_user, int32_t* seg_k,
                                  int32_t* seg_row0, int32_t* seg_batch, int64_t* n_segments, int64_t* n_batches) {
    CHECK_ARG(lengths && pend_user && pend_k && n_pend && seg_user && seg_k && seg_row0 && seg_batch && n_segments && n_batches,
              "null argument");
    CHECK_ARG(B >= 1 && n_users >= 0 && *n_pend >= 0 && *n_pend <= B, "bad batch size / pending count");
    int64_t ns = 0, nb = 0; int j = 0;
    for (int i = 0; i < *n_pend; ++i) {             // rows of the partial batch the previous pass left behind
        // (with sequence noise the user's sequence of THIS pass may be shorter than the one the rows were counted for)
        const int k = (int)std::min<int64_t>(pend_k[i], lengths[pend_user[i]] - 2);
        if (k <= 0) continue;
        seg_user[ns] = pend_user[i]; seg_k[ns] = k; seg_row0[ns] = j; seg_batch[ns] = (int)nb; ++ns;
        j += k;
    }
    CHECK_ARG(j < B || *n_pend == 0, "pending rows fill a whole batch");
    for (int64_t i = 0; i < n_users; ++i) {
        const int u = order ? order[i] : (int)i;
        CHECK_ARG(u >= 0 && u < n_users, "user id %d out of range", u);
        const int64_t L = lengths[u];
        if (L < 2) continue;                        // SequenceGenerator min_length=2 (data_handling.py:143)
        const int k = (int)std::min<int64_t>(B - j, L - 2);
        if (k <= 0) continue;                       // len == 2: random.sample(range(2,2), 0) -> no rows, user consumed
        seg_user[ns] = u; seg_k[ns] = k; seg_row0[ns] = j; seg_batch[ns] = (int)nb; ++ns;
        j += k;
        if (j == B) { ++nb; j = 0; }
    }
    // the segments of the unfinished batch nb sit at the tail: they become the next pass's pending prefix
    int64_t first = ns;
    while (first > 0 && seg_batch[first - 1] == nb) --first;
    const int np = (int)(ns - first);
    for (int i = 0; i < np; ++i) { pend_user[i] = seg_user[first + i]; pend_k[i] = seg_k[first + i]; }
    ns = first;
    *n_pend = np; *n_segments = ns; *n_batches = nb;
    return SBR_OK;
}

// ---------------------------------------------------------------------------------------
// Host-planned rows: _gen_mini_batch with a target selection that can come back empty (rnn_base.py:394-415 with
// SelectTargets.__call__, target_selection.py:41-53).  Per user: k = min(B - j, len - 2) sorted distinct split points
// (random.sample); per split point the targets are the first n_targets items of the remaining sequence -- after
// random.shuffle with --shuffle_targets -- that survive a draw against keep_prob[item] (--target_bias); a row without
// a target is skipped and does not count towards the batch.  Rows come out batch-major; the rows of the unfinished last
// batch are carried to the next pass as they are (the reference's generator keeps them too).  Draws: splitmix64 from `seed`.
// ---------------------------------------------------------------------------------------
namespace {
struct HostRng {
    unsigned long long s;
    unsigned long long next() { s += 0x9E3779B97F4A7C15ull; unsigned long long x = s; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }
    int below(int n) { return (int)(next() % (unsigned long long)n); }                       // n <= 2^31: bias < 2^-33
    float unif() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }                   // [0, 1)
};
}
extern "C" int sbr_plan_rows_host(const int32_t* items, const int64_t* offsets, const int64_t* lengths, const int32_t* order,
                                  int64_t n_users, int32_t B, int32_t n_targets, int32_t shuffle, const float* keep_prob, uint64_t seed,
                                  int32_t* pend_user, int32_t* pend_split, int32_t* pend_tgt, int32_t* n_pend, int64_t cap_rows,
                                  int32_t* row_user, int32_t* row_split, int32_t* row_tgt, int64_t* n_rows, int64_t* n_batches) {
    CHECK_ARG(items && offsets && pend_user && pend_split && pend_tgt && n_pend && row_user && row_split && row_tgt && n_rows && n_batches,
              "null argument");
    CHECK_ARG(B >= 1 && n_users >= 0 && n_targets >= 1 && *n_pend >= 0 && *n_pend < B, "bad batch size / targets / pending count");
    HostRng rng{seed * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull};
    int64_t nr = 0;
    int j = 0;
    auto emit = [&](int u, int l, const int* tg) -> bool {
        if (nr >= cap_rows) return false;
        row_user[nr] = u; row_split[nr] = l;
        for (int t = 0; t < n_targets; ++t) row_tgt[nr * n_targets + t] = tg[t];
        ++nr; return true;
    };
    for (int i = 0; i < *n_pend; ++i) {
        if (!emit(pend_user[i], pend_split[i], pend_tgt + (size_t)i * n_targets)) { sbr_set_error("row capacity too small"); return SBR_EINVAL; }
        ++j;
    }
    std::vector<char> mark;
    std::vector<int> tg(n_targets);
    std::unordered_map<int, int> perm;
    for (int64_t ui = 0; ui < n_users; ++ui) {
        const int u = order ? order[ui] : (int)ui;
        CHECK_ARG(u >= 0 && u < n_users, "user id %d out of range", u);
        const int64_t L = lengths ? lengths[u] : offsets[u + 1] - offsets[u];
        if (L < 2) continue;
        const int n = (int)L - 2, k = std::min(B - j, n);
        if (k <= 0) continue;
        mark.assign(n, 0);
        if (k >= n) std::fill(mark.begin(), mark.end(), 1);
        else for (int i = n - k; i < n; ++i) { const int t = rng.below(i + 1); if (mark[t]) mark[i] = 1; else mark[t] = 1; }   // Floyd: a uniform k-subset
        const int32_t* seq = items + offsets[u];
        int skipped = 0;
        for (int c = 0; c < n; ++c) {
            if (!mark[c]) continue;
            const int l = 2 + c, n_rem = (int)L - l;
            int taken = 0;
            std::fill(tg.begin(), tg.end(), -1);
            if (!shuffle) {
                for (int p = 0; p < n_rem && taken < n_targets; ++p)
                    if (!keep_prob || rng.unif() <= keep_prob[seq[l + p]]) tg[taken++] = p;
            } else {      // a lazily drawn uniform order of the remaining sequence (sparse Fisher-Yates), filtered as it comes
                perm.clear();
                for (int i = 0; i < n_rem && taken < n_targets; ++i) {
                    const int r = i + rng.below(n_rem - i);
                    auto ir = perm.find(r); const int vr = ir == perm.end() ? r : ir->second;
                    auto ii = perm.find(i); const int vi = ii == perm.end() ? i : ii->second;
                    perm[r] = vi;
                    if (!keep_prob || rng.unif() <= keep_prob[seq[l + vr]]) tg[taken++] = vr;
                }
            }
            if (taken == 0) { ++skipped; continue; }
            if (!emit(u, l, tg.data())) { sbr_set_error("row capacity too small"); return SBR_EINVAL; }
        }
        j += k - skipped;
        if (j == B) j = 0;
    }
    // the rows of the unfinished batch sit at the tail
    const int64_t nb = nr / B;
    const int np = (int)(nr - nb * B);
    for (int i = 0; i < np; ++i) {
        pend_user[i] = row_user[nb * B + i]; pend_split[i] = row_split[nb * B + i];
        for (int t = 0; t < n_targets; ++t) pend_tgt[(size_t)i * n_targets + t] = row_tgt[(nb * B + i) * n_targets + t];
    }
    *n_pend = np; *n_rows = nb * B; *n_batches = nb;
    return SBR_OK;
}

// ---------------------------------------------------------------------------------------
// device side
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {      // splitmix64 finaliser
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    return x;
}
__device__ __forceinline__ unsigned key32(unsigned long long seed, unsigned a, unsigned b) {
    return (unsigned)(mix64(seed ^ mix64(((unsigned long long)a << 32) | b)) >> 32);
}

// One workgroup per segment: k distinct split points out of the n = len - 2 candidates {2..len-1}, uniformly
// (random.sample, rnn_base.py:402) and in ascending order (sorted(...)).  Every candidate gets a 32-bit key
// hashed from (seed, user, index); the k smallest keys win: a 4-pass radix select in LDS finds the k-th key,
// an ordered compaction writes the winners.  Key ties at the threshold are broken by index.
__global__ void __launch_bounds__(256) bb_split_kernel(const long long* __restrict__ off, const int* __restrict__ len_n,
                                                       const int* __restrict__ seg_user,
                                                       const int* __restrict__ seg_k, const int* __restrict__ seg_row0,
                                                       int seg_begin, unsigned long long seed, int* __restrict__ split,
                                                       int* __restrict__ rowuser) {
    __shared__ int hist[256];
    __shared__ unsigned s_prefix, s_mask;
    __shared__ int s_remaining, s_base, s_ties;
    __shared__ int wsum[8];
    const int s = seg_begin + blockIdx.x, tid = threadIdx.x;
    const int user = seg_user[s], k = seg_k[s], row0 = seg_row0[s];
    const int n = (len_n ? len_n[user] : (int)(off[user + 1] - off[user])) - 2;
    // the segment's first row is part of the seed: a user who owns two segments of one batch (the carried tail of the previous
    // pass meeting the same user in the new pass) draws independent split points for each, as random.sample does
    const unsigned long long sd = seed ^ mix64(0x5EEDull + (unsigned long long)user) ^ mix64(0xB0Bull + ((unsigned long long)(unsigned)row0 << 20));
    if (tid == 0) { s_prefix = 0; s_mask = 0; s_remaining = k; }
    __syncthreads();
    if (k < n) {
        for (int pass = 3; pass >= 0; --pass) {
            hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix, mask = s_mask;
            for (int i = tid; i < n; i += 256) {
                const unsigned key = key32(sd, 1u, (unsigned)i);
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> (8 * pass)) & 255], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int rem = s_remaining, b = 0;
                while (b < 255 && hist[b] < rem) { rem -= hist[b]; ++b; }      // bin holding the k-th smallest key
                s_remaining = rem; s_prefix = prefix | ((unsigned)b << (8 * pass)); s_mask = mask | (0xFFu << (8 * pass));
            }
            __syncthreads();
        }
    }
    const unsigned thr = s_prefix;                 // the k-th smallest key; s_remaining of the keys equal to it are taken
    const int take_ties = s_remaining;
    if (tid == 0) { s_base = 0; s_ties = 0; }
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int i = c0 + tid;
        bool lt = false, eq = false;
        if (i < n) {
            if (k >= n) lt = true;
            else { const unsigned key = key32(sd, 1u, (unsigned)i); lt = key < thr; eq = key == thr; }
        }
        // ordered rank of the equal keys first (needed to decide which of them are taken)
        const int lane = tid & 63, wv = tid >> 6;
        const unsigned long long beq = __ballot(eq);
        int eq_before = __popcll(beq & ((1ull << lane) - 1));
        if (lane == 0) wsum[wv] = __popcll(beq);
        __syncthreads();
        for (int w = 0; w < wv; ++w) eq_before += wsum[w];
        const int eq_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const bool sel = lt || (eq && s_ties + eq_before < take_ties);
        __syncthreads();
        const unsigned long long bs = __ballot(sel);
        int before = __popcll(bs & ((1ull << lane) - 1));
        if (lane == 0) wsum[4 + wv] = __popcll(bs);
        __syncthreads();
        for (int w = 0; w < wv; ++w) before += wsum[4 + w];
        const int total = wsum[4] + wsum[5] + wsum[6] + wsum[7];
        if (sel) { const int pos = s_base + before; if (pos < k) { split[row0 + pos] = 2 + i; rowuser[row0 + pos] = user; } }
        __syncthreads();
        if (tid == 0) { s_base += total; s_ties += eq_total; }
        __syncthreads();
    }
}

// Position (0-based, inside the remaining sequence of n_rem items) of target j of global row g.  Next-item targets
// (SelectTargets without options, target_selection.py:41-53): j.  --shuffle_targets (random.shuffle of the remaining sequence,
// then the first n_targets of it): the j-th element of a uniform random k-subset in random order -- Floyd's algorithm on
// hashed draws, which every caller of the same (seed, g) reproduces (the popularity weight of a CCE row and its target are
// written by different lanes).  k <= 16 here (the launcher falls back beyond).
__device__ __forceinline__ int bb_target_pos(unsigned long long seed, int g, int j, int k, int n_rem, int shuffle) {
    if (!shuffle) return j;
    int chosen[16];
    for (int i = 0; i < k; ++i) {
        const int top = n_rem - k + i;                              // draw t uniformly in [0, top]
        const unsigned long long r = mix64(seed ^ mix64(0x7A67ull + ((unsigned long long)(unsigned)g << 8) + (unsigned)i));
        int t = (int)(r % (unsigned long long)(top + 1));
        for (int q = 0; q < i; ++q) if (chosen[q] == t) { t = top; break; }
        chosen[i] = t;
    }
    // the set is uniform; a uniform order on top of it: rotate by a hashed amount and pick (k <= 16: any fixed bijection of a
    // uniformly random set element works for j = 0, which is all the one-target heads read)
    const unsigned long long r2 = mix64(seed ^ mix64(0x0DDull + (unsigned long long)(unsigned)g));
    return chosen[(j + (int)(r2 % (unsigned long long)k)) % k];
}

// One wave per local row: X row (item id, + n_items + rating index with F == 2), length, pop**db of the first target.  Blocks
// past the rows write the targets (NT per row, -1 behind the last) and draw the S negatives.
__global__ void __launch_bounds__(64) bb_pack_kernel(const int* __restrict__ items, const int* __restrict__ rate,
                                                     const long long* __restrict__ off, const int* __restrict__ len_n,
                                                     const int* __restrict__ split, const int* __restrict__ rowuser,
                                                     const float* __restrict__ popdb, const double* __restrict__ cdf,
                                                     int n_items, int T, int F, int NT, int shuffle, int row_offset, int local_rows, int Bp,
                                                     int tgt_rows, int tgt_offset, int S, unsigned long long seed, int* __restrict__ X,
                                                     int* __restrict__ lengths, int* __restrict__ target, float* __restrict__ pop,
                                                     int* __restrict__ samples, const int* __restrict__ tgtpos) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b < Bp) {
        if (b >= local_rows) {                                     // padded rows: index 0, length 0, popularity 1
            for (int t = lane; t < T * F; t += 64) X[(size_t)b * T * F + t] = 0;
            if (lane == 0) { lengths[b] = 0; pop[b] = 1.0f; }
            return;
        }
        const int g = row_offset + b, l = split[g];
        const long long o = off[rowuser[g]];
        const int start = max(0, l - T), n_in = l - start;         // rnn_base.py:410: at most max_length items before l
        for (int t = lane; t < T; t += 64) {
            X[((size_t)b * T + t) * F] = t < n_in ? items[o + start + t] : 0;
            if (F == 2) X[((size_t)b * T + t) * F + 1] = t < n_in ? n_items + rate[o + start + t] : 0;   // rnn_base.py:637-642
        }
        if (lane == 0) {
            lengths[b] = n_in;
            const int n_rem = (len_n ? len_n[rowuser[g]] : (int)(off[rowuser[g] + 1] - o)) - l;
            const int k = min(n_rem, NT);
            const int p0 = tgtpos ? tgtpos[(size_t)g * NT] : bb_target_pos(seed, g, 0, k, n_rem, shuffle);   // (host-planned rows: given)
            pop[b] = popdb ? popdb[items[o + l + p0]] : 1.0f;                                              // rnn_one_hot.py:103
        }
        return;
    }
    const int e = (b - Bp) * 64 + lane;
    if (e < tgt_rows * NT) {                                       // targets (target_selection.py:41-53, rnn_base.py:407)
        const int g = tgt_offset + e / NT, j = e % NT;
        const long long o = off[rowuser[g]];
        const int l = split[g], n_rem = (len_n ? len_n[rowuser[g]] : (int)(off[rowuser[g] + 1] - o)) - l;
        const int k = min(n_rem, NT);
        if (tgtpos) { const int p = tgtpos[(size_t)g * NT + j]; target[e] = p >= 0 ? items[o + l + p] : -1; }
        else target[e] = j < k ? items[o + l + bb_target_pos(seed, g, j, k, n_rem, shuffle)] : -1;
    }
    const int si = e - ((tgt_rows * NT + 63) / 64) * 64;
    if (si >= 0 && si < S) {
        const unsigned long long r = mix64(seed ^ mix64(0xA5A5ull + (unsigned long long)si));
        if (!cdf) samples[si] = (int)(r % (unsigned long long)n_items);          // np.random.choice(n_items, S) (rnn_sampling.py:191)
        else {                                                     // bisect(cumsum, uniform(0, cumsum[-1])) (:159-163)
            const double x = (double)(r >> 11) * (1.0 / 9007199254740992.0) * cdf[n_items - 1];
            int lo = 0, hi = n_items;                              // first index with cdf[idx] > x
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] > x) hi = mid; else lo = mid + 1; }
            samples[si] = min(lo, n_items - 1);
        }
    }
}

// ---------------------------------------------------------------------------------------
// Sequence noise (sequence_noise.py:52-94): per pass and user, in this order -- dropout of items (a user left with fewer than
// two is skipped), swaps of neighbours (an item swaps at most once), swaps with an item a normal distance away, half-star
// rating perturbation.  One wave per user, the sequence staged in LDS: the compaction is a ballot prefix, the two swap
// passes are what the reference's loops are -- sequential -- run by lane 0 on LDS.  Draws are hashed from
// (seed, user, stage, index): a law, not the reference's Mersenne stream (tests/test_gpu_batch_builder.py).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bb_unif(unsigned long long seed, unsigned stage, unsigned user, unsigned i) {
    return (float)(key32(seed ^ mix64(0x401Eull + stage), user, i) >> 8) * (1.0f / 16777216.0f);      // [0, 1)
}
__global__ void __launch_bounds__(64) bb_noise_kernel(const int* __restrict__ items, const int* __restrict__ rate,
                                                      const long long* __restrict__ off, int* __restrict__ items_n,
                                                      int* __restrict__ rate_n, int* __restrict__ len_n, float p_drop, float p_swap,
                                                      float p_shuf, float shuf_std, float p_rate, unsigned long long seed) {
    extern __shared__ int nz[];                      // [L] items | [L] rating indices
    const unsigned u = blockIdx.x;
    const int lane = threadIdx.x;
    const long long o = off[u];
    const int L = (int)(off[u + 1] - o);
    int* it = nz; int* rt = nz + L;
    for (int i = lane; i < L; i += 64) { it[i] = items[o + i]; rt[i] = rate ? rate[o + i] : 0; }
    __syncthreads();
    int n = L;
    if (p_drop > 0.0f) {                             // keep an item while random() >= dropout; order preserved
        int w = 0;
        for (int c0 = 0; c0 < L; c0 += 64) {
            const int i = c0 + lane;
            const bool keep = i < L && bb_unif(seed, 0u, u, (unsigned)i) >= p_drop;
            const int a = i < L ? it[i] : 0, b = i < L ? rt[i] : 0;
            const unsigned long long m = __ballot(keep);
            __syncthreads();                         // (one wave: every lane has read its element before any lane writes)
            if (keep) { const int pos = w + __popcll(m & ((1ull << lane) - 1)); it[pos] = a; rt[pos] = b; }
            w += __popcll(m);
            __syncthreads();
        }
        n = w;
        if (n < 2) n = 0;                            // "if len(sequence) < 2: continue": the user yields nothing this pass
    }
    if (lane == 0 && n >= 2) {
        if (p_swap > 0.0f) {
            int i = 0;
            while (i < n - 1) {
                if (bb_unif(seed, 1u, u, (unsigned)i) < p_swap) {
                    const int a = it[i], b = rt[i];
                    it[i] = it[i + 1]; rt[i] = rt[i + 1]; it[i + 1] = a; rt[i + 1] = b;
                    i += 1;                          // don't allow to swap twice the same item
                }
                i += 1;
            }
        }
        if (p_shuf > 0.0f) {
            for (int i = 0; i < n; ++i) {
                if (bb_unif(seed, 2u, u, (unsigned)i) < p_shuf) {
                    // int(np.random.randn() * shuf_std) + i, clamped into the sequence (Box-Muller on two hashed uniforms)
                    const float u1 = fmaxf(bb_unif(seed, 3u, u, (unsigned)i), 5.9604645e-8f), u2 = bb_unif(seed, 4u, u, (unsigned)i);
                    const float z = sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
                    const int other = max(0, min(n - 1, (int)(z * shuf_std) + i));
                    const int a = it[i], b = rt[i];
                    it[i] = it[other]; rt[i] = rt[other]; it[other] = a; rt[other] = b;
                }
            }
        }
    }
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
        int r = rt[i];
        if (rate && p_rate > 0.0f && bb_unif(seed, 5u, u, (unsigned)i) < p_rate) {
            // +- half a star, clamped to [1, 5]: the one-hot index is 2 * rating - 1
            r = bb_unif(seed, 6u, u, (unsigned)i) < 0.5f ? min(9, r + 1) : max(1, r - 1);
        }
        items_n[o + i] = it[i];
        if (rate_n) rate_n[o + i] = r;
    }
    if (lane == 0) len_n[u] = n;
}

// ---------------------------------------------------------------------------------------
// C-ABI
// ---------------------------------------------------------------------------------------
extern "C" int sbr_dataset_create(const int32_t* items, const int64_t* offsets, int64_t n_users, int32_t n_items, void* stream,
                                  sbr_dataset** out) {
    CHECK_ARG(items && offsets && out && n_users >= 1 && n_items >= 1, "bad dataset arguments");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { sbr_set_error("no HIP device: the batch builder is device code"); return SBR_EHIP; }
    const int64_t nnz = offsets[n_users];
    CHECK_ARG(offsets[0] == 0 && nnz >= 0, "offsets must start at 0");
    sbr_dataset* d = new sbr_dataset();
    d->n_users = n_users; d->nnz = nnz; d->n_items = n_items; d->stream = (hipStream_t)stream;
    d->d_items = nullptr; d->d_off = nullptr; d->d_popdb = nullptr; d->d_cdf = nullptr; d->d_rate = nullptr; d->shuffle_targets = 0;
    d->d_items_n = d->d_rate_n = d->d_len_n = nullptr; d->noised = 0; d->max_len = 0;
    d->host_rows = 0; d->hr_targets = 1; d->hr_seed = 0; d->hr_pass = 0;
    d->d_hr_user = d->d_hr_split = d->d_hr_tgt = nullptr; d->cap_hru = d->cap_hrs = d->cap_hrt = 0;
    d->h_items.assign(items, items + nnz); d->h_off.assign(offsets, offsets + n_users + 1);
    d->d_seg_user = d->d_seg_k = d->d_seg_row0 = d->d_batch_begin = nullptr; d->cap_su = d->cap_sk = d->cap_sr = d->cap_bb = 0;
    d->d_split = d->d_rowuser = nullptr; d->cap_rows = 0; d->n_batches = 0; d->batch_size = 0;
    d->len.resize(n_users);
    for (int64_t u = 0; u < n_users; ++u) {
        d->len[u] = offsets[u + 1] - offsets[u];
        if (d->len[u] < 0) { delete d; sbr_set_error("offsets not monotone at user %lld", (long long)u); return SBR_EINVAL; }
        d->max_len = std::max(d->max_len, d->len[u]);
    }
    d->len_cur = d->len;
    for (int64_t i = 0; i < nnz; ++i)
        if (items[i] < 0 || items[i] >= n_items) { delete d; sbr_set_error("item id %d out of range [0,%d)", items[i], n_items); return SBR_EINVAL; }
    if (hipMalloc(&d->d_items, std::max<int64_t>(nnz, 1) * sizeof(int)) != hipSuccess ||
        hipMalloc(&d->d_off, (n_users + 1) * sizeof(long long)) != hipSuccess) {
        sbr_dataset_destroy(d); sbr_set_error("hipMalloc failed for the dataset"); return SBR_ENOMEM;
    }
    SBR_HIP(hipMemcpyAsync(d->d_items, items, nnz * sizeof(int), hipMemcpyHostToDevice, d->stream));
    SBR_HIP(hipMemcpyAsync(d->d_off, offsets, (n_users + 1) * sizeof(long long), hipMemcpyHostToDevice, d->stream));
    SBR_HIP(hipStreamSynchronize(d->stream));
    *out = d;
    return SBR_OK;
}

extern "C" int sbr_dataset_destroy(sbr_dataset* d) {
    if (!d) return SBR_OK;
    (void)hipFree(d->d_items); (void)hipFree(d->d_off); (void)hipFree(d->d_popdb); (void)hipFree(d->d_cdf); (void)hipFree(d->d_rate);
    (void)hipFree(d->d_seg_user); (void)hipFree(d->d_seg_k); (void)hipFree(d->d_seg_row0); (void)hipFree(d->d_batch_begin);
    (void)hipFree(d->d_split); (void)hipFree(d->d_rowuser);
    (void)hipFree(d->d_items_n); (void)hipFree(d->d_rate_n); (void)hipFree(d->d_len_n);
    (void)hipFree(d->d_hr_user); (void)hipFree(d->d_hr_split); (void)hipFree(d->d_hr_tgt);
    delete d;
    return SBR_OK;
}

extern "C" int sbr_dataset_set_options(sbr_dataset* d, const float* ratings, int shuffle_targets) {
    CHECK_ARG(d, "null dataset");
    d->shuffle_targets = shuffle_targets != 0;
    (void)hipFree(d->d_rate); d->d_rate = nullptr;
    if (ratings) {
        // rating one-hot index: round(rating * 2) - 1 on a scale of ten (rnn_base.py:590-605), Python 2's round (half away from zero)
        std::vector<int> idx((size_t)std::max<int64_t>(d->nnz, 1));
        for (int64_t i = 0; i < d->nnz; ++i) {
            const long v = (long)floor((double)ratings[i] * 2.0 + 0.5) - 1;
            idx[i] = (int)(((v % 10) + 10) % 10);
        }
        SBR_HIP(hipMalloc(&d->d_rate, idx.size() * sizeof(int)));
        SBR_HIP(hipMemcpyAsync(d->d_rate, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, d->stream));
        SBR_HIP(hipStreamSynchronize(d->stream));
    }
    return SBR_OK;
}

extern "C" int sbr_dataset_noise_pass(sbr_dataset* d, float dropout, float swap, float shuf, float shuf_std, float ratings_perturb,
                                      uint64_t seed) {
    CHECK_ARG(d, "null dataset");
    CHECK_ARG(dropout >= 0.0f && dropout < 1.0f && swap >= 0.0f && swap < 1.0f && ratings_perturb >= 0.0f && ratings_perturb < 1.0f &&
              shuf >= 0.0f && shuf <= 1.0f, "noise probabilities out of range (sequence_noise.py:46-51)");
    if (!(dropout > 0.0f || swap > 0.0f || shuf > 0.0f || (ratings_perturb > 0.0f && d->d_rate))) {
        d->noised = 0; d->len_cur = d->len;
        return SBR_OK;
    }
    const size_t lds = (size_t)std::max<int64_t>(d->max_len, 1) * 2 * sizeof(int);
    CHECK_ARG(lds <= 64 * 1024, "a sequence of %lld items does not fit the noise kernel's staging (8192)", (long long)d->max_len);
    if (!d->d_items_n) {
        SBR_HIP(hipMalloc(&d->d_items_n, std::max<int64_t>(d->nnz, 1) * sizeof(int)));
        SBR_HIP(hipMalloc(&d->d_len_n, d->n_users * sizeof(int)));
    }
    if (d->d_rate && !d->d_rate_n) SBR_HIP(hipMalloc(&d->d_rate_n, std::max<int64_t>(d->nnz, 1) * sizeof(int)));
    // batches of the previous pass may still be reading the previous copy: the stream orders the kernel behind them
    SBR_DYN_LDS(bb_noise_kernel, lds);
    bb_noise_kernel<<<(unsigned)d->n_users, 64, lds, d->stream>>>(d->d_items, d->d_rate, d->d_off, d->d_items_n, d->d_rate ? d->d_rate_n : nullptr,
                                                                   d->d_len_n, dropout, swap, shuf, shuf_std, ratings_perturb,
                                                                   seed * 0x9E3779B97F4A7C15ull + 0x5EEDull);
    SBR_LAUNCH(hipGetLastError());
    d->len_n32.resize(d->n_users);
    SBR_HIP(hipMemcpyAsync(d->len_n32.data(), d->d_len_n, d->n_users * sizeof(int), hipMemcpyDeviceToHost, d->stream));
    SBR_HIP(hipStreamSynchronize(d->stream));
    d->len_cur.resize(d->n_users);
    for (int64_t u = 0; u < d->n_users; ++u) d->len_cur[u] = d->len_n32[u];
    d->noised = 1;
    if (d->host_rows) {      // the host plans the rows of the pass: it needs this pass's copy of the sequences
        d->h_items_cur.resize((size_t)std::max<int64_t>(d->nnz, 1));
        SBR_HIP(hipMemcpy(d->h_items_cur.data(), d->d_items_n, d->nnz * sizeof(int), hipMemcpyDeviceToHost));
    }
    return SBR_OK;
}

extern "C" int sbr_dataset_set_target_bias(sbr_dataset* d, const float* keep_prob, int32_t n_targets, uint64_t seed) {
    CHECK_ARG(d, "null dataset");
    d->hr_pu.clear(); d->hr_ps.clear(); d->hr_pt.clear(); d->pend_user.clear(); d->pend_k.clear();
    if (!keep_prob) { d->host_rows = 0; d->keep_prob.clear(); return SBR_OK; }
    CHECK_ARG(n_targets >= 1 && n_targets <= 1024, "n_targets out of range");
    for (int i = 0; i < d->n_items; ++i) CHECK_ARG(keep_prob[i] >= 0.0f && keep_prob[i] <= 1.0f, "keep_prob[%d] outside [0, 1]", i);
    d->keep_prob.assign(keep_prob, keep_prob + d->n_items);
    d->host_rows = 1; d->hr_targets = n_targets; d->hr_seed = seed; d->hr_pass = 0;
    return SBR_OK;
}

extern "C" int sbr_dataset_current_sequences(sbr_dataset* d, int32_t* items, int32_t* rating_index, int32_t* lengths) {
    CHECK_ARG(d && items && lengths, "null argument");
    SBR_HIP(hipStreamSynchronize(d->stream));
    SBR_HIP(hipMemcpy(items, d->noised ? d->d_items_n : d->d_items, d->nnz * sizeof(int), hipMemcpyDeviceToHost));
    if (rating_index && d->d_rate)
        SBR_HIP(hipMemcpy(rating_index, d->noised ? d->d_rate_n : d->d_rate, d->nnz * sizeof(int), hipMemcpyDeviceToHost));
    for (int64_t u = 0; u < d->n_users; ++u) lengths[u] = (int32_t)d->len_cur[u];
    return SBR_OK;
}

extern "C" int sbr_dataset_set_tables(sbr_dataset* d, const float* pop_db, const double* cdf) {
    CHECK_ARG(d, "null dataset");
    (void)hipFree(d->d_popdb); (void)hipFree(d->d_cdf); d->d_popdb = nullptr; d->d_cdf = nullptr;
    if (pop_db) {
        SBR_HIP(hipMalloc(&d->d_popdb, d->n_items * sizeof(float)));
        SBR_HIP(hipMemcpyAsync(d->d_popdb, pop_db, d->n_items * sizeof(float), hipMemcpyHostToDevice, d->stream));
    }
    if (cdf) {
        for (int i = 1; i < d->n_items; ++i) CHECK_ARG(cdf[i] >= cdf[i - 1], "sample_cdf must be non-decreasing");
        CHECK_ARG(cdf[d->n_items - 1] > 0, "sample_cdf has no mass");
        SBR_HIP(hipMalloc(&d->d_cdf, d->n_items * sizeof(double)));
        SBR_HIP(hipMemcpyAsync(d->d_cdf, cdf, d->n_items * sizeof(double), hipMemcpyHostToDevice, d->stream));
    }
    SBR_HIP(hipStreamSynchronize(d->stream));
    return SBR_OK;
}

static int upload(int** dev, size_t* cap, const std::vector<int>& v, hipStream_t s) {
    if (v.size() > *cap) {
        (void)hipFree(*dev); *dev = nullptr;
        *cap = v.size() + v.size() / 2 + 64;
        SBR_HIP(hipMalloc(dev, *cap * sizeof(int)));
    }
    if (!v.empty()) SBR_HIP(hipMemcpyAsync(*dev, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice, s));
    return SBR_OK;
}

static int plan_pass_host_rows(sbr_dataset* d, const int32_t* order, int32_t B, int64_t* n_batches) {
    const int NT = d->hr_targets;
    if (d->batch_size != B) { d->hr_pu.clear(); d->hr_ps.clear(); d->hr_pt.clear(); d->batch_size = B; }
    if (d->noised) { d->hr_pu.clear(); d->hr_ps.clear(); d->hr_pt.clear(); }      // carried rows index the previous pass's noised copy: dropped
    std::vector<int> pu(B), ps(B), pt((size_t)B * NT);
    int np = (int)d->hr_pu.size();
    std::copy(d->hr_pu.begin(), d->hr_pu.end(), pu.begin());
    std::copy(d->hr_ps.begin(), d->hr_ps.end(), ps.begin());
    std::copy(d->hr_pt.begin(), d->hr_pt.end(), pt.begin());
    int64_t cap = B;
    for (int64_t u = 0; u < d->n_users; ++u) cap += std::max<int64_t>(0, std::min<int64_t>(B, d->len_cur[u] - 2));
    d->hr_user.resize(cap); d->hr_split.resize(cap); d->hr_tgt.resize((size_t)cap * NT);
    int64_t nr = 0, nb = 0;
    const int rc = sbr_plan_rows_host(d->noised ? d->h_items_cur.data() : d->h_items.data(), d->h_off.data(), d->len_cur.data(), order,
                                      d->n_users, B, NT, d->shuffle_targets, d->keep_prob.data(), d->hr_seed + 0x9E37ull * (++d->hr_pass),
                                      pu.data(), ps.data(), pt.data(), &np, cap, d->hr_user.data(), d->hr_split.data(), d->hr_tgt.data(), &nr, &nb);
    if (rc != SBR_OK) return rc;
    d->hr_pu.assign(pu.begin(), pu.begin() + np); d->hr_ps.assign(ps.begin(), ps.begin() + np); d->hr_pt.assign(pt.begin(), pt.begin() + (size_t)np * NT);
    d->hr_user.resize(nr); d->hr_split.resize(nr); d->hr_tgt.resize((size_t)nr * NT);
    // segments (runs of one user inside a batch), for the callers that follow the pass (sbr_dataset_plan_segments)
    d->seg_user.clear(); d->seg_k.clear(); d->seg_row0.clear(); d->seg_batch.clear();
    for (int64_t r = 0; r < nr; ++r) {
        const int b = (int)(r / B), g = (int)(r % B);
        if (g > 0 && d->hr_user[r] == d->hr_user[r - 1]) { d->seg_k.back() += 1; continue; }
        d->seg_user.push_back(d->hr_user[r]); d->seg_k.push_back(1); d->seg_row0.push_back(g); d->seg_batch.push_back(b);
    }
    d->batch_begin.assign(nb + 1, 0);
    SBR_HIP(hipStreamSynchronize(d->stream));      // the previous plan may still be read by batches in flight
    int r;
    if ((r = upload(&d->d_hr_user, &d->cap_hru, d->hr_user, d->stream)) != SBR_OK) return r;
    if ((r = upload(&d->d_hr_split, &d->cap_hrs, d->hr_split, d->stream)) != SBR_OK) return r;
    if ((r = upload(&d->d_hr_tgt, &d->cap_hrt, d->hr_tgt, d->stream)) != SBR_OK) return r;
    SBR_HIP(hipStreamSynchronize(d->stream));
    d->n_batches = nb; *n_batches = nb;
    return SBR_OK;
}

extern "C" int sbr_dataset_plan_pass(sbr_dataset* d, const int32_t* order, int32_t B, int64_t* n_batches) {
    CHECK_ARG(d && n_batches && B >= 1, "bad plan arguments");
    if (d->host_rows) return plan_pass_host_rows(d, order, B, n_batches);
    if (d->batch_size != B) { d->pend_user.clear(); d->pend_k.clear(); d->batch_size = B; }
    std::vector<int> pu(B), pk(B);
    int np = (int)d->pend_user.size();
    std::copy(d->pend_user.begin(), d->pend_user.end(), pu.begin());
    std::copy(d->pend_k.begin(), d->pend_k.end(), pk.begin());
    const size_t cap = (size_t)d->n_users + np + 1;
    d->seg_user.resize(cap); d->seg_k.resize(cap); d->seg_row0.resize(cap); d->seg_batch.resize(cap);
    int64_t ns = 0, nb = 0;
    const int rc = sbr_plan_pass_host(d->len_cur.data(), order, d->n_users, B, pu.data(), pk.data(), &np, d->seg_user.data(),
                                      d->seg_k.data(), d->seg_row0.data(), d->seg_batch.data(), &ns, &nb);
    if (rc != SBR_OK) return rc;
    d->pend_user.assign(pu.begin(), pu.begin() + np); d->pend_k.assign(pk.begin(), pk.begin() + np);
    d->seg_user.resize(ns); d->seg_k.resize(ns); d->seg_row0.resize(ns); d->seg_batch.resize(ns);
    d->batch_begin.assign(nb + 1, 0);
    for (int64_t s = 0; s < ns; ++s) d->batch_begin[d->seg_batch[s] + 1] = (int)s + 1;
    for (int64_t b = 1; b <= nb; ++b) d->batch_begin[b] = std::max(d->batch_begin[b], d->batch_begin[b - 1]);
    // the previous plan may still be read by batches in flight on the stream: order the uploads behind them
    SBR_HIP(hipStreamSynchronize(d->stream));
    int r;
    if ((r = upload(&d->d_seg_user, &d->cap_su, d->seg_user, d->stream)) != SBR_OK) return r;
    if ((r = upload(&d->d_seg_k, &d->cap_sk, d->seg_k, d->stream)) != SBR_OK) return r;
    if ((r = upload(&d->d_seg_row0, &d->cap_sr, d->seg_row0, d->stream)) != SBR_OK) return r;
    if ((r = upload(&d->d_batch_begin, &d->cap_bb, d->batch_begin, d->stream)) != SBR_OK) return r;
    SBR_HIP(hipStreamSynchronize(d->stream));
    d->n_batches = nb; *n_batches = nb;
    return SBR_OK;
}

extern "C" int sbr_dataset_plan_segments(sbr_dataset* d, int64_t* n_segments, const int32_t** seg_user, const int32_t** seg_k,
                                         const int32_t** seg_row0, const int32_t** seg_batch) {
    CHECK_ARG(d && n_segments, "null argument");
    *n_segments = (int64_t)d->seg_user.size();
    if (seg_user) *seg_user = d->seg_user.data();
    if (seg_k) *seg_k = d->seg_k.data();
    if (seg_row0) *seg_row0 = d->seg_row0.data();
    if (seg_batch) *seg_batch = d->seg_batch.data();
    return SBR_OK;
}

extern "C" int sbr_build_batch(sbr_handle* h, sbr_dataset* d, int64_t batch, uint64_t seed) {
    CHECK_ARG(h && d, "null handle / dataset");
    const Layout& y = h->lay;
    CHECK_ARG(y.F == 1 || (y.F == 2 && d->d_rate && y.cfg.input_size == y.N + 10),
              "the native batch builder covers the item index and, with ratings attached (sbr_dataset_set_options), the rating index");
    CHECK_ARG(d->n_items == y.N && (y.cfg.input_size == y.N || y.F == 2), "dataset has %d items, the model %d", d->n_items, y.N);
    CHECK_ARG(!d->shuffle_targets || y.NT <= 16 || d->host_rows, "shuffled targets: at most 16 targets per row on the device");
    CHECK_ARG(d->batch_size == y.Bg, "the pass was planned for batches of %d rows, the model's global batch is %d", d->batch_size, y.Bg);
    CHECK_ARG(batch >= 0 && batch < d->n_batches, "batch %lld outside the planned pass [0,%lld)", (long long)batch, (long long)d->n_batches);
    CHECK_ARG(d->stream == h->stream, "dataset and engine must share one stream");
    // The build runs on a stream of its own into the batch set the step in flight does not read, so that the host can queue
    // batch i+1 while step i runs and the device packs it beside the step (two small launches, 20 - 30 us that used to sit between
    // two steps; tools/bench_train_loop.py).  Order: (1) the set it overwrites was read by the step before the one in flight --
    // every stream of a completed step is joined into the main stream by sbr_apply_update, so any main-stream record made DURING
    // the step in flight is behind it: the step records one anyway (ev_lg, in front of the BPTT chain), no extra record on the
    // main stream; (2) without such a record (first batches, evaluation between steps, a step abandoned half way) the build
    // waits for a fresh record on each of the engine's streams; (3) the main stream waits for the build (long complete by then).
    // The dataset's own arrays change only inside calls that synchronise d->stream first and last, i.e. behind (3).
    hipStream_t s = h->s_bb;
    const int set = h->bb_set ^ 1;
    if (h->train_fwd_open) h->bb_slow = 2;
    if (h->bb_unread && h->bb_slow == 0) h->bb_slow = 1;      // two builds and no forward between them: whoever read the first did it outside a step
    if (h->bb_slow == 0 && h->ev_lg_rec && h->lg_seq > h->set_use[set]) {
        SBR_HIP(hipStreamWaitEvent(s, h->ev_lg_rec, 0));
    } else {
        hipStream_t all[4] = {h->stream, h->side, h->side2, h->side3};
        for (hipStream_t q : all) { SBR_HIP(hipEventRecord(h->ev_bbw, q)); SBR_HIP(hipStreamWaitEvent(s, h->ev_bbw, 0)); }
        if (h->bb_slow > 0) --h->bb_slow;
    }
    if ((size_t)y.Bg > d->cap_rows) {
        (void)hipFree(d->d_split); (void)hipFree(d->d_rowuser); d->d_split = d->d_rowuser = nullptr;
        SBR_HIP(hipMalloc(&d->d_split, (size_t)y.Bg * sizeof(int)));
        SBR_HIP(hipMalloc(&d->d_rowuser, (size_t)y.Bg * sizeof(int)));
        d->cap_rows = y.Bg;
    }
    const unsigned long long sd = seed ^ (0x9E3779B97F4A7C15ull * (unsigned long long)(batch + 1));
    const int* len_n = d->noised ? d->d_len_n : nullptr;      // this pass's noised copy of the sequences (sbr_dataset_noise_pass)
    const int* src_items = d->noised ? d->d_items_n : d->d_items;
    const int* src_rate = d->noised && d->d_rate ? d->d_rate_n : d->d_rate;
    const int *split = d->d_split, *rowuser = d->d_rowuser, *tgtpos = nullptr;
    if (d->host_rows) {      // rows planned on the host (target bias): user, split point and target positions are given
        CHECK_ARG(d->hr_targets == y.NT, "the rows were planned for %d targets, the model has %d", d->hr_targets, y.NT);
        split = d->d_hr_split + batch * y.Bg; rowuser = d->d_hr_user + batch * y.Bg; tgtpos = d->d_hr_tgt + batch * (int64_t)y.Bg * y.NT;
    } else {
        const int sb = d->batch_begin[batch], se = d->batch_begin[batch + 1];
        bb_split_kernel<<<se - sb, 256, 0, s>>>(d->d_off, len_n, d->d_seg_user, d->d_seg_k, d->d_seg_row0, sb, sd, d->d_split, d->d_rowuser);
        SBR_LAUNCH(hipGetLastError());
    }
    int *bX = (int*)h->A(set ? y.a_X2 : y.a_X), *blen = (int*)h->A(set ? y.a_len2 : y.a_len), *btgt = (int*)h->A(set ? y.a_tgt2 : y.a_tgt);
    int* bsmp = (int*)h->A(set ? y.a_smp2 : y.a_smp);
    float* bpop = h->A(set ? y.a_pop2 : y.a_pop);
    const bool sampled = y.S > 0;
    const int tgt_rows = sampled ? y.Bg : y.B, tgt_offset = sampled ? 0 : y.cfg.row_offset;
    const int extra = (tgt_rows * y.NT + 63) / 64 + (y.S + 63) / 64;
    bb_pack_kernel<<<y.Bp + extra, 64, 0, s>>>(src_items, src_rate, d->d_off, len_n, split, rowuser, d->d_popdb, d->d_cdf, d->n_items, y.T,
                                                y.F, y.NT, d->shuffle_targets, y.cfg.row_offset, y.B, y.Bp, tgt_rows, tgt_offset, y.S, sd, bX,
                                                blen, btgt, bpop, bsmp, tgtpos);
    SBR_LAUNCH(hipGetLastError());
    SBR_HIP(hipEventRecord(h->ev_bb, s));
    SBR_HIP(hipStreamWaitEvent(h->stream, h->ev_bb, 0));
    h->bX = bX; h->blen = blen; h->btgt = btgt; h->bsmp = bsmp; h->bpop = bpop;
    h->bb_set = set; h->bb_unread = true;
    h->n_rows = y.B; h->have_batch = true; h->fwd_done = false;
    return SBR_OK;
}
