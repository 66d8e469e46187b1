// HBM-bound and small kernels of the hot path (gfx950, wave64): embedding-bag gather (K1) and
// its scatter-add gradient (K6), softmax / cross-entropy (K8), sampled heads (K10-K12),
// optimizers (K13), test-path exclusion + top-k (K14).
#include "sbr_common.h"
#include "sbr_rec_p.h"
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------
// block-wide reductions (256 threads = 4 waves)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s = fmaxf(s, red[w]);
    return s;
}

// ---------------------------------------------------------------------------------------
// K1 embedding-bag gather: xt[t][b][:] = sum_f W_in[X[b][t][f]][:] + bias
// (sparse_lstm.py:368 / :755 / :1111).  One 16-byte piece per thread: a row of G*Hp floats is
// read by G*Hp/4 consecutive lanes (coalesced 16 B/lane), written the same way.
// ---------------------------------------------------------------------------------------
__global__ void gather_xt_kernel(const f32x4* __restrict__ Win, const f32x4* __restrict__ bias,
                                 const int* __restrict__ X, f32x4* __restrict__ xt, int T, int Bp, int F, int R4) {
    const size_t total = (size_t)T * Bp * R4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int f4 = (int)(i % R4);
        const size_t pos = i / R4;
        const int b = (int)(pos % Bp), t = (int)(pos / Bp);
        f32x4 v = bias[f4];
        const int* ids = X + ((size_t)b * T + t) * F;
        for (int f = 0; f < F; ++f) v += Win[(size_t)ids[f] * R4 + f4];
        xt[i] = v;
    }
}

__global__ void gather_concat_kernel(const f32x4* __restrict__ W, const int* __restrict__ X, f32x4* __restrict__ out, int T,
                                     int Bp, int F, int E4) {
    const size_t total = (size_t)T * Bp * F * E4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int e4 = (int)(i % E4);
        const size_t pf = i / E4;                      // (t*Bp + b)*F + f
        const int f = (int)(pf % F);
        const size_t pos = pf / F;
        const int b = (int)(pos % Bp), t = (int)(pos / Bp);
        out[i] = W[(size_t)X[((size_t)b * T + t) * F + f] * E4 + e4];
    }
}
hipError_t launch_gather_concat(hipStream_t s, const float* Wemb, const int* X, float* out, int T, int Bp, int F, int Ep) {
    const size_t total = (size_t)T * Bp * F * (Ep / 4);
    const int grid = (int)min((size_t)256 * 16, (total + 255) / 256);
    gather_concat_kernel<<<grid, 256, 0, s>>>((const f32x4*)Wemb, X, (f32x4*)out, T, Bp, F, Ep / 4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// --r_bi helpers.  The backwards layer of a level (Lasagne go_backwards=True over the left-aligned, masked sequence:
// the padded steps come first and copy hid_init, then the valid steps in reverse) is run as an ordinary forward scan
// over the row's valid steps REVERSED IN PLACE: rev(t) = len-1-t for t < len, t otherwise.  State after scan step s of
// the reversed row = the backwards layer's output at time len-1-s; its final state = the output at time 0, which is what
// only_return_final takes (sparse_lstm.py:485-486).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int rev_t(int t, int len) { return t < len ? len - 1 - t : t; }

__global__ void rev_rows_int_kernel(const int* __restrict__ X, const int* __restrict__ len, int* __restrict__ Xr, int T, int Bp, int F) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp * T * F) return;
    const int f = i % F, t = (i / F) % T, b = i / (F * T);
    Xr[i] = X[((size_t)b * T + rev_t(t, len[b])) * F + f];
}
hipError_t launch_rev_rows_int(hipStream_t s, const int* X, const int* len, int* Xr, int T, int Bp, int F) {
    const int n = Bp * T * F;
    rev_rows_int_kernel<<<(n + 255) / 256, 256, 0, s>>>(X, len, Xr, T, Bp, F);
    return hipGetLastError();
}

__global__ void rev_rows_kernel(const f32x4* __restrict__ src, const int* __restrict__ len, f32x4* __restrict__ dst, int T, int Bp, int W4) {
    const size_t total = (size_t)T * Bp * W4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W4);
        const size_t pos = i / W4;
        const int b = (int)(pos % Bp), t = (int)(pos / Bp);
        dst[i] = src[((size_t)rev_t(t, len[b]) * Bp + b) * W4 + w];
    }
}
hipError_t launch_rev_rows(hipStream_t s, const float* src, const int* len, float* dst, int T, int Bp, int W) {
    const size_t total = (size_t)T * Bp * (W / 4);
    rev_rows_kernel<<<(int)min((size_t)4096, (total + 255) / 256), 256, 0, s>>>((const f32x4*)src, len, (f32x4*)dst, T, Bp, W / 4);
    return hipGetLastError();
}

__global__ void cat_outputs_kernel(const f32x4* __restrict__ hf, const f32x4* __restrict__ hb, const int* __restrict__ len,
                                   f32x4* __restrict__ cat, int T, int Bp, int H4) {
    const size_t total = (size_t)T * Bp * 2 * H4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % (2 * H4));
        const size_t pos = i / (2 * H4);
        const int b = (int)(pos % Bp), t = (int)(pos / Bp);
        cat[i] = w < H4 ? hf[((size_t)(t + 1) * Bp + b) * H4 + w]
                        : hb[((size_t)(rev_t(t, len[b]) + 1) * Bp + b) * H4 + (w - H4)];
    }
}
hipError_t launch_cat_outputs(hipStream_t s, const float* hs_f, const float* hs_b, const int* len, float* cat, int T, int Bp, int Hp) {
    const size_t total = (size_t)T * Bp * 2 * (Hp / 4);
    cat_outputs_kernel<<<(int)min((size_t)4096, (total + 255) / 256), 256, 0, s>>>((const f32x4*)hs_f, (const f32x4*)hs_b, len,
                                                                                 (f32x4*)cat, T, Bp, Hp / 4);
    return hipGetLastError();
}

__global__ void hcat_kernel(const float* __restrict__ hf, const float* __restrict__ hb, float* __restrict__ out, int Bp, int Hp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Bp * 2 * Hp) return;
    const int w = i % (2 * Hp), b = i / (2 * Hp);
    out[i] = w < Hp ? hf[(size_t)b * Hp + w] : hb[(size_t)b * Hp + w - Hp];
}
hipError_t launch_hcat(hipStream_t s, const float* hf, const float* hb, float* out, int Bp, int Hp) {
    hcat_kernel<<<(Bp * 2 * Hp + 255) / 256, 256, 0, s>>>(hf, hb, out, Bp, Hp);
    return hipGetLastError();
}
__global__ void split_cols_kernel(const float* __restrict__ src, float* __restrict__ a, float* __restrict__ b, int rows, int Hp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 2 * Hp) return;
    const int w = i % (2 * Hp), r = i / (2 * Hp);
    if (w < Hp) a[(size_t)r * Hp + w] = src[i]; else b[(size_t)r * Hp + w - Hp] = src[i];
}
hipError_t launch_split_cols(hipStream_t s, const float* src, float* a, float* b, int rows, int Hp) {
    split_cols_kernel<<<(rows * 2 * Hp + 255) / 256, 256, 0, s>>>(src, a, b, rows, Hp);
    return hipGetLastError();
}

__global__ void uncat_kernel(const float* __restrict__ f, const float* __restrict__ r, const int* __restrict__ len,
                             float* __restrict__ out_f, float* __restrict__ out_b, int T, int Bp, int W, int Hp) {
    const size_t total = (size_t)T * Bp * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % W);
        const size_t pos = i / W;
        const int b = (int)(pos % Bp), t = (int)(pos / Bp);
        const int tr = rev_t(t, len[b]);
        const float d = f[i] + r[((size_t)tr * Bp + b) * W + w];                   // gradient wrt the level's input at time t
        if (!out_b) out_f[i] = d;
        else if (w < Hp) out_f[((size_t)t * Bp + b) * Hp + w] = d;                     // forward direction: its own time
        else out_b[((size_t)tr * Bp + b) * Hp + (w - Hp)] = d;                         // backwards direction: reversed time
    }
}
hipError_t launch_uncat(hipStream_t s, const float* f, const float* r, const int* len, float* out_f, float* out_b, int T, int Bp,
                        int W, int Hp) {
    const size_t total = (size_t)T * Bp * W;
    uncat_kernel<<<(int)min((size_t)4096, (total + 255) / 256), 256, 0, s>>>(f, r, len, out_f, out_b, T, Bp, W, Hp);
    return hipGetLastError();
}

hipError_t launch_gather_xt(hipStream_t s, const float* Win, const float* bias, const int* X, float* xt, int T, int Bp,
                            int F, int GHp, int) {
    const int R4 = GHp / 4;
    const size_t total = (size_t)T * Bp * R4;
    const int grid = (int)min((size_t)256 * 16, (total + 255) / 256);
    gather_xt_kernel<<<grid, 256, 0, s>>>((const f32x4*)Win, (const f32x4*)bias, X, (f32x4*)xt, T, Bp, F, R4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K6 scatter-add: dWin[X[b][t][f]][:] += dxt[t][b][:] for t < len[b]; duplicates accumulate
// (gradient of the advanced-indexing gather [3P]).
//
// Item ids follow a Zipf law, so per-element float atomics serialise in L2 on the popular rows
// (772 us at B=256, T=200, N=3706).  Instead the (position, id) pairs are counting-sorted by id
// (3 tiny integer kernels that depend only on the batch, run on a side stream under the BPTT
// chain), and one wave per chunk of 32 sorted entries sums the dxt rows of equal id in registers:
// rows are read once, coalesced, 16 B/lane; a row of dWin that lies entirely inside one chunk is
// written with plain stores, only segments that straddle a chunk boundary use atomics.
// ---------------------------------------------------------------------------------------
// Zipf ids: thousands of entries share the hottest id, so per-entry global atomics on cnt[]/cur[]
// serialise in one L2 channel (69 us each, and they slow every kernel running beside them).  For
// catalogues whose counters fit LDS (n_ids <= SCAT_LDS_IDS) each workgroup histograms its slice in
// LDS and issues ONE global atomic per distinct id it saw.
#define SCAT_LDS_IDS 36864    // 144 KB of LDS counters: catalogues up to ~37 k ids (C4's 26 744) take the LDS path
#define SCAT_BLOCK 1024
// Time-chunked keys (tch > 0): key = (t / tch) * ids_per_chunk + id, so that the entries of one chunk of time steps are
// contiguous in the sorted order and sorted by id inside it -- the scatter-add of a chunk can then run as soon as the BPTT
// chain has left that chunk (sbr_backward_recurrent, "tail overlap").  n_ids is the size of the KEY space.
__device__ __forceinline__ int scat_key(int id, int t, const SbrTChunks& tc, int ids_per_chunk) {
    if (tc.n <= 1) return id;
    int c = 0;
#pragma unroll
    for (int k = 1; k < SBR_TCHUNKS_MAX; ++k) c += (k < tc.n && t >= tc.lo[k]) ? 1 : 0;
    return c * ids_per_chunk + id;
}

__global__ void __launch_bounds__(SCAT_BLOCK) scat_count_lds_kernel(const int* __restrict__ X, const int* __restrict__ len,
                                                                    int T, int Bp, int F, int n_ids, int per_block,
                                                                    int* __restrict__ cnt, SbrTChunks tch, int ipc) {
    extern __shared__ int hist[];
    for (int i = threadIdx.x; i < n_ids; i += SCAT_BLOCK) hist[i] = 0;
    __syncthreads();
    const int total = T * Bp * F;
    const int lo = blockIdx.x * per_block, hi = min(total, lo + per_block);
    for (int i = lo + threadIdx.x; i < hi; i += SCAT_BLOCK) {
        const int f = i % F, pos = i / F, b = pos % Bp, t = pos / Bp;
        if (t < len[b]) atomicAdd(&hist[scat_key(X[((size_t)b * T + t) * F + f], t, tch, ipc)], 1);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_ids; i += SCAT_BLOCK) { const int c = hist[i]; if (c) atomicAdd(&cnt[i], c); }
}

__global__ void __launch_bounds__(SCAT_BLOCK) scat_fill_lds_kernel(const int* __restrict__ X, const int* __restrict__ len,
                                                                   int T, int Bp, int F, int n_ids, int per_block,
                                                                   int* __restrict__ cur, int* __restrict__ sid,
                                                                   int* __restrict__ spos, int concat, SbrTChunks tch, int ipc) {
    extern __shared__ int hist[];          // [n_ids] counts, then cursors
    for (int i = threadIdx.x; i < n_ids; i += SCAT_BLOCK) hist[i] = 0;
    __syncthreads();
    const int total = T * Bp * F;
    const int lo = blockIdx.x * per_block, hi = min(total, lo + per_block);
    for (int i = lo + threadIdx.x; i < hi; i += SCAT_BLOCK) {
        const int f = i % F, pos = i / F, b = pos % Bp, t = pos / Bp;
        if (t < len[b]) atomicAdd(&hist[scat_key(X[((size_t)b * T + t) * F + f], t, tch, ipc)], 1);
    }
    __syncthreads();
    // reserve a contiguous slot range per id with one global atomic; hist[id] becomes the block's cursor
    for (int i = threadIdx.x; i < n_ids; i += SCAT_BLOCK) { const int c = hist[i]; if (c) hist[i] = atomicAdd(&cur[i], c); }
    __syncthreads();
    for (int i = lo + threadIdx.x; i < hi; i += SCAT_BLOCK) {
        const int f = i % F, pos = i / F, b = pos % Bp, t = pos / Bp;
        if (t < len[b]) {
            const int id = scat_key(X[((size_t)b * T + t) * F + f], t, tch, ipc);
            const int slot = atomicAdd(&hist[id], 1);
            sid[slot] = id; spos[slot] = concat ? i : pos;
        }
    }
}

__global__ void scat_count_kernel(const int* __restrict__ X, const int* __restrict__ len, int T, int Bp, int F,
                                  int* __restrict__ cnt) {
    const int total = T * Bp * F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = i % F, pos = i / F, b = pos % Bp, t = pos / Bp;
        if (t < len[b]) atomicAdd(&cnt[X[((size_t)b * T + t) * F + f]], 1);
    }
}

// exclusive scan of cnt[0..n) -> offs[0..n], offs[n] = total; cur = copy of offs (fill cursors); one workgroup.
// LDS (n <= SCAT_LDS_IDS, e.g. the ~30 k keys of the time-chunked sort): the counters are staged in LDS with coalesced loads,
// every thread sums a contiguous run of ceil(n / 1024) of them, the 1024 run sums are scanned across the block once, every
// thread turns its run into prefixes in place, and the result leaves with coalesced stores.  Otherwise (large catalogues):
// one block-wide scan per 1024 counters, coalesced accesses throughout.
template <bool LDS>
__global__ void __launch_bounds__(1024) scat_scan_kernel(int* __restrict__ cnt, int n, int* __restrict__ offs,
                                                         int* __restrict__ cur) {
    extern __shared__ int stage[];
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!LDS) {
        if (threadIdx.x == 0) carry_s = 0;
        __syncthreads();
        for (int base = 0; base < n; base += 1024) {
            const int i = base + threadIdx.x;
            const int v = i < n ? cnt[i] : 0;
            int incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
            if (lane == 63) wsum[wave] = incl;
            __syncthreads();
            int woff = 0;
            for (int w = 0; w < wave; ++w) woff += wsum[w];
            const int carry = carry_s;
            const int excl = carry + woff + incl - v;
            if (i < n) { offs[i] = excl; cur[i] = excl; }
            __syncthreads();
            if (threadIdx.x == 1023) carry_s = carry + woff + incl;
            __syncthreads();
        }
        if (threadIdx.x == 0) offs[n] = carry_s;
        return;
    }
    for (int i = threadIdx.x; i < n; i += 1024) { stage[i] = cnt[i]; cnt[i] = 0; }     // (zero again: the next sort of this shape needs no memset)
    __syncthreads();
    const int per = (n + 1023) / 1024;
    const int lo = min(n, (int)threadIdx.x * per), hi = min(n, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += stage[i];
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + incl - sum;                      // exclusive prefix of this thread's run
    for (int i = lo; i < hi; ++i) { const int c = stage[i]; stage[i] = run; run += c; }
    if (threadIdx.x == 1023) offs[n] = woff + incl;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 1024) { const int v = stage[i]; offs[i] = v; cur[i] = v; }
}

// Large catalogues (the counters do not fit LDS): three small launches instead of one workgroup walking the whole array
// (1 M counters took it 2 ms: a thousand rounds of three barriers) -- per 4096-counter block the local exclusive prefixes
// and the block's total; the totals' exclusive scan (one workgroup; parked in `cnt`, which nobody reads any more); the add.
constexpr int SCAN_BLK = 4096;
__global__ void __launch_bounds__(1024) scat_scan_local_kernel(const int* __restrict__ cnt, int n, int* __restrict__ offs,
                                                               int* __restrict__ bsum) {
    __shared__ int wsum[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = blockIdx.x * SCAN_BLK + threadIdx.x * 4;
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i0 + k < n ? cnt[i0 + k] : 0;
    const int mine = v[0] + v[1] + v[2] + v[3];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int run = woff + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (i0 + k < n) offs[i0 + k] = run; run += v[k]; }
    if (threadIdx.x == 1023) bsum[blockIdx.x] = run;
}
__global__ void __launch_bounds__(1024) scat_scan_totals_kernel(const int* __restrict__ bsum, int nb, int* __restrict__ bpre,
                                                                int* __restrict__ total) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nb; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nb ? bsum[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const int carry = carry_s;
        if (i < nb) bpre[i] = carry + woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}
__global__ void __launch_bounds__(1024) scat_scan_add_kernel(const int* __restrict__ bpre, int n, int* __restrict__ offs,
                                                             int* __restrict__ cur) {
    const int carry = bpre[blockIdx.x];
    const int i0 = blockIdx.x * SCAN_BLK + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) { const int v = offs[i0 + k] + carry; offs[i0 + k] = v; cur[i0 + k] = v; }
}

__global__ void scat_fill_kernel(const int* __restrict__ X, const int* __restrict__ len, int T, int Bp, int F,
                                 int* __restrict__ cur, int* __restrict__ sid, int* __restrict__ spos, int concat) {
    const int total = T * Bp * F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = i % F, pos = i / F, b = pos % Bp, t = pos / Bp;
        if (t < len[b]) {
            const int id = X[((size_t)b * T + t) * F + f];
            const int slot = atomicAdd(&cur[id], 1);
            sid[slot] = id; spos[slot] = concat ? i : pos;
        }
    }
}

// The counting sort of catalogues beyond the LDS histogram, round 6: one global atomic per DISTINCT id of a workgroup's 1 024 entries
// instead of one per entry.  The two kernels above send every entry's atomic to cnt[id] / cur[id]; atomics on one address serialise at
// its L2 channel, and a step's hottest id holds thousands of its 51 200 entries (C5's batch: 3 612) -- ~50 us per kernel, all of it the
// hot counters, beside the head whose loads wait behind them (profiles/round6_variants.txt, calls s3 / a2).  Here a workgroup first
// groups its entries in an LDS hash table (id -> count; LDS atomics), then adds each distinct id's count once; the fill takes a base slot
// per distinct id the same way and every entry adds its rank inside the workgroup.
#define SCAT_AGG_SLOTS 2048
__device__ __forceinline__ int scat_agg_insert(int* __restrict__ hkey, int* __restrict__ hcnt, int id, int& rank) {
    unsigned h = ((unsigned)id * 2654435761u) >> 21;                  // 11 bits
    for (;;) {
        const int prev = atomicCAS(&hkey[h], -1, id);
        if (prev == -1 || prev == id) { rank = atomicAdd(&hcnt[h], 1); return (int)h; }
        h = (h + 1) & (SCAT_AGG_SLOTS - 1);
    }
}
__global__ void __launch_bounds__(1024) scat_count_agg_kernel(const int* __restrict__ X, const int* __restrict__ len, int T, int Bp, int F,
                                                              int* __restrict__ cnt) {
    __shared__ int hkey[SCAT_AGG_SLOTS], hcnt[SCAT_AGG_SLOTS];
    for (int k = threadIdx.x; k < SCAT_AGG_SLOTS; k += 1024) { hkey[k] = -1; hcnt[k] = 0; }
    __syncthreads();
    const int total = T * Bp * F, i = blockIdx.x * 1024 + threadIdx.x;
    if (i < total) {
        const int f = i % F, pos = i / F, b = pos % Bp, t = pos / Bp;
        if (t < len[b]) { int rank; (void)scat_agg_insert(hkey, hcnt, X[((size_t)b * T + t) * F + f], rank); }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SCAT_AGG_SLOTS; k += 1024) if (hkey[k] >= 0) atomicAdd(&cnt[hkey[k]], hcnt[k]);
}
__global__ void __launch_bounds__(1024) scat_fill_agg_kernel(const int* __restrict__ X, const int* __restrict__ len, int T, int Bp, int F,
                                                             int* __restrict__ cur, int* __restrict__ sid, int* __restrict__ spos, int concat) {
    __shared__ int hkey[SCAT_AGG_SLOTS], hcnt[SCAT_AGG_SLOTS], hbase[SCAT_AGG_SLOTS];
    for (int k = threadIdx.x; k < SCAT_AGG_SLOTS; k += 1024) { hkey[k] = -1; hcnt[k] = 0; }
    __syncthreads();
    const int total = T * Bp * F, i = blockIdx.x * 1024 + threadIdx.x;
    int id = -1, pos = 0, rank = 0, h = 0;
    if (i < total) {
        const int f = i % F; pos = i / F;
        const int b = pos % Bp, t = pos / Bp;
        if (t < len[b]) { id = X[((size_t)b * T + t) * F + f]; h = scat_agg_insert(hkey, hcnt, id, rank); }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < SCAT_AGG_SLOTS; k += 1024) if (hkey[k] >= 0) hbase[k] = atomicAdd(&cur[hkey[k]], hcnt[k]);
    __syncthreads();
    if (id >= 0) { const int slot = hbase[h] + rank; sid[slot] = id; spos[slot] = concat ? i : pos; }
}

int sbr_scatter_lds_ids() { return SCAT_LDS_IDS; }

hipError_t launch_scatter_sort(hipStream_t s, const int* X, const int* len, int T, int Bp, int F, int n_ids, int* cnt,
                               int* offs, int* cur, int* sid, int* spos, int concat, int tch, int n_tchunks, const SbrTChunks* bounds,
                               int* cnt_zero_n) {
    const int ipc = n_ids;                          // ids per time chunk
    SbrTChunks tc = bounds ? *bounds : sbr_uniform_tchunks(tch, n_tchunks);
    if (tch > 0) {
        n_ids *= n_tchunks;                         // key space
        if (n_ids > SCAT_LDS_IDS || tc.n != n_tchunks || tc.n > SBR_TCHUNKS_MAX || tc.lo[0] != 0 || tc.lo[tc.n] < T) return hipErrorInvalidValue;
    }
    // cnt_zero_n: how many leading counters the previous user of `cnt` left at zero (the LDS-path scan clears what it reads)
    if (!(cnt_zero_n && *cnt_zero_n >= n_ids)) {
        hipError_t e = hipMemsetAsync(cnt, 0, (size_t)n_ids * sizeof(int), s);
        if (e != hipSuccess) return e;
    }
    if (cnt_zero_n) *cnt_zero_n = n_ids <= SCAT_LDS_IDS ? n_ids : 0;
    const int total = T * Bp * F;
    if (n_ids <= SCAT_LDS_IDS) {
        const int per_block = 4096;
        const int grid = (total + per_block - 1) / per_block;
        const size_t lds = (size_t)n_ids * sizeof(int);
        SBR_DYN_LDS(scat_count_lds_kernel, lds);
        SBR_DYN_LDS(scat_fill_lds_kernel, lds);
        scat_count_lds_kernel<<<grid, SCAT_BLOCK, lds, s>>>(X, len, T, Bp, F, n_ids, per_block, cnt, tc, ipc);
        SBR_DYN_LDS(scat_scan_kernel<true>, lds);
        scat_scan_kernel<true><<<1, 1024, lds, s>>>(cnt, n_ids, offs, cur);
        scat_fill_lds_kernel<<<grid, SCAT_BLOCK, lds, s>>>(X, len, T, Bp, F, n_ids, per_block, cur, sid, spos, concat, tc, ipc);
    } else {
        const int grid = min(1024, (total + 255) / 256);
        const int agrid = (total + 1023) / 1024;
        scat_count_agg_kernel<<<agrid, 1024, 0, s>>>(X, len, T, Bp, F, cnt);
        const int nb = (n_ids + SCAN_BLK - 1) / SCAN_BLK;
        if (nb <= 1) scat_scan_kernel<false><<<1, 1024, 0, s>>>(cnt, n_ids, offs, cur);
        else {      // block totals in cur[0 .. nb) (rewritten by the add), their prefixes in cnt[0 .. nb) (not read any more)
            scat_scan_local_kernel<<<nb, 1024, 0, s>>>(cnt, n_ids, offs, cur);
            scat_scan_totals_kernel<<<1, 1024, 0, s>>>(cur, nb, cnt, offs + n_ids);
            scat_scan_add_kernel<<<nb, 1024, 0, s>>>(cnt, n_ids, offs, cur);
        }
        (void)grid;
        scat_fill_agg_kernel<<<agrid, 1024, 0, s>>>(X, len, T, Bp, F, cur, sid, spos, concat);
    }
    return hipGetLastError();
}

#define SCAT_FLY 8     // rows in flight per wave
// SCAT_CHUNK sorted entries per wave: 64 puts 800 waves on the chip for C2's 51 200 entries (200 workgroups, one wave per
// SIMD: latency-bound, 1.6 TB/s of row reads); smaller chunks trade more seam atomics for memory-level parallelism
// (SBR_SCAT_CHUNK = 16 / 32 / 64).
// key_lo > 0 or ACC: the launch covers the sorted entries of the keys [key_lo, key_lo + n_ids) only (one time chunk of the
// time-chunked sort, row id = key - key_lo) and ADDS to rows earlier launches of the same stream have written.
// POLL (overlapped step tail): ONE launch over all time chunks beside the running BPTT chain.  n_ids = ids per time chunk,
// keys = chunk * n_ids + id.  Waves take the sorted entries from the far end (late time steps are complete first); a wave
// waits (one lane polls poll.done, bounded) until the chain has left the time chunk of its FIRST entry -- the lowest of the
// wave, the order is ascending -- takes an agent-scope acquire, and adds its row sums with float atomics: an id may occur
// in every chunk, and chunks finish in different waves.
template <int NV, int SCAT_CHUNK, bool ACC = false, bool POLL = false>
__global__ void __launch_bounds__(256) scat_reduce_kernel(const f32x4* __restrict__ dxt, const int* __restrict__ sid,
                                                          const int* __restrict__ spos, const int* __restrict__ offs,
                                                          int n_ids, float* __restrict__ dWin, int R4, int Bp, int key_lo = 0,
                                                          int n_tchunks = 1, SbrTChunks tch = SbrTChunks(), SbrPoll poll = SbrPoll()) {
    // rows in flight per wave: narrow rows (NV = 1: at most 256 floats, C1's are 128) are latency, not bytes -- 16 of them
    constexpr int FLY = (NV == 1 && !POLL) ? 16 : SCAT_FLY;
    const int lane = threadIdx.x & 63;
    const int wave_global = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), n_waves = gridDim.x * (blockDim.x >> 6);
    const int total = offs[POLL ? n_ids * n_tchunks : key_lo + n_ids];
    // POLL: key_lo = the first key this launch takes (time chunk 0 -- the one the chain completes last -- is left to a launch
    // of its own behind the chain when key_lo = n_ids: see sbr_backward_recurrent)
    const int e_lo = POLL ? offs[key_lo] : 0;
    // POLL: the entries of the time chunks the chain completes LAST (chunks < poll.n_small: the last ~10 time steps with the
    // geometric bounds) are cut into SHORT pieces of SCAT_SHORT entries: what is left at the chain's end is then one round of
    // eight rows per wave instead of a 32-entry walk (four rounds and up to 32 flushes: 30 - 35 us behind the chain,
    // profiles/round3_f_timeline.txt)
    constexpr int SCAT_SHORT = 8;
    const int e_mid = POLL ? max(e_lo, min(total, offs[min(poll.n_small, n_tchunks) * n_ids])) : 0;
    const int nchA = POLL ? (total - e_mid + SCAT_CHUNK - 1) / SCAT_CHUNK : 0, nchB = POLL ? (e_mid - e_lo + SCAT_SHORT - 1) / SCAT_SHORT : 0;
    // POLL: a bounded number of waves (the launch must leave the chip to the GEMM that runs beside it) walks the sorted
    // entries from the far end, wave-chunk it, it + n_waves, ...
    for (int it = wave_global; ; it += n_waves) {
    int base, cnt;
    if (POLL) {
        if (it >= nchA + nchB) break;
        if (it < nchA) { base = e_mid + (nchA - 1 - it) * SCAT_CHUNK; cnt = min(SCAT_CHUNK, total - base); }
        else { base = e_lo + (nchB - 1 - (it - nchA)) * SCAT_SHORT; cnt = min(SCAT_SHORT, e_mid - base); }
    } else {
        base = offs[key_lo] + wave_global * SCAT_CHUNK;
        if (base >= total) return;
        cnt = min(SCAT_CHUNK, total - base);
    }
    const int lim = base + cnt;
    const int e = base + (lane & (SCAT_CHUNK - 1));
    const int my_id = e < lim ? sid[e] : -1;
    const int my_pos = e < lim ? spos[e] : 0;
    if (POLL) {
        const int t_need = tch.lo[min(__shfl(my_id, 0) / n_ids, SBR_TCHUNKS_MAX)];
        if (lane == 0) {
            const unsigned long long t0 = wall_clock64();
            const int* mine = poll.done + ((blockIdx.x + 16) & (SBR_DONE_COPIES - 1)) * SBR_DONE_STRIDE;
            for (;;) {
                const int v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 12) == poll.epoch && (v & 0xfff) <= t_need) break;
                if (wall_clock64() - t0 > SBR_POLL_TICKS) { atomicOr(poll.fault, 8); break; }
                poll_sleep((v >> 12) == poll.epoch ? (v & 0xfff) - t_need : 64);
            }
        }
        // (no acquire fence: on gfx950 that is an invalidate of the XCD's whole L2, ~15 000 cycles, for every released wave -- the
        // rows the chain wrote are read with sc1 loads below instead; sbr_gemm_x6.hip x6_poll_wait)
        __builtin_amdgcn_wave_barrier();
        if (poll.trace && lane == 0) {
            const int k = min((it - wave_global) / n_waves, 3);
            poll.trace[8192 + (wave_global * 4 + k) * 2] = wall_clock64();
        }
    }
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
    int cur_id = __shfl(my_id, 0);
#ifndef SCAT_DEBUG_NOFLUSH
#define SCAT_DEBUG_NOFLUSH 0      // 1 (tools/probes/variant_build.sh): the polling launch adds nothing -- WRONG gradients, for timing what its atomics cost
#endif
    auto flush = [&](int key) {
        if (POLL && SCAT_DEBUG_NOFLUSH) return;
        const bool owned = !POLL && offs[key] >= base && offs[key + 1] <= base + cnt;    // whole segment inside this chunk
        const int id = POLL ? key % n_ids : key - key_lo;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f4 = lane + 64 * v;
            if (f4 < R4) {
                float* dst = dWin + ((size_t)id * R4 + f4) * 4;
                if (owned) { if (ACC) *(f32x4*)dst += acc[v]; else *(f32x4*)dst = acc[v]; }
                else { atomicAdd(dst, acc[v][0]); atomicAdd(dst + 1, acc[v][1]); atomicAdd(dst + 2, acc[v][2]); atomicAdd(dst + 3, acc[v][3]); }
            }
            acc[v] = f32x4{0, 0, 0, 0};
        }
    };
    for (int i = 0; i < cnt; i += FLY) {
        f32x4 val[FLY][NV];
        int ids[FLY];
#pragma unroll
        for (int u = 0; u < FLY; ++u) {                 // rows in flight
            const int ii = min(i + u, cnt - 1);
            ids[u] = __shfl(my_id, ii);
            const size_t pos = (size_t)__shfl(my_pos, ii);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int f4 = lane + 64 * v;
                if constexpr (POLL)      // (lanes past the row load its last piece: no branch around the asm; zeroed once landed)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(val[u][v]) : "v"(dxt + pos * R4 + min(f4, R4 - 1)) : "memory");
                else val[u][v] = f4 < R4 ? dxt[pos * R4 + f4] : f32x4{0, 0, 0, 0};
            }
        }
        if constexpr (POLL) {      // (the loads above are invisible to the compiler's waitcnt insertion)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < FLY; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    asm volatile("" : "+v"(val[u][v]));
                    if (lane + 64 * v >= R4) val[u][v] = f32x4{0, 0, 0, 0};
                }
        }
#pragma unroll
        for (int u = 0; u < FLY; ++u) {
            if (i + u < cnt) {                                 // wave-uniform
                if (ids[u] != cur_id) { flush(cur_id); cur_id = ids[u]; }
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
            }
        }
    }
    flush(cur_id);
    if (POLL && poll.trace && lane == 0) {
        const int k = min((it - wave_global) / n_waves, 3);
        poll.trace[8192 + (wave_global * 4 + k) * 2 + 1] = wall_clock64();
    }
    if (!POLL) break;
    }
}

hipError_t launch_scatter_reduce_poll(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs,
                                      int n_ids, int n_tchunks, int tch, int max_entries, int GHp, int Bp, const SbrPoll& poll_in, int first_key,
                                      const SbrTChunks* bounds, int short_chunks, bool fence_on) {
    const SbrTChunks tc = bounds ? *bounds : sbr_uniform_tchunks(tch, n_tchunks);
    SbrPoll poll = poll_in;
    poll.n_small = std::max(0, std::min(short_chunks, n_tchunks));      // (the field counts the GEMM's short slabs there; here: time chunks cut short)

    const int R4 = GHp / 4, nv = (R4 + 63) / 64;
    const int wgs = 128;   // (round 3: 128 with the short pieces; 64 before)
    const int grid = std::max(1, std::min(wgs, ((max_entries + 7) / 8 + 3) / 4));
    // (the LDS this launch asks for is a FENCE, not storage: with it a workgroup does not fit beside the BPTT chain's, which claims
    // 124 KB of its CU's 160 for the same purpose -- sbr_rec_p.hip launch_bwd_p)
    const size_t fence = fence_on ? (size_t)40 * 1024 : 0;
#define SRP(NV) scat_reduce_kernel<NV, 32, true, true><<<grid, 256, fence, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4, Bp, first_key, n_tchunks, tc, poll)
    if (nv <= 1) SRP(1); else if (nv <= 2) SRP(2); else if (nv <= 4) SRP(4); else return hipErrorInvalidValue;
#undef SRP
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Overlapped step tail, round 3: the scatter-add WITHOUT one global atomic per piece and row.
// The polling launch above adds every piece's row sums to dW_in with float atomics: an id occurs in every time chunk and its
// entries are cut into pieces, ~10 000 flushes of 384 floats per C2 step.  On gfx950 agent-scope atomics are executed by the
// memory side, and 4 M of them beside the chain cost the chain 13 us, the polling GEMM's last slabs 10 us and this launch's
// own end 17 us (profiles/round3_w_trace.txt: the same step with the flush removed).
// Here a workgroup OWNS a range of ids: it walks the time chunks in the order the chain releases them, adds the rows of its
// ids' entries into accumulators in LDS (ds_add_f32), and stores each row ONCE when the chain has ended.  Ranges are cut in
// COST space, cost(id) = max(entries of id over all chunks, floor): unit u owns [u Q, (u + 1) Q) of the running sum P, so that
// units are balanced in entries and hold at most Q / floor + 2 ids (the LDS rows); an id that straddles a cut -- every hot id
// does -- is shared: in EVERY chunk each sharing unit takes the same fraction of that id's entries (the cut is uniform in
// time, so nobody is left with only the last time steps), and only those rows, at most two per unit, end in global atomics.
// In the time-chunked sort the entries of a unit in one chunk are one contiguous range of the sorted array.
// ---------------------------------------------------------------------------------------------------------------------
// the monitor's loop (tail_monitor_kernel below; one workgroup of 256 threads): see there
__device__ __forceinline__ void tail_monitor_loop(const SbrPoll& pl, int t_lo) {
    __shared__ int s_part[4];
    const int tid = threadIdx.x, tag = pl.epoch;
    int last = 0x1000;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        int m = 0;
        for (int i = tid; i < pl.n; i += 256) {
            const int v = __hip_atomic_load(pl.words + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            m = max(m, (v >> 12) == tag ? (v & 0xfff) : 0xfff);          // (a wave that has not published this step yet: nothing complete)
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = max(m, __shfl_xor(m, o));
        if ((tid & 63) == 0) s_part[tid >> 6] = m;
        __syncthreads();
        m = max(max(s_part[0], s_part[1]), max(s_part[2], s_part[3]));
        __syncthreads();
        if (m != last && m != 0xfff) {
            if (tid < SBR_DONE_COPIES) __hip_atomic_store(pl.done + tid * SBR_DONE_STRIDE, (tag << 12) | m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = m;
        }
        if (m <= t_lo) break;
        if (wall_clock64() - t0 > SBR_POLL_TICKS) { if (tid == 0) atomicOr(pl.fault, 8); break; }
        __builtin_amdgcn_s_sleep(4);
    }
}
__global__ void __launch_bounds__(1024) scat_cost_scan_kernel(const int* __restrict__ offs, int n_ids, int n_tchunks, int floor_cost,
                                                              int* __restrict__ P) {
    __shared__ int part[1024];
    const int per = (n_ids + 1023) / 1024, lo = threadIdx.x * per, hi = min(n_ids, lo + per);
    int sum = 0;
    for (int id = lo; id < hi; ++id) {
        int tot = 0;
        for (int c = 0; c < n_tchunks; ++c) tot += offs[c * n_ids + id + 1] - offs[c * n_ids + id];
        sum += max(tot, floor_cost);
    }
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;                               // exclusive
    for (int id = lo; id < hi; ++id) {
        int tot = 0;
        for (int c = 0; c < n_tchunks; ++c) tot += offs[c * n_ids + id + 1] - offs[c * n_ids + id];
        P[id] = run;
        run += max(tot, floor_cost);
    }
    if (threadIdx.x == 1023) P[n_ids] = part[1023];
}

template <int NV>
__global__ void __launch_bounds__(256) scat_lds_kernel(const f32x4* __restrict__ dxt, const int* __restrict__ sid, const int* __restrict__ spos,
                                                       const int* __restrict__ offs, const int* __restrict__ P, int n_ids, int n_tchunks,
                                                       float* __restrict__ dWin, int R4, SbrTChunks tch, SbrPoll poll, int rows_lds,
                                                       int monitor, int t_lo) {
    // monitor != 0: workgroup 0 -- the first on the chip -- is the tail's MONITOR (tail_monitor_loop) and nothing else; the
    // units are workgroups 1 .. gridDim.x - 1.  No stream of its own then -- hardware queues are few (a FOURTH side stream for
    // the monitor halved the throughput of everything, profiles/round3_A_variants.txt), and this launch is the first of the
    // tail on its stream anyway.
    if (monitor && blockIdx.x == 0) { tail_monitor_loop(poll, t_lo); return; }
    const int unit = (int)blockIdx.x - (monitor ? 1 : 0);
    extern __shared__ float rows[];                                  // [rows_lds][4][R4]: component-major rows (no bank conflicts)
    __shared__ int s_meta[4];
    __shared__ int s_e0[SBR_TCHUNKS_MAX + 1], s_e1[SBR_TCHUNKS_MAX + 1], s_kf[SBR_TCHUNKS_MAX + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int total = P[n_ids], U = (int)gridDim.x - (monitor ? 1 : 0), Q = (total + U - 1) / U;
    const int lo = unit * Q, hi = min(total, lo + Q);
    if (lo >= hi) return;
    if (tid == 0) {                                                  // the ids of P that hold cost positions lo and hi - 1
        int a = 0, b = n_ids - 1;
        while (a < b) { const int m = (a + b + 1) >> 1; if (P[m] <= lo) a = m; else b = m - 1; }
        s_meta[0] = a;
        b = n_ids - 1;
        while (a < b) { const int m = (a + b + 1) >> 1; if (P[m] <= hi - 1) a = m; else b = m - 1; }
        s_meta[1] = a;
    }
    __syncthreads();
    const int id_first = s_meta[0], id_last = s_meta[1];
    int n_rows = id_last - id_first + 1;
    if (n_rows > rows_lds) { if (tid == 0) atomicOr(poll.fault, 16); n_rows = rows_lds; }      // (cannot happen: launch_scatter_lds_poll sizes rows_lds)
    const long num0 = lo - P[id_first], den0 = P[id_first + 1] - P[id_first];
    const long num1 = hi - P[id_last], den1 = P[id_last + 1] - P[id_last];
    const int RW = 4 * R4;
    for (int i = tid; i < n_rows * RW; i += 256) rows[i] = 0.f;
    if (tid < n_tchunks) {                                           // this unit's range of the sorted entries in every time chunk
        const int kf = tid * n_ids + id_first, kl = tid * n_ids + id_last;
        const int b0 = offs[kf], b1 = offs[kl];
        s_e0[tid] = b0 + (int)(num0 * (offs[kf + 1] - b0) / den0);
        s_e1[tid] = b1 + (int)(num1 * (offs[kl + 1] - b1) / den1);
        s_kf[tid] = kf;
    }
    __syncthreads();
    int seen = 0xfff;
    const int* mine = poll.done + ((unit + 16) & (SBR_DONE_COPIES - 1)) * SBR_DONE_STRIDE;
    for (int c = n_tchunks - 1; c >= 0; --c) {
        const int kf = s_kf[c], e0 = s_e0[c], e1 = s_e1[c];
        if (e0 >= e1) continue;                                      // (uniform)
        const int t_need = tch.lo[c];
        // the first strip's keys and positions (written by the sort, long complete) are on their way while the workgroup waits
        const int base_0 = e0 + wave * SCAT_FLY, cnt_0 = min(SCAT_FLY, e1 - base_0);
        int key_0 = lane < cnt_0 ? sid[base_0 + lane] : -1;
        int pos_0 = lane < cnt_0 ? spos[base_0 + lane] : 0;
        if (seen > t_need) {                                         // the chain has not left this time chunk yet, as far as this workgroup knows
            if (tid == 0) {
                const unsigned long long t0 = wall_clock64();
                for (;;) {
                    const int v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((v >> 12) == poll.epoch && (v & 0xfff) <= t_need) { s_meta[2] = v & 0xfff; break; }
                    if (wall_clock64() - t0 > SBR_POLL_TICKS) { atomicOr(poll.fault, 8); s_meta[2] = 0; break; }
                    poll_sleep((v >> 12) == poll.epoch ? (v & 0xfff) - t_need : 64);
                }
            }
            __syncthreads();
            seen = s_meta[2];
            __syncthreads();
        }
        if (poll.trace && tid == 0) poll.trace[8192 + (unit * 16 + c) * 2] = wall_clock64();
        // strips of SCAT_FLY entries, round-robin over the four waves; rows read with sc1 loads (no acquire fence: scat_reduce_kernel)
        for (int base = base_0; base < e1; base += 4 * SCAT_FLY) {
            const int cnt = min(SCAT_FLY, e1 - base);
            const int my_key = base == base_0 ? key_0 : (lane < cnt ? sid[base + lane] : -1);
            const int my_pos = base == base_0 ? pos_0 : (lane < cnt ? spos[base + lane] : 0);
            f32x4 val[SCAT_FLY][NV];
#pragma unroll
            for (int u = 0; u < SCAT_FLY; ++u) {
                const size_t pos = (size_t)__shfl(my_pos, min(u, cnt - 1));
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(val[u][v]) : "v"(dxt + pos * R4 + min(lane + 64 * v, R4 - 1)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < SCAT_FLY; ++u)
#pragma unroll
                for (int v = 0; v < NV; ++v) asm volatile("" : "+v"(val[u][v]));
            f32x4 acc[NV];
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
            int cur = __shfl(my_key, 0);
            auto flush = [&](int key) {
                float* r = rows + (size_t)min(key - kf, n_rows - 1) * RW;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int f4 = lane + 64 * v;
                    if (f4 < R4) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) atomicAdd(r + e * R4 + f4, acc[v][e]);
                    }
                    acc[v] = f32x4{0, 0, 0, 0};
                }
            };
#pragma unroll
            for (int u = 0; u < SCAT_FLY; ++u) {
                if (u < cnt) {                                       // (uniform)
                    const int k = __shfl(my_key, u);
                    if (k != cur) { flush(cur); cur = k; }
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
                }
            }
            flush(cur);
        }
        if (poll.trace && tid == 0) poll.trace[8192 + (unit * 16 + c) * 2 + 1] = wall_clock64();
    }
    __syncthreads();
    // one store per row; the (at most two) rows shared with the neighbouring units go by atomics
    for (int r = wave; r < n_rows; r += 4) {
        const bool shared = (r == 0 && num0 > 0) || (r == n_rows - 1 && num1 < den1);
        const float* src = rows + (size_t)r * RW;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f4 = lane + 64 * v;
            f32x4 x = f32x4{0, 0, 0, 0};
            if (f4 < R4) { x[0] = src[f4]; x[1] = src[R4 + f4]; x[2] = src[2 * R4 + f4]; x[3] = src[3 * R4 + f4]; }
            const bool nz = x[0] != 0.f || x[1] != 0.f || x[2] != 0.f || x[3] != 0.f;
            if (f4 < R4 && nz) {
                float* dst = dWin + ((size_t)(id_first + r) * R4 + f4) * 4;
                if (shared) { atomicAdd(dst, x[0]); atomicAdd(dst + 1, x[1]); atomicAdd(dst + 2, x[2]); atomicAdd(dst + 3, x[3]); }
                else *(f32x4*)dst = x;
            }
        }
    }
    if (poll.trace && tid == 0) poll.trace[8192 + (unit * 16 + 15) * 2] = wall_clock64();
}

// LDS rows and cost floor of a launch: ids of an average unit = 16, + the ids the floor admits.  false: shape not supported.
static bool scat_lds_plan(int n_ids, int n_tchunks, int max_entries, int GHp, int units, int* floor_cost, int* rows_lds) {
    const int R4 = GHp / 4, nv = (R4 + 63) / 64;
    if ((GHp & 3) || nv > 2 || n_tchunks < 1 || n_tchunks > SBR_TCHUNKS_MAX || units < 1) return false;
    const int rows0 = 16;
    *floor_cost = std::max(1, (max_entries + units * rows0 - 1) / (units * rows0));
    const long q_max = ((long)max_entries + (long)n_ids * *floor_cost + units - 1) / units;
    *rows_lds = (int)(q_max / *floor_cost) + 3;
    return (size_t)*rows_lds * GHp * sizeof(float) <= 120 * 1024;
}
// behind the time-chunked sort (same stream): the running cost P[0 .. n_ids] of the ids.  false = launch_scatter_lds_poll will refuse too.
bool launch_scatter_cost_scan(hipStream_t s, const int* offs, int* P, int n_ids, int n_tchunks, int max_entries, int GHp, int units,
                              hipError_t* err) {
    int floor_cost = 0, rows_lds = 0;
    if (!scat_lds_plan(n_ids, n_tchunks, max_entries, GHp, units, &floor_cost, &rows_lds)) return false;
    scat_cost_scan_kernel<<<1, 1024, 0, s>>>(offs, n_ids, n_tchunks, floor_cost, P);
    *err = hipGetLastError();
    return true;
}
// false: the shape does not fit (rows too long for the LDS) -- the caller launches launch_scatter_reduce_poll instead
bool launch_scatter_lds_poll(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs, const int* P,
                             int n_ids, int n_tchunks, int max_entries, int GHp, const SbrPoll& poll, const SbrTChunks& bounds,
                             int units, hipError_t* err, bool monitor, int t_lo) {
    int floor_cost = 0, rows_lds = 0;
    if (!scat_lds_plan(n_ids, n_tchunks, max_entries, GHp, units, &floor_cost, &rows_lds)) return false;
    const int R4 = GHp / 4, nv = (R4 + 63) / 64, grid = units + (monitor ? 1 : 0), mon = monitor ? 1 : 0;
    const size_t lds = (size_t)rows_lds * GHp * sizeof(float);
    if (nv <= 1) { SBR_DYN_LDS(scat_lds_kernel<1>, lds); scat_lds_kernel<1><<<grid, 256, lds, s>>>((const f32x4*)dxt, sid, spos, offs, P, n_ids, n_tchunks, dWin, R4, bounds, poll, rows_lds, mon, t_lo); }
    else { SBR_DYN_LDS(scat_lds_kernel<2>, lds); scat_lds_kernel<2><<<grid, 256, lds, s>>>((const f32x4*)dxt, sid, spos, offs, P, n_ids, n_tchunks, dWin, R4, bounds, poll, rows_lds, mon, t_lo); }
    *err = hipGetLastError();
    return true;
}

// ---------------------------------------------------------------------------------------
// Wide rows (G*Hp >= 512 floats: the cluster layers, C3 / C4 / C5): segment-parallel scatter-add, no atomics on rows, fixed order.
// The wave-per-32-entries kernel above adds every segment that is not wholly inside a chunk with float atomics; with Zipf ids
// the hot rows' segments span hundreds of chunks and their adds serialise at the memory side (C4: 199 us for 210 MB of rows,
// 1 TB/s; C5 571 us) -- while most segments of a large catalogue are ONE entry, i.e. a row copy.  Three launches:
//   heads   one wave per sorted entry; a wave whose entry is not the first of its id leaves at once.  The head of a segment of
//           <= SCATW_SHORT entries sums its rows (SCATW_FLY in flight, 16 bytes per lane and piece) and stores dW_in[id]; the head
//           of a longer segment claims ceil(len / SCATW_SHORT) slots of the partial-row slab (one atomic on a counter) and files
//           (id, first entry, length, first slot) in the list of long segments
//   pieces  one workgroup per slab slot: the SCATW_SHORT entries of that piece -> the slot's partial row
//   merge   one workgroup per long segment: its slots summed in order -> dW_in[id]
// Every row has one writer and a fixed summation order (the slot numbers depend on which head reached the counter first, the sums
// do not): the gradient is bit-reproducible.  A long segment claims ceil(len / 64) <= len / 64 + 1 slots and has more than 64 entries:
// at most entries / 64 + entries / 65 slots and entries / 65 long segments exist (sbr_scatter_wide_slots; the launcher refuses a
// smaller slab -- round 4 sized it entries / 64 + 2, which segments of 65 .. 127 entries could overrun).
// (sparse_lstm.py:368: the AdvancedIncSubtensor the reference's backward builds.)
// ---------------------------------------------------------------------------------------
#define SCATW_SHORT 64
#define SCATW_FLY 4
struct ScatLong { int id, first, len, slot; };
template <int NV>
__global__ void __launch_bounds__(256) scat_heads_kernel(const f32x4* __restrict__ dxt, const int* __restrict__ sid,
                                                         const int* __restrict__ spos, const int* __restrict__ offs, int n_ids,
                                                         float* __restrict__ dWin, int R4, int* __restrict__ counters,
                                                         ScatLong* __restrict__ longs) {
    const int lane = threadIdx.x & 63;
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), total = offs[n_ids];
    if (e >= total) return;
    const int id = sid[e];
    if (e > 0 && sid[e - 1] == id) return;               // not the head of its segment
    const int len = offs[id + 1] - offs[id];
    if (len > SCATW_SHORT) {
        if (lane == 0) {
            const int np = (len + SCATW_SHORT - 1) / SCATW_SHORT;
            const int slot = atomicAdd(counters, np), k = atomicAdd(counters + 1, 1);
            longs[k] = ScatLong{id, e, len, slot};
        }
        return;
    }
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < len; i += SCATW_FLY) {
        f32x4 val[SCATW_FLY][NV];
#pragma unroll
        for (int u = 0; u < SCATW_FLY; ++u) {
            if (i + u < len) {                           // uniform (most segments of a large catalogue are one entry: a row copy)
                const size_t pos = (size_t)spos[e + i + u];
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int f4 = lane + 64 * v;
                    val[u][v] = f4 < R4 ? dxt[pos * R4 + f4] : f32x4{0, 0, 0, 0};
                }
            } else {
#pragma unroll
                for (int v = 0; v < NV; ++v) val[u][v] = f32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < SCATW_FLY; ++u)
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) { const int f4 = lane + 64 * v; if (f4 < R4) ((f32x4*)dWin)[(size_t)id * R4 + f4] = acc[v]; }
}

// one workgroup per slab slot: blockIdx.x = slot; the segment that owns it is found by a scan of the (short) list
template <int NV>
__global__ void __launch_bounds__(256) scat_pieces_kernel(const f32x4* __restrict__ dxt, const int* __restrict__ spos,
                                                          const int* __restrict__ counters, const ScatLong* __restrict__ longs,
                                                          f32x4* __restrict__ part, int R4) {
    const int slot = blockIdx.x, tid = threadIdx.x;
    if (slot >= counters[0]) return;
    const int nl = counters[1];
    __shared__ int s_seg;
    if (tid == 0) s_seg = -1;
    __syncthreads();
    for (int k = tid; k < nl; k += 256) {
        const ScatLong L = longs[k];
        if (slot >= L.slot && slot < L.slot + (L.len + SCATW_SHORT - 1) / SCATW_SHORT) s_seg = k;
    }
    __syncthreads();
    if (s_seg < 0) return;
    const ScatLong L = longs[s_seg];
    const int lo = L.first + (slot - L.slot) * SCATW_SHORT, hi = min(L.first + L.len, lo + SCATW_SHORT);
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
    for (int i = lo; i < hi; i += 8) {
        f32x4 val[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t pos = (size_t)spos[min(i + u, hi - 1)];
#pragma unroll
            for (int v = 0; v < NV; ++v) { const int f4 = tid + 256 * v; val[u][v] = f4 < R4 ? dxt[pos * R4 + f4] : f32x4{0, 0, 0, 0}; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i + u < hi) {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
            }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) { const int f4 = tid + 256 * v; if (f4 < R4) part[(size_t)slot * R4 + f4] = acc[v]; }
}

template <int NV>
__global__ void __launch_bounds__(256) scat_long_merge_kernel(const f32x4* __restrict__ part, const int* __restrict__ counters,
                                                              const ScatLong* __restrict__ longs, float* __restrict__ dWin, int R4) {
    const int k = blockIdx.x, tid = threadIdx.x;
    if (k >= counters[1]) return;
    const ScatLong L = longs[k];
    const int np = (L.len + SCATW_SHORT - 1) / SCATW_SHORT;
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
    for (int p0 = 0; p0 < np; p0 += 8) {
        f32x4 val[8][NV];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const size_t sl = (size_t)(L.slot + min(p0 + u, np - 1));
#pragma unroll
            for (int v = 0; v < NV; ++v) { const int f4 = tid + 256 * v; val[u][v] = f4 < R4 ? part[sl * R4 + f4] : f32x4{0, 0, 0, 0}; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p0 + u < np) {
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
            }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) { const int f4 = tid + 256 * v; if (f4 < R4) ((f32x4*)dWin)[(size_t)L.id * R4 + f4] = acc[v]; }
}

// part: n_slots * GHp floats (n_slots >= sbr_scatter_wide_slots(max_entries)); aux: 2 counters + n_slots ScatLong records (ints);
// false: shape not served (the caller takes launch_scatter_reduce)
bool launch_scatter_wide(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs, int n_ids,
                         int max_entries, int GHp, float* part, int* aux, int n_slots, hipError_t* err) {
    const int R4 = GHp / 4, nvw = (R4 + 63) / 64;
    if ((GHp & 3) || GHp < 512 || nvw > 8 || !part || !aux || n_slots < sbr_scatter_wide_slots((size_t)max_entries)) return false;
    if (hipMemsetAsync(aux, 0, 2 * sizeof(int), s) != hipSuccess) { *err = hipGetLastError(); return true; }
    int* counters = aux; ScatLong* longs = (ScatLong*)(aux + 4);
    const int gh = (max_entries + 3) / 4;
#define SW(NVW, NVB) do { \
        scat_heads_kernel<NVW><<<gh, 256, 0, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4, counters, longs); \
        scat_pieces_kernel<NVB><<<n_slots, 256, 0, s>>>((const f32x4*)dxt, spos, counters, longs, (f32x4*)part, R4); \
        scat_long_merge_kernel<NVB><<<n_slots, 256, 0, s>>>((const f32x4*)part, counters, longs, dWin, R4); } while (0)
    if (nvw <= 2) SW(2, 1); else if (nvw <= 4) SW(4, 1); else SW(8, 2);
#undef SW
    *err = hipGetLastError();
    return true;
}

// ---------------------------------------------------------------------------------------
// The optimizer step of one element / one 16-byte piece on registers (SbrScatStep in sbr_common.h): update_element's arithmetic,
// for the kernels that hold a finished gradient in registers (update_rows_aware_kernel, out_grad_step_kernel).
// (Round 5 also built a scatter-add that stepped the rows it completed -- launch_scatter_wide_step: correct, 20 % fewer bytes behind
// the chain, and slower (C4 1.324 -> 1.375 ms): its waves held a row's optimizer state beside the dxt rows of a segment and ran at
// 1.5 - 2.3 TB/s where the two-pass form streams at 3.6 - 6.2.  Removed in round 6; the numbers are in profiles/round5_a_*.)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void scat_step1(const SbrScatStep& st, float g, float& p, float& s0, float& s1) {
    switch (st.updater) {                                     // update_element's arithmetic (sbr_misc.hip K13), on registers
        case SBR_UPD_ADAGRAD: { const float acc = s0 + g * g; s0 = acc; p -= st.lr * g / sqrtf(acc + 1e-6f); return; }
        case SBR_UPD_RMSPROP: { const float acc = st.rho * s0 + (1.0f - st.rho) * g * g; s0 = acc; p -= st.lr * g / sqrtf(acc + 1e-6f); return; }
        case SBR_UPD_ADADELTA: {
            const float acc = st.rho * s0 + (1.0f - st.rho) * g * g;
            const float upd = g * sqrtf(s1 + 1e-6f) / sqrtf(acc + 1e-6f);
            s0 = acc; p -= st.lr * upd; s1 = st.rho * s1 + (1.0f - st.rho) * upd * upd; return;
        }
        case SBR_UPD_NESTEROV: { const float v = st.rho * s0 - st.lr * g; s0 = v; p += st.rho * v - st.lr * g; return; }
        default: {
            const float m = st.b1 * s0 + (1.0f - st.b1) * g;
            const float v = st.b2 * s1 + (1.0f - st.b2) * g * g;
            s0 = m; s1 = v; p -= st.a_t * m / (sqrtf(v) + 1e-8f); return;
        }
    }
}
__device__ __forceinline__ void scat_step4(const SbrScatStep& st, const f32x4& g, f32x4& p, f32x4& s0, f32x4& s1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { float pe = p[e], ae = s0[e], be = s1[e]; scat_step1(st, g[e], pe, ae, be); p[e] = pe; s0[e] = ae; s1[e] = be; }
}

// ---------------------------------------------------------------------------------------
// Wide rows, form 1 ("range" scatter-add; default for G*Hp <= 1024: C3 / C4): two passes, no atomics, fixed order.
// (Form 2, the segment-parallel kernels above, is the faster one ALONE -- 100 against 164 us at C4 -- but this one keeps its pace beside the
// weight-gradient GEMM it shares the chip with after the BPTT chain: 512 persistent workgroups instead of 12 800 short ones; measured
// C4 1.315 ms with it against 1.372 with form 2 on the same stream and 1.370 with form 2 alone in front of the GEMM: profiles/round4_variants_wide.txt.)
// The wave-per-32-entries kernel above adds every segment that is not wholly inside a chunk with float atomics; with Zipf ids
// the hot rows' segments span hundreds of chunks and their adds serialise at the memory side (C4: 199 us for 210 MB of rows,
// 1 TB/s; C5 571 us).  Here a workgroup owns a contiguous RANGE of the sorted entries, a thread owns one 16-byte piece of
// the row (a 1024-float row = one piece per thread: every row load is one fully coalesced 4 KB access, SCATR_FLY rows in
// flight), the range is walked once with the running sum in registers.  Segments inside the range are stored straight to
// dW_in; the range's FIRST and LAST segment -- the only ones another range can share -- go to a partial-row slab
// [range][2][row] with their ids beside them, and the merge pass (one workgroup per partial slot; the leader of a run of
// equal ids sums the run in slot order) writes those rows.  Every row has exactly one writer and a fixed summation order:
// the gradient is bit-reproducible.  (sparse_lstm.py:368: the AdvancedIncSubtensor the reference's backward builds.)
// ---------------------------------------------------------------------------------------
// (Round 6 measured a software-pipelined walk -- the range's (id, position) pairs staged in LDS, two register sets of SCATR_FLY rows
// alternating, the flushes counted out of its waits: ALONE on the chip 60.8 against 67.5 us at C4 (4.17 TB/s), but 194 against 112 us
// in the step, where it runs beside the weight-gradient GEMM: 80 registers instead of 48 halve what fits beside that launch's
// workgroups.  The two launches together are no faster than one after the other either way -- profiles/round6_variants.txt, calls d, e.)
#define SCATR_FLY 8
template <int NV>
__global__ void __launch_bounds__(256) scat_range_kernel(const f32x4* __restrict__ dxt, const int* __restrict__ sid,
                                                         const int* __restrict__ spos, const int* __restrict__ total_p,
                                                         float* __restrict__ dWin, int R4, f32x4* __restrict__ part,
                                                         int* __restrict__ part_id) {
    const int total = *total_p, nr = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    const int E = (total + nr - 1) / nr;
    const int lo = min(total, w * E), hi = min(total, lo + E);
    if (lo >= hi) { if (tid < 2) part_id[2 * w + tid] = -1; return; }
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
    const int first_id = sid[lo], last_id = sid[hi - 1];
    int cur = first_id;
    bool in_first = true;
    auto flush = [&](int id, bool last) {              // uniform arguments
        f32x4* dst = (in_first || last) ? part + ((size_t)(2 * w + (in_first ? 0 : 1)) * R4)
                                        : (f32x4*)dWin + (size_t)id * R4;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f4 = tid + 256 * v;
            if (f4 < R4) dst[f4] = acc[v];
            acc[v] = f32x4{0, 0, 0, 0};
        }
        in_first = false;
    };
    for (int i = lo; i < hi; i += SCATR_FLY) {
        f32x4 val[SCATR_FLY][NV];
        int ids[SCATR_FLY];
#pragma unroll
        for (int u = 0; u < SCATR_FLY; ++u) {
            const int e = min(i + u, hi - 1);
            ids[u] = sid[e];
            const size_t pos = (size_t)spos[e];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int f4 = tid + 256 * v;
                val[u][v] = f4 < R4 ? dxt[pos * R4 + f4] : f32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < SCATR_FLY; ++u) {
            if (i + u < hi) {                            // uniform
                const int id = __builtin_amdgcn_readfirstlane(ids[u]);
                if (id != cur) { flush(cur, false); cur = id; }
#pragma unroll
                for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
            }
        }
    }
    const bool single = in_first;                        // the whole range is one segment: it went (goes) to slot 0
    flush(cur, true);
    if (tid == 0) { part_id[2 * w] = first_id; part_id[2 * w + 1] = single ? -1 : last_id; }
}

// The merge of the shared first / last segments.  A hot id's run spans many slots (Zipf: item 0 is ~9 % of C4's entries = ~90 slots of
// 512 ranges); the leader summed them one dependent load after the other (47 us at C4).  Now: the run's end first (ids only), then
// its rows SCATR_FLY at a time.
template <int NV>
__global__ void __launch_bounds__(256) scat_range_merge_kernel(const f32x4* __restrict__ part, const int* __restrict__ part_id,
                                                               int n_slots, float* __restrict__ dWin, int R4) {
    const int p = blockIdx.x, tid = threadIdx.x;
    const int id = part_id[p];
    if (id < 0) return;
    // previous slot that holds a row: the last segment of the range in front, or (that range being one segment) its first
    int prev = -2;
    if (p & 1) prev = part_id[p - 1];
    else if (p >= 2) prev = part_id[p - 1] >= 0 ? part_id[p - 1] : part_id[p - 2];
    if (prev == id) return;                              // not the leader of its run
    int qend = p + 1;                                    // one past the run's last slot (empty slots inside a run are skipped below)
    while (qend < n_slots) { const int iq = part_id[qend]; if (iq >= 0 && iq != id) break; ++qend; }
    f32x4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) { const int f4 = tid + 256 * v; acc[v] = f4 < R4 ? part[(size_t)p * R4 + f4] : f32x4{0, 0, 0, 0}; }
    for (int q0 = p + 1; q0 < qend; q0 += SCATR_FLY) {
        f32x4 val[SCATR_FLY][NV];
#pragma unroll
        for (int u = 0; u < SCATR_FLY; ++u) {
            const int q = min(q0 + u, qend - 1);
            const bool live = q0 + u < qend && part_id[q] == id;       // uniform
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int f4 = tid + 256 * v;
                val[u][v] = (live && f4 < R4) ? part[(size_t)q * R4 + f4] : f32x4{0, 0, 0, 0};
            }
        }
#pragma unroll
        for (int u = 0; u < SCATR_FLY; ++u)              // slot order: the sum is reproducible
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) { const int f4 = tid + 256 * v; if (f4 < R4) ((f32x4*)dWin)[(size_t)id * R4 + f4] = acc[v]; }
}

// part: n_ranges * 2 * GHp floats, part_id: n_ranges * 2 ints; false: shape not served (the caller takes launch_scatter_reduce)
bool launch_scatter_range(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs, int n_ids,
                          int GHp, float* part, int* part_id, int n_ranges, hipError_t* err) {
    const int R4 = GHp / 4, nv = (R4 + 255) / 256;
    if ((GHp & 3) || GHp < 512 || nv > 4 || !part || !part_id || n_ranges < 1) return false;
    const int* total_p = offs + n_ids;
#define SRG(NV) do { scat_range_kernel<NV><<<n_ranges, 256, 0, s>>>((const f32x4*)dxt, sid, spos, total_p, dWin, R4, (f32x4*)part, part_id); \
                     scat_range_merge_kernel<NV><<<2 * n_ranges, 256, 0, s>>>((const f32x4*)part, part_id, 2 * n_ranges, dWin, R4); } while (0)
    if (nv <= 1) SRG(1); else if (nv <= 2) SRG(2); else SRG(4);
#undef SRG
    *err = hipGetLastError();
    return true;
}

// The plain scatter-add of rows up to 512 floats (round 6): scat_reduce_kernel's walk, 32 sorted entries per wave, with the partial rows
// of the segments that cross a wave's chunk COMBINED inside the workgroup before anything is added to memory.  A hot id spans many
// chunks, and every one of them used to add its partial row onto the same addresses with float atomics -- at C1 (3 706 ids, the hottest
// with thousands of the step's 51 200 entries) that serialisation was most of the launch (41 us for 26 MB).  Here the W waves of a
// workgroup take W consecutive chunks, leave at most two open partial rows each (the segment that began before the chunk, the one that
// goes on behind it) in LDS, and one wave adds up the runs of equal ids: a run that lies inside the workgroup's entries is STORED, only
// the runs that cross a workgroup boundary are added atomically -- W times fewer atomics on the hot rows.
template <int NV>
__global__ void __launch_bounds__(512) scat_reduce_comb_kernel(const f32x4* __restrict__ dxt, const int* __restrict__ sid,
                                                                              const int* __restrict__ spos, const int* __restrict__ offs,
                                                                              int n_ids, float* __restrict__ dWin, int R4) {
#ifndef COMB_CH
#define COMB_CH 32      // sorted entries per wave: 64 / 32 / 16 -> 29.3 / 21.5 / 22.6 us at C2 (profiles/round6_variants.txt, call y)
#endif
#ifndef COMB_FLY
#define COMB_FLY 8
#endif
    constexpr int CH = COMB_CH, FLY = NV >= 8 ? 2 : NV >= 4 ? 4 : COMB_FLY, W = 8;      // (wide rows: fewer in flight, the pieces stay near 64 registers)
    extern __shared__ __attribute__((aligned(16))) char comb_lds[];
    f32x4 (*slot)[NV * 64] = (f32x4 (*)[NV * 64])comb_lds;                         // [2 W][NV * 64]
    int* slot_id = (int*)(comb_lds + (size_t)2 * W * NV * 64 * sizeof(f32x4));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2 * W) slot_id[threadIdx.x] = -1;
    __syncthreads();
    const int total = offs[n_ids];
    const int wg_base = blockIdx.x * W * CH, wg_end = min(total, wg_base + W * CH);
    const int base = wg_base + wave * CH;
    // Whether a segment ends inside a chunk / a run inside the workgroup is read off the NEIGHBOURING entries' ids (the entries are sorted),
    // two loads per wave up front -- not off offs[key], offs[key + 1] at every flush: with a million ids those were two dependent trips to
    // HBM per distinct id of the chunk (C5: ~30 per wave), most of the launch.
    const int wg_prev = (wave == 0 && wg_base > 0 && wg_base < total) ? sid[wg_base - 1] : -1;
    const int wg_next = (wave == 0 && wg_end < total) ? sid[wg_end] : -1;
    auto row_out = [&](int key, const f32x4 (&a)[NV], bool store) {
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const int f4 = lane + 64 * v;
            if (f4 < R4) {
                float* dst = dWin + ((size_t)key * R4 + f4) * 4;
                if (store) *(f32x4*)dst = a[v];
                else { atomicAdd(dst, a[v][0]); atomicAdd(dst + 1, a[v][1]); atomicAdd(dst + 2, a[v][2]); atomicAdd(dst + 3, a[v][3]); }
            }
        }
    };
    if (base < total) {
        const int cnt = min(CH, total - base);
        const int e = base + lane;
        const int my_id = lane < cnt ? sid[e] : -1;
        const int my_pos = lane < cnt ? spos[e] : 0;
        const int prev_id = base > 0 ? sid[base - 1] : -1, next_id = base + cnt < total ? sid[base + cnt] : -1;      // (uniform)
        f32x4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
        int cur_id = __shfl(my_id, 0);
        auto flush = [&](int key) {
            if (key != prev_id && key != next_id) row_out(key, acc, true);      // the whole segment inside this chunk
            else {
                const int sl = wave * 2 + (key == prev_id ? 0 : 1);             // began before the chunk | goes on behind it
#pragma unroll
                for (int v = 0; v < NV; ++v) slot[sl][lane + 64 * v] = acc[v];
                if (lane == 0) slot_id[sl] = key;
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] = f32x4{0, 0, 0, 0};
        };
        for (int i = 0; i < cnt; i += FLY) {
            f32x4 val[FLY][NV];
            int ids[FLY];
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                const int ii = min(i + u, cnt - 1);
                ids[u] = __shfl(my_id, ii);
                const size_t pos = (size_t)__shfl(my_pos, ii);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const int f4 = lane + 64 * v;
                    val[u][v] = f4 < R4 ? dxt[pos * R4 + f4] : f32x4{0, 0, 0, 0};
                }
            }
#pragma unroll
            for (int u = 0; u < FLY; ++u) {
                if (i + u < cnt) {                                 // wave-uniform
                    if (ids[u] != cur_id) { flush(cur_id); cur_id = ids[u]; }
#pragma unroll
                    for (int v = 0; v < NV; ++v) acc[v] += val[u][v];
                }
            }
        }
        flush(cur_id);
    }
    __syncthreads();
    if (wave == 0) {       // runs of equal ids among the open partial rows, in entry order
        int cur = -1;
        f32x4 sum[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) sum[v] = f32x4{0, 0, 0, 0};
        auto out = [&](int key) { row_out(key, sum, key != wg_prev && key != wg_next); };      // a run inside the workgroup's entries: stored
        for (int sl = 0; sl < 2 * W; ++sl) {
            const int id = slot_id[sl];                            // (uniform)
            if (id < 0) continue;
            if (id != cur) {
                if (cur >= 0) out(cur);
                cur = id;
#pragma unroll
                for (int v = 0; v < NV; ++v) sum[v] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) sum[v] += slot[sl][lane + 64 * v];
        }
        if (cur >= 0) out(cur);
    }
}

hipError_t launch_scatter_reduce(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos,
                                 const int* offs, int n_ids, int max_entries, int GHp, int Bp, int key_lo, bool accumulate, int acc_chunk) {
    const int R4 = GHp / 4;
    const int chunk_env = 0;
    // narrow rows (at most 256 floats): 64 entries per wave -- what the launch waits for there is the float atomics of the chunks that hold
    // a piece of a hot id (every such chunk adds its partial row onto the same addresses), and twice the entries per chunk is half of them
    const int chunk = (accumulate || key_lo) ? (acc_chunk == 16 ? 16 : 32) : ((chunk_env == 16 || chunk_env == 32 || chunk_env == 64) ? chunk_env : (R4 <= 64 ? 64 : 32));
    const int chunks = (max_entries + chunk - 1) / chunk;
    const int grid = (chunks + 3) / 4;
    const int nv = (R4 + 63) / 64;
    // plain launch, rows up to 512 floats: the combining form, 8 waves x 32 entries per workgroup (alone on the chip: C2's shape 43.4 -> 21.5 us,
    // C1's 26.9 -> 14.8; C1's step 0.298 -> 0.286 ms: profiles/round6_variants.txt, call y).  (16 waves x 64 entries measured slower inside C1's
    // step than the walk below, which now serves the accumulating / keyed launches and wider rows only.)
    if (!accumulate && !key_lo && (nv <= 2 || nv == 4 || nv == 8) && max_entries > 0) {
        const int grid = (max_entries + 8 * COMB_CH - 1) / (8 * COMB_CH);
#define SRC(NV) do { const size_t lds = (size_t)16 * NV * 64 * sizeof(f32x4) + 64; SBR_DYN_LDS(scat_reduce_comb_kernel<NV>, lds); \
                     scat_reduce_comb_kernel<NV><<<grid, 512, lds, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4); } while (0)
        if (nv == 1) SRC(1); else if (nv == 2) SRC(2); else if (nv == 4) SRC(4); else SRC(8);
#undef SRC
        return hipGetLastError();
    }
#define SR(NV) do { if ((accumulate || key_lo) && chunk == 16) scat_reduce_kernel<NV, 16, true><<<grid, 256, 0, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4, Bp, key_lo); \
                    else if (accumulate || key_lo) scat_reduce_kernel<NV, 32, true><<<grid, 256, 0, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4, Bp, key_lo); \
                    else if (chunk == 16) scat_reduce_kernel<NV, 16><<<grid, 256, 0, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4, Bp); \
                    else if (chunk == 32) scat_reduce_kernel<NV, 32><<<grid, 256, 0, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4, Bp); \
                    else scat_reduce_kernel<NV, 64><<<grid, 256, 0, s>>>((const f32x4*)dxt, sid, spos, offs, n_ids, dWin, R4, Bp); } while (0)
    if (accumulate || key_lo) { if (nv <= 1) SR(1); else if (nv <= 2) SR(2); else if (nv <= 4) SR(4); else return hipErrorInvalidValue; }
    else if (nv <= 1) SR(1); else if (nv <= 2) SR(2); else if (nv <= 4) SR(4); else if (nv <= 8) SR(8); else if (nv <= 16) SR(16);
    else return hipErrorInvalidValue;
#undef SR
    return hipGetLastError();
}

// Gate of the overlapped step tail (sbr_backward_recurrent): returns once every wave of the running BPTT chain
// (rec_bwd_x6p<.., WT>) has published a progress word of this launch (epoch) at or below `target`, i.e. once all time
// steps >= target are complete AND written through to memory.  The kernels enqueued behind it on the same stream are
// ordinary consumers of those time steps.  One workgroup; the poll is a relaxed agent-scope load with s_sleep between
// polls; the spin is bounded (fault bit 3, reported with the cost) so that a chain that never starts cannot hang the GPU.
__global__ void __launch_bounds__(512) tail_gate_kernel(const int* __restrict__ progress, int n, int epoch, int target, int* __restrict__ fault) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            const int v = __hip_atomic_load(progress + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 12) == epoch && (v & 0xfff) <= target) break;
            if (wall_clock64() - t0 > SBR_POLL_TICKS) { atomicOr(fault, 8); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
}

// The MONITOR of an overlapped step tail (tail_monitor_loop: workgroup 0 of the LDS-row scatter-add launch; as this kernel on a stream
// of its own where that launch is not taken, and on the one stream of the serial mode): one workgroup folds the chain's per-wave progress words into
// `done` = (epoch << 12) | max t (SBR_DONE_COPIES copies, sbr_common.h SbrPoll) for as long as the chain runs -- relaxed
// agent-scope loads, stores only when the maximum moves -- and leaves when every wave has reached t_lo.  (Rounds 2 / 3a: a
// workgroup of the polling GEMM did this; the consumers then depended on WHEN that launch got onto the chip.)
__global__ void __launch_bounds__(256) tail_monitor_kernel(SbrPoll pl, int t_lo) { tail_monitor_loop(pl, t_lo); }
hipError_t launch_tail_monitor(hipStream_t s, const SbrPoll& poll, int t_lo) {
    tail_monitor_kernel<<<1, 256, 0, s>>>(poll, t_lo);
    return hipGetLastError();
}

hipError_t launch_tail_gate(hipStream_t s, const int* progress, int n, int epoch, int target, int* fault) {
    tail_gate_kernel<<<1, 512, 0, s>>>(progress, n, epoch, target, fault);
    return hipGetLastError();
}

// triage fallback: per-element float atomics (SBR_FLAG_ATOMIC_SCATTER)
__global__ void scatter_rows_kernel(float* __restrict__ dWin, const f32x4* __restrict__ dxt, const int* __restrict__ X,
                                    const int* __restrict__ len, int T, int Bp, int F, int R4) {
    const size_t total = (size_t)T * Bp * R4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int f4 = (int)(i % R4);
        const size_t pos = i / R4;
        const int b = (int)(pos % Bp), t = (int)(pos / Bp);
        if (t >= len[b]) continue;
        const f32x4 v = dxt[i];
        const int* ids = X + ((size_t)b * T + t) * F;
        for (int f = 0; f < F; ++f) {
            float* dst = dWin + ((size_t)ids[f] * R4 + f4) * 4;
            atomicAdd(dst + 0, v[0]); atomicAdd(dst + 1, v[1]); atomicAdd(dst + 2, v[2]); atomicAdd(dst + 3, v[3]);
        }
    }
}

hipError_t launch_scatter_rows(hipStream_t s, float* dWin, const float* dxt, const int* X, const int* len, int T,
                               int Bp, int F, int GHp) {
    const int R4 = GHp / 4;
    const size_t total = (size_t)T * Bp * R4;
    const int grid = (int)min((size_t)256 * 16, (total + 255) / 256);
    scatter_rows_kernel<<<grid, 256, 0, s>>>(dWin, (const f32x4*)dxt, X, len, T, Bp, F, R4);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K8 full softmax + categorical cross-entropy, forward and backward fused, one workgroup per
// row (rnn_one_hot.py:65-71): in: raw h.W_out; out (in place): dcost/dlogits.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) softmax_cce_kernel(float* __restrict__ logits, const float* __restrict__ bout,
                                                          const int* __restrict__ target, const float* __restrict__ pop,
                                                          float* __restrict__ rowcost, int N, long ld, int Bglobal) {
    __shared__ float red[4];
    const int r = blockIdx.x;
    float* x = logits + (size_t)r * ld;
    float mx = -INFINITY;
    for (int n = threadIdx.x; n < N; n += 256) { const float v = x[n] + bout[n]; x[n] = v; mx = fmaxf(mx, v); }
    mx = block_max(mx, red);
    float se = 0.0f;
    for (int n = threadIdx.x; n < N; n += 256) se += expf(x[n] - mx);
    se = block_sum(se, red);
    const int y = target[r];
    const float scale = 1.0f / (pop[r] * (float)Bglobal);
    const float xy = x[y];
    __syncthreads();
    const float inv = 1.0f / se;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float p = expf(x[n] - mx) * inv;
        x[n] = (p - (n == y ? 1.0f : 0.0f)) * scale;
    }
    if (threadIdx.x == 0) rowcost[r] = (logf(se) + mx - xy) * scale;
}

// Same, with the row held in LDS (N * 4 bytes <= ~150 KB, i.e. catalogues up to ~38 k items): one read and one write
// of the logits instead of three reads and two writes, 1024 threads per row, hardware exp2.  C4 (N = 26744): 133 -> ~30 us.
__global__ void __launch_bounds__(1024) softmax_cce_lds_kernel(float* __restrict__ logits, const float* __restrict__ bout,
                                                               const int* __restrict__ target, const float* __restrict__ pop,
                                                               float* __restrict__ rowcost, int N, long ld, int Bglobal) {
    extern __shared__ float row[];
    __shared__ float red[16];
    const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* x = logits + (size_t)r * ld;
    float mx = -INFINITY;
    for (int n = tid; n < N; n += 1024) { const float v = x[n] + bout[n]; row[n] = v; mx = fmaxf(mx, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
    __syncthreads();
    float se = 0.0f;
    for (int n = tid; n < N; n += 1024) { const float e = __builtin_amdgcn_exp2f((row[n] - mx) * 1.4426950408889634f); row[n] = e; se += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    if (lane == 0) red[wv] = se;
    const int y = target[r];
    const float scale = 1.0f / (pop[r] * (float)Bglobal);
    __syncthreads();
    se = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) se += red[w];                 // fixed order: deterministic
    const float inv = 1.0f / se;
    // cost = log(se) + mx - x_y with x_y recovered from e_y = exp(x_y - mx): keep it exact instead: re-read the logit
    const float xy = x[y] + bout[y];
    __syncthreads();                                           // everyone has read x[y] before the row is overwritten
    for (int n = tid; n < N; n += 1024) x[n] = (row[n] * inv - (n == y ? 1.0f : 0.0f)) * scale;
    if (tid == 0) rowcost[r] = (logf(se) + mx - xy) * scale;
}

hipError_t launch_softmax_cce(hipStream_t s, float* logits, const float* bout, const int* target, const float* pop,
                              float* rowcost, int rows, int N, long ld, int Bglobal) {
    if (rows <= 0) return hipSuccess;
    const size_t lds = (size_t)N * sizeof(float);
    if (lds <= 150 * 1024) {
        SBR_DYN_LDS(softmax_cce_lds_kernel, lds);
        softmax_cce_lds_kernel<<<rows, 1024, lds, s>>>(logits, bout, target, pop, rowcost, N, ld, Bglobal);
        return hipGetLastError();
    }
    softmax_cce_kernel<<<rows, 256, 0, s>>>(logits, bout, target, pop, rowcost, N, ld, Bglobal);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// RNNMargin's multi-target losses (rnn_margin.py:62-69), forward and backward fused, one workgroup per row.
// The reference builds dense (B, N) target / weight matrices on the host (:112-147); here a row is: the default target
// and the false-positive weight w = balance * n_pos / (N - n_pos - n_in) everywhere, then the overrides, few entries each
// and possibly repeated: the positives (target 1, weight -1), and LAST -- with unique interactions -- the row's input
// items (target 0, weight 0; an input item that is also a positive ends there).  The overridden logits are put aside
// before the dense pass overwrites the row with gradients; duplicates are recognised by comparing against the earlier
// entries (k <= T + NT entries in LDS).   loss / d loss / d p of one element:
//   hinge   relu((p - y) w)               | w if (p - y) w > 0
//   logit   sigmoid(p - y) w              | s (1 - s) w
//   logsig  -log(sigmoid((y - p) w))      | w (1 - sigmoid((y - p) w))
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void margin_elem(int loss, float p, float y, float w, float& l, float& g) {
    if (loss == SBR_LOSS_HINGE) { const float z = (p - y) * w; l = fmaxf(z, 0.0f); g = z > 0.0f ? w : 0.0f; }
    else if (loss == SBR_LOSS_LOGIT) { const float s = 1.0f / (1.0f + expf(y - p)); l = s * w; g = s * (1.0f - s) * w; }
    else { const float z = (y - p) * w; l = fmaxf(-z, 0.0f) + log1pf(expf(-fabsf(z))); g = w * (1.0f - 1.0f / (1.0f + expf(-z))); }
}

__global__ void __launch_bounds__(256) margin_loss_kernel(float* __restrict__ logits, const float* __restrict__ bout,
                                                          const int* __restrict__ target, int NT, const int* __restrict__ X,
                                                          const int* __restrict__ len, int T, int F, const float* __restrict__ dflt,
                                                          float* __restrict__ rowcost, int N, long ld, int Bglobal, int loss,
                                                          float balance, int unique) {
    extern __shared__ int m_ids[];                 // [NT + T] ids of the override entries, then their logits (floats)
    __shared__ float red[4];
    const int r = blockIdx.x, tid = threadIdx.x;
    float* x = logits + (size_t)r * ld;
    float* m_p = (float*)(m_ids + NT + T);
    const int n_in = len[r];
    int nt = 0;
    for (int j = 0; j < NT; ++j) nt += target[(size_t)r * NT + j] >= 0 ? 1 : 0;
    const int k = nt + (unique ? n_in : 0);
    for (int e = tid; e < k; e += 256) {
        const int id = e < nt ? target[(size_t)r * NT + e] : X[((size_t)r * T + (e - nt)) * F];
        m_ids[e] = id;
        m_p[e] = x[id] + bout[id];
    }
    __syncthreads();
    const float w_fp = balance * (float)nt / (float)(N - nt - n_in);
    const float inv = 1.0f / (float)Bglobal;
    float acc = 0.0f;
    for (int n = tid; n < N; n += 256) {
        float l, g;
        margin_elem(loss, x[n] + bout[n], dflt ? dflt[n] : 0.0f, w_fp, l, g);
        x[n] = g * inv;
        acc += l;
    }
    __syncthreads();
    for (int e = tid; e < k; e += 256) {
        const int id = m_ids[e];
        const bool seen = e >= nt;
        bool dup = false;
        for (int f = seen ? nt : 0; f < e; ++f) dup = dup || m_ids[f] == id;         // an earlier entry of the same kind
        if (!seen && unique) for (int f = nt; f < k; ++f) dup = dup || m_ids[f] == id;   // a positive that is also an input item
        if (dup) continue;
        float l0, g0, l1, g1;
        margin_elem(loss, m_p[e], dflt ? dflt[id] : 0.0f, w_fp, l0, g0);
        margin_elem(loss, m_p[e], seen ? 0.0f : 1.0f, seen ? 0.0f : -1.0f, l1, g1);
        x[id] = g1 * inv;
        acc += l1 - l0;
    }
    acc = block_sum(acc, red);
    if (tid == 0) rowcost[r] = acc * inv;
}

hipError_t launch_margin_loss(hipStream_t s, float* logits, const float* bout, const int* target, int NT, const int* X,
                              const int* len, int T, int F, const float* dflt, float* rowcost, int rows, int N, long ld,
                              int Bglobal, int loss, float balance, int unique) {
    if (rows <= 0) return hipSuccess;
    const size_t lds = (size_t)(NT + T) * (sizeof(int) + sizeof(float));
    SBR_DYN_LDS(margin_loss_kernel, lds);
    margin_loss_kernel<<<rows, 256, lds, s>>>(logits, bout, target, NT, X, len, T, F, dflt, rowcost, N, ld, Bglobal, loss, balance, unique);
    return hipGetLastError();
}

// predict path: x += b, optional softmax (rnn_base.py:188-194, rnn_sampling.py:144)
__global__ void __launch_bounds__(256) softmax_rows_kernel(float* __restrict__ logits, const float* __restrict__ bout, int N,
                                                           int do_softmax) {
    __shared__ float red[4];
    float* x = logits + (size_t)blockIdx.x * N;
    float mx = -INFINITY;
    for (int n = threadIdx.x; n < N; n += 256) { const float v = x[n] + bout[n]; x[n] = v; mx = fmaxf(mx, v); }
    if (!do_softmax) return;
    mx = block_max(mx, red);
    float se = 0.0f;
    for (int n = threadIdx.x; n < N; n += 256) se += expf(x[n] - mx);
    se = block_sum(se, red);
    const float inv = 1.0f / se;
    for (int n = threadIdx.x; n < N; n += 256) x[n] = expf(x[n] - mx) * inv;
}

hipError_t launch_softmax_rows(hipStream_t s, float* logits, const float* bout, int rows, int N, int do_softmax) {
    if (rows <= 0) return hipSuccess;
    softmax_rows_kernel<<<rows, 256, 0, s>>>(logits, bout, N, do_softmax);
    return hipGetLastError();
}

// db[n] = sum_r d[r][n] (+ d reg/db); rnn_one_hot.py:73-77 regularises the output bias only.
// Two stages (rows split over gridDim.y, partials summed in fixed order): deterministic, and
// enough workgroups to stream the (rows, N) matrix at HBM speed.
#define COLSUM_SPLIT 16
__global__ void colsum_partial_kernel(const float* __restrict__ d, int rows, int N, long ld, float* __restrict__ part) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int per = (rows + COLSUM_SPLIT - 1) / COLSUM_SPLIT;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    float s = 0.0f;
    for (int r = r0; r < r1; ++r) s += d[(size_t)r * ld + n];
    part[(size_t)blockIdx.y * N + n] = s;
}
__global__ void colsum_bias_kernel(const float* __restrict__ part, int N, float* __restrict__ db,
                                   const float* __restrict__ b, float reg) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < COLSUM_SPLIT; ++k) s += part[(size_t)k * N + n];
    if (reg > 0.0f) s += 2.0f * reg * b[n];
    else if (reg < 0.0f) s -= reg * (b[n] > 0.0f ? 1.0f : (b[n] < 0.0f ? -1.0f : 0.0f));
    db[n] = s;
}
// cost += reg * sum(b^2)  or  |reg| * sum|b|   (single workgroup, fixed order)
__global__ void __launch_bounds__(256) reg_cost_kernel(const float* __restrict__ b, int N, float reg, float* cost) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int n = threadIdx.x; n < N; n += 256) s += reg > 0.0f ? b[n] * b[n] : fabsf(b[n]);
    s = block_sum(s, red);
    if (threadIdx.x == 0) *cost += fabsf(reg) * s;
}

hipError_t launch_colsum_bias(hipStream_t s, const float* d, int rows, int N, long ld, float* db, const float* b,
                              float reg, float* cost, float* ws) {
    colsum_partial_kernel<<<dim3((N + 255) / 256, COLSUM_SPLIT), 256, 0, s>>>(d, rows, N, ld, ws);
    colsum_bias_kernel<<<(N + 255) / 256, 256, 0, s>>>(ws, N, db, b, reg);
    if (reg != 0.0f && cost) reg_cost_kernel<<<1, 256, 0, s>>>(b, N, reg, cost);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) sum_cost_kernel(const float* __restrict__ rowcost, int rows, float* cost) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int r = threadIdx.x; r < rows; r += 256) s += rowcost[r];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *cost = s;
}
hipError_t launch_sum_cost(hipStream_t s, const float* rowcost, int rows, float* cost) {
    sum_cost_kernel<<<1, 256, 0, s>>>(rowcost, rows, cost);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Sampled heads (BlackoutLayer, sparse_lstm.py:42-54; losses rnn_sampling.py:68-91)
// ---------------------------------------------------------------------------------------
__global__ void build_cells_kernel(const int* target, const int* samples, int Bg, int S, int* cells) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < Bg + S) cells[c] = c < Bg ? target[c] : samples[c - Bg];   // T.concatenate((targets, output_cells)) :50
}
hipError_t launch_build_cells(hipStream_t s, const int* target, const int* samples, int Bg, int S, int* cells) {
    build_cells_kernel<<<(Bg + S + 255) / 256, 256, 0, s>>>(target, samples, Bg, S, cells);
    return hipGetLastError();
}

// W_out is stored item-major [N][Hp], so W[:, cells] (sparse_lstm.py:52) is a coalesced row gather
__global__ void gather_rows_kernel(const f32x4* __restrict__ W, const float* __restrict__ b, const int* __restrict__ cells,
                                   int C, int R4, f32x4* __restrict__ Wc, float* __restrict__ bc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * R4) return;
    const int c = i / R4, f4 = i % R4;
    Wc[i] = W[(size_t)cells[c] * R4 + f4];
    if (f4 == 0) bc[c] = b[cells[c]];
}
hipError_t launch_gather_rows(hipStream_t s, const float* W, const float* b, const int* cells, int C, int Hp, float* Wc,
                              float* bc) {
    const int R4 = Hp / 4;
    gather_rows_kernel<<<(C * R4 + 255) / 256, 256, 0, s>>>((const f32x4*)W, b, cells, C, R4, (f32x4*)Wc, bc);
    return hipGetLastError();
}

__device__ __forceinline__ float sigmf(float x) { return 1.0f / (1.0f + expf(-x)); }

// one workgroup per local row r; act row (C = Bg+S columns) in: h.Wc^T, out: dcost/dact
__global__ void __launch_bounds__(256) sampled_loss_kernel(float* __restrict__ act, const float* __restrict__ bc,
                                                           const float* __restrict__ pop, float* __restrict__ rowcost,
                                                           int Bg, int S, int row_offset, int loss, int Bglobal) {
    __shared__ float red[4];
    const int r = blockIdx.x, C = Bg + S, pos = row_offset + r;
    float* a = act + (size_t)r * C;
    const float scale = 1.0f / (pop[r] * (float)Bglobal);
    if (bc) { for (int c = threadIdx.x; c < C; c += 256) a[c] += bc[c]; }      // + b[output_cells] :54 (the cluster head's scores have none)
    __syncthreads();
    const float apos = a[pos];
    float L;
    if (loss == SBR_LOSS_BLACKOUT || loss == SBR_LOSS_SCCE) {         // rnn_sampling.py:68-72; SCCE: rnn_cluster.py:158-162
        const bool blackout = loss == SBR_LOSS_BLACKOUT;
        float mx = -INFINITY;
        for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, a[c]);
        mx = block_max(mx, red);
        float se = 0.0f;
        for (int c = threadIdx.x; c < C; c += 256) se += expf(a[c] - mx);
        se = block_sum(se, red);
        const float inv = 1.0f / se;
        const float ppos = expf(apos - mx) * inv;
        // sum_k dLdp_k p_k  and  sum of -log(1-p) over the negatives
        float dot = 0.0f, lneg = 0.0f;
        for (int c = threadIdx.x; c < C; c += 256) {
            const float p = expf(a[c] - mx) * inv;
            if (c >= Bg && blackout) { dot += p / (1.0f - p); lneg -= logf(1.0f - p); }
        }
        dot = block_sum(dot, red) - 1.0f;                             // positive: (-1/p_pos) * p_pos
        lneg = block_sum(lneg, red);
        L = -logf(ppos) + lneg;
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            const float p = expf(a[c] - mx) * inv;
            float dldp = 0.0f;
            if (c >= Bg && blackout) dldp = 1.0f / (1.0f - p);
            if (c == pos) dldp += -1.0f / p;
            a[c] = p * (dldp - dot) * scale;
        }
    } else {
        float lsum = 0.0f, dpos = 0.0f;
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            if (c == pos) continue;                                   // written after the reduction
            float da = 0.0f;
            if (c >= Bg) {
                const float v = a[c], diff = v - apos;
                if (loss == SBR_LOSS_BPR) {                           // :80-84  -log(sigmoid(-diff)) = softplus(diff)
                    lsum += fmaxf(diff, 0.0f) + log1pf(expf(-fabsf(diff)));
                    const float dd = sigmf(diff) / (float)S;
                    da = dd; dpos += dd;
                } else if (loss == SBR_LOSS_BPRELU) {                 // rnn_cluster.py:173-175: leaky_rectify(diff + 0.5), leakiness 0.01 [3P]
                    const float yv = diff + 0.5f;
                    lsum += yv > 0.0f ? yv : 0.01f * yv;
                    const float dd = (yv > 0.0f ? 1.0f : 0.01f) / (float)S;
                    da = dd; dpos += dd;
                } else if (loss == SBR_LOSS_LIN) {                    // rnn_cluster.py:164-167: SUM of the negatives - the positive
                    lsum += v;
                    da = 1.0f;
                } else {                                              // TOP1 :86-91
                    const float s1 = sigmf(diff), s2 = sigmf(v * v);
                    lsum += s1 + s2;
                    const float d1 = s1 * (1.0f - s1) / (float)S;
                    da = d1 + s2 * (1.0f - s2) * 2.0f * v / (float)S; dpos += d1;
                }
            }
            a[c] = da * scale;
        }
        lsum = block_sum(lsum, red);
        dpos = block_sum(dpos, red);
        L = lsum / (float)S;
        if (loss == SBR_LOSS_LIN) { L = lsum - apos; dpos = 1.0f; }
        if (threadIdx.x == 0) a[pos] = -dpos * scale;
    }
    if (threadIdx.x == 0) rowcost[r] = L * scale;
}

hipError_t launch_sampled_loss(hipStream_t s, float* act, const float* bc, const float* pop, float* rowcost, int rows,
                               int Bg, int S, int row_offset, int loss, int Bglobal) {
    if (rows <= 0) return hipSuccess;
    sampled_loss_kernel<<<rows, 256, 0, s>>>(act, bc, pop, rowcost, Bg, S, row_offset, loss, Bglobal);
    return hipGetLastError();
}

// dW_out^T[cells[c]][:] += dWc[c][:], db_out[cells[c]] += dbc[c]; duplicates accumulate [3P]
__global__ void scatter_cells_kernel(float* __restrict__ dW, float* __restrict__ db, const float* __restrict__ dWc,
                                     const float* __restrict__ dbc, const int* __restrict__ cells, int C, int Hp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * Hp) return;
    const int c = i / Hp, k = i % Hp;
    atomicAdd(&dW[(size_t)cells[c] * Hp + k], dWc[i]);
    if (k == 0) atomicAdd(&db[cells[c]], dbc[c]);
}
hipError_t launch_scatter_cells(hipStream_t s, float* dW, float* db, const float* dWc, const float* dbc, const int* cells,
                                int C, int Hp) {
    scatter_cells_kernel<<<(C * Hp + 255) / 256, 256, 0, s>>>(dW, db, dWc, dbc, cells, C, Hp);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K13 optimizers: lasagne.updates.{adagrad,adadelta,rmsprop,nesterov_momentum,adam} [3P]
// (update_manager.py:24-82), applied densely to the whole flat parameter section.
// ---------------------------------------------------------------------------------------
// one element's step: gi = its gradient; p, s0, s1 are read and written in place
__device__ __forceinline__ void update_element(int updater, float gi, float* __restrict__ p, float* __restrict__ s0, float* __restrict__ s1,
                                               size_t i, float lr, float rho, float b1, float b2, float a_t) {
    float pi = p[i];
    if (updater == SBR_UPD_ADAGRAD) {                 // eps 1e-6
        const float acc = s0[i] + gi * gi;
        s0[i] = acc; pi -= lr * gi / sqrtf(acc + 1e-6f);
    } else if (updater == SBR_UPD_RMSPROP) {          // eps 1e-6
        const float acc = rho * s0[i] + (1.0f - rho) * gi * gi;
        s0[i] = acc; pi -= lr * gi / sqrtf(acc + 1e-6f);
    } else if (updater == SBR_UPD_ADADELTA) {         // eps 1e-6
        const float acc = rho * s0[i] + (1.0f - rho) * gi * gi;
        const float upd = gi * sqrtf(s1[i] + 1e-6f) / sqrtf(acc + 1e-6f);
        s0[i] = acc; pi -= lr * upd;
        s1[i] = rho * s1[i] + (1.0f - rho) * upd * upd;
    } else if (updater == SBR_UPD_NESTEROV) {         // sgd + apply_nesterov_momentum
        const float v = rho * s0[i] - lr * gi;
        s0[i] = v; pi += rho * v - lr * gi;
    } else {                                          // adam, eps 1e-8
        const float m = b1 * s0[i] + (1.0f - b1) * gi;
        const float v = b2 * s1[i] + (1.0f - b2) * gi * gi;
        s0[i] = m; s1[i] = v;
        pi -= a_t * m / (sqrtf(v) + 1e-8f);
    }
    p[i] = pi;
}

__global__ void update_kernel(int updater, float* __restrict__ p, float* __restrict__ g, float* __restrict__ s0,
                              float* __restrict__ s1, size_t n, float lr, float rho, float b1, float b2, float a_t,
                              size_t gap_at, size_t gap_len) {
    // n elements of [0, gap_at) u [gap_at + gap_len, ...): two parameter ranges in one launch
    for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
        const size_t i = k < gap_at ? k : k + gap_len;
        const float gi = g[i];
        g[i] = 0.0f;                                      // the gradient section is clean for the next step (no memset)
        update_element(updater, gi, p, s0, s1, i, lr, rho, b1, b2, a_t);
    }
}

// The optimizer step of a parameter block whose gradient still lies in split-K slabs (dW_hid of the overlapped step tail): the
// slab reduction of gemm_splitk_reduce_v4 -- same grouping, same summation order, so the same gradient bits -- with the step
// applied by the thread that holds the finished sum.  One launch and one pass over the block less at the end of the step; the
// gradient array itself is not written (it stays as the last update left it: zero).
__global__ void __launch_bounds__(256) update_from_slabs_kernel(int updater, const f32x4* __restrict__ ws, int nsplit, size_t n,
                                                                float* __restrict__ p, float* __restrict__ s0, float* __restrict__ s1,
                                                                float lr, float rho, float b1, float b2, float a_t) {
    __shared__ f32x4 red[16][16];
    const size_t n4 = n / 4;
    const int j = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t i4 = (size_t)blockIdx.x * 16 + j;
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    if (i4 < n4) {
        int z = grp;
        for (; z + 48 < nsplit; z += 64) {
            a0 += ws[(size_t)z * n4 + i4]; a1 += ws[(size_t)(z + 16) * n4 + i4];
            a2 += ws[(size_t)(z + 32) * n4 + i4]; a3 += ws[(size_t)(z + 48) * n4 + i4];
        }
        for (; z < nsplit; z += 16) a0 += ws[(size_t)z * n4 + i4];
    }
    red[grp][j] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int jj = threadIdx.x >> 2, e = threadIdx.x & 3;             // one element per thread
    const size_t i = ((size_t)blockIdx.x * 16 + jj) * 4 + e;
    if (i >= n) return;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][jj][e];
    update_element(updater, t, p, s0, s1, i, lr, rho, b1, b2, a_t);
}

hipError_t launch_update_from_slabs(hipStream_t s, int updater, const float* ws, int nslabs, float* p, float* s0, float* s1, size_t n,
                                    float lr, float rho, float b1, float b2, long t) {
    if (n == 0) return hipSuccess;
    if ((n & 3) || ((uintptr_t)ws & 15)) return hipErrorInvalidValue;
    float a_t = 0.0f;
    if (updater == SBR_UPD_ADAM)
        a_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
    update_from_slabs_kernel<<<(unsigned)((n / 4 + 15) / 16), 256, 0, s>>>(updater, (const f32x4*)ws, nslabs, n, p, s0, s1, lr, rho, b1, b2, a_t);
    return hipGetLastError();
}

// The dense pass over an item-indexed block, aware of which rows the batch touched (offs = the scatter's segment offsets): an
// untouched row's gradient is zero and stays zero, so its step reads and writes p, s0, s1 only (6 passes instead of 8: C4 touches 40 %
// of its 26 744 rows per batch); a touched row's gradient is read and cleared as update_kernel does.  One streaming launch, 16
// bytes per lane, update_element's arithmetic for both kinds of row.
__global__ void __launch_bounds__(256) update_rows_aware_kernel(SbrScatStep st, float* __restrict__ gr, int n_rows, int R4,
                                                                const int* __restrict__ offs) {
    const size_t n4 = (size_t)n_rows * R4;
    f32x4* __restrict__ p = (f32x4*)st.p; f32x4* __restrict__ s0 = (f32x4*)st.s0; f32x4* __restrict__ s1 = (f32x4*)st.s1;
    f32x4* __restrict__ g = (f32x4*)gr;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / (unsigned)R4);
        const bool touched = offs[r + 1] > offs[r];
        f32x4 gv = {0, 0, 0, 0};
        if (touched) { gv = g[i]; g[i] = f32x4{0, 0, 0, 0}; }
        f32x4 pv = p[i], av = s0[i], bv = s1 ? s1[i] : f32x4{0, 0, 0, 0};
        scat_step4(st, gv, pv, av, bv);
        p[i] = pv; s0[i] = av;
        if (s1) s1[i] = bv;
    }
}
hipError_t launch_update_rows_aware(hipStream_t s, int updater, float* p, float* g, float* s0, float* s1, int n_rows, int row_floats,
                                    const int* offs, float lr, float rho, float b1, float b2, long t) {
    if (n_rows <= 0) return hipSuccess;
    if ((row_floats & 3) || !offs) return hipErrorInvalidValue;
    SbrScatStep st; st.p = p; st.s0 = s0; st.s1 = s1; st.last = nullptr; st.updater = updater; st.t_to = (int)t;
    st.lr = lr; st.rho = rho; st.b1 = b1; st.b2 = b2; st.a_t = 0.0f;
    if (updater == SBR_UPD_ADAM)
        st.a_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
    const size_t n4 = (size_t)n_rows * (row_floats / 4);
    const int grid = (int)min((size_t)256 * 16, (n4 + 255) / 256);
    update_rows_aware_kernel<<<grid, 256, 0, s>>>(st, g, n_rows, row_floats / 4, offs);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// The output layer's gradient AND its optimizer step in one launch (round 5; single-call steps of the dense heads, no bias
// regulariser): dW_out^T[n][k] = sum_rows dlogits[row][n] h[row][k], db[n] = sum_rows dlogits[row][n], cost = sum_rows rowcost
// (rnn_one_hot.py:65-77 backward + update_manager.py:24-82).  Before: sum_cost, two column-sum kernels, the dW_out GEMM (+ its
// split-K reduction) and update_kernel -- five to six launches, 50 - 60 us with their gaps on the side stream, IN FRONT of the
// polling weight-gradient GEMM of the overlapped tail, which therefore started 68 us into a 138 us BPTT chain and ended 39 us
// behind it (profiles/round4_z_c2_timeline.txt).  Here a workgroup owns 16 items: its four waves split the batch rows, every
// wave accumulates D[k][item] over its rows on v_mfma_f32_16x16x4_f32 (k slot q = row r0 + q; operands straight from L2: 64
// contiguous bytes per 16 lanes), the partial sums meet in LDS, and the thread that holds four consecutive k of an item steps
// W_out^T[item][k .. k + 3] (update_element's arithmetic: scat_step4); the gradient arrays are never written (they stay zero).
// ---------------------------------------------------------------------------------------
template <int HP>
__global__ void __launch_bounds__(256) out_grad_step_kernel(const float* __restrict__ dlog, const float* __restrict__ h,
                                                            const float* __restrict__ rowcost, float* __restrict__ cost, SbrScatStep st,
                                                            float* __restrict__ bp, float* __restrict__ bs0, float* __restrict__ bs1,
                                                            int R, int N, int Nl) {
    constexpr int KG = HP / 16;
    __shared__ __attribute__((aligned(16))) f32x4 part[4][KG][64];
    __shared__ float dbp[4][16];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const bool col_in = n0 + j < N;
    const int rpw = ((R + 3) / 4 + 3) / 4 * 4;                       // rows per wave, a multiple of 4
    const int r_lo = wave * rpw, r_hi = min(R, r_lo + rpw);
    f32x4 acc[KG];
#pragma unroll
    for (int kt = 0; kt < KG; ++kt) acc[kt] = f32x4{0, 0, 0, 0};
    float dbv = 0.0f;
    for (int r0 = r_lo; r0 < r_hi; r0 += 16) {                      // four row groups per round: their 4 + 4 KG loads in flight together
        float d[4], hv[4][KG];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = r0 + 4 * u + q;
            const bool in = row < r_hi;
            d[u] = (in && col_in) ? dlog[(size_t)row * Nl + n0 + j] : 0.0f;
            const float* hr = h + (size_t)(in ? row : r_lo) * HP + j;
#pragma unroll
            for (int kt = 0; kt < KG; ++kt) hv[u][kt] = in ? hr[16 * kt] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            dbv += d[u];
#pragma unroll
            for (int kt = 0; kt < KG; ++kt) acc[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[u][kt], d[u], acc[kt], 0, 0, 0);
        }
    }
    asm volatile("s_nop 15");                                        // MFMA D -> VALU read across the loop exit
#pragma unroll
    for (int kt = 0; kt < KG; ++kt) part[wave][kt][lane] = acc[kt];
    dbv += __shfl_xor(dbv, 16); dbv += __shfl_xor(dbv, 32);
    if (q == 0) dbp[wave][j] = dbv;
    if (blockIdx.x == 0) {                                           // the batch cost: fixed order
        float c = 0.0f;
        for (int r = tid; r < R; r += 256) c += rowcost[r];
        c = block_sum(c, red);
        if (tid == 0) *cost = c;
    }
    __syncthreads();
    for (int i = tid; i < KG * 64; i += 256) {
        const int kt = i >> 6, ln = i & 63, item = n0 + (ln & 15), k = 16 * kt + 4 * (ln >> 4);
        if (item >= N) continue;
        const f32x4 g = (part[0][kt][ln] + part[1][kt][ln]) + (part[2][kt][ln] + part[3][kt][ln]);
        const size_t o = (size_t)item * HP + k;
        f32x4 pv = *(const f32x4*)(st.p + o), av = *(const f32x4*)(st.s0 + o), bv = st.s1 ? *(const f32x4*)(st.s1 + o) : f32x4{0, 0, 0, 0};
        scat_step4(st, g, pv, av, bv);
        *(f32x4*)(st.p + o) = pv; *(f32x4*)(st.s0 + o) = av;
        if (st.s1) *(f32x4*)(st.s1 + o) = bv;
    }
    if (tid < 16 && n0 + tid < N) {
        const float g = (dbp[0][tid] + dbp[1][tid]) + (dbp[2][tid] + dbp[3][tid]);
        float pe = bp[n0 + tid], ae = bs0[n0 + tid], be = bs1 ? bs1[n0 + tid] : 0.0f;
        scat_step1(st, g, pe, ae, be);
        bp[n0 + tid] = pe; bs0[n0 + tid] = ae;
        if (bs1) bs1[n0 + tid] = be;
    }
}

// W: W_out^T [N][Hp] with its state arrays (s1 NULL: one state array), b: b_out [N] likewise; false: shape not served
bool launch_out_grad_step(hipStream_t s, const float* dlogits, const float* h_last, const float* rowcost, float* cost, int updater,
                          float* W, float* Ws0, float* Ws1, float* b, float* bs0, float* bs1, int R, int N, int Nl, int Hp, float lr,
                          float rho, float b1, float b2, long t, hipError_t* err) {
    if (!(Hp == 32 || Hp == 64 || Hp == 128) || R < 1 || N < 1) return false;
    SbrScatStep st; st.p = W; st.s0 = Ws0; st.s1 = Ws1; st.last = nullptr; st.updater = updater; st.t_to = (int)t;
    st.lr = lr; st.rho = rho; st.b1 = b1; st.b2 = b2; st.a_t = 0.0f;
    if (updater == SBR_UPD_ADAM)
        st.a_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
    const int grid = (N + 15) / 16;
    if (Hp == 128) out_grad_step_kernel<128><<<grid, 256, 0, s>>>(dlogits, h_last, rowcost, cost, st, b, bs0, bs1, R, N, Nl);
    else if (Hp == 64) out_grad_step_kernel<64><<<grid, 256, 0, s>>>(dlogits, h_last, rowcost, cost, st, b, bs0, bs1, R, N, Nl);
    else out_grad_step_kernel<32><<<grid, 256, 0, s>>>(dlogits, h_last, rowcost, cost, st, b, bs0, bs1, R, N, Nl);
    *err = hipGetLastError();
    return true;
}

hipError_t launch_update(hipStream_t s, int updater, float* p, float* g, float* s0, float* s1, size_t n, float lr,
                         float rho, float b1, float b2, long t, size_t gap_at, size_t gap_len) {
    if (n == 0) return hipSuccess;
    float a_t = 0.0f;
    if (updater == SBR_UPD_ADAM)
        a_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
    const int grid = (int)min((size_t)256 * 16, (n + 255) / 256);
    update_kernel<<<grid, 256, 0, s>>>(updater, p, g, s0, s1, n, lr, rho, b1, b2, a_t, gap_at, gap_len);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// K14 test path: exclude seen items, ordered top-k (rnn_base.py:196-211)
// ---------------------------------------------------------------------------------------
__global__ void exclude_seen_kernel(float* __restrict__ scores, const int* __restrict__ X, const int* __restrict__ len,
                                    int rows, int T, int F, int N, float value) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * T) return;
    const int r = i / T, t = i % T;
    if (t < len[r]) scores[(size_t)r * N + X[((size_t)r * T + t) * F]] = value;   // exclude[i, item ids] = 1
}
hipError_t launch_exclude_seen(hipStream_t s, float* scores, const int* X, const int* len, int rows, int T, int F, int N, float value) {
    if (rows <= 0) return hipSuccess;
    exclude_seen_kernel<<<(rows * T + 255) / 256, 256, 0, s>>>(scores, X, len, rows, T, F, N, value);
    return hipGetLastError();
}

// k rounds of arg-max per row; ties -> lowest id; destroys the row (winners set to -inf).  Only finite scores and +inf are
// candidates (NaN never wins a comparison and -inf is an excluded or already emitted item): when a row runs out of
// candidates -- a user who has seen more than N - k items with exclude_seen, or NaN scores -- the remaining places are
// filled with -1 instead of re-emitting an excluded id or an out-of-range one.
__global__ void __launch_bounds__(256) topk_kernel(float* __restrict__ scores, int N, int k, int* __restrict__ ids) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    float* x = scores + (size_t)blockIdx.x * N;
    for (int it = 0; it < k; ++it) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int n = threadIdx.x; n < N; n += 256) {
            const float v = x[n];
            if (v > bv || (v == bv && n < bi && v > -INFINITY)) { bv = v; bi = n; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = bv; ri[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w)
                if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
            ids[(size_t)blockIdx.x * k + it] = bi < N ? bi : -1;
            if (bi < N) x[bi] = -INFINITY;
        }
        __syncthreads();
    }
}
hipError_t launch_topk(hipStream_t s, float* scores, int rows, int N, int k, int* ids) {
    if (rows <= 0) return hipSuccess;
    topk_kernel<<<rows, 256, 0, s>>>(scores, N, k, ids);
    return hipGetLastError();
}

__global__ void fill_kernel(float* p, float v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
hipError_t launch_fill(hipStream_t s, float* p, float v, size_t n) {
    const int grid = (int)min((size_t)256 * 16, (n + 255) / 256);
    fill_kernel<<<grid, 256, 0, s>>>(p, v, n);
    return hipGetLastError();
}
