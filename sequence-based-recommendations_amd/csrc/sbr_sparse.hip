// Row-sparse optimizer steps and gradient exchange for the large-catalogue parameter blocks.
//
// The reference applies lasagne.updates.* densely to every parameter (update_manager.py:24-82): at 100 k - 1 M items
// that is an elementwise pass over the whole item-major W_in [input_size][G*H] (and, for the sampled heads, W_out
// [N][H] + b_out) per step -- 3.6 GB (C3) to 72 GB (C5) of HBM traffic for rows whose gradient is exactly zero:
// only the rows the batch gathers (sparse_lstm.py:368) and the sampled cells (sparse_lstm.py:50-54) receive one.
//
// What a step does to a row with zero gradient, per updater (same formulas as update_kernel in sbr_misc.hip):
//   adagrad   nothing (acc += 0, p -= lr * 0 / sqrt(acc + eps))                         -> skipping is EXACT
//   rmsprop   acc *= rho                                                                -> k skipped steps: acc *= rho^k
//   adadelta  acc *= rho, delta *= rho, p unchanged                                     -> both *= rho^k
//   nesterov  v *= rho; p += rho * v                                                    -> geometric sum
//   adam      m *= b1; v *= b2; p -= a_t * m / (sqrt(v) + eps)   (the row keeps moving) -> replayed step by step
// Every row of a sparse block carries `last[row]` = the step through which it is current.  Before a row is read
// (gathered by the forward pass, ranked by predict / top-k, exported by get_params) or stepped with a real gradient, it
// is CAUGHT UP: the zero-gradient steps it missed are replayed -- Adam literally, one (m, v, p) update per missed step
// with that step's own a_t, in the dense kernel's arithmetic, until the update is smaller than half an ulp of p and
// provably stays so (then only m and v still change: one pow each); the others in closed form (a short loop for small k
// so that the result is the dense kernel's bit for bit, pow for the long gaps).  "Lazy-exact": the dense oracle is
// matched to float32 rounding over runs that touch rows intermittently (tests/test_gpu_sparse_update.py).
//
// Candidates are lists of row ids WITH duplicates (the batch's item ids, the sampled cells, the ids gathered from other
// ranks): the first wave lane to raise last[id] with atomicMax owns the row, the others skip it.
#include "sbr_common.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SpUpd {
    int updater;
    float lr, rho, b1, b2;
    const float* at;     // adam: a_t = lr * sqrt(1 - b2^t) / (1 - b1^t) for t = 1..n_at (host double -> float, as the dense launch computes it)
    int n_at;            // beyond the table a_t == (float)lr
    int early_exit;      // adam: a missed step's update shrinks monotonically (checked on the host from b1, b2 and the table)
};

__device__ __forceinline__ float sp_at(const SpUpd& u, long t) { return t <= (long)u.n_at ? u.at[t - 1] : u.lr; }

// k zero-gradient steps t0+1 .. t0+k on the NE elements a lane holds of one row.  The elements advance together, step by
// step, so that their dependent chains (mul, sqrt, div, sub per step) interleave: one element at a time the replay is
// latency-bound (measured 264 us for C3's batch rows with gaps <= 25; the loop below issues NE independent chains).
template <int NE>
__device__ __forceinline__ void sp_catch_up(const SpUpd& u, float (&p)[NE], float (&s0)[NE], float (&s1)[NE], int k, long t0) {
    if (k <= 0) return;
    switch (u.updater) {
        case SBR_UPD_ADAGRAD: return;
        case SBR_UPD_RMSPROP:
            if (k <= 32) { for (int j = 0; j < k; ++j) _Pragma("unroll") for (int e = 0; e < NE; ++e) s0[e] = u.rho * s0[e]; }
            else { const float f = powf(u.rho, (float)k); _Pragma("unroll") for (int e = 0; e < NE; ++e) s0[e] *= f; }
            return;
        case SBR_UPD_ADADELTA:
            if (k <= 32) { for (int j = 0; j < k; ++j) _Pragma("unroll") for (int e = 0; e < NE; ++e) { s0[e] = u.rho * s0[e]; s1[e] = u.rho * s1[e]; } }
            else { const float f = powf(u.rho, (float)k); _Pragma("unroll") for (int e = 0; e < NE; ++e) { s0[e] *= f; s1[e] *= f; } }
            return;
        case SBR_UPD_NESTEROV:
            if (k <= 32) {
                for (int j = 0; j < k; ++j)
                    _Pragma("unroll") for (int e = 0; e < NE; ++e) { const float v = u.rho * s0[e]; s0[e] = v; p[e] += u.rho * v; }
            } else {    // p += rho * sum_{j=1..k} rho^j v0 ; v = rho^k v0
                const float rk = powf(u.rho, (float)k);
                const float f = u.rho * u.rho * (1.0f - rk) / (1.0f - u.rho);
                _Pragma("unroll") for (int e = 0; e < NE; ++e) { p[e] += s0[e] * f; s0[e] *= rk; }
            }
            return;
        default: {      // adam: replay until every element's update is below half an ulp of its p (and shrinking) or m is gone.
            // Per missed step and element: m *= b1; the root of v advances by sqrt(b2) (one multiply instead of a square
            // root: the relative drift after j steps is <= j * 6e-8, on an update that is itself a vanishing share of p); one
            // hardware reciprocal (1 ulp).  The dense kernel's sqrtf + division cost ~3x the instructions here, where -- unlike
            // there -- arithmetic and not HBM is the bound.
            float rt[NE];
#pragma unroll
            for (int e = 0; e < NE; ++e) rt[e] = sqrtf(s1[e]);
            const float rb2 = sqrtf(u.b2);
            int j = 0;
            for (; j < k; ++j) {
                const float a_t = sp_at(u, t0 + 1 + j);
                bool live = false;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const float m = u.b1 * s0[e];
                    rt[e] *= rb2;
                    const float upd = a_t * m * __builtin_amdgcn_rcpf(rt[e] + 1e-8f);
                    s0[e] = m; p[e] -= upd;
                    live = live || !(m == 0.0f || (u.early_exit && fabsf(upd) < fabsf(p[e]) * 1.4901161e-8f));   // 2^-26 |p|
                }
                if (!live || j >= 8190) { ++j; break; }
            }
            const float f2 = powf(u.b2, (float)k);               // v after all k steps, from the exact start value
#pragma unroll
            for (int e = 0; e < NE; ++e) s1[e] *= f2;
            if (j < k) {
                const float f1 = powf(u.b1, (float)(k - j));
#pragma unroll
                for (int e = 0; e < NE; ++e) s0[e] *= f1;
            }
            return;
        }
    }
}

// one real step (gradient g) -- the dense update_kernel's arithmetic
__device__ __forceinline__ void sp_step(const SpUpd& u, float& p, float g, float& s0, float& s1, float a_t) {
    switch (u.updater) {
        case SBR_UPD_ADAGRAD: { const float acc = s0 + g * g; s0 = acc; p -= u.lr * g / sqrtf(acc + 1e-6f); return; }
        case SBR_UPD_RMSPROP: { const float acc = u.rho * s0 + (1.0f - u.rho) * g * g; s0 = acc; p -= u.lr * g / sqrtf(acc + 1e-6f); return; }
        case SBR_UPD_ADADELTA: {
            const float acc = u.rho * s0 + (1.0f - u.rho) * g * g;
            const float upd = g * sqrtf(s1 + 1e-6f) / sqrtf(acc + 1e-6f);
            s0 = acc; p -= u.lr * upd; s1 = u.rho * s1 + (1.0f - u.rho) * upd * upd; return;
        }
        case SBR_UPD_NESTEROV: { const float v = u.rho * s0 - u.lr * g; s0 = v; p += u.rho * v - u.lr * g; return; }
        default: {
            const float m = u.b1 * s0 + (1.0f - u.b1) * g;
            const float v = u.b2 * s1 + (1.0f - u.b2) * g * g;
            s0 = m; s1 = v; p -= a_t * m / (sqrtf(v) + 1e-8f); return;
        }
    }
}

// SRC 0: the current batch (X [Bp][T][F], len [Bp]: entry i = (b, t, f), valid when t < len[b]);
//     1: an id list of *n_list (device) or n_host entries;   2: every row 0 .. n_rows-1 (flush)
// STEP false: catch the row up to step t_to;  true: catch it up to t_to - 1, then apply step t_to with its gradient
// (which is cleared, as the dense kernel clears what it consumes).
// A wave takes SP_CPW candidates at a time and then walks the rows it owns one after the other, all 64 lanes on one row:
// a row pass is a dependent load -> math -> store chain of ~2 us, so the candidates per wave bound the length of the
// serial chain (64 per wave measured 600 us for C3's 51 200 candidates, the kernel being nothing but 64-deep chains);
// two 256-float pieces of a row are loaded before the first is used and replayed together (four pieces: 256 VGPRs, one
// wave per SIMD).
// (8 until round 4; 2 since round 5: the rows a wave owns are walked one after the other, each a load -> math -> store round trip, and
// the waves of the sorted list's long tail owned all 8 of theirs -- C3 1.397 -> 1.378 (4) -> 1.373 ms (2): profiles/round5_variants.txt call e)
#ifndef SP_CPW
#define SP_CPW 2
#endif
#define SP_NV 2
template <int SRC, bool STEP>
__global__ void __launch_bounds__(256) sp_rows_kernel(SbrSparseRows r, SpUpd u, const int* __restrict__ X, const int* __restrict__ len,
                                                      int T, int Bp, int F, const int* __restrict__ list, const int* __restrict__ n_list,
                                                      int n_host, int t_to) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int total = SRC == 0 ? Bp * T * F : SRC == 1 ? (n_list ? *n_list : n_host) : r.n_rows;
    const float a_t = STEP ? sp_at(u, t_to) : 0.0f;
    for (int base = wave * SP_CPW; base < total; base += nwaves * SP_CPW) {
        const int i = base + lane;
        int id = -1;
        if (lane < SP_CPW && i < total) {
            if (SRC == 0) { const int b = i / (T * F), t = (i / F) % T; if (t < len[b]) id = X[i]; }
            else if (SRC == 1) {
                id = list[i];
                // (an entry that repeats its predecessor is that predecessor's business: the step's list is the SORTED entries of the
                // batch, a hot id fills thousands of consecutive places of it, and every one of them used to send its own atomicMax to
                // the same last[id] -- round 6, call b2)
                if (i > 0 && list[i - 1] == id) id = -1;
            }
            else id = i;
        }
        int old = -1;
        bool own = false;
        if (id >= 0 && id < r.n_rows) {
            old = r.last[id];
            if (old < t_to) { old = SRC == 2 ? old : atomicMax(&r.last[id], t_to); own = old < t_to; if (SRC == 2) r.last[id] = t_to; }
        }
        unsigned long long owners = __ballot(own);
        while (owners) {
            const int src = __ffsll((long long)owners) - 1;
            owners &= owners - 1;
            const int rid = __shfl(id, src), rold = __shfl(old, src);
            const int k = (STEP ? t_to - 1 : t_to) - rold;            // zero-gradient steps rold+1 .. rold+k
            if (!STEP && (k <= 0 || u.updater == SBR_UPD_ADAGRAD)) continue;
            for (int pr = 0; pr < r.npairs; ++pr) {
                const size_t ro = r.off[pr] + (size_t)rid * r.stride[pr];
                const int w = r.width[pr];
                if (w >= 4) {
                    for (int c0 = 0; c0 < w; c0 += 256 * SP_NV) {
                        f32x4 p[SP_NV], s0[SP_NV], s1[SP_NV], g[SP_NV];
#pragma unroll
                        for (int v = 0; v < SP_NV; ++v) {
                            const int c = c0 + v * 256 + lane * 4;
                            const bool in = c < w;
                            p[v] = in ? *(const f32x4*)(r.p + ro + c) : f32x4{0, 0, 0, 0};
                            s0[v] = in ? *(const f32x4*)(r.s0 + ro + c) : f32x4{0, 0, 0, 0};
                            s1[v] = (in && r.s1) ? *(const f32x4*)(r.s1 + ro + c) : f32x4{0, 0, 0, 0};
                            g[v] = (in && STEP) ? *(const f32x4*)(r.g + ro + c) : f32x4{0, 0, 0, 0};
                        }
                        {
                            float pe[4 * SP_NV], ae[4 * SP_NV], be[4 * SP_NV];
#pragma unroll
                            for (int v = 0; v < SP_NV; ++v)
#pragma unroll
                                for (int e = 0; e < 4; ++e) { pe[4 * v + e] = p[v][e]; ae[4 * v + e] = s0[v][e]; be[4 * v + e] = s1[v][e]; }
                            sp_catch_up<4 * SP_NV>(u, pe, ae, be, k, (long)rold);       // (pieces beyond the row hold zeros: m == 0)
#pragma unroll
                            for (int v = 0; v < SP_NV; ++v)
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    if (STEP) sp_step(u, pe[4 * v + e], g[v][e], ae[4 * v + e], be[4 * v + e], a_t);
                                    p[v][e] = pe[4 * v + e]; s0[v][e] = ae[4 * v + e]; s1[v][e] = be[4 * v + e];
                                }
                        }
#pragma unroll
                        for (int v = 0; v < SP_NV; ++v) {
                            const int c = c0 + v * 256 + lane * 4;
                            if (c >= w) continue;
                            *(f32x4*)(r.p + ro + c) = p[v];
                            *(f32x4*)(r.s0 + ro + c) = s0[v];
                            if (r.s1) *(f32x4*)(r.s1 + ro + c) = s1[v];
                            if (STEP) *(f32x4*)(r.g + ro + c) = f32x4{0, 0, 0, 0};
                        }
                    }
                } else if (lane < w) {                                  // bias rows: one float
                    const size_t o = ro + lane;
                    float pe[1] = {r.p[o]}, a[1] = {r.s0[o]}, b[1] = {r.s1 ? r.s1[o] : 0.0f};
                    sp_catch_up<1>(u, pe, a, b, k, (long)rold);
                    if (STEP) { const float gg = r.g[o]; r.g[o] = 0.0f; sp_step(u, pe[0], gg, a[0], b[0], a_t); }
                    r.p[o] = pe[0]; r.s0[o] = a[0]; if (r.s1) r.s1[o] = b[0];
                }
            }
        }
    }
}

static SpUpd make_upd(const SbrSparseUpd& c) {
    SpUpd u; u.updater = c.updater; u.lr = c.lr; u.rho = c.rho; u.b1 = c.b1; u.b2 = c.b2; u.at = c.at; u.n_at = c.n_at;
    u.early_exit = c.early_exit;
    return u;
}
// one wave per SP_CPW candidates (4 waves per workgroup), at most 16 k workgroups (grid-stride beyond)
static inline int sp_grid(long total) { return (int)std::max<long>(1, std::min<long>(16384, (total + 4 * SP_CPW - 1) / (4 * SP_CPW))); }

hipError_t launch_sparse_catch_up_batch(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, const int* X, const int* len, int T,
                                        int Bp, int F, int t_to) {
    sp_rows_kernel<0, false><<<sp_grid((long)Bp * T * F), 256, 0, s>>>(r, make_upd(c), X, len, T, Bp, F, nullptr, nullptr, 0, t_to);
    return hipGetLastError();
}
hipError_t launch_sparse_catch_up_list(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, const int* list, const int* n_dev,
                                       int n_host, int n_max, int t_to) {
    sp_rows_kernel<1, false><<<sp_grid(n_max), 256, 0, s>>>(r, make_upd(c), nullptr, nullptr, 0, 0, 0, list, n_dev, n_host, t_to);
    return hipGetLastError();
}
hipError_t launch_sparse_flush(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, int t_to) {
    sp_rows_kernel<2, false><<<sp_grid(r.n_rows), 256, 0, s>>>(r, make_upd(c), nullptr, nullptr, 0, 0, 0, nullptr, nullptr, 0, t_to);
    return hipGetLastError();
}
hipError_t launch_sparse_step_list(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, const int* list, const int* n_dev,
                                   int n_host, int n_max, int t_to) {
    sp_rows_kernel<1, true><<<sp_grid(n_max), 256, 0, s>>>(r, make_upd(c), nullptr, nullptr, 0, 0, 0, list, n_dev, n_host, t_to);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Sparse gradient exchange (data parallel): instead of all-reducing the whole block, every rank packs the gradient
// rows it touched -- (row id, the row of every pair, concatenated) -- the ranks all-gather those, and every rank adds
// all ranks' rows, in rank order, into its (now empty) gradient block: the replicas stay bit-identical.
// ---------------------------------------------------------------------------------------
// list: candidate ids with duplicates; mark[id] = epoch claims a row; count: running number of packed rows
__global__ void __launch_bounds__(256) sp_pack_kernel(SbrSparseRows r, const int* __restrict__ list, const int* __restrict__ n_list, int n_host,
                                                      int* __restrict__ mark, int epoch, int* __restrict__ ids_out,
                                                      float* __restrict__ rows_out, int W, int* __restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int total = n_list ? *n_list : n_host;
    for (int base = wave * SP_CPW; base < total; base += nwaves * SP_CPW) {
        const int i = base + lane;
        const int id = (lane < SP_CPW && i < total) ? list[i] : -1;
        bool own = false;
        if (id >= 0 && id < r.n_rows && mark[id] != epoch) own = atomicMax(&mark[id], epoch) != epoch;
        int slot = 0;
        if (own) { slot = atomicAdd(count, 1); ids_out[slot] = id; }
        unsigned long long owners = __ballot(own);
        while (owners) {
            const int src = __ffsll((long long)owners) - 1;
            owners &= owners - 1;
            const int rid = __shfl(id, src), rslot = __shfl(slot, src);
            float* dst = rows_out + (size_t)rslot * W;
            int col = 0;
            for (int pr = 0; pr < r.npairs; ++pr) {
                float* g = r.g + r.off[pr] + (size_t)rid * r.stride[pr];
                for (int c = lane; c < r.width[pr]; c += 64) { dst[col + c] = g[c]; g[c] = 0.0f; }
                col += r.width[pr];
            }
        }
    }
}

// rows of ONE rank (ids unique): g[id] += row; the ids are appended to the step's candidate list
__global__ void __launch_bounds__(256) sp_unpack_kernel(SbrSparseRows r, const int* __restrict__ ids, const float* __restrict__ rows, int n,
                                                        int W, int* __restrict__ cand) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    for (int j = wave; j < n; j += nwaves) {
        const int id = ids[j];
        if (lane == 0) cand[j] = id;
        if (id < 0 || id >= r.n_rows) continue;
        const float* src = rows + (size_t)j * W;
        int col = 0;
        for (int pr = 0; pr < r.npairs; ++pr) {
            float* g = r.g + r.off[pr] + (size_t)id * r.stride[pr];
            for (int c = lane; c < r.width[pr]; c += 64) g[c] += src[col + c];
            col += r.width[pr];
        }
    }
}

// ... with the rank's row count on the device (ids[0]; the ids follow): slots j >= count of the candidate list get -1
__global__ void __launch_bounds__(256) sp_unpack_dev_kernel(SbrSparseRows r, const int* __restrict__ ids, const float* __restrict__ rows, int cap,
                                                            int W, int* __restrict__ cand) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * (blockDim.x >> 6);
    const int n = min(max(ids[0], 0), cap);
    for (int j = wave; j < cap; j += nwaves) {
        const int id = j < n ? ids[1 + j] : -1;
        if (lane == 0) cand[j] = id;
        if (id < 0 || id >= r.n_rows) continue;
        const float* src = rows + (size_t)j * W;
        int col = 0;
        for (int pr = 0; pr < r.npairs; ++pr) {
            float* g = r.g + r.off[pr] + (size_t)id * r.stride[pr];
            for (int c = lane; c < r.width[pr]; c += 64) g[c] += src[col + c];
            col += r.width[pr];
        }
    }
}

hipError_t launch_sparse_unpack_add_dev(hipStream_t s, const SbrSparseRows& r, const int* ids, const float* rows, int cap, int W, int* cand) {
    if (cap <= 0) return hipSuccess;
    sp_unpack_dev_kernel<<<std::max(1, std::min(16384, (cap + 3) / 4)), 256, 0, s>>>(r, ids, rows, cap, W, cand);
    return hipGetLastError();
}

hipError_t launch_sparse_pack(hipStream_t s, const SbrSparseRows& r, const int* list, const int* n_dev, int n_host, int n_max, int* mark,
                              int epoch, int* ids_out, float* rows_out, int W, int* count) {
    hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s);
    if (e != hipSuccess) return e;
    sp_pack_kernel<<<sp_grid(n_max), 256, 0, s>>>(r, list, n_dev, n_host, mark, epoch, ids_out, rows_out, W, count);
    return hipGetLastError();
}
hipError_t launch_sparse_unpack_add(hipStream_t s, const SbrSparseRows& r, const int* ids, const float* rows, int n, int W, int* cand) {
    if (n <= 0) return hipSuccess;
    sp_unpack_kernel<<<std::max(1, std::min(16384, (n + 3) / 4)), 256, 0, s>>>(r, ids, rows, n, W, cand);
    return hipGetLastError();
}
