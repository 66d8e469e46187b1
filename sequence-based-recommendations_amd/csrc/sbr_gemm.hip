// Generic float32 GEMM on v_mfma_f32_16x16x4_f32 (exact f32: bitwise an fmaf chain) for gfx950.
// Used for the dense pieces of the hot path: output projection logits = h.W_out (K7), its
// backward pair (K9), the split-K weight gradients dW_hid = hs^T.dhi / dW_in = x^T.dxt that run
// after the BPTT chain, and the layer>=2 input projections (K15).
//
//   C[m][n] = sum_k A(m,k) * B(k,n) (+ bias[n]),  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
//
// Workgroup = 256 threads = 4 waves, 64x64 output tile, BK = 16; each wave owns a 32x32
// sub-tile = 2x2 MFMA tiles.  Global -> registers -> LDS with the next K-tile prefetched into
// registers while the MFMAs of the current one run.  Split-K over grid.z writes partial slabs
// that a second kernel sums in fixed order (deterministic).
#include "sbr_common.h"
#include <algorithm>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define BM 64
#define BN 64
#define BK 16
#define LDT 68   // LDS row stride (floats): 64 + 4 keeps 16-B alignment and breaks the 64-float bank period

struct GemmArgs {
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    float* C; long ldc;
    int M, N, K;
    const float* bias;
    int a_blk_Bp, b_blk_Bp;   // > 0: operand is a tile-blocked activation [pos = t*Bp + row][col]
    int kchunk;      // K range per grid.z slice (multiple of BK)
    float* ws;       // split-K slabs (NULL when gridDim.z == 1): slab z at ws + z*ws_stride, row stride ws_ld
    long ws_ld; size_t ws_stride;
};

__global__ void __launch_bounds__(256) gemm_f32_mfma(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float As[BK * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int j = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);

    // thread -> element mapping for the global loads: 4 elements per thread per operand,
    // contiguous along whichever dimension has unit stride
    const bool a_kfast = g.sak == 1 || g.a_blk_Bp > 0;
    const int a_m = a_kfast ? (tid >> 2) : ((tid & 15) << 2);
    const int a_k = a_kfast ? ((tid & 3) << 2) : (tid >> 4);
    const bool b_nfast = g.sbn == 1 || g.b_blk_Bp > 0;
    const int b_n = b_nfast ? ((tid & 15) << 2) : (tid >> 2);
    const int b_k = b_nfast ? (tid >> 4) : ((tid & 3) << 2);

    float ra[4], rb[4];
    auto a_addr = [&](int m, int k) -> size_t {   // blocked A: rows = positions, K columns
        if (g.a_blk_Bp > 0) { const int t = m / g.a_blk_Bp; return sbr_blocked_index(t, m - t * g.a_blk_Bp, k, g.a_blk_Bp, g.K); }
        return (size_t)((long)m * g.sam + (long)k * g.sak);
    };
    auto b_addr = [&](int k, int n) -> size_t {   // blocked B: k = positions, N columns
        if (g.b_blk_Bp > 0) { const int t = k / g.b_blk_Bp; return sbr_blocked_index(t, k - t * g.b_blk_Bp, n, g.b_blk_Bp, g.N); }
        return (size_t)((long)k * g.sbk + (long)n * g.sbn);
    };
    auto load_tile = [&](int k0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + a_m + (a_kfast ? 0 : e), k = k0 + a_k + (a_kfast ? e : 0);
            ra[e] = (m < g.M && k < kend) ? g.A[a_addr(m, k)] : 0.0f;
            const int n = n0 + b_n + (b_nfast ? e : 0), kb = k0 + b_k + (b_nfast ? 0 : e);
            rb[e] = (n < g.N && kb < kend) ? g.B[b_addr(kb, n)] : 0.0f;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            As[(a_k + (a_kfast ? e : 0)) * LDT + a_m + (a_kfast ? 0 : e)] = ra[e];
            Bs[(b_k + (b_nfast ? 0 : e)) * LDT + b_n + (b_nfast ? e : 0)] = rb[e];
        }
    };

    f32x4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0, 0, 0, 0};

    if (kbeg < kend) {
        load_tile(kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += BK) {
            __syncthreads();               // previous tile's MFMA reads are done
            store_tile();
            __syncthreads();
            if (k0 + BK < kend) load_tile(k0 + BK);
#pragma unroll
            for (int ks = 0; ks < BK / 4; ++ks) {
                float af[2], bf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) af[a] = As[(ks * 4 + q) * LDT + wm * 32 + a * 16 + j];
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[b] = Bs[(ks * 4 + q) * LDT + wn * 32 + b * 16 + j];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        }
    }

    asm volatile("s_nop 15");   // MFMA D -> VALU read hazard across the loop exit (see sbr_rec.hip)
    float* out = g.ws ? g.ws + (size_t)blockIdx.z * g.ws_stride : g.C;
    const long ld = g.ws ? g.ws_ld : g.ldc;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn * 32 + b * 16 + j;
            if (n >= g.N) continue;
            const float bv = (!g.ws && g.bias) ? g.bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 32 + a * 16 + q * 4 + r;
                if (m < g.M) out[(long)m * ld + n] = acc[a][b][r] + bv;
            }
        }
}

// C = sum of the split-K slabs (+ bias): 64 outputs x 4 slab groups per workgroup, 4 loads in flight per
// thread, fixed summation order (deterministic).  Slabs are read once, coalesced.
__global__ void __launch_bounds__(256) gemm_splitk_reduce(const float* __restrict__ ws, int nsplit, int M, int N,
                                                          float* __restrict__ C, long ldc, const float* __restrict__ bias) {
    __shared__ float red[4][64];
    const size_t MN = (size_t)M * N;
    const size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < MN) {
        int z = grp;
        for (; z + 12 < nsplit; z += 16) {
            s0 += ws[(size_t)z * MN + i]; s1 += ws[(size_t)(z + 4) * MN + i];
            s2 += ws[(size_t)(z + 8) * MN + i]; s3 += ws[(size_t)(z + 12) * MN + i];
        }
        for (; z < nsplit; z += 4) s0 += ws[(size_t)z * MN + i];
    }
    red[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0 || i >= MN) return;
    const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    const int m = (int)(i / N), n = (int)(i % N);
    C[(long)m * ldc + n] = s + (bias ? bias[n] : 0.0f);
}

// The same for many slabs (the weight gradients: 170 slabs of 128 x 384): 16-byte loads, 64 outputs x 16 slab groups per
// workgroup, 4 loads in flight per thread, fixed summation order.  Needs M*N % 4 == 0 and 16-byte aligned slabs.
__global__ void __launch_bounds__(256) gemm_splitk_reduce_v4(const f32x4* __restrict__ ws, int nsplit, int M, int N,
                                                             float* __restrict__ C, long ldc, const float* __restrict__ bias) {
    __shared__ f32x4 red[16][16];
    const size_t MN4 = (size_t)M * N / 4;
    const int j = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const size_t i4 = (size_t)blockIdx.x * 16 + j;
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    if (i4 < MN4) {
        int z = grp;
        for (; z + 48 < nsplit; z += 64) {
            s0 += ws[(size_t)z * MN4 + i4]; s1 += ws[(size_t)(z + 16) * MN4 + i4];
            s2 += ws[(size_t)(z + 32) * MN4 + i4]; s3 += ws[(size_t)(z + 48) * MN4 + i4];
        }
        for (; z < nsplit; z += 16) s0 += ws[(size_t)z * MN4 + i4];
    }
    red[grp][j] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int jj = threadIdx.x >> 2, e = threadIdx.x & 3;             // one output element per thread
    const size_t i = ((size_t)blockIdx.x * 16 + jj) * 4 + e;
    if (i >= (size_t)M * N) return;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][jj][e];
    const int m = (int)(i / N), n = (int)(i % N);
    C[(long)m * ldc + n] = t + (bias ? bias[n] : 0.0f);
}

// Few slabs of a large output (the dense layer GEMMs of C5: 4 - 16 slabs of 512 x 2048): one 16-byte piece of the output per
// thread, the slabs four at a time, fixed order.  The scalar kernel above read 4 bytes per lane and slab: 206 us for 64 MB.
__global__ void __launch_bounds__(256) gemm_splitk_reduce_v4s(const f32x4* __restrict__ ws, int nsplit, int M, int N,
                                                              float* __restrict__ C, long ldc, const float* __restrict__ bias) {
    const size_t MN4 = (size_t)M * N / 4, i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i4 >= MN4) return;
    f32x4 s0 = {0, 0, 0, 0}, s1 = s0, s2 = s0, s3 = s0;
    int z = 0;
    for (; z + 4 <= nsplit; z += 4) {
        s0 += ws[(size_t)z * MN4 + i4]; s1 += ws[(size_t)(z + 1) * MN4 + i4];
        s2 += ws[(size_t)(z + 2) * MN4 + i4]; s3 += ws[(size_t)(z + 3) * MN4 + i4];
    }
    for (; z < nsplit; ++z) s0 += ws[(size_t)z * MN4 + i4];
    f32x4 t = (s0 + s1) + (s2 + s3);
    const size_t i = i4 * 4;
    const int m = (int)(i / N), n = (int)(i % N);       // N % 4 == 0: the four elements share a row
    if (bias) { t[0] += bias[n]; t[1] += bias[n + 1]; t[2] += bias[n + 2]; t[3] += bias[n + 3]; }
    float* dst = C + (long)m * ldc + n;
    if ((ldc & 3) == 0 && ((uintptr_t)C & 15) == 0) *(f32x4*)dst = t;
    else { dst[0] = t[0]; dst[1] = t[1]; dst[2] = t[2]; dst[3] = t[3]; }
}

static hipError_t splitk_reduce(hipStream_t s, const float* ws, int nslabs, int M, int N, float* C, long ldc, const float* bias) {
    const size_t n = (size_t)M * N;
    if (nslabs < 32 && n >= (1u << 16) && (N & 3) == 0 && ((uintptr_t)ws & 15) == 0) {
        gemm_splitk_reduce_v4s<<<(unsigned)((n / 4 + 255) / 256), 256, 0, s>>>((const f32x4*)ws, nslabs, M, N, C, ldc, bias);
        return hipGetLastError();
    }
    if (nslabs >= 32 && (n & 3) == 0 && ((uintptr_t)ws & 15) == 0)
        gemm_splitk_reduce_v4<<<(unsigned)((n / 4 + 15) / 16), 256, 0, s>>>((const f32x4*)ws, nslabs, M, N, C, ldc, bias);
    else
        gemm_splitk_reduce<<<(unsigned)((n + 63) / 64), 256, 0, s>>>(ws, nslabs, M, N, C, ldc, bias);
    return hipGetLastError();
}

// triage: one thread per output element
__global__ void gemm_naive(GemmArgs g) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)g.M * g.N) return;
    const int m = i / g.N, n = i % g.N;
    float s = 0.0f;
    for (int k = 0; k < g.K; ++k) {
        size_t ia = (size_t)((long)m * g.sam + (long)k * g.sak), ib = (size_t)((long)k * g.sbk + (long)n * g.sbn);
        if (g.a_blk_Bp > 0) { const int t = m / g.a_blk_Bp; ia = sbr_blocked_index(t, m - t * g.a_blk_Bp, k, g.a_blk_Bp, g.K); }
        if (g.b_blk_Bp > 0) { const int t = k / g.b_blk_Bp; ib = sbr_blocked_index(t, k - t * g.b_blk_Bp, n, g.b_blk_Bp, g.N); }
        s = fmaf(g.A[ia], g.B[ib], s);
    }
    g.C[(long)m * g.ldc + n] = s + (g.bias ? g.bias[n] : 0.0f);
}

// SBR_FLAG_F32_MFMA: keep every GEMM on the exact-f32 kernel (set per call by the API layer; a handle is not thread-safe)
// (per host thread: two handles driven from different threads must not steal or leak each other's mode / hint -- ADVICE round 4)
static thread_local bool g_gemm_exact_f32 = false;
void sbr_gemm_set_exact_f32(bool on) { g_gemm_exact_f32 = on; }
static thread_local int g_gemm_planes = 3;
void sbr_gemm_set_planes(int planes) { g_gemm_planes = planes == 1 ? 1 : 3; }
// Hint for the NEXT launch_gemm call only (consumed and cleared by it; a handle is not thread-safe): the operand planes of the
// tiled kernel -- 2: the two-plane fp16 split (three MFMAs per product instead of bf16x6's six) for operands the caller knows to be
// bounded: hidden states behind tanh / sigmoid gates, weights, gradients that have passed the clip at +-100, with sa / sb the
// power-of-two scales that bring A / B into fp16's range (gemm_x6_kernel NP = 2); 1: plain bf16 operands, split-K allowed (the
// layer GEMMs under SBR_FLAG_BF16_LAYERS).  0: none (bf16x6).
static thread_local int g_hint_planes = 0; static thread_local float g_hint_sa = 1.0f, g_hint_sb = 1.0f;
void sbr_gemm_hint(int planes, float sa, float sb) { g_hint_planes = planes; g_hint_sa = sa; g_hint_sb = sb; }

// split-K partial products only: writes exactly `nsplit` slabs [z][M][N] at ws (no reduction)
hipError_t launch_gemm_slabs(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M,
                             int N, int K, float* ws, int nsplit, long ws_ld, size_t slab_stride) {
    if (M <= 0 || N <= 0 || nsplit < 1) return hipSuccess;
    if (!g_gemm_exact_f32) {
        const int kc = ((K + nsplit - 1) / nsplit + 31) / 32 * 32;   // slices past K write zero slabs
        hipError_t e = hipSuccess;
        // nsplit == 1 would take the "write C" form: identical here (slab 0, no bias)
        if (launch_gemm_x6(s, A, sam, sak, B, sbk, sbn, ws, ws_ld, M, N, K, nullptr, nsplit, kc, slab_stride, &e)) return e;
    }
    GemmArgs g{A, sam, sak, B, sbk, sbn, ws, N, M, N, K, nullptr, 0, 0, K, ws, ws_ld, slab_stride};
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    g.kchunk = ((K + nsplit - 1) / nsplit + BK - 1) / BK * BK;   // slices past K write zero slabs
    gemm_f32_mfma<<<dim3(tn, tm, nsplit), 256, 0, s>>>(g);
    return hipGetLastError();
}
bool launch_gemm_slabs_x6(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M, int N,
                          int K, float* ws, int nsplit, long ws_ld, size_t slab_stride, const float* B2, long sbk2, int n_split,
                          hipError_t* err, int planes, float sa, float sb) {
    if (g_gemm_exact_f32 || M <= 0 || N <= 0 || nsplit < 1) return false;
    const int kc = ((K + nsplit - 1) / nsplit + 31) / 32 * 32;
    return launch_gemm_x6(s, A, sam, sak, B, sbk, sbn, ws, ws_ld, M, N, K, nullptr, nsplit, kc, slab_stride, err, B2, sbk2, n_split, false,
                          planes, sa, sb);
}
bool launch_gemm_slabs_x6_poll(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M, int N,
                               int K, float* ws, int n_groups, long ws_ld, size_t slab_stride, const float* B2, long sbk2,
                               int n_split, hipError_t* err, int planes, float sa, float sb, const SbrPoll& poll) {
    if (g_gemm_exact_f32 || M <= 0 || N <= 0 || n_groups < 1 || !poll.slab_lo) return false;
    return launch_gemm_x6(s, A, sam, sak, B, sbk, sbn, ws, ws_ld, M, N, K, nullptr, n_groups, 32, slab_stride, err, B2, sbk2,
                          n_split, false, planes, sa, sb, &poll);
}
int sbr_tail_slab_table(int K, int rows_per_step, int cap, double growth, int max_rows, std::vector<int>& lo) {
    const int max_n = std::max(1, max_rows / 32);
    for (double scale = 1.0; ; scale *= 1.5) {
        lo.assign(1, 0);
        for (int r = 0; r < K; ) {
            const double t = (double)r / std::max(1, rows_per_step);
            int n = (int)std::floor((growth * t - 1.0) * scale);
            n = std::max((int)scale, std::min(max_n, n));
            r += std::min(32 * n, K - r);
            lo.push_back(r);
        }
        if ((int)lo.size() - 1 <= cap || scale > 1e6) break;
    }
    return (int)lo.size() - 1;
}
hipError_t launch_splitk_reduce(hipStream_t s, const float* ws, int nslabs, int M, int N, float* C, long ldc,
                                const float* bias) {
    return splitk_reduce(s, ws, nslabs, M, N, C, ldc, bias);
}

hipError_t launch_gemm(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C,
                       long ldc, int M, int N, int K, const float* bias, float* ws, size_t ws_floats, bool simple,
                       int a_blk_Bp, int b_blk_Bp, int* keep_slabs) {
    if (keep_slabs) *keep_slabs = 0;
    const int hint_planes = g_hint_planes; const float hint_sa = g_hint_sa, hint_sb = g_hint_sb;
    g_hint_planes = 0; g_hint_sa = g_hint_sb = 1.0f;
    if (M <= 0 || N <= 0) return hipSuccess;
    GemmArgs g{A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias, a_blk_Bp, b_blk_Bp, K, nullptr, (long)N, (size_t)M * N};
    if (simple) {
        const size_t n = (size_t)M * N;
        gemm_naive<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(g);
        return hipGetLastError();
    }
    if (g_gemm_planes == 1 && a_blk_Bp == 0 && b_blk_Bp == 0) {
        // plain bf16 operands: one pass over K in a fixed order (no split-K), so that an output element never depends on M
        hipError_t e = hipSuccess;
        const int t128 = ((M + 127) / 128) * ((N + 127) / 128);
        if (launch_gemm_x6(s, A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, bias, 1, (K + 31) / 32 * 32, (size_t)M * N, &e, nullptr, 0, 0,
                           t128 < 128 || M < 96, 1))
            return e;
    }
    if (!g_gemm_exact_f32 && a_blk_Bp == 0 && b_blk_Bp == 0 && M >= 48 && N >= 48 && K >= 32) {
        const int small_below = 128;      // (fewer large tiles than this on the chip: the 64 x 64 x 64 tile -- profiles/round1_j_*)
        auto plan = [&](int tile, int& ns, int& kc) {
            const int t = ((M + tile - 1) / tile) * ((N + tile - 1) / tile);
            ns = 1;
            if (ws && K >= 512) {
                ns = std::max(1, 512 / t);
                ns = std::min(ns, K / 128);
                ns = (int)std::min<size_t>((size_t)ns, ws_floats / ((size_t)M * N));
                ns = std::max(ns, 1);
            }
            kc = ((K + ns - 1) / ns + 31) / 32 * 32;
            ns = std::max(1, (K + kc - 1) / kc);
            return t * ns;
        };
        int ns, kc;
        // fewer than ~128 workgroups of the 128x128 tile: 64x64 tiles put four times as many on the chip
        const bool small = plan(128, ns, kc) < small_below || M < 96 || N < 96;
        if (small) plan(64, ns, kc);
        hipError_t e = hipSuccess;
        if (launch_gemm_x6(s, A, sam, sak, B, sbk, sbn, ns > 1 ? ws : C, ns > 1 ? (long)N : ldc, M, N, K, bias, ns, kc,
                           (size_t)M * N, &e, nullptr, 0, 0, small, hint_planes ? hint_planes : 3, hint_sa, hint_sb)) {
            if (e == hipSuccess && ns > 1) {
                if (keep_slabs && !bias) *keep_slabs = ns;                // the consumer adds the slabs
                else e = splitk_reduce(s, ws, ns, M, N, C, ldc, bias);
            }
            return e;
        }
    }
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    int nsplit = 1;
    if (ws && K >= 8 * BK) {
        nsplit = 1024 / (tm * tn);
        nsplit = min(nsplit, K / (4 * BK));
        nsplit = (int)min((size_t)nsplit, ws_floats / ((size_t)M * N));
        nsplit = max(nsplit, 1);
    }
    int kchunk = ((K + nsplit - 1) / nsplit + BK - 1) / BK * BK;
    if (kchunk <= 0) kchunk = BK;
    nsplit = max(1, (K + kchunk - 1) / kchunk);
    g.kchunk = kchunk;
    g.ws = nsplit > 1 ? ws : nullptr;
    gemm_f32_mfma<<<dim3(tn, tm, nsplit), 256, 0, s>>>(g);
    if (nsplit > 1) {
        const size_t n = (size_t)M * N;
        gemm_splitk_reduce<<<(unsigned)((n + 63) / 64), 256, 0, s>>>(ws, nsplit, M, N, C, ldc, bias);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Dedicated weight-gradient kernel: slab[z][Hp][GHp] = sum over the positions of K-slice z of
//     hs[pos][:]^T (x) dz[pos][:]      (dW_hid = hs_prev^T . d hid_input, K = T*B positions)
// Both operands are position-major rows, which is exactly the f32 MFMA fragment shape
// (A[i][k] = hs[pos+k][unit i], B[k][j] = dz[pos+k][col j]): fragments are loaded straight from
// global/L2 as 64-byte segments, no LDS, no transposes.  One workgroup = 8 waves as 2 (units) x 4
// (columns); every workgroup produces a full Hp x GHp slab for its slice of positions, so each
// activation row is read exactly once chip-wide.  GRU: columns >= 2*Hp come from the compact
// candidate-gate array (sbr_rec.hip), the others from dxt.
// ---------------------------------------------------------------------------------------
template <int MT, int NT>
__global__ void __launch_bounds__(512) wgrad_kernel(const float* __restrict__ hs, const float* __restrict__ dxt,
                                                    const float* __restrict__ dhc, float* __restrict__ slabs, int Hp, int GHp,
                                                    int split_col, int npos, int pps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;
    const int m0 = wm * MT * 16, n0 = wn * NT * 16;
    const int pbeg = blockIdx.x * pps, pend = min(npos, pbeg + pps);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4{0, 0, 0, 0};
    // per-column-tile source (uniform per tile): dxt row stride GHp, or the compact array with stride Hp
    const float* bsrc[NT]; int bld[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n0 + b * 16;
        if (dhc && n >= split_col) { bsrc[b] = dhc + (n - split_col) + j; bld[b] = Hp; }
        else { bsrc[b] = dxt + n + j; bld[b] = GHp; }
    }
    const float* asrc = hs + m0 + j;
    // fragments of later k-steps (4 positions each) are fetched before the MFMAs of k-step p: with one workgroup per CU nothing
    // else hides the L2/HBM latency of these loads.  How many k-steps ahead depends on what a step holds: the small layers (C1:
    // one float of A and two of B per lane and step) were a chain of exposed round trips with one step in flight -- 47 us for
    // 33 MB at LSTM-20 (profiles/round6_variants.txt, call y); S stages in flight, S - 1 steps ahead.
    constexpr int S = MT + NT <= 3 ? 8 : MT + NT <= 6 ? 4 : 2;
    float af[S][MT], bf[S][NT];
    auto fetch = [&](int p, int s) {
        const int pos = p + q;
        const bool ok = pos < pend;
#pragma unroll
        for (int a = 0; a < MT; ++a) af[s][a] = ok ? asrc[(size_t)pos * Hp + a * 16] : 0.0f;
#pragma unroll
        for (int b = 0; b < NT; ++b) bf[s][b] = ok ? bsrc[b][(size_t)pos * bld[b]] : 0.0f;
    };
    auto mma = [&](int s) {
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s][a], bf[s][b], acc[a][b], 0, 0, 0);
    };
#pragma unroll
    for (int s = 0; s < S - 1; ++s) fetch(pbeg + 4 * s, s);
    for (int p = pbeg; p < pend; p += 4 * S) {
#pragma unroll
        for (int s = 0; s < S; ++s) {                      // (steps past the slice fetch zeros: their MFMAs add nothing)
            fetch(p + 4 * (s + S - 1), (s + S - 1) % S);
            __builtin_amdgcn_sched_barrier(0);
            mma(s);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 15");
    float* out = slabs + (size_t)blockIdx.x * Hp * GHp;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(m0 + a * 16 + q * 4 + r) * GHp + n0 + b * 16 + j] = acc[a][b][r];
}

// returns false when the shape has no instantiation (caller falls back to the generic GEMM)
bool launch_wgrad_slabs(hipStream_t s, const float* hs, const float* dxt, const float* dhc, float* slabs, int Hp, int GHp,
                        int npos, int nslices, hipError_t* err) {
    const int mt = Hp / 32, nt = GHp / 64;
    if (Hp % 32 || GHp % 64) return false;
    const int pps = ((npos + nslices - 1) / nslices + 3) / 4 * 4;
    const int split = dhc ? 2 * Hp : GHp;
#define WG(MT, NT) wgrad_kernel<MT, NT><<<nslices, 512, 0, s>>>(hs, dxt, dhc, slabs, Hp, GHp, split, npos, pps)
    if (mt == 4 && nt == 6) WG(4, 6); else if (mt == 4 && nt == 8) WG(4, 8); else if (mt == 4 && nt == 2) WG(4, 2);
    else if (mt == 2 && nt == 3) WG(2, 3); else if (mt == 2 && nt == 4) WG(2, 4); else if (mt == 2 && nt == 1) WG(2, 1);
    else if (mt == 1 && nt == 2) WG(1, 2); else return false;
#undef WG
    *err = hipGetLastError();
    return true;
}
