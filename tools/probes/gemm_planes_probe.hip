// Probe (round 6; VERDICT rounds 4 / 5, "gemm_x6_kernel on operand planes written once"): an f32-class GEMM whose operands arrive as the
// two fp16 planes of the 2-way split (a = a1 + a2 / 2048), both k-contiguous, moved global -> LDS by LDS-DMA (no register round trip, no
// VALU split), two stages deep, products a1 b1 + (a1 b2 + a2 b1) / 2048 as gemm_x6_kernel NP = 2.
//
//   C[m][n] = so * sum_k A(m, k) * B(n, k)  (+ bias[n])
//
// Measured inside the library on C5's layer GEMMs (tools/gemm_planes_probe.py at commit "plane GEMM probe", profiles/round6_m_gemm_planes.txt):
// input projection (51200 x 2048 x 512) 400 us against gemm_x6_kernel's 570 (268 against 188 TFLOP/s f32-equivalent), its backward
// (51200 x 512 x 2048) 403 against 537; same errors against float64 (3.4e-7 / 6.6e-7: the same arithmetic).  1.4x, not the 2 - 3x the
// instruction counts suggest: 32 KB of LDS-DMA per workgroup and k step is 8.2 TB/s chip-wide = 32 GB/s per CU, the fill rate of the LDS-DMA
// path (MI355X_MICROARCH.md: ldsdma-fill).  NOT integrated: the producers (recurrent chains) would have to store their fp16 pairs as plane
// arrays -- K-permuted for the gradient --, the contraction-over-time GEMMs (dW) need transposed planes, and a split pre-pass costs what
// the GEMM gains (420 MB of dxt at C5: 140 us against 135); what it could buy is ~4 % of C5.  Kept as the measured ceiling of the idea.
// Workgroup = 256 threads, 128 x 128 tile, k step 32; a stage of LDS is [A | B][plane][128 rows][64 bytes], the 16-byte chunk of a row
// XOR-swizzled by (row >> 2) & 3 on the GLOBAL side of the copy (an LDS-DMA instruction writes lane l at M0 + 16 l).
// Build: hipcc --offload-arch=gfx950 -O3 -o gemm_planes_probe gemm_planes_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void lds_dma_x4(unsigned lds_addr, const void* ubase, unsigned boff, unsigned long long mask) {
    asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                 :: "s"(lds_addr), "v"(boff), "s"(ubase), "s"(mask) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int CNT>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CNT) : "memory"); }

typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4p __attribute__((ext_vector_type(4)));

struct GemmPlArgs {
    const _Float16* A1; const _Float16* A2; long lda;      // [M][lda]
    const _Float16* B1; const _Float16* B2; long ldb;      // [N][ldb]
    float* C; long ldc;
    const float* bias;
    int M, N, K;
    float so;
};

#define GPL_STAGE 32768
__global__ void __launch_bounds__(256, 2) gemm_pl_kernel(GemmPlArgs g) {
    extern __shared__ __attribute__((aligned(16))) char sm_pl[];      // [2 stages][A, B][2 planes][128 rows][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, j = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    // copy role: operand-plane `wave` (0: A1, 1: A2, 2: B1, 3: B2); piece b = rows 16 b .. 16 b + 15, lane l -> row l >> 2, LDS chunk l & 3
    const _Float16* src = wave == 0 ? g.A1 : wave == 1 ? g.A2 : wave == 2 ? g.B1 : g.B2;
    const long ld = wave < 2 ? g.lda : g.ldb;
    const char* base = (const char*)(src + (long)(wave < 2 ? m0 : n0) * ld);
    unsigned voff[8];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const int row = b * 16 + (lane >> 2);
        voff[b] = (unsigned)((long)row * ld * 2 + (((lane & 3) ^ ((row >> 2) & 3)) << 4));
    }
    const unsigned lds0 = (unsigned)(size_t)sm_pl;       // (LDS addresses are 32-bit)
    const unsigned my_plane = lds0 + wave * 8192;
    auto issue = [&](int k0, int stage) {
        const char* kb = base + (long)k0 * 2;
#pragma unroll
        for (int b = 0; b < 8; ++b) lds_dma_x4(my_plane + stage * GPL_STAGE + b * 1024, kb, voff[b], ~0ull);
    };
    // fragment addresses: A tile mi -> row wm*64 + mi*16 + j, B tile ni -> row wn*64 + ni*16 + j; chunk q swizzled by the row
    unsigned fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 64 + t * 16 + j, rb = wn * 64 + t * 16 + j;
        fa[t] = (unsigned)(ra * 64 + ((q ^ ((ra >> 2) & 3)) << 4));
        fb[t] = (unsigned)(16384 + rb * 64 + ((q ^ ((rb >> 2) & 3)) << 4));
    }
    const f32x4 z = f32x4{0, 0, 0, 0};
    f32x4 acc[4][4], acl[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { acc[a][b] = z; acl[a][b] = z; }

    issue(0, 0);
    for (int k0 = 0, st = 0; k0 < g.K; k0 += 32, st ^= 1) {
        const bool more = k0 + 32 < g.K;                 // uniform
        if (more) {
            issue(k0 + 32, st ^ 1);                      // (everybody left that stage behind the barrier at the end of the last iteration)
            wait_vm<8>();                                // this wave's pieces of the current stage have landed (the counter retires in order)
        } else wait_vm<0>();
        __syncthreads();                                 // ... and everybody else's
        const char* sp = sm_pl + st * GPL_STAGE;
        f16x8p a1[4], a2[4], b1[4], b2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a1[t] = *(const f16x8p*)(sp + fa[t]); a2[t] = *(const f16x8p*)(sp + 8192 + fa[t]);
            b1[t] = *(const f16x8p*)(sp + fb[t]); b2[t] = *(const f16x8p*)(sp + 8192 + fb[t]);
        }
        // products as mfma(B rows, A rows): the accumulators hold the transposed tiles (lane (j, q): row m = j, columns 4q .. 4q + 3)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acl[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2[ni], a1[mi], acl[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acl[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[ni], a2[mi], acl[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
        __syncthreads();                                 // the stage is free for the copy of k0 + 64
    }
    float* out = g.C + (long)m0 * g.ldc + n0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = wn * 64 + ni * 16 + 4 * q;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(acl[mi][ni][r], 1.0f / 2048.0f, acc[mi][ni][r]) * g.so;
            if (g.bias) {
                const f32x4 bv = *(const f32x4*)(g.bias + n0 + n);
                v += bv;
            }
            *(f32x4*)(out + (long)(wm * 64 + mi * 16 + j) * g.ldc + n) = v;
        }
}

// ---------------------------------------------------------------------------------------
// The same on a 256 x 128 x 32 tile (round 6, second probe): 512 threads, eight waves of 64 x 64 outputs, THREE LDS stages of 48 KB
// ([A1 | A2: 256 rows][B1 | B2: 128 rows] x 64 bytes), the copy of k step i + 2 issued behind the one barrier of step i, each wave six
// 1 KB pieces per stage.  No split, no operand registers in flight, plane-0 fragments first so that the first sixteen MFMAs start while
// plane 1 is still being read.
// ---------------------------------------------------------------------------------------
#define GPW_STAGE 49152
__global__ void __launch_bounds__(512, 1) gemm_pl256_kernel(GemmPlArgs g) {
    extern __shared__ __attribute__((aligned(16))) char sm_pw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, j = lane & 15, q = lane >> 4;
    const int tiles_n = g.N / 128, nt = (g.M / 256) * tiles_n;
    int lin = blockIdx.x;
    if ((nt & 7) == 0) lin = (lin & 7) * (nt >> 3) + (lin >> 3);
    const int m0 = (lin / tiles_n) * 256, n0 = (lin % tiles_n) * 128;
    // this wave's six pieces of a stage: piece p = 6 wave + i; p < 16: A1 rows 16 p .., < 32: A2, < 40: B1, else B2
    const char* pb[6]; unsigned pl[6]; bool isb[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int p = wave * 6 + i;
        const _Float16* src = p < 16 ? g.A1 : p < 32 ? g.A2 : p < 40 ? g.B1 : g.B2;
        const int rb = p < 16 ? p : p < 32 ? p - 16 : p < 40 ? p - 32 : p - 40;
        isb[i] = p >= 32;
        const long ld = isb[i] ? g.ldb : g.lda;
        pb[i] = (const char*)(src + ((long)(isb[i] ? n0 : m0) + rb * 16) * ld);
        pl[i] = (unsigned)p * 1024u;
    }
    const unsigned swz = (unsigned)(((lane & 3) ^ ((lane >> 4) & 3)) << 4);
    const unsigned voa = (unsigned)((long)(lane >> 2) * g.lda * 2) + swz, vob = (unsigned)((long)(lane >> 2) * g.ldb * 2) + swz;
    const unsigned lds0 = (unsigned)(size_t)sm_pw;
    auto issue = [&](int k0, int stage) {
#pragma unroll
        for (int i = 0; i < 6; ++i) lds_dma_x4(lds0 + stage * GPW_STAGE + pl[i], pb[i] + (long)k0 * 2, isb[i] ? vob : voa, ~0ull);
    };
    unsigned fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 64 + t * 16 + j, rb = wn * 64 + t * 16 + j;
        fa[t] = (unsigned)(ra * 64 + ((q ^ ((ra >> 2) & 3)) << 4));
        fb[t] = (unsigned)(32768 + rb * 64 + ((q ^ ((rb >> 2) & 3)) << 4));
    }
    const f32x4 z = f32x4{0, 0, 0, 0};
    f32x4 acc[4][4], acl[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { acc[a][b] = z; acl[a][b] = z; }
    const int ns = g.K / 32;
    issue(0, 0);
    if (ns > 1) issue(32, 1);
    int st = 0;
    for (int i = 0; i < ns; ++i) {
        if (i + 1 < ns) wait_vm<6>(); else wait_vm<0>();          // this wave's pieces of stage i (the counter retires in order)
        __syncthreads();                                          // ... everybody's; and everybody has read stage i - 1
        if (i + 2 < ns) issue((i + 2) * 32, st == 0 ? 2 : st - 1);
        const char* sp = sm_pw + st * GPW_STAGE;
        f16x8p a1[4], a2[4], b1[4], b2[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { a1[t] = *(const f16x8p*)(sp + fa[t]); b1[t] = *(const f16x8p*)(sp + fb[t]); }
#pragma unroll
        for (int t = 0; t < 4; ++t) { a2[t] = *(const f16x8p*)(sp + 16384 + fa[t]); b2[t] = *(const f16x8p*)(sp + 8192 + fb[t]); }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[ni], a1[mi], acc[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acl[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b2[ni], a1[mi], acl[mi][ni], 0, 0, 0);
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) acl[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1[ni], a2[mi], acl[mi][ni], 0, 0, 0);
        st = st == 2 ? 0 : st + 1;
    }
    float* out = g.C + (long)m0 * g.ldc + n0;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int n = wn * 64 + ni * 16 + 4 * q;
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(acl[mi][ni][r], 1.0f / 2048.0f, acc[mi][ni][r]) * g.so;
            if (g.bias) { const f32x4 bv = *(const f32x4*)(g.bias + n0 + n); v += bv; }
            *(f32x4*)(out + (long)(wm * 64 + mi * 16 + j) * g.ldc + n) = v;
        }
}
bool launch_gemm_planes256(hipStream_t s, const _Float16* A1, const _Float16* A2, long lda, const _Float16* B1, const _Float16* B2, long ldb,
                           float* C, long ldc, int M, int N, int K, const float* bias, float so, hipError_t* err) {
    if ((M & 255) || (N & 127) || (K & 31) || (lda & 7) || (ldb & 7) || (ldc & 3)) return false;
    if ((long)16 * lda * 2 >= (1l << 31) || (long)16 * ldb * 2 >= (1l << 31)) return false;
    GemmPlArgs g{A1, A2, lda, B1, B2, ldb, C, ldc, bias, M, N, K, so};
    (void)hipFuncSetAttribute((const void*)gemm_pl256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * GPW_STAGE);
    gemm_pl256_kernel<<<dim3((M / 256) * (N / 128)), 512, 3 * GPW_STAGE, s>>>(g);
    *err = hipGetLastError();
    return true;
}

// true = launched (err holds the launch status); false: shape / alignment not served
bool launch_gemm_planes(hipStream_t s, const _Float16* A1, const _Float16* A2, long lda, const _Float16* B1, const _Float16* B2, long ldb,
                        float* C, long ldc, int M, int N, int K, const float* bias, float so, hipError_t* err) {
    if ((M & 127) || (N & 127) || (K & 31) || (lda & 7) || (ldb & 7) || (ldc & 3)) return false;
    if ((((uintptr_t)A1 | (uintptr_t)A2 | (uintptr_t)B1 | (uintptr_t)B2 | (uintptr_t)C) & 15) || (bias && ((uintptr_t)bias & 15))) return false;
    if ((long)128 * lda * 2 >= (1l << 31) || (long)128 * ldb * 2 >= (1l << 31)) return false;
    GemmPlArgs g{A1, A2, lda, B1, B2, ldb, C, ldc, bias, M, N, K, so};
    (void)hipFuncSetAttribute((const void*)gemm_pl_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * GPL_STAGE);
    gemm_pl_kernel<<<dim3(N / 128, M / 128), 256, 2 * GPL_STAGE, s>>>(g);
    *err = hipGetLastError();
    return true;
}

// ---------------------------------------------------------------------------------------
// f32 -> the two fp16 planes of scale * x (one streaming pass: weights once per step; test hook for activations)
// rows x cols, input row stride ld_in (floats); output [rows][ld_out] fp16.  TRANSPOSE: out[c][r] = in[r][c] (a weight stored
// [K][N] as the B operand [N][K]); PERM4: output column = (c % gw) * 4 + c / gw for c < 4 gw (gate-major columns -> unit-major with
// the gate fastest: the K order in which rec_bwd_c16 holds a thread's four gate gradients)
// ---------------------------------------------------------------------------------------
template <bool TRANSPOSE>
__global__ void __launch_bounds__(256) split_planes_kernel(const float* __restrict__ in, long ld_in, int rows, int cols, float scale,
                                                           _Float16* __restrict__ o1, _Float16* __restrict__ o2, long ld_out, int perm_gw) {
    __shared__ float tile[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(long)r * ld_in + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int orow, ocol; float v;
        if (TRANSPOSE) { orow = c0 + ty + 8 * i; ocol = r0 + tx; v = tile[tx][ty + 8 * i]; if (orow >= cols || ocol >= rows) continue; }
        else { orow = r0 + ty + 8 * i; ocol = c0 + tx; v = tile[ty + 8 * i][tx]; if (orow >= rows || ocol >= cols) continue; }
        if (perm_gw > 0) {                               // (the contraction index of this operand is its column / transposed row)
            int& kk = TRANSPOSE ? orow : ocol;
            if (kk < 4 * perm_gw) kk = (kk % perm_gw) * 4 + kk / perm_gw;
        }
        float x = v * scale;
        asm("" : "+v"(x));                               // one rounding to fp16 for both uses (split2_f16)
        const _Float16 a1 = (_Float16)x;
        o1[(long)orow * ld_out + ocol] = a1;
        o2[(long)orow * ld_out + ocol] = (_Float16)((x - (float)a1) * 2048.0f);
    }
}
hipError_t launch_split_planes(hipStream_t s, const float* in, long ld_in, int rows, int cols, float scale, _Float16* o1, _Float16* o2,
                               long ld_out, bool transpose, int perm_gw) {
    const dim3 grid((cols + 31) / 32, (rows + 31) / 32);
    if (transpose) split_planes_kernel<true><<<grid, 256, 0, s>>>(in, ld_in, rows, cols, scale, o1, o2, ld_out, perm_gw);
    else split_planes_kernel<false><<<grid, 256, 0, s>>>(in, ld_in, rows, cols, scale, o1, o2, ld_out, perm_gw);
    return hipGetLastError();
}

int main() {
    const int M = 51200, N = 2048, K = 512;
    std::vector<float> hA((size_t)M * K), hB((size_t)K * N);
    srand(3);
    for (auto& x : hA) x = 2.0f * rand() / RAND_MAX - 1.0f;
    for (auto& x : hB) x = 0.2f * rand() / RAND_MAX - 0.1f;
    float *dA, *dB, *dC; _Float16 *a1, *a2, *b1, *b2;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dB, hB.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
    hipMalloc(&a1, hA.size() * 2); hipMalloc(&a2, hA.size() * 2); hipMalloc(&b1, hB.size() * 2); hipMalloc(&b2, hB.size() * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    launch_split_planes(0, dA, K, M, K, 1.0f, a1, a2, K, false, 0);
    launch_split_planes(0, dB, N, K, N, 1.0f, b1, b2, K, true, 0);
    hipError_t e = hipSuccess;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
    hipMemset(dC, 0, (size_t)M * N * 4);
    auto go = [&]() { if (which) launch_gemm_planes256(0, a1, a2, K, b1, b2, K, dC, N, M, N, K, nullptr, 1.0f, &e);
                      else launch_gemm_planes(0, a1, a2, K, b1, b2, K, dC, N, M, N, K, nullptr, 1.0f, &e); };
    for (int i = 0; i < 3; ++i) go();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) go();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<float> hC((size_t)N);
    double worst = 0, big = 0;
    for (int m = 0; m < M; m += 997) {
        hipMemcpy(hC.data(), dC + (size_t)m * N, (size_t)N * 4, hipMemcpyDeviceToHost);
        for (int n = 0; n < N; n += 61) {
            double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)m * K + k] * hB[(size_t)k * N + n];
            worst = fmax(worst, fabs(s - hC[n])); big = fmax(big, fabs(s));
        }
    }
    printf("planes GEMM (%s tile) %d x %d x %d: %.1f us, %.1f TFLOP/s f32-equivalent, max error %.2e of the largest entry (%s)\n",
           which ? "256 x 128, three stages" : "128 x 128, two stages", M, N, K, ms * 100.0,
           2.0 * M * N * K / (ms * 100.0) / 1e6, worst / big, hipGetErrorString(e));
    }
    return 0;
}
