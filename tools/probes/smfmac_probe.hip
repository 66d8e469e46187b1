// Probe: v_smfmac_f32_16x16x64_f16 on gfx950 -- (1) operand layout, found empirically with one-hot A operands and coded B
// operands, then checked with a random sparse product against the host; (2) issue rate against the dense
// v_mfma_f32_16x16x32_f16 for the recurrent chains' patterns (independent accumulators / one dependent chain, one and two
// waves per SIMD).  Why: the 4-row tiles of rec_*_x6p fill the 16 tile rows with plane copies; with the A operand 2:4-sparse the
// rows can select DIFFERENT weight planes out of an interleaved B operand, i.e. one instruction per (gate, k-block) instead of two.
// Build: hipcc --offload-arch=gfx950 -O3 -o smfmac_probe smfmac_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));

template <int ABID>
__device__ __forceinline__ f32x4 smf(const f16x8& a, const f16x16& b, const f32x4& c, int idx) {
    return __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, ABID);
}

// one wave per block; block = case: one-hot A (lane_a, e_a), every 2-bit index field = sel; B element (lane, e) = 16 lane + e
__global__ void __launch_bounds__(64) dump_kernel(float* out, int abid) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int sel = c & 3, e_a = (c >> 2) & 7, lane_a = c >> 5;
    f16x8 a;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)((lane == lane_a && e == e_a) ? 1.0f : 0.0f);
    f16x16 b;
    for (int e = 0; e < 16; ++e) b[e] = (_Float16)(float)(16 * lane + e);
    int idx = sel * 0x5555;                                  // low half: every field = sel
    idx |= ((sel ^ 1) * 0x5555) << 16;                       // high half: every field = sel ^ 1 (to see which half ABID takes)
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = abid ? smf<1>(a, b, d, idx) : smf<0>(a, b, d, idx);
    for (int k = 0; k < 4; ++k) out[((size_t)c * 64 + lane) * 4 + k] = d[k];
}

// general product with caller-given operands (one wave)
__global__ void __launch_bounds__(64) prod_kernel(const _Float16* A, const _Float16* B, const int* idx, float* out) {
    const int lane = threadIdx.x;
    f16x8 a; f16x16 b;
    for (int e = 0; e < 8; ++e) a[e] = A[lane * 8 + e];
    for (int e = 0; e < 16; ++e) b[e] = B[lane * 16 + e];
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    d = smf<0>(a, b, d, idx[lane]);
    for (int k = 0; k < 4; ++k) out[lane * 4 + k] = d[k];
}

// timing.  mode 0: dense mfma 16x16x32 f16, NACC accumulators round-robin; mode 1: smfmac 16x16x64 f16 likewise.
template <int MODE, int NACC>
__global__ void __launch_bounds__(512) time_kernel(float* out, unsigned long long* cyc, int steps) {
    const int wave = threadIdx.x >> 6;
    f16x8 a[3]; f16x16 b[3];
    for (int i = 0; i < 3; ++i) {
        for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.001f * ((threadIdx.x & 63) + i + e));
        for (int e = 0; e < 16; ++e) b[i][e] = (_Float16)(0.002f * ((threadIdx.x & 63) + 3 * i + e));
    }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int idx = 0x88888888;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            if (MODE == 0) {
                f16x8 bb;
                for (int e = 0; e < 8; ++e) bb[e] = b[k % 3][e];
                acc[k % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(k + 1) % 3], bb, acc[k % NACC], 0, 0, 0);
            } else {
                acc[k % NACC] = smf<0>(a[(k + 1) % 3], b[k % 3], acc[k % NACC], idx);
            }
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float r = 0.f;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// the chain's step shape: zero / bias the accumulator from LDS (ds_read_b128), 12 smfmac, combine (VALU) -- against v_mov init
template <int INIT>      // 0: v_mov zero init, 1: ds_read_b128 init
__global__ void __launch_bounds__(512) step_kernel(float* out, unsigned long long* cyc, int steps) {
    __shared__ __attribute__((aligned(16))) float zer[512 * 4];
    const int wave = threadIdx.x >> 6;
    for (int k = 0; k < 4; ++k) zer[threadIdx.x * 4 + k] = k == 0 ? 1e-3f * threadIdx.x : 0.f;
    f16x8 a[3]; f16x16 b[3];
    for (int i = 0; i < 3; ++i) {
        for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.001f * ((threadIdx.x & 63) + i + e));
        for (int e = 0; e < 16; ++e) b[i][e] = (_Float16)(0.002f * ((threadIdx.x & 63) + 3 * i + e));
    }
    const int idx = 0x88888888;
    float r = 0.f;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            if (INIT) { asm volatile("" ::: "memory"); acc[g] = *(const f32x4*)(zer + threadIdx.x * 4); }
            else { acc[g] = f32x4{r, 0.f, 0.f, 0.f}; asm volatile("" : "+v"(acc[g])); }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[g] = smf<0>(a[(k + g) % 3], b[(k + 2 * g) % 3], acc[g], idx);
#pragma unroll
        for (int g = 0; g < 3; ++g) r += fmaf(acc[g][1] + acc[g][2], 1.0f / 2048.0f, acc[g][0]) * 1e-6f;
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

int main() {
    // ---- 1. layout
    const int NC = 64 * 8 * 4;
    float* d_out; hipMalloc(&d_out, (size_t)NC * 256 * 4);
    std::vector<float> h((size_t)NC * 256);
    int map_qb[2][64][8][4], map_eb[2][64][8][4], map_m[2][64][8][4];
    for (int abid = 0; abid < 2; ++abid) {
        dump_kernel<<<NC, 64>>>(d_out, abid);
        if (hipDeviceSynchronize() != hipSuccess) { printf("dump kernel failed\n"); return 1; }
        hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int c = 0; c < NC; ++c) {
            const int sel = c & 3, e_a = (c >> 2) & 7, lane_a = c >> 5;
            int m = -1, qb = -1, eb = -1, cnt = 0, ok = 1;
            for (int l = 0; l < 64; ++l) for (int k = 0; k < 4; ++k) {
                const float v = h[((size_t)c * 64 + l) * 4 + k];
                if (v == 0.f) continue;            // (B code 0 = lane 0, e 0 is invisible: handled by the count below)
                ++cnt;
                const int mm = 4 * (l >> 4) + k, n = l & 15, code = (int)v, lb = code / 16, e = code % 16;
                if ((lb & 15) != n) ok = 0;
                if (m < 0) { m = mm; qb = lb >> 4; eb = e; }
                else if (m != mm || qb != (lb >> 4) || eb != e) ok = 0;
            }
            if (!ok || cnt < 15 || cnt > 16) { if (bad < 12) printf("abid %d case lane_a %d e_a %d sel %d: irregular (cnt %d m %d qb %d eb %d)\n", abid, lane_a, e_a, sel, cnt, m, qb, eb); ++bad; }
            map_m[abid][lane_a][e_a][sel] = m; map_qb[abid][lane_a][e_a][sel] = qb; map_eb[abid][lane_a][e_a][sel] = eb;
        }
        printf("abid %d: %d irregular cases of %d\n", abid, bad, NC);
        // does the map depend only on (qa, e_a, sel), and is the row lane_a & 15?
        int dep = 0, rowbad = 0;
        for (int la = 0; la < 64; ++la) for (int e = 0; e < 8; ++e) for (int s = 0; s < 4; ++s) {
            if (map_m[abid][la][e][s] != (la & 15)) ++rowbad;
            const int l0 = la & 48;
            if (map_qb[abid][la][e][s] != map_qb[abid][l0][e][s] || map_eb[abid][la][e][s] != map_eb[abid][l0][e][s]) ++dep;
        }
        printf("abid %d: row != lane_a %% 16 in %d cases; map differs between rows of one lane group in %d cases\n", abid, rowbad, dep);
        printf("abid %d: A (qa, e_a) with index field value sel -> B (qb, e_b)   [the low idx half holds sel, the high half sel ^ 1]\n", abid);
        for (int qa = 0; qa < 4; ++qa) {
            printf("  qa %d:", qa);
            for (int e = 0; e < 8; ++e) {
                printf("  e%d[", e);
                for (int s = 0; s < 4; ++s) printf("%d.%d%s", map_qb[abid][qa * 16][e][s], map_eb[abid][qa * 16][e][s], s < 3 ? " " : "");
                printf("]");
            }
            printf("\n");
        }
    }
    // ---- 1b. random sparse product against the host, with the map found for abid 0
    {
        std::vector<_Float16> A(64 * 8), B(64 * 16); std::vector<int> idx(64);
        srand(7);
        for (auto& v : A) v = (_Float16)(float)(rand() % 7 - 3);
        for (auto& v : B) v = (_Float16)(float)(rand() % 9 - 4);
        int selof[64][8];
        for (int l = 0; l < 64; ++l) {
            int w = 0;
            for (int g = 0; g < 4; ++g) {
                int s0 = rand() % 4, s1 = rand() % 4;
                while (s1 == s0) s1 = rand() % 4;
                if (s0 > s1) { int t = s0; s0 = s1; s1 = t; }
                selof[l][2 * g] = s0; selof[l][2 * g + 1] = s1;
                w |= (s0 | (s1 << 2)) << (4 * g);
            }
            idx[l] = w | (0x1234 << 16);
        }
        _Float16 *dA, *dB; int* dI; float* dO;
        hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dI, 256); hipMalloc(&dO, 1024);
        hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dI, idx.data(), 256, hipMemcpyHostToDevice);
        prod_kernel<<<1, 64>>>(dA, dB, dI, dO);
        float o[256]; hipMemcpy(o, dO, 1024, hipMemcpyDeviceToHost);
        int wrong = 0;
        for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
            float want = 0.f;
            for (int qa = 0; qa < 4; ++qa) for (int e = 0; e < 8; ++e) {
                const int la = qa * 16 + m, s = selof[la][e];
                const int qb = map_qb[0][la][e][s], eb = map_eb[0][la][e][s];
                if (qb < 0) continue;
                want += (float)A[la * 8 + e] * (float)B[(qb * 16 + n) * 16 + eb];
            }
            const float got = o[((m >> 2) * 16 + n) * 4 + (m & 3)];
            if (got != want) { if (wrong < 8) printf("  product: D[%d][%d] = %g, expected %g\n", m, n, got, want); ++wrong; }
        }
        printf("random sparse product (index fields bits [2e+1 : 2e] of the lane's own register, ascending within a group): %d of 256 wrong\n", wrong);
    }
    // ---- 2. timing
    float* out; unsigned long long* cyc; hipMalloc(&out, 64 * 512 * 4); hipMalloc(&cyc, 64 * 8 * 8);
    unsigned long long hc[8];
    const int steps = 4000;
#define RUN(NAME, KERNEL, THREADS) do { KERNEL<<<4, THREADS>>>(out, cyc, steps); hipDeviceSynchronize(); \
        hipMemcpy(hc, cyc, sizeof(hc), hipMemcpyDeviceToHost); \
        printf("%-58s %d waves: cycles per 12 ops, wave 0 %.0f", NAME, THREADS / 64, (double)hc[0] / steps); \
        if (THREADS > 256) printf(", wave 4 %.0f", (double)hc[4] / steps); printf("\n"); } while (0)
    RUN("dense mfma 16x16x32 f16, 3 accumulators", (time_kernel<0, 3>), 256);
    RUN("dense mfma 16x16x32 f16, 3 accumulators", (time_kernel<0, 3>), 512);
    RUN("dense mfma 16x16x32 f16, 2 accumulators", (time_kernel<0, 2>), 256);
    RUN("dense mfma 16x16x32 f16, 1 accumulator (dependent chain)", (time_kernel<0, 1>), 256);
    RUN("smfmac 16x16x64 f16, 3 accumulators", (time_kernel<1, 3>), 256);
    RUN("smfmac 16x16x64 f16, 3 accumulators", (time_kernel<1, 3>), 512);
    RUN("smfmac 16x16x64 f16, 2 accumulators", (time_kernel<1, 2>), 256);
    RUN("smfmac 16x16x64 f16, 2 accumulators", (time_kernel<1, 2>), 512);
    RUN("smfmac 16x16x64 f16, 1 accumulator (dependent chain)", (time_kernel<1, 1>), 256);
    RUN("smfmac 16x16x64 f16, 1 accumulator (dependent chain)", (time_kernel<1, 1>), 512);
    RUN("step shape: v_mov init, 12 smfmac, combine", (step_kernel<0>), 256);
    RUN("step shape: v_mov init, 12 smfmac, combine", (step_kernel<0>), 512);
    RUN("step shape: ds_read_b128 init, 12 smfmac, combine", (step_kernel<1>), 256);
    RUN("step shape: ds_read_b128 init, 12 smfmac, combine", (step_kernel<1>), 512);
    return 0;
}
