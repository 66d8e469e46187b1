// Probe: could the recurrent chain's f32 products run as a 2-way fp16 split with 3 MFMAs ("f16x3") instead of the 3-way
// bf16 split with 6 ("bf16x6")?  K = 128 dot products (one recurrent step: h . W_hid column) against fp64, for
//   fmaf     plain f32 FMA chain                       bf16x6   what the kernels run
//   f16x3    a = a1 + a2 in fp16, a1b1 + a1b2 + a2b1   f16x3s   low parts scaled by 2^11 (kept out of the subnormal range),
//                                                               own accumulator, combined with one FMA
// on h in [-1, 1], W ~ N(0, 0.1), and the same with tiny activations (|h| ~ 1e-4: fp16 subnormal low parts).
// Also prints whether the matrix pipe flushes fp16 subnormal inputs.    Build: tools/probes/build.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define KB 4

__device__ inline void split3(float v, __bf16& b1, __bf16& b2, __bf16& b3) {
    b1 = (__bf16)v; float r = v - (float)b1; b2 = (__bf16)r; r -= (float)b2; b3 = (__bf16)r;
}
// A: 16 x 128 row-major, B: 128 x 16 row-major, D: 16 x 16
__global__ void probe(const float* A, const float* B, float* D, int mode) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    for (int kb = 0; kb < KB; ++kb) {
        float av[8], bv[8];
        for (int e = 0; e < 8; ++e) { av[e] = A[i * 128 + kb * 32 + 8 * q + e]; bv[e] = B[(kb * 32 + 8 * q + e) * 16 + i]; }
        if (mode == 1) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                __bf16 x1, x2, x3;
                split3(av[e], x1, x2, x3); a[0][e] = x1; a[1][e] = x2; a[2][e] = x3;
                split3(bv[e], x1, x2, x3); b[0][e] = x1; b[1][e] = x2; b[2][e] = x3;
            }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
        } else {
            const float S = mode == 3 ? 2048.f : 1.f;
            f16x8 a1, a2, b1, b2;
            for (int e = 0; e < 8; ++e) {
                a1[e] = (_Float16)av[e]; a2[e] = (_Float16)((av[e] - (float)a1[e]) * S);
                b1[e] = (_Float16)bv[e]; b2[e] = (_Float16)((bv[e] - (float)b1[e]) * S);
            }
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, acc2, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, acc, 0, 0, 0);
        }
    }
    const float inv = mode == 3 ? 1.f / 2048.f : 1.f;
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = fmaf(acc2[r], inv, acc[r]);
}
__global__ void subnormal(float* out) {      // one product of two values whose fp16 forms are subnormal x normal
    const int l = threadIdx.x;
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    if ((l >> 4) == 0) { a[0] = (_Float16)3.0e-6f; b[0] = (_Float16)1024.f; }     // 3e-6 is subnormal in fp16 (min normal 6.1e-5)
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (l == 0) out[0] = acc[0];
}
static double gauss() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
int main() {
    static float hA[16 * 128], hB[128 * 16], hD[256];
    float *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    subnormal<<<1, 64>>>(dD); hipMemcpy(hD, dD, 4, hipMemcpyDeviceToHost);
    printf("fp16 subnormal input 3.0e-6 * 1024 through the matrix pipe = %.6e (exact %.6e): %s\n", hD[0], (double)(float)(_Float16)3.0e-6f * 1024,
           hD[0] == 0.f ? "FLUSHED" : "kept");
    const char* names[4] = {"fmaf  ", "bf16x6", "f16x3 ", "f16x3s"};
    for (int scen = 0; scen < 3; ++scen) {
        srand(7 + scen);
        const double hs = scen == 0 ? 1.0 : scen == 1 ? 1e-4 : 1.0, ws = scen == 2 ? 3.0 : 0.1;
        for (int k = 0; k < 16 * 128; ++k) hA[k] = (float)(((double)rand() / RAND_MAX * 2 - 1) * hs);
        for (int k = 0; k < 128 * 16; ++k) hB[k] = (float)(gauss() * ws);
        hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
        printf("scenario %d: |h| <= %g, W ~ N(0, %g)\n", scen, hs, ws);
        for (int mode = 0; mode < 4; ++mode) {
            if (mode) { probe<<<1, 64>>>(dA, dB, dD, mode); hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); }
            double worst = 0, se = 0, sr = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
                double ref = 0; float f = 0;
                for (int k = 0; k < 128; ++k) { ref += (double)hA[i * 128 + k] * hB[k * 16 + j]; f = fmaf(hA[i * 128 + k], hB[k * 16 + j], f); }
                const double got = mode ? hD[i * 16 + j] : f, e = fabs(got - ref);
                worst = fmax(worst, e); se += e * e; sr += ref * ref;
            }
            printf("   %s  max err %.3e   rms err / rms value %.3e\n", names[mode], worst, sqrt(se / sr));
        }
    }
    return 0;
}
