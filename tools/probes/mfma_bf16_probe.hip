// Probe: operand layout of v_mfma_f32_16x16x32_bf16 on gfx950 and accuracy of the 3-way bf16 split
// ("bf16x6": 6 bf16 MFMAs reproduce an f32 product to f32 rounding).  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline void split3(float v, __bf16& b1, __bf16& b2, __bf16& b3) {
    b1 = (__bf16)v; float r = v - (float)b1; b2 = (__bf16)r; r -= (float)b2; b3 = (__bf16)r;
}
// A: 16x32 row-major, B: 32x16 row-major, D: 16x16.  mode 0: single bf16 product; 1: bf16x6
__global__ void probe(const float* A, const float* B, float* D, int mode) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    bf16x8 a[3], b[3];
    for (int e = 0; e < 8; ++e) {
        __bf16 x1, x2, x3;
        split3(A[i * 32 + 8 * q + e], x1, x2, x3); a[0][e] = x1; a[1][e] = x2; a[2][e] = x3;
        split3(B[(8 * q + e) * 16 + i], x1, x2, x3); b[0][e] = x1; b[1][e] = x2; b[2][e] = x3;
    }
    f32x4 acc = {0, 0, 0, 0};
    if (mode == 0) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[2], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[1], b[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[0], b[0], acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) D[(4 * q + r) * 16 + i] = acc[r];
}
int main() {
    float hA[512], hB[512], hD[256], *dA, *dB, *dD;
    srand(1);
    for (int k = 0; k < 512; ++k) { hA[k] = (float)rand() / RAND_MAX * 2 - 1; hB[k] = (float)rand() / RAND_MAX * 2 - 1; }
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
    hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(dA, dB, dD, mode);
        hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
        double worst = 0, worst32 = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double ref = 0; float f = 0;
            for (int k = 0; k < 32; ++k) { ref += (double)hA[i * 32 + k] * hB[k * 16 + j]; f = fmaf(hA[i * 32 + k], hB[k * 16 + j], f); }
            worst = fmax(worst, fabs(hD[i * 16 + j] - ref)); worst32 = fmax(worst32, fabs(f - ref));
        }
        printf("mode %d (%s): max |D - fp64 ref| = %.3e   (plain f32 fmaf chain: %.3e)\n", mode, mode ? "bf16x6" : "bf16x1", worst, worst32);
    }
    return 0;
}
