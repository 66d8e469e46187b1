// Probe: v_mfma_f32_16x16x32_bf16 issue rate for the recurrent kernels' pattern (3 independent accumulators, 72 MFMAs
// per "step") with one and with two waves per SIMD, and with VALU filler work in the partner wave.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_share_probe mfma_share_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)

// mode 0: every wave runs MFMAs.  mode 1: waves >= 4 run dependent VALU chains instead (a partner in its gate math).
// mode 2: waves >= 4 idle (s_sleep).
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc, int steps, int mode) {
    const int wave = threadIdx.x >> 6;
    bf16x8 a[6], b[3];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (threadIdx.x + i + e));
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)(0.002f * (threadIdx.x + i * 3 + e));
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v = threadIdx.x * 1e-3f;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (mode >= 10 ? (wave == 0 || wave == mode - 10) : (mode == 0 || wave < 4)) {
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int term = 0; term < 6; ++term)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = MF(a[(term + g) % 6], b[(term + k) % 3], acc[g]);
            }
        }
    } else if (mode >= 10) {
    } else if (mode == 1) {
        for (int s = 0; s < steps * 60; ++s) { v = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v)); v = fmaf(v, 1.0001f, 0.5f); v = v * v - 0.3f; }
    } else {
        for (int s = 0; s < steps * 8; ++s) __builtin_amdgcn_s_sleep(8);
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + v;
}
int main() {
    float* out; unsigned long long* cyc; hipMalloc(&out, 64 * 512 * 4); hipMalloc(&cyc, 64 * 8 * 8);
    unsigned long long h[8];
    const int steps = 2000;
    for (int threads = 256; threads <= 512; threads += 256)
        for (int mode = 0; mode < 3; ++mode) {
            if (threads == 256 && mode) continue;
            probe<<<4, threads, 0, 0>>>(out, cyc, steps, mode);
            hipDeviceSynchronize();
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("threads %d mode %d: cycles per 72-MFMA step by wave:", threads, mode);
            for (int w = 0; w < threads / 64; ++w) printf(" %.0f", (double)h[w] / steps);
            printf("\n");
        }
    for (int k = 1; k < 8; ++k) {
        probe<<<4, 512, 0, 0>>>(out, cyc, steps, 10 + k);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("waves 0 and %d only: wave 0 %.0f, wave %d %.0f\n", k, (double)h[0] / steps, k, (double)h[k] / steps);
    }
    return 0;
}
