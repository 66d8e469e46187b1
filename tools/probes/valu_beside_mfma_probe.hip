// Probe: what one wave's VALU work costs while its SIMD partner streams v_mfma_f32_16x16x32_bf16, and what it costs
// the partner.  Waves 0-3: MFMA stream (or idle); waves 4-7: a block of N independent ops of one kind per "step".
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_beside_mfma_probe valu_beside_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MF(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)

template <int KIND>   // 0 fma, 1 exp2, 2 rcp, 3 cndmask, 4 s_nop (issue only), 5 ds_read_b32, 6 dependent exp2->add->rcp chain
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc, int steps, int mfma_on, float* lds_dummy, int prio, int swap) {
    __shared__ float sh[1024];
    int wave = threadIdx.x >> 6;
    if (swap) wave ^= 4;            // swap: the OLDER waves do the VALU work, the younger ones the MFMAs
    sh[threadIdx.x] = threadIdx.x; sh[threadIdx.x + 512] = 1.0f;
    bf16x8 a[6], b[3];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(0.001f * (threadIdx.x + i + e));
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)(0.002f * (threadIdx.x + i * 3 + e));
    f32x4 acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (wave < 4) {
        __builtin_amdgcn_s_setprio(0);
        if (mfma_on)
            for (int s = 0; s < steps; ++s)
#pragma unroll
                for (int k = 0; k < 24; ++k)
#pragma unroll
                    for (int g = 0; g < 3; ++g) acc[g] = MF(a[(k + g) % 6], b[k % 3], acc[g]);
    } else {
        if (prio) __builtin_amdgcn_s_setprio(3);
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int k = 0; k < 8; ++k)          // 64 ops per step, 8 independent chains
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == 0) v[i] = fmaf(v[i], 1.0001f, 0.5f);
                    if (KIND == 1) v[i] = __builtin_amdgcn_exp2f(v[i]);
                    if (KIND == 2) v[i] = __builtin_amdgcn_rcpf(v[i]);
                    if (KIND == 3) v[i] = (threadIdx.x & (1 << k)) ? v[i] : v[(i + 1) & 7];
                    if (KIND == 4) asm volatile("s_nop 0");
                    if (KIND == 5) v[i] += sh[(threadIdx.x + i * 64 + k) & 1023];
                    if (KIND == 6) v[i] = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v[i]));
                }
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float r = acc[0][0] + acc[1][1] + acc[2][2];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int KIND> void run(const char* name, float* out, unsigned long long* cyc, int prio = 0, int swap = 0) {
    unsigned long long h[8];
    const int steps = 2000;
    double res[2][2];
    for (int on = 0; on < 2; ++on) {
        probe<KIND><<<4, 512, 0, 0>>>(out, cyc, steps, on, out, prio, swap);
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        res[on][0] = (double)h[swap ? 4 : 0] / steps; res[on][1] = (double)h[swap ? 0 : 4] / steps;
    }
    const int nops = KIND == 6 ? 64 * 3 : 64;
    printf("%-14s alone: %6.1f cyc/op | beside MFMAs: %6.1f cyc/op, MFMA wave %5.0f per 72 (+%.1f per partner op issued meanwhile)\n", name,
           res[0][1] / nops, res[1][1] / nops, res[1][0], (res[1][0] - 1160.0) / (nops * res[1][0] / res[1][1]));
}
int main() {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 64 * 512 * 4); (void)hipMalloc(&cyc, 64 * 8 * 8);
    run<0>("v_fma_f32", out, cyc); run<1>("v_exp_f32", out, cyc); run<2>("v_rcp_f32", out, cyc); 
    run<4>("s_nop", out, cyc); run<6>("exp,add,rcp", out, cyc);
    printf("-- VALU wave at s_setprio 3, MFMA wave at 0\n");
    run<0>("v_fma_f32", out, cyc, 1); run<1>("v_exp_f32", out, cyc, 1); run<6>("exp,add,rcp", out, cyc, 1);
    printf("-- roles swapped: older waves do the VALU work\n");
    run<0>("v_fma_f32", out, cyc, 0, 1); run<1>("v_exp_f32", out, cyc, 0, 1); run<6>("exp,add,rcp", out, cyc, 0, 1);
    printf("-- swapped, VALU wave at s_setprio 3\n");
    run<0>("v_fma_f32", out, cyc, 1, 1); run<6>("exp,add,rcp", out, cyc, 1, 1);
    return 0;
}
