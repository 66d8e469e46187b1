// Probe: what VALU work costs when it sits in the gaps of the SAME wave's MFMA stream (one wave per SIMD), against the
// same work issued by the SIMD partner beside that stream (valu_beside_mfma_probe: 10 - 20 cycles per op).  The question
// behind it: should one wave own two unit tiles of the recurrent step and interleave the gate math of one with the MFMAs of
// the other, instead of two waves per SIMD taking turns?
//   part 1: 48 x v_mfma_f32_16x16x32_f16 per "step" on six accumulators, KPG filler ops after every MFMA -- independent
//           chains or ONE dependent chain, in the gate-math mix (fma, exp2, add, rcp).
//   part 2: the shape of a real step: groups of 12 MFMAs; beside group 2 and group 4 a dependent chain of NCH ops (the gate
//           math of one tile) whose first op reads the accumulators of the group before -- compiler-scheduled and pinned.
//   part 3: the same MFMAs with an s_nop N behind each (two waves per SIMD, waves 4-7 run a dependent VALU chain): does a
//           stream that gives the issue port away between its MFMAs let the partner's VALU work through?
// Build: hipcc --offload-arch=gfx950 -O3 -o own_valu_probe own_valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define MF(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_f16(A, B, C, 0, 0, 0)

__device__ __forceinline__ float chain_op(float v, int i) {
    switch (i & 3) {
        case 0: return fmaf(v, 1.0001f, 0.25f);
        case 1: return __builtin_amdgcn_exp2f(v);
        case 2: return v + 1.0f;
        default: return __builtin_amdgcn_rcpf(v);
    }
}

template <int KPG, int DEP>
__global__ void __launch_bounds__(256) probe_gaps(float* out, unsigned long long* cyc, int steps) {
    f16x8 a[6], b[3];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.001f * (threadIdx.x + i + e));
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.002f * (threadIdx.x + i * 3 + e));
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v[4];
    for (int i = 0; i < 4; ++i) v[i] = threadIdx.x * 1e-3f + i;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int k = 0; k < 48; ++k) {
            acc[k % 6] = MF(a[k % 6], b[k % 3], acc[k % 6]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KPG; ++i) {
                const int c = DEP ? 0 : (k * KPG + i) & 3;
                v[c] = chain_op(v[c], k * KPG + i);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    float r = v[0] + v[1] + v[2] + v[3];
    for (int i = 0; i < 6; ++i) r += acc[i][0] + acc[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

// part 2: four groups of 12 MFMAs (two tiles x two K halves); the chain of tile X (NCH dependent ops, first op reads the
// accumulators group 1 has just written) is placed beside group 2's MFMAs -- PINNED one op per MFMA (and the rest behind) or
// left to the compiler -- and tile Y's chain beside group 4.
template <int NCH, int PIN>
__global__ void __launch_bounds__(256) probe_step(float* out, unsigned long long* cyc, int steps) {
    f16x8 a[4], b[6];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.001f * (threadIdx.x + i + e));
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.002f * (threadIdx.x + i * 3 + e));
    f32x4 accX[6], accY[6];
    float hx = 0.1f, hy = 0.2f;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int s = 0; s < steps; ++s) {
        float vx, vy;
        // group 1: tile X, K half 2 (its K half 1 ran beside the previous chain)
#pragma unroll
        for (int k = 0; k < 12; ++k) accX[k % 6] = MF(a[k & 3], b[k % 6], (k < 6 ? f32x4{hx, 0, 0, 0} : accX[k % 6]));
        __builtin_amdgcn_sched_barrier(0);
        vx = accX[0][0] + accX[1][1] + accX[2][0] + accX[3][1] + accX[4][0] + accX[5][1];
        // group 2: tile Y, K half 2, beside chain X
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            accY[k % 6] = MF(a[k & 3], b[k % 6], (k < 6 ? f32x4{hy, 0, 0, 0} : accY[k % 6]));
            if (PIN) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < (NCH + 11) / 12; ++i) if (k * ((NCH + 11) / 12) + i < NCH) vx = chain_op(vx, k * ((NCH + 11) / 12) + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!PIN) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) vx = chain_op(vx, i);
        }
        hx = vx;
        __builtin_amdgcn_sched_barrier(0);
        vy = accY[0][0] + accY[1][1] + accY[2][0] + accY[3][1] + accY[4][0] + accY[5][1];
        // groups 3, 4: both tiles over K half 1 of the next step (24 MFMAs), beside chain Y
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            if (k < 12) accX[k % 6] = MF(a[k & 3], b[k % 6], accX[k % 6]); else accY[k % 6] = MF(a[k & 3], b[k % 6], accY[k % 6]);
            if (PIN) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < (NCH + 23) / 24; ++i) if (k * ((NCH + 23) / 24) + i < NCH) vy = chain_op(vy, k * ((NCH + 23) / 24) + i);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (!PIN) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) vy = chain_op(vy, i);
        }
        hy = vy;
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = hx + hy + accX[0][0] + accY[0][0];
}

// part 3: two waves per SIMD; waves 0-3 stream 24 MFMAs per step with s_nop NOP behind each, waves 4-7 run one dependent chain of
// 36 ops per step
template <int NOP>
__global__ void __launch_bounds__(512) probe_nop(float* out, unsigned long long* cyc, int steps) {
    const int wave = threadIdx.x >> 6;
    f16x8 a[6], b[3];
    for (int i = 0; i < 6; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.001f * (threadIdx.x + i + e));
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.002f * (threadIdx.x + i * 3 + e));
    f32x4 acc[6];
    for (int i = 0; i < 6; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float v = threadIdx.x * 1e-3f;
    __syncthreads();
    const unsigned long long t0 = clock64();
    if (wave < 4) {
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                acc[k % 6] = MF(a[k % 6], b[k % 3], acc[k % 6]);
                __builtin_amdgcn_sched_barrier(0);
                if (NOP >= 0) asm volatile("s_nop %0" :: "n"(NOP < 0 ? 0 : NOP));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int i = 0; i < 36; ++i) v = chain_op(v, i);
        }
    }
    const unsigned long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    float r = v;
    for (int i = 0; i < 6; ++i) r += acc[i][0];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}

static float* g_out; static unsigned long long* g_cyc;
template <typename F> static double run1(F launch, int wave, int steps) {
    unsigned long long h[8];
    launch();
    hipDeviceSynchronize();
    launch();
    hipDeviceSynchronize();
    hipMemcpy(h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
    return (double)h[wave] / steps;
}
int main() {
    (void)hipMalloc(&g_out, 64 * 512 * 4); (void)hipMalloc(&g_cyc, 64 * 8 * 8);
    const int steps = 2000;
    printf("== part 1: one wave per SIMD, 48 MFMAs per step, KPG filler ops behind every MFMA (cycles per step; 768 = bare MFMA issue)\n");
#define P1(K) printf("KPG %d: independent chains %7.1f | one dependent chain %7.1f\n", K, \
        run1([&] { probe_gaps<K, 0><<<4, 256>>>(g_out, g_cyc, steps); }, 0, steps), run1([&] { probe_gaps<K, 1><<<4, 256>>>(g_out, g_cyc, steps); }, 0, steps));
    P1(0) P1(1) P1(2) P1(3) P1(4)
    printf("== part 2: step shape (12 + 12 + 24 MFMAs, a dependent chain of NCH ops beside the second group and one beside the last two)\n");
#define P2(N) printf("NCH %2d: compiler-placed %7.1f | pinned into the gaps %7.1f\n", N, \
        run1([&] { probe_step<N, 0><<<4, 256>>>(g_out, g_cyc, steps); }, 0, steps), run1([&] { probe_step<N, 1><<<4, 256>>>(g_out, g_cyc, steps); }, 0, steps));
    P2(0) P2(12) P2(24) P2(36)
    printf("== part 3: two waves per SIMD: waves 0-3 24 MFMAs per step + s_nop N behind each, waves 4-7 a dependent chain of 36 ops per step\n");
#define P3(N) printf("s_nop %2d: MFMA wave %7.1f per step (384 = bare) | chain wave %7.1f per step = %5.1f per op\n", N, \
        run1([&] { probe_nop<N><<<4, 512>>>(g_out, g_cyc, steps); }, 0, steps), run1([&] { probe_nop<N><<<4, 512>>>(g_out, g_cyc, steps); }, 4, steps), \
        run1([&] { probe_nop<N><<<4, 512>>>(g_out, g_cyc, steps); }, 4, steps) / 36.0);
    P3(-1) P3(0) P3(1) P3(3) P3(5) P3(7) P3(9) P3(11)
    return 0;
}
