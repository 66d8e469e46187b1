// Probe (round 6): what the operand split of the f32-class GEMMs costs per element on the VALU -- x -> (fp16(x), fp16((x - fp16(x)) * 2048)),
// eight elements packed into two 16-byte vectors, 8 waves per CU (two per SIMD) as in gemm_x6w_kernel -- alone, with the LDS write, and as
// bf16 round-to-nearest.  Prints shader cycles per wave-level element group (8 elements per lane).
// Build: hipcc --offload-arch=gfx950 -O3 -o split_probe split_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512, 1) k(const float* in, float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[512 * 48];
    f32x4 v0 = *(const f32x4*)(in + threadIdx.x * 8), v1 = *(const f32x4*)(in + threadIdx.x * 8 + 4);
    float accum = 0.f;
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        f16x8 h1, h2; bf16x8 b1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float x = (c < 4 ? v0[c & 3] : v1[c & 3]) * 1.0009765625f;
            asm("" : "+v"(x));
            if (MODE == 2) { b1[c] = (__bf16)x; }
            else { const _Float16 a1 = (_Float16)x; h1[c] = a1; h2[c] = (_Float16)((x - (float)a1) * 2048.0f); }
        }
        if (MODE == 0) { asm volatile("" :: "v"(h1), "v"(h2)); }
        else if (MODE == 1) { *(f16x8*)(lds + threadIdx.x * 48) = h1; *(f16x8*)(lds + threadIdx.x * 48 + 16) = h2; }
        else if (MODE == 2) { asm volatile("" :: "v"(b1)); }
        else if (MODE == 3) { *(f16x8*)(lds + threadIdx.x * 48) = h1; *(f16x8*)(lds + threadIdx.x * 48 + 16) = h2; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
        v0[0] += 1e-7f; asm volatile("" : "+v"(v0), "+v"(v1));
    }
    const long long t1 = clock64();
    if (MODE == 1 || MODE == 3) accum = *(float*)(lds + ((threadIdx.x * 52) & 16383));
    out[blockIdx.x * 512 + threadIdx.x] = accum + v0[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 512 * 8 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    hipMemset(in, 0, 512 * 8 * 4);
    const int iters = 2000;
    const char* names[4] = {"f16 split, 8 elements, registers only", "f16 split + 2 ds_write_b128", "bf16 RNE, 8 elements", "f16 split + 2 ds_write_b128 + wait"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) k<0><<<256, 512>>>(in, out, cyc, iters);
            if (m == 1) k<1><<<256, 512>>>(in, out, cyc, iters);
            if (m == 2) k<2><<<256, 512>>>(in, out, cyc, iters);
            if (m == 3) k<3><<<256, 512>>>(in, out, cyc, iters);
            hipDeviceSynchronize();
        }
        long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
        printf("%-44s %7.1f cycles per iteration of a wave (8 waves per CU) = %.1f per element\n", names[m], s / 256 / iters, s / 256 / iters / 8);
    }
    return 0;
}
