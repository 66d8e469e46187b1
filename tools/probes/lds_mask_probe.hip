// Probe: does a ds_read_b128 with 16 of 64 lanes active occupy the LDS pipe for less time than a full one?
// 8 waves per workgroup issue back-to-back reads; cycles per wave-instruction = LDS pipe time / 8.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc, int iters, int masked, int bcast) {
    __shared__ __attribute__((aligned(16))) char sh[32768];
    for (int i = threadIdx.x; i < 8192; i += 512) ((float*)sh)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // bcast: the h-plane pattern of the recurrent kernels (4 lanes share an address); else one address per lane
    unsigned addr = bcast ? (unsigned)(((lane & 15) >> 2) * 288 + (lane >> 4) * 16) : (unsigned)(lane * 16 + (threadIdx.x >> 6) * 1024);
    addr += (unsigned)(size_t)sh;
    f32x4 v0 = {0, 0, 0, 0}, v1 = v0, v2 = v0, v3 = v0;
    const unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (masked) asm volatile("s_mov_b32 exec_lo, 0x11111111\n\ts_mov_b32 exec_hi, 0x11111111" ::: "memory");
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1152\n\tds_read_b128 %2, %4 offset:2304\n\tds_read_b128 %3, %4 offset:64\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(addr) : "memory");
        if (masked) asm volatile("s_mov_b64 exec, -1" ::: "memory");
    }
    const unsigned long long t1 = clock64();
    if (lane == 0) cyc[threadIdx.x >> 6] = t1 - t0;
    out[threadIdx.x] = v0[0] + v1[1] + v2[2] + v3[3];
}
int main() {
    float* out; unsigned long long* cyc; (void)hipMalloc(&out, 4096); (void)hipMalloc(&cyc, 64);
    unsigned long long h[8];
    const int iters = 4000;
    for (int bcast = 0; bcast < 2; ++bcast)
        for (int masked = 0; masked < 2; ++masked) {
            probe<<<1, 512, 0, 0>>>(out, cyc, iters, masked, bcast);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            printf("%s addresses, %s exec: %.1f cycles per ds_read_b128 per wave (8 waves) = %.1f LDS-pipe cycles each\n",
                   bcast ? "shared (h-plane pattern)" : "distinct", masked ? "1-in-4" : "full", (double)h[0] / iters / 4, (double)h[0] / iters / 4 / 8);
        }
    return 0;
}
