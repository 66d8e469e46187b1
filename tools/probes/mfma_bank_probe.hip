// Probe: does the matrix pipe's issue rate depend on how the three 4-register operand tuples of
// v_mfma_f32_16x16x32_bf16 are aligned in the VGPR file?  (Two builds of rec_fwd_x6p with identical instruction streams
// ran 177 vs 198 us; the only difference was A/B tuples at register 4n+2 vs 4n with the accumulators at 4n.)
// 96 independent-accumulator MFMAs per iteration, physical registers named in the asm.   Build: tools/probes/build.sh
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP6(X) X X X X X X
#define BODY(A, B) \
    "v_mfma_f32_16x16x32_bf16 v[0:3], " A ", " B ", v[0:3]\n" \
    "v_mfma_f32_16x16x32_bf16 v[4:7], " A ", " B ", v[4:7]\n" \
    "v_mfma_f32_16x16x32_bf16 v[8:11], " A ", " B ", v[8:11]\n" \
    "v_mfma_f32_16x16x32_bf16 v[12:15], " A ", " B ", v[12:15]\n"
#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15", \
             "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"

template <int MODE>
__global__ void probe(unsigned long long* out, int iters) {
    unsigned long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) asm volatile(REP6(BODY("v[16:19]", "v[20:23]")) ::: CLOB);        // A, B, C all at 4n
        if (MODE == 1) asm volatile(REP6(BODY("v[18:21]", "v[22:25]")) ::: CLOB);        // A, B at 4n+2
        if (MODE == 2) asm volatile(REP6(BODY("v[18:21]", "v[20:23]")) ::: CLOB);        // A at 4n+2, B at 4n
        if (MODE == 3) asm volatile(REP6(BODY("v[16:19]", "v[22:25]")) ::: CLOB);        // A at 4n, B at 4n+2
        if (MODE == 4) asm volatile(REP6(BODY("v[20:23]", "v[20:23]")) ::: CLOB);        // A and B the same tuple at 4n
    }
    unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    const char* names[5] = {"A 4n   B 4n   C 4n", "A 4n+2 B 4n+2 C 4n", "A 4n+2 B 4n   C 4n", "A 4n   B 4n+2 C 4n", "A = B 4n      C 4n"};
    const int iters = 2000;
    for (int m = 0; m < 5; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            if (m == 0) probe<0><<<1, 64>>>(d, iters); if (m == 1) probe<1><<<1, 64>>>(d, iters); if (m == 2) probe<2><<<1, 64>>>(d, iters);
            if (m == 3) probe<3><<<1, 64>>>(d, iters); if (m == 4) probe<4><<<1, 64>>>(d, iters);
            hipDeviceSynchronize();
        }
        unsigned long long h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("%s : %.2f cycles per MFMA (one wave, 24 MFMAs per iteration)\n", names[m], (double)h / iters / 24.0);
    }
    return 0;
}
