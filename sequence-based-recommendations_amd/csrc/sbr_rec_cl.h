// Device helpers shared by the cluster recurrent kernels (sbr_rec_cl.hip: 8-row tiles; sbr_rec_c16.hip: 16-row tiles).
#pragma once
#include "sbr_cell.h"
#include <type_traits>
#include <cstdlib>

#define CL_SENT 0xFFFFFFFFu
#define CL_SPIN_LIMIT 400000

typedef unsigned long long u64;

__device__ __forceinline__ u64 cl_load(const float* p) {
    return __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// fast = every member of the cluster runs on the same XCC: one L2 is the coherence point, an ordinary store
// (L1 is write-through) is visible to the members' L1-bypassing loads as soon as it reaches that L2.
// Otherwise the store must write through to memory (sc1): measured ~3000-6000 cycles more per step.
__device__ __forceinline__ void cl_store4(float* p, const f32x4 v, bool fast) {
    if (fast) { *(f32x4*)p = v; return; }
    union { float f[2]; u64 u; } a, b;
    a.f[0] = v[0]; a.f[1] = v[1]; b.f[0] = v[2]; b.f[1] = v[3];
    __hip_atomic_store((u64*)p, a.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((u64*)(p + 2), b.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cl_has_sentinel(u64 v) {
    return (unsigned)v == CL_SENT || (unsigned)(v >> 32) == CL_SENT;
}

// cluster / member of this workgroup; false = padding workgroup (no tile)
__device__ __forceinline__ bool cl_ids(const RecArgs& a, int C, int ntiles, int& tile, int& m) {
    const int bid = blockIdx.x;
    if (a.cl_linear) { tile = bid / C; m = bid % C; }             // (experiment) members on consecutive ids = different XCDs
    else { const int x = bid & 7, y = bid >> 3; m = y % C; tile = (y / C) * 8 + x; }
    return tile < ntiles;
}

// Start-of-launch handshake: every member publishes the XCC it runs on (HW_REG_XCC_ID) and reads the others'.
// Returns true when the whole cluster shares one XCC (the placement in the header makes that the normal case;
// nothing breaks when it does not hold -- the kernels then publish with write-through stores).
__device__ __forceinline__ bool cl_same_xcc(const RecArgs& a, int C, int tile, int mem, int* lds_flag, bool& dead) {
    const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));           // hwreg(HW_REG_XCC_ID, 0, 4)
    int* slots = a.clx + (size_t)tile * C;
    if (threadIdx.x == 0) {
        *lds_flag = 1;
        __hip_atomic_store(&slots[mem], (a.epoch << 4) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        int v = 0, tries = 0;
        while (true) {
            v = __hip_atomic_load(&slots[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 4) == a.epoch) break;
            if (++tries > CL_SPIN_LIMIT) { dead = true; atomicOr(a.fault, 1); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if ((v & 15) != xcc || (v >> 4) != a.epoch) *lds_flag = 0;
    }
    __syncthreads();
    const bool same = *lds_flag != 0;
    __syncthreads();
    return same;
}

// Polls NP 16-byte pieces per thread; piece p covers floats [4*c4, 4*c4+3] of tile row r, where
// p = tid + i*256, r = p / (W/4), c4 = p % (W/4).  src(r, col) returns the address.
// fast (whole cluster on one XCC): 16-byte sc1 loads (bypass the CU's L1, served by the shared L2).  hipcc lowers
// agent-scope atomic loads to sc1 only up to 8 bytes (0.54-0.70x the 16-byte rate), hence the inline asm: these
// loads are invisible to the compiler's waitcnt insertion, so the wait is explicit and the values are re-defined
// after it to pin their uses behind it.  (Tried and rejected: sc0 loads hit the stale L1 line; an L1 invalidate
// per poll, buffer_inv sc1, costs ~15000 cycles.)
template <int NP, typename SRC>
__device__ __forceinline__ int cl_fetch(f32x4 (&v)[NP], SRC src, int W, bool fast, bool& dead, int* fault) {
    const int tid = threadIdx.x;
    int tries = 0;
    while (true) {
        if (fast) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int p = tid + i * 256, r = p / (W >> 2), c4 = p % (W >> 2);
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(src(r, 4 * c4)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(v[i]));
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int p = tid + i * 256, r = p / (W >> 2), c4 = p % (W >> 2);
                const float* q = src(r, 4 * c4);
                union { u64 u[2]; f32x4 f; } x;
                x.u[0] = cl_load(q); x.u[1] = cl_load(q + 2);
                v[i] = x.f;
            }
        }
        unsigned mx = 0u;                                // the sentinel is the largest 32-bit pattern: one v_max3_u32 per two words
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = max(mx, __float_as_uint(v[i][e]));
        const bool ok = mx != CL_SENT;
        if (ok || dead) break;
        if (++tries > CL_SPIN_LIMIT) { dead = true; atomicOr(fault, 1); break; }   // bounded: never hang the GPU
        __builtin_amdgcn_s_sleep(1);
    }
    return tries;
}

// "f16x3" (sbr_rec_p.hip, split2_f16): an operand as a1 + a2 / 2048 in two fp16 planes, a product in three MFMAs.  Forward: h is
// in [-1, 1] unless the layer rectifies; backward: dhi has passed the reference's gradient clip (<= 100), scaled by 2^9.
typedef _Float16 f16x8c __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4c __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2c __attribute__((ext_vector_type(2)));
constexpr float CL_F16_LO = 2048.0f, CL_F16_DSCALE = 512.0f;
__device__ __forceinline__ f32x4 cl_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) { return MFMA_BF16(a, b, c); }
__device__ __forceinline__ f32x4 cl_mfma(const f16x8c& a, const f16x8c& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void cl_split2(float v, _Float16& a1, _Float16& a2) {
    asm("" : "+v"(v));                                   // one rounding to fp16 for both uses (see split2_f16)
    a1 = (_Float16)v;
    a2 = (_Float16)((v - (float)a1) * CL_F16_LO);
}
// splits the fetched pieces into three bf16 planes (F16: two fp16 planes of scale * v) [plane][R][ROWB bytes]
template <int NP, bool F16 = false>
__device__ __forceinline__ void cl_publish(const f32x4 (&v)[NP], char* planes, int W, int ROWB, int PLANEB, float scale = 1.0f) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = tid + i * 256, r = p / (W >> 2), c4 = p % (W >> 2);
        if constexpr (F16) {
            f16x4c h1, h2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { _Float16 a1, a2; cl_split2(v[i][e] * scale, a1, a2); h1[e] = a1; h2[e] = a2; }
            char* base = planes + r * ROWB + c4 * 8;
            *(f16x4c*)(base) = h1;
            *(f16x4c*)(base + PLANEB) = h2;
            continue;
        }
        bf16x4 p1, p2, p3;
        split3x4(v[i], p1, p2, p3);
        char* base = planes + r * ROWB + c4 * 8;
        *(bf16x4*)(base) = p1;
        *(bf16x4*)(base + PLANEB) = p2;
        *(bf16x4*)(base + 2 * PLANEB) = p3;
    }
}


typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8 bytes to an exchange array (see cl_store4 for `fast`)
__device__ __forceinline__ void cl_store2(float* p, const f32x2 v, bool fast) {
    if (fast) { *(f32x2*)p = v; return; }
    union { float f[2]; u64 u; } a; a.f[0] = v[0]; a.f[1] = v[1];
    __hip_atomic_store((u64*)p, a.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// SBR_FLAG_PROFILE_REC: cycles per phase of a step (tools/cl_prof.py); the kernels declare pc[], p_t, prof
#define CL_TICK(i) do { if (prof) { const u64 n_ = clock64(); pc[i] += n_ - p_t; p_t = n_; } } while (0)

__device__ __forceinline__ void cl_store1(unsigned* p, unsigned v, bool fast) {
    if (fast) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned cl_f16_bits(_Float16 x) { union { _Float16 h; unsigned short s; } u; u.h = x; return u.s; }
// Two fp16 planes of v, one 32-bit word per lane: even lanes hold plane 0 of units (j, j + 1), odd lanes plane 1 of (j - 1, j)
__device__ __forceinline__ unsigned cl_pair_word(float v, int j) {
    _Float16 a1, a2;
    cl_split2(v, a1, a2);
    const unsigned s1 = cl_f16_bits(a1), s2 = cl_f16_bits(a2);
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)((j & 1) ? s1 : s2), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
    return (j & 1) ? (recv | (s2 << 16)) : (s1 | (recv << 16));
}

static inline int cl_grid(const RecArgs& a, int C, int R) {
    const int ntiles = a.Bp / R;
    return a.cl_linear ? ntiles * C : (ntiles + 7) / 8 * 8 * C;
}
#define CL_LAUNCH(KERNEL, C, R, LDS) do { \
        SBR_DYN_LDS(KERNEL, (LDS)); \
        KERNEL<<<cl_grid(a, C, R), 256, LDS, s>>>(a); } while (0)
// sbr_rec_c16.hip
hipError_t launch_rec_forward_c16(hipStream_t s, const RecArgs& a);
hipError_t launch_rec_backward_c16(hipStream_t s, const RecArgs& a);
hipError_t sbr_rec_c16_fill(hipStream_t s, const RecArgs& a);
