// Helpers shared by the pipelined recurrent kernels (sbr_rec_p.hip: 128 units, two waves per SIMD; sbr_rec_q.hip:
// 32 / 64 units, one wave per SIMD).
#pragma once
#include <type_traits>
#include "sbr_cell.h"

#define X6P_SPIN_LIMIT (1 << 21)
#define X6P_NLOG2E (-1.4426950408889634f)

namespace {

// uniform base + 32-bit per-lane byte offset (+ immediate): one global_load/store with an SGPR base, no VALU
__device__ __forceinline__ float ldf(const void* base, unsigned boff, int imm = 0) {
    return *(const float*)((const char*)base + (size_t)boff + imm);
}
__device__ __forceinline__ int ldi(const void* base, unsigned boff, int imm = 0) {
    return *(const int*)((const char*)base + (size_t)boff + imm);
}
__device__ __forceinline__ void stf(void* base, unsigned boff, float v, int imm = 0) {
    *(float*)((char*)base + (size_t)boff + imm) = v;
}

// one lane adds 1 to an LDS counter: exec is all ones around every call site
__device__ __forceinline__ void lds_inc(unsigned addr, int one) {
    asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(addr), "v"(one) : "memory");
}
// store with a scalar base: the compiler's own form adds the per-step offset on the VALU
__device__ __forceinline__ void st_s(const void* ubase, unsigned boff, float v) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(boff), "v"(v), "s"(ubase) : "memory");
}
template <int IMM>
__device__ __forceinline__ void st_si(const void* ubase, unsigned boff, float v) {
    asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(boff), "v"(v), "s"(ubase), "n"(IMM) : "memory");
}
}  // namespace
