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
// WT: write-through (sc1) -- the value reaches memory that every XCD sees, for consumers that read it while this kernel
// is still running (MI355X_MICROARCH.md, inter-workgroup visibility)
template <int IMM, bool WT = false>
__device__ __forceinline__ void st_si(const void* ubase, unsigned boff, float v) {
    if constexpr (WT) asm volatile("global_store_dword %0, %1, %2 offset:%3 sc1" :: "v"(boff), "v"(v), "s"(ubase), "n"(IMM) : "memory");
    else asm volatile("global_store_dword %0, %1, %2 offset:%3" :: "v"(boff), "v"(v), "s"(ubase), "n"(IMM) : "memory");
}
__device__ __forceinline__ void st_wt(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
// rec_bwd_x6p<.., WT>: loads with a scalar base that the compiler's vmcnt bookkeeping does not see (its own wait for the last
// of five visible loads would be vmcnt(0) and drain the write-through stores issued behind them), and the wait that makes
// their results valid.  Between the two the destination registers must not be read: the kernel copies nothing out of
// them there, and tests/test_isa_lint.py checks the generated code for it.
__device__ __forceinline__ void ld_s(float& dst, const void* ubase, unsigned boff) {
    asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(boff), "s"(ubase) : "memory");
}
template <int CNT>
__device__ __forceinline__ void wait_vm5(float& a, float& b, float& c, float& d, float& e) {
    asm volatile("s_waitcnt vmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(CNT) : "memory");
}
template <int CNT>
__device__ __forceinline__ void wait_vm1(float& a) {
    asm volatile("s_waitcnt vmcnt(%1)" : "+v"(a) : "n"(CNT) : "memory");
}
// a copy the compiler can neither delay nor fold (ordered with the loads above)
__device__ __forceinline__ float copy_now(float v) {
    float r;
    asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// one lane publishes a progress word, write-through
__device__ __forceinline__ void publish_word(int* slot, int word) {
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_dword %0, %1, off sc1\n\ts_mov_b64 exec, -1" :: "v"(slot), "v"(word) : "memory");
}
// ... once `dep` has been computed (a value that depends on the loads the publication speaks for)
__device__ __forceinline__ void publish_word_after(int* slot, int word, float dep) {
    asm volatile("s_mov_b64 exec, 1\n\tglobal_store_dword %0, %1, off sc1\n\ts_mov_b64 exec, -1" :: "v"(slot), "v"(word), "v"(dep) : "memory");
}
// ... after everything this wave has stored before is complete
__device__ __forceinline__ void publish_progress(int* slot, int word) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    publish_word(slot, word);
}
// Consumers of the running chain poll a progress word; how long to sleep between two polls when `gap` time steps (~1 us
// each) are still missing -- thousands of pollers of one word must not load the fabric the chain's own loads go through.
// pollers give up after this many ticks of the 100 MHz wall clock (1.5 s) and raise the fault flag
#define SBR_POLL_TICKS 150000000ull
__device__ __forceinline__ void poll_sleep(int gap) {
    if (gap < 4) { __builtin_amdgcn_s_sleep(8); return; }
    const int n = min(gap >> 2, 16);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(32);
}
}  // namespace
