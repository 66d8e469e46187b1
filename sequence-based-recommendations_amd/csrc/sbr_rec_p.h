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
__device__ __forceinline__ void st_s4(const void* ubase, unsigned boff, f32x4 v) {
    asm volatile("global_store_dwordx4 %0, %1, %2" :: "v"(boff), "v"(v), "s"(ubase) : "memory");
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
// rec_bwd_x6p<.., WT>: one dword per lane from global memory straight into LDS (LDS-DMA): lane i's value lands at
// lds_addr + 4 i.  Invisible to the compiler's vmcnt bookkeeping (its own waits for visible loads would also drain the
// write-through stores issued behind them); the kernel waits by hand (wait_vm) and reads the ring with ordinary LDS loads.
// lds_addr must be wave-uniform (it travels in M0).
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // M0 is "reserved": the clobber is what tells the compiler it is overwritten
__device__ __forceinline__ void lds_dma(unsigned lds_addr, const void* ubase, unsigned boff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" :: "s"(lds_addr), "v"(boff), "s"(ubase) : "memory", "m0");
}
// 16 bytes per lane (lane i's four dwords at lds_addr + 16 i) for the lanes of `mask` only: one piece moves up to 1 KiB
__device__ __forceinline__ void lds_dma_x4(unsigned lds_addr, const void* ubase, unsigned boff, unsigned long long mask) {
    asm volatile("s_mov_b32 m0, %0\n\ts_mov_b64 exec, %3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, -1"
                 :: "s"(lds_addr), "v"(boff), "s"(ubase), "s"(mask) : "memory", "m0");
}
#pragma clang diagnostic pop
template <int CNT>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(CNT) : "memory"); }
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
