// float32 GEMM on the bf16 matrix pipe ("bf16x6", see sbr_rec.hip): every f32 operand element is split exactly into
// three bf16 planes when its tile is written to LDS, and the six products of order >= 2^-18 are issued on
// v_mfma_f32_16x16x32_bf16 -- f32-rounding-class results at 6 x 16 cycles per 16x16x32 block instead of 8 x 32 on
// v_mfma_f32_16x16x4_f32 (2.67x less matrix-pipe time; the split costs ~7 VALU per loaded element, amortised over a
// 128-wide tile).  Serves the dense pieces of the hot path whose shapes fill a 128x128 tile: output projection
// logits = h.W_out and its backward pair (rnn_one_hot.py:65, K7/K9), the weight gradients dW_hid / dW_in after the
// BPTT chain, the layer >= 2 input projections.
//
//   C[m][n] = sum_k A(m,k) * B(k,n) (+ bias[n]),  A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]
//
// Workgroup = 256 threads, 128x128 output tile, BK = 32 (one MFMA K).  Threads 0..127 load the A tile, 128..255 the
// B tile: 4 rows x 8 k each, vectorised along whichever dimension has unit stride (16/8/4-byte loads by alignment),
// next tile prefetched into registers while the MFMAs of the current one run.  LDS holds each operand as three bf16
// planes [128 rows][32 k] with an 80-byte row stride (conflict-free ds_read_b128 / ds_write_b128).  Each wave owns a
// 64x64 sub-tile = 4x4 MFMA tiles.  Split-K over grid.z writes partial slabs (summed by gemm_splitk_reduce).
//
// Measured on C4's weight-gradient shape (M=256 N=1024 K=51200, 189 us = 142 TFLOP/s f32-equivalent): with 1/6 of the
// MFMAs the kernel still takes 123 us, i.e. the matrix pipe runs at full rate when it runs and ~65 % of the time is the
// serial split / LDS-write / barrier / LDS-read part of each k-step (two workgroups per CU overlap it only partly).
// Tried and rejected: a second stage of global prefetch (no change: latency is already covered); double-buffered
// LDS with the split interleaved into the MFMA stream at one workgroup per CU (311 us: 376 registers push operands
// into AGPRs and a single wave per SIMD cannot keep the pipe fed -- the same lesson as rec_fwd_x6s NT=2).
#include "sbr_rec_p.h"
#include <type_traits>


struct GemmX6Args {
    const float* A; long sam, sak;
    const float* B; long sbk, sbn;
    float* C; long ldc; size_t slab_stride;
    int M, N, K, kchunk;
    const float* bias;
    const float* B2; long sbk2; int n_split;   // columns n >= n_split of B come from B2[k*sbk2 + (n - n_split)] (GRU weight gradients)
    float sa, sb, so;                          // NP == 2 (fp16 x3): power-of-two scales of the operands on the way in, 1 / (sa sb) on the way out
    SbrPoll poll;                              // words != NULL: consumer of a running BPTT chain (sbr_common.h)
};

// Overlapped step tail.  A workgroup polls the monitor's word `done` (tail_monitor_kernel, sbr_misc.hip; its copy of it: SbrPoll)
// from one lane until the first time step of its next slab is complete.  Spins are bounded (fault bit 3).  t_need: the slab's
// first time step.  seen: the last value of `done` this workgroup has read (0xfff before the first) -- a persistent
// workgroup whose next slab is already released does not pay another round trip to the memory side.
// The chain-written operand is then read with agent-coherent (sc1) loads -- x6_issue_sc1 -- and NOT behind an acquire fence:
// on gfx950 that fence is `buffer_inv sc1`, which drops every non-coherent line of the XCD's L2 and takes ~15 000 cycles
// (sbr_rec_cl.hip); the chain stored write-through, and sc1 loads are served by the memory side.
__device__ __forceinline__ void x6_poll_wait(const SbrPoll& pl, int t_need, int tid, int& seen) {
    __shared__ int s_seen;
    if (seen <= t_need) return;                              // (uniform)
    const int tag = pl.epoch;
    if (tid == 0) {
        const unsigned long long t0 = wall_clock64();
        const int* mine = pl.done + ((blockIdx.z * 3 + blockIdx.x + blockIdx.y) & (SBR_DONE_COPIES - 1)) * SBR_DONE_STRIDE;
        for (;;) {
            const int v = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 12) == tag && (v & 0xfff) <= t_need) { s_seen = v & 0xfff; break; }
            if (wall_clock64() - t0 > SBR_POLL_TICKS) { atomicOr(pl.fault, 8); s_seen = 0; break; }
            poll_sleep((v >> 12) == tag ? (v & 0xfff) - t_need : 64);
        }
    }
    __syncthreads();
    seen = s_seen;
    __syncthreads();
}
// 4 rows x 8 k as in x6_load (interior tiles only), with loads that every XCD's stores reach (sc1: served by the memory side,
// not by a line this XCD's L2 may still hold from the previous training step).  The loads are invisible to the compiler's
// waitcnt insertion: the registers stay where the loads put them until x6_sc1_landed() has waited, and only then become v.
template <bool RFAST>
__device__ __forceinline__ void x6_issue_sc1(const float* __restrict__ p, long stride, f32x4 (&t)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float* r = RFAST ? p + (long)i * stride : p + (long)(i >> 1) * stride + (i & 1) * 4;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[i]) : "v"(r) : "memory");
    }
}
// YOUNGER: the wave has issued one more stage of 8 loads behind this one (the vector-memory counter retires in order)
template <bool RFAST, bool YOUNGER = false>
__device__ __forceinline__ void x6_sc1_landed(f32x4 (&t)[8], float (&v)[4][8]) {
    if constexpr (YOUNGER) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(t[i]));
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (RFAST) v[e][i] = t[i][e];
            else v[i >> 1][(i & 1) * 4 + e] = t[i][e];
        }
}
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 x6_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) { return MFMA_BF16(a, b, c); }
__device__ __forceinline__ f32x4 x6_mfma(const f16x8g& a, const f16x8g& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// 4 rows x 8 k of one operand tile into v[row][k], from p = &operand(r0, k0).  RFAST: unit stride runs along the rows
// (srow == 1), else along k (sk == 1).  VEC: floats per load instruction (4; the scalar form is kept for the ragged edge path only).  nr / nk: rows
// and k's inside the matrix (4 / 8 for interior tiles: the branch-free path).
template <int VEC, bool RFAST>
__device__ __forceinline__ void x6_load(const float* __restrict__ p, long stride, int nr, int nk, float (&v)[4][8]) {
    if (nr == 4 && nk == 8) {
        if (!RFAST) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* r = p + (long)i * stride;                      // stride = srow
                if (VEC == 4) {
                    const f32x4 t0 = *(const f32x4*)r, t1 = *(const f32x4*)(r + 4);
                    v[i][0] = t0[0]; v[i][1] = t0[1]; v[i][2] = t0[2]; v[i][3] = t0[3];
                    v[i][4] = t1[0]; v[i][5] = t1[1]; v[i][6] = t1[2]; v[i][7] = t1[3];
                } else {
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk) v[i][kk] = r[kk];
                }
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float* r = p + (long)kk * stride;                     // stride = sk
                if (VEC == 4) { const f32x4 t = *(const f32x4*)r; v[0][kk] = t[0]; v[1][kk] = t[1]; v[2][kk] = t[2]; v[3][kk] = t[3]; }
                else { v[0][kk] = r[0]; v[1][kk] = r[1]; v[2][kk] = r[2]; v[3][kk] = r[3]; }
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            v[i][kk] = (i < nr && kk < nk) ? (RFAST ? p[(long)kk * stride + i] : p[(long)i * stride + kk]) : 0.0f;
}

// TW = MFMA tiles per wave and dimension, KH = 32-wide k blocks per step: <4, 1> is the 128x128x32 workgroup tile
// described above; <2, 2> is a 64x64x64 tile for problems that would put fewer than ~128 of the large tiles on the chip
// (C2's logits 256 x 3706 x 128 and dh 256 x 128 x 3706: 58 / 56 workgroups of the large tile, ~230 of the small one).
// Either way 128 threads load one operand tile, 4 rows x 8 k each.
// NP = bf16 planes per operand: 3 = the exact split above (six MFMA terms); 1 = plain bf16 operands (round to nearest
// even), ONE v_mfma_f32_16x16x32_bf16 per block with f32 accumulation -- the "bf16 MFMA output projection" of the 1 M-item
// configuration (BASELINE.json configs[4]; SBR_FLAG_BF16_PROJECTION): logits to ~3e-3 of their spread instead of f32
// rounding, a sixth of the matrix-pipe time and a third of the LDS traffic.
// PL: consumer of a running BPTT chain (g.poll): both operands through x6_issue_sc1.
// (PL: one workgroup per CU is what the overlapped tail runs anyway -- 192 of them beside the chain's 64 -- so the polling form may
// use the registers of two: a second stage of operand loads in flight, X6_PL_STAGES)
#ifndef X6_PL_STAGES
#define X6_PL_STAGES 2
#endif
template <int VA, bool RA, int VB, bool RB, int TW, int KH, int NP, bool PL = false>
__global__ void __launch_bounds__(256, PL ? 1 : 2) gemm_x6_kernel(GemmX6Args g) {
    constexpr int TM = 32 * TW, TK = 32 * KH, ROW = 64 * KH + 16, PLANE = TM * ROW, KC = 4 * KH;
    using OPV = std::conditional_t<NP == 2, f16x8g, bf16x8>;
    __shared__ __attribute__((aligned(16))) char sA[NP * PLANE];
    __shared__ __attribute__((aligned(16))) char sB[NP * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int j = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TM;
    int zz = blockIdx.z;
    // loader role: A tile (waves 0,1) or B tile (waves 2,3); rows rg*4..+3, k-chunk kc*8..+7
    const bool ldB = tid >= 128;
    const int lt = tid & 127, rg = lt / KC, kc = lt % KC;
    const int r0 = (ldB ? n0 : m0) + rg * 4;
    const int nr = max(0, min(4, (ldB ? g.N : g.M) - r0));
    long srow = ldB ? g.sbn : g.sam, sk = ldB ? g.sbk : g.sak;
    const float* base = (ldB ? g.B : g.A) + (long)r0 * srow;
    if (ldB && g.B2 && r0 >= g.n_split) { sk = g.sbk2; base = g.B2 + (long)(r0 - g.n_split) * srow; }
    const float* src = base;                               // advanced by TK k per step
    const long kstep = TK * sk;
    char* sdst = (ldB ? sB : sA) + (rg * 4) * ROW + kc * 16;

    float v[4][8];
    f32x4 tq[8], tq2[8];      // (dead without PL / with one stage)
    int kbeg = 0, kend = 0;
    // PL: one stage of 8 sc1 loads for the k step at `k0` into t (a k chunk beyond the slab: zeros, no loads -- a wave whose
    // lanes hold different chunks still executes the 8 load instructions, so the in-order count per stage is the same)
    auto issue = [&](f32x4 (&t)[8], int k0) {
        const int nk = max(0, min(8, kend - (k0 + kc * 8)));
        if (nk < 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] = f32x4{0, 0, 0, 0};
        }
        else if (ldB) x6_issue_sc1<RB>(src, RB ? sk : srow, t);
        else x6_issue_sc1<RA>(src, RA ? sk : srow, t);
        src += kstep;
    };
    auto load = [&](int k0) {
        const int nk = max(0, min(8, kend - (k0 + kc * 8)));
        if (ldB) x6_load<VB, RB>(src, RB ? sk : srow, nr, nk, v);
        else x6_load<VA, RA>(src, RA ? sk : srow, nr, nk, v);
        src += kstep;
    };
    const f32x4 z = f32x4{0, 0, 0, 0};
    f32x4 acc[TW][TW], acl[NP == 2 ? TW : 1][NP == 2 ? TW : 1];      // acl: the low-order products of the fp16 form
#pragma unroll
    for (int a = 0; a < TW; ++a)
#pragma unroll
        for (int b = 0; b < TW; ++b) { acc[a][b] = z; if constexpr (NP == 2) acl[a][b] = z; }
    const float opscale = ldB ? g.sb : g.sa;

    // one k step: `t` = the stage that holds its operands (PL), younger = another stage was issued behind it
    auto k_step = [&](f32x4 (&t)[8], int k0, bool younger) {
        if constexpr (PL) {
            if (younger) { if (ldB) x6_sc1_landed<RB, true>(t, v); else x6_sc1_landed<RA, true>(t, v); }
            else { if (ldB) x6_sc1_landed<RB, false>(t, v); else x6_sc1_landed<RA, false>(t, v); }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            bf16x8 p1, p2, p3;
            if constexpr (NP == 1) {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) p1[kk] = (__bf16)v[i][kk];
                *(bf16x8*)(sdst + i * ROW) = p1;
            } else if constexpr (NP == 2) {
                f16x8g h1, h2;
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    float x = v[i][kk] * opscale;
                    asm("" : "+v"(x));                     // one rounding to fp16 for both uses (sbr_rec_p.hip split2_f16)
                    const _Float16 a1 = (_Float16)x;
                    h1[kk] = a1; h2[kk] = (_Float16)((x - (float)a1) * 2048.0f);
                }
                *(f16x8g*)(sdst + i * ROW) = h1;
                *(f16x8g*)(sdst + i * ROW + PLANE) = h2;
            } else {
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) { __bf16 a, b, c; split3(v[i][kk], a, b, c); p1[kk] = a; p2[kk] = b; p3[kk] = c; }
                *(bf16x8*)(sdst + i * ROW) = p1;
                *(bf16x8*)(sdst + i * ROW + PLANE) = p2;
                *(bf16x8*)(sdst + i * ROW + 2 * PLANE) = p3;
            }
        }
        __syncthreads();
        if constexpr (PL && X6_PL_STAGES == 2) { if (k0 + 2 * TK < kend) issue(t, k0 + 2 * TK); }      // this stage's registers are free again
        else if constexpr (PL) { if (k0 + TK < kend) issue(t, k0 + TK); }
        else { if (k0 + TK < kend) load(k0 + TK); }        // in flight while this tile's MFMAs run
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            OPV a[NP][TW], b[NP][TW];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int t = 0; t < TW; ++t) {
                    a[p][t] = *(const OPV*)(sA + p * PLANE + (wm * 16 * TW + t * 16 + j) * ROW + kh * 64 + q * 16);
                    b[p][t] = *(const OPV*)(sB + p * PLANE + (wn * 16 * TW + t * 16 + j) * ROW + kh * 64 + q * 16);
                }
            // smallest terms first: a1b3, a3b1, a2b2, a1b2, a2b1, a1b1; TW*TW independent accumulators per term
#define X6_TERM(PA, PB) _Pragma("unroll") for (int mi = 0; mi < TW; ++mi) _Pragma("unroll") for (int ni = 0; ni < TW; ++ni) \
                acc[mi][ni] = x6_mfma(b[PB][ni], a[PA][mi], acc[mi][ni]);
#define X6_TERL(PA, PB) _Pragma("unroll") for (int mi = 0; mi < TW; ++mi) _Pragma("unroll") for (int ni = 0; ni < TW; ++ni) \
                acl[mi][ni] = x6_mfma(b[PB][ni], a[PA][mi], acl[mi][ni]);
            if constexpr (NP == 1) { X6_TERM(0, 0) }
            else if constexpr (NP == 2) { X6_TERL(0, 1) X6_TERL(1, 0) X6_TERM(0, 0) }
            else { X6_TERM(0, NP - 1) X6_TERM(NP - 1, 0) X6_TERM(1, 1) X6_TERM(0, 1) X6_TERM(1, 0) X6_TERM(0, 0) }
#undef X6_TERM
#undef X6_TERL
        }
        __syncthreads();
    };
    auto run_slab = [&]() {                               // acc += A[:, kbeg .. kend) . B[kbeg .. kend, :]
        src = base + (long)(kbeg + kc * 8) * sk;
        if constexpr (PL && X6_PL_STAGES == 2) {
            // two k steps of operands in flight: the k loop of a polling group was one exposed round trip to the memory side per
            // step (sc1 loads, ~3 us against ~1 us of split + MFMAs): 51 200 rows took the launch ~115 us of pure work, 1.2x
            // the rate at which the chain releases them -- a late start was never caught up (profiles/round5_b_c2_timeline.txt)
            if (kbeg < kend) issue(tq, kbeg);
            if (kbeg + TK < kend) issue(tq2, kbeg + TK);
            for (int k0 = kbeg; k0 < kend; k0 += 2 * TK) {
                k_step(tq, k0, k0 + TK < kend);
                if (k0 + TK < kend) k_step(tq2, k0 + TK, k0 + 2 * TK < kend);
            }
        } else {
            if constexpr (PL) { if (kbeg < kend) issue(tq, kbeg); }
            else { if (kbeg < kend) load(kbeg); }
            for (int k0 = kbeg; k0 < kend; k0 += TK) k_step(tq, k0, false);
        }
    };
    if constexpr (PL) {
        // Persistent groups: gridDim.z groups of workgroups (one per N / M tile each) share the nz slabs round-robin in the
        // order the chain releases them (slab nz-1 first); a workgroup keeps its accumulators across its slabs and stores ONE
        // partial at the end (slab `group` of the workspace).  So the launch occupies a FIXED number of workgroups, all
        // resident from the start -- with one workgroup per slab the early, long slabs held the CUs' registers and the
        // scatter-add beside this launch got its waves only when the chain had ended (profiles/round3_u_trace.txt).
        const int nz = g.poll.n_slabs, groups = (int)gridDim.z, group = (int)blockIdx.z;
        int seen = 0xfff;
        for (int sl = nz - 1 - group; sl >= 0; sl -= groups) {
            kbeg = g.poll.slab_lo[sl]; kend = min(g.K, g.poll.slab_lo[sl + 1]);
            x6_poll_wait(g.poll, kbeg / g.poll.rows_per_step, tid, seen);
            unsigned long long* tr = g.poll.trace && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 ? g.poll.trace + 16 + 2 * sl : nullptr;
            if (tr) tr[0] = wall_clock64();
            run_slab();
            if (tr) tr[1] = wall_clock64();
        }
        zz = group;
    } else {
        kbeg = blockIdx.z * g.kchunk; kend = min(g.K, kbeg + g.kchunk);
        run_slab();
    }
    // The products above are mfma(B rows, A rows): the accumulators hold the TRANSPOSED 16x16 tiles, i.e. lane (j, q) has row
    // m = j and the four consecutive columns n = 4q .. 4q+3 of each tile -- one 16-byte store per tile and lane, 64 contiguous
    // bytes per row and quarter wave.  (Round 3: the untransposed form wrote 64 guarded dwords per lane, ~8 us of a workgroup's
    // life whatever its K -- profiles/round3_s_trace.txt.)
    float* out = g.C + (size_t)zz * g.slab_stride;
    const bool inner = (g.ldc & 3) == 0 && ((uintptr_t)out & 15) == 0 && m0 + TM <= g.M && n0 + TM <= g.N;   // uniform
    auto tile_out = [&](int mi, int ni) -> f32x4 {
        f32x4 val = acc[mi][ni];
        if constexpr (NP == 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) val[r] = fmaf(acl[mi][ni][r], 1.0f / 2048.0f, val[r]) * g.so;
        }
        return val;
    };
    if (inner && !g.bias) {                      // split-K slabs, whole tiles: nothing but the stores
#pragma unroll
        for (int mi = 0; mi < TW; ++mi)
#pragma unroll
            for (int ni = 0; ni < TW; ++ni)
                *(f32x4*)(out + (long)(m0 + wm * 16 * TW + mi * 16 + j) * g.ldc + n0 + wn * 16 * TW + ni * 16 + 4 * q) = tile_out(mi, ni);
    } else {
#pragma unroll
        for (int mi = 0; mi < TW; ++mi) {
            const int m = m0 + wm * 16 * TW + mi * 16 + j;
#pragma unroll
            for (int ni = 0; ni < TW; ++ni) {
                const int n = n0 + wn * 16 * TW + ni * 16 + 4 * q;
                f32x4 val = tile_out(mi, ni);
                if (g.bias) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) val[r] += g.bias[n + r];
                }
                float* dst = out + (long)m * g.ldc + n;
                if (inner) *(f32x4*)dst = val;
                else if (m < g.M) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (n + r < g.N) dst[r] = val[r];
                }
            }
        }
    }
    if (PL && g.poll.trace && tid == 0 && blockIdx.x == 0 && blockIdx.y == 0) g.poll.trace[8 + (blockIdx.z & 7)] = wall_clock64();   // (end of a workgroup)
}

static inline bool x6_aligned(const float* p, long other_stride) {   // 16-byte loads along the unit-stride dimension
    return ((uintptr_t)p & 15) == 0 && (other_stride & 3) == 0;
}

// true = launched (err holds the launch status).  false = the caller uses the f32 kernel: shape too small for a
// 128x128 tile, no unit stride, or an operand whose rows are not 16-byte aligned (e.g. N = 3706 item columns: scalar
// loads would make the split the bottleneck).  nsplit == 1: C (row stride ldc, + bias); nsplit > 1: slab z at
// C + z*slab_stride, row stride ldc.
bool launch_gemm_x6(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                    int M, int N, int K, const float* bias, int nsplit, int kchunk, size_t slab_stride, hipError_t* err,
                    const float* B2, long sbk2, int n_split, bool small, int planes, float sa, float sb, const SbrPoll* poll) {
    if (planes == 1) { if (M < 1 || N < 48 || K < 32) return false; }      // any number of rows: a row's scores must not depend on
    else if (M < (small ? 48 : 96) || N < (small ? 48 : 96) || K < 32) return false;   // how many rows share the call
    if (B2 && (sbn != 1 || (n_split & 3) || !x6_aligned(B2, sbk2))) return false;
    if (!(sam == 1 || sak == 1) || !(sbk == 1 || sbn == 1)) return false;
    const bool ra = sak != 1, rb = sbk != 1;                        // unit stride along the rows (m / n) instead of k
    if (!x6_aligned(A, ra ? sak : sam) || !x6_aligned(B, rb ? sbk : sbn)) return false;
    GemmX6Args g{A, sam, sak, B, sbk, sbn, C, ldc, slab_stride, M, N, K, kchunk, nsplit > 1 ? nullptr : bias, B2, sbk2, n_split,
                 sa, sb, 1.0f / (sa * sb), SbrPoll{nullptr, 0, nullptr, 0, 1, nullptr, 0, 0, nullptr, nullptr, 0}};
    if (poll) {      // (interior tiles only: x6_issue_sc1)
        if (small || nsplit < 1 || planes < 2 || (M & 127) || (N & 127) || (K & 7) || !poll->slab_lo || poll->n_slabs < 1) return false;
        g.poll = *poll;
    }
    const int tile = small ? 64 : 128;
    const dim3 grid((N + tile - 1) / tile, (M + tile - 1) / tile, nsplit);
#define X6_GO(TW, KH, NP, PL) do { \
        if (ra && rb) gemm_x6_kernel<4, true, 4, true, TW, KH, NP, PL><<<grid, 256, 0, s>>>(g); \
        else if (ra) gemm_x6_kernel<4, true, 4, false, TW, KH, NP, PL><<<grid, 256, 0, s>>>(g); \
        else if (rb) gemm_x6_kernel<4, false, 4, true, TW, KH, NP, PL><<<grid, 256, 0, s>>>(g); \
        else gemm_x6_kernel<4, false, 4, false, TW, KH, NP, PL><<<grid, 256, 0, s>>>(g); } while (0)
    if (poll) { if (planes == 2) X6_GO(4, 1, 2, true); else X6_GO(4, 1, 3, true); }
    else if (planes == 1) { if (small) X6_GO(2, 2, 1, false); else X6_GO(4, 1, 1, false); }
    else if (planes == 2) { if (small) X6_GO(2, 2, 2, false); else X6_GO(4, 1, 2, false); }
    else if (small) X6_GO(2, 2, 3, false); else X6_GO(4, 1, 3, false);
#undef X6_GO
    *err = hipGetLastError();
    return true;
}
