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

// Two operand elements -> their fp16 planes, packed (low half = the first): h1 = fp16(x s), h2 = fp16((x s - h1) 2048), three VALU
// instructions per element (the mixed-precision FMA reads an fp16 half as a source and writes one as a result: no conversion back, no
// pack) where the C form below takes five or six.  Same values bit for bit: x s is exact (s a power of two), so is x s - h1.
__device__ __forceinline__ void x6_split_pair(float x0, float x1, float s, float k2048, unsigned& h1, unsigned& h2) {
    unsigned d1, d2; float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d1) : "v"(x0), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(d1) : "v"(x1), "v"(s));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(s), "v"(d1));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(s), "v"(d1));
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "=v"(d2) : "v"(r0), "v"(k2048));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel_hi:[0,0,0]" : "+v"(d2) : "v"(r1), "v"(k2048));
    h1 = d1; h2 = d2;
}

// ---------------------------------------------------------------------------------------
// Wide tile (round 6): 256 x 128 x 32 per workgroup of 512 threads, LDS double-buffered, one barrier per k step.
// gemm_x6_kernel's 128 x 128 tile is bound by what a k step costs beside its MFMAs -- per loaded element one split (VALU), one LDS
// write, and the element is used for 128 outputs; two barriers per step keep the split of step k + 1 out of the MFMA phase of step
// k (the 2:1 ratio: C5's layer-2 input projection 51 200 x 2 048 x 512 runs at 23 % of the fp16 matrix pipe).  Here a workgroup's
// eight waves own 64 x 64 outputs each: 48 MFMAs (768 cycles of matrix pipe) per wave and k step against 24 loaded elements per
// thread, the next step's operands are split and written to the OTHER LDS stage while this step's MFMAs run, and the raw f32 of the
// step after that is in flight in registers.  f32 operands in (same call sites, same arithmetic per product as gemm_x6_kernel NP = 2 /
// NP = 1, same accumulation order over K inside a slab: results are bitwise those of the 128-wide kernel).
// Shapes: M % 256 == 0, N % 128 == 0, K and the K slab % 32 == 0, 16-byte aligned rows; everything else stays on gemm_x6_kernel.
// RA / RB: unit stride along the rows (m / n) instead of k, as above.  grid = (tiles, K slabs).
template <bool RA, bool RB, int NP>
__global__ void __launch_bounds__(512, 1) gemm_x6w_kernel(GemmX6Args g) {
    constexpr int TM = 256, TN = 128, ROW = 80, PA = TM * ROW, PB = TN * ROW, STAGE = NP * (PA + PB);
    using OPV = std::conditional_t<NP == 2, f16x8g, bf16x8>;
    using EL = std::conditional_t<NP == 2, _Float16, __bf16>;
    typedef EL el4 __attribute__((ext_vector_type(4)));
    typedef EL el2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 15, q = lane >> 4;
    // tile of this workgroup: consecutive workgroup ids go to different XCDs, so ids that share an XCD walk the n tiles of one m
    // tile together (its 256 rows of A stay in that XCD's L2)
    const int tiles_n = g.N / TN, nt = (g.M / TM) * tiles_n;
    int lin = blockIdx.x;
    if ((nt & 7) == 0) lin = (lin & 7) * (nt >> 3) + (lin >> 3);
    const int m0 = (lin / tiles_n) * TM, n0 = (lin % tiles_n) * TN;
    const int kbeg = blockIdx.y * g.kchunk, kend = min(g.K, kbeg + g.kchunk);

    // loaders: every thread moves 16 elements of the A tile and 8 of the B tile per k step
    const float* pa; const float* pb; long sa_k, sb_k; int da, db;      // source, its step per k, byte offset inside a plane
    if (RA) { const int mg = tid & 63, kq = tid >> 6; pa = g.A + (m0 + 4 * mg) + (long)(kbeg + 4 * kq) * g.sak; da = 4 * mg * ROW + kq * 8; }
    else { const int r = tid >> 1, kh = tid & 1; pa = g.A + (long)(m0 + r) * g.sam + kbeg + 16 * kh; da = r * ROW + kh * 32; }
    if (RB) { const int ng = tid & 31, kp = tid >> 5; pb = g.B + (n0 + 4 * ng) + (long)(kbeg + 2 * kp) * g.sbk; db = 4 * ng * ROW + kp * 4; }
    else { const int n = tid >> 2, kq = tid & 3; pb = g.B + (long)(n0 + n) * g.sbn + kbeg + 8 * kq; db = n * ROW + kq * 16; }
    sa_k = RA ? g.sak : 1; sb_k = RB ? g.sbk : 1;
    f32x4 ra[4], rb[2];
    auto load = [&]() {                                    // raw f32 of the next k step (32 k further on)
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *(const f32x4*)(RA ? pa + (long)i * sa_k : pa + 4 * i);
#pragma unroll
        for (int i = 0; i < 2; ++i) rb[i] = *(const f32x4*)(RB ? pb + (long)i * sb_k : pb + 4 * i);
        pa += 32 * sa_k; pb += 32 * sb_k;
    };
    auto conv = [&](float x, float scale, EL& e1, EL& e2) {
        if constexpr (NP == 2) {
            x *= scale;
            asm("" : "+v"(x));                             // one rounding to fp16 for both uses (split2_f16, sbr_rec_p.hip)
            e1 = (_Float16)x; e2 = (_Float16)((x - (float)e1) * 2048.0f);
        } else { e1 = (__bf16)x; e2 = e1; }
    };
    auto store = [&](char* st) {                           // split + LDS write of what `load` fetched
        char* A0 = st + da; char* B0 = st + NP * PA + db;
        if constexpr (RA) {                                // ra[i][e] = A(m = 4 mg + e, k = 4 kq + i): row e gets 4 k = 8 bytes per plane
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                el4 h1, h2;
#pragma unroll
                for (int i = 0; i < 4; ++i) { EL a, b; conv(ra[i][e], g.sa, a, b); h1[i] = a; h2[i] = b; }
                *(el4*)(A0 + e * ROW) = h1;
                if constexpr (NP == 2) *(el4*)(A0 + e * ROW + PA) = h2;
            }
        } else {                                           // ra[i] = 4 consecutive k of the thread's row: 16 k = 32 bytes per plane
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                if constexpr (NP == 2) {
                    u32x4 w1, w2;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        unsigned p1, p2;
                        x6_split_pair(ra[2 * hh + (c >> 1)][2 * (c & 1)], ra[2 * hh + (c >> 1)][2 * (c & 1) + 1], g.sa, 2048.0f, p1, p2);
                        w1[c] = p1; w2[c] = p2;
                    }
                    *(u32x4*)(A0 + hh * 16) = w1;
                    *(u32x4*)(A0 + hh * 16 + PA) = w2;
                } else {
                    OPV h1;
#pragma unroll
                    for (int c = 0; c < 8; ++c) { EL a, b; conv(ra[2 * hh + (c >> 2)][c & 3], g.sa, a, b); h1[c] = a; }
                    *(OPV*)(A0 + hh * 16) = h1;
                }
            }
        }
        if constexpr (RB) {                                // rb[i][e] = B(k = 2 kp + i, n = 4 ng + e): row e gets 2 k = 4 bytes per plane
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if constexpr (NP == 2) {
                    unsigned p1, p2;
                    x6_split_pair(rb[0][e], rb[1][e], g.sb, 2048.0f, p1, p2);
                    *(unsigned*)(B0 + e * ROW) = p1;
                    *(unsigned*)(B0 + e * ROW + PB) = p2;
                } else {
                    el2 h1;
#pragma unroll
                    for (int i = 0; i < 2; ++i) { EL a, b; conv(rb[i][e], g.sb, a, b); h1[i] = a; }
                    *(el2*)(B0 + e * ROW) = h1;
                }
            }
        } else {
            if constexpr (NP == 2) {
                u32x4 w1, w2;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    unsigned p1, p2;
                    x6_split_pair(rb[c >> 1][2 * (c & 1)], rb[c >> 1][2 * (c & 1) + 1], g.sb, 2048.0f, p1, p2);
                    w1[c] = p1; w2[c] = p2;
                }
                *(u32x4*)B0 = w1;
                *(u32x4*)(B0 + PB) = w2;
            } else {
                OPV h1;
#pragma unroll
                for (int c = 0; c < 8; ++c) { EL a, b; conv(rb[c >> 2][c & 3], g.sb, a, b); h1[c] = a; }
                *(OPV*)B0 = h1;
            }
        }
    };
    const f32x4 z = f32x4{0, 0, 0, 0};
    f32x4 acc[4][4], acl[NP == 2 ? 4 : 1][NP == 2 ? 4 : 1];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { acc[a][b] = z; if constexpr (NP == 2) acl[a][b] = z; }
    const int fa = (wm * 64 + j) * ROW + q * 16, fb = NP * PA + (wn * 64 + j) * ROW + q * 16;
    if (kbeg < kend) {
        load();
        store(smw);
        __syncthreads();
        if (kbeg + 32 < kend) load();
    }
    // One k step.  ST: the next step's operands (raw f32 in ra / rb) are split and written to the other LDS stage; LD: the raw f32 of the
    // step after that is requested.  Everything of a step is ONE scheduling region, and its order is prescribed (sched_group_barrier):
    // a wave's split instructions go BETWEEN its own MFMAs -- a matrix instruction occupies the pipe for 16 cycles (32 with the SIMD's
    // other wave taking turns) and the wave issues three to seven VALU instructions meanwhile.  Phases instead of interleaving --
    // the split of all eight waves, then their MFMAs, or the two waves of a SIMD in opposite order -- cost 4 580 cycles per step
    // against 1 536 of matrix pipe: a VALU instruction beside the OTHER wave's MFMA stream issues once per matrix instruction
    // (profiles/round6_variants.txt, call q).
    int cur = 0;
    auto step = [&](auto st_tag, auto ld_tag) {
        constexpr bool ST = decltype(st_tag)::value, LD = decltype(ld_tag)::value;
        const char* st = smw + cur * STAGE;
        OPV a[NP][4], b[NP][4];
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[p][t] = *(const OPV*)(st + fa + p * PA + t * 16 * ROW);
                b[p][t] = *(const OPV*)(st + fb + p * PB + t * 16 * ROW);
            }
#define X6W_TERM(ACC, PA_, PB_) _Pragma("unroll") for (int mi = 0; mi < 4; ++mi) _Pragma("unroll") for (int ni = 0; ni < 4; ++ni) \
            ACC[mi][ni] = x6_mfma(b[PB_][ni], a[PA_][mi], ACC[mi][ni]);
        X6W_TERM(acc, 0, 0)                                // (first: needs plane 0 only, the reads of plane 1 land meanwhile)
        if constexpr (ST) store(smw + (cur ^ 1) * STAGE);
        if constexpr (NP == 2) { X6W_TERM(acl, 0, 1) }
        if constexpr (LD) load();
        if constexpr (NP == 2) { X6W_TERM(acl, 1, 0) }
#undef X6W_TERM
        // the prescribed order
        __builtin_amdgcn_sched_group_barrier(0x100, 8 * NP, 0);                 // fragment reads
        constexpr int NMF = NP == 2 ? 48 : 16, NFEED = NP == 2 ? 32 : 12, VPER = NP == 2 ? 4 : 8;
#pragma unroll
        for (int i = 0; i < NFEED; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (ST) __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);
            if (ST && (i % (NFEED / 6)) == NFEED / 6 - 1) __builtin_amdgcn_sched_group_barrier(0x200, NP, 0);   // 6 x NP LDS writes in all
        }
        if (LD) __builtin_amdgcn_sched_group_barrier(0x020, 6, 0);
#pragma unroll
        for (int i = NFEED; i < NMF; ++i) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __syncthreads();
        cur ^= 1;
    };
    {
        const int ns = (kend - kbeg) / 32;                 // steps; raw f32 of step 1 is in flight
        int i = 0;
        for (; i + 2 < ns; ++i) step(std::true_type{}, std::true_type{});
        if (i + 1 < ns) { step(std::true_type{}, std::false_type{}); ++i; }
        if (i < ns) step(std::false_type{}, std::false_type{});
    }
    float* out = g.C + (size_t)blockIdx.y * g.slab_stride;
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int m = m0 + wm * 64 + mi * 16 + j, n = n0 + wn * 64 + ni * 16 + 4 * q;      // (transposed products: lane (j, q) holds row j, columns 4q..4q+3)
            f32x4 val = acc[mi][ni];
            if constexpr (NP == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) val[r] = fmaf(acl[mi][ni][r], 1.0f / 2048.0f, val[r]) * g.so;
            }
            if (g.bias) {
#pragma unroll
                for (int r = 0; r < 4; ++r) val[r] += g.bias[n + r];
            }
            *(f32x4*)(out + (long)m * g.ldc + n) = val;
        }
}

static thread_local bool g_x6_no_wide = false;
void sbr_gemm_x6_no_wide(bool on) { g_x6_no_wide = on; }

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
    // the wide tile where the shape fills it (see gemm_x6w_kernel); g_x6_no_wide: parity tests of the two kernels against each other
    // (A with unit stride along m -- the weight-gradient GEMMs, contraction over time -- stays on the 128-wide kernel: eight 8-byte LDS
    //  writes per thread and step instead of four 16-byte ones, measured slower there: 582 against 476 us for 512 x 2 048 x 51 200)
    if (!poll && !small && !g_x6_no_wide && !ra && (planes == 1 || planes == 2) && !B2 && (M & 255) == 0 && (N & 127) == 0 && (K & 31) == 0 &&
        (kchunk & 31) == 0 && (ldc & 3) == 0 && ((uintptr_t)C & 15) == 0 && (slab_stride & 3) == 0 &&
        (size_t)(M / 256) * (N / 128) * nsplit >= 128) {
        const dim3 wgrid((M / 256) * (N / 128), nsplit);
        const size_t lds = (size_t)2 * planes * (256 + 128) * 80;
#define X6W_GO(NP) do { \
            if (rb) { SBR_DYN_LDS((gemm_x6w_kernel<false, true, NP>), lds); gemm_x6w_kernel<false, true, NP><<<wgrid, 512, lds, s>>>(g); } \
            else { SBR_DYN_LDS((gemm_x6w_kernel<false, false, NP>), lds); gemm_x6w_kernel<false, false, NP><<<wgrid, 512, lds, s>>>(g); } } while (0)
        if (planes == 2) X6W_GO(2); else X6W_GO(1);
#undef X6W_GO
        *err = hipGetLastError();
        return true;
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
