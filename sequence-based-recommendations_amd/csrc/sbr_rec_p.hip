// Pipelined bf16x6 recurrent kernels for Hp = 128 on 4-row tiles ("x6p"): the kernels of BASELINE config C2.
//
// Same arithmetic, data layout and LDS operand planes as rec_fwd_x6s / rec_bwd_x6s (sbr_rec.hip: one workgroup of
// eight waves per 4-row tile, W_hid planes 1-2 in registers and plane 3 in LDS, every lane finishes ONE (row, unit)
// pair on the duplicate MFMA columns).  Two things differ:
//
// 1. No workgroup barrier in the step loop.  The two waves that own the 32 units of k-block kb bump an LDS counter
//    after publishing their slice of h_{t+1} (forward) / dhi_t (backward); a consumer reads counter then planes (the
//    LDS executes one wave's instructions in order, so planes read after a counter value >= target are the published
//    ones) and re-reads both while the counter is short.  The waves of a SIMD pair drift apart by about half a step
//    (measured with the timeline counters below), so one wave's gate math issues while its partner keeps the matrix
//    pipe busy.  Double buffering still suffices: a wave overwrites the buffer of step t only after its step-(t+1)
//    MFMAs, which needed every wave's step-(t+1) slice, which each wave published after its own step-t operand reads.
//
// 2. The step loop is written for instruction count.  With two waves per SIMD the VALU issue port carries 144 MFMAs
//    (576 issue cycles of the 2304 the pipe is busy) plus both waves' non-MFMA instructions at 4+ cycles each; the
//    x6s loop spends ~250 of those per wave and step, most of them 64-bit address arithmetic, and that, not the matrix
//    pipe, set its 3760 cycles per step.  Here every global access is (uniform base advanced on the scalar unit) +
//    (32-bit per-lane byte offset computed once), profiling is a template parameter, and so is the fused gather.
#include "sbr_cell.h"

#define X6P_SPIN_LIMIT (1 << 21)
#ifndef X6P_DBG
#define X6P_DBG 0        // timing experiments only (tools/probes/x6p_variants.sh): wrong results by design
#endif

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

constexpr int HP = 128, R = 4, KBH = HP / 32;

}  // namespace

// ---------------------------------------------------------------------------------------
// forward (GRU / Vanilla: all three W_hid planes stay in registers)
//
// MFMA roles: A operand = h planes (LDS), rows m = 4*row + copy, i.e. every batch row fills four consecutive rows of
// the 16-row tile; B operand = W_hid (registers), columns = the wave's 16 units.  Lane (j, q) then holds
// D[4q .. 4q+3][j] = four copies of (row q, unit wave*16 + j): it finishes that ONE pair from accumulator element 0,
// no select needed.  Bias enters as the C operand of a gate's first MFMA.
//
// LDS: h planes [2 buffers][3 planes][R rows][HROW] | cnt[2]: waves 0-3 / 4-7 add 1 after publishing their slice of
// h_{t+1} (4 per step and half) | tok[4]: wave 4+p adds 1 after issuing its step's MFMAs (the pipe gate of wave p).
//
// One step of one wave:  N1  operand planes + counters: k-blocks 0,1 (published by waves 0-3) for everybody, 2,3 too
//                            for waves 4-7; waves 0-3 wait at the pipe gate
//                        M   72 MFMAs, bare; waves 0-3 fetch k-blocks 2,3 in the middle of k-block 1
//                        N2  gate math, publish + counter, stores of step t, loads for step t+1
// N1 and N2 run under the SIMD partner's M, where a VALU instruction costs ~20 cycles (one issue slot per partner
// MFMA, tools/probes/valu_beside_mfma_probe.hip) and a scalar one ~4: per-step addresses advance on the SALU, stores
// and counter updates are single instructions with scalar bases / precomputed operands.
// ---------------------------------------------------------------------------------------
namespace {
// one lane adds 1 to an LDS counter: exec is all ones around every call site
__device__ __forceinline__ void lds_inc(unsigned addr, int one) {
    asm volatile("s_mov_b64 exec, 1\n\tds_add_u32 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(addr), "v"(one) : "memory");
}
// store with a scalar base: the compiler's own form adds the per-step offset on the VALU
__device__ __forceinline__ void st_s(const void* ubase, unsigned boff, float v) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(boff), "v"(v), "s"(ubase) : "memory");
}
}  // namespace

template <int CELL, bool FUSE, bool PROF>
__global__ void __launch_bounds__(512) rec_fwd_x6p(RecArgs a) {
    constexpr int G = Gates<CELL>::G, KB = KBH, GHP = G * HP;
    static_assert(G <= 3, "W_hid plane 3 does not fit the register file with four gates");
    constexpr int HROW = HP * 2 + 32, PLANEB = R * HROW, BUFB = 3 * PLANEB;
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    char* hbuf = smem_p;
    int* cnt = (int*)(hbuf + 2 * BUFB);                  // [2] publish counters, [4..7] pipe-gate counters
    int* tok = cnt + 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * R + q;                  // this lane's pair: (row q of the tile, unit u)
    const int u = wave * 16 + j;
    const int T = a.T, Bp = a.Bp;
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    const bool roleA = wave < 4;                         // waves w and w + 4 share a SIMD; the older one owns the pipe
    const unsigned lds_cnt_mine = (unsigned)(size_t)(cnt + (roleA ? 0 : 1)), lds_tok = (unsigned)(size_t)(tok + (wave & 3));
    const int one = 1;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 16));
    tmax = __builtin_amdgcn_readfirstlane(max(tmax, __shfl_xor(tmax, 32)));  // workgroup-uniform: all four rows

    bf16x8 W1[G][KB], W2[G][KB], W3[G][KB];              // B operands: lane (j, q) holds W[kb*32 + 8q + e][unit j]
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 b1, b2, b3;
                split3(a.Whid[(size_t)(kb * 32 + 8 * q + e) * GHP + g * HP + u], b1, b2, b3);
                W1[g][kb][e] = b1; W2[g][kb][e] = b2; W3[g][kb][e] = b3;
            }

    // per-lane byte offsets, computed once; the per-step part of every address is uniform and advances on the SALU
    const unsigned bo_h = (unsigned)(row * HP + u) * 4u;                                       // hs rows
    const unsigned bo_g = (unsigned)sbr_blocked_index(0, row, u, Bp, HP) * 4u;                 // saved activations
    const unsigned bo_x = (unsigned)(row * GHP + u) * 4u;                                      // xt rows (not fused)
    const unsigned bo_id = (unsigned)(row * T) * 4u;                                           // ids of this row
    const size_t st_h = (size_t)Bp * HP * 4, st_x = (size_t)Bp * GHP * 4;                      // bytes per time step

    float h = a.hinit[u], cst = 0.f;
    stf(a.hs, bo_h, h);
    const unsigned lds_pub = (unsigned)(q * HROW + u * 2);            // where this lane's h goes inside a plane set
    const unsigned lds_rd = (unsigned)((j >> 2) * HROW + q * 16);     // A operand: tile row m = j holds batch row j >> 2
    auto publish_h = [&](int buf) {
        __bf16 p1, p2, p3;
        split3(h, p1, p2, p3);
        char* base = hbuf + buf * BUFB + lds_pub;
        *(__bf16*)(base) = p1; *(__bf16*)(base + PLANEB) = p2; *(__bf16*)(base + 2 * PLANEB) = p3;
    };
    publish_h(0);

    // Input of step t: a row of xt, or (layer 0, one index per step) gathered here: W_in[id[row][t]] (+ b as the
    // MFMA C operand) (sparse_lstm.py:755 / :1111).  Row t+1 is requested at the end of step t and used after the
    // MFMAs of step t+1; its id was requested one step before that.  Time indices are clamped, not branched around.
    // The bias rides in as the C operand of a gate's first MFMA, except for GRU's candidate gate, whose recurrent
    // part is multiplied by r before the input part (with its bias) is added (sparse_lstm.py:786-792).
    float x[G];
    f32x4 biasv[G];
    float bias_c = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float b = FUSE ? a.gbias[g * HP + u] : 0.f;
        if (CELL == CELL_GRU && g == 2) { bias_c = b; b = 0.f; }
        biasv[g] = f32x4{b, b, b, b};
    }
    auto load_id = [&](int t) -> int { return FUSE ? ldi((const char*)a.gX + (size_t)min(t, T - 1) * 4, bo_id) : 0; };
    auto load_x = [&](int t, int id) {
        if (FUSE) {
            const unsigned bo = (unsigned)id * (unsigned)(GHP * 4) + (unsigned)u * 4u;   // < 2^32: checked by the launcher
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = ldf(a.gWin, bo, g * HP * 4);
        } else {
            const char* xt_t = (const char*)a.xt + (size_t)min(t, T - 1) * st_x;
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = ldf(xt_t, bo_x, g * HP * 4);
        }
    };
    load_x(0, load_id(0));
    int id_next = load_id(1);
    __syncthreads();
    unsigned long long p_c0 = 0, p_r0 = 0, p_spin = 0, p_tok = 0, p_seg[3] = {0, 0, 0}, p_ta = 0, p_tb = 0;
    if (PROF) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    unsigned long long* tl = PROF && blockIdx.x == 0 && lane == 0 ? a.prof + (8 + wave) * 8 : nullptr;   // step-100 timeline

    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    size_t off_t = 0;                                              // t * st_h
    for (int t = 0; t < tmax; ++t) {
        if (PROF) p_ta = clock64();
        if (PROF && tl && (t == 100 || t == 101)) tl[t == 100 ? 0 : 7] = p_ta;
        // ---- N1
        const char* hb = hbuf + (t & 1) * BUFB + lds_rd;
        bf16x8 hp[KB][3];
        int fl[2];
        auto load_half = [&](int half) {                          // counter first, then planes: the LDS keeps a wave's order
            fl[half] = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kb = 2 * half; kb < 2 * half + 2; ++kb) {
                hp[kb][0] = *(const bf16x8*)(hb + kb * 64);
                hp[kb][1] = *(const bf16x8*)(hb + kb * 64 + PLANEB);
                hp[kb][2] = *(const bf16x8*)(hb + kb * 64 + 2 * PLANEB);
            }
        };
        auto ensure_half = [&](int half) {                        // the four producers of this half have published h_t
            if (!(X6P_DBG & 32) && __builtin_amdgcn_readfirstlane(fl[half]) < 4 * t) {
                unsigned long long w0 = 0;
                if (PROF) w0 = clock64();
                int spins = 0;
#pragma clang loop unroll(disable)
                do {
                    asm volatile("" ::: "memory");
                    load_half(half);
                    if (++spins > X6P_SPIN_LIMIT) { atomicOr(a.fault, 2); break; }   // bounded: never hang the GPU
                } while (__builtin_amdgcn_readfirstlane(fl[half]) < 4 * t);
                if (PROF) p_spin += clock64() - w0;
            }
            // the planes are waited for HERE, where the two paths join: no counter waits between the MFMAs
            asm volatile("" :: "v"(hp[2 * half][0]), "v"(hp[2 * half][1]), "v"(hp[2 * half][2]),
                               "v"(hp[2 * half + 1][0]), "v"(hp[2 * half + 1][1]), "v"(hp[2 * half + 1][2]));
        };
        load_half(0);
        if (!roleA) load_half(1);
        ensure_half(0);
        if (!roleA) ensure_half(1);
        __builtin_amdgcn_s_setprio(0);
        // The matrix pipe serves the OLDER wave of a SIMD pair first, strictly (tools/probes/mfma_share_probe.hip: two
        // MFMA streams on one SIMD run 1160 / 2321 cycles per 72, not 1740 / 1740).  A wave 0-3 that started its step
        // as soon as its own group's k-blocks were there would starve its partner's last MFMAs, whose results
        // everybody waits for: so it holds back until the partner has issued its whole step.  Waves 4-7 need no gate,
        // they only ever get the gaps.
        if (roleA && a.x6_pipe >= 2 && !(X6P_DBG & 32)) {
            unsigned long long w0 = 0;
            if (PROF) w0 = clock64();
            int v = __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), spins = 0;
#pragma clang loop unroll(disable)
            while (__builtin_amdgcn_readfirstlane(v) < t) {
                v = __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (++spins > X6P_SPIN_LIMIT) { atomicOr(a.fault, 4); break; }
            }
            if (PROF) p_tok += clock64() - w0;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PROF) { p_tb = clock64(); p_seg[0] += p_tb - p_ta; }
        // ---- M
        f32x4 acc[G];
#define X6P_TERM(HOP, WOP) _Pragma("unroll") for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(HOP, WOP, acc[g]);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb == KB / 2 && roleA) ensure_half(1);
            __builtin_amdgcn_sched_barrier(0);
            if (PROF && tl && t == 100) tl[1 + kb] = clock64();
            if (!((X6P_DBG & 8) && !roleA) && !((X6P_DBG & 16) && roleA)) {
            if (kb == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(hp[kb][0], W3[g][kb], biasv[g]);
            } else { X6P_TERM(hp[kb][0], W3[g][kb]) }
            X6P_TERM(hp[kb][2], W1[g][kb])
            X6P_TERM(hp[kb][1], W2[g][kb])
            } else if (kb == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = biasv[g];
            }
            if (kb == KB / 2 - 1 && roleA) {                      // late enough for the partner's gate math to have
                __builtin_amdgcn_sched_barrier(0);                // published, early enough to hide the LDS latency
                load_half(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!((X6P_DBG & 8) && !roleA) && !((X6P_DBG & 16) && roleA)) {
            X6P_TERM(hp[kb][0], W2[g][kb])
            X6P_TERM(hp[kb][1], W1[g][kb])
            X6P_TERM(hp[kb][0], W1[g][kb])
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef X6P_TERM
        if (!roleA) lds_inc(lds_tok, one);
        asm volatile("s_nop 15");                                 // MFMA D -> VALU read hazard (see rec_fwd_mfma)
        __builtin_amdgcn_s_setprio(3);
        if (PROF) { const unsigned long long tc = clock64(); p_seg[1] += tc - p_tb; p_tb = tc; }
        if (PROF && tl && t == 100) tl[5] = p_tb;
        // ---- N2
        {
            float as[G], xc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { as[g] = acc[g][0]; xc[g] = x[g]; }
            if (CELL == CELL_GRU) xc[G - 1] += bias_c;
            if (X6P_DBG & 1) { h = 0.5f * h + 0.01f * (as[0] + xc[0] + as[G - 1] * xc[G - 1]); sv[0] = as[0]; sv[1] = xc[0]; }
            else cell_forward<CELL, true>(xc, as, t < mylen, h, cst, 0.f, 0.f, 0.f, sv);
        }
        if (t + 1 < tmax) {
            publish_h((t + 1) & 1);
            lds_inc(lds_cnt_mine, one);
            if (PROF) p_seg[2] += clock64() - p_tb;
            if (PROF && tl && t == 100) tl[6] = clock64();
        }
        if (!(X6P_DBG & 4)) {                                     // what BPTT needs of step t
            if (CELL != CELL_VANILLA) {
#pragma unroll
                for (int k = 0; k < 4; ++k) st_s((const char*)a.g[k] + off_t, bo_g, sv[k]);
            }
            st_s((const char*)a.hs + off_t + st_h, bo_h, h);
        }
        off_t += st_h;
        if (!(X6P_DBG & 2)) { load_x(t + 1, id_next); id_next = load_id(t + 2); }
    }
    for (int t = tmax; t < T; ++t) {                              // past the tile's longest row: the state is carried
        stf((char*)a.hs + off_t + st_h, bo_h, h);
        off_t += st_h;
    }
    if (PROF && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        const unsigned long long tot = clock64() - p_c0;
        o[0] = tot; o[1] = wall_clock64() - p_r0; o[2] = tot - p_spin - p_tok; o[3] = p_spin; o[4] = p_tok;
        o[5] = p_seg[0]; o[6] = p_seg[1]; o[7] = p_seg[2];
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static size_t fwd_lds_bytes() { return 2 * 3 * R * (size_t)(HP * 2 + 32) + 64; }

bool sbr_rec_x6p_ok(const RecArgs& a) {
    if (!a.x6_pipe || a.f32_mfma || a.Hp != HP || a.rpt != R || !a.x6_split || a.G > 3) return false;
    if ((size_t)a.Bp * a.G * HP * 4 >= ((size_t)1 << 32)) return false;          // 32-bit per-lane byte offsets
    if (a.gX && (size_t)a.n_in * a.G * HP * 4 >= ((size_t)1 << 32)) return false; // ... also into W_in (fused gather)
    return true;
}

template <int CELL>
static hipError_t launch_fwd_p(hipStream_t s, const RecArgs& a) {
    const size_t lds = fwd_lds_bytes();
    const int nb = a.Bp / R;
#define X6P_LAUNCH(KERNEL) do { \
        (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        KERNEL<<<nb, 512, lds, s>>>(a); } while (0)
    const bool fuse = a.gX != nullptr;
    if (a.prof) { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, true>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, true>)); }
    else { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, false>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, false>)); }
#undef X6P_LAUNCH
    return hipGetLastError();
}

hipError_t launch_rec_forward_x6p(hipStream_t s, const RecArgs& a) {
    return a.cell == SBR_CELL_GRU ? launch_fwd_p<CELL_GRU>(s, a) : launch_fwd_p<CELL_VANILLA>(s, a);
}
