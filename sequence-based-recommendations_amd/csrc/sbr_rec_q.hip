// bf16x6 recurrent kernels for layers of 32 / 64 padded units on 4-row tiles ("x6q"): BASELINE config C1 (LSTM-20)
// and the reference CLI's default width (--r_l 50).  One workgroup of Hp/16 waves = one wave per SIMD per 4-row tile for
// all T steps; arithmetic and global layout as rec_fwd_x6s / rec_bwd_x6s (sbr_rec.hip), the per-step code as in the
// 128-unit kernels of sbr_rec_p.hip minus everything that deals with two waves sharing a matrix pipe:
//   * no workgroup barrier in the step loop: every wave adds 1 to ONE LDS counter after publishing its 16 units;
//     consumers read counter then planes and re-read while the counter is short (bounded, fault flag);
//   * all three W_hid planes in registers (<= 96 VGPRs at these widths, LSTM included), all operand planes of a step
//     fetched before its MFMAs: the MFMA phase is bare;
//   * activations are the MFMA A operand with every batch row filling four tile rows, so lane (j, q) finishes
//     (row q, unit j) from accumulator element 0; biases ride in as the C operand; GRU's sigmoid gates pre-scaled;
//   * scalar-advanced addresses, single-instruction stores, truncating bf16 split, uniform branch while no row of the
//     tile is masked.
// With one wave per SIMD nothing overlaps the gate math, so a step is (MFMAs) + (everything else) serially and the
// instruction count of "everything else" is the whole game at these sizes (C1: 24 MFMAs = 384 cycles per step).
#include "sbr_rec_p.h"
#include <type_traits>
#include <cstdlib>

namespace {
constexpr int RQ = 4;
// "f16x3" (sbr_rec_p.hip: split2_f16): an operand as a1 + a2 / 2048 in two fp16 planes, a product in three MFMAs instead of
// bf16x6's six; the forward's operand is h in [-1, 1] (not behind a rectifier), the backward's gradient operand has passed
// the reference's clip (<= 100) and is scaled by 2^9.  Same switches as the 128-unit kernels (SBR_X6_F16, SBR_X6_F16_BWD).
typedef _Float16 f16x8q __attribute__((ext_vector_type(8)));
constexpr float Q_F16_LO = 2048.0f, Q_F16_DSCALE = 512.0f;
__device__ __forceinline__ f32x4 mfma_q(const f16x8q& a, const f16x8q& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void split2_q(float v, _Float16& a1, _Float16& a2) {
    asm("" : "+v"(v));                                   // one rounding to fp16 for both uses (see split2_f16)
    a1 = (_Float16)v;
    a2 = (_Float16)((v - (float)a1) * Q_F16_LO);
}
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int CELL, int HQ, bool FUSE, bool F16>
__global__ void __launch_bounds__(HQ * 4) rec_fwd_x6q(RecArgs a) {
    using OPV = std::conditional_t<F16, f16x8q, bf16x8>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int G = Gates<CELL>::G, KB = HQ / 32, NW = HQ / 16, GHP = G * HQ;
    static_assert(G * KB * 12 <= 144, "W_hid planes must fit the register file");
    constexpr int HROW = HQ * 2 + 32, PLANEB = RQ * HROW, BUFB = 3 * PLANEB;
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    char* hbuf = smem_q;
    int* cnt = (int*)(hbuf + 2 * BUFB);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * RQ + q;                 // this lane's pair: (row q of the tile, unit u)
    const int u = wave * 16 + j;
    const int T = a.T, Bp = a.Bp;
    if (threadIdx.x == 0) cnt[0] = 0;
    const unsigned lds_cnt = (unsigned)(size_t)cnt;
    const int one = 1;

    const int mylen = a.len[row];
    int tmax = mylen, tmin = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 16)); tmin = min(tmin, __shfl_xor(tmin, 16));
    tmax = __builtin_amdgcn_readfirstlane(max(tmax, __shfl_xor(tmax, 32)));  // workgroup-uniform: all four rows
    tmin = __builtin_amdgcn_readfirstlane(min(tmin, __shfl_xor(tmin, 32)));  // steps below it: no row is masked

    OPV W1[G][KB], W2[G][KB], W3[F16 ? 1 : G][F16 ? 1 : KB];   // B operands: lane (j, q) holds W[kb*32 + 8q + e][unit j]
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = (CELL == CELL_GRU && g < 2) ? X6P_NLOG2E : 1.0f;     // sigmoid gates: see the gate math
                const float w = sc * a.Whid[(size_t)(kb * 32 + 8 * q + e) * GHP + g * HQ + u];
                if constexpr (F16) {
                    _Float16 b1, b2;
                    split2_q(w, b1, b2);
                    W1[g][kb][e] = b1; W2[g][kb][e] = b2;
                } else {
                    __bf16 b1, b2, b3;
                    split3(w, b1, b2, b3);
                    W1[g][kb][e] = b1; W2[g][kb][e] = b2; W3[g][kb][e] = b3;
                }
            }

    const unsigned bo_h = (unsigned)(row * HQ + u) * 4u;                                       // hs / cs rows
    const unsigned bo_g = (unsigned)sbr_blocked_index(0, row, u, Bp, HQ) * 4u;                 // saved activations
    const unsigned bo_x = (unsigned)(row * GHP + u) * 4u;                                      // xt rows (not fused)
    const unsigned bo_id = (unsigned)(row * T) * 4u;                                           // ids of this row
    const size_t st_h = (size_t)Bp * HQ * 4, st_x = (size_t)Bp * GHP * 4;                      // bytes per time step

    float h = a.hinit[u], cst = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (CELL == CELL_LSTM) {
        cst = a.cinit[u];
        pi = a.peep[u]; pf = a.peep[HQ + u]; po = a.peep[2 * HQ + u];
        stf(a.cs, bo_h, cst);
    }
    stf(a.hs, bo_h, h);
    const unsigned lds_pub = (unsigned)(q * HROW + u * 2);            // where this lane's h goes inside a plane set
    const unsigned lds_rd = (unsigned)((j >> 2) * HROW + q * 16);     // A operand: tile row m = j holds batch row j >> 2
    auto publish_h = [&](int buf) {
        char* base = hbuf + buf * BUFB + lds_pub;
        if constexpr (F16) {
            _Float16 h1, h2;
            split2_q(h, h1, h2);
            *(_Float16*)(base) = h1;
            *(_Float16*)(base + PLANEB) = h2;
        } else {
        unsigned p1, p2, p3;
        split3_trunc(h, p1, p2, p3);
        *(unsigned short*)(base) = (unsigned short)(p1 >> 16);
        *(unsigned short*)(base + PLANEB) = (unsigned short)(p2 >> 16);
        *(unsigned short*)(base + 2 * PLANEB) = (unsigned short)(p3 >> 16);
        }
    };
    publish_h(0);

    // biases ride in as the C operand of a gate's first MFMA, except for GRU's candidate gate (its recurrent part is
    // multiplied by r before the input part with its bias is added, sparse_lstm.py:786-792)
    f32x4 biasv[G];
    float bias_c = 0.f;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float b = FUSE ? a.gbias[g * HQ + u] : 0.f;
        if (CELL == CELL_GRU && g == 2) { bias_c = b; b = 0.f; }
        if (CELL == CELL_GRU && g < 2) b *= X6P_NLOG2E;
        biasv[g] = f32x4{b, b, b, b};
    }
    auto load_id = [&](int t) -> int { return FUSE ? ldi((const char*)a.gX + (size_t)min(t, T - 1) * 4, bo_id) : 0; };
    auto load_x = [&](float (&xd)[G], int t, int id) {
        if (FUSE) {
            const unsigned bo = (unsigned)id * (unsigned)(GHP * 4) + (unsigned)u * 4u;   // < 2^32: checked by the launcher
#pragma unroll
            for (int g = 0; g < G; ++g) xd[g] = ldf(a.gWin, bo, g * HQ * 4);
        } else {
            const char* xt_t = (const char*)a.xt + (size_t)min(t, T - 1) * st_x;
#pragma unroll
            for (int g = 0; g < G; ++g) xd[g] = ldf(xt_t, bo_x, g * HQ * 4);
        }
    };
    // THREE register sets for a step's input row, used by the steps in turn: the set a step has consumed is refilled for the step three
    // ahead, so its loads have 2.6 steps to land (round 6; one set, refilled at the end of step t for step t + 1, gave them the 0.6 of a
    // step between the request and the gate math -- ~0.25 us against an L2 round trip of ~0.5: exposed in every step.  C1: rec_fwd_x6q
    // 97.6 us with one set, 87.5 with two, 81.5 with three, 82.9 with four; the backward chain below: 120.8 / 103.1 / 99.3 / 94.9)
    float xa[G], xb[G], xc[G];
    load_x(xa, 0, load_id(0));
    load_x(xb, 1, load_id(1));
    load_x(xc, 2, load_id(2));
    int ida = load_id(3), idb = load_id(4), idc = load_id(5);
    __syncthreads();

    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    size_t off_t = 0;                                              // t * st_h
    auto fstep = [&](int t, float (&x)[G], int& idn) {
        const char* hb = hbuf + (t & 1) * BUFB + lds_rd;
        OPV hp[KB][NPL];
        int fl;
        auto load_all = [&]() {                                   // counter first, then planes: the LDS keeps a wave's order
            fl = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                hp[kb][0] = *(const OPV*)(hb + kb * 64);
                hp[kb][1] = *(const OPV*)(hb + kb * 64 + PLANEB);
                if constexpr (!F16) hp[kb][NPL - 1] = *(const OPV*)(hb + kb * 64 + 2 * PLANEB);
            }
        };
        load_all();
        if (__builtin_amdgcn_readfirstlane(fl) < NW * t) {       // not every wave has published h_t yet
            int spins = 0;
#pragma clang loop unroll(disable)
            do {
                asm volatile("" ::: "memory");
                load_all();
                if (++spins > X6P_SPIN_LIMIT) { atomicOr(a.fault, 2); break; }   // bounded: never hang the GPU
            } while (__builtin_amdgcn_readfirstlane(fl) < NW * t);
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) asm volatile("" :: "v"(hp[kb][0]), "v"(hp[kb][1]), "v"(hp[kb][NPL - 1]));
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[G];
        if constexpr (F16) {      // acc: h1 w1 (+ bias as the C operand); lo: the low-order products h2 w1 + h1 w2 (/ 2048)
            const f32x4 z4 = f32x4{0, 0, 0, 0};
            f32x4 lo[G];
#pragma unroll
            for (int g = 0; g < G; ++g) lo[g] = z4;
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = mfma_q(hp[kb][1], W1[g][kb], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = mfma_q(hp[kb][0], W2[g][kb], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = mfma_q(hp[kb][0], W1[g][kb], kb == 0 ? biasv[g] : acc[g]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][0] = fmaf(lo[g][0], 1.0f / Q_F16_LO, acc[g][0]);
        } else {
#define X6Q_TERM(HOP, WOP) _Pragma("unroll") for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(HOP, WOP, acc[g]);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(hp[kb][0], W3[g][kb], biasv[g]);
            } else { X6Q_TERM(hp[kb][0], W3[g][kb]) }
            X6Q_TERM(hp[kb][2], W1[g][kb])
            X6Q_TERM(hp[kb][1], W2[g][kb])
            X6Q_TERM(hp[kb][0], W2[g][kb])
            X6Q_TERM(hp[kb][1], W1[g][kb])
            X6Q_TERM(hp[kb][0], W1[g][kb])
        }
#undef X6Q_TERM
        }
        __builtin_amdgcn_sched_barrier(0);                        // (the MFMA D -> VALU read hazard right below is padded by hipcc: same basic block)
        {
            float hn, cn = cst;
            if (CELL == CELL_GRU) {                               // sparse_lstm.py:780-803; r, u columns pre-scaled by -log2(e)
                constexpr int IU = G > 1 ? 1 : 0, IC = G > 2 ? 2 : 0;
                const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(x[0], X6P_NLOG2E, acc[0][0])));
                const float ug = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(x[IU], X6P_NLOG2E, acc[IU][0])));
                const float hc = acc[IC][0];
                const float cc = tanh_fast(fmaf(rg, hc, x[IC] + bias_c));
                hn = fmaf(ug, cc - h, h);                         // (1 - u) h + u c
                sv[0] = rg; sv[1] = ug; sv[2] = cc; sv[3] = hc;
            } else if (CELL == CELL_LSTM) {                       // sparse_lstm.py:397-414
                constexpr int I1 = G > 1 ? 1 : 0, I2 = G > 2 ? 2 : 0, I3 = G > 3 ? 3 : 0;
                const float ig = sigm_fast(x[0] + acc[0][0] + cst * pi);
                const float fg = sigm_fast(x[I1] + acc[I1][0] + cst * pf);
                const float gg = tanh_fast(x[I2] + acc[I2][0]);
                cn = fg * cst + ig * gg;
                const float og = sigm_fast(x[I3] + acc[I3][0] + cn * po);
                hn = og * tanh_fast(cn);
                sv[0] = ig; sv[1] = fg; sv[2] = gg; sv[3] = og;
            } else {
                { const float pre = x[0] + acc[0][0]; hn = a.relu ? fmaxf(pre, 0.0f) : tanh_fast(pre); }
            }
            if (t < tmin) { asm volatile("" : "+v"(hn)); h = hn; cst = cn; }          // uniform branch: no selects
            else { const bool m = t < mylen; h = m ? hn : h; cst = m ? cn : cst; }
        }
        if (t + 1 < tmax) {
            publish_h((t + 1) & 1);
            lds_inc(lds_cnt, one);
        }
        if (CELL != CELL_VANILLA) {                               // what BPTT needs of step t
#pragma unroll
            for (int k = 0; k < 4; ++k) st_s((const char*)a.g[k] + off_t, bo_g, sv[k]);
        }
        st_s((const char*)a.hs + off_t + st_h, bo_h, h);
        if (CELL == CELL_LSTM) st_s((const char*)a.cs + off_t + st_h, bo_h, cst);
        off_t += st_h;
        load_x(x, t + 3, idn); idn = load_id(t + 6);
    };
    {
        int t = 0;
        for (; t + 2 < tmax; t += 3) { fstep(t, xa, ida); fstep(t + 1, xb, idb); fstep(t + 2, xc, idc); }
        if (t < tmax) fstep(t, xa, ida);
        if (t + 1 < tmax) fstep(t + 1, xb, idb);
    }
    for (int t = tmax; t < T; ++t) {                              // past the tile's longest row: the state is carried
        stf((char*)a.hs + off_t + st_h, bo_h, h);
        if (CELL == CELL_LSTM) stf((char*)a.cs + off_t + st_h, bo_h, cst);
        off_t += st_h;
    }
}

// ---------------------------------------------------------------------------------------
// backward:  dh_{t-1}[row][unit] += sum_k dhi_t[row][k] * W_hid[unit][k],  k over the G*Hp gate columns.
// A operand = dhi planes (LDS), B operand = the W_hid rows of the wave's 16 units (registers); every operand plane of
// the step is fetched before its MFMAs.  Chunked BPTT protocol (t_lo / t_hi / state / part) as in rec_bwd_x6s.
// ---------------------------------------------------------------------------------------
template <int CELL, int HQ, bool EXT, bool F16>
__global__ void __launch_bounds__(HQ * 4) rec_bwd_x6q(RecArgs a) {
    using OPV = std::conditional_t<F16, f16x8q, bf16x8>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int G = Gates<CELL>::G, NW = HQ / 16, GHP = G * HQ, KB = GHP / 32;
    static_assert(KB * 24 <= 200, "weights + operand planes must fit the register file");
    constexpr int DROW = GHP * 2 + 32, PLANEB = RQ * DROW, BUFB = 3 * PLANEB;
    extern __shared__ __attribute__((aligned(16))) char smem_q[];
    char* dbuf = smem_q;                                 // [2][3 planes][R rows][DROW]
    int* cnt = (int*)(dbuf + 2 * BUFB);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * RQ + q;
    const int u = wave * 16 + j;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    if (threadIdx.x == 0) cnt[0] = 0;
    const unsigned lds_cnt = (unsigned)(size_t)cnt;
    const int one = 1;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 16));
    tmax = __builtin_amdgcn_readfirstlane(max(tmax, __shfl_xor(tmax, 32)));

    OPV W1[KB], W2[KB], W3[F16 ? 1 : KB];                // B operands: lane (j, q) holds W_hid[unit j][kb*32 + 8q + e]
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float* src = a.Whid + (size_t)u * GHP + kb * 32 + 8 * q;
        const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = e < 4 ? lo[e & 3] : hi[e & 3];
            if constexpr (F16) {
                _Float16 b1, b2;
                split2_q(w, b1, b2);
                W1[kb][e] = b1; W2[kb][e] = b2;
            } else {
                __bf16 b1, b2, b3;
                split3(w, b1, b2, b3);
                W1[kb][e] = b1; W2[kb][e] = b2; W3[kb][e] = b3;
            }
        }
    }

    const unsigned bo_h = (unsigned)(row * HQ + u) * 4u;
    const unsigned bo_g = (unsigned)sbr_blocked_index(0, row, u, Bp, HQ) * 4u;
    const unsigned bo_x = (unsigned)(row * GHP + u) * 4u;
    const size_t st_h = (size_t)Bp * HQ * 4, st_x = (size_t)Bp * GHP * 4;
    const unsigned lds_pub = (unsigned)(q * DROW + u * 2), lds_rd = (unsigned)((j >> 2) * DROW + q * 16);

    const bool first = a.t_hi >= T, last = a.t_lo <= 0;          // first / last launch of the chunked chain
    float dh = 0.f, dc = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (first) { if (a.dh_last) dh = a.dh_last[(size_t)row * HQ + u]; }
    else {
        dh = a.state[(size_t)row * HQ + u];
        if (CELL == CELL_LSTM) dc = a.state[((size_t)Bp + row) * HQ + u];
    }
    if (CELL == CELL_LSTM) { pi = a.peep[u]; pf = a.peep[HQ + u]; po = a.peep[2 * HQ + u]; }
    float sdb[G], sdp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = 0.f;

    // what a step reads of the forward pass; four sets, used by the steps in turn and refilled four steps ahead (see rec_fwd_x6q)
    struct Saved { float sv[4], hprev, cprev, dhe; };
    Saved SA, SB, SC, SD;
#pragma unroll
    for (int k = 0; k < 4; ++k) { SA.sv[k] = 0.f; SB.sv[k] = 0.f; SC.sv[k] = 0.f; SD.sv[k] = 0.f; }
    SA.hprev = SA.cprev = SA.dhe = SB.hprev = SB.cprev = SB.dhe = SC.hprev = SC.cprev = SC.dhe = SD.hprev = SD.cprev = SD.dhe = 0.f;
    float cnew = 0.f, hnew = 0.f;
    auto load_saved = [&](Saved& S, size_t o) {                  // activations of the step at byte offset o = t * st_h
        S.hprev = ldf((const char*)a.hs + o, bo_h);
        if (CELL != CELL_VANILLA) {
#pragma unroll
            for (int k = 0; k < 4; ++k) S.sv[k] = ldf((const char*)a.g[k] + o, bo_g);
        }
        if (CELL == CELL_LSTM) S.cprev = ldf((const char*)a.cs + o, bo_h);
        if (EXT) S.dhe = ldf((const char*)a.dh_ext + o, bo_h);
    };
    __syncthreads();

    const int t_live = min(a.t_hi, tmax);                         // steps [t_live, t_hi) are masked for the whole tile
    for (int t = a.t_hi - 1; t >= max(t_live, a.t_lo); --t) {     // zero rows; dh_ext still accumulates
        if (EXT) dh += a.dh_ext[((size_t)t * Bp + row) * HQ + u];
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[((size_t)t * Bp + row) * GHP + g * HQ + u] = 0.f;
        if (CELL == CELL_GRU) a.dhi[((size_t)t * Bp + row) * HQ + u] = 0.f;
    }
    if (t_live > a.t_lo) {
        load_saved(SA, (size_t)(t_live - 1) * st_h);
        load_saved(SB, (size_t)max(t_live - 2, a.t_lo) * st_h);
        load_saved(SC, (size_t)max(t_live - 3, a.t_lo) * st_h);
        load_saved(SD, (size_t)max(t_live - 4, a.t_lo) * st_h);
        const size_t o1 = (size_t)t_live * Bp * HQ + (size_t)row * HQ + u;
        if (CELL == CELL_LSTM) cnew = a.cs[o1];
        if (CELL == CELL_VANILLA) hnew = a.hs[o1];
    }
    size_t off_h = (size_t)(t_live - 1) * st_h, off_x = (size_t)(t_live - 1) * st_x;   // of step t
    int n = 0;                                                    // steps done
    auto bstep = [&](int t, Saved& S) {
        float (&sv)[4] = S.sv; float& hprev = S.hprev; float& cprev = S.cprev;
        if (EXT) dh += S.dhe;
        char* lds = dbuf + (n & 1) * BUFB;
        float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
        cell_backward<CELL, true>(t < mylen, clip, dh, dc, sv, hprev, cprev, cnew, hnew, pi, pf, po, dxi, dhi, dp, a.relu != 0);
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[g] += dxi[g];
        if (CELL == CELL_LSTM) { sdp[0] += dp[0]; sdp[1] += dp[1]; sdp[2] += dp[2]; }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            char* base = lds + lds_pub + g * HQ * 2;
            if constexpr (F16) {
                _Float16 d1, d2;
                split2_q(dhi[g] * Q_F16_DSCALE, d1, d2);          // |dhi| <= clip <= 100: below fp16's 65504
                *(_Float16*)(base) = d1;
                *(_Float16*)(base + PLANEB) = d2;
            } else {
            unsigned p1, p2, p3;
            split3_trunc(dhi[g], p1, p2, p3);
            *(unsigned short*)(base) = (unsigned short)(p1 >> 16);
            *(unsigned short*)(base + PLANEB) = (unsigned short)(p2 >> 16);
            *(unsigned short*)(base + 2 * PLANEB) = (unsigned short)(p3 >> 16);
            }
        }
        lds_inc(lds_cnt, one);
        if (CELL == CELL_LSTM) cnew = cprev;
        if (CELL == CELL_VANILLA) hnew = hprev;
        {
            const char* dx_t = (const char*)a.dxt + off_x;
            st_si<0>(dx_t, bo_x, dxi[0]);
            if (G > 1) st_si<HQ * 4>(dx_t, bo_x, dxi[G > 1 ? 1 : 0]);
            if (G > 2) st_si<2 * HQ * 4>(dx_t, bo_x, dxi[G > 2 ? 2 : 0]);
            if (G > 3) st_si<3 * HQ * 4>(dx_t, bo_x, dxi[G > 3 ? 3 : 0]);
            if (CELL == CELL_GRU) st_si<0>((const char*)a.dhi + off_h, bo_h, dhi[G - 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        load_saved(S, t - 4 >= a.t_lo ? off_h - 4 * st_h : (size_t)a.t_lo * st_h);      // step t - 4 into the set this step has consumed: unconditional, clamped
        __builtin_amdgcn_sched_barrier(0);
        off_h -= st_h; off_x -= st_x;
        const char* db = lds + lds_rd;
        OPV dpl[KB][NPL];
        int fl;
        auto load_all = [&]() {
            fl = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                dpl[kb][0] = *(const OPV*)(db + kb * 64);
                dpl[kb][1] = *(const OPV*)(db + kb * 64 + PLANEB);
                if constexpr (!F16) dpl[kb][NPL - 1] = *(const OPV*)(db + kb * 64 + 2 * PLANEB);
            }
        };
        load_all();
        if (__builtin_amdgcn_readfirstlane(fl) < NW * (n + 1)) {
            int spins = 0;
#pragma clang loop unroll(disable)
            do {
                asm volatile("" ::: "memory");
                load_all();
                if (++spins > X6P_SPIN_LIMIT) { atomicOr(a.fault, 2); break; }
            } while (__builtin_amdgcn_readfirstlane(fl) < NW * (n + 1));
        }
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) asm volatile("" :: "v"(dpl[kb][0]), "v"(dpl[kb][1]), "v"(dpl[kb][NPL - 1]));
        __builtin_amdgcn_sched_barrier(0);
        const f32x4 z4 = f32x4{0, 0, 0, 0};
        f32x4 acc[3] = {z4, z4, z4};
        if constexpr (F16) {      // acc[0]: d1 w1; acc[1], acc[2]: the low-order products (/ 2048); all / 2^9 (the operand's scale)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                acc[1] = mfma_q(dpl[kb][1], W1[kb], acc[1]);
                acc[2] = mfma_q(dpl[kb][0], W2[kb], acc[2]);
                acc[0] = mfma_q(dpl[kb][0], W1[kb], acc[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            dh += fmaf(acc[1][0] + acc[2][0], 1.0f / Q_F16_LO, acc[0][0]) * (1.0f / Q_F16_DSCALE);
        } else {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            acc[0] = MFMA_BF16(dpl[kb][0], W3[kb], acc[0]);
            acc[1] = MFMA_BF16(dpl[kb][2], W1[kb], acc[1]);
            acc[2] = MFMA_BF16(dpl[kb][1], W2[kb], acc[2]);
            acc[0] = MFMA_BF16(dpl[kb][0], W2[kb], acc[0]);
            acc[1] = MFMA_BF16(dpl[kb][1], W1[kb], acc[1]);
            acc[2] = MFMA_BF16(dpl[kb][0], W1[kb], acc[2]);
        }
        __builtin_amdgcn_sched_barrier(0);                        // (MFMA D -> VALU read hazard: padded by hipcc, same basic block)
        dh += acc[0][0] + acc[1][0] + acc[2][0];
        }
    };
    {
        int t = t_live - 1;
        for (; t - 3 >= a.t_lo; t -= 4) { bstep(t, SA); ++n; bstep(t - 1, SB); ++n; bstep(t - 2, SC); ++n; bstep(t - 3, SD); ++n; }
        if (t >= a.t_lo) { bstep(t, SA); ++n; }
        if (t - 1 >= a.t_lo) { bstep(t - 1, SB); ++n; }
        if (t - 2 >= a.t_lo) { bstep(t - 2, SC); ++n; }
    }

    if (!last) {                                                  // hand dh / dc to the next chunk launch
        a.state[(size_t)row * HQ + u] = dh;
        if (CELL == CELL_LSTM) a.state[((size_t)Bp + row) * HQ + u] = dc;
    }
    float* part = a.part + ((size_t)a.chunk * gridDim.x + blockIdx.x) * (GHP + 5 * HQ);
    float v[G + 5];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = sdb[g];
    v[G] = sdp[0]; v[G + 1] = sdp[1]; v[G + 2] = sdp[2];
    v[G + 3] = last ? dc : 0.f; v[G + 4] = last ? dh : 0.f;      // init-state gradients come from the last chunk only
#pragma unroll
    for (int k = 0; k < G + 5; ++k) {
        float sum = v[k];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);                               // over the tile's 4 rows (q)
        v[k] = sum;
    }
    if (q == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) part[g * HQ + u] = v[g];
#pragma unroll
        for (int k = 0; k < 5; ++k) part[GHP + k * HQ + u] = v[G + k];
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool sbr_rec_x6q_ok(const RecArgs& a) {
    if (!a.x6_pipe || a.f32_mfma || !(a.Hp == 32 || a.Hp == 64) || a.rpt != RQ || !a.x6_split || a.prof) return false;
    if ((size_t)a.Bp * a.G * a.Hp * 4 >= ((size_t)1 << 32)) return false;               // 32-bit per-lane byte offsets
    if (a.gX && (size_t)a.n_in * a.G * a.Hp * 4 >= ((size_t)1 << 32)) return false;     // ... also into W_in (fused gather)
    return true;
}

#define X6Q_LAUNCH(KERNEL, THREADS, LDS) do { \
        SBR_DYN_LDS(KERNEL, (LDS)); \
        KERNEL<<<nb, THREADS, LDS, s>>>(a); } while (0)

template <int CELL, int HQ>
static hipError_t launch_fwd_q(hipStream_t s, const RecArgs& a) {
    const size_t lds = 2 * 3 * RQ * (size_t)(HQ * 2 + 32) + 64;
    const int nb = a.Bp / RQ;
    const char* fe = getenv("SBR_X6_F16");                         // read per launch: the tests flip it
    const bool f16 = (fe ? atoi(fe) != 0 : true) && !a.relu;       // (a rectified state is unbounded)
    if (f16) { if (a.gX) X6Q_LAUNCH((rec_fwd_x6q<CELL, HQ, true, true>), HQ * 4, lds); else X6Q_LAUNCH((rec_fwd_x6q<CELL, HQ, false, true>), HQ * 4, lds); }
    else { if (a.gX) X6Q_LAUNCH((rec_fwd_x6q<CELL, HQ, true, false>), HQ * 4, lds); else X6Q_LAUNCH((rec_fwd_x6q<CELL, HQ, false, false>), HQ * 4, lds); }
    return hipGetLastError();
}
template <int CELL, int HQ>
static hipError_t launch_bwd_q(hipStream_t s, const RecArgs& a) {
    const size_t lds = 2 * 3 * RQ * (size_t)(Gates<CELL>::G * HQ * 2 + 32) + 64;
    const int nb = a.Bp / RQ;
    const char* fe = getenv("SBR_X6_F16_BWD");                     // read per launch: the tests flip it
    const bool f16 = (fe ? atoi(fe) != 0 : true) && a.clip > 0.0f && a.clip <= 100.0f;      // the clip bounds the gradient operand
    if (f16) { if (a.dh_ext) X6Q_LAUNCH((rec_bwd_x6q<CELL, HQ, true, true>), HQ * 4, lds); else X6Q_LAUNCH((rec_bwd_x6q<CELL, HQ, false, true>), HQ * 4, lds); }
    else { if (a.dh_ext) X6Q_LAUNCH((rec_bwd_x6q<CELL, HQ, true, false>), HQ * 4, lds); else X6Q_LAUNCH((rec_bwd_x6q<CELL, HQ, false, false>), HQ * 4, lds); }
    return hipGetLastError();
}
#undef X6Q_LAUNCH

#define X6Q_DISPATCH(FN) \
    switch (a.cell) { \
        case SBR_CELL_LSTM: return a.Hp == 32 ? FN<CELL_LSTM, 32>(s, a) : FN<CELL_LSTM, 64>(s, a); \
        case SBR_CELL_GRU: return a.Hp == 32 ? FN<CELL_GRU, 32>(s, a) : FN<CELL_GRU, 64>(s, a); \
        default: return a.Hp == 32 ? FN<CELL_VANILLA, 32>(s, a) : FN<CELL_VANILLA, 64>(s, a); \
    }
hipError_t launch_rec_forward_x6q(hipStream_t s, const RecArgs& a) { X6Q_DISPATCH(launch_fwd_q) }
hipError_t launch_rec_backward_x6q(hipStream_t s, const RecArgs& a) { X6Q_DISPATCH(launch_bwd_q) }
#undef X6Q_DISPATCH
