// "x6r": the 128-unit recurrent kernels with ONE wave per SIMD that owns TWO unit tiles (round 3).  Same arithmetic, global data
// layout, lane <-> (row, unit) mapping and packed fp16 planes as rec_*_x6p (sbr_rec_p.hip: 4-row tiles, one workgroup per tile
// for all T steps, W_hid in registers), so either direction pairs with the other family's kernel; what changes is who overlaps
// with whom.
//
// x6p runs two waves per SIMD and lets one wave's gate math run under its partner's MFMA phase.  Measured (tools/probes/
// own_valu_probe, profiles/round3_b_probes.txt): a VALU instruction beside the PARTNER's MFMA stream costs 10 - 20 cycles, in
// the gaps of the wave's OWN stream the first one per MFMA is free and further ones ~4 -- and the s_nop trick (a stream that
// leaves the issue port alone between its MFMAs) does not change the partner's price.  With the packed planes a step is 24
// MFMAs per wave (384 cycles) against ~35 VALU instructions of gate math, publication and bookkeeping per wave at partner
// prices: the step had become a chain of those (1575 cycles forward against 768 of MFMA issue per SIMD).
//
// Here wave w owns the unit tiles X = [16 w, 16 w + 16) and Y = [64 + 16 w, ...): 48 MFMAs per step in four groups of 12
// (GRU) -- tile x K half, the K halves being the h planes published by the X tiles (units 0 .. 63) and by the Y tiles -- and the
// gate math of one tile sits in the gaps of the OTHER tile's MFMAs, in the same instruction stream:
//
//     wait Y(t) | G3: X x K_y | G4: Y x K_y  ||  gate math X(t) | publish X(t+1) | read K_x(t+1) | gate math Y(t) | publish Y(t+1)
//               | G1: X x K_x(t+1) | G2: Y x K_x(t+1) | read K_y(t+1) ...
//
// The publish -> visible -> operand-read latency of the X half runs under the gate math of Y, that of the Y half under G1 and
// G2 of the next step.  LDS: h planes [2 buffers][2 planes][R][HROW], two counters (X / Y halves published: 4 per step each),
// the fused gather's offset table and row ring (one piece per tile, wave and step).  No workgroup barrier in the loop.
#include "sbr_rec_p.h"

#ifndef X6R_V2
#define X6R_V2 1         // forward: schedule 2 (see the loop); 0: the first schedule (stores in a burst, reads where they are used)
#endif
#ifndef X6R_SCHED
#define X6R_SCHED 1      // 1: the gate math of tile X is left to the compiler's scheduler inside G4's region; 0: fenced behind it
#endif

namespace {

constexpr int HP = 128, R = 4, KB = HP / 32;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float F16_LO = 2048.0f;
__device__ __forceinline__ f32x4 mf(const f16x8& a, const f16x8& b, const f32x4& c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void split2(float v, _Float16& a1, _Float16& a2) {      // see split2_f16 in sbr_rec_p.hip (v pinned by the caller)
    a1 = (_Float16)v;
    a2 = (_Float16)((v - (float)a1) * F16_LO);
}

}  // namespace

template <int CELL, bool FUSE>
__global__ void __launch_bounds__(256) rec_fwd_x6r(RecArgs a) {
    constexpr int G = Gates<CELL>::G, GHP = G * HP;
    constexpr int HROW = HP * 2 + 32, PLANEB = R * HROW, BUFB = 2 * PLANEB;
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    char* hbuf = smem_r;
    int* cnt = (int*)(hbuf + 2 * BUFB);                  // [0]: X halves published, [1]: Y halves (4 per step each)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * R + q;
    const int T = a.T, Bp = a.Bp;
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    const int one = 1;
    const unsigned lds_cnt = (unsigned)(size_t)cnt;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 16));
    tmax = __builtin_amdgcn_readfirstlane(max(tmax, __shfl_xor(tmax, 32)));

    // tile 0 = X (units 16 w ..), tile 1 = Y (units 64 + 16 w ..)
    int u[2];
    u[0] = wave * 16 + j; u[1] = 64 + wave * 16 + j;
    f16x8 W1[2][G][KB], W2[2][G][KB];                     // B operands: lane (j, q) holds W_hid[kb*32 + 8q + e][gate g, unit u]
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sc = (CELL == CELL_GRU && g < 2) ? X6P_NLOG2E : 1.0f;
                    float w = sc * a.Whid[(size_t)(kb * 32 + 8 * q + e) * GHP + g * HP + u[tl]];
                    asm("" : "+v"(w));
                    _Float16 b1, b2;
                    split2(w, b1, b2);
                    W1[tl][g][kb][e] = b1; W2[tl][g][kb][e] = b2;
                }

    unsigned bo_h[2], bo_g[2], bo_x[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        bo_h[tl] = (unsigned)(row * HP + u[tl]) * 4u;
        bo_g[tl] = (unsigned)sbr_blocked_index(0, row, u[tl], Bp, HP) * 4u;
        bo_x[tl] = (unsigned)(row * GHP + u[tl]) * 4u;
    }
    const size_t st_h = (size_t)Bp * HP * 4, st_x = (size_t)Bp * GHP * 4;

    float h[2], c[2] = {0.f, 0.f}, pi[2] = {0.f, 0.f}, pf[2] = {0.f, 0.f}, po[2] = {0.f, 0.f};
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        h[tl] = a.hinit[u[tl]];
        stf(a.hs, bo_h[tl], h[tl]);
        if (CELL == CELL_LSTM) {
            c[tl] = a.cinit[u[tl]]; pi[tl] = a.peep[u[tl]]; pf[tl] = a.peep[HP + u[tl]]; po[tl] = a.peep[2 * HP + u[tl]];
            stf(a.cs, bo_h[tl], c[tl]);
        }
    }
    const unsigned lds_rd = (unsigned)((j >> 2) * HROW + q * 16 + (j & 1) * PLANEB);      // packed planes (sbr_rec_p.hip)
    unsigned lds_pub[2];
    lds_pub[0] = (unsigned)(q * HROW + u[0] * 2); lds_pub[1] = (unsigned)(q * HROW + u[1] * 2);
    auto publish = [&](int tl, int buf) {
        char* base = hbuf + buf * BUFB + lds_pub[tl];
        _Float16 h1, h2;
        split2(h[tl], h1, h2);
        *(_Float16*)(base) = h1;
        *(_Float16*)(base + PLANEB) = h2;
    };
    publish(0, 0); publish(1, 0);

    float x[2][G];
    f32x4 biasv[2][G];
    float bias_c[2] = {0.f, 0.f};
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float b = FUSE ? a.gbias[g * HP + u[tl]] : 0.f;
            if (CELL == CELL_GRU && g == 2) { bias_c[tl] = b; b = 0.f; }
            if (CELL == CELL_GRU && g < 2) b *= X6P_NLOG2E;
            biasv[tl][g] = f32x4{b, 0.f, 0.f, 0.f};
        }
    // fused gather: as in rec_fwd_x6p, one 16-byte-per-lane LDS-DMA piece per TILE, wave and step, XPD steps ahead
    constexpr int XPD = 4, NSF = CELL == CELL_VANILLA ? 1 : (CELL == CELL_LSTM ? 6 : 5), XSTG = G * 256;
    constexpr int NVM = 2 * NSF + 2;                               // vector-memory operations of one iteration: stores of both tiles, two pieces
    constexpr int XOFF_OFF = (2 * BUFB + 64 + 255) & ~255;
    const int xring_off = (XOFF_OFF + R * T * 4 + 255) & ~255;
    unsigned* xo_tab = (unsigned*)(smem_r + XOFF_OFF);
    const unsigned xring_wave = (unsigned)(size_t)(smem_r + xring_off) + (unsigned)wave * (XPD * 2 * XSTG);
    const char* xring_lane = smem_r + xring_off + wave * (XPD * 2 * XSTG) + lane * 4;
    const unsigned* xo_row = xo_tab + ((lane >> 2) & 3) * T;
    const unsigned bo_lane = (unsigned)((lane >> 4) * HP + wave * 16 + (lane & 3) * 4) * 4u;      // tile X; tile Y: + 64 units
    constexpr unsigned long long XMASK = G >= 4 ? ~0ull : ((1ull << (16 * G)) - 1ull);
    unsigned bo_nxt = 0;
    auto dma_x = [&](unsigned bo, int slot) {
        lds_dma_x4(xring_wave + (unsigned)slot * (2 * XSTG), a.gWin, bo, XMASK);
        lds_dma_x4(xring_wave + (unsigned)slot * (2 * XSTG) + XSTG, a.gWin, bo + 64 * 4, XMASK);
    };
    auto load_x = [&](int t) {
        const char* xt_t = (const char*)a.xt + (size_t)min(t, T - 1) * st_x;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
            for (int g = 0; g < G; ++g) x[tl][g] = ldf(xt_t, bo_x[tl], g * HP * 4);
    };
    if constexpr (FUSE) {
        for (int i = threadIdx.x; i < R * T; i += 256) {
            const int r = i / T;
            xo_tab[i] = (unsigned)a.gX[(size_t)(blockIdx.x * R + r) * T + (i - r * T)] * (unsigned)(GHP * 4);
        }
        __syncthreads();
#pragma unroll
        for (int d = 0; d < XPD; ++d) dma_x(xo_row[min(d, T - 1)] + bo_lane, d);
        bo_nxt = xo_row[min(XPD, T - 1)] + bo_lane;
        wait_vm<0>();
    } else {
        load_x(0);
    }
    __syncthreads();

    int tmin = mylen;
    tmin = min(tmin, __shfl_xor(tmin, 16));
    tmin = __builtin_amdgcn_readfirstlane(min(tmin, __shfl_xor(tmin, 32)));

    // operand reads of one K half (two k-blocks) of h_t: counter first, then the planes (the LDS keeps a wave's order); re-read
    // while the counter is short.  Bounded: a spin that gives up raises the fault flag instead of hanging the GPU.
    f16x8 hp[KB];
    [[maybe_unused]] auto read_half = [&](int half, int t) {
        const char* hb = hbuf + (t & 1) * BUFB + lds_rd + half * 128;
        int fl = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        hp[2 * half] = *(const f16x8*)(hb);
        hp[2 * half + 1] = *(const f16x8*)(hb + 64);
        if (__builtin_amdgcn_readfirstlane(fl) < 4 * t) {
            int spins = 0;
#pragma clang loop unroll(disable)
            do {
                asm volatile("" ::: "memory");
                fl = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
                hp[2 * half] = *(const f16x8*)(hb);
                hp[2 * half + 1] = *(const f16x8*)(hb + 64);
                if (++spins > X6P_SPIN_LIMIT) { atomicOr(a.fault, 2); break; }
            } while (__builtin_amdgcn_readfirstlane(fl) < 4 * t);
        }
    };
    f32x4 acc[2][G], acl[2][G];
    // one group: tile tl over K half `half`; first: the accumulators start from the bias / zero
    auto group = [&](int tl, int half, bool first) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kb = 2 * half + kk;
#pragma unroll
            for (int g = 0; g < G; ++g) acl[tl][g] = mf(hp[kb], W2[tl][g][kb], (first && kk == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acl[tl][g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[tl][g] = mf(hp[kb], W1[tl][g][kb], (first && kk == 0) ? biasv[tl][g] : acc[tl][g]);
        }
    };
    float sv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // gate math of tile tl for step t (sparse_lstm.py:780-803 / :397-423 / :1133-1150), as in rec_fwd_x6p
    auto gate = [&](int tl, int t) {
        float aa[G];
#pragma unroll
        for (int g = 0; g < G; ++g) aa[g] = fmaf(acc[tl][g][1] + acl[tl][g][0], 1.0f / F16_LO, acc[tl][g][0]);
        float hn;
        if (CELL == CELL_GRU) {
            constexpr int IU = G > 1 ? 1 : 0, IC = G > 2 ? 2 : 0;
            const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(x[tl][0], X6P_NLOG2E, aa[0])));
            const float ug = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(x[tl][IU], X6P_NLOG2E, aa[IU])));
            const float hc = aa[IC];
            const float cc = tanh_fast(fmaf(rg, hc, x[tl][IC] + bias_c[tl]));
            hn = fmaf(ug, cc - h[tl], h[tl]);
            sv[tl][0] = rg; sv[tl][1] = ug; sv[tl][2] = cc; sv[tl][3] = hc;
        } else if (CELL == CELL_LSTM) {
            hn = h[tl];
            cell_forward<CELL_LSTM, true>(x[tl], aa, t < mylen, hn, c[tl], pi[tl], pf[tl], po[tl], sv[tl]);
        } else {
            const float pre = x[tl][0] + aa[0];
            hn = a.relu ? fmaxf(pre, 0.0f) : tanh_fast(pre);
        }
        if (CELL == CELL_LSTM) h[tl] = hn;
        else if (t < tmin) h[tl] = hn;
        else h[tl] = t < mylen ? hn : h[tl];
        asm volatile("" : "+v"(h[tl]));                              // pinned for the split
    };
    float sv_st[2][4], h_st[2], c_st[2];                             // what the next step's first group stores
    auto store_step = [&](size_t off) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            if (CELL != CELL_VANILLA) {
#pragma unroll
                for (int k = 0; k < 4; ++k) st_s((const char*)a.g[k] + off, bo_g[tl], sv_st[tl][k]);
            }
            st_s((const char*)a.hs + off + st_h, bo_h[tl], h_st[tl]);
            if (CELL == CELL_LSTM) st_s((const char*)a.cs + off + st_h, bo_h[tl], c_st[tl]);
        }
    };
    auto take_x = [&](int xslot) {                                   // the rows of this step out of the ring (DMA issued XPD iterations ago)
        if constexpr (FUSE) {
            wait_vm<(XPD - 1) * NVM>();
            const char* xp = xring_lane + xslot * (2 * XSTG);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int g = 0; g < G; ++g) x[tl][g] = *(const float*)(xp + tl * XSTG + g * 256);
        }
    };

    size_t off_t = 0;
    int xslot = 0;
#if X6R_V2
    // Schedule 2: nothing but the gate math of tile Y stands outside an MFMA stream.  The stores of step t - 1 go one per gap into
    // G3, the operand reads of a K half are ISSUED a group early (K_y inside G1 / G2's region, K_x behind Y's gate math) and only
    // checked where they are used, the fused gather's pieces, table and row reads share the region of Y's gate math.
    int flh[2] = {0, 0};
    auto issue_half = [&](int half, int t) {                         // counter, then planes (the LDS keeps a wave's order)
        const char* hb = hbuf + (t & 1) * BUFB + lds_rd + half * 128;
        flh[half] = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
        hp[2 * half] = *(const f16x8*)(hb);
        hp[2 * half + 1] = *(const f16x8*)(hb + 64);
    };
    auto check_half = [&](int half, int t) {                         // ... re-read while the counter was short
        if (__builtin_amdgcn_readfirstlane(flh[half]) < 4 * t) {
            const char* hb = hbuf + (t & 1) * BUFB + lds_rd + half * 128;
            int spins = 0;
#pragma clang loop unroll(disable)
            do {
                asm volatile("" ::: "memory");
                flh[half] = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
                hp[2 * half] = *(const f16x8*)(hb);
                hp[2 * half + 1] = *(const f16x8*)(hb + 64);
                if (++spins > X6P_SPIN_LIMIT) { atomicOr(a.fault, 2); break; }
            } while (__builtin_amdgcn_readfirstlane(flh[half]) < 4 * t);
        }
        asm volatile("" :: "v"(hp[2 * half]), "v"(hp[2 * half + 1]));
    };
    auto store_one = [&](size_t off, int i) {                        // store i of the 2 NSF a step saves (both tiles)
        const int tl = i / NSF, k = i % NSF;
        if (CELL != CELL_VANILLA && k < 4) st_s((const char*)a.g[k] + off, bo_g[tl], sv_st[tl][k]);
        else if (k == (CELL == CELL_VANILLA ? 0 : 4)) st_s((const char*)a.hs + off + st_h, bo_h[tl], h_st[tl]);
        else st_s((const char*)a.cs + off + st_h, bo_h[tl], c_st[tl]);
    };
    // G3 with one store behind every MFMA pair
    auto group3_stores = [&](size_t off) {
        int si = 0;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int kb = 2 + kk;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acl[0][g] = mf(hp[kb], W2[0][g][kb], acl[0][g]);
                acc[0][g] = mf(hp[kb], W1[0][g][kb], acc[0][g]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < (2 * NSF + 2 * G - 1) / (2 * G); ++r)
                    if (si < 2 * NSF) { store_one(off, si); ++si; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if (tmax > 0) {
        issue_half(0, 0);
        check_half(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 0, true);
        issue_half(1, 0);
        group(1, 0, true);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (FUSE) { if (tmax > 0) take_x(0); }
    for (int t = 0; t < tmax; ++t) {
        check_half(1, t);
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) group3_stores(off_t - st_h); else group(0, 1, false);      // G3: X x K_y (+ the stores of step t - 1)
        __builtin_amdgcn_sched_barrier(0);
        group(1, 1, false);                                          // G4: Y x K_y ...
        gate(0, t);                                                  // ... with X's gate math in its gaps
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 1 < tmax;
        if (more) {
            publish(0, (t + 1) & 1);
            lds_inc(lds_cnt, one);
        }
        __builtin_amdgcn_sched_barrier(0);
        gate(1, t);                                                  // Y's gate math: the X half becomes visible meanwhile
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sv_st[tl][k] = sv[tl][k];
            h_st[tl] = h[tl]; c_st[tl] = c[tl];
        }
        off_t += st_h;
        if constexpr (FUSE) {
            dma_x(bo_nxt, xslot);
            xslot = xslot + 1 == XPD ? 0 : xslot + 1;
            bo_nxt = xo_row[min(t + XPD + 1, T - 1)] + bo_lane;
            // the rows of step t + 1 (x is free: both gates have read it): their pieces left XPD - 1 iterations ago, younger than
            // them are the stores and pieces of the XPD - 1 iterations since
            wait_vm<(XPD - 1) * NVM>();
            const char* xp = xring_lane + xslot * (2 * XSTG);
#pragma unroll
            for (int tl = 0; tl < 2; ++tl)
#pragma unroll
                for (int g = 0; g < G; ++g) x[tl][g] = *(const float*)(xp + tl * XSTG + g * 256);
        } else load_x(t + 1);
        if (more) issue_half(0, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            publish(1, (t + 1) & 1);
            lds_inc(lds_cnt + 4, one);
            check_half(0, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            group(0, 0, true);                                       // G1 of step t + 1
            issue_half(1, t + 1);                                    // (Y was published a group ago)
            group(1, 0, true);                                       // G2
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#else
    if (tmax > 0) {
        read_half(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 0, true); group(1, 0, true);                        // G1, G2 of step 0
        __builtin_amdgcn_sched_barrier(0);
    }
    for (int t = 0; t < tmax; ++t) {
        take_x(xslot);
        read_half(1, t);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 1, false);                                          // G3: X x K_y
        __builtin_amdgcn_sched_barrier(0);
        if (t > 0) { store_step(off_t - st_h); __builtin_amdgcn_sched_barrier(0); }      // the stores of step t - 1, among the MFMAs
        group(1, 1, false);                                          // G4: Y x K_y ...
        if (!X6R_SCHED) __builtin_amdgcn_sched_barrier(0);
        gate(0, t);                                                  // ... with X's gate math in its gaps
        __builtin_amdgcn_sched_barrier(0);
        const bool more = t + 1 < tmax;
        if (more) {
            publish(0, (t + 1) & 1);
            lds_inc(lds_cnt, one);
        }
        __builtin_amdgcn_sched_barrier(0);
        gate(1, t);                                                  // Y's gate math: the X half becomes visible meanwhile
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            publish(1, (t + 1) & 1);
            lds_inc(lds_cnt + 4, one);
        }
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
            for (int k = 0; k < 4; ++k) sv_st[tl][k] = sv[tl][k];
            h_st[tl] = h[tl]; c_st[tl] = c[tl];
        }
        off_t += st_h;
        if constexpr (FUSE) {
            dma_x(bo_nxt, xslot);
            xslot = xslot + 1 == XPD ? 0 : xslot + 1;
            bo_nxt = xo_row[min(t + XPD + 1, T - 1)] + bo_lane;
        } else load_x(t + 1);
        if (more) {
            __builtin_amdgcn_sched_barrier(0);
            read_half(0, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            group(0, 0, true); group(1, 0, true);                    // G1, G2 of step t + 1: the Y half becomes visible meanwhile
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#endif
    if (tmax > 0) store_step(off_t - st_h);
    for (int t = tmax; t < T; ++t) {                                 // past the tile's longest row: the state is carried
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            stf((char*)a.hs + off_t + st_h, bo_h[tl], h[tl]);
            if (CELL == CELL_LSTM) stf((char*)a.cs + off_t + st_h, bo_h[tl], c[tl]);
        }
        off_t += st_h;
    }
}

// ---------------------------------------------------------------------------------------
// backward: dh_{t-1}[row][unit] += sum_k dhi_t[row][k] W_hid[unit][k], k over the G*HP gate columns.  Same roles as the forward:
// the K halves are the dhi planes published by the X tiles (columns of units 0 .. 63 of every gate) and by the Y tiles; a step is
//     G4(n-1): Y x K_y  ||  gate math X(t) | publish X | read K_x | gate math Y(t) | publish Y | G1: X x K_x | G2: Y x K_x | read K_y
//     | G3: X x K_y -> dh_X(t-1) | ...
// WTM 0: saved activations one step ahead in registers, plain stores;  1: through the LDS ring (LDS-DMA), write-through stores,
// progress words for the overlapped step tail (rec_bwd_x6p: same protocol; a wave speaks for the two progress slots w and w + 4,
// the consumers fold eight words per workgroup).  Chunked BPTT (t_lo / t_hi / state / part) as in rec_bwd_x6p.
// ---------------------------------------------------------------------------------------
constexpr float F16_DSCALE_R = 512.0f;
template <int CELL, int WTM>
__global__ void __launch_bounds__(256) rec_bwd_x6r(RecArgs a) {
    constexpr bool WT = WTM == 1, RING = WTM != 0;
    constexpr int G = Gates<CELL>::G, GHP = G * HP, KBT = GHP / 32, KU = HP / 32, NH = KBT / 2;     // NH k-blocks per K half
    constexpr int DROW = GHP * 2 + 32, PLANEB = R * DROW, BUFB = 2 * PLANEB;
    extern __shared__ __attribute__((aligned(16))) char smem_r[];
    char* dbuf = smem_r;
    int* cnt = (int*)(dbuf + 2 * BUFB);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * R + q;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
    const int one = 1;
    const unsigned lds_cnt = (unsigned)(size_t)cnt;
    int u[2];
    u[0] = wave * 16 + j; u[1] = 64 + wave * 16 + j;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 16));
    tmax = __builtin_amdgcn_readfirstlane(max(tmax, __shfl_xor(tmax, 32)));

    // k-blocks of a K half, in the order they are visited: half 0 = columns of units 0 .. 63 of every gate, half 1 = units 64 .. 127
    auto kb_of = [](int half, int i) { return (i / 2) * KU + (i % 2) + half * 2; };
    f16x8 W1[2][KBT], W2[2][KBT];                        // B operands: lane (j, q) holds W_hid[unit u][kb*32 + 8q + e]
#pragma unroll
    for (int tl = 0; tl < 2; ++tl)
#pragma unroll
        for (int kb = 0; kb < KBT; ++kb) {
            const float* src = a.Whid + (size_t)u[tl] * GHP + kb * 32 + 8 * q;
            const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float w = e < 4 ? lo[e & 3] : hi[e & 3];
                asm("" : "+v"(w));
                _Float16 b1, b2;
                split2(w, b1, b2);
                W1[tl][kb][e] = b1; W2[tl][kb][e] = b2;
            }
        }
    unsigned bo_h[2], bo_g[2], bo_x[2], lds_pub[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        bo_h[tl] = (unsigned)(row * HP + u[tl]) * 4u;
        bo_g[tl] = (unsigned)sbr_blocked_index(0, row, u[tl], Bp, HP) * 4u;
        bo_x[tl] = (unsigned)(row * GHP + u[tl]) * 4u;
        lds_pub[tl] = (unsigned)(q * DROW + u[tl] * 2);
    }
    const size_t st_h = (size_t)Bp * HP * 4, st_x = (size_t)Bp * GHP * 4;
    const unsigned lds_rd = (unsigned)((j >> 2) * DROW + q * 16 + (j & 1) * PLANEB);

    const bool first = a.t_hi >= T, last = a.t_lo <= 0;
    float dh[2] = {0.f, 0.f}, dc[2] = {0.f, 0.f};
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        if (first) {
            if (a.n_dh_slabs) {                                       // dh_last arrives as unreduced split-K slabs of the dh GEMM
                const float* p = a.dh_slabs + (size_t)row * HP + u[tl];
                const size_t st = (size_t)Bp * HP;
                int z = 0;
                for (; z + 8 <= a.n_dh_slabs; z += 8) {
                    float v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(z + k) * st];
#pragma unroll
                    for (int k = 0; k < 8; ++k) dh[tl] += v[k];
                }
                for (; z < a.n_dh_slabs; ++z) dh[tl] += p[(size_t)z * st];
            } else if (a.dh_last) dh[tl] = a.dh_last[(size_t)row * HP + u[tl]];
        } else {
            dh[tl] = a.state[(size_t)row * HP + u[tl]];
            if (CELL == CELL_LSTM) dc[tl] = a.state[(size_t)Bp * HP + (size_t)row * HP + u[tl]];
        }
    }
    const char* const sbase = CELL == CELL_LSTM ? (const char*)a.cs : (const char*)a.hs;
    float pi[2] = {0.f, 0.f}, pf[2] = {0.f, 0.f}, po[2] = {0.f, 0.f}, sdp[2][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}}, sdb[2][G];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        if (CELL == CELL_LSTM) { pi[tl] = a.peep[u[tl]]; pf[tl] = a.peep[HP + u[tl]]; po[tl] = a.peep[2 * HP + u[tl]]; }
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[tl][g] = 0.f;
    }
    float sv[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, hprev[2] = {0.f, 0.f}, hnew[2] = {0.f, 0.f};

    // the ring of saved activations (RING): per wave, stage and tile NL arrays x 256 bytes; two 16-byte-per-lane pieces per tile and
    // step (hs / cs row pieces from 16 lanes, the four tile-blocked gate arrays from 64), PD steps ahead (rec_bwd_x6p)
    constexpr int PD = 4, NL = CELL == CELL_VANILLA ? 1 : 5, NST = G + (CELL == CELL_GRU ? 1 : 0);
    constexpr int NLI = CELL == CELL_VANILLA ? 1 : 2;
    constexpr int STG = NL * 256, RING_OFF = (2 * BUFB + 64 + 255) & ~255;
    constexpr int NLW = 2 * NLI, NSW = 2 * NST;                    // load / store instructions of one iteration (both tiles)
    constexpr int VMN = (PD - 1) * (NLW + NSW);                    // younger than the loads of the step being taken (stores come behind take_saved)
    const unsigned ring_wave = (unsigned)(size_t)(smem_r + RING_OFF) + (unsigned)wave * (PD * 2 * STG);
    const char* ring_lane = smem_r + RING_OFF + wave * (PD * 2 * STG) + lane * 4;
    unsigned bo_ga[2] = {0, 0}, bo_hs4[2];
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int ubase = tl * 64 + wave * 16;
        bo_hs4[tl] = (unsigned)((blockIdx.x * R + ((lane >> 2) & 3)) * HP + ubase + (lane & 3) * 4) * 4u;
        if (CELL != CELL_VANILLA) {
            const unsigned b0 = (unsigned)sbr_blocked_index(0, blockIdx.x * R, ubase, Bp, HP) * 4u;
            const int k = lane >> 4;
            const char* gk = k == 0 ? (const char*)a.g[0] : k == 1 ? (const char*)a.g[1] : k == 2 ? (const char*)a.g[2] : (const char*)a.g[3];
            bo_ga[tl] = b0 + (unsigned)(size_t)(gk - sbase) + (unsigned)(lane & 15) * 16u;
        }
    }
    auto dma_saved = [&](size_t o, int slot) {
        const char* base = sbase + o;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            const unsigned m = ring_wave + (unsigned)slot * (2 * STG) + tl * STG;
            lds_dma_x4(m, base, bo_hs4[tl], 0xFFFFull);
            if (CELL != CELL_VANILLA) lds_dma_x4(m + 256, base, bo_ga[tl], ~0ull);
        }
    };
    auto take_saved = [&](int slot) {
        wait_vm<VMN>();
        const char* p = ring_lane + slot * (2 * STG);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            hprev[tl] = *(const float*)(p + tl * STG);
            if (CELL != CELL_VANILLA) {
                sv[tl][0] = *(const float*)(p + tl * STG + 256); sv[tl][1] = *(const float*)(p + tl * STG + 512);
                sv[tl][2] = *(const float*)(p + tl * STG + 768); sv[tl][3] = *(const float*)(p + tl * STG + 1024);
            }
        }
    };
    auto load_saved = [&](size_t o) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            hprev[tl] = ldf(sbase + o, bo_h[tl]);
            if (CELL != CELL_VANILLA) {
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[tl][k] = ldf((const char*)a.g[k] + o, bo_g[tl]);
            }
        }
    };
    unsigned long long wt_c0 = 0, wt_r0 = 0;
    if (WT) { wt_c0 = clock64(); wt_r0 = wall_clock64(); }
    __syncthreads();

    const int t_live = min(a.t_hi, tmax);                         // steps [t_live, t_hi) are masked for the whole tile: zero rows
    for (int t = a.t_hi - 1; t >= max(t_live, a.t_lo); --t) {
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float* p = a.dxt + ((size_t)t * Bp + row) * GHP + g * HP + u[tl];
                if (WT) st_wt(p, 0.f); else *p = 0.f;
            }
            if (CELL == CELL_GRU) {
                float* p = a.dhi + ((size_t)t * Bp + row) * HP + u[tl];
                if (WT) st_wt(p, 0.f); else *p = 0.f;
            }
        }
    }
    int* prog_slot = nullptr; int prog_next = 0; const int prog_tag = a.prog_epoch << 12;
    // this wave's two progress words (slots w and w + 4 of the workgroup's eight): lanes 0 and 1 store them
    auto publish2 = [&](int word) {
        int* slot = prog_slot + (lane & 1) * 4;
        asm volatile("s_mov_b64 exec, 3\n\tglobal_store_dword %0, %1, off sc1\n\ts_mov_b64 exec, -1" :: "v"(slot), "v"(word) : "memory");
    };
    if constexpr (WT) {
        prog_slot = a.progress + blockIdx.x * 8 + wave;
        const int tl0 = max(t_live, a.t_lo);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish2(prog_tag | tl0);
        prog_next = tl0 - a.prog_every;
    }
    if (t_live > a.t_lo) {
        if constexpr (RING) {
#pragma unroll
            for (int d = 0; d < PD; ++d) dma_saved((size_t)max(t_live - 1 - d, a.t_lo) * st_h, d);
            wait_vm<0>();
            take_saved(0);
        } else load_saved((size_t)(t_live - 1) * st_h);
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            if (CELL == CELL_VANILLA) hnew[tl] = a.hs[(size_t)t_live * Bp * HP + (size_t)row * HP + u[tl]];
            if (CELL == CELL_LSTM) hnew[tl] = a.cs[(size_t)t_live * Bp * HP + (size_t)row * HP + u[tl]];
        }
    }
    size_t off_h = (size_t)(t_live - 1) * st_h, off_x = (size_t)(t_live - 1) * st_x;      // of step t

    f16x8 dp[2][NH];                                               // operand k-blocks of the two K halves
    auto read_half = [&](int half, int n) {
        const char* db = dbuf + (n & 1) * BUFB + lds_rd;
        int fl = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < NH; ++i) dp[half][i] = *(const f16x8*)(db + kb_of(half, i) * 64);
        if (__builtin_amdgcn_readfirstlane(fl) < 4 * (n + 1)) {
            int spins = 0;
#pragma clang loop unroll(disable)
            do {
                asm volatile("" ::: "memory");
                fl = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < NH; ++i) dp[half][i] = *(const f16x8*)(db + kb_of(half, i) * 64);
                if (++spins > X6P_SPIN_LIMIT) { atomicOr(a.fault, 2); break; }
            } while (__builtin_amdgcn_readfirstlane(fl) < 4 * (n + 1));
        }
    };
    f32x4 acc[2][2];                                               // per tile: [0] d . w1 (element 0: d1 w1, element 1: d2 w1), [1] d . w2
    auto group = [&](int tl, int half, bool firstg) {
#pragma unroll
        for (int i = 0; i < NH; ++i) {
            const int kb = kb_of(half, i);
            acc[tl][1] = mf(dp[half][i], W2[tl][kb], (firstg && i == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[tl][1]);
            acc[tl][0] = mf(dp[half][i], W1[tl][kb], (firstg && i == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[tl][0]);
        }
    };
    auto fold = [&](int tl) {                                      // dh of the tile for the step below
        dh[tl] += fmaf(acc[tl][0][1] + acc[tl][1][0], 1.0f / F16_LO, acc[tl][0][0]) * (1.0f / F16_DSCALE_R);
    };
    float dxi[2][G], dhi[2][G];
    auto gate = [&](int tl, int t, int n) {                        // gate math of step t for the tile, publication of its dhi planes
        float dpl[3] = {0.f, 0.f, 0.f};
        if (CELL == CELL_LSTM) {
            cell_backward<CELL, true>(t < mylen, clip, dh[tl], dc[tl], sv[tl], 0.f, hprev[tl], hnew[tl], 0.f, pi[tl], pf[tl], po[tl],
                                      dxi[tl], dhi[tl], dpl, false);
            sdp[tl][0] += dpl[0]; sdp[tl][1] += dpl[1]; sdp[tl][2] += dpl[2];
        } else
            cell_backward<CELL, true>(t < mylen, clip, dh[tl], dc[tl], sv[tl], hprev[tl], 0.f, 0.f, hnew[tl], 0.f, 0.f, 0.f, dxi[tl], dhi[tl],
                                      dpl, a.relu != 0);
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[tl][g] += dxi[tl][g];
        char* lds = dbuf + (n & 1) * BUFB + lds_pub[tl];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float d = dhi[tl][g] * F16_DSCALE_R;                   // |dhi| <= clip <= 100: below fp16's 65504
            asm volatile("" : "+v"(d));
            _Float16 d1, d2;
            split2(d, d1, d2);
            *(_Float16*)(lds + g * HP * 2) = d1;
            *(_Float16*)(lds + g * HP * 2 + PLANEB) = d2;
        }
        if (CELL != CELL_GRU) hnew[tl] = hprev[tl];
    };
    auto store_step = [&](size_t ox, size_t oh) {                  // dxt (and GRU's compact candidate slice of dhi) of both tiles
        const char* dx_t = (const char*)a.dxt + ox;
#pragma unroll
        for (int tl = 0; tl < 2; ++tl) {
            st_si<0, WT>(dx_t, bo_x[tl], dxi[tl][0]);
            if (G > 1) st_si<HP * 4, WT>(dx_t, bo_x[tl], dxi[tl][G > 1 ? 1 : 0]);
            if (G > 2) st_si<2 * HP * 4, WT>(dx_t, bo_x[tl], dxi[tl][G > 2 ? 2 : 0]);
            if (G > 3) st_si<3 * HP * 4, WT>(dx_t, bo_x[tl], dxi[tl][G > 3 ? 3 : 0]);
            if (CELL == CELL_GRU) st_si<0, WT>((const char*)a.dhi + oh, bo_h[tl], dhi[tl][G - 1]);
        }
    };

    int n = 0, slot = 0;
    for (int t = t_live - 1; t >= a.t_lo; --t, ++n) {
        // ---- gate math of step t: tile X beside G4 of the step above (its MFMAs were issued at the bottom of the last iteration),
        // then tile Y, whose dh that group completes
        if constexpr (WT) {
            // every operation but the youngest NLW + NSW (the previous iteration's loads and stores) has been waited for: step t + 2
            // is complete and written through
            if (n >= 1 && t + 2 <= prog_next) {
                wait_vm<NLW + NSW>();
                publish2(prog_tag | (t + 2));
                prog_next = t + 2 - a.prog_every;
            }
        }
        gate(0, t, n);
        lds_inc(lds_cnt, one);
        __builtin_amdgcn_sched_barrier(0);
        if (n > 0) fold(1);                                        // G4 of the step above: dh of tile Y is complete
        gate(1, t, n);
        lds_inc(lds_cnt + 4, one);
        __builtin_amdgcn_sched_barrier(0);
        // ---- loads for step t - 1, stores of step t
        if constexpr (RING) {
            dma_saved(t - PD >= a.t_lo ? off_h - (size_t)PD * st_h : (size_t)a.t_lo * st_h, slot);
            slot = slot + 1 == PD ? 0 : slot + 1;
            __builtin_amdgcn_sched_barrier(0);
            take_saved(slot);
            __builtin_amdgcn_sched_barrier(0);
            store_step(off_x, off_h);
        } else {
            store_step(off_x, off_h);
            __builtin_amdgcn_sched_barrier(0);
            load_saved(t > a.t_lo ? off_h - st_h : off_h);
        }
        off_h -= st_h; off_x -= st_x;
        __builtin_amdgcn_sched_barrier(0);
        // ---- the products of step t: X and Y over the X half, X over the Y half (-> dh of tile X), Y over the Y half (-> tile Y,
        // folded at the top of the next iteration / behind the loop)
        read_half(0, n);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 0, true); group(1, 0, true);
        __builtin_amdgcn_sched_barrier(0);
        read_half(1, n);
        __builtin_amdgcn_sched_barrier(0);
        group(0, 1, false);
        __builtin_amdgcn_sched_barrier(0);
        group(1, 1, false);
        fold(0);
    }
    if (n > 0) fold(1);
    if constexpr (WT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish2(prog_tag | a.t_lo);
        if (blockIdx.x == 0 && wave == 0 && lane == 0) {
            unsigned long long* cc = (unsigned long long*)(a.progress + gridDim.x * 8 + 128);
            cc[0] = clock64() - wt_c0; cc[1] = wall_clock64() - wt_r0;
        }
    } else if constexpr (RING) wait_vm<0>();

    float* part = a.part + ((size_t)a.chunk * gridDim.x + blockIdx.x) * (GHP + 5 * HP);
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        if (!last) {
            a.state[(size_t)row * HP + u[tl]] = dh[tl];
            if (CELL == CELL_LSTM) a.state[(size_t)Bp * HP + (size_t)row * HP + u[tl]] = dc[tl];
        }
        float v[G + 5];
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = sdb[tl][g];
        v[G] = sdp[tl][0]; v[G + 1] = sdp[tl][1]; v[G + 2] = sdp[tl][2];
        v[G + 3] = CELL == CELL_LSTM && last ? dc[tl] : 0.f;
        v[G + 4] = last ? dh[tl] : 0.f;
#pragma unroll
        for (int k = 0; k < G + 5; ++k) {
            float sum = v[k];
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            v[k] = sum;
        }
        if (q == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) part[g * HP + u[tl]] = v[g];
#pragma unroll
            for (int k = 0; k < 5; ++k) part[GHP + k * HP + u[tl]] = v[G + k];
        }
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool sbr_rec_x6r_fwd_ok(const RecArgs& a) {
    const char* e = getenv("SBR_X6R");                               // read per launch: the tests flip it
    if (e && atoi(e) == 0) return false;
    return !a.prof && a.Hp == HP && a.rpt == R;
}

template <int CELL>
static hipError_t launch_fwd_r(hipStream_t s, const RecArgs& a) {
    constexpr int G = Gates<CELL>::G;
    size_t lds = 2 * 2 * R * (size_t)(HP * 2 + 32) + 64;
    if (a.gX) {
        lds = ((lds + 255) & ~(size_t)255) + (size_t)R * a.T * 4;
        lds = ((lds + 255) & ~(size_t)255) + (size_t)4 * 4 * 2 * G * 256;      // 4 waves x XPD stages x two tiles
    }
    const int nb = a.Bp / R;
    if (a.gX) {
        (void)hipFuncSetAttribute((const void*)rec_fwd_x6r<CELL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        rec_fwd_x6r<CELL, true><<<nb, 256, lds, s>>>(a);
    } else {
        (void)hipFuncSetAttribute((const void*)rec_fwd_x6r<CELL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        rec_fwd_x6r<CELL, false><<<nb, 256, lds, s>>>(a);
    }
    return hipGetLastError();
}

hipError_t launch_rec_forward_x6r(hipStream_t s, const RecArgs& a) {
    return a.cell == SBR_CELL_GRU ? launch_fwd_r<CELL_GRU>(s, a) : a.cell == SBR_CELL_LSTM ? launch_fwd_r<CELL_LSTM>(s, a)
                                                                 : launch_fwd_r<CELL_VANILLA>(s, a);
}

bool sbr_rec_x6r_bwd_ok(const RecArgs& a) {
    const char* e = getenv("SBR_X6R_BWD");                           // read per launch: the tests flip it
    if (e && atoi(e) == 0) return false;
    return !a.prof && !a.dh_ext && a.Hp == HP && a.rpt == R;
}

template <int CELL>
static hipError_t launch_bwd_r(hipStream_t s, const RecArgs& a) {
    constexpr int G = Gates<CELL>::G;
    size_t lds = 2 * 2 * R * (size_t)(G * HP * 2 + 32) + 64;
    lds = ((lds + 255) & ~(size_t)255) + 4 * 4 * 2 * (size_t)(G == 1 ? 1 : 5) * 256;      // + the ring (4 waves x PD stages x two tiles)
    const int nb = a.Bp / R;
    if (a.progress) {
        (void)hipFuncSetAttribute((const void*)rec_bwd_x6r<CELL, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        rec_bwd_x6r<CELL, 1><<<nb, 256, lds, s>>>(a);
    } else {
        (void)hipFuncSetAttribute((const void*)rec_bwd_x6r<CELL, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        rec_bwd_x6r<CELL, 0><<<nb, 256, lds, s>>>(a);
    }
    return hipGetLastError();
}

hipError_t launch_rec_backward_x6r(hipStream_t s, const RecArgs& a) {
    return a.cell == SBR_CELL_GRU ? launch_bwd_r<CELL_GRU>(s, a) : a.cell == SBR_CELL_LSTM ? launch_bwd_r<CELL_LSTM>(s, a)
                                                                 : launch_bwd_r<CELL_VANILLA>(s, a);
}
