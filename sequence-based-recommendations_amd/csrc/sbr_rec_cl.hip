// Cluster recurrent kernels for wide layers (Hp >= 256) on gfx950: W_hid no longer fits one CU
// (LSTM-256: 1.5 MB of bf16 planes against 512 KB of VGPRs + 160 KB of LDS), so a CLUSTER of C = Hp/32
// workgroups owns one tile of R batch rows for all T steps, each workgroup keeping the W_hid slice of its
// 32 hidden units resident (planes 1,2 in VGPRs, plane 3 in LDS) exactly as the single-CU bf16x6 kernels do.
//
// What crosses workgroups per step is what the kernels write to HBM anyway:
//   forward : h_t            -> the hs array      (every member needs all Hp values of h_{t-1})
//   backward: dhi_t (= dxt_t, GRU: + the compact candidate slice) -> the dxt / dhc arrays
// so the exchange costs no extra traffic.  It is synchronised by the data itself: the launcher fills the
// array with a NaN sentinel (0xFFFFFFFF), producers publish with agent-scope write-through stores
// (global_store ... sc1), consumers poll the exact 8-byte pieces they need with agent-scope loads until no
// sentinel is left.  No flags, no fences, no grid barrier: one L2 round trip per step.
//
// Placement: workgroup ids are dispatched round-robin over the 8 XCDs, so the members of a cluster are the
// ids {8*(grp*C + m) + x, m = 0..C-1}: same XCD, same L2, and a group of 8 clusters is contiguous in
// dispatch order (no deadlock when the grid exceeds the resident capacity).  Correctness does not depend
// on that placement (agent-scope accesses are coherent across XCDs), only the latency does.
//
// Inside a workgroup: 4 waves = 2 unit tiles x 2 K-halves; the K-half partial accumulators of the upper wave
// are added through LDS by the lower wave, which also runs the gate math for its 16 units.
// Cell math: sbr_cell.h (sparse_lstm.py:377-425, :764-805, :1120-1152); BPTT pinned by oracle/rnn_oracle.py.
#include "sbr_rec_cl.h"

// bool switches of the two chains, read per launch (the tests flip them); the same conditions as the 128-unit kernels
static bool cl_f16_fwd(const RecArgs& a) {
    const char* fe = getenv("SBR_X6_F16");
    return (fe ? atoi(fe) != 0 : true) && !a.relu;
}
static bool cl_f16_bwd(const RecArgs& a) {
    const char* fe = getenv("SBR_X6_F16_BWD");
    return (fe ? atoi(fe) != 0 : true) && a.clip > 0.0f && a.clip <= 100.0f;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------

// With R = 8 live rows the MFMA columns 8..15 hold a duplicate of rows 0..7 (the B operand repeats them), so the
// accumulators of lane (j, q) and lane (j + 8, q) are identical: the lower lane finishes units 0,1 of its four,
// the upper lane units 2,3 -- every lane does useful gate math and the per-step VALU chain is halved.
template <int CELL, int HP, int R, bool F16>
__global__ void __launch_bounds__(256) rec_fwd_cl(RecArgs a) {
    static_assert(R == 8, "the element split over duplicate MFMA columns assumes 8 live rows");
    using OPV = std::conditional_t<F16, f16x8c, bf16x8>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int G = Gates<CELL>::G, C = HP / 32, KH = HP / 2, KBW = KH / 32, GHP = G * HP;
    constexpr int HROW = HP * 2 + 32, PLANEB = R * HROW;
    constexpr int W3_BYTES = F16 ? 0 : G * KBW * 4 * 1024;
    constexpr int NP = R * HP / 4 / 256;
    static_assert(R * HP / 4 % 256 == 0, "piece count");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [G][KBW][4 waves][64 lanes][16 B]
    char* hpl = smem_c + W3_BYTES;                       // [3 (F16: 2) planes][R rows][HROW]
    char* red = hpl + NPL * PLANEB;                      // [2 unit tiles][G][64 lanes][16 B]
    int tile, mem;
    if (!cl_ids(a, C, a.Bp / R, tile, mem)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = wave & 1, kh = wave >> 1;
    const int j = lane & 15, q = lane >> 4;
    const int rl = j & 7, eh = j >> 3;                   // tile-local row; which half of the lane's 4 units it finishes
    const int row = tile * R + rl;
    const int T = a.T, Bp = a.Bp;
    const int ub = mem * 32 + ut * 16;                   // first unit of this wave's tile
    const int u2 = ub + q * 4 + eh * 2;                  // the 2 units this lane finishes
    const int k0 = kh * KH;
    const bool fin = kh == 0;                            // this wave finishes (reduces, gate math, stores)
    bool dead = false;
    const bool fast = cl_same_xcc(a, C, tile, mem, (int*)red, dead);

    const int mylen = a.len[row];
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    // A operand planes: lane (unit j of the tile, k-group q) holds W_hid[k0 + kb*32 + 8q + e][g*HP + ub + j]
    OPV W1[G][KBW], W2[G][KBW];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            if constexpr (F16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 b1, b2;
                    cl_split2(a.Whid[(size_t)(k0 + kb * 32 + 8 * q + e) * GHP + g * HP + ub + j], b1, b2);
                    W1[g][kb][e] = b1; W2[g][kb][e] = b2;
                }
            } else {
            bf16x8 w3v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 b1, b2, b3;
                split3(a.Whid[(size_t)(k0 + kb * 32 + 8 * q + e) * GHP + g * HP + ub + j], b1, b2, b3);
                W1[g][kb][e] = b1; W2[g][kb][e] = b2; w3v[e] = b3;
            }
            *(bf16x8*)(w3 + ((g * KBW + kb) * 4 + wave) * 1024 + lane * 16) = w3v;
            }
        }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const f32x2 z2 = f32x2{0, 0};
    const f32x4 z = f32x4{0, 0, 0, 0};
    f32x2 h = z2, c = z2, pi = z2, pf = z2, po = z2;
    if (fin) {
        h = *(const f32x2*)&a.hinit[u2];
        if (CELL == CELL_LSTM) {
            c = *(const f32x2*)&a.cinit[u2];
            pi = *(const f32x2*)&a.peep[u2]; pf = *(const f32x2*)&a.peep[HP + u2]; po = *(const f32x2*)&a.peep[2 * HP + u2];
            *(f32x2*)&a.cs[(size_t)row * HP + u2] = c;
        }
        cl_store2(&a.hs[(size_t)row * HP + u2], h, fast);
    }
    // Input of step t: a row of xt, or (layer 0, one index per step) W_in[id[row][t]] + b gathered here, id two steps
    // ahead, row one step ahead (see rec_fwd_x6s)
    const bool fuse = a.gX != nullptr;
    f32x2 x[G], xn[G], bias[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bias[g] = (fuse && fin) ? *(const f32x2*)&a.gbias[g * HP + u2] : z2;
    auto load_id = [&](int t) -> int { return fuse ? a.gX[(size_t)row * T + (t < T ? t : T - 1)] : 0; };
    auto load_x = [&](int t, int id, f32x2 (&d)[G]) {
        const float* src = fuse ? a.gWin + (size_t)id * GHP + u2 : a.xt + ((size_t)(t < T ? t : T - 1) * Bp + row) * GHP + u2;
#pragma unroll
        for (int g = 0; g < G; ++g) d[g] = *(const f32x2*)&src[g * HP];
    };
    int id_next = 0, id_nn = 0;
    if (fin) { id_next = load_id(1); load_x(0, load_id(0), x); }
    __syncthreads();                                     // W plane 3 visible
    // SBR_FLAG_PROFILE_REC: cycles per phase (exchange wait | publish + barrier | LDS reads + MFMA | reduce
    // barrier | gate math + stores), tools/cl_prof.py
    u64 pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_c0 = 0, p_r0 = 0, p_t = 0, p_tries = 0;
    const bool prof = a.prof != nullptr;
    if (prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }

    // the per-step GEMM of this wave: acc[g] = sum over its K half of W_hid^T . h_{t-1}  (B operand from the LDS planes)
    auto mfma_phase = [&](f32x4 (&acc)[G]) {
        const char* hb = hpl + rl * HROW + k0 * 2 + q * 16;
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = z;
        if constexpr (F16) {      // acc: a1 w1; lo: the low-order products a2 w1 + a1 w2 (/ 2048 at the end)
            f32x4 lo[G];
#pragma unroll
            for (int g = 0; g < G; ++g) lo[g] = z;
            OPV hp[2][2];
            auto load_ops = [&](int kb, int s) {
                hp[s][0] = *(const OPV*)(hb + kb * 64);
                hp[s][1] = *(const OPV*)(hb + kb * 64 + PLANEB);
            };
            load_ops(0, 0);
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
                const int s = kb & 1;
                if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = cl_mfma(W1[g][kb], hp[s][1], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = cl_mfma(W2[g][kb], hp[s][0], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = cl_mfma(W1[g][kb], hp[s][0], acc[g]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 15");
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] += lo[g] * (1.0f / CL_F16_LO);
        } else {
        bf16x8 hp[2][3], wp[2][G];
        auto load_ops = [&](int kb, int s) {
            hp[s][0] = *(const bf16x8*)(hb + kb * 64);
            hp[s][1] = *(const bf16x8*)(hb + kb * 64 + PLANEB);
            hp[s][2] = *(const bf16x8*)(hb + kb * 64 + 2 * PLANEB);
#pragma unroll
            for (int g = 0; g < G; ++g) wp[s][g] = *(const bf16x8*)(w3 + ((g * KBW + kb) * 4 + wave) * 1024 + lane * 16);
        };
        load_ops(0, 0);
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(wp[s][g], hp[s][0], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W1[g][kb], hp[s][2], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W2[g][kb], hp[s][1], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W2[g][kb], hp[s][0], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W1[g][kb], hp[s][1], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W1[g][kb], hp[s][0], acc[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15");                        // MFMA D -> VALU read hazard (the reader may sit behind a branch)
        }
    };
    // h_{t-1} of all Hp units (slot t of hs, written by the C members of the cluster) -> bf16 planes in LDS
    auto exchange = [&](int t) {
        f32x4 v[NP];
        const float* base = a.hs + ((size_t)t * Bp + (size_t)tile * R) * HP;
        p_tries += cl_fetch<NP>(v, [&](int r, int col) { return base + (size_t)r * HP + col; }, HP, fast, dead, a.fault);
        CL_TICK(0);
        cl_publish<NP, F16>(v, hpl, HP, HROW, PLANEB);
    };
    if (prof) p_t = clock64();

    // Two roles, two loops with the same barrier sequence (2 per live step).  Separate loops keep the finishing
    // waves' prefetched xt registers free of phi copies (a copy at a block end makes hipcc wait for the loads).
    if (fin) {
        for (int t = 0; t < tmax; ++t) {                 // tmax is uniform over the whole cluster (same rows)
            load_x(t + 1, id_next, xn);                  // unconditional (clamped): see rec_bwd_cl
            id_nn = load_id(t + 2);
            exchange(t);
            __syncthreads();
            CL_TICK(1);
            f32x4 acc[G];
            mfma_phase(acc);
            CL_TICK(2);
            __syncthreads();                             // partials visible; every wave is done reading hpl
            CL_TICK(3);
            f32x2 as2[G], sv[4];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x2 part = *(const f32x2*)(red + ((ut * G + g) * 64 + lane) * 16 + eh * 8);
                as2[g] = (eh ? f32x2{acc[g][2], acc[g][3]} : f32x2{acc[g][0], acc[g][1]}) + part;
            }
            const bool m = t < mylen;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float xs[G], as[G], s[4];
#pragma unroll
                for (int g = 0; g < G; ++g) { xs[g] = x[g][e] + bias[g][e]; as[g] = as2[g][e]; }
                float hh = h[e], cc = c[e];
                cell_forward<CELL, true>(xs, as, m, hh, cc, pi[e], pf[e], po[e], s, a.relu != 0);
                h[e] = hh; c[e] = cc;
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[k][e] = s[k];
            }
            // x <- xn BEFORE this step's stores are issued: vmcnt retires in order, so a copy placed after them
            // would also wait for their acknowledgements (~1000 cycles measured)
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = xn[g];
            id_next = id_nn;
            __builtin_amdgcn_sched_barrier(0);
            const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u2;
            cl_store2(&a.hs[o], h, fast);                // first: the other members are waiting for it
            if (CELL == CELL_LSTM) *(f32x2*)&a.cs[o] = c;
            if (CELL != CELL_VANILLA) {
                const size_t og = sbr_blocked_index(t, row, u2, Bp, HP);
#pragma unroll
                for (int k = 0; k < 4; ++k) *(f32x2*)&a.g[k][og] = sv[k];
            }
            CL_TICK(4);
        }
        for (int t = tmax; t < T; ++t) {                 // past the tile's longest row: the state is carried
            const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u2;
            *(f32x2*)&a.hs[o] = h;
            if (CELL == CELL_LSTM) *(f32x2*)&a.cs[o] = c;
        }
    } else {
        for (int t = 0; t < tmax; ++t) {
            exchange(t);
            __syncthreads();
            CL_TICK(1);
            f32x4 acc[G];
            mfma_phase(acc);
#pragma unroll
            for (int g = 0; g < G; ++g) *(f32x4*)(red + ((ut * G + g) * 64 + lane) * 16) = acc[g];
            CL_TICK(2);
            __syncthreads();
            CL_TICK(3);
        }
    }
    if (prof && lane == 0 && tile < 4) {
        u64* o = a.prof + (((size_t)tile * C + mem) * 4 + wave) * 16;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_tries;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[3 + i] = pc[i];
    }
}

// ---------------------------------------------------------------------------------------
// backward (BPTT).  D[unit k][row] = sum_col W_hid[k][col] * dhi[row][col] over ALL G*Hp columns: each member
// computes the rows of W_hid it owns (its 32 units) and needs the whole dhi_t row tile from the cluster.
// ---------------------------------------------------------------------------------------
template <int CELL, int HP, int R, bool F16>
__global__ void __launch_bounds__(256) rec_bwd_cl(RecArgs a) {
    static_assert(R == 8, "the element split over duplicate MFMA columns assumes 8 live rows");
    using OPV = std::conditional_t<F16, f16x8c, bf16x8>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int G = Gates<CELL>::G, C = HP / 32, GHP = G * HP, KH = GHP / 2, KBW = KH / 32;
    constexpr int DROW = GHP * 2 + 32, PLANEB = R * DROW;
    constexpr int W3_BYTES = F16 ? 0 : KBW * 4 * 1024;
    constexpr int NP = R * GHP / 4 / 256;
    static_assert(R * GHP / 4 % 256 == 0, "piece count");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [KBW][4 waves][64][16 B]
    char* dpl = smem_c + W3_BYTES;                       // [3 (F16: 2)][R][DROW]
    char* red = dpl + NPL * PLANEB;                      // [2 unit tiles][64 lanes][16 B]
    int tile, mem;
    if (!cl_ids(a, C, a.Bp / R, tile, mem)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = wave & 1, kh = wave >> 1;
    const int j = lane & 15, q = lane >> 4;
    const int rl = j & 7, eh = j >> 3;
    const int row = tile * R + rl;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    const int ub = mem * 32 + ut * 16;
    const int u2 = ub + q * 4 + eh * 2;
    const int k0 = kh * KH;
    const bool fin = kh == 0;
    bool dead = false;
    const bool fast = cl_same_xcc(a, C, tile, mem, (int*)red, dead);

    const int mylen = a.len[row];
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    // A operand planes: lane (unit j of the tile, k-group q) holds W_hid[ub + j][k0 + kb*32 + 8q + e]
    OPV W1[KBW], W2[KBW];
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
        const float* src = a.Whid + (size_t)(ub + j) * GHP + k0 + kb * 32 + 8 * q;
        const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
        if constexpr (F16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 b1, b2;
                cl_split2(e < 4 ? lo[e & 3] : hi[e & 3], b1, b2);
                W1[kb][e] = b1; W2[kb][e] = b2;
            }
        } else {
        bf16x8 w3v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 b1, b2, b3;
            split3(e < 4 ? lo[e & 3] : hi[e & 3], b1, b2, b3);
            W1[kb][e] = b1; W2[kb][e] = b2; w3v[e] = b3;
        }
        *(bf16x8*)(w3 + (kb * 4 + wave) * 1024 + lane * 16) = w3v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const f32x2 z2 = f32x2{0, 0};
    const f32x4 z4 = f32x4{0, 0, 0, 0};
    f32x2 dh = z2, dc = z2, pi = z2, pf = z2, po = z2;
    if (fin && a.dh_last) dh = *(const f32x2*)&a.dh_last[(size_t)row * HP + u2];
    if (CELL == CELL_LSTM) { pi = *(const f32x2*)&a.peep[u2]; pf = *(const f32x2*)&a.peep[HP + u2]; po = *(const f32x2*)&a.peep[2 * HP + u2]; }
    f32x2 sdb[G], sdp[3];
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = z2;
    sdp[0] = z2; sdp[1] = z2; sdp[2] = z2;

    f32x2 sv[4], hprev = z2, cprev = z2, cnew = z2, hnew = z2;
#pragma unroll
    for (int k = 0; k < 4; ++k) sv[k] = z2;
    auto load_saved = [&](int t) {
        const size_t o = ((size_t)t * Bp + row) * HP + u2;
        hprev = *(const f32x2*)&a.hs[o];
        if (CELL != CELL_VANILLA) {
            const size_t og = sbr_blocked_index(t, row, u2, Bp, HP);
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = *(const f32x2*)&a.g[k][og];
        }
        if (CELL == CELL_LSTM) cprev = *(const f32x2*)&a.cs[o];
    };
    __syncthreads();                                     // W plane 3 visible
    // cycles per phase: gate math + publish stores | exchange wait | split + barrier | LDS reads + MFMA | reduce barrier
    u64 pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_c0 = 0, p_r0 = 0, p_t = 0, p_tries = 0;
    const bool prof = a.prof != nullptr;
    if (prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }

    auto mfma_phase = [&]() -> f32x4 {                   // this wave's K half of dhi_t . W_hid^T for its 16 units
        const char* db = dpl + rl * DROW + k0 * 2 + q * 16;
        f32x4 acc[3] = {z4, z4, z4};
        if constexpr (F16) {      // acc[0]: d1 w1; acc[1], acc[2]: the low-order products (/ 2048), all / 2^9 (the operand's scale)
            OPV dp[2][2];
            auto load_ops = [&](int kb, int s) {
                dp[s][0] = *(const OPV*)(db + kb * 64);
                dp[s][1] = *(const OPV*)(db + kb * 64 + PLANEB);
            };
            load_ops(0, 0);
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
                const int s = kb & 1;
                if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = cl_mfma(W1[kb], dp[s][1], acc[1]);
                acc[2] = cl_mfma(W2[kb], dp[s][0], acc[2]);
                acc[0] = cl_mfma(W1[kb], dp[s][0], acc[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 15");
            return (acc[0] + (acc[1] + acc[2]) * (1.0f / CL_F16_LO)) * (1.0f / CL_F16_DSCALE);
        } else {
        bf16x8 dp[2][3], wp[2];
        auto load_ops = [&](int kb, int s) {
            dp[s][0] = *(const bf16x8*)(db + kb * 64);
            dp[s][1] = *(const bf16x8*)(db + kb * 64 + PLANEB);
            dp[s][2] = *(const bf16x8*)(db + kb * 64 + 2 * PLANEB);
            wp[s] = *(const bf16x8*)(w3 + (kb * 4 + wave) * 1024 + lane * 16);
        };
        load_ops(0, 0);
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = MFMA_BF16(wp[s], dp[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dp[s][2], acc[1]);
            acc[2] = MFMA_BF16(W2[kb], dp[s][1], acc[2]);
            acc[0] = MFMA_BF16(W2[kb], dp[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dp[s][1], acc[1]);
            acc[2] = MFMA_BF16(W1[kb], dp[s][0], acc[2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15");
        return acc[0] + acc[1] + acc[2];
        }
    };
    auto exchange = [&](int t) {                         // dhi_t of all G*Hp columns -> bf16 planes in LDS
        f32x4 v[NP];
        const float* bx = a.dxt + ((size_t)t * Bp + (size_t)tile * R) * GHP;
        const float* bc = a.dhi + ((size_t)t * Bp + (size_t)tile * R) * HP;
        p_tries += cl_fetch<NP>(v, [&](int r, int col) {
            return (CELL == CELL_GRU && col >= 2 * HP) ? bc + (size_t)r * HP + (col - 2 * HP) : bx + (size_t)r * GHP + col;
        }, GHP, fast, dead, a.fault);
        CL_TICK(1);
        cl_publish<NP, F16>(v, dpl, GHP, DROW, PLANEB, F16 ? CL_F16_DSCALE : 1.0f);
    };
    if (prof) p_t = clock64();

    // Two roles, two loops with the same barrier sequence (see rec_fwd_cl)
    if (fin) {
        for (int t = T - 1; t >= tmax; --t) {            // whole tile masked: zero rows, nobody waits for them
            if (a.dh_ext) dh += *(const f32x2*)&a.dh_ext[((size_t)t * Bp + row) * HP + u2];
#pragma unroll
            for (int g = 0; g < G; ++g) *(f32x2*)&a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u2] = z2;
            if (CELL == CELL_GRU) *(f32x2*)&a.dhi[((size_t)t * Bp + row) * HP + u2] = z2;
        }
        if (tmax > 0) {
            load_saved(tmax - 1);
            const size_t o1 = ((size_t)tmax * Bp + row) * HP + u2;
            if (CELL == CELL_LSTM) cnew = *(const f32x2*)&a.cs[o1];
            if (CELL == CELL_VANILLA) hnew = *(const f32x2*)&a.hs[o1];
        }
        for (int t = tmax - 1; t >= 0; --t) {
            if (a.dh_ext) dh += *(const f32x2*)&a.dh_ext[((size_t)t * Bp + row) * HP + u2];
            const bool m = t < mylen;
            f32x2 vxi[G], vhi[G];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float s[4] = {sv[0][e], sv[1][e], sv[2][e], sv[3][e]};
                float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
                float dhh = dh[e], dcc = dc[e];
                cell_backward<CELL, true>(m, clip, dhh, dcc, s, hprev[e], cprev[e], cnew[e], hnew[e], pi[e], pf[e], po[e], dxi, dhi, dp, a.relu != 0);
                dh[e] = dhh; dc[e] = dcc;
#pragma unroll
                for (int g = 0; g < G; ++g) { vxi[g][e] = dxi[g]; vhi[g][e] = dhi[g]; sdb[g][e] += dxi[g]; }
                sdp[0][e] += dp[0]; sdp[1][e] += dp[1]; sdp[2][e] += dp[2];
            }
            if (CELL == CELL_LSTM) cnew = cprev;
            if (CELL == CELL_VANILLA) hnew = hprev;
            // Saved activations of step t-1, in flight across the exchange and the MFMA phase.  Three things keep
            // them off the critical path (each measured): issued AFTER the gate math (sched_barrier: hipcc otherwise
            // hoists them above the wait for this step's values, which then covers them too), BEFORE this step's
            // stores (vmcnt retires in order: behind the stores they also wait for the store acknowledgements),
            // and unconditionally (clamped index: a branch here makes the waitcnt pass fall back to vmcnt(0)).
            __builtin_amdgcn_sched_barrier(0);
            load_saved(t > 0 ? t - 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
            // dhi == dxi except the GRU candidate gate: dxt doubles as the exchange array
#pragma unroll
            for (int g = 0; g < G; ++g) cl_store2(&a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u2], vxi[g], fast);
            if (CELL == CELL_GRU) cl_store2(&a.dhi[((size_t)t * Bp + row) * HP + u2], vhi[2], fast);
            CL_TICK(0);
            exchange(t);
            __syncthreads();
            CL_TICK(2);
            const f32x4 sum = mfma_phase();
            CL_TICK(3);
            __syncthreads();                             // partials visible; every wave is done reading dpl
            dh += (eh ? f32x2{sum[2], sum[3]} : f32x2{sum[0], sum[1]}) + *(const f32x2*)(red + (ut * 64 + lane) * 16 + eh * 8);
            CL_TICK(4);
        }
    } else {
        for (int t = tmax - 1; t >= 0; --t) {
            CL_TICK(0);
            exchange(t);
            __syncthreads();
            CL_TICK(2);
            const f32x4 sum = mfma_phase();
            *(f32x4*)(red + (ut * 64 + lane) * 16) = sum;
            CL_TICK(3);
            __syncthreads();
            CL_TICK(4);
        }
    }
    if (prof && lane == 0 && tile < 4) {
        u64* o = a.prof + (((size_t)tile * C + mem) * 4 + wave) * 16;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_tries;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[3 + i] = pc[i];
    }

    if (fin) {
        float* part = a.part + (size_t)tile * (GHP + 5 * HP);
        f32x2 v[G + 5];
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = sdb[g];
        v[G] = sdp[0]; v[G + 1] = sdp[1]; v[G + 2] = sdp[2];
        v[G + 3] = dc; v[G + 4] = dh;
#pragma unroll
        for (int k = 0; k < G + 5; ++k)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float sum = v[k][e];
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);     // over the 8 rows; lanes j and j+8 hold different units
                v[k][e] = sum;
            }
        if (rl == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) *(f32x2*)&part[g * HP + u2] = v[g];
#pragma unroll
            for (int k = 0; k < 5; ++k) *(f32x2*)&part[GHP + k * HP + u2] = v[G + k];
        }
    }
}

// ---------------------------------------------------------------------------------------
// General cluster kernels (Hp = 512): UT unit tiles per workgroup, K split KS ways over the waves (UT * KS = 4 waves),
// R live rows per tile (R = 8: lanes j, j+8 share a row and finish 2 units each; R = 4: four copies, 1 unit each).
// C = Hp / (16 * UT) members per cluster; with UT = 1 at Hp = 512 a cluster is 32 workgroups = one whole XCD.
// Same exchange protocol, same layouts as rec_fwd_cl / rec_bwd_cl above (which stay as the tuned Hp = 256 instances).
// Sizing at LSTM-512: forward (UT 1, KS 4, R 8): W planes 1,2 = 128 VGPRs per wave, plane 3 = 64 KB LDS, h planes 25 KB;
// backward (UT 1, KS 4, R 4): K = 2048 columns, 128 VGPRs, 64 KB, dhi planes 49.5 KB (8 rows would need 99 KB).
// ---------------------------------------------------------------------------------------
template <int N> struct VecN { float v[N]; };
template <int N> __device__ __forceinline__ VecN<N> ldn(const float* p) {
    VecN<N> r;
    if (N == 2) { const f32x2 t = *(const f32x2*)p; r.v[0] = t[0]; r.v[N - 1] = t[1]; }
    else r.v[0] = p[0];
    return r;
}
template <int N> __device__ __forceinline__ void stn(float* p, const VecN<N>& x) {
    if (N == 2) *(f32x2*)p = f32x2{x.v[0], x.v[N - 1]};
    else p[0] = x.v[0];
}
template <int N> __device__ __forceinline__ void cl_storen(float* p, const VecN<N>& x, bool fast) {
    if (N == 2) { cl_store2(p, f32x2{x.v[0], x.v[N - 1]}, fast); return; }
    if (fast) p[0] = x.v[0];
    else __hip_atomic_store(p, x.v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float pick4f(const f32x4 v, int c) { return c == 0 ? v[0] : (c == 1 ? v[1] : (c == 2 ? v[2] : v[3])); }

template <int CELL, int HP, int UT, int KS, int R, bool F16>
__global__ void __launch_bounds__(256) rec_fwd_clg(RecArgs a) {
    static_assert(UT * KS == 4 && (R == 8 || R == 4), "4 waves; 8 or 4 live rows");
    using OPV = std::conditional_t<F16, f16x8c, bf16x8>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int G = Gates<CELL>::G, C = HP / (16 * UT), KW = HP / KS, KBW = KW / 32, GHP = G * HP, EPL = R / 4;
    constexpr int HROW = HP * 2 + 32, PLANEB = R * HROW;
    constexpr int W3_BYTES = F16 ? 0 : G * KBW * 4 * 1024;
    constexpr int NP = R * HP / 4 / 256;
    static_assert(R * HP / 4 % 256 == 0 && NP >= 1, "piece count");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [G][KBW][4 waves][64 lanes][16 B]
    char* hpl = smem_c + W3_BYTES;                       // [3 (F16: 2) planes][R rows][HROW]
    char* red = hpl + NPL * PLANEB;                      // [KS-1][UT][G][64 lanes][16 B]
    int tile, mem;
    if (!cl_ids(a, C, a.Bp / R, tile, mem)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = wave % UT, kq = wave / UT;
    const int j = lane & 15, q = lane >> 4;
    const int rl = j & (R - 1), eh = j / R;              // tile-local row; which EPL of the lane's 4 units this copy finishes
    const int row = tile * R + rl;
    const int T = a.T, Bp = a.Bp;
    const int ub = (mem * UT + ut) * 16;                 // first unit of this wave's tile
    const int u = ub + q * 4 + eh * EPL;
    const int k0 = kq * KW;
    const bool fin = kq == 0;
    bool dead = false;
    const bool fast = cl_same_xcc(a, C, tile, mem, (int*)red, dead);

    const int mylen = a.len[row];
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < R; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    OPV W1[G][KBW], W2[G][KBW];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            if constexpr (F16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 b1, b2;
                    cl_split2(a.Whid[(size_t)(k0 + kb * 32 + 8 * q + e) * GHP + g * HP + ub + j], b1, b2);
                    W1[g][kb][e] = b1; W2[g][kb][e] = b2;
                }
            } else {
            bf16x8 w3v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 b1, b2, b3;
                split3(a.Whid[(size_t)(k0 + kb * 32 + 8 * q + e) * GHP + g * HP + ub + j], b1, b2, b3);
                W1[g][kb][e] = b1; W2[g][kb][e] = b2; w3v[e] = b3;
            }
            *(bf16x8*)(w3 + ((g * KBW + kb) * 4 + wave) * 1024 + lane * 16) = w3v;
            }
        }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    typedef VecN<EPL> V;
    V h, c, pi, pf, po;
#pragma unroll
    for (int e = 0; e < EPL; ++e) { h.v[e] = 0.f; c.v[e] = 0.f; pi.v[e] = 0.f; pf.v[e] = 0.f; po.v[e] = 0.f; }
    if (fin) {
        h = ldn<EPL>(&a.hinit[u]);
        if (CELL == CELL_LSTM) {
            c = ldn<EPL>(&a.cinit[u]);
            pi = ldn<EPL>(&a.peep[u]); pf = ldn<EPL>(&a.peep[HP + u]); po = ldn<EPL>(&a.peep[2 * HP + u]);
            stn<EPL>(&a.cs[(size_t)row * HP + u], c);
        }
        cl_storen<EPL>(&a.hs[(size_t)row * HP + u], h, fast);
    }
    const bool fuse = a.gX != nullptr;
    V x[G], xn[G], bias[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) { bias[g].v[e] = 0.f; x[g].v[e] = 0.f; xn[g].v[e] = 0.f; }
        if (fuse && fin) bias[g] = ldn<EPL>(&a.gbias[g * HP + u]);
    }
    auto load_id = [&](int t) -> int { return fuse ? a.gX[(size_t)row * T + (t < T ? t : T - 1)] : 0; };
    auto load_x = [&](int t, int id, V (&d)[G]) {
        const float* src = fuse ? a.gWin + (size_t)id * GHP + u : a.xt + ((size_t)(t < T ? t : T - 1) * Bp + row) * GHP + u;
#pragma unroll
        for (int g = 0; g < G; ++g) d[g] = ldn<EPL>(&src[g * HP]);
    };
    int id_next = 0, id_nn = 0;
    if (fin) { id_next = load_id(1); load_x(0, load_id(0), x); }
    __syncthreads();                                     // W plane 3 visible

    const f32x4 z = f32x4{0, 0, 0, 0};
    auto mfma_phase = [&](f32x4 (&acc)[G]) {
        const char* hb = hpl + rl * HROW + k0 * 2 + q * 16;
#pragma unroll
        for (int g = 0; g < G; ++g) acc[g] = z;
        if constexpr (F16) {      // see rec_fwd_cl
            f32x4 lo[G];
#pragma unroll
            for (int g = 0; g < G; ++g) lo[g] = z;
            OPV hp[2][2];
            auto load_ops = [&](int kb, int s) {
                hp[s][0] = *(const OPV*)(hb + kb * 64);
                hp[s][1] = *(const OPV*)(hb + kb * 64 + PLANEB);
            };
            load_ops(0, 0);
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
                const int s = kb & 1;
                if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = cl_mfma(W1[g][kb], hp[s][1], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = cl_mfma(W2[g][kb], hp[s][0], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = cl_mfma(W1[g][kb], hp[s][0], acc[g]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 15");
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] += lo[g] * (1.0f / CL_F16_LO);
        } else {
        bf16x8 hp[2][3], wp[2][G];
        auto load_ops = [&](int kb, int s) {
            hp[s][0] = *(const bf16x8*)(hb + kb * 64);
            hp[s][1] = *(const bf16x8*)(hb + kb * 64 + PLANEB);
            hp[s][2] = *(const bf16x8*)(hb + kb * 64 + 2 * PLANEB);
#pragma unroll
            for (int g = 0; g < G; ++g) wp[s][g] = *(const bf16x8*)(w3 + ((g * KBW + kb) * 4 + wave) * 1024 + lane * 16);
        };
        load_ops(0, 0);
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(wp[s][g], hp[s][0], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W1[g][kb], hp[s][2], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W2[g][kb], hp[s][1], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W2[g][kb], hp[s][0], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W1[g][kb], hp[s][1], acc[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = MFMA_BF16(W1[g][kb], hp[s][0], acc[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15");
        }
    };
    auto exchange = [&](int t) {
        f32x4 v[NP];
        const float* base = a.hs + ((size_t)t * Bp + (size_t)tile * R) * HP;
        (void)cl_fetch<NP>(v, [&](int r, int col) { return base + (size_t)r * HP + col; }, HP, fast, dead, a.fault);
        cl_publish<NP, F16>(v, hpl, HP, HROW, PLANEB);
    };

    if (fin) {
        for (int t = 0; t < tmax; ++t) {
            load_x(t + 1, id_next, xn);
            id_nn = load_id(t + 2);
            exchange(t);
            __syncthreads();
            f32x4 acc[G];
            mfma_phase(acc);
            __syncthreads();                             // partials of the other K parts visible; hpl free again
            V sv[4];
            float as[G][EPL];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                f32x4 s4 = acc[g];
#pragma unroll
                for (int p = 0; p < KS - 1; ++p) s4 += *(const f32x4*)(red + (((p * UT + ut) * G + g) * 64 + lane) * 16);
#pragma unroll
                for (int e = 0; e < EPL; ++e) as[g][e] = pick4f(s4, eh * EPL + e);
            }
            const bool m = t < mylen;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float xs[G], ag[G], s[4];
#pragma unroll
                for (int g = 0; g < G; ++g) { xs[g] = x[g].v[e] + bias[g].v[e]; ag[g] = as[g][e]; }
                float hh = h.v[e], cc = c.v[e];
                cell_forward<CELL, true>(xs, ag, m, hh, cc, pi.v[e], pf.v[e], po.v[e], s, a.relu != 0);
                h.v[e] = hh; c.v[e] = cc;
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[k].v[e] = s[k];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = xn[g];
            id_next = id_nn;
            __builtin_amdgcn_sched_barrier(0);
            const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u;
            cl_storen<EPL>(&a.hs[o], h, fast);           // first: the other members are waiting for it
            if (CELL == CELL_LSTM) stn<EPL>(&a.cs[o], c);
            if (CELL != CELL_VANILLA) {
                const size_t og = sbr_blocked_index(t, row, u, Bp, HP);
#pragma unroll
                for (int k = 0; k < 4; ++k) stn<EPL>(&a.g[k][og], sv[k]);
            }
        }
        for (int t = tmax; t < T; ++t) {
            const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u;
            stn<EPL>(&a.hs[o], h);
            if (CELL == CELL_LSTM) stn<EPL>(&a.cs[o], c);
        }
    } else {
        for (int t = 0; t < tmax; ++t) {
            exchange(t);
            __syncthreads();
            f32x4 acc[G];
            mfma_phase(acc);
#pragma unroll
            for (int g = 0; g < G; ++g) *(f32x4*)(red + ((((kq - 1) * UT + ut) * G + g) * 64 + lane) * 16) = acc[g];
            __syncthreads();
        }
    }
}

template <int CELL, int HP, int UT, int KS, int R, bool F16>
__global__ void __launch_bounds__(256) rec_bwd_clg(RecArgs a) {
    static_assert(UT * KS == 4 && (R == 8 || R == 4), "4 waves; 8 or 4 live rows");
    using OPV = std::conditional_t<F16, f16x8c, bf16x8>;
    constexpr int NPL = F16 ? 2 : 3;
    constexpr int G = Gates<CELL>::G, C = HP / (16 * UT), GHP = G * HP, KW = GHP / KS, KBW = KW / 32, EPL = R / 4;
    constexpr int DROW = GHP * 2 + 32, PLANEB = R * DROW;
    constexpr int W3_BYTES = F16 ? 0 : KBW * 4 * 1024;
    constexpr int NP = R * GHP / 4 / 256;
    static_assert(R * GHP / 4 % 256 == 0 && NP >= 1, "piece count");
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [KBW][4 waves][64][16 B]
    char* dpl = smem_c + W3_BYTES;                       // [3 (F16: 2)][R][DROW]
    char* red = dpl + NPL * PLANEB;                      // [KS-1][UT][64 lanes][16 B]
    int tile, mem;
    if (!cl_ids(a, C, a.Bp / R, tile, mem)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ut = wave % UT, kq = wave / UT;
    const int j = lane & 15, q = lane >> 4;
    const int rl = j & (R - 1), eh = j / R;
    const int row = tile * R + rl;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    const int ub = (mem * UT + ut) * 16;
    const int u = ub + q * 4 + eh * EPL;
    const int k0 = kq * KW;
    const bool fin = kq == 0;
    bool dead = false;
    const bool fast = cl_same_xcc(a, C, tile, mem, (int*)red, dead);

    const int mylen = a.len[row];
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < R; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    OPV W1[KBW], W2[KBW];
#pragma unroll
    for (int kb = 0; kb < KBW; ++kb) {
        const float* src = a.Whid + (size_t)(ub + j) * GHP + k0 + kb * 32 + 8 * q;
        const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
        if constexpr (F16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 b1, b2;
                cl_split2(e < 4 ? lo[e & 3] : hi[e & 3], b1, b2);
                W1[kb][e] = b1; W2[kb][e] = b2;
            }
        } else {
        bf16x8 w3v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 b1, b2, b3;
            split3(e < 4 ? lo[e & 3] : hi[e & 3], b1, b2, b3);
            W1[kb][e] = b1; W2[kb][e] = b2; w3v[e] = b3;
        }
        *(bf16x8*)(w3 + (kb * 4 + wave) * 1024 + lane * 16) = w3v;
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    typedef VecN<EPL> V;
    V dh, dc, pi, pf, po, sdb[G], sdp[3], sv[4], hprev, cprev, cnew, hnew;
    auto zero = [](V& x) {
#pragma unroll
        for (int e = 0; e < EPL; ++e) x.v[e] = 0.f;
    };
    zero(dh); zero(dc); zero(pi); zero(pf); zero(po); zero(hprev); zero(cprev); zero(cnew); zero(hnew);
#pragma unroll
    for (int g = 0; g < G; ++g) zero(sdb[g]);
    zero(sdp[0]); zero(sdp[1]); zero(sdp[2]);
#pragma unroll
    for (int k = 0; k < 4; ++k) zero(sv[k]);
    if (fin && a.dh_last) dh = ldn<EPL>(&a.dh_last[(size_t)row * HP + u]);
    if (CELL == CELL_LSTM) { pi = ldn<EPL>(&a.peep[u]); pf = ldn<EPL>(&a.peep[HP + u]); po = ldn<EPL>(&a.peep[2 * HP + u]); }
    auto load_saved = [&](int t) {
        const size_t o = ((size_t)t * Bp + row) * HP + u;
        hprev = ldn<EPL>(&a.hs[o]);
        if (CELL != CELL_VANILLA) {
            const size_t og = sbr_blocked_index(t, row, u, Bp, HP);
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = ldn<EPL>(&a.g[k][og]);
        }
        if (CELL == CELL_LSTM) cprev = ldn<EPL>(&a.cs[o]);
    };
    __syncthreads();                                     // W plane 3 visible

    const f32x4 z4 = f32x4{0, 0, 0, 0};
    auto mfma_phase = [&]() -> f32x4 {
        const char* db = dpl + rl * DROW + k0 * 2 + q * 16;
        f32x4 acc[3] = {z4, z4, z4};
        if constexpr (F16) {      // see rec_bwd_cl
            OPV dp[2][2];
            auto load_ops = [&](int kb, int s) {
                dp[s][0] = *(const OPV*)(db + kb * 64);
                dp[s][1] = *(const OPV*)(db + kb * 64 + PLANEB);
            };
            load_ops(0, 0);
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
                const int s = kb & 1;
                if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = cl_mfma(W1[kb], dp[s][1], acc[1]);
                acc[2] = cl_mfma(W2[kb], dp[s][0], acc[2]);
                acc[0] = cl_mfma(W1[kb], dp[s][0], acc[0]);
                __builtin_amdgcn_sched_barrier(0);
            }
            asm volatile("s_nop 15");
            return (acc[0] + (acc[1] + acc[2]) * (1.0f / CL_F16_LO)) * (1.0f / CL_F16_DSCALE);
        } else {
        bf16x8 dp[2][3], wp[2];
        auto load_ops = [&](int kb, int s) {
            dp[s][0] = *(const bf16x8*)(db + kb * 64);
            dp[s][1] = *(const bf16x8*)(db + kb * 64 + PLANEB);
            dp[s][2] = *(const bf16x8*)(db + kb * 64 + 2 * PLANEB);
            wp[s] = *(const bf16x8*)(w3 + (kb * 4 + wave) * 1024 + lane * 16);
        };
        load_ops(0, 0);
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < KBW) load_ops(kb + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = MFMA_BF16(wp[s], dp[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dp[s][2], acc[1]);
            acc[2] = MFMA_BF16(W2[kb], dp[s][1], acc[2]);
            acc[0] = MFMA_BF16(W2[kb], dp[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dp[s][1], acc[1]);
            acc[2] = MFMA_BF16(W1[kb], dp[s][0], acc[2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15");
        return acc[0] + acc[1] + acc[2];
        }
    };
    auto exchange = [&](int t) {
        f32x4 v[NP];
        const float* bx = a.dxt + ((size_t)t * Bp + (size_t)tile * R) * GHP;
        const float* bc = a.dhi + ((size_t)t * Bp + (size_t)tile * R) * HP;
        (void)cl_fetch<NP>(v, [&](int r, int col) {
            return (CELL == CELL_GRU && col >= 2 * HP) ? bc + (size_t)r * HP + (col - 2 * HP) : bx + (size_t)r * GHP + col;
        }, GHP, fast, dead, a.fault);
        cl_publish<NP, F16>(v, dpl, GHP, DROW, PLANEB, F16 ? CL_F16_DSCALE : 1.0f);
    };

    if (fin) {
        for (int t = T - 1; t >= tmax; --t) {
            if (a.dh_ext) { const V e = ldn<EPL>(&a.dh_ext[((size_t)t * Bp + row) * HP + u]);
#pragma unroll
                for (int i = 0; i < EPL; ++i) dh.v[i] += e.v[i]; }
            V zz; zero(zz);
#pragma unroll
            for (int g = 0; g < G; ++g) stn<EPL>(&a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u], zz);
            if (CELL == CELL_GRU) stn<EPL>(&a.dhi[((size_t)t * Bp + row) * HP + u], zz);
        }
        if (tmax > 0) {
            load_saved(tmax - 1);
            const size_t o1 = ((size_t)tmax * Bp + row) * HP + u;
            if (CELL == CELL_LSTM) cnew = ldn<EPL>(&a.cs[o1]);
            if (CELL == CELL_VANILLA) hnew = ldn<EPL>(&a.hs[o1]);
        }
        for (int t = tmax - 1; t >= 0; --t) {
            if (a.dh_ext) { const V e = ldn<EPL>(&a.dh_ext[((size_t)t * Bp + row) * HP + u]);
#pragma unroll
                for (int i = 0; i < EPL; ++i) dh.v[i] += e.v[i]; }
            const bool m = t < mylen;
            V vxi[G], vhi[G];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float s[4] = {sv[0].v[e], sv[1].v[e], sv[2].v[e], sv[3].v[e]};
                float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
                float dhh = dh.v[e], dcc = dc.v[e];
                cell_backward<CELL, true>(m, clip, dhh, dcc, s, hprev.v[e], cprev.v[e], cnew.v[e], hnew.v[e], pi.v[e], pf.v[e],
                                          po.v[e], dxi, dhi, dp, a.relu != 0);
                dh.v[e] = dhh; dc.v[e] = dcc;
#pragma unroll
                for (int g = 0; g < G; ++g) { vxi[g].v[e] = dxi[g]; vhi[g].v[e] = dhi[g]; sdb[g].v[e] += dxi[g]; }
                sdp[0].v[e] += dp[0]; sdp[1].v[e] += dp[1]; sdp[2].v[e] += dp[2];
            }
            if (CELL == CELL_LSTM) cnew = cprev;
            if (CELL == CELL_VANILLA) hnew = hprev;
            __builtin_amdgcn_sched_barrier(0);
            load_saved(t > 0 ? t - 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < G; ++g) cl_storen<EPL>(&a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u], vxi[g], fast);
            if (CELL == CELL_GRU) cl_storen<EPL>(&a.dhi[((size_t)t * Bp + row) * HP + u], vhi[2], fast);
            exchange(t);
            __syncthreads();
            f32x4 sum = mfma_phase();
            __syncthreads();
#pragma unroll
            for (int p = 0; p < KS - 1; ++p) sum += *(const f32x4*)(red + ((p * UT + ut) * 64 + lane) * 16);
#pragma unroll
            for (int e = 0; e < EPL; ++e) dh.v[e] += pick4f(sum, eh * EPL + e);
        }
        float* part = a.part + (size_t)tile * (GHP + 5 * HP);
        V v[G + 5];
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = sdb[g];
        v[G] = sdp[0]; v[G + 1] = sdp[1]; v[G + 2] = sdp[2];
        v[G + 3] = dc; v[G + 4] = dh;
#pragma unroll
        for (int k = 0; k < G + 5; ++k)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                float sum = v[k].v[e];
#pragma unroll
                for (int o = 1; o < R; o <<= 1) sum += __shfl_xor(sum, o);
                v[k].v[e] = sum;
            }
        if (rl == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) stn<EPL>(&part[g * HP + u], v[g]);
#pragma unroll
            for (int k = 0; k < 5; ++k) stn<EPL>(&part[GHP + k * HP + u], v[G + k]);
        }
    } else {
        for (int t = tmax - 1; t >= 0; --t) {
            exchange(t);
            __syncthreads();
            const f32x4 sum = mfma_phase();
            *(f32x4*)(red + (((kq - 1) * UT + ut) * 64 + lane) * 16) = sum;
            __syncthreads();
        }
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
bool sbr_rec_cluster_ok(const RecArgs& a) {
    return a.cluster && !a.f32_mfma && (a.Hp == 256 || a.Hp == 512) && a.Bp % SBR_CL_ROWS == 0;
}
// rows per tile of the backward launch (part[] has Bp / rows blocks)
// the 16-row kernels: both chains on fp16 planes (one answer for the forward and the backward launch of a step: they share
// the exchange arrays' and the partial sums' layout), SBR_CL16=0 keeps the 8-row kernels
bool sbr_rec_c16_ok(const RecArgs& a) {
    const char* ce = getenv("SBR_CL16");                           // read per launch: the tests flip it
    const bool on = ce ? atoi(ce) != 0 : true;
    if (!(on && sbr_rec_cluster_ok(a) && a.xh && a.pring && a.Bp % 16 == 0 && cl_f16_fwd(a) && cl_f16_bwd(a))) return false;
    if (a.cell != SBR_CELL_VANILLA && a.g[0])             // the saved gates are ONE region [T][Bp][Hp][4] under the four arrays
        for (int k = 1; k < 4; ++k) if (a.g[k] != a.g[0] + (size_t)k * a.T * a.Bp * a.Hp) return false;
    return true;
}
int sbr_rec_cluster_bwd_rows(const RecArgs& a) {
    if (sbr_rec_c16_ok(a)) return 16;
    return a.Hp == 512 && !cl_f16_bwd(a) ? 4 : SBR_CL_ROWS;
}
size_t sbr_rec_c16_ring_floats(int Bp, int Hp) { return (size_t)SBR_C16_RING * (Bp / 16) * (Hp / 16) * (Hp / 16) * 256; }

template <int CELL, int HP>
static hipError_t fwd_cl(hipStream_t s, const RecArgs& a) {
    constexpr int G = Gates<CELL>::G, R = SBR_CL_ROWS;
    if (sbr_rec_c16_ok(a)) return launch_rec_forward_c16(s, a);
    hipError_t e = hipMemsetAsync(a.hs, 0xFF, (size_t)(a.T + 1) * a.Bp * HP * sizeof(float), s);   // sentinel: see the header
    if (e != hipSuccess) return e;
    if (HP == 256) {
        if (cl_f16_fwd(a)) {
            const size_t lds = 2 * (size_t)R * (HP * 2 + 32) + 2 * G * 1024;
            CL_LAUNCH((rec_fwd_cl<CELL, 256, R, true>), 256 / 32, R, lds);
        } else {
            const size_t lds = (size_t)G * (HP / 64) * 4 * 1024 + 3 * (size_t)R * (HP * 2 + 32) + 2 * G * 1024;
            CL_LAUNCH((rec_fwd_cl<CELL, 256, R, false>), 256 / 32, R, lds);
        }
    } else {   // 512: one unit tile per workgroup, K in four parts, a cluster = 32 workgroups
        if (cl_f16_fwd(a)) {
            const size_t lds = 2 * (size_t)R * (HP * 2 + 32) + 3 * G * 1024;
            CL_LAUNCH((rec_fwd_clg<CELL, 512, 1, 4, R, true>), 512 / 16, R, lds);
        } else {
            const size_t lds = (size_t)G * (HP / 4 / 32) * 4 * 1024 + 3 * (size_t)R * (HP * 2 + 32) + 3 * G * 1024;
            CL_LAUNCH((rec_fwd_clg<CELL, 512, 1, 4, R, false>), 512 / 16, R, lds);
        }
    }
    return hipGetLastError();
}
template <int CELL, int HP>
static hipError_t bwd_cl(hipStream_t s, const RecArgs& a) {
    constexpr int G = Gates<CELL>::G, GHP = G * HP;
    if (!a.sentinel_done) {                               // else: sbr_rec_bwd_cl_fill ran on the side stream during the output phase
        const hipError_t e = sbr_rec_bwd_cl_fill(s, a);
        if (e != hipSuccess) return e;
    }
    if (sbr_rec_c16_ok(a)) return launch_rec_backward_c16(s, a);
    if (HP == 256) {
        constexpr int R = SBR_CL_ROWS;
        if (cl_f16_bwd(a)) {
            const size_t lds = 2 * (size_t)R * (GHP * 2 + 32) + 2 * 1024;
            CL_LAUNCH((rec_bwd_cl<CELL, 256, R, true>), 256 / 32, R, lds);
        } else {
            const size_t lds = (size_t)(GHP / 64) * 4 * 1024 + 3 * (size_t)R * (GHP * 2 + 32) + 2 * 1024;
            CL_LAUNCH((rec_bwd_cl<CELL, 256, R, false>), 256 / 32, R, lds);
        }
    } else {
        if (cl_f16_bwd(a)) {     // two fp16 planes and no W plane in LDS: 8-row tiles fit (66 KB), half the workgroup rounds
            constexpr int R = 8;
            const size_t lds = 2 * (size_t)R * (GHP * 2 + 32) + 3 * 1024;
            CL_LAUNCH((rec_bwd_clg<CELL, 512, 1, 4, R, true>), 512 / 16, R, lds);
        } else {
            constexpr int R = 4;
            const size_t lds = (size_t)(GHP / 4 / 32) * 4 * 1024 + 3 * (size_t)R * (GHP * 2 + 32) + 3 * 1024;
            CL_LAUNCH((rec_bwd_clg<CELL, 512, 1, 4, R, false>), 512 / 16, R, lds);
        }
    }
    return hipGetLastError();
}

// the sentinel fill of the backward exchange arrays (dxt, GRU: + the compact candidate slice)
hipError_t sbr_rec_bwd_cl_fill(hipStream_t s, const RecArgs& a) {
    if (sbr_rec_c16_ok(a)) return sbr_rec_c16_fill(s, a);
    hipError_t e = hipMemsetAsync(a.dxt, 0xFF, (size_t)a.T * a.Bp * a.G * a.Hp * sizeof(float), s);
    if (e == hipSuccess && a.cell == SBR_CELL_GRU) e = hipMemsetAsync(a.dhi, 0xFF, (size_t)a.T * a.Bp * a.Hp * sizeof(float), s);
    return e;
}

#define CL_DISPATCH(FN) \
    if (a.Hp == 512) { \
        switch (a.cell) { case SBR_CELL_LSTM: return FN<CELL_LSTM, 512>(s, a); case SBR_CELL_GRU: return FN<CELL_GRU, 512>(s, a); \
                          default: return FN<CELL_VANILLA, 512>(s, a); } \
    } \
    switch (a.cell) { case SBR_CELL_LSTM: return FN<CELL_LSTM, 256>(s, a); case SBR_CELL_GRU: return FN<CELL_GRU, 256>(s, a); \
                      default: return FN<CELL_VANILLA, 256>(s, a); }

hipError_t launch_rec_forward_cl(hipStream_t s, const RecArgs& a) { CL_DISPATCH(fwd_cl) }
hipError_t launch_rec_backward_cl(hipStream_t s, const RecArgs& a) { CL_DISPATCH(bwd_cl) }
