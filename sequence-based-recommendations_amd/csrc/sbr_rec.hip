// Recurrent forward / BPTT kernels for gfx950 (CDNA4, wave64).
//
// Math follows the reference's scan step functions (neural_networks/sparse_lstm.py:377-425
// LSTM, :764-805 GRU, :1120-1152 Vanilla); the backward is the hand-derived BPTT pinned by
// oracle/rnn_oracle.py (the reference obtains it from theano.grad through scan).
//
// Design (one workgroup = one 16-row tile of the batch for ALL T steps, no inter-workgroup sync):
//   * v_mfma_f32_16x16x4_f32 computes D[unit][row] = sum_k W_hid[k][unit] * h[row][k]:
//     the A operand is a W_hid fragment (register-resident for Hp <= 128: one 16-unit tile x G
//     gates x Hp/4 k-steps per wave = 96 VGPRs for GRU-128), the B operand is h_{t-1} read from
//     LDS (written by all waves at the end of the previous step, double-buffered: one
//     __syncthreads per step).  k is permuted (lane group q owns k in [q*Hp/4, (q+1)*Hp/4)) so a
//     lane's B values are contiguous in LDS (ds_read_b128).
//   * With D[unit][row], the accumulator layout gives every lane 4 CONSECUTIVE hidden units of
//     ONE batch row -> xt loads, state, and all saved-activation stores are 16-byte accesses in
//     the natural [t][row][unit] layout; gate math stays in registers (c_t, h_t never leave the
//     wave except h_t -> LDS for the next step's B operand).
//   * Rows are left-aligned (rnn_one_hot.py:90-101): steps t >= len(row) copy state
//     (sparse_lstm.py:422-423); steps beyond the tile's longest row skip the MFMA work.
#include "sbr_common.h"

#include "sbr_cell.h"

// ---------------------------------------------------------------------------------------
// LDS tile layout shared by both MFMA kernels: 16 rows (batch rows of the tile) x K values, K split
// in 4 chunks (one per MFMA k-lane-group q).  Chunk stride CH = multiple of 64 floats and row
// stride 4*CH+4 make every ds_read_b128 of the B operand conflict-free (rows land on distinct
// 16-byte slots of the 256-byte bank row for each of the instruction's four 16-lane groups).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ int lds_chunk(int kper) { return (kper + 63) / 64 * 64; }

// Saved gate activations g[k] (forward -> BPTT only) live in a tile-blocked layout
//   [t][row tile of 16][unit tile of 16][row 16][unit 16]
// so that the 16x16 tile one wave owns is ONE contiguous KiB: a wave-wide 16-B store/load touches
// 8 full 128-B lines instead of 16 half lines of 16 different rows (the row-major form made the
// CU's memory pipeline, not the MFMA pipe, the per-step bottleneck).
__device__ __forceinline__ size_t gate_index(int t, int row, int u, int Bp, int Hp) {
    return sbr_blocked_index(t, row, u, Bp, Hp);
}
// xt: blocked when the gather wrote it (layer 0), row-major when a GEMM did (layers >= 1)
__device__ __forceinline__ size_t xt_index(int blocked, int t, int row, int col, int Bp, int ncols) {
    return blocked ? sbr_blocked_index(t, row, col, Bp, ncols) : ((size_t)t * Bp + row) * ncols + col;
}

// ---------------------------------------------------------------------------------------
// MFMA persistent forward.  KS_RES > 0: Hp = 4*KS_RES compile-time, W_hid fragments in VGPRs.
// KS_RES == 0: runtime Hp, fragments streamed from L2 every step.  NT unit tiles per wave.
// Per step: [B operand: ds_read_b128] -> [G*Hp/4 MFMAs per tile] -> [gate math in registers]
// -> [16-B stores of h_t and the activations BPTT needs] -> [prefetch xt of step t+1] ->
// [h_t -> LDS] -> one __syncthreads.  The xt prefetch is issued BEFORE the barrier so that its
// HBM latency hides under the barrier wait and the next step's MFMA phase.
// ---------------------------------------------------------------------------------------
template <int CELL, int NT, int KS_RES>
__global__ void __launch_bounds__(KS_RES > 0 ? KS_RES * 16 : 1024) rec_fwd_mfma(RecArgs a) {
    constexpr int G = Gates<CELL>::G;
    const int Hp = KS_RES > 0 ? 4 * KS_RES : a.Hp;
    const int KS = Hp / 4, GHp = G * Hp;
    const int CH = lds_chunk(KS), ROW = 4 * CH + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][16][ROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + j;
    const int T = a.T, Bp = a.Bp;

    const int mylen = a.len[row];
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    // register-resident A fragments: W[n][g][kk] = W_hid[q*KS+kk][g*Hp + tile*16 + j]
    float W[NT][G][KS_RES > 0 ? KS_RES : 1];
    if (KS_RES > 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int kk = 0; kk < KS_RES; ++kk)
                    W[n][g][kk] = a.Whid[(size_t)(q * KS + kk) * GHp + g * Hp + (wave * NT + n) * 16 + j];
        // vmcnt(0) through the builtin (the waitcnt pass sees it): without it hipcc, unable to tell the
        // one-off W loads from the per-step stores on the in-order vmcnt counter, makes every loop
        // iteration wait for the PREVIOUS step's stores right after its first MFMAs (~1.5k cycles/step)
        __builtin_amdgcn_s_waitcnt(0x0F70);
    }

    f32x4 h[NT], c[NT], pi[NT], pf[NT], po[NT];
    int wofs[NT];          // where this lane's 4 units live inside an LDS row
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int u0 = (wave * NT + n) * 16 + q * 4;
        wofs[n] = (u0 / KS) * CH + (u0 % KS);
        h[n] = *(const f32x4*)&a.hinit[u0];
        c[n] = f32x4{0, 0, 0, 0}; pi[n] = c[n]; pf[n] = c[n]; po[n] = c[n];
        if (CELL == CELL_LSTM) {
            c[n] = *(const f32x4*)&a.cinit[u0];
            pi[n] = *(const f32x4*)&a.peep[u0]; pf[n] = *(const f32x4*)&a.peep[Hp + u0];
            po[n] = *(const f32x4*)&a.peep[2 * Hp + u0];
            *(f32x4*)&a.cs[(size_t)row * Hp + u0] = c[n];
        }
        *(f32x4*)&a.hs[(size_t)row * Hp + u0] = h[n];
        *(f32x4*)&smem[j * ROW + wofs[n]] = h[n];
    }

    // xt is fetched one whole step ahead (issued at the top of step t for step t+1, BEFORE step t's
    // stores): on the in-order vmcnt counter the loads then never queue behind fresh stores
    f32x4 x[NT][G], xn[NT][G];
    auto load_x = [&](int t, f32x4 (&d)[NT][G]) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g)
                d[n][g] = *(const f32x4*)&a.xt[((size_t)t * Bp + row) * GHp + g * Hp + (wave * NT + n) * 16 + q * 4];
    };
    if (tmax > 0) load_x(0, x);
    __syncthreads();
    unsigned long long p_c0 = 0, p_r0 = 0, p_work = 0, p_bar = 0, p_ta = 0, p_mfma = 0, p_epi = 0;
    if (a.prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }

    for (int t = 0; t < T; ++t) {
        if (a.prof) p_ta = clock64();
        if (t < tmax) {                                           // workgroup-uniform
            if (t + 1 < tmax) load_x(t + 1, xn);
            const float* hb = smem + (t & 1) * 16 * ROW + j * ROW + q * CH;
            f32x4 acc[NT][G];
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < G; ++g) acc[n][g] = f32x4{0, 0, 0, 0};
            if (KS_RES > 0) {
                f32x4 hv[KS_RES / 4 > 0 ? KS_RES / 4 : 1];
#pragma unroll
                for (int k4 = 0; k4 < KS_RES / 4; ++k4) hv[k4] = *(const f32x4*)&hb[4 * k4];
#pragma unroll
                for (int kk = 0; kk < KS_RES; ++kk)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
#pragma unroll
                        for (int g = 0; g < G; ++g)
                            acc[n][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[n][g][kk], hv[kk >> 2][kk & 3], acc[n][g], 0, 0, 0);
            } else {
                for (int kc = 0; kc < KS; kc += 4) {
                    const f32x4 hv = *(const f32x4*)&hb[kc];
                    const float* wrow = a.Whid + (size_t)(q * KS + kc) * GHp + wave * NT * 16 + j;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
#pragma unroll
                            for (int g = 0; g < G; ++g)
                                acc[n][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wrow[(size_t)kk * GHp + g * Hp + n * 16], hv[kk],
                                                                                 acc[n][g], 0, 0, 0);
                }
            }
            // MFMA D -> VALU read is a software-managed hazard (8-pass XDL: 12 wait states).  hipcc pads it
            // inside a basic block, but a branch between the last MFMA and the first read (streamed-W
            // loop back-edge, or any `if`) got only `s_nop 0`: pad explicitly, 16 cycles per step.
            asm volatile("s_nop 15");
            const bool m = t < mylen;
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int u0 = (wave * NT + n) * 16 + q * 4;
                f32x4 sv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xs[G], as[G], s[4];
#pragma unroll
                    for (int g = 0; g < G; ++g) { xs[g] = x[n][g][e]; as[g] = acc[n][g][e]; }
                    float hh = h[n][e], cc = c[n][e];
                    cell_forward<CELL, true>(xs, as, m, hh, cc, pi[n][e], pf[n][e], po[n][e], s, a.relu != 0);
                    h[n][e] = hh; c[n][e] = cc;
#pragma unroll
                    for (int k = 0; k < 4; ++k) sv[k][e] = s[k];
                }
                if (CELL != CELL_VANILLA) {
                    const size_t o = gate_index(t, row, u0, Bp, Hp);
#pragma unroll
                    for (int k = 0; k < 4; ++k) *(f32x4*)&a.g[k][o] = sv[k];
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int u0 = (wave * NT + n) * 16 + q * 4;
            const size_t o = ((size_t)(t + 1) * Bp + row) * Hp + u0;
            *(f32x4*)&a.hs[o] = h[n];
            if (CELL == CELL_LSTM) *(f32x4*)&a.cs[o] = c[n];
        }
        if (t + 1 < tmax) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < G; ++g) x[n][g] = xn[n][g];
#pragma unroll
            for (int n = 0; n < NT; ++n)
                *(f32x4*)&smem[((t + 1) & 1) * 16 * ROW + j * ROW + wofs[n]] = h[n];
            if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
            __syncthreads();
            if (a.prof) p_bar += clock64() - p_ta;
        }
    }
    if (a.prof && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_work; o[3] = p_bar;
        o[4] = p_mfma; o[5] = p_epi;
    }
}

// ---------------------------------------------------------------------------------------
// MFMA persistent backward (BPTT).  D[unit k][row] = sum_j W_hid[k][j] * dhi[row][j]:
// A = W_hid rows of the wave's unit tile (KS_RES > 0: register resident, JS = G*Hp/4 j-steps),
// B = this step's dhi tile in LDS.  dxt/dhi rows are streamed out for the scatter-add and the
// split-K weight-gradient GEMM (dW_hid = hs_prev^T . dhi), which run after the chain.
// The saved activations of step t-1 are prefetched before the barrier of step t so their HBM
// latency hides under the MFMA phase; c_t / h_t of the step above are carried in registers.
// ---------------------------------------------------------------------------------------
template <int CELL, int NT>
struct SavedAct { f32x4 sv[NT][4]; f32x4 hprev[NT]; f32x4 cprev[NT]; };

template <int CELL, int NT, int KS_RES>
__global__ void __launch_bounds__(KS_RES > 0 ? KS_RES * 16 : 1024) rec_bwd_mfma(RecArgs a, int dbuf) {
    constexpr int G = Gates<CELL>::G;
    constexpr int JS_RES = G * KS_RES;
    const int Hp = KS_RES > 0 ? 4 * KS_RES : a.Hp;
    const int GHp = G * Hp, JS = GHp / 4;
    const int CH = lds_chunk(JS), ROW = 4 * CH + 4;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // [dbuf ? 2 : 1][16][ROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + j;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;

    const int mylen = a.len[row];
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    // A fragments: Wb[n][jj] = W_hid[tile*16 + j][q*JS + jj]
    float Wb[NT][JS_RES > 0 ? JS_RES : 1];
    if (KS_RES > 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int j4 = 0; j4 < JS_RES / 4; ++j4) {
                const f32x4 w = *(const f32x4*)&a.Whid[(size_t)((wave * NT + n) * 16 + j) * GHp + q * JS + 4 * j4];
#pragma unroll
                for (int e = 0; e < 4; ++e) Wb[n][4 * j4 + e] = w[e];
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);   // see rec_fwd_mfma
    }

    f32x4 dh[NT], dc[NT], pi[NT], pf[NT], po[NT];
    f32x4 sdb[NT][G], sdp[NT][3];
    int wofs[NT][G];       // LDS row offset of this lane's 4 units of gate g
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int u0 = (wave * NT + n) * 16 + q * 4;
        const f32x4 z = f32x4{0, 0, 0, 0};
        dh[n] = a.dh_last ? *(const f32x4*)&a.dh_last[(size_t)row * Hp + u0] : z;
        dc[n] = z; pi[n] = z; pf[n] = z; po[n] = z;
        if (CELL == CELL_LSTM) {
            pi[n] = *(const f32x4*)&a.peep[u0]; pf[n] = *(const f32x4*)&a.peep[Hp + u0];
            po[n] = *(const f32x4*)&a.peep[2 * Hp + u0];
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            sdb[n][g] = z;
            const int col = g * Hp + u0;
            wofs[n][g] = (col / JS) * CH + (col % JS);
        }
        sdp[n][0] = z; sdp[n][1] = z; sdp[n][2] = z;
    }

    SavedAct<CELL, NT> cur, nxt;
    f32x4 cnew[NT], hnew[NT];
    auto load_saved = [&](int t, SavedAct<CELL, NT>& d) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const size_t o = ((size_t)t * Bp + row) * Hp + (wave * NT + n) * 16 + q * 4;
            d.hprev[n] = *(const f32x4*)&a.hs[o];
            if (CELL != CELL_VANILLA) {
                const size_t og = gate_index(t, row, (wave * NT + n) * 16 + q * 4, Bp, Hp);
#pragma unroll
                for (int k = 0; k < 4; ++k) d.sv[n][k] = *(const f32x4*)&a.g[k][og];
            }
            if (CELL == CELL_LSTM) d.cprev[n] = *(const f32x4*)&a.cs[o];
        }
    };
    bool have = false;
    unsigned long long p_c0 = 0, p_r0 = 0, p_work = 0, p_bar = 0, p_ta = 0;
    if (a.prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }

    for (int t = T - 1; t >= 0; --t) {
        if (a.prof) p_ta = clock64();
        if (a.dh_ext) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
                dh[n] += *(const f32x4*)&a.dh_ext[((size_t)t * Bp + row) * Hp + (wave * NT + n) * 16 + q * 4];
        }
        if (t >= tmax) {                                          // whole tile masked: zero rows
            const f32x4 z = f32x4{0, 0, 0, 0};
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const size_t o = ((size_t)t * Bp + row) * GHp + g * Hp + (wave * NT + n) * 16 + q * 4;
                    *(f32x4*)&a.dxt[o] = z;
                    if (CELL == CELL_GRU && g == 2) *(f32x4*)&a.dhi[((size_t)t * Bp + row) * Hp + (wave * NT + n) * 16 + q * 4] = z;
                }
            continue;
        }
        if (!have) {                                              // first active step: nothing prefetched yet
            load_saved(t, cur);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const size_t o1 = ((size_t)(t + 1) * Bp + row) * Hp + (wave * NT + n) * 16 + q * 4;
                cnew[n] = CELL == CELL_LSTM ? *(const f32x4*)&a.cs[o1] : f32x4{0, 0, 0, 0};
                hnew[n] = CELL == CELL_VANILLA ? *(const f32x4*)&a.hs[o1] : f32x4{0, 0, 0, 0};
            }
            have = true;
        }
        const bool m = t < mylen;
        float* lds = smem + (dbuf ? (t & 1) : 0) * 16 * ROW;
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const int u0 = (wave * NT + n) * 16 + q * 4;
            f32x4 vxi[G], vhi[G];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                if (CELL != CELL_VANILLA) { s[0] = cur.sv[n][0][e]; s[1] = cur.sv[n][1][e]; s[2] = cur.sv[n][2][e]; s[3] = cur.sv[n][3][e]; }
                float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
                float dhh = dh[n][e], dcc = dc[n][e];
                const float cpv = CELL == CELL_LSTM ? cur.cprev[n][e] : 0.f;
                cell_backward<CELL, true>(m, clip, dhh, dcc, s, cur.hprev[n][e], cpv, cnew[n][e], hnew[n][e], pi[n][e],
                                          pf[n][e], po[n][e], dxi, dhi, dp, a.relu != 0);
                dh[n][e] = dhh; dc[n][e] = dcc;
#pragma unroll
                for (int g = 0; g < G; ++g) { vxi[g][e] = dxi[g]; vhi[g][e] = dhi[g]; sdb[n][g][e] += dxi[g]; }
                sdp[n][0][e] += dp[0]; sdp[n][1][e] += dp[1]; sdp[n][2][e] += dp[2];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const size_t og = ((size_t)t * Bp + row) * GHp + g * Hp + u0;
                *(f32x4*)&a.dxt[og] = vxi[g];
                if (CELL == CELL_GRU && g == 2) *(f32x4*)&a.dhi[((size_t)t * Bp + row) * Hp + u0] = vhi[g];
                *(f32x4*)&lds[j * ROW + wofs[n][g]] = vhi[g];
            }
        }
        if (t > 0) load_saved(t - 1, nxt);                        // prefetch: in flight across the MFMA phase
        if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
        __syncthreads();
        if (a.prof) { const unsigned long long tc = clock64(); p_bar += tc - p_ta; p_ta = tc; }
        const float* db = lds + j * ROW + q * CH;
        f32x4 acc[NT][2];
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc[n][0] = f32x4{0, 0, 0, 0}; acc[n][1] = f32x4{0, 0, 0, 0}; }
        if (KS_RES > 0) {
#pragma unroll
            for (int j4 = 0; j4 < JS_RES / 4; ++j4) {
                const f32x4 bv = *(const f32x4*)&db[4 * j4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[n][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wb[n][4 * j4 + e], bv[e], acc[n][e & 1], 0, 0, 0);
            }
        } else {
            for (int j4 = 0; j4 < JS / 4; ++j4) {
                const f32x4 bv = *(const f32x4*)&db[4 * j4];
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const f32x4 w = *(const f32x4*)&a.Whid[(size_t)((wave * NT + n) * 16 + j) * GHp + q * JS + 4 * j4];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        acc[n][e & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[e], bv[e], acc[n][e & 1], 0, 0, 0);
                }
            }
        }
        asm volatile("s_nop 15");                                  // MFMA D -> VALU read hazard, see rec_fwd_mfma
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            dh[n] += acc[n][0] + acc[n][1];
            if (CELL == CELL_LSTM) cnew[n] = cur.cprev[n];          // c_{t-1} is c_t of the step below
            if (CELL == CELL_VANILLA) hnew[n] = cur.hprev[n];
        }
        cur = nxt;
        if (!dbuf) __syncthreads();
        if (a.prof) p_work += clock64() - p_ta;
    }
    if (a.prof && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_work; o[3] = p_bar;
    }

    // per-workgroup partial sums -> part[block][ G*Hp | dpi | dpf | dpo | dcinit | dhinit ]
    float* part = a.part + (size_t)blockIdx.x * (GHp + 5 * Hp);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int u0 = (wave * NT + n) * 16 + q * 4;
        f32x4 v[G + 5];
#pragma unroll
        for (int g = 0; g < G; ++g) v[g] = sdb[n][g];
        v[G] = sdp[n][0]; v[G + 1] = sdp[n][1]; v[G + 2] = sdp[n][2]; v[G + 3] = dc[n]; v[G + 4] = dh[n];
#pragma unroll
        for (int k = 0; k < G + 5; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s = v[k][e];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
                v[k][e] = s;
            }
        if (j == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) *(f32x4*)&part[g * Hp + u0] = v[g];
#pragma unroll
            for (int k = 0; k < 5; ++k) *(f32x4*)&part[GHp + k * Hp + u0] = v[G + k];
        }
    }
}

// ---------------------------------------------------------------------------------------
// "bf16x6" recurrent kernels: the same persistent structure, but the per-step GEMM runs on
// v_mfma_f32_16x16x32_bf16 with every f32 operand split exactly into three bf16 planes
// (v = b1 + b2 + b3, residual < 2^-26 |v|) and the six products of order >= 2^-18 kept:
//     a.w ~= a1w1 + a1w2 + a2w1 + a1w3 + a2w2 + a3w1      (dropped terms < 2^-26 |a||w|)
// bf16 x bf16 products are exact in f32 and the MFMA accumulates in f32, so the result is
// f32-rounding class (tools/probes/mfma_bf16_probe.hip measures 2.7e-7 vs 6.2e-7 for an fmaf chain
// on K=32) -- at 6 x 16 = 96 MFMA cycles per 16x16x32 block instead of 8 x 32 = 256 on the f32
// MFMA (v_mfma_f32_16x16x4_f32), i.e. 2.67x less time on the matrix pipe, which is what bounds
// the 2*T-step dependent chain.  Planes 1,2 of W_hid stay in VGPRs, plane 3 in LDS; h_t (resp.
// dhi_t) is split by the producing wave and published to LDS as three bf16 planes.
// ---------------------------------------------------------------------------------------
template <int CELL, int HP>
__global__ void __launch_bounds__(HP * 4) rec_fwd_x6(RecArgs a) {
    constexpr int G = Gates<CELL>::G, KB = HP / 32, NW = HP / 16, GHP = G * HP;
    constexpr int HROW = HP * 2 + 32;                    // bytes per row of one h plane (conflict-free b128 reads)
    constexpr int W3_BYTES = G * KB * NW * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [G][KB][NW][64 lanes][16 B]
    char* hbuf = smem_c + W3_BYTES;                      // [2][3 planes][16 rows][HROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    // a workgroup owns a.rpt (<= 16) batch rows: with fewer rows per CU the per-CU store rate (~13 B/clk,
    // the real bound of this chain once the matrix pipe is on bf16) is spread over more CUs; MFMA columns
    // j >= rpt compute on a duplicate row and are never stored
    const int rpt = a.rpt;
    const bool live = j < rpt;
    const int row = blockIdx.x * rpt + (live ? j : j % rpt);
    const int T = a.T, Bp = a.Bp;
    const int u0 = wave * 16 + q * 4;

    const int mylen = live ? a.len[row] : 0;
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    // A operand planes: lane (unit j of the tile, k-group q) holds W_hid[kb*32 + 8q + e][g*HP + tile*16 + j]
    bf16x8 W1[G][KB], W2[G][KB];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            bf16x8 w3v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 b1, b2, b3;
                split3(a.Whid[(size_t)(kb * 32 + 8 * q + e) * GHP + g * HP + wave * 16 + j], b1, b2, b3);
                W1[g][kb][e] = b1; W2[g][kb][e] = b2; w3v[e] = b3;
            }
            *(bf16x8*)(w3 + ((g * KB + kb) * NW + wave) * 1024 + lane * 16) = w3v;
        }
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // see rec_fwd_mfma: clear the VMEM scoreboard

    f32x4 h, c, pi, pf, po;
    {
        const f32x4 z = f32x4{0, 0, 0, 0};
        h = *(const f32x4*)&a.hinit[u0];
        c = z; pi = z; pf = z; po = z;
        if (CELL == CELL_LSTM) {
            c = *(const f32x4*)&a.cinit[u0];
            pi = *(const f32x4*)&a.peep[u0]; pf = *(const f32x4*)&a.peep[HP + u0]; po = *(const f32x4*)&a.peep[2 * HP + u0];
            if (live) *(f32x4*)&a.cs[(size_t)row * HP + u0] = c;
        }
        if (live) *(f32x4*)&a.hs[(size_t)row * HP + u0] = h;
    }
    auto publish_h = [&](int buf) {                      // three bf16 planes of this lane's 4 units
        bf16x4 p1, p2, p3;
        split3x4(h, p1, p2, p3);
        char* base = hbuf + (size_t)buf * 3 * 16 * HROW + j * HROW + u0 * 2;
        *(bf16x4*)(base) = p1; *(bf16x4*)(base + 16 * HROW) = p2; *(bf16x4*)(base + 32 * HROW) = p3;
    };
    publish_h(0);

    f32x4 x[G], xn[G];
    auto load_x = [&](int t, f32x4 (&d)[G]) {
#pragma unroll
        for (int g = 0; g < G; ++g) d[g] = *(const f32x4*)&a.xt[((size_t)t * Bp + row) * GHP + g * HP + u0];
    };
    if (tmax > 0) load_x(0, x);
    __syncthreads();
    unsigned long long p_c0 = 0, p_r0 = 0, p_work = 0, p_bar = 0, p_ta = 0;
    if (a.prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }

    for (int t = 0; t < T; ++t) {
        if (a.prof) p_ta = clock64();
        f32x4 sv[4];
        if (t < tmax) {                                           // workgroup-uniform
            if (t + 1 < tmax) load_x(t + 1, xn);
            const char* hb = hbuf + (size_t)(t & 1) * 3 * 16 * HROW + j * HROW + q * 16;
            f32x4 acc[G];
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = f32x4{0, 0, 0, 0};
            // operand reads of k-block kb+1 are issued BEFORE the MFMAs of k-block kb (register double
            // buffer + sched_barrier): left alone, hipcc waits lgkmcnt(0) in front of every MFMA group and
            // exposes the LDS latency KB times per step
            bf16x8 hp[2][3], wp[2][G];
            auto load_ops = [&](int kb, int s) {
                hp[s][0] = *(const bf16x8*)(hb + kb * 64);
                hp[s][1] = *(const bf16x8*)(hb + kb * 64 + 16 * HROW);
                hp[s][2] = *(const bf16x8*)(hb + kb * 64 + 32 * HROW);
#pragma unroll
                for (int g = 0; g < G; ++g) wp[s][g] = *(const bf16x8*)(w3 + ((g * KB + kb) * NW + wave) * 1024 + lane * 16);
            };
            load_ops(0, 0);
            __builtin_amdgcn_s_setprio(0);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int s = kb & 1;
                if (kb + 1 < KB) load_ops(kb + 1, s ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                // smallest terms first; gates interleaved so consecutive MFMAs hit different accumulators
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
            asm volatile("s_nop 15");                             // MFMA D -> VALU read hazard (see rec_fwd_mfma)
            // the gate-math / publish phase is VALU+LDS work that shares the SIMD's issue port with the partner
            // wave's MFMA stream: give it priority, or it is starved for the whole length of that stream
            __builtin_amdgcn_s_setprio(3);
            const bool m = t < mylen;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float xs[G], as[G], s[4];
#pragma unroll
                for (int g = 0; g < G; ++g) { xs[g] = x[g][e]; as[g] = acc[g][e]; }
                float hh = h[e], cc = c[e];
                cell_forward<CELL, true>(xs, as, m, hh, cc, pi[e], pf[e], po[e], s, a.relu != 0);
                h[e] = hh; c[e] = cc;
#pragma unroll
                for (int k = 0; k < 4; ++k) sv[k][e] = s[k];
            }
        }
        if (t + 1 < tmax) {
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = xn[g];
            publish_h((t + 1) & 1);
            if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
            __syncthreads();
            if (a.prof) p_bar += clock64() - p_ta;
        }
        // this step's stores are issued AFTER the barrier: they leave the critical path (h_t is already
        // published) and drain under the next step's MFMA phase
        if (live) {
            if (CELL != CELL_VANILLA && t < tmax) {
                const size_t o = gate_index(t, row, u0, Bp, HP);
#pragma unroll
                for (int k = 0; k < 4; ++k) *(f32x4*)&a.g[k][o] = sv[k];
            }
            const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u0;
            *(f32x4*)&a.hs[o] = h;
            if (CELL == CELL_LSTM) *(f32x4*)&a.cs[o] = c;
        }
    }
    if (a.prof && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_work; o[3] = p_bar;
    }
}

template <int CELL, int HP>
__global__ void __launch_bounds__(HP * 4) rec_bwd_x6(RecArgs a, int dbuf) {
    constexpr int G = Gates<CELL>::G, NW = HP / 16, GHP = G * HP, KB = GHP / 32;
    constexpr int DROW = GHP * 2 + 32;                   // bytes per row of one dhi plane
    constexpr int W3_BYTES = KB * NW * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [KB][NW][64][16 B]
    char* dbufp = smem_c + W3_BYTES;                     // [dbuf ? 2 : 1][3][rpt][DROW]: only live rows are stored;
                                                         // phantom MFMA columns read (broadcast) row j % rpt
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    // a workgroup owns a.rpt (<= 16) batch rows: with fewer rows per CU the per-CU store rate (~13 B/clk,
    // the real bound of this chain once the matrix pipe is on bf16) is spread over more CUs; MFMA columns
    // j >= rpt compute on a duplicate row and are never stored
    const int rpt = a.rpt;
    const bool live = j < rpt;
    const int row = blockIdx.x * rpt + (live ? j : j % rpt);
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    const int u0 = wave * 16 + q * 4;

    const int mylen = live ? a.len[row] : 0;
    int tmax = mylen;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) tmax = max(tmax, __shfl_xor(tmax, o));

    // A operand planes: lane (unit j of the tile, k-group q) holds W_hid[tile*16 + j][kb*32 + 8q + e]
    bf16x8 W1[KB], W2[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float* src = a.Whid + (size_t)(wave * 16 + j) * GHP + kb * 32 + 8 * q;
        const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
        bf16x8 w3v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 b1, b2, b3;
            split3(e < 4 ? lo[e & 3] : hi[e & 3], b1, b2, b3);
            W1[kb][e] = b1; W2[kb][e] = b2; w3v[e] = b3;
        }
        *(bf16x8*)(w3 + (kb * NW + wave) * 1024 + lane * 16) = w3v;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const f32x4 z4 = f32x4{0, 0, 0, 0};
    const bool first = a.t_hi >= T, last = a.t_lo <= 0;          // first / last launch of the chunked chain
    f32x4 dh = z4, dc = z4, pi = z4, pf = z4, po = z4;
    if (live) {
        if (first) { if (a.dh_last) dh = *(const f32x4*)&a.dh_last[(size_t)row * HP + u0]; }
        else {
            dh = *(const f32x4*)&a.state[(size_t)row * HP + u0];
            if (CELL == CELL_LSTM) dc = *(const f32x4*)&a.state[((size_t)Bp + row) * HP + u0];
        }
    }
    if (CELL == CELL_LSTM) { pi = *(const f32x4*)&a.peep[u0]; pf = *(const f32x4*)&a.peep[HP + u0]; po = *(const f32x4*)&a.peep[2 * HP + u0]; }
    f32x4 sdb[G], sdp[3];
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = z4;
    sdp[0] = z4; sdp[1] = z4; sdp[2] = z4;

    SavedAct<CELL, 1> cur;
    f32x4 cnew = z4, hnew = z4;
    auto load_saved = [&](int t, SavedAct<CELL, 1>& d) {
        const size_t o = ((size_t)t * Bp + row) * HP + u0;
        d.hprev[0] = *(const f32x4*)&a.hs[o];
        if (CELL != CELL_VANILLA) {
            const size_t og = gate_index(t, row, u0, Bp, HP);
#pragma unroll
            for (int k = 0; k < 4; ++k) d.sv[0][k] = *(const f32x4*)&a.g[k][og];
        }
        if (CELL == CELL_LSTM) d.cprev[0] = *(const f32x4*)&a.cs[o];
    };
    bool have = false;
    unsigned long long p_c0 = 0, p_r0 = 0, p_work = 0, p_bar = 0, p_ta = 0;
    if (a.prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    __syncthreads();                                              // W plane 3 visible

    for (int t = a.t_hi - 1; t >= a.t_lo; --t) {
        if (a.prof) p_ta = clock64();
        if (a.dh_ext && live) dh += *(const f32x4*)&a.dh_ext[((size_t)t * Bp + row) * HP + u0];
        if (t >= tmax) {                                          // whole tile masked: zero rows
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const size_t o = ((size_t)t * Bp + row) * GHP + g * HP + u0;
                if (live) {
                    *(f32x4*)&a.dxt[o] = z4;
                    if (CELL == CELL_GRU && g == 2) *(f32x4*)&a.dhi[((size_t)t * Bp + row) * HP + u0] = z4;
                }
            }
            continue;
        }
        if (!have) {
            load_saved(t, cur);
            const size_t o1 = ((size_t)(t + 1) * Bp + row) * HP + u0;
            if (CELL == CELL_LSTM) cnew = *(const f32x4*)&a.cs[o1];
            if (CELL == CELL_VANILLA) hnew = *(const f32x4*)&a.hs[o1];
            have = true;
        }
        const bool m = t < mylen;
        const int prow = rpt * DROW;                              // bytes per plane
        char* lds = dbufp + (size_t)(dbuf ? (t & 1) : 0) * 3 * prow;
        {
            f32x4 vxi[G], vhi[G];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                if (CELL != CELL_VANILLA) { s[0] = cur.sv[0][0][e]; s[1] = cur.sv[0][1][e]; s[2] = cur.sv[0][2][e]; s[3] = cur.sv[0][3][e]; }
                float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
                float dhh = dh[e], dcc = dc[e];
                const float cpv = CELL == CELL_LSTM ? cur.cprev[0][e] : 0.f;
                cell_backward<CELL, true>(m, clip, dhh, dcc, s, cur.hprev[0][e], cpv, cnew[e], hnew[e], pi[e], pf[e], po[e],
                                          dxi, dhi, dp, a.relu != 0);
                dh[e] = dhh; dc[e] = dcc;
#pragma unroll
                for (int g = 0; g < G; ++g) { vxi[g][e] = dxi[g]; vhi[g][e] = dhi[g]; sdb[g][e] += dxi[g]; }
                sdp[0][e] += dp[0]; sdp[1][e] += dp[1]; sdp[2][e] += dp[2];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const size_t og = ((size_t)t * Bp + row) * GHP + g * HP + u0;
                if (live) {
                    *(f32x4*)&a.dxt[og] = vxi[g];
                    if (CELL == CELL_GRU && g == 2) *(f32x4*)&a.dhi[((size_t)t * Bp + row) * HP + u0] = vhi[g];
                }
                if (live) {
                    bf16x4 p1, p2, p3;
                    split3x4(vhi[g], p1, p2, p3);
                    char* base = lds + j * DROW + (g * HP + u0) * 2;
                    *(bf16x4*)(base) = p1; *(bf16x4*)(base + prow) = p2; *(bf16x4*)(base + 2 * prow) = p3;
                }
            }
        }
        // c_{t-1} / h_{t-1} double as c_t / h_t of the step below; then refill `cur` in place for step
        // t-1: the loads are in flight across the barrier and the whole MFMA phase
        if (CELL == CELL_LSTM) cnew = cur.cprev[0];
        if (CELL == CELL_VANILLA) hnew = cur.hprev[0];
        if (t > a.t_lo) load_saved(t - 1, cur);
        if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
        __syncthreads();
        if (a.prof) { const unsigned long long tc = clock64(); p_bar += tc - p_ta; p_ta = tc; }
        const char* db = lds + (j % rpt) * DROW + q * 16;
        f32x4 acc[3] = {z4, z4, z4};
        bf16x8 dp[2][3], wp[2];
        auto load_ops = [&](int kb, int s) {                     // see rec_fwd_x6: software-pipelined operand reads
            dp[s][0] = *(const bf16x8*)(db + kb * 64);
            dp[s][1] = *(const bf16x8*)(db + kb * 64 + prow);
            dp[s][2] = *(const bf16x8*)(db + kb * 64 + 2 * prow);
            wp[s] = *(const bf16x8*)(w3 + (kb * NW + wave) * 1024 + lane * 16);
        };
        load_ops(0, 0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < KB) load_ops(kb + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            // three independent accumulators break the MFMA dependency chain
            acc[0] = MFMA_BF16(wp[s], dp[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dp[s][2], acc[1]);
            acc[2] = MFMA_BF16(W2[kb], dp[s][1], acc[2]);
            acc[0] = MFMA_BF16(W2[kb], dp[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dp[s][1], acc[1]);
            acc[2] = MFMA_BF16(W1[kb], dp[s][0], acc[2]);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_nop 15");                                 // MFMA D -> VALU read hazard
        __builtin_amdgcn_s_setprio(3);                            // gate-math phase ahead: see rec_fwd_x6
        if (live) dh += acc[0] + acc[1] + acc[2];                 // phantom columns read a live row: keep their dh at 0
        if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
        if (!dbuf) __syncthreads();
        if (a.prof) p_bar += clock64() - p_ta;
    }
    if (a.prof && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_work; o[3] = p_bar;
    }

    if (!last && live) {                                          // hand dh / dc to the next chunk launch
        *(f32x4*)&a.state[(size_t)row * HP + u0] = dh;
        if (CELL == CELL_LSTM) *(f32x4*)&a.state[((size_t)Bp + row) * HP + u0] = dc;
    }
    float* part = a.part + ((size_t)a.chunk * gridDim.x + blockIdx.x) * (GHP + 5 * HP);
    f32x4 v[G + 5];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = sdb[g];
    v[G] = sdp[0]; v[G + 1] = sdp[1]; v[G + 2] = sdp[2];
    v[G + 3] = last ? dc : z4; v[G + 4] = last ? dh : z4;       // init-state gradients come from the last chunk only
#pragma unroll
    for (int k = 0; k < G + 5; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sum = v[k][e];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o);
            v[k][e] = sum;
        }
    if (j == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) *(f32x4*)&part[g * HP + u0] = v[g];
#pragma unroll
        for (int k = 0; k < 5; ++k) *(f32x4*)&part[GHP + k * HP + u0] = v[G + k];
    }
}

// ---------------------------------------------------------------------------------------
// bf16x6 kernels for 4-row tiles ("x6s": split gate math).  With rpt = 4 the MFMA columns 4..15 repeat rows
// 0..3, so the four lanes (r, q), (r+4, q), (r+8, q), (r+12, q) hold identical accumulators.  Instead of
// letting three of them idle, copy c = j >> 2 finishes unit c of the lane's four: every lane runs the cell
// math for ONE (row, unit) pair -- the VALU part of the per-step critical path shrinks 4x, and all loads and
// stores are 4 bytes per lane over 16 consecutive units of a row (full 64-byte segments).
// Same data layout, same LDS operand planes (4 rows), same chunked-BPTT protocol as rec_fwd_x6 / rec_bwd_x6.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float pick4(const f32x4 v, int c) {
    return c == 0 ? v[0] : (c == 1 ? v[1] : (c == 2 ? v[2] : v[3]));
}

template <int CELL, int HP, int NT = 1>
__global__ void __launch_bounds__(HP * 4 / NT) rec_fwd_x6s(RecArgs a) {
    // NT unit tiles per wave (only NT = 1 is launched).  Measured and rejected: NT = 2 at HP = 128 (4 waves, one per
    // SIMD, to remove the ~900-cycle skew between the two waves that share a matrix pipe): 414 registers put the
    // resident W planes in AGPRs, every MFMA operand then needs v_accvgpr_read copies, and a single wave issues one
    // MFMA per ~28 cycles instead of 16: 5280 cycles per step against 3760 with two waves per SIMD.
    constexpr int G = Gates<CELL>::G, KB = HP / 32, NW = HP / 16, NWV = NW / NT, GHP = G * HP, R = 4;
    constexpr int HROW = HP * 2 + 32, PLANEB = R * HROW;
    constexpr int W3_BYTES = G * KB * NW * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [G][KB][NW tiles][64 lanes][16 B]
    char* hbuf = smem_c + W3_BYTES;                      // [2][3 planes][R rows][HROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int r = j & 3, c = j >> 2;                     // tile row; which of the lane's 4 accumulator units it finishes
    const int row = blockIdx.x * R + r;
    const int T = a.T, Bp = a.Bp;
    int u[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) u[n] = (wave + n * NWV) * 16 + q * 4 + c;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 1));
    tmax = max(tmax, __shfl_xor(tmax, 2));

    bf16x8 W1[NT][G][KB], W2[NT][G][KB];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const int tile = wave + n * NWV;
                bf16x8 w3v;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    __bf16 b1, b2, b3;
                    split3(a.Whid[(size_t)(kb * 32 + 8 * q + e) * GHP + g * HP + tile * 16 + j], b1, b2, b3);
                    W1[n][g][kb][e] = b1; W2[n][g][kb][e] = b2; w3v[e] = b3;
                }
                *(bf16x8*)(w3 + ((g * KB + kb) * NW + tile) * 1024 + lane * 16) = w3v;
            }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    float h[NT], cst[NT], pi[NT], pf[NT], po[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        h[n] = a.hinit[u[n]]; cst[n] = 0.f; pi[n] = 0.f; pf[n] = 0.f; po[n] = 0.f;
        if (CELL == CELL_LSTM) {
            cst[n] = a.cinit[u[n]];
            pi[n] = a.peep[u[n]]; pf[n] = a.peep[HP + u[n]]; po[n] = a.peep[2 * HP + u[n]];
            a.cs[(size_t)row * HP + u[n]] = cst[n];
        }
        a.hs[(size_t)row * HP + u[n]] = h[n];
    }
    auto publish_h = [&](int buf) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            __bf16 p1, p2, p3;
            split3(h[n], p1, p2, p3);
            char* base = hbuf + (size_t)buf * 3 * PLANEB + r * HROW + u[n] * 2;
            *(__bf16*)(base) = p1; *(__bf16*)(base + PLANEB) = p2; *(__bf16*)(base + 2 * PLANEB) = p3;
        }
    };
    publish_h(0);

    // Input of step t: a row of xt, or (layer 0, one index per step) gathered here: W_in[id[row][t]] + b
    // (sparse_lstm.py:368 / :755 / :1111).  The id is fetched two steps ahead, the row one step ahead; the
    // prefetches are unconditional with clamped indices (a branch around them costs a vmcnt(0), see sbr_rec_cl.hip).
    const bool fuse = a.gX != nullptr;
    float x[NT][G], xn[NT][G], bias[NT][G];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int g = 0; g < G; ++g) bias[n][g] = fuse ? a.gbias[g * HP + u[n]] : 0.f;
    auto load_id = [&](int t) -> int { return fuse ? a.gX[(size_t)row * T + (t < T ? t : T - 1)] : 0; };
    auto load_x = [&](int t, int id, float (&d)[NT][G]) {
        const float* src = fuse ? a.gWin + (size_t)id * GHP : a.xt + ((size_t)(t < T ? t : T - 1) * Bp + row) * GHP;
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g) d[n][g] = src[g * HP + u[n]];
    };
    int id_next = load_id(1), id_nn = 0;
    load_x(0, load_id(0), x);
    __syncthreads();
    unsigned long long p_c0 = 0, p_r0 = 0, p_work = 0, p_bar = 0, p_ta = 0;
    if (a.prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }

    // The values a step saves for BPTT (gate activations, h_t, c_t) are stored from INSIDE the next step's MFMA
    // stream (after its first k-block): address math and store issue then hide under the matrix pipe instead of
    // sitting between the barrier and the first MFMA (~650 cycles per step measured).  h / cst / sv still hold
    // step t-1's values there: the gate math that overwrites them comes after the MFMA loop.
    float sv[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int k = 0; k < 4; ++k) sv[n][k] = 0.f;
    auto store_step = [&](int t) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (CELL != CELL_VANILLA) {
                const size_t o = gate_index(t, row, u[n], Bp, HP);
#pragma unroll
                for (int k = 0; k < 4; ++k) a.g[k][o] = sv[n][k];
            }
            const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u[n];
            a.hs[o] = h[n];
            if (CELL == CELL_LSTM) a.cs[o] = cst[n];
        }
    };
    for (int t = 0; t < tmax; ++t) {                              // tmax is workgroup-uniform
        if (a.prof) p_ta = clock64();
        load_x(t + 1, id_next, xn);
        id_nn = load_id(t + 2);
        const char* hb = hbuf + (size_t)(t & 1) * 3 * PLANEB + r * HROW + q * 16;
        f32x4 acc[NT][G];
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[n][g] = f32x4{0, 0, 0, 0};
        bf16x8 hp[2][3], wp[2][NT][G];
        auto load_ops = [&](int kb, int s) {
            hp[s][0] = *(const bf16x8*)(hb + kb * 64);
            hp[s][1] = *(const bf16x8*)(hb + kb * 64 + PLANEB);
            hp[s][2] = *(const bf16x8*)(hb + kb * 64 + 2 * PLANEB);
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    wp[s][n][g] = *(const bf16x8*)(w3 + ((g * KB + kb) * NW + wave + n * NWV) * 1024 + lane * 16);
        };
        load_ops(0, 0);
        __builtin_amdgcn_s_setprio(0);
#define X6S_TERM(WOP, HOP) _Pragma("unroll") for (int n = 0; n < NT; ++n) _Pragma("unroll") for (int g = 0; g < G; ++g) \
            acc[n][g] = MFMA_BF16(WOP, HOP, acc[n][g]);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < KB) load_ops(kb + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            X6S_TERM(wp[s][n][g], hp[s][0])
            X6S_TERM(W1[n][g][kb], hp[s][2])
            X6S_TERM(W2[n][g][kb], hp[s][1])
            X6S_TERM(W2[n][g][kb], hp[s][0])
            X6S_TERM(W1[n][g][kb], hp[s][1])
            X6S_TERM(W1[n][g][kb], hp[s][0])
            __builtin_amdgcn_sched_barrier(0);
            if (kb == 0 && t > 0) { store_step(t - 1); __builtin_amdgcn_sched_barrier(0); }
        }
#undef X6S_TERM
        asm volatile("s_nop 15");                                 // MFMA D -> VALU read hazard (see rec_fwd_mfma)
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            float as[G], xb[G];
#pragma unroll
            for (int g = 0; g < G; ++g) { as[g] = pick4(acc[n][g], c); xb[g] = x[n][g] + bias[n][g]; }
            cell_forward<CELL, true>(xb, as, t < mylen, h[n], cst[n], pi[n], pf[n], po[n], sv[n], a.relu != 0);
        }
        if (t + 1 < tmax) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int g = 0; g < G; ++g) x[n][g] = xn[n][g];
            id_next = id_nn;
            publish_h((t + 1) & 1);
            if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
            __syncthreads();
            if (a.prof) p_bar += clock64() - p_ta;
        }
    }
    if (tmax > 0) store_step(tmax - 1);
    for (int t = tmax; t < T; ++t) {                              // past the tile's longest row: the state is carried
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u[n];
            a.hs[o] = h[n];
            if (CELL == CELL_LSTM) a.cs[o] = cst[n];
        }
    }
    if (a.prof && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_work; o[3] = p_bar;
    }
}

template <int CELL, int HP>
__global__ void __launch_bounds__(HP * 4) rec_bwd_x6s(RecArgs a, int dbuf) {
    constexpr int G = Gates<CELL>::G, NW = HP / 16, GHP = G * HP, KB = GHP / 32, R = 4;
    constexpr int DROW = GHP * 2 + 32, PLANEB = R * DROW;
    constexpr int W3_BYTES = KB * NW * 1024;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* w3 = smem_c;                                   // [KB][NW][64][16 B]
    char* dbufp = smem_c + W3_BYTES;                     // [dbuf ? 2 : 1][3][R][DROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int r = j & 3, c = j >> 2;
    const int row = blockIdx.x * R + r;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    const int u = wave * 16 + q * 4 + c;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 1));
    tmax = max(tmax, __shfl_xor(tmax, 2));

    bf16x8 W1[KB], W2[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float* src = a.Whid + (size_t)(wave * 16 + j) * GHP + kb * 32 + 8 * q;
        const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
        bf16x8 w3v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            __bf16 b1, b2, b3;
            split3(e < 4 ? lo[e & 3] : hi[e & 3], b1, b2, b3);
            W1[kb][e] = b1; W2[kb][e] = b2; w3v[e] = b3;
        }
        *(bf16x8*)(w3 + (kb * NW + wave) * 1024 + lane * 16) = w3v;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);

    const f32x4 z4 = f32x4{0, 0, 0, 0};
    const bool first = a.t_hi >= T, last = a.t_lo <= 0;          // first / last launch of the chunked chain
    float dh = 0.f, dc = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (first) { if (a.dh_last) dh = a.dh_last[(size_t)row * HP + u]; }
    else {
        dh = a.state[(size_t)row * HP + u];
        if (CELL == CELL_LSTM) dc = a.state[((size_t)Bp + row) * HP + u];
    }
    if (CELL == CELL_LSTM) { pi = a.peep[u]; pf = a.peep[HP + u]; po = a.peep[2 * HP + u]; }
    float sdb[G], sdp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = 0.f;

    float sv[4] = {0.f, 0.f, 0.f, 0.f}, hprev = 0.f, cprev = 0.f, cnew = 0.f, hnew = 0.f;
    auto load_saved = [&](int t) {
        const size_t o = ((size_t)t * Bp + row) * HP + u;
        hprev = a.hs[o];
        if (CELL != CELL_VANILLA) {
            const size_t og = gate_index(t, row, u, Bp, HP);
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = a.g[k][og];
        }
        if (CELL == CELL_LSTM) cprev = a.cs[o];
    };
    unsigned long long p_c0 = 0, p_r0 = 0, p_work = 0, p_bar = 0, p_ta = 0;
    if (a.prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    __syncthreads();                                              // W plane 3 visible

    const int t_live = min(a.t_hi, tmax);                         // steps [t_live, t_hi) are masked for the whole tile
    for (int t = a.t_hi - 1; t >= max(t_live, a.t_lo); --t) {     // zero rows; dh_ext still accumulates
        if (a.dh_ext) dh += a.dh_ext[((size_t)t * Bp + row) * HP + u];
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u] = 0.f;
        if (CELL == CELL_GRU) a.dhi[((size_t)t * Bp + row) * HP + u] = 0.f;
    }
    if (t_live > a.t_lo) {
        load_saved(t_live - 1);
        const size_t o1 = ((size_t)t_live * Bp + row) * HP + u;
        if (CELL == CELL_LSTM) cnew = a.cs[o1];
        if (CELL == CELL_VANILLA) hnew = a.hs[o1];
    }
    for (int t = t_live - 1; t >= a.t_lo; --t) {
        if (a.prof) p_ta = clock64();
        if (a.dh_ext) dh += a.dh_ext[((size_t)t * Bp + row) * HP + u];
        char* lds = dbufp + (size_t)(dbuf ? (t & 1) : 0) * 3 * PLANEB;
        float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
        cell_backward<CELL, true>(t < mylen, clip, dh, dc, sv, hprev, cprev, cnew, hnew, pi, pf, po, dxi, dhi, dp, a.relu != 0);
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[g] += dxi[g];
        sdp[0] += dp[0]; sdp[1] += dp[1]; sdp[2] += dp[2];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            __bf16 p1, p2, p3;
            split3(dhi[g], p1, p2, p3);
            char* base = lds + r * DROW + (g * HP + u) * 2;
            *(__bf16*)(base) = p1; *(__bf16*)(base + PLANEB) = p2; *(__bf16*)(base + 2 * PLANEB) = p3;
        }
        if (CELL == CELL_LSTM) cnew = cprev;
        if (CELL == CELL_VANILLA) hnew = hprev;
        // saved activations of step t-1: unconditional (clamped) and fenced, see sbr_rec_cl.hip on in-order vmcnt
        __builtin_amdgcn_sched_barrier(0);
        load_saved(t > a.t_lo ? t - 1 : t);
        __builtin_amdgcn_sched_barrier(0);
        if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
        __syncthreads();
        if (a.prof) { const unsigned long long tc = clock64(); p_bar += tc - p_ta; p_ta = tc; }
        const char* db = lds + r * DROW + q * 16;
        f32x4 acc[3] = {z4, z4, z4};
        bf16x8 dpl[2][3], wp[2];
        auto load_ops = [&](int kb, int s) {
            dpl[s][0] = *(const bf16x8*)(db + kb * 64);
            dpl[s][1] = *(const bf16x8*)(db + kb * 64 + PLANEB);
            dpl[s][2] = *(const bf16x8*)(db + kb * 64 + 2 * PLANEB);
            wp[s] = *(const bf16x8*)(w3 + (kb * NW + wave) * 1024 + lane * 16);
        };
        load_ops(0, 0);
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int s = kb & 1;
            if (kb + 1 < KB) load_ops(kb + 1, s ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = MFMA_BF16(wp[s], dpl[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dpl[s][2], acc[1]);
            acc[2] = MFMA_BF16(W2[kb], dpl[s][1], acc[2]);
            acc[0] = MFMA_BF16(W2[kb], dpl[s][0], acc[0]);
            acc[1] = MFMA_BF16(W1[kb], dpl[s][1], acc[1]);
            acc[2] = MFMA_BF16(W1[kb], dpl[s][0], acc[2]);
            __builtin_amdgcn_sched_barrier(0);
            if (kb == 0) {
                // this step's outputs leave from inside the MFMA stream (see rec_fwd_x6s): off the gate-math ->
                // publish -> barrier critical section, hidden under the matrix pipe
#pragma unroll
                for (int g = 0; g < G; ++g) a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u] = dxi[g];
                if (CELL == CELL_GRU) a.dhi[((size_t)t * Bp + row) * HP + u] = dhi[2];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_nop 15");                                 // MFMA D -> VALU read hazard
        __builtin_amdgcn_s_setprio(3);
        dh += pick4(acc[0] + acc[1] + acc[2], c);
        if (a.prof) { const unsigned long long tc = clock64(); p_work += tc - p_ta; p_ta = tc; }
        if (!dbuf) __syncthreads();
        if (a.prof) p_bar += clock64() - p_ta;
    }
    if (a.prof && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_work; o[3] = p_bar;
    }

    if (!last) {                                                  // hand dh / dc to the next chunk launch
        a.state[(size_t)row * HP + u] = dh;
        if (CELL == CELL_LSTM) a.state[((size_t)Bp + row) * HP + u] = dc;
    }
    float* part = a.part + ((size_t)a.chunk * gridDim.x + blockIdx.x) * (GHP + 5 * HP);
    float v[G + 5];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = sdb[g];
    v[G] = sdp[0]; v[G + 1] = sdp[1]; v[G + 2] = sdp[2];
    v[G + 3] = last ? dc : 0.f; v[G + 4] = last ? dh : 0.f;      // init-state gradients come from the last chunk only
#pragma unroll
    for (int k = 0; k < G + 5; ++k) {
        float sum = v[k];
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);                                // over the 4 rows (lanes j, j+4, .. hold other units)
        v[k] = sum;
    }
    if (r == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) part[g * HP + u] = v[g];
#pragma unroll
        for (int k = 0; k < 5; ++k) part[GHP + k * HP + u] = v[G + k];
    }
}

// ---------------------------------------------------------------------------------------
// Triage ("simple") kernels: one launch per time step, one thread per (row, unit), plain FMAs.
// Same scalar cell math, none of the MFMA/LDS machinery (SBR_FLAG_SIMPLE_REC).
// ---------------------------------------------------------------------------------------
template <int CELL>
__global__ void rec_fwd_step_simple(RecArgs a, int t) {
    constexpr int G = Gates<CELL>::G;
    const int Hp = a.Hp, GHp = G * Hp, Bp = a.Bp;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bp * Hp) return;
    const int row = idx / Hp, k = idx % Hp;
    const size_t o0 = ((size_t)t * Bp + row) * Hp, o1 = ((size_t)(t + 1) * Bp + row) * Hp;
    float h, c = 0.f;
    if (t == 0) {
        h = a.hinit[k]; a.hs[(size_t)row * Hp + k] = h;
        if (CELL == CELL_LSTM) { c = a.cinit[k]; a.cs[(size_t)row * Hp + k] = c; }
    } else {
        h = a.hs[o0 + k];
        if (CELL == CELL_LSTM) c = a.cs[o0 + k];
    }
    float acc[G], x[G], s[4];
    for (int g = 0; g < G; ++g) {
        float v = 0.f;
        for (int kk = 0; kk < Hp; ++kk) {
            const float hv = t == 0 ? a.hinit[kk] : a.hs[o0 + kk];
            v = fmaf(hv, a.Whid[(size_t)kk * GHp + g * Hp + k], v);
        }
        acc[g] = v; x[g] = a.xt[((size_t)t * Bp + row) * GHp + g * Hp + k];
    }
    float pi = 0.f, pf = 0.f, po = 0.f;
    if (CELL == CELL_LSTM) { pi = a.peep[k]; pf = a.peep[Hp + k]; po = a.peep[2 * Hp + k]; }
    cell_forward<CELL>(x, acc, t < a.len[row], h, c, pi, pf, po, s, a.relu != 0);
    if (CELL != CELL_VANILLA)
        for (int q = 0; q < 4; ++q) a.g[q][gate_index(t, row, k, Bp, Hp)] = s[q];
    a.hs[o1 + k] = h;
    if (CELL == CELL_LSTM) a.cs[o1 + k] = c;
}

// elementwise part of one backward step; dhstate/dcstate [Bp][Hp] carry dh, dc between launches
template <int CELL>
__global__ void rec_bwd_elem_simple(RecArgs a, int t, float* dhstate, float* dcstate) {
    constexpr int G = Gates<CELL>::G;
    const int Hp = a.Hp, GHp = G * Hp, Bp = a.Bp;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bp * Hp) return;
    const int row = idx / Hp, k = idx % Hp;
    const size_t o0 = ((size_t)t * Bp + row) * Hp + k, o1 = ((size_t)(t + 1) * Bp + row) * Hp + k;
    float dh = dhstate[idx], dc = dcstate[idx];
    if (t == a.T - 1) { dh = a.dh_last ? a.dh_last[idx] : 0.f; dc = 0.f; }
    if (a.dh_ext) dh += a.dh_ext[o0];
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (CELL != CELL_VANILLA)
        for (int q = 0; q < 4; ++q) s[q] = a.g[q][gate_index(t, row, k, Bp, Hp)];
    float cprev = 0.f, cnew = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (CELL == CELL_LSTM) { cprev = a.cs[o0]; cnew = a.cs[o1]; pi = a.peep[k]; pf = a.peep[Hp + k]; po = a.peep[2 * Hp + k]; }
    float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
    cell_backward<CELL>(t < a.len[row], a.clip, dh, dc, s, a.hs[o0], cprev, cnew, a.hs[o1], pi, pf, po, dxi, dhi, dp, a.relu != 0);
    for (int g = 0; g < G; ++g) {
        const size_t og = ((size_t)t * Bp + row) * GHp + g * Hp + k;
        a.dxt[og] = dxi[g];
        if (CELL == CELL_GRU && g == 2) a.dhi[((size_t)t * Bp + row) * Hp + k] = dhi[g];
        atomicAdd(&a.part[g * Hp + k], dxi[g]);
    }
    if (CELL == CELL_LSTM) {
        atomicAdd(&a.part[GHp + k], dp[0]); atomicAdd(&a.part[GHp + Hp + k], dp[1]); atomicAdd(&a.part[GHp + 2 * Hp + k], dp[2]);
    }
    dhstate[idx] = dh; dcstate[idx] = dc;
}

template <int CELL>
__global__ void rec_bwd_matvec_simple(RecArgs a, int t, float* dhstate, float* dcstate) {
    constexpr int G = Gates<CELL>::G;
    const int Hp = a.Hp, GHp = G * Hp, Bp = a.Bp;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Bp * Hp) return;
    const int row = idx / Hp, k = idx % Hp;
    // grad wrt hid_input: LSTM/Vanilla = dxt; GRU = [dxt_r, dxt_u, dhi_c] (only the candidate slice differs)
    const float* d = a.dxt + ((size_t)t * Bp + row) * GHp;
    const float* dc2 = a.dhi + ((size_t)t * Bp + row) * Hp;
    float v = 0.f;
    for (int jj = 0; jj < GHp; ++jj) {
        const float dv = (CELL == CELL_GRU && jj >= 2 * Hp) ? dc2[jj - 2 * Hp] : d[jj];
        v = fmaf(dv, a.Whid[(size_t)k * GHp + jj], v);
    }
    const float dh = dhstate[idx] + v;
    dhstate[idx] = dh;
    if (t == 0) {   // init-state gradients: column sums over rows
        atomicAdd(&a.part[GHp + 4 * Hp + k], dh);
        if (CELL == CELL_LSTM) atomicAdd(&a.part[GHp + 3 * Hp + k], dcstate[idx]);
    }
}

// 64 outputs per workgroup x 4 slices of the block range; fixed summation order (deterministic)
__global__ void __launch_bounds__(256) rec_reduce_partials(const float* __restrict__ part, int nblk, int G, int Hp,
                                                           float* db, float* dpeep, float* dcinit, float* dhinit) {
    __shared__ float red[4][64];
    const int GHp = G * Hp, n = GHp + 5 * Hp;
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < n) {
        int b = grp;
        for (; b + 12 < nblk; b += 16) {                       // 4 independent loads in flight
            s0 += part[(size_t)b * n + e]; s1 += part[(size_t)(b + 4) * n + e];
            s2 += part[(size_t)(b + 8) * n + e]; s3 += part[(size_t)(b + 12) * n + e];
        }
        for (; b < nblk; b += 4) s0 += part[(size_t)b * n + e];
    }
    red[grp][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (grp != 0 || e >= n) return;
    const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
    if (e < GHp) { db[e] = s; return; }
    const int k = (e - GHp) / Hp, u = (e - GHp) % Hp;
    if (k < 3) { if (dpeep) dpeep[k * Hp + u] = s; }
    else if (k == 3) { if (dcinit) dcinit[u] = s; }
    else dhinit[u] = s;
}

hipError_t launch_rec_reduce_partials(hipStream_t s, const float* part, int nblk, int G, int Hp, int cell, float* db,
                                      float* dpeep, float* dcinit, float* dhinit) {
    const int n = G * Hp + 5 * Hp;
    const bool lstm = cell == SBR_CELL_LSTM;
    rec_reduce_partials<<<(n + 63) / 64, 256, 0, s>>>(part, nblk, G, Hp, db, lstm ? dpeep : nullptr,
                                                        lstm ? dcinit : nullptr, dhinit);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------------------
template <int CELL>
static hipError_t launch_fwd_cell(hipStream_t s, const RecArgs& a, bool simple) {
    const int Hp = a.Hp;
    if (simple) {
        const int n = a.Bp * Hp, blk = 256, grid = (n + blk - 1) / blk;
        for (int t = 0; t < a.T; ++t) rec_fwd_step_simple<CELL><<<grid, blk, 0, s>>>(a, t);
        return hipGetLastError();
    }
    if (sbr_rec_cluster_ok(a)) return launch_rec_forward_cl(s, a);
    if (sbr_rec_x6p_ok(a)) return launch_rec_forward_x6p(s, a);
    if (sbr_rec_x6q_ok(a)) return launch_rec_forward_x6q(s, a);
    const int nblk = a.Bp / 16;
#define LAUNCH_DYN(KERNEL, GRID, BLOCK, LDS, ...) do { \
        SBR_DYN_LDS(KERNEL, (LDS)); \
        KERNEL<<<GRID, BLOCK, LDS, s>>>(__VA_ARGS__); } while (0)
    if (!a.f32_mfma && (Hp == 32 || Hp == 64 || Hp == 128)) {
        const size_t l6 = (size_t)Gates<CELL>::G * (Hp / 32) * (Hp / 16) * 1024 + 2 * 3 * 16 * (size_t)(Hp * 2 + 32);
        if (l6 <= 160 * 1024) {
            const int nb6 = a.Bp / a.rpt;
            if (a.rpt == 4 && a.x6_split) {   // 4-row tiles: one (row, unit) pair per lane
                const size_t l4 = (size_t)Gates<CELL>::G * (Hp / 32) * (Hp / 16) * 1024 + 2 * 3 * 4 * (size_t)(Hp * 2 + 32);
                if (Hp == 32) LAUNCH_DYN((rec_fwd_x6s<CELL, 32>), nb6, Hp * 4, l4, a);
                else if (Hp == 64) LAUNCH_DYN((rec_fwd_x6s<CELL, 64>), nb6, Hp * 4, l4, a);
                else LAUNCH_DYN((rec_fwd_x6s<CELL, 128>), nb6, Hp * 4, l4, a);
                return hipGetLastError();
            }
            if (Hp == 32) LAUNCH_DYN((rec_fwd_x6<CELL, 32>), nb6, Hp * 4, l6, a);
            else if (Hp == 64) LAUNCH_DYN((rec_fwd_x6<CELL, 64>), nb6, Hp * 4, l6, a);
            else LAUNCH_DYN((rec_fwd_x6<CELL, 128>), nb6, Hp * 4, l6, a);
            return hipGetLastError();
        }
    }
    const size_t lds = 2 * 16 * (size_t)(4 * ((Hp / 4 + 63) / 64 * 64) + 4) * sizeof(float);
#define FWD_RES(KS) LAUNCH_DYN((rec_fwd_mfma<CELL, 1, KS>), nblk, (Hp / 16) * 64, lds, a)
    if (Hp == 16) FWD_RES(4);
    else if (Hp == 32) FWD_RES(8);
    else if (Hp == 64) FWD_RES(16);
    else if (Hp == 128) FWD_RES(32);
    else {
        const int tiles = Hp / 16;
        if (tiles <= 16) LAUNCH_DYN((rec_fwd_mfma<CELL, 1, 0>), nblk, tiles * 64, lds, a);
        else if (tiles <= 32) LAUNCH_DYN((rec_fwd_mfma<CELL, 2, 0>), nblk, (tiles / 2) * 64, lds, a);
        else if (tiles <= 64) LAUNCH_DYN((rec_fwd_mfma<CELL, 4, 0>), nblk, (tiles / 4) * 64, lds, a);
        else return hipErrorInvalidValue;
    }
#undef FWD_RES
    return hipGetLastError();
}

bool sbr_rec_fwd_can_fuse_gather(const RecArgs& a, bool simple) {
    const int Hp = a.Hp;
    if (simple || a.f32_mfma) return false;
    if (sbr_rec_cluster_ok(a)) return true;               // rec_fwd_cl gathers its own rows too
    if (!(Hp == 32 || Hp == 64 || Hp == 128)) return false;
    if (sbr_rec_x6p_ok(a) && !sbr_rec_x6p_fuse_ok(a)) return false;    // rec_fwd_x6p: row offsets in LDS (T), 32-bit offsets into W_in (n_in)
    const size_t l4 = (size_t)a.G * (Hp / 32) * (Hp / 16) * 1024 + 2 * 3 * 4 * (size_t)(Hp * 2 + 32);
    return a.rpt == 4 && a.x6_split && l4 <= 160 * 1024;
}

hipError_t launch_rec_forward(hipStream_t s, const RecArgs& a, bool simple) {
    switch (a.cell) {
        case SBR_CELL_LSTM: return launch_fwd_cell<CELL_LSTM>(s, a, simple);
        case SBR_CELL_GRU: return launch_fwd_cell<CELL_GRU>(s, a, simple);
        default: return launch_fwd_cell<CELL_VANILLA>(s, a, simple);
    }
}

template <int CELL>
static hipError_t launch_bwd_cell(hipStream_t s, const RecArgs& a, bool simple) {
    const int Hp = a.Hp, G = Gates<CELL>::G, GHp = G * Hp;
    const int nblk = a.Bp / 16;
    if (simple) {
        // part block 0 accumulates (atomics); other blocks stay zero
        hipError_t e = hipMemsetAsync(a.part, 0, (size_t)nblk * (GHp + 5 * Hp) * sizeof(float), s);
        if (e != hipSuccess) return e;
        float* st = nullptr;   // dh/dc carry: reuse the tail of the dhi/dxt-independent scratch: allocate ad hoc
        e = hipMalloc(&st, (size_t)2 * a.Bp * Hp * sizeof(float));
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(st, 0, (size_t)2 * a.Bp * Hp * sizeof(float), s);
        const int n = a.Bp * Hp, blk = 256, grid = (n + blk - 1) / blk;
        for (int t = a.T - 1; t >= 0; --t) {
            rec_bwd_elem_simple<CELL><<<grid, blk, 0, s>>>(a, t, st, st + (size_t)a.Bp * Hp);
            rec_bwd_matvec_simple<CELL><<<grid, blk, 0, s>>>(a, t, st, st + (size_t)a.Bp * Hp);
        }
        e = hipGetLastError();
        (void)hipStreamSynchronize(s);
        (void)hipFree(st);
        return e;
    }
    if (sbr_rec_cluster_ok(a)) return launch_rec_backward_cl(s, a);
    if (sbr_rec_x6p_ok(a)) return launch_rec_backward_x6p(s, a);
    if (sbr_rec_x6q_ok(a)) return launch_rec_backward_x6q(s, a);
    if (!a.f32_mfma && (Hp == 32 || Hp == 64 || Hp == 128)) {
        const size_t w3b = (size_t)(GHp / 32) * (Hp / 16) * 1024, one6 = 3 * (size_t)a.rpt * (GHp * 2 + 32);
        const int db6 = (w3b + 2 * one6 <= 160 * 1024) ? 1 : 0;
        const size_t l6 = w3b + (db6 ? 2 : 1) * one6;
        if (l6 <= 160 * 1024) {
            const int nb6 = a.Bp / a.rpt;
            if (a.rpt == 4 && a.x6_split) {
                if (Hp == 32) LAUNCH_DYN((rec_bwd_x6s<CELL, 32>), nb6, Hp * 4, l6, a, db6);
                else if (Hp == 64) LAUNCH_DYN((rec_bwd_x6s<CELL, 64>), nb6, Hp * 4, l6, a, db6);
                else LAUNCH_DYN((rec_bwd_x6s<CELL, 128>), nb6, Hp * 4, l6, a, db6);
                return hipGetLastError();
            }
            if (Hp == 32) LAUNCH_DYN((rec_bwd_x6<CELL, 32>), nb6, Hp * 4, l6, a, db6);
            else if (Hp == 64) LAUNCH_DYN((rec_bwd_x6<CELL, 64>), nb6, Hp * 4, l6, a, db6);
            else LAUNCH_DYN((rec_bwd_x6<CELL, 128>), nb6, Hp * 4, l6, a, db6);
            return hipGetLastError();
        }
    }
    const size_t one = 16 * (size_t)(4 * ((GHp / 4 + 63) / 64 * 64) + 4) * sizeof(float);
    const int dbuf = (2 * one <= 150 * 1024) ? 1 : 0;
    const size_t lds = dbuf ? 2 * one : one;
#define BWD_RES(KS) LAUNCH_DYN((rec_bwd_mfma<CELL, 1, KS>), nblk, (Hp / 16) * 64, lds, a, dbuf)
    if (Hp == 16) BWD_RES(4);
    else if (Hp == 32) BWD_RES(8);
    else if (Hp == 64) BWD_RES(16);
    else if (Hp == 128) BWD_RES(32);
    else {
        const int tiles = Hp / 16;
        if (tiles <= 16) LAUNCH_DYN((rec_bwd_mfma<CELL, 1, 0>), nblk, tiles * 64, lds, a, dbuf);
        else if (tiles <= 32) LAUNCH_DYN((rec_bwd_mfma<CELL, 2, 0>), nblk, (tiles / 2) * 64, lds, a, dbuf);
        else if (tiles <= 64) LAUNCH_DYN((rec_bwd_mfma<CELL, 4, 0>), nblk, (tiles / 4) * 64, lds, a, dbuf);
        else return hipErrorInvalidValue;
    }
#undef BWD_RES
    return hipGetLastError();
}

static bool uses_x6_bwd(const RecArgs& a) {
    const int Hp = a.Hp, GHp = a.G * Hp;
    if (a.f32_mfma || !(Hp == 32 || Hp == 64 || Hp == 128)) return false;
    const size_t w3b = (size_t)(GHp / 32) * (Hp / 16) * 1024, one6 = 3 * (size_t)a.rpt * (GHp * 2 + 32);
    return w3b + one6 <= 160 * 1024;
}
int sbr_rec_bwd_blocks(const RecArgs& a, bool simple) {
    if (simple) return a.Bp / 16;                 // simple kernels accumulate into block 0, others zeroed
    if (sbr_rec_cluster_ok(a)) return a.Bp / sbr_rec_cluster_bwd_rows(a);
    return uses_x6_bwd(a) ? a.Bp / a.rpt : a.Bp / 16;
}

bool sbr_rec_bwd_chunkable(const RecArgs& a, bool simple) { return !simple && uses_x6_bwd(a); }

hipError_t launch_rec_backward(hipStream_t s, const RecArgs& a, bool simple) {
    switch (a.cell) {
        case SBR_CELL_LSTM: return launch_bwd_cell<CELL_LSTM>(s, a, simple);
        case SBR_CELL_GRU: return launch_bwd_cell<CELL_GRU>(s, a, simple);
        default: return launch_bwd_cell<CELL_VANILLA>(s, a, simple);
    }
}
