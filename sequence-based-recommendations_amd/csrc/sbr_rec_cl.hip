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
#include "sbr_cell.h"
#include <type_traits>
#include <cstdlib>

#define CL_SENT 0xFFFFFFFFu
#define CL_SPIN_LIMIT 400000

typedef unsigned long long u64;

__device__ __forceinline__ u64 cl_load(const float* p) {
    return __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// fast = every member of the cluster runs on the same XCC: one L2 is the coherence point, an ordinary store
// (L1 is write-through) is visible to the members' L1-bypassing loads as soon as it reaches that L2.
// Otherwise the store must write through to memory (sc1): measured ~3000-6000 cycles more per step.
__device__ __forceinline__ void cl_store4(float* p, const f32x4 v, bool fast) {
    if (fast) { *(f32x4*)p = v; return; }
    union { float f[2]; u64 u; } a, b;
    a.f[0] = v[0]; a.f[1] = v[1]; b.f[0] = v[2]; b.f[1] = v[3];
    __hip_atomic_store((u64*)p, a.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((u64*)(p + 2), b.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool cl_has_sentinel(u64 v) {
    return (unsigned)v == CL_SENT || (unsigned)(v >> 32) == CL_SENT;
}

// cluster / member of this workgroup; false = padding workgroup (no tile)
__device__ __forceinline__ bool cl_ids(const RecArgs& a, int C, int ntiles, int& tile, int& m) {
    const int bid = blockIdx.x;
    if (a.cl_linear) { tile = bid / C; m = bid % C; }             // (experiment) members on consecutive ids = different XCDs
    else { const int x = bid & 7, y = bid >> 3; m = y % C; tile = (y / C) * 8 + x; }
    return tile < ntiles;
}

// Start-of-launch handshake: every member publishes the XCC it runs on (HW_REG_XCC_ID) and reads the others'.
// Returns true when the whole cluster shares one XCC (the placement in the header makes that the normal case;
// nothing breaks when it does not hold -- the kernels then publish with write-through stores).
__device__ __forceinline__ bool cl_same_xcc(const RecArgs& a, int C, int tile, int mem, int* lds_flag, bool& dead) {
    const int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11));           // hwreg(HW_REG_XCC_ID, 0, 4)
    int* slots = a.clx + (size_t)tile * C;
    if (threadIdx.x == 0) {
        *lds_flag = 1;
        __hip_atomic_store(&slots[mem], (a.epoch << 4) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        int v = 0, tries = 0;
        while (true) {
            v = __hip_atomic_load(&slots[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 4) == a.epoch) break;
            if (++tries > CL_SPIN_LIMIT) { dead = true; atomicOr(a.fault, 1); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if ((v & 15) != xcc || (v >> 4) != a.epoch) *lds_flag = 0;
    }
    __syncthreads();
    const bool same = *lds_flag != 0;
    __syncthreads();
    return same;
}

// Polls NP 16-byte pieces per thread; piece p covers floats [4*c4, 4*c4+3] of tile row r, where
// p = tid + i*256, r = p / (W/4), c4 = p % (W/4).  src(r, col) returns the address.
// fast (whole cluster on one XCC): 16-byte sc1 loads (bypass the CU's L1, served by the shared L2).  hipcc lowers
// agent-scope atomic loads to sc1 only up to 8 bytes (0.54-0.70x the 16-byte rate), hence the inline asm: these
// loads are invisible to the compiler's waitcnt insertion, so the wait is explicit and the values are re-defined
// after it to pin their uses behind it.  (Tried and rejected: sc0 loads hit the stale L1 line; an L1 invalidate
// per poll, buffer_inv sc1, costs ~15000 cycles.)
template <int NP, typename SRC>
__device__ __forceinline__ int cl_fetch(f32x4 (&v)[NP], SRC src, int W, bool fast, bool& dead, int* fault) {
    const int tid = threadIdx.x;
    int tries = 0;
    while (true) {
        if (fast) {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int p = tid + i * 256, r = p / (W >> 2), c4 = p % (W >> 2);
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(src(r, 4 * c4)) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(v[i]));
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int p = tid + i * 256, r = p / (W >> 2), c4 = p % (W >> 2);
                const float* q = src(r, 4 * c4);
                union { u64 u[2]; f32x4 f; } x;
                x.u[0] = cl_load(q); x.u[1] = cl_load(q + 2);
                v[i] = x.f;
            }
        }
        unsigned mx = 0u;                                // the sentinel is the largest 32-bit pattern: one v_max3_u32 per two words
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = max(mx, __float_as_uint(v[i][e]));
        const bool ok = mx != CL_SENT;
        if (ok || dead) break;
        if (++tries > CL_SPIN_LIMIT) { dead = true; atomicOr(fault, 1); break; }   // bounded: never hang the GPU
        __builtin_amdgcn_s_sleep(1);
    }
    return tries;
}

// "f16x3" (sbr_rec_p.hip, split2_f16): an operand as a1 + a2 / 2048 in two fp16 planes, a product in three MFMAs.  Forward: h is
// in [-1, 1] unless the layer rectifies; backward: dhi has passed the reference's gradient clip (<= 100), scaled by 2^9.
typedef _Float16 f16x8c __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4c __attribute__((ext_vector_type(4)));
constexpr float CL_F16_LO = 2048.0f, CL_F16_DSCALE = 512.0f;
__device__ __forceinline__ f32x4 cl_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) { return MFMA_BF16(a, b, c); }
__device__ __forceinline__ f32x4 cl_mfma(const f16x8c& a, const f16x8c& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void cl_split2(float v, _Float16& a1, _Float16& a2) {
    asm("" : "+v"(v));                                   // one rounding to fp16 for both uses (see split2_f16)
    a1 = (_Float16)v;
    a2 = (_Float16)((v - (float)a1) * CL_F16_LO);
}
// bool switches of the two chains, read per launch (the tests flip them); the same conditions as the 128-unit kernels
static bool cl_f16_fwd(const RecArgs& a) {
    const char* fe = getenv("SBR_X6_F16");
    return (fe ? atoi(fe) != 0 : true) && !a.relu;
}
static bool cl_f16_bwd(const RecArgs& a) {
    const char* fe = getenv("SBR_X6_F16_BWD");
    return (fe ? atoi(fe) != 0 : true) && a.clip > 0.0f && a.clip <= 100.0f;
}

// splits the fetched pieces into three bf16 planes (F16: two fp16 planes of scale * v) [plane][R][ROWB bytes]
template <int NP, bool F16 = false>
__device__ __forceinline__ void cl_publish(const f32x4 (&v)[NP], char* planes, int W, int ROWB, int PLANEB, float scale = 1.0f) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = tid + i * 256, r = p / (W >> 2), c4 = p % (W >> 2);
        if constexpr (F16) {
            f16x4c h1, h2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { _Float16 a1, a2; cl_split2(v[i][e] * scale, a1, a2); h1[e] = a1; h2[e] = a2; }
            char* base = planes + r * ROWB + c4 * 8;
            *(f16x4c*)(base) = h1;
            *(f16x4c*)(base + PLANEB) = h2;
            continue;
        }
        bf16x4 p1, p2, p3;
        split3x4(v[i], p1, p2, p3);
        char* base = planes + r * ROWB + c4 * 8;
        *(bf16x4*)(base) = p1;
        *(bf16x4*)(base + PLANEB) = p2;
        *(bf16x4*)(base + 2 * PLANEB) = p3;
    }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 8 bytes to an exchange array (see cl_store4 for `fast`)
__device__ __forceinline__ void cl_store2(float* p, const f32x2 v, bool fast) {
    if (fast) { *(f32x2*)p = v; return; }
    union { float f[2]; u64 u; } a; a.f[0] = v[0]; a.f[1] = v[1];
    __hip_atomic_store((u64*)p, a.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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
#define CL_TICK(i) do { if (prof) { const u64 n_ = clock64(); pc[i] += n_ - p_t; p_t = n_; } } while (0)

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
// "c16": full 16-row tiles, one 16-unit tile per workgroup, fp16 planes only (round 2).
//
// The kernels above keep 8 live rows per MFMA tile (half of its columns are duplicates) and 32 or 16 units per workgroup;
// at B = 256 that is one workgroup per CU with a per-step critical path of exchange + split of the WHOLE exchanged tile by
// every member + 48 MFMAs + the gate math of 2 elements per lane on two of the four waves (5500 / 7700 cycles per step at
// C4), and at Hp = 512 four to eight rounds of 256 workgroups.  Here a cluster is C = Hp / 16 workgroups around a 16-row tile:
// every MFMA column is a live row, a workgroup's W_hid slice is 64 (Hp = 256) or 128 (512) VGPRs per lane as two fp16 planes
// (nothing in LDS), every thread finishes exactly ONE (row, unit) element, and two workgroups fit a CU -- B = 256 at Hp = 512
// is one round of 512 resident workgroups.
//
//   forward   h_t travels PRE-SPLIT: the producer of an element writes its two fp16 halves into an exchange array whose
//             tile image [plane][row][Hp fp16, 16-byte chunks XOR-swizzled by row] IS the LDS image the MFMA phase reads
//             (conflict-free without padding), so a member copies 16 KB per step instead of splitting 4096 values.
//             K is split over the four waves; partial sums meet in LDS and wave w finishes tile row 4q + w of lane (j, q).
//   backward  the OUTPUT is exchanged: member m multiplies its own dhi columns (64 fp16 x 16 rows, straight from its gate
//             math through LDS) with W_hid[all Hp units][its columns] and sends every destination member the 16 x 16 block
//             of partial sums for that member's units (1 KB, f32); a member receives C blocks, adds them and resets them to
//             the sentinel.  32 KB per member and step instead of the 64 KB (Hp = 256) an input exchange of the 16-row dhi
//             tile would move, no split by the receivers, no K-split reduction.  The blocks live in a ring of four steps
//             (a slot is rewritten three publications of its reader later: its reset has long been acknowledged).
// hs / cs / saved gates / dxt / dhi are stored as always (plain stores, off the critical path).
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void cl_store1(unsigned* p, unsigned v, bool fast) {
    if (fast) *p = v;
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned cl_f16_bits(_Float16 x) { union { _Float16 h; unsigned short s; } u; u.h = x; return u.s; }
// Two fp16 planes of v, one 32-bit word per lane: even lanes hold plane 0 of units (j, j + 1), odd lanes plane 1 of (j - 1, j)
__device__ __forceinline__ unsigned cl_pair_word(float v, int j) {
    _Float16 a1, a2;
    cl_split2(v, a1, a2);
    const unsigned s1 = cl_f16_bits(a1), s2 = cl_f16_bits(a2);
    const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)((j & 1) ? s1 : s2), 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
    return (j & 1) ? (recv | (s2 << 16)) : (s1 | (recv << 16));
}

template <int CELL, int HP>
__global__ void __launch_bounds__(256, 2) rec_fwd_c16(RecArgs a) {
    constexpr int G = Gates<CELL>::G, C = HP / 16, GHP = G * HP, KBW = HP / 32 / 4;      // k-blocks per wave
    constexpr int ROWB = HP * 2, PLANEB = 16 * ROWB, TILEB = 2 * PLANEB;                 // one exchanged h tile
    constexpr int NP = TILEB / 16 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* hpl = smem_c;                                  // [2 planes][16 rows][ROWB], chunks swizzled
    char* red = smem_c + TILEB;                          // [4 waves][G][64 lanes][16 B]
    int tile, mem;
    const int ntiles = a.Bp / 16;
    if (!cl_ids(a, C, ntiles, tile, mem)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rl = 4 * q + wave;                         // the tile row this thread finishes (accumulator element `wave`)
    const int row = tile * 16 + rl, u = mem * 16 + j;
    const int T = a.T, Bp = a.Bp;
    bool dead = false;
    const bool fast = cl_same_xcc(a, C, tile, mem, (int*)red, dead);

    const int mylen = a.len[row];
    int tmax = 0;
    for (int i = 0; i < 16; ++i) tmax = max(tmax, a.len[tile * 16 + i]);

    // B operand planes: lane (unit j, k-group q) holds W_hid[(wave*KBW + kb)*32 + 8q + e][g*HP + u]
    f16x8c W1[G][KBW], W2[G][KBW];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 b1, b2;
                cl_split2(a.Whid[(size_t)((wave * KBW + kb) * 32 + 8 * q + e) * GHP + g * HP + u], b1, b2);
                W1[g][kb][e] = b1; W2[g][kb][e] = b2;
            }

    float h = a.hinit[u], c = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (CELL == CELL_LSTM) { c = a.cinit[u]; pi = a.peep[u]; pf = a.peep[HP + u]; po = a.peep[2 * HP + u]; }
    // this lane's word of the exchanged tile: plane j & 1, row rl, units (u & ~1, + 1)
    const int ue = (mem * 16 + (j & ~1));
    const unsigned xoff = (unsigned)((j & 1) * PLANEB + rl * ROWB + (((ue >> 3) ^ rl) << 4) + (ue & 7) * 2);
    // The exchange array is a ring of XRING time steps (slot t % XRING holds h_{t-1}), small enough to live in the XCD's L2
    // with every line complete: with one tile image per time step ([T + 1] slots, each written once by 4-byte stores of 16
    // workgroups) a poll took ~2600 cycles whoever arrived first -- reading a line the L2 holds partially written first
    // fetches the rest of it from memory.  A slot is a sentinel again before it is reused: a member resets ITS OWN words of
    // slot t - 1 once its poll of step t has succeeded (every member has then published step t, so has finished reading
    // t - 1), and writes them again for step t + 3 -- behind two more polls, whose waits have seen that reset acknowledged.
    constexpr int XRING = 4;
    char* const xh = (char*)a.xh + (size_t)tile * TILEB;
    const size_t xstep = (size_t)ntiles * TILEB;
    cl_store1((unsigned*)(xh + xoff), cl_pair_word(h, j), fast);
    a.hs[(size_t)row * HP + u] = h;
    if (CELL == CELL_LSTM) a.cs[(size_t)row * HP + u] = c;

    const bool fuse = a.gX != nullptr;
    float x[G], xn[G], bias[G];
#pragma unroll
    for (int g = 0; g < G; ++g) { bias[g] = fuse ? a.gbias[g * HP + u] : 0.f; x[g] = 0.f; xn[g] = 0.f; }
    auto load_id = [&](int t) -> int { return fuse ? a.gX[(size_t)row * T + (t < T ? t : T - 1)] : 0; };
    auto load_x = [&](int t, int id, float (&d)[G]) {
        const float* src = fuse ? a.gWin + (size_t)id * GHP + u : a.xt + ((size_t)(t < T ? t : T - 1) * Bp + row) * GHP + u;
#pragma unroll
        for (int g = 0; g < G; ++g) d[g] = src[g * HP];
    };
    int id_next = load_id(1), id_nn = 0;
    load_x(0, load_id(0), x);

    u64 pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_c0 = 0, p_r0 = 0, p_t = 0, p_tries = 0;
    const bool prof = a.prof != nullptr;
    if (prof) { p_c0 = clock64(); p_r0 = wall_clock64(); p_t = p_c0; }
    const f32x4 z = f32x4{0, 0, 0, 0};
    // A operand: lane (batch row j, k-group q) reads chunk (wave*KBW + kb)*4 + q of row j, swizzled by the row
    const char* hb = hpl + j * ROWB;

    // The vector-memory counter retires in order, so the wait of a poll also covers every store issued in front of it, and
    // the ~7 stores of a step took ~3000 cycles to be acknowledged (measured: "exchange wait" 2990 cycles at 0.05 re-polls per
    // step).  Only the exchange word is stored in front of the poll; hs / cs / the saved gates of step t - 1 wait in
    // registers and leave behind the poll of step t, a full step before the next wait.
    // ... and they are issued behind the barrier, between the operand reads and the MFMAs, with running offsets (hs / cs
    // and the tile-blocked gate arrays both advance Bp * Hp floats per step): nothing of it sits between poll and barrier.
    float h_pend = 0.f, c_pend = 0.f, sv_pend[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t st_step = (size_t)Bp * HP;
    size_t o_h = st_step + (size_t)row * HP + u, o_g = sbr_blocked_index(0, row, u, Bp, HP);   // of the step being stored
    auto store_step = [&]() {                            // results of the step behind o_h / o_g (h_t = slot t + 1)
        a.hs[o_h] = h_pend;
        if (CELL == CELL_LSTM) a.cs[o_h] = c_pend;
        if (CELL != CELL_VANILLA) {
#pragma unroll
            for (int k = 0; k < 4; ++k) a.g[k][o_g] = sv_pend[k];
        }
        o_h += st_step; o_g += st_step;
    };
    for (int t = 0; t < tmax; ++t) {
        load_x(t + 1, id_next, xn);                      // unconditional (clamped), as in rec_fwd_cl
        id_nn = load_id(t + 2);
        {   // h_{t-1} of all Hp units, pre-split by its producers: a straight copy into LDS
            f32x4 v[NP];
            const float* base = (const float*)(xh + (size_t)(t & (XRING - 1)) * xstep);
            p_tries += cl_fetch<NP>(v, [&](int r, int) { return base + (size_t)r * 4; }, 4, fast, dead, a.fault);
            CL_TICK(0);
#pragma unroll
            for (int i = 0; i < NP; ++i) *(f32x4*)(hpl + (threadIdx.x + i * 256) * 16) = v[i];
        }
        __syncthreads();
        CL_TICK(1);
        f32x4 acc[G], lo[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g] = z; lo[g] = z; }
        {
            f16x8c hp[KBW][2];
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
                const int ch = (((wave * KBW + kb) * 4 + q) ^ j) << 4;
                hp[kb][0] = *(const f16x8c*)(hb + ch);
                hp[kb][1] = *(const f16x8c*)(hb + ch + PLANEB);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (t > 0) {                                 // (under the operand reads' latency)
                cl_store1((unsigned*)(xh + (size_t)((t - 1) & (XRING - 1)) * xstep + xoff), CL_SENT, fast);
                store_step();
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kb = 0; kb < KBW; ++kb) {
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = cl_mfma(hp[kb][1], W1[g][kb], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) lo[g] = cl_mfma(hp[kb][0], W2[g][kb], lo[g]);
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = cl_mfma(hp[kb][0], W1[g][kb], acc[g]);
            }
            asm volatile("s_nop 15");                    // MFMA D -> VALU read hazard
        }
#pragma unroll
        for (int g = 0; g < G; ++g) *(f32x4*)(red + ((wave * G + g) * 64 + lane) * 16) = acc[g] + lo[g] * (1.0f / CL_F16_LO);
        CL_TICK(2);
        __syncthreads();                                 // partial sums of the four K parts visible; hpl free again
        CL_TICK(3);
        float xs[G], as[G], sv[4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sum = 0.f;
#pragma unroll
            for (int sw = 0; sw < 4; ++sw) sum += *(const float*)(red + ((sw * G + g) * 64 + lane) * 16 + wave * 4);
            as[g] = sum; xs[g] = x[g] + bias[g];
        }
        cell_forward<CELL, true>(xs, as, t < mylen, h, c, pi, pf, po, sv, false);
#pragma unroll
        for (int g = 0; g < G; ++g) x[g] = xn[g];        // before this step's stores are issued (vmcnt retires in order)
        id_next = id_nn;
        __builtin_amdgcn_sched_barrier(0);
        cl_store1((unsigned*)(xh + (size_t)((t + 1) & (XRING - 1)) * xstep + xoff), cl_pair_word(h, j), fast);   // the cluster waits for it
        h_pend = h; c_pend = c;
#pragma unroll
        for (int k = 0; k < 4; ++k) sv_pend[k] = sv[k];
        CL_TICK(4);
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CL_TICK(5); }      // (counters only: the exchange store's own round trip)
    }
    if (tmax > 0) store_step();
    for (int t = tmax; t < T; ++t) {                     // past the tile's longest row: the state is carried
        const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u;
        a.hs[o] = h;
        if (CELL == CELL_LSTM) a.cs[o] = c;
    }
    if (prof && lane == 0 && tile * C + mem < 32) {
        u64* o = a.prof + (((size_t)tile * C + mem) * 4 + wave) * 16;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_tries;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[3 + i] = pc[i];
    }
}

template <int CELL, int HP>
__global__ void __launch_bounds__(256, 2) rec_bwd_c16(RecArgs a) {
    constexpr int G = Gates<CELL>::G, C = HP / 16, GHP = G * HP, NT = C / 4;             // N tiles (destination members) per wave
    constexpr int KBL = G == 1 ? 1 : 2;                  // k-blocks of the member's own columns (G*16, zero-padded to 32 / 64)
    constexpr int AROW = KBL * 64, APLANE = 16 * AROW, RPB = 256 / AROW, CPR = AROW / 16;   // rows per bank period, chunks per row
    constexpr int NP = C / 4;                            // 1 KB blocks this thread's wave fetches a piece of
    constexpr int RING = 4;
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    char* apl = smem_c;                                  // [2 planes][16 rows][AROW]: this member's dhi columns, swizzled
    char* red = smem_c + 2 * APLANE;                     // [4 waves][64 lanes][16 B]
    int tile, mem;
    const int ntiles = a.Bp / 16;
    if (!cl_ids(a, C, ntiles, tile, mem)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rl = 4 * q + wave;
    const int row = tile * 16 + rl, u = mem * 16 + j;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    bool dead = false;
    const bool fast = cl_same_xcc(a, C, tile, mem, (int*)red, dead);

    const int mylen = a.len[row];
    int tmax = 0;
    for (int i = 0; i < 16; ++i) tmax = max(tmax, a.len[tile * 16 + i]);

    // B operand planes: N tile n = wave*NT + i (units 16n + j of dh), K = this member's columns kk = kb*32 + 8q + e:
    // gate kk / 16, unit mem*16 + kk % 16 -- eight consecutive floats of a W_hid row; columns past G*16 are zero
    f16x8c W1[NT][KBL], W2[NT][KBL];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int kb = 0; kb < KBL; ++kb) {
            const int g = 2 * kb + (q >> 1);
            const float* src = a.Whid + (size_t)((wave * NT + i) * 16 + j) * GHP + (g < G ? g : 0) * HP + mem * 16 + (q & 1) * 8;
            const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 b1, b2;
                cl_split2(g < G ? (e < 4 ? lo[e & 3] : hi[e & 3]) : 0.0f, b1, b2);
                W1[i][kb][e] = b1; W2[i][kb][e] = b2;
            }
        }
    if (G * 16 < KBL * 32) {                             // the padding columns of the A planes stay zero
        for (int i = threadIdx.x; i < 2 * APLANE / 4; i += 256) ((unsigned*)apl)[i] = 0u;
    }

    float dh = 0.f, dc = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (a.dh_last) dh = a.dh_last[(size_t)row * HP + u];
    if (CELL == CELL_LSTM) { pi = a.peep[u]; pf = a.peep[HP + u]; po = a.peep[2 * HP + u]; }
    float sdb[G], sdp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = 0.f;
    float sv[4] = {0.f, 0.f, 0.f, 0.f}, hprev = 0.f, cprev = 0.f, cnew = 0.f, hnew = 0.f;
    auto load_saved = [&](int t) {
        const size_t o = ((size_t)t * Bp + row) * HP + u;
        hprev = a.hs[o];
        if (CELL != CELL_VANILLA) {
            const size_t og = sbr_blocked_index(t, row, u, Bp, HP);
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = a.g[k][og];
        }
        if (CELL == CELL_LSTM) cprev = a.cs[o];
    };
    // this lane's word of a gate's 16 columns in the A planes: plane j & 1, row rl, local columns g*16 + (j & ~1), + 1
    const unsigned aoff = (unsigned)((j & 1) * APLANE + rl * AROW + (j & 6) * 2);
    const int aswz = (rl / RPB) & (CPR - 1);
    // A operand: lane (batch row j, k-group q) reads chunk kb*4 + q of row j
    const char* ab = apl + j * AROW;
    const int rswz = (j / RPB) & (CPR - 1);
    // partial-sum blocks: ring[slot][tile][destination][source][1 KB as (q, j, 4 rows)]
    const size_t slotf = (size_t)ntiles * C * C * 256;
    float* const pmine = a.pring + ((size_t)tile * C + mem) * C * 256;          // blocks addressed to this member
    float* const psend = a.pring + (size_t)tile * C * C * 256 + (size_t)mem * 256 + lane * 4;   // + destination * C * 256

    u64 pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_c0 = 0, p_r0 = 0, p_t = 0, p_tries = 0;
    const bool prof = a.prof != nullptr;
    if (prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    const f32x4 z4 = f32x4{0, 0, 0, 0};
    __syncthreads();

    for (int t = T - 1; t >= tmax; --t) {                // whole tile masked: zero rows, nobody waits for them
        if (a.dh_ext) dh += a.dh_ext[((size_t)t * Bp + row) * HP + u];
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u] = 0.f;
        if (CELL == CELL_GRU) a.dhi[((size_t)t * Bp + row) * HP + u] = 0.f;
    }
    if (tmax > 0) {
        load_saved(tmax - 1);
        const size_t o1 = ((size_t)tmax * Bp + row) * HP + u;
        if (CELL == CELL_LSTM) cnew = a.cs[o1];
        if (CELL == CELL_VANILLA) hnew = a.hs[o1];
    }
    if (prof) p_t = clock64();
    int n = 0;                                           // steps done: ring slot n % RING
    size_t o_x = ((size_t)(tmax - 1) * Bp + row) * GHP + u, o_d = ((size_t)(tmax - 1) * Bp + row) * HP + u;   // of step t (used for t >= 0 only)
    for (int t = tmax - 1; t >= 0; --t, ++n) {
        if (a.dh_ext) dh += a.dh_ext[((size_t)t * Bp + row) * HP + u];
        float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
        cell_backward<CELL, true>(t < mylen, clip, dh, dc, sv, hprev, cprev, cnew, hnew, pi, pf, po, dxi, dhi, dp, a.relu != 0);
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[g] += dxi[g];
        sdp[0] += dp[0]; sdp[1] += dp[1]; sdp[2] += dp[2];
        if (CELL == CELL_LSTM) cnew = cprev;
        if (CELL == CELL_VANILLA) hnew = hprev;
        // this member's dhi columns -> the A planes (scaled: |dhi| <= clip <= 100, see rec_bwd_x6p)
#pragma unroll
        for (int g = 0; g < G; ++g)
            *(unsigned*)(apl + aoff + (((2 * g + (j >> 3)) ^ aswz) << 4)) = cl_pair_word(dhi[g] * CL_F16_DSCALE, j);
        __builtin_amdgcn_sched_barrier(0);
        load_saved(t > 0 ? t - 1 : 0);                   // in flight across the MFMA phase and the exchange (see rec_bwd_cl)
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        CL_TICK(0);
        {
            f16x8c dpv[KBL][2];
#pragma unroll
            for (int kb = 0; kb < KBL; ++kb) {
                const int ch = ((kb * 4 + q) ^ rswz) << 4;
                dpv[kb][0] = *(const f16x8c*)(ab + ch);
                dpv[kb][1] = *(const f16x8c*)(ab + ch + APLANE);
            }
            float* dst = psend + (size_t)(n & (RING - 1)) * slotf + (size_t)(wave * NT) * C * 256;
#pragma unroll
            for (int i0 = 0; i0 < NT; i0 += 4) {         // four destination blocks at a time (accumulator registers)
                f32x4 hi[4], l1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { hi[i] = z4; l1[i] = z4; }
#pragma unroll
                for (int kb = 0; kb < KBL; ++kb) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) l1[i] = cl_mfma(dpv[kb][1], W1[i0 + i][kb], l1[i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) l1[i] = cl_mfma(dpv[kb][0], W2[i0 + i][kb], l1[i]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) hi[i] = cl_mfma(dpv[kb][0], W1[i0 + i][kb], hi[i]);
                }
                asm volatile("s_nop 15");
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    cl_store4(dst + (size_t)(i0 + i) * C * 256, (hi[i] + l1[i] * (1.0f / CL_F16_LO)) * (1.0f / CL_F16_DSCALE), fast);
            }
        }
        CL_TICK(1);
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CL_TICK(5); }      // (counters only: the block stores' round trip)
        f32x4 sum = z4;
        {   // the C blocks addressed to this member: wave w takes sources 4 i + w; then back to the sentinel
            f32x4 v[NP];
            float* base = pmine + (size_t)(n & (RING - 1)) * slotf;
            p_tries += cl_fetch<NP>(v, [&](int r, int) { return (const float*)base + (size_t)r * 4; }, 4, fast, dead, a.fault);
            CL_TICK(2);
#pragma unroll
            for (int i = 0; i < NP; ++i) sum += v[i];
        }
        *(f32x4*)(red + (wave * 64 + lane) * 16) = sum;
        __syncthreads();                                 // the four waves' sums visible; every wave is done reading the A planes
        CL_TICK(3);
        float add = 0.f;
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) add += *(const float*)(red + (sw * 64 + lane) * 16 + wave * 4);
        dh += add;
        // Behind the poll (in front of it its wait would cover them too, see rec_fwd_c16) and behind the reduction (nobody
        // waits for them): the blocks just read go back to the sentinel, dxt / dhi of this step leave
        {
            const f32x4 sent = f32x4{__uint_as_float(CL_SENT), __uint_as_float(CL_SENT), __uint_as_float(CL_SENT), __uint_as_float(CL_SENT)};
            float* base = pmine + (size_t)(n & (RING - 1)) * slotf;
#pragma unroll
            for (int i = 0; i < NP; ++i) cl_store4(base + (size_t)(threadIdx.x + i * 256) * 4, sent, fast);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[o_x + g * HP] = dxi[g];
        if (CELL == CELL_GRU) a.dhi[o_d] = dhi[2];
        o_x -= (size_t)Bp * GHP; o_d -= (size_t)Bp * HP;
        CL_TICK(4);
    }
    if (prof && lane == 0 && tile * C + mem < 32) {
        u64* o = a.prof + (((size_t)tile * C + mem) * 4 + wave) * 16;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_tries;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[3 + i] = pc[i];
    }

    // bias / peephole / initial-state gradient partial sums of this tile: over its 16 rows = over q and over the waves
    float v[G + 5];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = sdb[g];
    v[G] = sdp[0]; v[G + 1] = sdp[1]; v[G + 2] = sdp[2]; v[G + 3] = dc; v[G + 4] = dh;
    __syncthreads();
    float* redf = (float*)smem_c;                        // [4 waves][G + 5][16 units]
#pragma unroll
    for (int k = 0; k < G + 5; ++k) {
        float sum = v[k];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (q == 0) redf[(wave * (G + 5) + k) * 16 + j] = sum;
    }
    __syncthreads();
    if (wave == 0 && q == 0) {
        float* part = a.part + (size_t)tile * (GHP + 5 * HP);
#pragma unroll
        for (int k = 0; k < G + 5; ++k) {
            const float sum = redf[k * 16 + j] + redf[((G + 5) + k) * 16 + j] + redf[(2 * (G + 5) + k) * 16 + j] + redf[(3 * (G + 5) + k) * 16 + j];
            if (k < G) part[k * HP + u] = sum; else part[GHP + (k - G) * HP + u] = sum;
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
    return on && sbr_rec_cluster_ok(a) && a.xh && a.pring && a.Bp % 16 == 0 && cl_f16_fwd(a) && cl_f16_bwd(a);
}
int sbr_rec_cluster_bwd_rows(const RecArgs& a) {
    if (sbr_rec_c16_ok(a)) return 16;
    return a.Hp == 512 && !cl_f16_bwd(a) ? 4 : SBR_CL_ROWS;
}
size_t sbr_rec_c16_ring_floats(int Bp, int Hp) { return (size_t)4 * (Bp / 16) * (Hp / 16) * (Hp / 16) * 256; }

static inline int cl_grid(const RecArgs& a, int C, int R) {
    const int ntiles = a.Bp / R;
    return a.cl_linear ? ntiles * C : (ntiles + 7) / 8 * 8 * C;
}

#define CL_LAUNCH(KERNEL, C, R, LDS) do { \
        SBR_DYN_LDS(KERNEL, (LDS)); \
        KERNEL<<<cl_grid(a, C, R), 256, LDS, s>>>(a); } while (0)

template <int CELL, int HP>
static hipError_t fwd_cl(hipStream_t s, const RecArgs& a) {
    constexpr int G = Gates<CELL>::G, R = SBR_CL_ROWS;
    if (sbr_rec_c16_ok(a)) {
        hipError_t e = hipMemsetAsync(a.xh, 0xFF, (size_t)4 * a.Bp * HP * sizeof(float), s);   // the ring's slots: sentinel
        if (e != hipSuccess) return e;
        const size_t lds = (size_t)2 * 16 * HP * 2 + 4 * G * 1024;
        CL_LAUNCH((rec_fwd_c16<CELL, HP>), HP / 16, 16, lds);
        return hipGetLastError();
    }
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
    if (sbr_rec_c16_ok(a)) {
        const size_t lds = 2 * 16 * (size_t)(G == 1 ? 64 : 128) + 4 * 1024;
        CL_LAUNCH((rec_bwd_c16<CELL, HP>), HP / 16, 16, lds);
        return hipGetLastError();
    }
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
    if (sbr_rec_c16_ok(a))      // the ring of partial-sum blocks (every block a launch leaves behind is a sentinel again: this fill
        return hipMemsetAsync(a.pring, 0xFF, sbr_rec_c16_ring_floats(a.Bp, a.Hp) * sizeof(float), s);   // only guards an aborted launch)
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
