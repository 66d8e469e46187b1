// 16-row cluster recurrent kernels for wide layers (Hp = 256 / 512) on gfx950: rec_fwd_c16 / rec_bwd_c16.
// The compiled scan of sparse_lstm.py:377-425 (LSTM), :764-805 (GRU), :1120-1152 (Vanilla) and its BPTT (theano.grad of the
// scan with grad_clip, sparse_lstm.py:474-481), pinned by oracle/rnn_oracle.py.
//
// A cluster is C = Hp / 16 workgroups around one 16-row tile of the batch for all T steps: every MFMA column is a live row,
// a workgroup keeps the W_hid slice of its 16 hidden units resident as two fp16 planes (64 VGPRs per lane at Hp = 256, 128 at
// 512, nothing in LDS), every thread finishes exactly ONE (row, unit) element per step, and two workgroups fit a CU -- B = 256
// at Hp = 512 is one round of 512 resident workgroups.  Placement, the XCC handshake and `fast`: sbr_rec_cl.h.
//
// A step of either chain is ONE hand-off through the XCD's L2 (~0.5 us on this chip) plus what the workgroup does between two
// hand-offs.  The vector-memory counter of a wave retires IN ORDER, and the wait behind a poll (vmcnt(0): the poll's loads are
// the youngest operations) therefore covers everything the wave has issued before it.  Round 6 rebuilt both step loops around
// that one fact (profiles/round6_*_cluster_phases*.txt):
//
//   * NOTHING but the exchange store is issued between two polls' last instruction and the next poll.  The inputs of the
//     next step (forward: the W_in row / xt row of step t + 1; backward: the saved activations of step t - 2 and the gradient
//     from the layer above) are requested right BEHIND a poll and have a whole step to arrive -- issued in front of it (as in
//     rounds 2-5) their HBM latency WAS the "exchange wait": 2060 cycles forward, 1280 + 1300 backward.
//   * forward: h travels pre-split (two fp16 halves per value) through an exchange image that IS the MFMA operand layout --
//     [k-block][plane][k-group q][row][8 fp16]: the 16 bytes lane (row, q) feeds to the matrix instruction are contiguous
//     and a wave's piece is one contiguous KiB.  A wave polls exactly the K quarter it multiplies, straight into the operand
//     registers: no LDS staging, no barrier between hand-off and MFMAs (poll + operand fetch under one s_waitcnt).  The
//     partial sums of the four K quarters meet in LDS (double-buffered by the step's parity: one barrier per step).
//   * backward: the OUTPUT is exchanged (member m multiplies its own dhi columns with W_hid[all units][its columns] and sends
//     every member the 16 x 16 block of partial sums for that member's units), and a block validates ITSELF: the lowest
//     mantissa bit of each of its floats carries the parity of the ring lap, so the reader tells this lap's block from the
//     last lap's without anybody resetting anything -- a third of the blocks' traffic and NP stores per thread and step
//     (rounds 2-5 wrote a NaN sentinel back) are gone, and the ring is two steps deep instead of four (fits the L2s).
//   * the four gate values BPTT needs of a (t, row, unit) are one 16-byte element [t][row][unit][4] in the region of
//     RecArgs.g[0..3] (as in rec_*_x6p): one store forward, one load backward instead of four each.
#include "sbr_rec_cl.h"
#include <utility>
#include <cstdlib>

namespace {

// 16 bytes per lane through a uniform base, a 32-bit per-lane offset and an immediate: no address arithmetic per poll.
// Invisible to the compiler's vmcnt bookkeeping: the poll waits by hand.
template <int IMM>
__device__ __forceinline__ void c16_ld16(f32x4& v, const void* sbase, unsigned voff) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3 sc1" : "=v"(v) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
template <int NP, int... I>
__device__ __forceinline__ void c16_ld_all(f32x4 (&v)[NP], const void* sbase, unsigned voff, std::integer_sequence<int, I...>) {
    (c16_ld16<(I & 3) * 1024>(v[I], sbase, voff + (unsigned)(I >> 2) * 4096u), ...);
}

// Polls NP consecutive KiB pieces (lane l: bytes [16 l, 16 l + 16) of each) until `valid` accepts all of them.
// FAST (whole cluster on one XCC): 16-byte sc1 loads (bypass the CU's L1, served by the shared L2); otherwise 8-byte
// agent-scope atomic loads.  FAST is a template parameter of the whole kernel body: with both forms in one loop, loads the
// compiler can see (the atomics) pending on the shared back edge make it guard the asm's destination registers with waits
// BETWEEN the four loads -- four serial round trips instead of one.  Bounded: a poll that never succeeds raises the fault
// flag instead of hanging the GPU.
template <int NP, bool FAST, typename VALID>
__device__ __forceinline__ int c16_poll(f32x4 (&v)[NP], const char* sbase, unsigned voff, bool& dead, int* fault, VALID valid) {
    int tries = 0;
    while (true) {
        if constexpr (FAST) {
            c16_ld_all<NP>(v, sbase, voff, std::make_integer_sequence<int, NP>{});
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(v[i]));
        } else {
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const float* q = (const float*)(sbase + voff + i * 1024);
                union { u64 u[2]; f32x4 f; } x;
                x.u[0] = cl_load(q); x.u[1] = cl_load(q + 2);
                v[i] = x.f;
            }
        }
        if (valid(v) || dead) break;
        if (++tries > CL_SPIN_LIMIT) { dead = true; atomicOr(fault, 1); break; }
        __builtin_amdgcn_s_sleep(1);
    }
    return tries;
}
// A copy the compiler can neither move nor fold: hipcc places the copies of loop-carried prefetch registers where it likes
// (at the loop header, behind the NEXT request -- with a vmcnt(0) in front).  Behind a poll's wait the source is complete.
__device__ __forceinline__ float c16_mov(float src) {
    float d;
    asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(src));
    return d;
}

// A thread's gate gradients of its (row, unit) as fp16 pairs, K-contiguous: the K index of a member's columns is
// unit * GP + gate slot (GP = 4: i, f, c, o / r, u, c, 0; GP = 2: the one gate, 0), so the GP values one thread holds are GP
// consecutive K positions -- one 8-byte (4-byte) store per plane, no lane exchange (rounds 2-5 ordered K gate-major and paired
// neighbouring units through a DPP move: 11 VALU instructions per gate).  Padding slots carry zeros.
template <int G, int GP>
__device__ __forceinline__ void c16_pack_gates(const float (&dhi)[G], float scale, f16x4c& p1, f16x4c& p2) {
#pragma unroll
    for (int gs = 0; gs < 4; ++gs) {
        _Float16 a1 = (_Float16)0.0f, a2 = (_Float16)0.0f;
        if (gs < G && gs < GP) cl_split2(dhi[gs < G ? gs : 0] * scale, a1, a2);
        p1[gs] = a1; p2[gs] = a2;
    }
}

// no word is the NaN sentinel (the largest 32-bit pattern: one v_max3_u32 per two words)
template <int NP>
__device__ __forceinline__ bool c16_no_sentinel(const f32x4 (&v)[NP]) {
    unsigned mx = 0u;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = max(mx, __float_as_uint(v[i][e]));
    return mx != CL_SENT;
}
// every word carries this lap's parity in its lowest bit
template <int NP>
__device__ __forceinline__ bool c16_all_tagged(const f32x4 (&v)[NP], unsigned par) {
    if (par) {
        unsigned an = ~0u;
#pragma unroll
        for (int i = 0; i < NP; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) an &= __float_as_uint(v[i][e]);
        return (an & 1u) != 0u;
    }
    unsigned o = 0u;
#pragma unroll
    for (int i = 0; i < NP; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) o |= __float_as_uint(v[i][e]);
    return (o & 1u) == 0u;
}

}  // namespace

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int CELL, int HP, bool FAST>
__device__ __forceinline__ void c16_fwd_body(const RecArgs& a, const int tile, const int mem, bool dead, char* smem_c) {
    constexpr bool fast = FAST;
    constexpr int G = Gates<CELL>::G, C = HP / 16, GHP = G * HP, KBW = HP / 32 / 4;      // k-blocks per wave
    constexpr int TILEB = 64 * HP;                       // one exchanged h tile: [HP / 32 k-blocks][2 planes][4 q][16 rows][16 B]
    constexpr int NPW = 2 * KBW;                         // KiB pieces a wave polls: its k-blocks x 2 planes
    constexpr int REDB = 4 * G * 1024;                   // one set of partial sums [4 waves][G][64 lanes][16 B]
    char* red = smem_c;                                  // [2 (step parity)][REDB]
    const int ntiles = a.Bp / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rl = 4 * q + wave;                         // the tile row this thread finishes (accumulator element `wave`)
    const int row = tile * 16 + rl, u = mem * 16 + j;
    const int T = a.T, Bp = a.Bp;

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
    // this lane's word of the exchanged tile: plane j & 1, row rl, units (ue, ue + 1) = K positions of the next step
    const int ue = mem * 16 + (j & ~1);
    const unsigned xoff = (unsigned)((((((ue >> 5) * 2 + (j & 1)) * 4 + ((ue >> 3) & 3)) * 16 + rl) * 16) + (ue & 7) * 2);
    // ... and the pieces this lane polls: k-blocks wave*KBW .. + KBW - 1, both planes, as lane (row j, k-group q) = lane
    const unsigned poff = (unsigned)(wave * NPW * 1024 + lane * 16);
    // The exchange array is a ring of XRING time steps (slot t % XRING holds h_{t-1}), small enough to live in the XCD's L2.
    // A slot is a sentinel again before it is reused, without a single owner: a member resets ITS OWN words of slot t - 1
    // behind the reduce barrier of step t (all four waves' polls have then succeeded: every member has published step t, so
    // has finished reading t - 1), and writes them again for step t + 3 -- behind two more polls, whose waits have seen
    // that reset acknowledged.
    constexpr int XRING = 4;
    char* const xh = (char*)a.xh + (size_t)tile * TILEB;
    const size_t xstep = (size_t)ntiles * TILEB;
    cl_store1((unsigned*)(xh + xoff), cl_pair_word(h, j), fast);

    const bool fuse = a.gX != nullptr;
    float bias[G];
#pragma unroll
    for (int g = 0; g < G; ++g) bias[g] = fuse ? a.gbias[g * HP + u] : 0.f;
    auto load_id = [&](int t) -> int { return fuse ? a.gX[(size_t)row * T + (t < T ? t : T - 1)] : 0; };
    auto load_x = [&](int t, int id, float (&d)[G]) {
        const float* src = fuse ? a.gWin + (size_t)id * GHP + u : a.xt + ((size_t)(t < T ? t : T - 1) * Bp + row) * GHP + u;
#pragma unroll
        for (int g = 0; g < G; ++g) d[g] = src[g * HP];
    };
    // The input row of step t + 1 is requested behind the poll of step t into xn and moves to x behind the poll of step t + 1
    // (whose wait has it in; c16_mov).  The id of step t + 2 travels one request ahead of its row.
    float x[G], xn[G];
#pragma unroll
    for (int g = 0; g < G; ++g) x[g] = 0.f;
    int idn = load_id(1);
    load_x(0, load_id(0), xn);

    u64 pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_c0 = 0, p_r0 = 0, p_t = 0, p_tries = 0;
    const bool prof = a.prof != nullptr;
    if (prof) { p_c0 = clock64(); p_r0 = wall_clock64(); p_t = p_c0; }
    const f32x4 z = f32x4{0, 0, 0, 0};

    // results of step t - 1 wait in registers and leave behind the poll of step t (header): hs / cs and the 16-byte gate element
    // (no branch around them: the initial state is "step -1" -- slot 0 of hs / cs; its gate element lands on step 0's, which the
    // next iteration overwrites)
    float h_pend = h, c_pend = c, sv_pend[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t st_step = (size_t)Bp * HP;
    size_t o_h = (size_t)row * HP + u, o_g = ((size_t)row * HP + u) * 4;   // of the step being stored
    auto store_step = [&](bool advance_g) {              // h_t = slot t + 1
        a.hs[o_h] = h_pend;
        if (CELL == CELL_LSTM) a.cs[o_h] = c_pend;
        if (CELL != CELL_VANILLA) *(f32x4*)(a.g[0] + o_g) = f32x4{sv_pend[0], sv_pend[1], sv_pend[2], sv_pend[3]};
        o_h += st_step; o_g += advance_g ? 4 * st_step : 0;
    };
    // (a wait the compiler knows about, in front of the loop: its pass places ONE wait at a loop header for both incoming
    // edges, and with the prologue's requests pending on the entry edge that wait is vmcnt(0) in every iteration)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    for (int t = 0; t < tmax; ++t) {
        // h_{t-1}, the K quarter this wave multiplies, pre-split by its producers: straight into the A operand registers
        f32x4 v[NPW];
        p_tries += c16_poll<NPW, FAST>(v, xh + (size_t)(t & (XRING - 1)) * xstep, poff, dead, a.fault,
                                 [](const f32x4 (&w)[NPW]) { return c16_no_sentinel<NPW>(w); });
        CL_TICK(0);
        // behind the poll: a whole step until the next wait
#pragma unroll
        for (int g = 0; g < G; ++g) x[g] = c16_mov(xn[g]);
        load_x(t + 1, idn, xn);                          // unconditional (clamped)
        idn = load_id(t + 2);
        store_step(t > 0);
        f32x4 acc[G], lo[G];
#pragma unroll
        for (int g = 0; g < G; ++g) { acc[g] = z; lo[g] = z; }
#pragma unroll
        for (int kb = 0; kb < KBW; ++kb) {
            const f16x8c h0 = (f16x8c)v[2 * kb], h1 = (f16x8c)v[2 * kb + 1];
#pragma unroll
            for (int g = 0; g < G; ++g) lo[g] = cl_mfma(h1, W1[g][kb], lo[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) lo[g] = cl_mfma(h0, W2[g][kb], lo[g]);
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = cl_mfma(h0, W1[g][kb], acc[g]);
        }
        asm volatile("s_nop 15");                        // MFMA D -> VALU read hazard
        char* const rb = red + (t & 1) * REDB;
#pragma unroll
        for (int g = 0; g < G; ++g) *(f32x4*)(rb + ((wave * G + g) * 64 + lane) * 16) = acc[g] + lo[g] * (1.0f / CL_F16_LO);
        CL_TICK(1);
        __syncthreads();                                 // partial sums of the four K parts visible; every member has published step t
        CL_TICK(2);
        float xs[G], as[G], sv[4];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float sum = 0.f;
#pragma unroll
            for (int sw = 0; sw < 4; ++sw) sum += *(const float*)(rb + ((sw * G + g) * 64 + lane) * 16 + wave * 4);
            as[g] = sum; xs[g] = x[g] + bias[g];
        }
        cell_forward<CELL, true>(xs, as, t < mylen, h, c, pi, pf, po, sv, false);
        __builtin_amdgcn_sched_barrier(0);
        cl_store1((unsigned*)(xh + (size_t)((t + 1) & (XRING - 1)) * xstep + xoff), cl_pair_word(h, j), fast);   // the cluster waits for it
        cl_store1((unsigned*)(xh + (size_t)((t - 1) & (XRING - 1)) * xstep + xoff), CL_SENT, fast);   // (t = 0: slot 3, a sentinel already)
        h_pend = h; c_pend = c;
#pragma unroll
        for (int k = 0; k < 4; ++k) sv_pend[k] = sv[k];
        CL_TICK(3);
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CL_TICK(5); }      // (counters only: everything issued since the poll)
    }
    store_step(tmax > 0);                                // (tmax = 0: the initial state into slot 0)
    for (int t = tmax; t < T; ++t) {                     // past the tile's longest row: the state is carried
        const size_t o = ((size_t)(t + 1) * Bp + row) * HP + u;
        a.hs[o] = h;
        if (CELL == CELL_LSTM) a.cs[o] = c;
    }
    if (prof && lane == 0 && tile * C + mem < min(32, a.Bp / 8)) {      // (the counters' region: Bp / 16 * 128 words per direction)
        u64* o = a.prof + (((size_t)tile * C + mem) * 4 + wave) * 16;
        o[0] = clock64() - p_c0; o[1] = wall_clock64() - p_r0; o[2] = p_tries;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[3 + i] = pc[i];
    }
}

// ---------------------------------------------------------------------------------------
// backward (BPTT)
// ---------------------------------------------------------------------------------------
template <int CELL, int HP>
__global__ void __launch_bounds__(256, 2) rec_fwd_c16(RecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    int tile, mem;
    if (!cl_ids(a, HP / 16, a.Bp / 16, tile, mem)) return;
    bool dead = false;
    if (cl_same_xcc(a, HP / 16, tile, mem, (int*)smem_c, dead)) c16_fwd_body<CELL, HP, true>(a, tile, mem, dead, smem_c);
    else c16_fwd_body<CELL, HP, false>(a, tile, mem, dead, smem_c);
}

template <int CELL, int HP, bool FAST>
__device__ __forceinline__ void c16_bwd_body(const RecArgs& a, const int tile, const int mem, bool dead, char* smem_c) {
    constexpr bool fast = FAST;
    constexpr int G = Gates<CELL>::G, C = HP / 16, GHP = G * HP, NT = C / 4;             // N tiles (destination members) per wave
    constexpr int KBL = G == 1 ? 1 : 2;                  // k-blocks of the member's own columns (G*16, zero-padded to 32 / 64)
    constexpr int AROW = KBL * 64, APLANE = 16 * AROW, RPB = 256 / AROW, CPR = AROW / 16;   // rows per bank period, chunks per row
    constexpr int NP = C / 4;                            // 1 KB blocks this thread's wave fetches a piece of
    constexpr int RING = SBR_C16_RING;
    char* apl = smem_c;                                  // [2 planes][16 rows][AROW]: this member's dhi columns, swizzled
    char* red = smem_c + 2 * APLANE;                     // [4 waves][64 lanes][16 B]
    const int ntiles = a.Bp / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rl = 4 * q + wave;
    const int row = tile * 16 + rl, u = mem * 16 + j;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;

    const int mylen = a.len[row];
    int tmax = 0;
    for (int i = 0; i < 16; ++i) tmax = max(tmax, a.len[tile * 16 + i]);

    // B operand planes: N tile n = wave*NT + i (units 16n + j of dh), K = this member's columns kk = kb*32 + 8q + e =
    // unit * GP + gate slot (c16_pack_gates); slots past G are zero
    constexpr int GP = 2 * KBL;
    f16x8c W1[NT][KBL], W2[NT][KBL];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int kb = 0; kb < KBL; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = kb * 32 + 8 * q + e, gs = kk % GP, ul = kk / GP;
                _Float16 b1, b2;
                cl_split2(gs < G ? a.Whid[(size_t)((wave * NT + i) * 16 + j) * GHP + (gs < G ? gs : 0) * HP + mem * 16 + ul] : 0.0f, b1, b2);
                W1[i][kb][e] = b1; W2[i][kb][e] = b2;
            }

    float dh = 0.f, dc = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (a.dh_last) dh = a.dh_last[(size_t)row * HP + u];
    if (CELL == CELL_LSTM) { pi = a.peep[u]; pf = a.peep[HP + u]; po = a.peep[2 * HP + u]; }
    float sdb[G], sdp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = 0.f;
    // Saved activations of a step: the 16-byte gate element, h_{t-1} (slot t of hs), c_{t-1}; plus the gradient from the layer
    // above.  The gate math of a step comes FIRST (the hand-off follows it), so what step t - 2 needs is requested behind the
    // poll of step t, into `nxt`: two hand-offs to arrive.  Behind a poll `nxt` (requested behind the LAST poll, complete now)
    // moves to `cur` (c16_mov) and is requested again.
    struct Saved { f32x4 sv; float hprev, cprev, dhe; };
    const float* const dhe_src = a.dh_ext;               // (never null here: the launcher points it at hs and clears dhe_on)
    const bool dhe_on = a.dhe_on != 0;
    auto load_saved = [&](int t, Saved& s) {
        const size_t o = ((size_t)t * Bp + row) * HP + u;
        s.hprev = a.hs[o];
        if (CELL != CELL_VANILLA) s.sv = *(const f32x4*)(a.g[0] + o * 4);
        if (CELL == CELL_LSTM) s.cprev = a.cs[o];
        s.dhe = dhe_src[o];                              // (no branch around a request: hipcc's wait pass would fall back to vmcnt(0))
    };
    Saved cur, nxt;
    cur.sv = nxt.sv = f32x4{0, 0, 0, 0};
    cur.hprev = cur.cprev = cur.dhe = nxt.hprev = nxt.cprev = nxt.dhe = 0.f;
    float cnew = 0.f, hnew = 0.f;
    // this thread's GP columns in the A planes: row rl, bytes [2 GP j, 2 GP (j + 1)) of the row, 16-byte chunks swizzled by the row
    const int aswz = (rl / RPB) & (CPR - 1);
    const unsigned aoff = (unsigned)(rl * AROW + ((((j * GP * 2) >> 4) ^ aswz) << 4) + ((j * GP * 2) & 15));
    // A operand: lane (batch row j, k-group q) reads chunk kb*4 + q of row j
    const char* ab = apl + j * AROW;
    const int rswz = (j / RPB) & (CPR - 1);
    // partial-sum blocks: ring[slot][tile][destination][source][1 KB as (q, j, 4 rows)]; wave w receives sources w*NP .. + NP - 1
    const size_t slotb = (size_t)ntiles * C * C * 1024;
    const char* const pmine = (const char*)a.pring + ((size_t)tile * C + mem) * C * 1024;    // blocks addressed to this member
    const unsigned poff = (unsigned)(wave * NP * 1024 + lane * 16);
    float* const psend = a.pring + (size_t)tile * C * C * 256 + (size_t)mem * 256 + lane * 4;   // + destination * C * 256

    u64 pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_c0 = 0, p_r0 = 0, p_t = 0, p_tries = 0;
    const bool prof = a.prof != nullptr;
    if (prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    const f32x4 z4 = f32x4{0, 0, 0, 0};
    __syncthreads();

    float dhx = 0.f;
    for (int t = T - 1; t >= tmax; --t) {                // whole tile masked: zero rows, nobody waits for them
        if (dhe_on) dhx += a.dh_ext[((size_t)t * Bp + row) * HP + u];
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u] = 0.f;
        if (CELL == CELL_GRU) a.dhi[((size_t)t * Bp + row) * HP + u] = 0.f;
    }
    dh += dhx;
    if (tmax > 0) {
        load_saved(tmax - 1, cur);
        load_saved(tmax > 1 ? tmax - 2 : 0, nxt);
        const size_t o1 = ((size_t)tmax * Bp + row) * HP + u;
        if (CELL == CELL_LSTM) cnew = a.cs[o1];
        if (CELL == CELL_VANILLA) hnew = a.hs[o1];
    }
    if (prof) p_t = clock64();
    int n = 0;                                           // steps done: ring slot n % RING, lap n / RING
    size_t o_x = ((size_t)(tmax - 1) * Bp + row) * GHP + u, o_d = ((size_t)(tmax - 1) * Bp + row) * HP + u;   // of step t (used for t >= 0 only)
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // (see rec_fwd_c16)
    for (int t = tmax - 1; t >= 0; --t, ++n) {
        if (dhe_on) dh += cur.dhe;
        float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
        {
            const float sv[4] = {cur.sv[0], cur.sv[1], cur.sv[2], cur.sv[3]};
            cell_backward<CELL, true>(t < mylen, clip, dh, dc, sv, cur.hprev, cur.cprev, cnew, hnew, pi, pf, po, dxi, dhi, dp, a.relu != 0);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[g] += dxi[g];
        sdp[0] += dp[0]; sdp[1] += dp[1]; sdp[2] += dp[2];
        if (CELL == CELL_LSTM) cnew = cur.cprev;
        if (CELL == CELL_VANILLA) hnew = cur.hprev;
        // this member's dhi columns -> the A planes (scaled: |dhi| <= clip <= 100, see rec_bwd_x6p)
        {
            f16x4c p1, p2;
            c16_pack_gates<G, GP>(dhi, CL_F16_DSCALE, p1, p2);
            if constexpr (GP == 4) { *(f16x4c*)(apl + aoff) = p1; *(f16x4c*)(apl + aoff + APLANE) = p2; }
            else { *(f16x2c*)(apl + aoff) = f16x2c{p1[0], p1[1]}; *(f16x2c*)(apl + aoff + APLANE) = f16x2c{p2[0], p2[1]}; }
        }
        __syncthreads();
        CL_TICK(0);
        const unsigned par = (unsigned)(n / RING) & 1u;
        {
            f16x8c dpv[KBL][2];
#pragma unroll
            for (int kb = 0; kb < KBL; ++kb) {
                const int ch = ((kb * 4 + q) ^ rswz) << 4;
                dpv[kb][0] = *(const f16x8c*)(ab + ch);
                dpv[kb][1] = *(const f16x8c*)(ab + ch + APLANE);
            }
            float* dst = psend + (size_t)(n % RING) * (slotb / 4) + (size_t)(wave * NT) * C * 256;
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
                for (int i = 0; i < 4; ++i) {
                    f32x4 bv = (hi[i] + l1[i] * (1.0f / CL_F16_LO)) * (1.0f / CL_F16_DSCALE);
#pragma unroll
                    for (int e = 0; e < 4; ++e) bv[e] = __uint_as_float((__float_as_uint(bv[e]) & ~1u) | par);   // this lap's parity
                    cl_store4(dst + (size_t)(i0 + i) * C * 256, bv, fast);
                }
            }
        }
        CL_TICK(1);
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CL_TICK(5); }      // (counters only: the block stores' round trip)
        f32x4 sum = z4;
        {   // the blocks addressed to this member (this wave: NP consecutive sources)
            f32x4 v[NP];
            p_tries += c16_poll<NP, FAST>(v, pmine + (size_t)(n % RING) * slotb, poff, dead, a.fault,
                                    [par](const f32x4 (&w)[NP]) { return c16_all_tagged<NP>(w, par); });
            CL_TICK(2);
#pragma unroll
            for (int i = 0; i < NP; ++i) sum += v[i];
        }
        *(f32x4*)(red + (wave * 64 + lane) * 16) = sum;
        CL_TICK(3);
        __syncthreads();                                 // the four waves' sums visible; every wave is done reading the A planes
        CL_TICK(4);
        float pr[4];
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) pr[sw] = *(const float*)(red + (sw * 64 + lane) * 16 + wave * 4);
        __builtin_amdgcn_sched_barrier(0);
        // behind the poll, off the path from the blocks to the next gate math (under the reads' latency): the step after next's
        // saved activations, this step's dxt / dhi
#pragma unroll
        for (int e = 0; e < 4; ++e) cur.sv[e] = c16_mov(nxt.sv[e]);
        cur.hprev = c16_mov(nxt.hprev); cur.cprev = c16_mov(nxt.cprev); cur.dhe = c16_mov(nxt.dhe);
        load_saved(t > 2 ? t - 2 : 0, nxt);
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[o_x + g * HP] = dxi[g];
        if (CELL == CELL_GRU) a.dhi[o_d] = dhi[2];
        o_x -= (size_t)Bp * GHP; o_d -= (size_t)Bp * HP;
        __builtin_amdgcn_sched_barrier(0);
        dh += (pr[0] + pr[1]) + (pr[2] + pr[3]);
        CL_TICK(6);
    }
    if (prof && lane == 0 && tile * C + mem < min(32, a.Bp / 8)) {      // (the counters' region: Bp / 16 * 128 words per direction)
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
// backward, Hp = 512: the exchange in TWO levels.
// With the one-level output exchange a member of a 32-member cluster sends and receives 32 KB of partial-sum blocks per step
// (512 workgroups: 32 MB of stores per step on a memory system that writes every store through): 6 us per step, 1.2 ms per
// layer where the forward takes 0.57.  Here the cluster is 8 GROUPS of four members:
//   level 1  the four members of a group share their dhi columns (4 x 64 columns x 16 rows, pre-split fp16 pairs, in the MFMA
//            operand layout exactly as the forward exchanges h: 4 KB out, 16 KB in per member, staged through LDS);
//   level 2  member s of a group multiplies the group's K = 256 columns for ITS QUARTER of the destinations (units 128 s ..
//            128 s + 127: 8 blocks of 16 x 16 partial sums, every block complete over the group's K) and sends them; a member
//            receives one block per group: 8 KB out, 8 KB in, self-validating as above.
// 12 KB of stores per member and step instead of 32 (+ no resets), the same 192 MFMAs per workgroup, the same 128 weight
// registers -- at the price of a second hand-off per step, which two resident workgroups per CU cover for each other.
// ---------------------------------------------------------------------------------------
template <int CELL, bool FAST>
__device__ __forceinline__ void c16_bwd2_body(const RecArgs& a, const int tile, const int mem, bool dead, char* smem_c) {
    constexpr bool fast = FAST;
    constexpr int HP = 512, G = Gates<CELL>::G, C = HP / 16, GHP = G * HP;
    constexpr int S = 4, NG = C / S;                     // members per group, groups per cluster
    constexpr int KBL = G == 1 ? 1 : 2;                  // k-blocks of one member's columns (G*16, zero-padded to 32 / 64)
    constexpr int GP = 2 * KBL;                          // gate slots of a member in the image (the padding slots carry zeros)
    constexpr int KBG = S * KBL;                         // k-blocks of a group's columns
    constexpr int IMG1 = KBG * 2048;                     // level-1 image of a group: [KBG][2 planes][4 q][16 rows][16 B]
    constexpr int NP1 = KBG / 4 * 2;                     // KiB pieces of it a wave polls and stages
    constexpr int NT2 = 2;                               // destination blocks per wave (8 per member)
    constexpr int NP2 = NG / 4;                          // incoming blocks per wave
    constexpr int XRING = 4, RING = SBR_C16_RING;
    char* stg = smem_c;                                  // the group's image, as exchanged
    char* red = smem_c + IMG1;                           // [4 waves][64 lanes][16 B]
    const int ntiles = a.Bp / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int rl = 4 * q + wave;
    const int row = tile * 16 + rl, u = mem * 16 + j;
    const int gi = mem / S, sm = mem % S;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;

    const int mylen = a.len[row];
    int tmax = 0;
    for (int i = 0; i < 16; ++i) tmax = max(tmax, a.len[tile * 16 + i]);

    // B operand planes: N tile i of this wave = destination member d = 8 sm + 2 wave + i (units 16 d + j of dh), K = the group's
    // columns kbg*32 + 8q + e: member kbg / KBL of the group, its column kk = (kbg % KBL)*32 + 8q + e = unit * GP + gate slot
    f16x8c W1[NT2][KBG], W2[NT2][KBG];
#pragma unroll
    for (int i = 0; i < NT2; ++i)
#pragma unroll
        for (int kbg = 0; kbg < KBG; ++kbg)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int kk = (kbg % KBL) * 32 + 8 * q + e, gs = kk % GP, ul = kk / GP;
                const int d = 8 * sm + 2 * wave + i;
                _Float16 b1, b2;
                cl_split2(gs < G ? a.Whid[(size_t)(d * 16 + j) * GHP + (gs < G ? gs : 0) * HP + (gi * S + kbg / KBL) * 16 + ul] : 0.0f, b1, b2);
                W1[i][kbg][e] = b1; W2[i][kbg][e] = b2;
            }

    float dh = 0.f, dc = 0.f, pi = 0.f, pf = 0.f, po = 0.f;
    if (a.dh_last) dh = a.dh_last[(size_t)row * HP + u];
    if (CELL == CELL_LSTM) { pi = a.peep[u]; pf = a.peep[HP + u]; po = a.peep[2 * HP + u]; }
    float sdb[G], sdp[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = 0.f;
    struct Saved { f32x4 sv; float hprev, cprev, dhe; };
    const float* const dhe_src = a.dh_ext;               // (never null here: see rec_bwd_c16)
    const bool dhe_on = a.dhe_on != 0;
    auto load_saved = [&](int t, Saved& s) {
        const size_t o = ((size_t)t * Bp + row) * HP + u;
        s.hprev = a.hs[o];
        if (CELL != CELL_VANILLA) s.sv = *(const f32x4*)(a.g[0] + o * 4);
        if (CELL == CELL_LSTM) s.cprev = a.cs[o];
        s.dhe = dhe_src[o];
    };
    Saved cur, nxt;
    cur.sv = nxt.sv = f32x4{0, 0, 0, 0};
    cur.hprev = cur.cprev = cur.dhe = nxt.hprev = nxt.cprev = nxt.dhe = 0.f;
    float cnew = 0.f, hnew = 0.f;

    // level 1: ring[slot][tile][group][IMG1] behind the blocks' ring; this lane's word of gate slot gs: plane j & 1, row rl,
    // columns (sm*KBL*32 + gs*16 + (j & ~1), + 1) of the group
    constexpr size_t RING2_BYTES_PER_TILE = (size_t)C * NG * 1024;
    char* const ring2 = (char*)a.pring;
    char* const ring1 = ring2 + (size_t)RING * ntiles * RING2_BYTES_PER_TILE;
    const size_t slot1b = (size_t)ntiles * NG * IMG1;
    char* const x1 = ring1 + ((size_t)tile * NG + gi) * IMG1;
    // this thread's GP columns (member-local kk = GP j ..): k-block kk >> 5, k-group (kk >> 3) & 3, bytes 2 (kk & 7) of the lane's 16
    const unsigned xoff1 = (unsigned)(((sm * KBL + ((j * GP) >> 5)) * 2 * 4 + (((j * GP) >> 3) & 3)) * 256 + rl * 16 + ((j * GP) & 7) * 2);   // + 1024: plane 1
    const unsigned poff1 = (unsigned)(wave * NP1 * 1024 + lane * 16);
    // level 2: ring[slot][tile][destination][source group][1 KB as (q, j, 4 rows)]; wave w receives groups w*NP2 .. + NP2 - 1
    const size_t slot2b = (size_t)ntiles * RING2_BYTES_PER_TILE;
    const char* const pmine = ring2 + ((size_t)tile * C + mem) * NG * 1024;
    const unsigned poff2 = (unsigned)(wave * NP2 * 1024 + lane * 16);
    char* const psend = ring2 + (((size_t)tile * C + 8 * sm + 2 * wave) * NG + gi) * 1024 + lane * 16;   // + i * NG * 1024

    u64 pc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, p_c0 = 0, p_r0 = 0, p_t = 0, p_tries = 0;
    const bool prof = a.prof != nullptr;
    if (prof) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    const f32x4 z4 = f32x4{0, 0, 0, 0};
    __syncthreads();

    float dhx = 0.f;
    for (int t = T - 1; t >= tmax; --t) {                // whole tile masked: zero rows, nobody waits for them
        if (dhe_on) dhx += a.dh_ext[((size_t)t * Bp + row) * HP + u];
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[((size_t)t * Bp + row) * GHP + g * HP + u] = 0.f;
        if (CELL == CELL_GRU) a.dhi[((size_t)t * Bp + row) * HP + u] = 0.f;
    }
    dh += dhx;
    if (tmax > 0) {
        load_saved(tmax - 1, cur);
        load_saved(tmax > 1 ? tmax - 2 : 0, nxt);
        const size_t o1 = ((size_t)tmax * Bp + row) * HP + u;
        if (CELL == CELL_LSTM) cnew = a.cs[o1];
        if (CELL == CELL_VANILLA) hnew = a.hs[o1];
    }
    if (prof) p_t = clock64();
    int n = 0;                                           // steps done
    size_t o_x = ((size_t)(tmax - 1) * Bp + row) * GHP + u, o_d = ((size_t)(tmax - 1) * Bp + row) * HP + u;   // of step t (used for t >= 0 only)
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // (see rec_fwd_c16)
    for (int t = tmax - 1; t >= 0; --t, ++n) {
        if (dhe_on) dh += cur.dhe;
        float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
        {
            const float sv[4] = {cur.sv[0], cur.sv[1], cur.sv[2], cur.sv[3]};
            cell_backward<CELL, true>(t < mylen, clip, dh, dc, sv, cur.hprev, cur.cprev, cnew, hnew, pi, pf, po, dxi, dhi, dp, a.relu != 0);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[g] += dxi[g];
        sdp[0] += dp[0]; sdp[1] += dp[1]; sdp[2] += dp[2];
        if (CELL == CELL_LSTM) cnew = cur.cprev;
        if (CELL == CELL_VANILLA) hnew = cur.hprev;
        // level 1: this member's dhi columns to its group (scaled: |dhi| <= clip <= 100, see rec_bwd_x6p); padding slots: zeros
        {
            char* const xb = x1 + (size_t)(n & (XRING - 1)) * slot1b + xoff1;
            f16x4c p1, p2;
            c16_pack_gates<G, GP>(dhi, CL_F16_DSCALE, p1, p2);
            if constexpr (GP == 4) {
                union { f16x4c h; f32x2 f; } w1, w2; w1.h = p1; w2.h = p2;
                cl_store2((float*)xb, w1.f, fast); cl_store2((float*)(xb + 1024), w2.f, fast);
            } else {
                union { f16x2c h; unsigned u; } w1, w2; w1.h = f16x2c{p1[0], p1[1]}; w2.h = f16x2c{p2[0], p2[1]};
                cl_store1((unsigned*)xb, w1.u, fast); cl_store1((unsigned*)(xb + 1024), w2.u, fast);
            }
        }
        CL_TICK(0);
        {   // the group's columns, the K quarter this wave stages
            f32x4 v[NP1];
            p_tries += c16_poll<NP1, FAST>(v, x1 + (size_t)(n & (XRING - 1)) * slot1b, poff1, dead, a.fault,
                                           [](const f32x4 (&w)[NP1]) { return c16_no_sentinel<NP1>(w); });
            CL_TICK(1);
#pragma unroll
            for (int i = 0; i < NP1; ++i) *(f32x4*)(stg + poff1 + i * 1024) = v[i];
        }
        __syncthreads();                                 // the image complete in LDS; every member of the group has published step n
        CL_TICK(2);
        // behind the poll and the barrier: this step's dxt / dhi, the reset of this member's words of the last step's image
#pragma unroll
        for (int g = 0; g < G; ++g) a.dxt[o_x + g * HP] = dxi[g];
        if (CELL == CELL_GRU) a.dhi[o_d] = dhi[2];
        o_x -= (size_t)Bp * GHP; o_d -= (size_t)Bp * HP;
        {
            char* const xb = x1 + (size_t)((n - 1) & (XRING - 1)) * slot1b + xoff1;      // (n = 0: slot 3, a sentinel already)
            if constexpr (GP == 4) {
                const f32x2 sent = f32x2{__uint_as_float(CL_SENT), __uint_as_float(CL_SENT)};
                cl_store2((float*)xb, sent, fast); cl_store2((float*)(xb + 1024), sent, fast);
            } else { cl_store1((unsigned*)xb, CL_SENT, fast); cl_store1((unsigned*)(xb + 1024), CL_SENT, fast); }
        }
        const unsigned par = (unsigned)(n / RING) & 1u;
        {
            f32x4 hi[NT2], l1[NT2];
#pragma unroll
            for (int i = 0; i < NT2; ++i) { hi[i] = z4; l1[i] = z4; }
            const char* sb = stg + lane * 16;
            f16x8c d0 = *(const f16x8c*)(sb), d1 = *(const f16x8c*)(sb + 1024);
#pragma unroll
            for (int kbg = 0; kbg < KBG; ++kbg) {
                const f16x8c c0 = d0, c1 = d1;
                if (kbg + 1 < KBG) { d0 = *(const f16x8c*)(sb + (kbg + 1) * 2048); d1 = *(const f16x8c*)(sb + (kbg + 1) * 2048 + 1024); }
#pragma unroll
                for (int i = 0; i < NT2; ++i) l1[i] = cl_mfma(c1, W1[i][kbg], l1[i]);
#pragma unroll
                for (int i = 0; i < NT2; ++i) l1[i] = cl_mfma(c0, W2[i][kbg], l1[i]);
#pragma unroll
                for (int i = 0; i < NT2; ++i) hi[i] = cl_mfma(c0, W1[i][kbg], hi[i]);
            }
            asm volatile("s_nop 15");
            char* const dst = psend + (size_t)(n % RING) * slot2b;
#pragma unroll
            for (int i = 0; i < NT2; ++i) {
                f32x4 bv = (hi[i] + l1[i] * (1.0f / CL_F16_LO)) * (1.0f / CL_F16_DSCALE);
#pragma unroll
                for (int e = 0; e < 4; ++e) bv[e] = __uint_as_float((__float_as_uint(bv[e]) & ~1u) | par);   // this lap's parity
                cl_store4((float*)(dst + (size_t)i * NG * 1024), bv, fast);
            }
        }
        CL_TICK(3);
        if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); CL_TICK(5); }      // (counters only: everything issued since the level-1 poll)
        f32x4 sum = z4;
        {   // one block per group, addressed to this member
            f32x4 v[NP2];
            p_tries += c16_poll<NP2, FAST>(v, pmine + (size_t)(n % RING) * slot2b, poff2, dead, a.fault,
                                           [par](const f32x4 (&w)[NP2]) { return c16_all_tagged<NP2>(w, par); });
            CL_TICK(4);
#pragma unroll
            for (int i = 0; i < NP2; ++i) sum += v[i];
        }
        *(f32x4*)(red + (wave * 64 + lane) * 16) = sum;
        __syncthreads();                                 // the four waves' sums visible; every wave is done reading the staged image
        CL_TICK(6);
        float pr[4];
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) pr[sw] = *(const float*)(red + (sw * 64 + lane) * 16 + wave * 4);
        __builtin_amdgcn_sched_barrier(0);
        // behind the poll, under the reads' latency: the step after next's saved activations
#pragma unroll
        for (int e = 0; e < 4; ++e) cur.sv[e] = c16_mov(nxt.sv[e]);
        cur.hprev = c16_mov(nxt.hprev); cur.cprev = c16_mov(nxt.cprev); cur.dhe = c16_mov(nxt.dhe);
        load_saved(t > 2 ? t - 2 : 0, nxt);
        __builtin_amdgcn_sched_barrier(0);
        dh += (pr[0] + pr[1]) + (pr[2] + pr[3]);
        CL_TICK(7);
    }
    if (prof && lane == 0 && tile * C + mem < min(32, a.Bp / 8)) {      // (the counters' region: Bp / 16 * 128 words per direction)
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

template <int CELL>
__global__ void __launch_bounds__(256, 2) rec_bwd_c16t(RecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    int tile, mem;
    if (!cl_ids(a, 32, a.Bp / 16, tile, mem)) return;
    bool dead = false;
    if (cl_same_xcc(a, 32, tile, mem, (int*)smem_c, dead)) c16_bwd2_body<CELL, true>(a, tile, mem, dead, smem_c);
    else c16_bwd2_body<CELL, false>(a, tile, mem, dead, smem_c);
}

template <int CELL, int HP>
__global__ void __launch_bounds__(256, 2) rec_bwd_c16(RecArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_c[];
    int tile, mem;
    if (!cl_ids(a, HP / 16, a.Bp / 16, tile, mem)) return;
    bool dead = false;
    if (cl_same_xcc(a, HP / 16, tile, mem, (int*)smem_c, dead)) c16_bwd_body<CELL, HP, true>(a, tile, mem, dead, smem_c);
    else c16_bwd_body<CELL, HP, false>(a, tile, mem, dead, smem_c);
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
// Hp = 512: the backward exchange in two levels (rec_bwd_c16t); SBR_C16_TWO_LEVEL=0: one level, as at Hp = 256
static bool c16_two_level() {
    static const int two = [] { const char* e = getenv("SBR_C16_TWO_LEVEL"); return e ? atoi(e) : 1; }();
    return two != 0;
}
template <int CELL, int HP>
static hipError_t fwd_c16(hipStream_t s, const RecArgs& a) {
    constexpr int G = Gates<CELL>::G;
    hipError_t e = hipMemsetAsync(a.xh, 0xFF, (size_t)4 * a.Bp * HP * sizeof(float), s);   // the ring's slots: sentinel
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)2 * 4 * G * 1024;
    CL_LAUNCH((rec_fwd_c16<CELL, HP>), HP / 16, 16, lds);
    return hipGetLastError();
}
template <int CELL, int HP>
static hipError_t bwd_c16(hipStream_t s, const RecArgs& a_in) {
    constexpr int G = Gates<CELL>::G;
    const size_t lds = 2 * 16 * (size_t)(G == 1 ? 64 : 128) + 4 * 1024;
    RecArgs a = a_in;
    a.dhe_on = a.dh_ext != nullptr;
    if (!a.dh_ext) a.dh_ext = a.hs;                      // requested every step, used only under dhe_on (no pointer select in the kernel)
    if constexpr (HP == 512) {
        if (c16_two_level()) {
            const size_t lds2 = (size_t)4 * (G == 1 ? 1 : 2) * 2048 + 4 * 1024;
            CL_LAUNCH((rec_bwd_c16t<CELL>), 32, 16, lds2);
            return hipGetLastError();
        }
    }
    CL_LAUNCH((rec_bwd_c16<CELL, HP>), HP / 16, 16, lds);
    return hipGetLastError();
}
#define C16_DISPATCH(FN) \
    if (a.Hp == 512) { \
        switch (a.cell) { case SBR_CELL_LSTM: return FN<CELL_LSTM, 512>(s, a); case SBR_CELL_GRU: return FN<CELL_GRU, 512>(s, a); \
                          default: return FN<CELL_VANILLA, 512>(s, a); } \
    } \
    switch (a.cell) { case SBR_CELL_LSTM: return FN<CELL_LSTM, 256>(s, a); case SBR_CELL_GRU: return FN<CELL_GRU, 256>(s, a); \
                      default: return FN<CELL_VANILLA, 256>(s, a); }

hipError_t launch_rec_forward_c16(hipStream_t s, const RecArgs& a) { C16_DISPATCH(fwd_c16) }
hipError_t launch_rec_backward_c16(hipStream_t s, const RecArgs& a) { C16_DISPATCH(bwd_c16) }
// The ring of partial-sum blocks starts a launch with every word's lowest bit set: lap 0 expects it clear (rec_bwd_c16).
// (two levels: + the sentinel of the level-1 images behind them; 1 MB per tile instead of 2)
hipError_t sbr_rec_c16_fill(hipStream_t s, const RecArgs& a) {
    size_t bytes = sbr_rec_c16_ring_floats(a.Bp, a.Hp) * sizeof(float);
    if (a.Hp == 512 && c16_two_level()) bytes = (size_t)(a.Bp / 16) * (SBR_C16_RING * 32 * 8 * 1024 + 4 * 8 * 16384);
    return hipMemsetAsync(a.pring, 0xFF, bytes, s);
}
