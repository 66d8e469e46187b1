// Pipelined recurrent kernels for layers of 128 units on 4-row tiles ("x6p"; GRU, Vanilla, and LSTM on fp16 planes): the
// kernels of BASELINE config C2.  Same arithmetic and global data layout as rec_fwd_x6s / rec_bwd_x6s (sbr_rec.hip: one workgroup
// of eight waves, two per SIMD, per 4-row tile for all T steps; f32 operands split exactly into three bf16 planes, six
// MFMA terms); the schedule inside the workgroup is rebuilt around three facts measured on the MI355X
// (tools/probes/mfma_share_probe.hip, valu_beside_mfma_probe.hip, lds_mask_probe.hip):
//
//   a. Waves w and w + 4 share a SIMD, and its matrix pipe serves the OLDER one first, strictly: two MFMA streams on
//      one SIMD finish in 1160 / 2321 cycles per 72 MFMAs, not 1740 / 1740.  s_setprio does not change that.
//   b. Beside a partner wave that streams MFMAs, a VALU instruction of the other wave costs ~20 cycles (one issue slot
//      per partner MFMA; 2.6 - 8.5 cycles alone), whichever wave is older.  Scalar, LDS and memory instructions cost
//      what they always cost.
//   c. A ds_read_b128 occupies the LDS pipe ~5 cycles whatever the exec mask.
//
// So a step is two serial MFMA phases per SIMD (2 x 1160 cycles) with every wave's other work running under its
// partner's phase, and it lasts max(both phases, one wave's phase + its other work at 20 cycles per VALU instruction).
// With a barrier per step (x6s) all eight waves do their gate math at the same time while the pipe idles: 3760 cycles.
//
//   1. No workgroup barrier in the step loop: LDS counters per group of four producer waves; a consumer reads counter
//      then planes (the LDS keeps a wave's order, so planes read after a sufficient counter value are the published
//      ones) and re-reads both while the counter is short.  Double buffering still suffices: a wave overwrites the
//      buffer of step t only after its step-(t+1) MFMAs, which needed every wave's step-(t+1) slice, which each wave
//      published after its own step-t operand reads.  Every spin is bounded and raises the fault flag
//      (sbr_read_cost reports it) instead of hanging the GPU.
//   2. A gate on the matrix pipe: waves 0-3 start their MFMA phase only when their partner has issued its own (fact a:
//      otherwise they starve the partner's last MFMAs, whose results everybody waits for).
//   3. All three W_hid planes in registers (144 VGPRs; a fourth gate fits as two fp16 planes only: LSTM runs the F16 forms,
//      see sbr_rec_x6p_ok); activations are the
//      A operand with every batch row filling four tile rows, so lane (j, q) finishes (row q, unit j) from accumulator
//      element 0 without a select; biases ride in as the MFMA C operand; sigmoid gates are pre-scaled by -log2(e).
//   4. Everything outside the MFMA phase is written for VALU count (fact b): scalar-advanced addresses, single-
//      instruction stores and counter updates, v_med3 clip, truncating bf16 split, one loop per role, no packed-f32
//      (SLP) gate math, a uniform branch instead of a select while no row is masked.
//
//   5. Forward only: products as a 2-way fp16 split, three MFMAs instead of six (f16x3, see split2_f16 below): 228 -> 175 us.
//
// Measured at C2 (cycles per step, 2320 = MFMA issue of bf16x6): forward 3760 -> 2650 (-> 2000 with f16x3), backward 4100 -> 3150.  Tried and
// rejected: all operands fetched before the MFMA phase by every wave (LDS burst, +35 us), a three-deep operand ring in
// the backward (+15 us), a delay at the gate or a signal more than one MFMA term early (+5 .. +35 us; one term early is what runs, -4 us), splitting the backward over K instead
// of over the output units (every gate-math step would then need all eight waves' partial sums: no overlap left).
// The backward's 36 operand reads per wave and step (K = 384; the forward has 12) are not what holds it back: with half
// of them skipped (a probe build, round 1) rec_bwd takes 256 instead of 263 us, with three quarters skipped 250 us.
#include "sbr_rec_p.h"

#ifndef X6P_BWD_NS
#define X6P_BWD_NS 4     // backward: operand slots / k-blocks fetched ahead of their MFMAs
#endif
#ifndef X6P_BWD_LA
#define X6P_BWD_LA 3     // (round 4: with one sparse instruction per k-block an LDS read one k-block ahead left every iteration waiting for it: 155 -> 136 us)
#endif
#ifndef X6P_FWD_V2
#define X6P_FWD_V2 6     // forward, packed form: 1 = the input row (+ bias) rides in as the MFMA C operand, 2 = the stores of step
#endif                   // t - 1 are issued inside the MFMA phase of step t, 4 = the next row offset is read before the MFMAs
#ifndef X6P_BWD_DEFER
#define X6P_BWD_DEFER 0  // backward, ring forms: the stores of step t leave from inside its own MFMA phase (behind the first k-block)
#endif
#ifndef X6P_FWD_TOK
#define X6P_FWD_TOK 1    // forward, packed form: waves 4-7 open the pipe gate this many MFMA groups (of G) before their last MFMA
#endif
#ifndef X6P_G4
#define X6P_G4 1         // the four saved gate values of a (t, row, unit) as ONE 16-byte element [t][row][unit][4] in the region of RecArgs.g[0..3]
#endif                   // (forward: one dwordx4 store per lane and step instead of four; backward: one ds_read_b128 from the ring); 0: round 3's four arrays
#ifndef X6P_SPEC1
#define X6P_SPEC1 0      // forward, sparse form: waves 0-3 fetch the second half of the operands at the top of the step too, on spec (measured:
                         // rec_fwd 113.5 -> 117 us -- two more ds_read_b128 per wave and step on an LDS pipe that is the busy resource)
#endif
#ifndef X6P_H1_AT
#define X6P_H1_AT 0      // forward, sparse form, waves 0-3: the second operand half is requested behind the instructions of this k-block (1: the
#endif                   // last one in front of its use, as in rounds 1 - 3 -- with three instructions per k-block that read's whole round trip was
                         // exposed: rec_fwd 115.0; 0: one k-block earlier: 112.8 us; -1: in front of the first k-block's instructions: 113.5)
#ifndef X6P_ROLES
#define X6P_ROLES 1      // 0: waves 0-3 run the loop of waves 4-7 too (all operands fetched at the top of the step, no pipe gate)
#endif
#ifndef X6P_TOK_EARLY
#define X6P_TOK_EARLY 1  // the pipe gate's token is read together with the step's first operand reads: a separate read was one exposed LDS
#endif                   // round trip (~140 cycles) per step of waves 0-3, and the token is there long before (profiles/round4_d_rec_phases.txt)
#ifndef X6P_GATE
#define X6P_GATE(a) false   // the matrix-pipe gate of rounds 1 - 3 (waves 0-3 wait for their partner's last MFMA): with one sparse instruction per
#endif                      // k-block it only cost its reads (SBR_X6_PIPE=2 was never faster after round 4); compiled out, -DX6P_GATE(a)=((a).x6_pipe>=2) brings it back
#ifndef X6P_SYNC
#define X6P_SYNC 0       // 1: one workgroup barrier per step instead of the counters / the pipe gate / the role split (experiment:
#endif                   // with two MFMAs per product the matrix phase is short enough for the synchronous schedule to compete)
#ifndef X6P_PACK
#define X6P_PACK 1       // 0: the fp16 forms with three MFMAs per product (round 2), for same-box A/B runs (SBR_LIB)
#endif
#ifndef X6P_SPARSE
#define X6P_SPARSE 1     // packed planes on the 2:4-sparse matrix instruction: ONE v_smfmac_f32_16x16x64_f16 per (gate, k-block), see
#endif                   // "Sparse planes" below; 0: two dense MFMAs (round 3)
#ifndef X6P_SP_INIT
#define X6P_SP_INIT 1    // sparse forms: the accumulators' initial value (bias | zeros) comes from LDS (ds_read_b128 at the top of the step;
                         // bit 0: forward, bit 1: backward) or from VALU moves (0)
#endif

namespace {

constexpr int HP = 128, R = 4, KBH = HP / 32;

// "f16x3": an f32 operand as a1 + a2 / 2048 with a1 = fp16(a), a2 = fp16((a - a1) * 2048) (the scale keeps the low part
// out of fp16's subnormal range), a product as a1 b1 + (a1 b2 + a2 b1) / 2048: three MFMAs instead of bf16x6's six, two
// accumulators.  Measured against fp64 on K = 128 dot products of recurrent-step operands (tools/probes/
// mfma_f16x3_probe.hip): rms error 1.1e-7 of the rms value (bf16x6 1.5e-7, a plain f32 FMA chain 2.1e-7), the same for
// activations of magnitude 1e-4; the matrix pipe keeps fp16 subnormal inputs.  Range: |a| < 65504 -- hidden states are
// in [-1, 1], weights far below; relative precision decays below |a| ~ 1e-4 * 2^-13 (absolute floor 1.5e-11), which is
// why only the FORWARD chain uses it: gradients span too many binades without a per-row scale.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x16 __attribute__((ext_vector_type(16)));
constexpr float F16_LO = 2048.0f;
// "Sparse planes" (round 4): one matrix instruction per (gate, k-block).  v_smfmac_f32_16x16x64_f16 multiplies a 2:4-sparse A (16 x 64,
// stored compressed as 16 x 32 + two index bits per kept element) with a dense B (64 x 16) in the 16 cycles the dense 16x16x32 takes
// (tools/probes/smfmac_probe.hip, profiles/round4_a_probe_smfmac.txt: 200 cycles per 12, one dependent accumulator chain included).
// The 4-row tile's sixteen A rows are (batch row, copy c); with the index bits each copy picks its OWN half of a B operand that
// interleaves the two weight planes: every group of four dense K positions is (W1[k], W2[k], W1[k+1], W2[k+1]), copy c keeps
// positions (P, 2 + P) with P = c & 1 and carries plane c >> 1 of the activations.  One instruction then leaves
//     accumulator element 0: a1 w1     1: a1 w2     2: a2 w1     3: a2 w2 (below f32 rounding, unused)
// for the lane's (row, unit) -- the three products the packed form needs two dense MFMAs for.  Operand layout, found with one-hot
// operands by the probe: A lane (i, qa) holds the kept elements of dense k = 16 qa .. 16 qa + 15 (eight values: real k = 8 qa + r,
// one ds_read_b128 of the plane as before), index field r at bits [2r + 1 : 2r] of the lane's own index register (ABID picks the 16-bit
// half; both halves are filled); B lane (j, qb) element e holds dense k = 32 (e >> 3) + 8 qb + (e & 7), i.e. plane e & 1 of
// real k = 16 (e >> 3) + 4 qb + ((e & 7) >> 1).  smfmac accumulates in place (no C operand): a step's accumulators start from
// {bias, 0, 0, 0} by moves (X6P_SP_INIT 0) or LDS reads (1).
__device__ __forceinline__ f32x4 smfmac16(const f16x8& a, const f16x16& b, const f32x4& c, int idx) {
    return __builtin_amdgcn_smfmac_f32_16x16x64_f16(a, b, c, idx, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(const bf16x8& a, const bf16x8& b, const f32x4& c) { return MFMA_BF16(a, b, c); }
[[maybe_unused]] __device__ __forceinline__ f32x4 mfma16(const f16x8& a, const f16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// v must be a value the compiler cannot look through (callers pin it with an empty asm): when it can see the expression
// that produced v, it forms a1 twice -- v_fma_mixlo_f16 straight from that expression (one rounding) for the subtraction,
// v_cvt_pk_f16_f32 of the f32-rounded value for the stored plane -- and the two differ in one element of ~2^13 (double
// rounding), each worth an fp16 ulp of that operand (found as a 30x larger error in one hidden unit of GRU [64, 128]).
__device__ __forceinline__ void split2_f16(float v, _Float16& a1, _Float16& a2) {
    a1 = (_Float16)v;
    a2 = (_Float16)((v - (float)a1) * F16_LO);
}

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
template <int CELL, bool FUSE, bool PROF, bool F16>
__global__ void __launch_bounds__(512) rec_fwd_x6p(RecArgs a) {
    using OPV = std::conditional_t<F16, f16x8, bf16x8>;      // operand fragment: fp16 (two planes) or bf16 (three planes)
    constexpr int NP = F16 ? 2 : 3;
    constexpr bool PK = F16 && X6P_PACK;                     // packed planes: two MFMAs per product, see "Packed planes" above
    constexpr bool SP = PK && X6P_SPARSE;                    // ... one sparse instruction per product, see "Sparse planes" above
    constexpr int G = Gates<CELL>::G, KB = KBH, GHP = G * HP;
    static_assert(G <= 3 || F16, "three W_hid planes of four gates do not fit the register file: an LSTM runs on the two fp16 planes only");
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
    int spin_limit = X6P_SPIN_LIMIT;                     // (drops to 64 once this wave has raised the fault flag)
    const unsigned lds_tok = (unsigned)(size_t)(tok + (wave & 3));
    const int one = 1;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 16));
    tmax = __builtin_amdgcn_readfirstlane(max(tmax, __shfl_xor(tmax, 32)));  // workgroup-uniform: all four rows

    OPV W1[SP ? 1 : G][SP ? 1 : KB], W2[SP ? 1 : G][SP ? 1 : KB], W3[F16 ? 1 : G][F16 ? 1 : KB];   // B operands: lane (j, q) holds W[kb*32 + 8q + e][unit j]
    f16x16 WS[SP ? G : 1][SP ? KB : 1];                  // SP: both planes interleaved, lane (j, q) holds real k = 16 (e >> 3) + 4 q + ((e & 7) >> 1)
    if constexpr (SP) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb)
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float sc = (CELL == CELL_GRU && g < 2) ? X6P_NLOG2E : 1.0f;
                    float w = sc * a.Whid[(size_t)(kb * 32 + 16 * (r >> 2) + 4 * q + (r & 3)) * GHP + g * HP + u];
                    _Float16 b1, b2;
                    asm("" : "+v"(w));                       // see split2_f16 (not volatile)
                    split2_f16(w, b1, b2);
                    WS[g][kb][2 * r] = b1; WS[g][kb][2 * r + 1] = b2;
                }
    } else
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float sc = (CELL == CELL_GRU && g < 2) ? X6P_NLOG2E : 1.0f;     // sigmoid gates: see the gate math
                float w = sc * a.Whid[(size_t)(kb * 32 + 8 * q + e) * GHP + g * HP + u];
                if constexpr (F16) {
                    _Float16 b1, b2;
                    asm("" : "+v"(w));                                  // see split2_f16.  NOT volatile: volatile asms keep their order and
                                                                        // serialised the 96 loads of this prologue (+20 us per launch)
                    split2_f16(w, b1, b2);
                    W1[g][kb][e] = b1; W2[g][kb][e] = b2;
                } else {
                    __bf16 b1, b2, b3;
                    split3(w, b1, b2, b3);
                    W1[g][kb][e] = b1; W2[g][kb][e] = b2; W3[g][kb][e] = b3;
                }
            }

    // per-lane byte offsets, computed once; the per-step part of every address is uniform and advances on the SALU
    const unsigned bo_h = (unsigned)(row * HP + u) * 4u;                                       // hs rows
    const unsigned bo_g = X6P_G4 ? (unsigned)(row * HP + u) * 16u : (unsigned)sbr_blocked_index(0, row, u, Bp, HP) * 4u;   // saved activations
    const unsigned bo_x = (unsigned)(row * GHP + u) * 4u;                                      // xt rows (not fused)
    const size_t st_h = (size_t)Bp * HP * 4, st_x = (size_t)Bp * GHP * 4;                      // bytes per time step

    float h = a.hinit[u];
    stf(a.hs, bo_h, h);
    float c = 0.f, pi = 0.f, pf = 0.f, po = 0.f;                 // LSTM: cell state and the peepholes of this lane's unit
    if (CELL == CELL_LSTM) {
        c = a.cinit[u]; pi = a.peep[u]; pf = a.peep[HP + u]; po = a.peep[2 * HP + u];
        stf(a.cs, bo_h, c);
    }
    const unsigned lds_pub = (unsigned)(q * HROW + u * 2);            // where this lane's h goes inside a plane set
    // A operand: tile row m = j holds batch row j >> 2; PK: plane j & 1 of it (rows 4r + 2, 4r + 3 repeat rows 4r, 4r + 1)
    // (SP: tile rows 4r, 4r + 1 hold plane 0, rows 4r + 2, 4r + 3 plane 1; the row's index bits choose the weight plane)
    const unsigned lds_rd = (unsigned)((j >> 2) * HROW + q * 16 + (SP ? ((j >> 1) & 1) * PLANEB : PK ? (j & 1) * PLANEB : 0));
    const int spidx = (j & 1) ? (int)0xDDDDDDDDu : (int)0x88888888u;      // kept positions of every group of four: (1, 3) / (0, 2)
    auto publish_h = [&](int buf) {
        char* base = hbuf + buf * BUFB + lds_pub;
        if constexpr (F16) {
            _Float16 h1, h2;
            split2_f16(h, h1, h2);
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

    // Input of step t: a row of xt, or (layer 0, one index per step) gathered here: W_in[id[row][t]] (+ b as the
    // MFMA C operand) (sparse_lstm.py:755 / :1111).  Row t+1 is requested at the end of step t and used after the
    // MFMAs of step t+1; its id was requested one step before that.  Time indices are clamped, not branched around.
    // The bias rides in as the C operand of a gate's first MFMA, except for GRU's candidate gate, whose recurrent
    // part is multiplied by r before the input part (with its bias) is added (sparse_lstm.py:786-792).
    float x[G];
    f32x4 biasv[G];
    float bias_c = 0.f, bsc[G], xc_pre = 0.f;
    unsigned bo_nxt2 = 0;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float b = FUSE ? a.gbias[g * HP + u] : 0.f;
        if (CELL == CELL_GRU && g == 2) { bias_c = b; b = 0.f; }
        if (CELL == CELL_GRU && g < 2) b *= X6P_NLOG2E;
        bsc[g] = b;
        biasv[g] = PK ? f32x4{b, 0.f, 0.f, 0.f} : f32x4{b, b, b, b};      // PK: element 1 collects the low-order products
    }
    // FUSE: the rows travel XPD time steps ahead of their use through a ring in LDS (LDS-DMA, sbr_rec_p.h: one dword per lane
    // and gate; invisible to the compiler's vmcnt bookkeeping, waited for by hand), their byte offsets inside W_in come from
    // a table the prologue builds in LDS from the tile's ids (R x T entries).  With the row one step ahead in registers and
    // the id a step before that, every step of the fp16 chain (half the MFMA time of bf16x6) waited for far memory, and
    // the address arithmetic (a 64-bit multiply-add, a 64-bit add) sat on the VALU beside the partner's MFMA stream:
    // 177 us against 153 us for the same kernel reading a pre-gathered xt.
    // Per iteration the wave issues: stores of step t (NSF) < the DMA piece of step t + XPD.
    constexpr int XPD = 4, NSF = CELL == CELL_VANILLA ? 1 : (CELL == CELL_LSTM ? 2 : 1) + (X6P_G4 ? 1 : 4), XSTG = G * 256;
    constexpr int XOFF_OFF = (2 * BUFB + 64 + 255) & ~255;
    const int xring_off = (XOFF_OFF + R * T * 4 + 255) & ~255;
    unsigned* xo_tab = (unsigned*)(smem_p + XOFF_OFF);
    const unsigned xring_wave = (unsigned)(size_t)(smem_p + xring_off) + (unsigned)wave * (XPD * XSTG);
    const char* xring_lane = smem_p + xring_off + wave * (XPD * XSTG) + lane * 4;
    // ONE 16-byte-per-lane piece per step: lane l < 16 G fetches the four units 4 (l % 4) .. of gate l / 16 from the W_in row of
    // batch row (l / 4) % 4; it lands as [gate][row][unit], where lane (j, q) finds gate g at 256 g + 4 (16 q + j)
    const unsigned* xo_row = xo_tab + ((lane >> 2) & 3) * T;     // the table row of the batch row this lane fetches for
    const unsigned bo_lane = (unsigned)((lane >> 4) * HP + wave * 16 + (lane & 3) * 4) * 4u;
    constexpr unsigned long long XMASK = G >= 4 ? ~0ull : ((1ull << (16 * G)) - 1ull);
    unsigned bo_nxt = 0;                                         // byte offset the NEXT piece fetches from
    auto dma_x = [&](unsigned bo, int slot) {
        lds_dma_x4(xring_wave + (unsigned)slot * XSTG, a.gWin, bo, XMASK);
    };
    auto load_x = [&](int t) {                                   // not fused: a row of xt, one step ahead in registers
        const char* xt_t = (const char*)a.xt + (size_t)min(t, T - 1) * st_x;
#pragma unroll
        for (int g = 0; g < G; ++g) x[g] = ldf(xt_t, bo_x, g * HP * 4);
    };
    // SP, X6P_SP_INIT: {bias, 0, 0, 0} of every (gate, unit) in LDS -- a step's accumulators start as one ds_read_b128 each
    const f32x4* bz_lane = (const f32x4*)(smem_p + (FUSE ? ((xring_off + 8 * XPD * XSTG + 255) & ~255) : XOFF_OFF)) + u;
    if constexpr (SP && (X6P_SP_INIT & 1)) {
        if (q == 0) {
#pragma unroll
            for (int g = 0; g < G; ++g) ((f32x4*)bz_lane)[g * HP] = biasv[g];
        }
    }
    if constexpr (FUSE) {
        for (int i = threadIdx.x; i < R * T; i += 512) {
            const int r = i / T;
            xo_tab[i] = (unsigned)a.gX[(size_t)(blockIdx.x * R + r) * T + (i - r * T)] * (unsigned)(GHP * 4);   // < 2^32: checked by the launcher
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
    unsigned long long p_c0 = 0, p_r0 = 0, p_spin = 0, p_tok = 0, p_seg[3] = {0, 0, 0}, p_ta = 0, p_tb = 0;
    if (PROF) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    unsigned long long* tl = PROF && blockIdx.x == 0 && lane == 0 ? a.prof + (8 + wave) * 8 : nullptr;   // step-100 timeline

    float sv[4] = {0.f, 0.f, 0.f, 0.f};
    // X6P_FWD_V2 & 2: the NSF stores of step t - 1 leave from inside the MFMA phase of step t -- behind its first k-block, where
    // the issue slots are free -- instead of between the publication and the next step's operand reads.  (Fused gather: the
    // wait at the top of the loop counts NSF stores + one DMA piece per iteration; the rows of steps < XPD have landed before
    // the loop, so the first iteration's missing stores change nothing.)
    constexpr bool DEFER = PK && (X6P_FWD_V2 & 2);
    float sv_st[4] = {0.f, 0.f, 0.f, 0.f}, h_st = 0.f, c_st = 0.f;
    auto store_step = [&](size_t off, const float* svv, float hv, float cv) {
        if (CELL != CELL_VANILLA) {
            if constexpr (X6P_G4) st_s4((const char*)a.g[0] + 4 * off, bo_g, f32x4{svv[0], svv[1], svv[2], svv[3]});
            else {
#pragma unroll
                for (int k = 0; k < 4; ++k) st_s((const char*)a.g[k] + off, bo_g, svv[k]);
            }
        }
        st_s((const char*)a.hs + off + st_h, bo_h, hv);
        if (CELL == CELL_LSTM) st_s((const char*)a.cs + off + st_h, bo_h, cv);
    };
    size_t off_t = 0;                                              // t * st_h
    int tmin = mylen;                                              // steps below it: no row of the tile is masked
    tmin = min(tmin, __shfl_xor(tmin, 16));
    tmin = __builtin_amdgcn_readfirstlane(min(tmin, __shfl_xor(tmin, 32)));
    const unsigned lds_cnt0 = (unsigned)(size_t)cnt;              // LDS addresses of the two publish counters
    const unsigned my_cnt = roleA ? lds_cnt0 : lds_cnt0 + 4;      // (the wave's group, whichever loop it runs: X6P_ROLES)
    // One loop per role (waves 0-3 / 4-7): a role test inside the loop costs VALU instructions in every step, some
    // of them between MFMAs, and values defined under it get copies at the join (see sbr_rec_cl.hip).
    auto steps = [&](auto role_tag) {
    constexpr bool RA = decltype(role_tag)::value;
    int xslot = 0;                                                // FUSE: ring slot of step t = t % XPD
    for (int t = 0; t < tmax; ++t) {
        if (PROF) p_ta = clock64();
        if (PROF && tl && (t == 100 || t == 101)) tl[t == 100 ? 0 : 7] = p_ta;
        if constexpr (FUSE) {
            // the row of this step: its DMA was issued XPD iterations ago; younger than it are XPD - 1 whole iterations
            wait_vm<(XPD - 1) * (NSF + 1)>();
            const char* xp = xring_lane + xslot * XSTG;
#pragma unroll
            for (int g = 0; g < G; ++g) x[g] = *(const float*)(xp + g * 256);
        }
        if constexpr (PK && (X6P_FWD_V2 & 1)) {                    // x (+ bias) into element 0 of the C operands: two ops less behind the MFMAs
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (CELL == CELL_GRU && g == 2) xc_pre = x[2] + bias_c;
                else if (CELL == CELL_GRU) biasv[g][0] = fmaf(x[g], X6P_NLOG2E, bsc[g]);
                else biasv[g][0] = x[g] + bsc[g];
            }
        }
        if constexpr (FUSE && PK && (X6P_FWD_V2 & 4)) {            // (LDS latency under the MFMA phase instead of in front of the next step)
            bo_nxt2 = xo_row[min(t + XPD + 1, T - 1)];
        }
        // ---- N1
        f32x4 acc0[G];                                            // SP: what the step's accumulators start from
        if constexpr (SP && (X6P_SP_INIT & 1)) {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int g = 0; g < G; ++g) acc0[g] = bz_lane[g * HP];
        }
        const char* hb = hbuf + (t & 1) * BUFB + lds_rd;
        OPV hp[KB][3];
        int fl[2];
        int tokv = 0;                                             // RA: the pipe gate's token, read with the first operands (one LDS
        auto load_half = [&](int half) {                          // round trip instead of two).  Counter first, then planes: the LDS keeps a wave's order
            fl[half] = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (X6P_TOK_EARLY && RA && half == 0 && X6P_GATE(a)) tokv = __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int kb = 2 * half; kb < 2 * half + 2; ++kb) {
                hp[kb][0] = *(const OPV*)(hb + kb * 64);
                hp[kb][1] = PK ? hp[kb][0] : *(const OPV*)(hb + kb * 64 + PLANEB);
                hp[kb][2] = NP == 3 ? *(const OPV*)(hb + kb * 64 + 2 * PLANEB) : hp[kb][1];
            }
        };
        auto ensure_half = [&](int half) {                        // the four producers of this half have published h_t
            if (!X6P_SYNC && __builtin_amdgcn_readfirstlane(fl[half]) < 4 * t) {
                unsigned long long w0 = 0;
                if (PROF) w0 = clock64();
                int spins = 0;
#pragma clang loop unroll(disable)
                do {
                    asm volatile("" ::: "memory");
                    load_half(half);
                    if (++spins > spin_limit) { atomicOr(a.fault, 2); spin_limit = 64; break; }   // bounded: never hang the GPU (and a broken launch ends soon)
                } while (__builtin_amdgcn_readfirstlane(fl[half]) < 4 * t);
                if (PROF) p_spin += clock64() - w0;
            }
            // the planes are waited for HERE, where the two paths join: no counter waits between the MFMAs
            asm volatile("" :: "v"(hp[2 * half][0]), "v"(hp[2 * half][1]), "v"(hp[2 * half][2]),
                               "v"(hp[2 * half + 1][0]), "v"(hp[2 * half + 1][1]), "v"(hp[2 * half + 1][2]));
        };
        load_half(0);
        if (!RA || (SP && X6P_SPEC1)) load_half(1);               // (waves 0-3, SP: on spec -- their partners' half is normally published by now)
        ensure_half(0);
        if (!RA) ensure_half(1);
        // The matrix pipe serves the OLDER wave of a SIMD pair first, strictly (tools/probes/mfma_share_probe.hip: two
        // MFMA streams on one SIMD run 1160 / 2321 cycles per 72, not 1740 / 1740).  A wave 0-3 that started its step
        // as soon as its own group's k-blocks were there would starve its partner's last MFMAs, whose results
        // everybody waits for: so it holds back until the partner has issued its whole step.  Waves 4-7 need no gate,
        // they only ever get the gaps.
        if (RA && X6P_GATE(a)) {
            unsigned long long w0 = 0;
            if (PROF) w0 = clock64();
            int v = X6P_TOK_EARLY ? tokv : __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), spins = 0;
#pragma clang loop unroll(disable)
            while (__builtin_amdgcn_readfirstlane(v) < t) {
                v = __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (++spins > spin_limit) { atomicOr(a.fault, 4); spin_limit = 64; break; }
            }
            if (PROF) p_tok += clock64() - w0;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PROF) { p_tb = clock64(); p_seg[0] += p_tb - p_ta; }
        // ---- M
        f32x4 acc[G], acl[G];                                     // acl: the low-order products of the fp16 form
#define X6P_TERM(HOP, WOP) _Pragma("unroll") for (int g = 0; g < G; ++g) acc[g] = mfma16(HOP, WOP, acc[g]);
#define X6P_TERL(HOP, WOP) _Pragma("unroll") for (int g = 0; g < G; ++g) acl[g] = mfma16(HOP, WOP, acl[g]);
        if constexpr (SP) {
            // acc[g]: {h1 w1 (+ bias), h1 w2, h2 w1, h2 w2}: one sparse instruction per (gate, k-block)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g] = (X6P_SP_INIT & 1) ? acc0[g] : biasv[g];
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb == KB / 2 && RA) ensure_half(1);
                __builtin_amdgcn_sched_barrier(0);
                if (PROF && tl && t == 100) tl[1 + kb] = clock64();
                if (!RA && kb == KB - 1 && X6P_FWD_TOK >= 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }
                if (X6P_H1_AT < 0 && kb == 0 && RA) {             // the second operand half: requested in FRONT of the first k-block's instructions
                    load_half(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = smfmac16(hp[kb][0], WS[g][kb], acc[g], spidx);
                if (X6P_H1_AT >= 0 && kb == (X6P_H1_AT < KB / 2 - 1 ? X6P_H1_AT : KB / 2 - 1) && RA) {   // ... or behind those of k-block X6P_H1_AT
                    __builtin_amdgcn_sched_barrier(0);
                    if (!X6P_SPEC1 || __builtin_amdgcn_readfirstlane(fl[1]) < 4 * t) load_half(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (DEFER && kb == 0 && t > 0) { store_step(off_t - st_h, sv_st, h_st, c_st); __builtin_amdgcn_sched_barrier(0); }
                if (!RA && kb == KB - 1 && X6P_FWD_TOK == 0) { if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][0] = fmaf(acc[g][1] + acc[g][2], 1.0f / F16_LO, acc[g][0]);
        } else if constexpr (PK) {
            // acc[g][0] = h1 w1 (+ bias), acc[g][1] = h2 w1, acl[g][0] = h1 w2 (acl[g][1] = h2 w2: below f32 rounding, dropped)
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb == KB / 2 && RA) ensure_half(1);
                __builtin_amdgcn_sched_barrier(0);
                if (PROF && tl && t == 100) tl[1 + kb] = clock64();
                if (!RA && kb == KB - 1 && X6P_FWD_TOK == 2) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int g = 0; g < G; ++g) { acl[g] = mfma16(hp[kb][0], W2[g][kb], kb == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acl[g]); }
                if (kb == KB / 2 - 1 && RA) {                     // see the bf16 form below
                    __builtin_amdgcn_sched_barrier(0);
                    load_half(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!RA && kb == KB - 1 && X6P_FWD_TOK == 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int g = 0; g < G; ++g) { acc[g] = mfma16(hp[kb][0], W1[g][kb], kb == 0 ? biasv[g] : acc[g]); }
                __builtin_amdgcn_sched_barrier(0);
                if (DEFER && kb == 0 && t > 0) { store_step(off_t - st_h, sv_st, h_st, c_st); __builtin_amdgcn_sched_barrier(0); }
                if (!RA && kb == KB - 1 && X6P_FWD_TOK == 0) { if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][0] = fmaf(acc[g][1] + acl[g][0], 1.0f / F16_LO, acc[g][0]);
        } else if constexpr (F16) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                if (kb == KB / 2 && RA) ensure_half(1);
                __builtin_amdgcn_sched_barrier(0);
                if (PROF && tl && t == 100) tl[1 + kb] = clock64();
                if (kb == 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) acl[g] = mfma16(hp[kb][1], W1[g][kb], f32x4{0.f, 0.f, 0.f, 0.f});
                } else { X6P_TERL(hp[kb][1], W1[g][kb]) }
                if (kb == KB / 2 - 1 && RA) {                     // see the bf16 form below
                    __builtin_amdgcn_sched_barrier(0);
                    load_half(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                X6P_TERL(hp[kb][0], W2[g][kb])
                if (!RA && kb == KB - 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }
                if (kb == 0) {
#pragma unroll
                    for (int g = 0; g < G; ++g) acc[g] = mfma16(hp[kb][0], W1[g][kb], biasv[g]);
                } else { X6P_TERM(hp[kb][0], W1[g][kb]) }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < G; ++g) acc[g][0] = fmaf(acl[g][0], 1.0f / F16_LO, acc[g][0]);
        } else {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb == KB / 2 && RA) ensure_half(1);
            __builtin_amdgcn_sched_barrier(0);
            if (PROF && tl && t == 100) tl[1 + kb] = clock64();
            if (kb == 0) {
#pragma unroll
                for (int g = 0; g < G; ++g) acc[g] = mfma16(hp[kb][0], W3[g][kb], biasv[g]);
            } else { X6P_TERM(hp[kb][0], W3[g][kb]) }
            X6P_TERM(hp[kb][2], W1[g][kb])
            X6P_TERM(hp[kb][1], W2[g][kb])
            if (kb == KB / 2 - 1 && RA) {                         // late enough for the partner's gate math to have
                __builtin_amdgcn_sched_barrier(0);                // published, early enough to hide the LDS latency
                load_half(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            X6P_TERM(hp[kb][0], W2[g][kb])
            X6P_TERM(hp[kb][1], W1[g][kb])
            // The partner needs >= ~100 cycles to notice the counter (LDS add + its polling read): tell it G MFMAs (48
            // cycles at GRU) before the last one is issued, so that less of that reaction time is idle matrix pipe.  Not
            // earlier: an older wave that starts while this one still has MFMAs to issue stalls them for its whole first
            // half (measured: 2 G MFMAs early gains nothing in the forward, 3 terms early loses 5 us).
            if (!RA && kb == KB - 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }
            X6P_TERM(hp[kb][0], W1[g][kb])
            __builtin_amdgcn_sched_barrier(0);
        }
        }
#undef X6P_TERM
#undef X6P_TERL
        // MFMA D -> VALU read hazard: hipcc pads it inside a basic block (the accumulators are read right below); the
        // profiling build has branches in between and pads by hand (see rec_fwd_mfma)
        if (PROF) asm volatile("s_nop 15");
        if (PROF) { const unsigned long long tc = clock64(); p_seg[1] += tc - p_tb; p_tb = tc; }
        if (PROF && tl && t == 100) tl[5] = p_tb;
        // ---- N2: gate math (sparse_lstm.py:780-803 / :1133-1150).  For GRU the r and u columns of W_hid and their
        // bias were scaled by -log2(e) when the planes were built, so a sigmoid is fma, exp2, add, rcp.
        {
            float hn;
            constexpr bool XC = PK && (X6P_FWD_V2 & 1);                // x already sits in the accumulators
            if (CELL == CELL_GRU) {
                constexpr int IU = G > 1 ? 1 : 0, IC = G > 2 ? 2 : 0;
                const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(XC ? acc[0][0] : fmaf(x[0], X6P_NLOG2E, acc[0][0])));
                const float ug = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(XC ? acc[IU][0] : fmaf(x[IU], X6P_NLOG2E, acc[IU][0])));
                const float hc = acc[IC][0];
                const float cc = tanh_fast(fmaf(rg, hc, XC ? xc_pre : x[IC] + bias_c));
                hn = fmaf(ug, cc - h, h);                         // (1 - u) h + u c
                sv[0] = rg; sv[1] = ug; sv[2] = cc; sv[3] = hc;
            } else if (CELL == CELL_LSTM) {                       // sparse_lstm.py:397-423 (peepholes, masked rows copy h and c)
                float aa[G];
#pragma unroll
                for (int g = 0; g < G; ++g) aa[g] = acc[g][0];
                hn = h;
                const float xz[4] = {0.f, 0.f, 0.f, 0.f};
                cell_forward<CELL_LSTM, true>(XC ? xz : x, aa, t < mylen, hn, c, pi, pf, po, sv);
            } else {
                { const float pre = XC ? acc[0][0] : x[0] + acc[0][0]; hn = a.relu ? fmaxf(pre, 0.0f) : tanh_fast(pre); }
            }
            if (CELL == CELL_LSTM) { h = hn; if (F16) asm volatile("" : "+v"(h)); }   // (cell_forward has applied the mask)
            else if (t < tmin) { asm volatile("" : "+v"(hn)); h = hn; }            // uniform branch: no select while no row is masked
            else { h = t < mylen ? hn : h; if (F16) asm volatile("" : "+v"(h)); }   // (pinned for split2_f16 either way)
        }
        if (t + 1 < tmax) {
            publish_h((t + 1) & 1);
            if (X6P_SYNC) __syncthreads(); else
            lds_inc(my_cnt, one);
            if (PROF) p_seg[2] += clock64() - p_tb;
            if (PROF && tl && t == 100) tl[6] = clock64();
        }
        if constexpr (DEFER) {                                    // stored from inside the next step's MFMA phase (or behind the loop)
#pragma unroll
            for (int k = 0; k < 4; ++k) sv_st[k] = sv[k];
            h_st = h; c_st = c;
        } else store_step(off_t, sv, h, c);                       // what BPTT needs of step t
        off_t += st_h;
        if constexpr (FUSE) {
            dma_x(bo_nxt, xslot);                                 // the row of step t + XPD into the slot this step has read
            xslot = xslot + 1 == XPD ? 0 : xslot + 1;
            if constexpr (PK && (X6P_FWD_V2 & 4)) bo_nxt = bo_nxt2 + bo_lane;
            else bo_nxt = xo_row[min(t + XPD + 1, T - 1)] + bo_lane;
        } else load_x(t + 1);
    }
    };
    if (roleA && !X6P_SYNC && X6P_ROLES) steps(std::true_type{}); else steps(std::false_type{});
    if (DEFER && tmax > 0) store_step(off_t - st_h, sv_st, h_st, c_st);      // the last step's
    for (int t = tmax; t < T; ++t) {                              // past the tile's longest row: the state is carried
        stf((char*)a.hs + off_t + st_h, bo_h, h);
        if (CELL == CELL_LSTM) stf((char*)a.cs + off_t + st_h, bo_h, c);
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
// backward (GRU / Vanilla).  Same lane <-> (row, unit) mapping, counters and pipe gate as the forward kernel.
//   dh_{t-1}[row][unit] += sum_k dhi_t[row][k] * W_hid[unit][k],  k over the G*HP gate columns (12 k-blocks at GRU)
// A operand = dhi planes (LDS, published by the gate math of all eight waves), B operand = W_hid rows of the wave's 16
// units (three planes, registers).  K is three times the forward's, so the operand planes cannot all be fetched up
// front: k-block i+1 is read while k-block i's six MFMAs run.  k-blocks are visited producer-group-major: first the
// six whose columns belong to units 0-63 (waves 0-3), then the six of waves 4-7, which waves 0-3 reach just after
// their partner has published them.
// Chunked BPTT protocol (t_lo / t_hi / state / part) as in rec_bwd_x6s.
// ---------------------------------------------------------------------------------------
// F16: the products as the forward's 2-way fp16 split (three MFMAs per block instead of six).  The A operand is a GRADIENT
// here; what makes the fp16 range safe is the reference's own clip: every element of dhi has passed grad_clip(+-100)
// (sparse_lstm.py:768-772, :789-791; recurrent_layers.py:19), so dhi * 2^9 < 65504 always, and the split keeps an absolute
// floor of 2^-36 / 2^9 = 3e-14 below that (gradients of this path are 1e-12 .. 1e-2: f32-class relative error down to
// ~1e-9, then absolute).  The launcher takes this form only while 0 < clip <= 100.
// WT: the overlapped step tail (sbr_backward_recurrent).  dxt / dhi leave the CU write-through and every wave publishes
// how far it has come (RecArgs.progress), so that the weight-gradient GEMM and the embedding scatter-add of a finished
// chunk of time steps can run on the idle CUs while the chain continues.  A wave knows its stores of step t+1 are complete
// when the loads it issued after them have returned (the vector memory counter retires in order) -- at the top of step t,
// where the gate math needs those loads anyway: the explicit vmcnt(0) there waits for nothing new.
constexpr float F16_DSCALE = 512.0f;
// WTM: 0 = the saved activations one step ahead in registers (compiler-visible loads); 1 = through the LDS ring, write-through
// stores, progress words (the overlapped tail); 2 = through the LDS ring only (plain stores, nothing published).
template <int CELL, bool EXT, bool PROF, bool F16, int WTM = 0>
__global__ void __launch_bounds__(512) rec_bwd_x6p(RecArgs a) {
    constexpr bool WT = WTM == 1, RING = WTM != 0;
    using OPV = std::conditional_t<F16, f16x8, bf16x8>;
    constexpr int NP = F16 ? 2 : 3;
    constexpr bool PK = F16 && X6P_PACK;                 // packed planes: two MFMAs per product ("Packed planes" above)
    constexpr bool SP = PK && X6P_SPARSE;                // ... one sparse instruction per product ("Sparse planes" above)
    constexpr int G = Gates<CELL>::G, GHP = G * HP, KB = GHP / 32, KU = HP / 32;     // KU k-blocks per gate
    static_assert(G <= 3 || F16, "three W_hid planes of four gates do not fit the register file: an LSTM runs on the two fp16 planes only");
    constexpr int DROW = GHP * 2 + 32, PLANEB = R * DROW, BUFB = NP * PLANEB;
    extern __shared__ __attribute__((aligned(16))) char smem_p[];
    char* dbuf = smem_p;                                 // [2][3 planes][R rows][DROW]
    int* cnt = (int*)(dbuf + 2 * BUFB);
    int* tok = cnt + 4;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * R + q;
    const int u = wave * 16 + j;
    const int T = a.T, Bp = a.Bp;
    const float clip = a.clip;
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    const bool roleA = wave < 4;
    int spin_limit = X6P_SPIN_LIMIT;
    const unsigned lds_tok = (unsigned)(size_t)(tok + (wave & 3));
    const int one = 1;

    const int mylen = a.len[row];
    int tmax = mylen;
    tmax = max(tmax, __shfl_xor(tmax, 16));
    tmax = __builtin_amdgcn_readfirstlane(max(tmax, __shfl_xor(tmax, 32)));

    OPV W1[SP ? 1 : KB], W2[SP ? 1 : KB], W3[F16 ? 1 : KB];               // B operands: lane (j, q) holds W_hid[unit j][kb*32 + 8q + e]
    f16x16 WS[SP ? KB : 1];                              // SP: both planes interleaved, columns kb*32 + 16 (e >> 3) + 4 q + ((e & 7) >> 1)
    if constexpr (SP) {
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float* src = a.Whid + (size_t)u * GHP + kb * 32 + 4 * q;
            const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 16);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float w = r < 4 ? lo[r & 3] : hi[r & 3];
                _Float16 b1, b2;
                asm("" : "+v"(w));                       // see split2_f16 (not volatile)
                split2_f16(w, b1, b2);
                WS[kb][2 * r] = b1; WS[kb][2 * r + 1] = b2;
            }
        }
    } else
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const float* src = a.Whid + (size_t)u * GHP + kb * 32 + 8 * q;
        const f32x4 lo = *(const f32x4*)src, hi = *(const f32x4*)(src + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float w = e < 4 ? lo[e & 3] : hi[e & 3];
            if constexpr (F16) {
                _Float16 b1, b2;
                asm("" : "+v"(w));                       // see split2_f16 (not volatile: the loads of this prologue stay unordered)
                split2_f16(w, b1, b2);
                W1[kb][e] = b1; W2[kb][e] = b2;
            } else {
                __bf16 b1, b2, b3;
                split3(w, b1, b2, b3);
                W1[kb][e] = b1; W2[kb][e] = b2; W3[kb][e] = b3;
            }
        }
    }

    const unsigned bo_h = (unsigned)(row * HP + u) * 4u;
    const unsigned bo_g = X6P_G4 ? (unsigned)(row * HP + u) * 16u : (unsigned)sbr_blocked_index(0, row, u, Bp, HP) * 4u;
    const unsigned bo_x = (unsigned)(row * GHP + u) * 4u;
    const size_t st_h = (size_t)Bp * HP * 4, st_x = (size_t)Bp * GHP * 4;
    const unsigned lds_pub = (unsigned)(q * DROW + u * 2),
                   lds_rd = (unsigned)((j >> 2) * DROW + q * 16 + (SP ? ((j >> 1) & 1) * PLANEB : PK ? (j & 1) * PLANEB : 0));
    const int spidx = (j & 1) ? (int)0xDDDDDDDDu : (int)0x88888888u;      // SP: kept positions of every group of four, see rec_fwd_x6p

    const bool first = a.t_hi >= T, last = a.t_lo <= 0;          // first / last launch of the chunked chain
    float dh = 0.f, dc = 0.f;
    if (first) {
        if (a.n_dh_slabs) {                                       // dh_last arrives as unreduced split-K slabs of the dh GEMM
            const float* p = a.dh_slabs + (size_t)row * HP + u;
            const size_t st = (size_t)Bp * HP;
            int z = 0;
            for (; z + 8 <= a.n_dh_slabs; z += 8) {               // eight loads in flight (one after the other: 6 us)
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(z + k) * st];
#pragma unroll
                for (int k = 0; k < 8; ++k) dh += v[k];
            }
            for (; z < a.n_dh_slabs; ++z) dh += p[(size_t)z * st];
        } else if (a.dh_last) dh = a.dh_last[(size_t)row * HP + u];
    }
    else {
        dh = a.state[(size_t)row * HP + u];
        if (CELL == CELL_LSTM) dc = a.state[(size_t)Bp * HP + (size_t)row * HP + u];
    }
    // LSTM: the row-major saved array is cs (slot t = c_{t-1}; h_{t-1} is not needed by the gate math), and "hprev" / "hnew"
    // below carry c_{t-1} / c_t
    const char* const sbase = CELL == CELL_LSTM ? (const char*)a.cs : (const char*)a.hs;
    float pi = 0.f, pf = 0.f, po = 0.f, sdp[3] = {0.f, 0.f, 0.f};
    if (CELL == CELL_LSTM) { pi = a.peep[u]; pf = a.peep[HP + u]; po = a.peep[2 * HP + u]; }
    float sdb[G];
#pragma unroll
    for (int g = 0; g < G; ++g) sdb[g] = 0.f;

    float sv[4] = {0.f, 0.f, 0.f, 0.f}, hprev = 0.f, hnew = 0.f, dhe = 0.f;
    // WT: the saved activations travel PD time steps ahead of their use through a ring in LDS (LDS-DMA, sbr_rec_p.h): beside
    // the consumers' traffic the one step of the register prefetch below was not enough (chain 204 us against 184 alone).
    // Per wave and stage: NL arrays x 256 bytes (lane i's dword at + 4 i).  Ordering by the in-order vector-memory counter: an
    // iteration issues [publish] < loads of step t - PD < stores of step t.
    constexpr int PD = 4, NL = CELL == CELL_VANILLA ? 1 : 5, NST = G + (CELL == CELL_GRU ? 1 : 0);
    constexpr int NLI = CELL == CELL_VANILLA ? 1 : 2;             // load INSTRUCTIONS of a step (16-byte pieces, below)
    constexpr int STG = NL * 256, RING_OFF = (2 * 3 * R * DROW + 64 + 255) & ~255;
    // BDEF: the NST stores of a step are issued inside its MFMA phase, i.e. BEHIND take_saved: the iteration's own stores are
    // not in flight yet when the ring is read
    constexpr bool BDEF = RING && PK && X6P_BWD_DEFER;
    constexpr int VMN = (BDEF ? 0 : NST) + (PD - 1) * (NLI + NST);   // younger than the loads of the step being taken
    const f32x4* zslot = (const f32x4*)(smem_p + RING_OFF + 8 * PD * STG);      // SP, X6P_SP_INIT: sixteen zero bytes behind the ring
    if (SP && (X6P_SP_INIT & 2) && threadIdx.x == 0) *(f32x4*)zslot = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned ring_wave = (unsigned)(size_t)(smem_p + RING_OFF) + (unsigned)wave * (PD * STG);
    const char* ring_lane = smem_p + RING_OFF + wave * (PD * STG) + lane * 4;
    // Two 16-byte-per-lane pieces per step instead of five dword ones (a piece costs its issue slot whatever it moves): the
    // four gate arrays are tile-blocked, so the wave's 4 rows x 16 units of each are 256 contiguous bytes -- lane l fetches
    // chunk l % 16 of array l / 16; hs is row-major -- lanes 0-15 fetch the four 64-byte row pieces.  Both land in the
    // ring as [array][row][unit], where lane (j, q) finds its values at 4 (16 q + j).  One scalar base per step (hs + o): the
    // gate arrays' distance from hs rides in the lanes' offsets (same arena, same bytes per step: sbr_rec_x6p_tail_ok).
    unsigned bo_ga = 0;
    const unsigned bo_hs4 = (unsigned)((blockIdx.x * R + ((lane >> 2) & 3)) * HP + wave * 16 + (lane & 3) * 4) * 4u;
    // X6P_G4: the gate values are one 16-byte element per (row, unit): lane (j, q) fetches ITS OWN element (it lands at + 16 lane and is
    // read back by one ds_read_b128); a step is 4 x the bytes of hs, so the piece has its own scalar base
    if (X6P_G4) bo_ga = bo_g;
    else if (CELL != CELL_VANILLA) {
        const unsigned b0 = (unsigned)sbr_blocked_index(0, blockIdx.x * R, wave * 16, Bp, HP) * 4u;     // the wave's block
        const int k = lane >> 4;
        const char* gk = k == 0 ? (const char*)a.g[0] : k == 1 ? (const char*)a.g[1] : k == 2 ? (const char*)a.g[2] : (const char*)a.g[3];
        bo_ga = b0 + (unsigned)(size_t)(gk - sbase) + (unsigned)(lane & 15) * 16u;
    }
    auto dma_saved = [&](size_t o, int slot) {                   // activations of the step at byte offset o -> ring slot
        const unsigned m = ring_wave + (unsigned)slot * STG;
        const char* base = sbase + o;
        lds_dma_x4(m, base, bo_hs4, 0xFFFFull);
        if (CELL != CELL_VANILLA) {
            if constexpr (X6P_G4) lds_dma_x4(m + 256, (const char*)a.g[0] + 4 * o, bo_ga, ~0ull);
            else lds_dma_x4(m + 256, base, bo_ga, ~0ull);
        }
    };
    // The ring is read one iteration BEFORE the values are used (at the bottom of the iteration in front, into the registers
    // the gate math has just released), so that the LDS latency runs under the MFMA phase: read at the top of the
    // iteration that needs them it was 115 exposed cycles per step (tools/tail_prof.py).
    auto take_saved = [&](int slot) {                            // (these loads were issued PD - 1 iterations ago)
        wait_vm<VMN>();
        const char* p = ring_lane + slot * STG;
        hprev = *(const float*)p;
        if (CELL != CELL_VANILLA) {
            if constexpr (X6P_G4) {
                const f32x4 v = *(const f32x4*)(p + 256 + lane * 12);      // (p is ring + 4 lane: the element sits at ring + 256 + 16 lane)
                sv[0] = v[0]; sv[1] = v[1]; sv[2] = v[2]; sv[3] = v[3];
            } else {
            sv[0] = *(const float*)(p + 256); sv[1] = *(const float*)(p + 512);
            sv[2] = *(const float*)(p + 768); sv[3] = *(const float*)(p + 1024);
            }
        }
    };
    auto load_saved = [&](size_t o) {                            // activations of the step at byte offset o = t * st_h
        hprev = ldf(sbase + o, bo_h);
        if (CELL != CELL_VANILLA) {
            if constexpr (X6P_G4) {
                const f32x4 v = *(const f32x4*)((const char*)a.g[0] + 4 * o + (size_t)bo_g);
                sv[0] = v[0]; sv[1] = v[1]; sv[2] = v[2]; sv[3] = v[3];
            } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) sv[k] = ldf((const char*)a.g[k] + o, bo_g);
            }
        }
        if (EXT) dhe = ldf((const char*)a.dh_ext + o, bo_h);
    };
    unsigned long long p_c0 = 0, p_r0 = 0, p_spin = 0, p_tok = 0, p_n = 0, p_m = 0, p_vm = 0;
    if (PROF) { p_c0 = clock64(); p_r0 = wall_clock64(); }
    unsigned long long wt_c0 = 0, wt_r0 = 0;
    if (WT) { wt_c0 = clock64(); wt_r0 = wall_clock64(); }
    __syncthreads();

    const int t_live = min(a.t_hi, tmax);                         // steps [t_live, t_hi) are masked for the whole tile
    for (int t = a.t_hi - 1; t >= max(t_live, a.t_lo); --t) {     // zero rows; dh_ext still accumulates
        if (EXT) dh += a.dh_ext[((size_t)t * Bp + row) * HP + u];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float* p = a.dxt + ((size_t)t * Bp + row) * GHP + g * HP + u;
            if (WT) st_wt(p, 0.f); else *p = 0.f;
        }
        if (CELL == CELL_GRU) {
            float* p = a.dhi + ((size_t)t * Bp + row) * HP + u;
            if (WT) st_wt(p, 0.f); else *p = 0.f;
        }
    }
    int* prog_slot = nullptr; int prog_next = 0; const int prog_tag = a.prog_epoch << 12;
    if constexpr (WT) {
        prog_slot = a.progress + blockIdx.x * 8 + wave;
        const int tl = max(t_live, a.t_lo);
        publish_progress(prog_slot, prog_tag | tl);               // every step >= tl is complete (zero rows above)
        prog_next = tl - a.prog_every;                            // publish again at or below this step
    }
    if (t_live > a.t_lo) {
        if constexpr (RING) {    // fill the ring: steps t_live - 1 .. t_live - PD (clamped), all landed before the loop starts
#pragma unroll
            for (int d = 0; d < PD; ++d) dma_saved((size_t)max(t_live - 1 - d, a.t_lo) * st_h, d);
            wait_vm<0>();
            take_saved(0);                                        // the first step's values
        } else load_saved((size_t)(t_live - 1) * st_h);
        if (CELL == CELL_VANILLA) hnew = a.hs[(size_t)t_live * Bp * HP + (size_t)row * HP + u];
        if (CELL == CELL_LSTM) hnew = a.cs[(size_t)t_live * Bp * HP + (size_t)row * HP + u];
    }
    // k-blocks in the order they are visited: columns of units 0-63 (all gates), then of units 64-127
    constexpr int NH = KB / 2;
    auto korder = [](int i) { return (i % NH) / 2 * KU + (i % NH) % 2 + (i / NH) * 2; };
    size_t off_h = (size_t)(t_live - 1) * st_h, off_x = (size_t)(t_live - 1) * st_x;   // of step t
    const unsigned lds_cnt0 = (unsigned)(size_t)cnt;
    const unsigned my_cnt = roleA ? lds_cnt0 : lds_cnt0 + 4;
    float dxi_st[G], dhc_st = 0.f; size_t offx_st = 0, offh_st = 0;      // BDEF: what the MFMA phase stores
#pragma unroll
    for (int g = 0; g < G; ++g) dxi_st[g] = 0.f;
    auto steps = [&](auto role_tag) {                             // one loop per role, see rec_fwd_x6p
    constexpr bool RA = decltype(role_tag)::value;
    int n = 0;                                                    // steps done
    int slot = 0;                                                 // WT: ring slot of step t = n % PD
    for (int t = t_live - 1; t >= a.t_lo; --t, ++n) {
        unsigned long long q_top = 0;
        if (PROF) q_top = clock64();
        // ---- N: gate math of step t (needs dh complete), publish dhi, stores, loads for step t-1
        if (EXT) dh += dhe;
        char* lds = dbuf + (n & 1) * BUFB;
        float dxi[G], dhi[G], dp[3] = {0.f, 0.f, 0.f};
        if (CELL == CELL_LSTM) {
            cell_backward<CELL, true>(t < mylen, clip, dh, dc, sv, 0.f, hprev, hnew, 0.f, pi, pf, po, dxi, dhi, dp, false);
            sdp[0] += dp[0]; sdp[1] += dp[1]; sdp[2] += dp[2];
        } else
        cell_backward<CELL, true>(t < mylen, clip, dh, dc, sv, hprev, 0.f, 0.f, hnew, 0.f, 0.f, 0.f, dxi, dhi, dp, a.relu != 0);
        if constexpr (WT) {
            // Progress: everything but the youngest NL + NST operations (the loads and stores the previous iteration issued) is
            // waited for -- two iterations old by now, normally long complete -- so step t + 2 is complete and written through.
            if (n >= 1 && t + 2 <= prog_next) {                   // uniform
                wait_vm<NLI + NST>();
                publish_word_after(prog_slot, prog_tag | (t + 2), dxi[0]);
                prog_next = t + 2 - a.prog_every;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) sdb[g] += dxi[g];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            char* base = lds + lds_pub + g * HP * 2;
            if constexpr (F16) {
                float d = dhi[g] * F16_DSCALE;            // |dhi| <= clip <= 100: below fp16's 65504
                asm volatile("" : "+v"(d));               // pinned for split2_f16
                _Float16 d1, d2;
                split2_f16(d, d1, d2);
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
        if (X6P_SYNC) __syncthreads(); else
        lds_inc(my_cnt, one);
        if (CELL != CELL_GRU) hnew = hprev;
        if constexpr (RING) {                                     // loads first: see the progress note above
            __builtin_amdgcn_sched_barrier(0);
            dma_saved(t - PD >= a.t_lo ? off_h - (size_t)PD * st_h : (size_t)a.t_lo * st_h, slot);   // step t - PD into the slot just read
            slot = slot + 1 == PD ? 0 : slot + 1;
            if constexpr (BDEF) {
#pragma unroll
                for (int g = 0; g < G; ++g) dxi_st[g] = dxi[g];
                dhc_st = dhi[G - 1]; offx_st = off_x; offh_st = off_h;
            } else {
            const char* dx_t = (const char*)a.dxt + off_x;
            st_si<0, WT>(dx_t, bo_x, dxi[0]);
            if (G > 1) st_si<HP * 4, WT>(dx_t, bo_x, dxi[G > 1 ? 1 : 0]);
            if (G > 2) st_si<2 * HP * 4, WT>(dx_t, bo_x, dxi[G > 2 ? 2 : 0]);
            if (G > 3) st_si<3 * HP * 4, WT>(dx_t, bo_x, dxi[G > 3 ? 3 : 0]);
            if (CELL == CELL_GRU) st_si<0, WT>((const char*)a.dhi + off_h, bo_h, dhi[G - 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            { unsigned long long q0 = 0; if (PROF) q0 = clock64();
              take_saved(slot);                                   // step t - 1 (slot has moved on to it)
              if (PROF) p_vm += clock64() - q0; }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            const char* dx_t = (const char*)a.dxt + off_x;
            st_si<0>(dx_t, bo_x, dxi[0]);
            if (G > 1) st_si<HP * 4>(dx_t, bo_x, dxi[G > 1 ? 1 : 0]);
            if (G > 2) st_si<2 * HP * 4>(dx_t, bo_x, dxi[G > 2 ? 2 : 0]);
            if (G > 3) st_si<3 * HP * 4>(dx_t, bo_x, dxi[G > 3 ? 3 : 0]);
            if (CELL == CELL_GRU) st_si<0>((const char*)a.dhi + off_h, bo_h, dhi[G - 1]);
            __builtin_amdgcn_sched_barrier(0);
            load_saved(t > a.t_lo ? off_h - st_h : off_h);            // step t-1: unconditional, clamped
            __builtin_amdgcn_sched_barrier(0);
        }
        off_h -= st_h; off_x -= st_x;
        unsigned long long q_n = 0;
        if (PROF) { q_n = clock64(); p_n += q_n - q_top; }
        // ---- operands: a ring of NS k-block slots, fetched LA k-blocks ahead of their MFMAs; the pipe gate
        constexpr int LA = X6P_BWD_LA < NH ? X6P_BWD_LA : NH - 1, NS = LA < X6P_BWD_NS ? X6P_BWD_NS : LA + 1;   // (a Vanilla step has four k-blocks: the lookahead stays inside a half)
        static_assert(LA >= 1 && NS > LA, "operand ring");
        const char* db = lds + lds_rd;
        OPV dpl[NS][NP];
        int fl[2];
        auto load_kb = [&](int i) {
            const int kb = korder(i), s = i % NS;
            dpl[s][0] = *(const OPV*)(db + kb * 64);
            dpl[s][1] = PK ? dpl[s][0] : *(const OPV*)(db + kb * 64 + PLANEB);
            if constexpr (NP == 3) dpl[s][2] = *(const OPV*)(db + kb * 64 + 2 * PLANEB);
        };
        int tokv = 0;                                             // RA: the pipe gate's token, read with the first flag (see rec_fwd_x6p)
        auto load_flag = [&](int half) {
            fl[half] = __hip_atomic_load(cnt + half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (X6P_TOK_EARLY && RA && half == 0 && X6P_GATE(a)) tokv = __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
        };
        auto use = [&](int i) { asm volatile("" :: "v"(dpl[i % NS][0]), "v"(dpl[i % NS][1]), "v"(dpl[i % NS][NP - 1])); };
        // the half's producers have published this step; else re-read the counter and the k-blocks [i0, i1) fetched on spec
        auto ensure_half = [&](int half, int i0, int i1) {
            if (!X6P_SYNC && __builtin_amdgcn_readfirstlane(fl[half]) < 4 * (n + 1)) {
                unsigned long long w0 = 0;
                if (PROF) w0 = clock64();
                int spins = 0;
#pragma clang loop unroll(disable)
                do {
                    asm volatile("" ::: "memory");
                    load_flag(half);
#pragma unroll
                    for (int i = i0; i < i1; ++i) load_kb(i);
                    if (++spins > spin_limit) { atomicOr(a.fault, 2); spin_limit = 64; break; }
                } while (__builtin_amdgcn_readfirstlane(fl[half]) < 4 * (n + 1));
                if (PROF) p_spin += clock64() - w0;
            }
            use(i0);
        };
        f32x4 accz = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (SP && (X6P_SP_INIT & 2)) { asm volatile("" ::: "memory"); accz = *zslot; }
        load_flag(0);
#pragma unroll
        for (int i = 0; i < LA; ++i) load_kb(i);
        if (!RA) load_flag(1);
        ensure_half(0, 0, LA);
        __builtin_amdgcn_s_setprio(0);
        if (RA && X6P_GATE(a)) {                               // the matrix-pipe gate, see rec_fwd_x6p
            unsigned long long w0 = 0;
            if (PROF) w0 = clock64();
            int v = X6P_TOK_EARLY ? tokv : __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), spins = 0;
#pragma clang loop unroll(disable)
            while (__builtin_amdgcn_readfirstlane(v) < n) {
                v = __hip_atomic_load(tok + (wave & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (++spins > spin_limit) { atomicOr(a.fault, 4); spin_limit = 64; break; }
            }
            if (PROF) p_tok += clock64() - w0;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- M
        const f32x4 z4 = f32x4{0, 0, 0, 0};
        f32x4 acc[3] = {z4, z4, z4};
        if constexpr (SP && (X6P_SP_INIT & 2)) acc[0] = accz;     // (in-place accumulation: zeros by one LDS read, issued in the N phase)
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int s = i % NS, kb = korder(i);
            if (i + LA < KB) {
                if (i + LA == NH) load_flag(1);                   // (waves 4-7 looked before the gate; harmless to look again)
                load_kb(i + LA);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SP) {       // acc[0]: {d1 w1, d1 w2, d2 w1, d2 w2}, one dependent chain (full issue rate: smfmac_probe)
                if (!RA && i == KB - 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }   // one instruction early
                acc[0] = smfmac16(dpl[s][0], WS[kb], acc[0], spidx);
                if (BDEF && i == 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    const char* dx_t = (const char*)a.dxt + offx_st;
                    st_si<0, WT>(dx_t, bo_x, dxi_st[0]);
                    if (G > 1) st_si<HP * 4, WT>(dx_t, bo_x, dxi_st[G > 1 ? 1 : 0]);
                    if (G > 2) st_si<2 * HP * 4, WT>(dx_t, bo_x, dxi_st[G > 2 ? 2 : 0]);
                    if (G > 3) st_si<3 * HP * 4, WT>(dx_t, bo_x, dxi_st[G > 3 ? 3 : 0]);
                    if (CELL == CELL_GRU) st_si<0, WT>((const char*)a.dhi + offh_st, bo_h, dhc_st);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (PK) {       // acc[0][0]: d1 w1, acc[0][1]: d2 w1, acc[1][0]: d1 w2 (acc[1][1] = d2 w2 is dropped)
                acc[1] = mfma16(dpl[s][0], W2[kb], acc[1]);
                if (!RA && i == KB - 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }   // one MFMA early
                acc[0] = mfma16(dpl[s][0], W1[kb], acc[0]);
                if (BDEF && i == 1) {                               // this step's dxt / dhi, from where the issue slots are free
                    __builtin_amdgcn_sched_barrier(0);
                    const char* dx_t = (const char*)a.dxt + offx_st;
                    st_si<0, WT>(dx_t, bo_x, dxi_st[0]);
                    if (G > 1) st_si<HP * 4, WT>(dx_t, bo_x, dxi_st[G > 1 ? 1 : 0]);
                    if (G > 2) st_si<2 * HP * 4, WT>(dx_t, bo_x, dxi_st[G > 2 ? 2 : 0]);
                    if (G > 3) st_si<3 * HP * 4, WT>(dx_t, bo_x, dxi_st[G > 3 ? 3 : 0]);
                    if (CELL == CELL_GRU) st_si<0, WT>((const char*)a.dhi + offh_st, bo_h, dhc_st);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if constexpr (F16) {      // acc[0]: d1 w1;  acc[1], acc[2]: the low-order products d2 w1, d1 w2 (/ 2048 at the end)
                if (!RA && i == KB - 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }   // three MFMAs early
                acc[1] = mfma16(dpl[s][1], W1[kb], acc[1]);
                acc[2] = mfma16(dpl[s][0], W2[kb], acc[2]);
                acc[0] = mfma16(dpl[s][0], W1[kb], acc[0]);
            } else {
            acc[0] = mfma16(dpl[s][0], W3[kb], acc[0]);
            acc[1] = mfma16(dpl[s][NP - 1], W1[kb], acc[1]);
            acc[2] = mfma16(dpl[s][1], W2[kb], acc[2]);
            if (!RA && i == KB - 1) { __builtin_amdgcn_sched_barrier(0); if (!X6P_SYNC && X6P_GATE(a)) lds_inc(lds_tok, one); __builtin_amdgcn_sched_barrier(0); }   // three MFMAs early, see rec_fwd_x6p
            acc[0] = mfma16(dpl[s][0], W2[kb], acc[0]);
            acc[1] = mfma16(dpl[s][1], W1[kb], acc[1]);
            acc[2] = mfma16(dpl[s][0], W1[kb], acc[2]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 == NH) ensure_half(1, NH, NH + LA < KB ? NH + LA : KB);
            else if (i + 1 < KB) use(i + 1);
        }
        asm volatile("s_nop 15");                                 // MFMA D -> VALU read hazard
        __builtin_amdgcn_s_setprio(3);
        if (PROF) p_m += clock64() - q_n;
        if constexpr (SP) dh += fmaf(acc[0][1] + acc[0][2], 1.0f / F16_LO, acc[0][0]) * (1.0f / F16_DSCALE);
        else if constexpr (PK) dh += fmaf(acc[0][1] + acc[1][0], 1.0f / F16_LO, acc[0][0]) * (1.0f / F16_DSCALE);
        else if constexpr (F16) dh += fmaf(acc[1][0] + acc[2][0], 1.0f / F16_LO, acc[0][0]) * (1.0f / F16_DSCALE);
        else dh += acc[0][0] + acc[1][0] + acc[2][0];
    }
    };
    if (roleA && !X6P_SYNC && X6P_ROLES) steps(std::true_type{}); else steps(std::false_type{});
    if constexpr (WT) publish_progress(prog_slot, prog_tag | a.t_lo);
    if constexpr (WT) {      // shader cycles and 100 MHz wall-clock ticks of this launch (block 0, wave 0): the chain's effective clock
        if (blockIdx.x == 0 && wave == 0 && lane == 0) {
            unsigned long long* c = (unsigned long long*)(a.progress + gridDim.x * 8 + 128);
            c[0] = clock64() - wt_c0; c[1] = wall_clock64() - wt_r0; c[2] = wt_r0; c[3] = wall_clock64();
        }
    }
    else if constexpr (RING) wait_vm<0>();                       // (the last iterations' clamped loads)
    if (PROF && lane == 0 && blockIdx.x < (unsigned)(a.Bp >> 4)) {
        unsigned long long* o = a.prof + ((size_t)blockIdx.x * 16 + wave) * 8;
        const unsigned long long tot = clock64() - p_c0;
        o[0] = tot; o[1] = wall_clock64() - p_r0; o[2] = tot - p_spin - p_tok; o[3] = p_spin; o[4] = p_tok;
        o[5] = p_n; o[6] = p_m; o[7] = p_vm;      // loop top -> loads / stores issued (incl. o[7]: the wait for the ring) | operands + gate + MFMAs
    }

    if (!last) {                                                  // hand dh (dc) to the next chunk launch
        a.state[(size_t)row * HP + u] = dh;
        if (CELL == CELL_LSTM) a.state[(size_t)Bp * HP + (size_t)row * HP + u] = dc;
    }
    float* part = a.part + ((size_t)a.chunk * gridDim.x + blockIdx.x) * (GHP + 5 * HP);
    float v[G + 5];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = sdb[g];
    v[G] = sdp[0]; v[G + 1] = sdp[1]; v[G + 2] = sdp[2];          // peephole gradients (LSTM; zero otherwise)
    v[G + 3] = CELL == CELL_LSTM && last ? dc : 0.f;
    v[G + 4] = last ? dh : 0.f;                                   // init-state gradient comes from the last chunk only
#pragma unroll
    for (int k = 0; k < G + 5; ++k) {
        float sum = v[k];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);                               // over the tile's 4 rows (q)
        v[k] = sum;
    }
    if (q == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) part[g * HP + u] = v[g];
#pragma unroll
        for (int k = 0; k < 5; ++k) part[GHP + k * HP + u] = v[G + k];
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
static size_t fwd_lds_bytes() { return 2 * 3 * R * (size_t)(HP * 2 + 32) + 64; }

static bool x6p_f16_fwd(const RecArgs& a) {                        // forward products as fp16 x3 (see split2_f16); a rectified
    const char* fe = getenv("SBR_X6_F16");                         // state is unbounded, the fp16 split needs |h| < 65504
    return (fe ? atoi(fe) != 0 : true) && !a.relu;                 // (read per launch: the tests flip it)
}
static bool x6p_f16_bwd(const RecArgs& a) {     // the operand that carries gradients is bounded by the reference's own gradient clip
    const char* fe = getenv("SBR_X6_F16_BWD");
    return (fe ? atoi(fe) != 0 : true) && a.clip > 0.0f && a.clip <= 100.0f;
}

int sbr_rec_x6p_f16_terms() { return X6P_PACK ? (X6P_SPARSE ? 1 : 2) : 3; }

bool sbr_rec_x6p_ok(const RecArgs& a) {
    if (!a.x6_pipe || a.f32_mfma || a.Hp != HP || a.rpt != R || !a.x6_split) return false;
    // four gates: W_hid fits the register file as two fp16 planes only, so an LSTM runs here while BOTH directions take their
    // fp16 forms (one answer for the forward and the backward launch of a step: they share the saved activations' layout)
    if (a.G > 3 && !(x6p_f16_fwd(a) && x6p_f16_bwd(a))) return false;
    if ((size_t)a.Bp * a.G * HP * 4 >= ((size_t)1 << 32)) return false;          // 32-bit per-lane byte offsets
    // (NOT a function of a.gX: the forward and the backward launch of a step must get the same answer -- they share the saved
    // gates' layout (X6P_G4).  Whether the gather can be fused is sbr_rec_fwd_can_fuse_gather's decision, made before gX is set;
    // launch_fwd_p refuses a fused launch outside those bounds instead of silently taking another kernel family.)
    if (X6P_G4 && a.cell != SBR_CELL_VANILLA && a.g[0])                            // one region [T][Bp][HP][4] under the four arrays
        for (int k = 1; k < 4; ++k) if (a.g[k] != a.g[0] + (size_t)k * a.T * a.Bp * HP) return false;
    if (X6P_G4 && (size_t)a.Bp * HP * 16 >= ((size_t)1 << 32)) return false;
    return true;
}

// the fused gather of rec_fwd_x6p: 32-bit per-lane byte offsets into W_in, and the tile's row-offset table (R x T) in LDS
bool sbr_rec_x6p_fuse_ok(const RecArgs& a) {
    return (size_t)a.n_in * a.G * HP * 4 < ((size_t)1 << 32) && a.T <= SBR_X6P_FUSE_MAX_T;
}

template <int CELL>
static hipError_t launch_fwd_p(hipStream_t s, const RecArgs& a) {
    size_t lds = fwd_lds_bytes();
    if (a.gX) {      // fused gather: + the row-offset table (R x T) and the ring of rows (8 waves x XPD = 4 stages x G x 256 bytes)
        lds = ((lds + 255) & ~(size_t)255) + (size_t)R * a.T * 4;
        lds = ((lds + 255) & ~(size_t)255) + (size_t)8 * 4 * Gates<CELL>::G * 256;
    }
    if (X6P_PACK && X6P_SPARSE && (X6P_SP_INIT & 1)) lds = ((lds + 255) & ~(size_t)255) + (size_t)Gates<CELL>::G * HP * 16;   // + the accumulators' initial values
    if (a.fence_kb > 0 && a.Bp / R <= 192 && (size_t)a.fence_kb * 1024 > lds && a.fence_kb <= 160) lds = (size_t)a.fence_kb * 1024;   // see launch_bwd_p
    const int nb = a.Bp / R;
#define X6P_LAUNCH(KERNEL) do { \
        SBR_DYN_LDS(KERNEL, lds); \
        KERNEL<<<nb, 512, lds, s>>>(a); } while (0)
    const bool fuse = a.gX != nullptr;
    if (fuse && !sbr_rec_x6p_fuse_ok(a)) return hipErrorInvalidValue;          // (sbr_rec_fwd_can_fuse_gather says when)
    const bool f16 = x6p_f16_fwd(a);
    if constexpr (CELL == CELL_LSTM) {
        if (!f16) return hipErrorInvalidValue;                     // (sbr_rec_x6p_ok says when)
        if (a.prof) { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, true, true>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, true, true>)); }
        else { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, false, true>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, false, true>)); }
    } else
    if (a.prof && f16) { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, true, true>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, true, true>)); }
    else if (a.prof) { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, true, false>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, true, false>)); }
    else if (f16) { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, false, true>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, false, true>)); }
    else { if (fuse) X6P_LAUNCH((rec_fwd_x6p<CELL, true, false, false>)); else X6P_LAUNCH((rec_fwd_x6p<CELL, false, false, false>)); }
#undef X6P_LAUNCH
    return hipGetLastError();
}

template <int CELL>
static hipError_t launch_bwd_p(hipStream_t s, const RecArgs& a) {
    constexpr int G = Gates<CELL>::G;
    size_t lds = 2 * 3 * R * (size_t)(G * HP * 2 + 32) + 64;
    lds = ((lds + 255) & ~(size_t)255) + 8 * 4 * (size_t)(G == 1 ? 1 : 5) * 256;             // + the prefetch ring (PD = 4 stages per wave)
    lds += 256;                                                                               // + the zero slot of the sparse form
    // Overlapped tail: the chain's workgroup claims most of its CU's LDS, so that the consumers that run beside it -- the polling
    // weight-gradient GEMM (40 KB of LDS per workgroup, MFMAs on the same SIMDs) and the scatter-add (which asks for LDS it does not
    // use, for this purpose) -- are placed on the other 192 CUs instead of sharing the chain's matrix pipes, issue slots and L1
    // path.  SBR_TAIL_FENCE_KB=0: no fence (round 2).
    if (a.fence_kb > 0 && a.Bp / R <= 192) {      // (one workgroup per CU must still leave CUs to the consumers: up to B = 768)
        const size_t fence = (size_t)a.fence_kb * 1024;
        if (fence > lds && fence <= 160 * 1024) lds = fence;
    }
    const int nb = a.Bp / R;
#define X6P_LAUNCH(KERNEL) do { \
        SBR_DYN_LDS(KERNEL, lds); \
        KERNEL<<<nb, 512, lds, s>>>(a); } while (0)
    const bool ext = a.dh_ext != nullptr;
    const bool f16 = x6p_f16_bwd(a);                               // fp16 x3 products for the BPTT chain
    if (a.progress && (ext || !f16)) return hipErrorInvalidValue;               // (sbr_rec_x6p_tail_ok says when)
    if (CELL == CELL_LSTM && !f16) return hipErrorInvalidValue;                 // (sbr_rec_x6p_ok)
    if (a.prof && f16 && !ext) {      // in-kernel counters for the fp16x3 forms too (tools/tail_prof.py)
        if (a.progress) X6P_LAUNCH((rec_bwd_x6p<CELL, false, true, true, 1>)); else X6P_LAUNCH((rec_bwd_x6p<CELL, false, true, true, 0>));
        return hipGetLastError();
    }
    // (the LDS ring WITHOUT consumers beside the chain -- WT = 2 -- measured 194 us against 189 with the register prefetch: not launched)
    if (a.progress) { X6P_LAUNCH((rec_bwd_x6p<CELL, false, false, true, 1>)); }
    else if constexpr (CELL == CELL_LSTM) { if (ext) X6P_LAUNCH((rec_bwd_x6p<CELL, true, false, true>)); else X6P_LAUNCH((rec_bwd_x6p<CELL, false, false, true>)); }
    else if (a.prof) { if (ext) X6P_LAUNCH((rec_bwd_x6p<CELL, true, true, false>)); else X6P_LAUNCH((rec_bwd_x6p<CELL, false, true, false>)); }
    else if (f16) { if (ext) X6P_LAUNCH((rec_bwd_x6p<CELL, true, false, true>)); else X6P_LAUNCH((rec_bwd_x6p<CELL, false, false, true>)); }
    else { if (ext) X6P_LAUNCH((rec_bwd_x6p<CELL, true, false, false>)); else X6P_LAUNCH((rec_bwd_x6p<CELL, false, false, false>)); }
#undef X6P_LAUNCH
    return hipGetLastError();
}

// the write-through / progress form of the backward kernel exists for the fp16x3 products of a top (single) layer
bool sbr_rec_x6p_tail_ok(const RecArgs& a) {
    const bool f16 = x6p_f16_bwd(a);
    const char* sbase = a.cell == SBR_CELL_LSTM ? (const char*)a.cs : (const char*)a.hs;
    for (int k = 0; k < 4 && a.cell != SBR_CELL_VANILLA; ++k) {      // one scalar base serves hs (cs) and the gate arrays (LDS-DMA loads)
        const ptrdiff_t d = (const char*)a.g[k] - sbase;
        if (d < 0 || d >= ((ptrdiff_t)1 << 31)) return false;
    }
    return sbr_rec_x6p_ok(a) && f16 && !a.dh_ext && a.T < 4096 &&
           (size_t)a.T * a.Bp * a.G * HP * 4 < ((size_t)1 << 31);                    // buffer stores: 31-bit scalar offsets
}

hipError_t launch_rec_backward_x6p(hipStream_t s, const RecArgs& a) {
    return a.cell == SBR_CELL_GRU ? launch_bwd_p<CELL_GRU>(s, a) : a.cell == SBR_CELL_LSTM ? launch_bwd_p<CELL_LSTM>(s, a)
                                                                 : launch_bwd_p<CELL_VANILLA>(s, a);
}

hipError_t launch_rec_forward_x6p(hipStream_t s, const RecArgs& a) {
    return a.cell == SBR_CELL_GRU ? launch_fwd_p<CELL_GRU>(s, a) : a.cell == SBR_CELL_LSTM ? launch_fwd_p<CELL_LSTM>(s, a)
                                                                 : launch_fwd_p<CELL_VANILLA>(s, a);
}
