// The full-softmax head of one training step in ONE launch (round 5): logits = h_last . W_out + b, softmax +
// categorical cross-entropy and its gradient, dh = dlogits . W_out^T  (rnn_one_hot.py:65-71: DenseLayer(softmax) +
// categorical_crossentropy / target popularity, mean over the batch).
//
// Before: three dependent launches on the main stream between the two recurrent chains -- logits GEMM (9 us at C2), softmax +
// CCE (6 us), dh GEMM (10 us) -- with their launch gaps: 34 us of a 342 us step in which nothing else could run
// (profiles/round4_z_c2_timeline.txt).  The work itself is 0.24 GFLOP per GEMM and 3.8 MB of logits: latency, not throughput.
//
// Here a workgroup owns 16 batch rows x one CHUNK of the catalogue (CW columns, a multiple of 16), and the grid -- row blocks x
// column chunks, at most one workgroup per CU: at most 256 -- is co-resident, so the softmax's row statistics cross the chunks
// inside the kernel:
//   0. the chunk's rows of W_out^T [CW][HP] (item-major: a row is contiguous) go to LDS once, f32, row stride HP + 4 floats;
//      every wave keeps its 16 rows of h_last in registers as the MFMA's B operand.
//   1. logits tile by tile (a wave takes the 16-column tiles wave, wave + 4, ...): D[item][row] = sum_k W[item][k] h[row][k] on
//      v_mfma_f32_16x16x4_f32 (exact f32 products -- no operand split, nothing to bound), one ds_read_b128 per four
//      instructions (k slot q of instruction m is k0 + 4 q + m).  A lane ends with four consecutive items of one batch row, the
//      tile stays in registers.
//   2. row max / sum of exponentials over the chunk (lanes -> waves through LDS), published as one 16-byte piece per (row block,
//      chunk, row) whose dwords validate themselves against this launch's epoch (value, value ^ epoch: no ordering needed, a
//      torn piece is simply not valid yet); every workgroup of the row block polls the CC pieces of its rows (agent-scope loads)
//      and combines them in chunk order -- every workgroup the same bits.  A chunk whose workgroup has not published within
//      60 us is not waited for: its statistics are recomputed by whoever misses them (its W rows through the same LDS image,
//      the same code), so every workgroup can always finish on its own -- two processes on one GPU can interleave two such
//      grids so that neither is ever co-resident.
//   3. dlogits = (softmax - onehot) / (pop B) from the registers, stored once (the output layer's gradient kernels on the side
//      stream read them); the row's cost by the lane that holds its target column.
//   4. dh[row][k] = sum_items dlogits[row][item] W[item][k] with the SAME LDS image (item = reduction index: one ds_read_b32 per
//      instruction, conflict-free at stride HP + 4) and the dlogits registers as the B operand exactly as phase 3 left them
//      (k slot q of instruction e of tile t is item 16 t + 4 q + e); the four waves' partial sums meet in LDS and leave as ONE
//      split-K slab [Bp][HP] per chunk -- what rec_bwd_x6p's prologue already adds up (RecArgs.dh_slabs), else
//      gemm_splitk_reduce follows.
// Served: CCE, one direction, full batches (rows == Bp), HP in {32, 64, 128}, and a catalogue whose chunk fits LDS
// (CW (HP + 4) 4 bytes + 3 KB <= 160 KB with CC = min(16, 256 / (Bp / 16)) chunks: N <= 4 864 at B = 256, HP = 128: C1, C2).  Everything
// else keeps the three launches.  SBR_HEAD_FUSE=0 switches it off.
#include "sbr_common.h"
#include "sbr_rec_p.h"
#include <cstdlib>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct HeadArgs {
    const float* h;        // [Bp][HP] h_last
    const float* W;        // [N][HP]  W_out^T
    const float* b;        // [N]
    const int* tgt;        // [Bp]
    const float* pop;      // [Bp]
    float* dlog;           // [Bp][Nl]
    float* rowcost;        // [Bp]
    unsigned long long wait_ticks;   // HEAD_WAIT_TICKS, or SBR_HEAD_WAIT_TICKS (tests: 0 = every foreign chunk is recomputed)
    float* slabs;          // [CC][Bp][HP]
    unsigned* stats;       // [RB][CC][16][4]
    int* fault;
    int N, Nl, CW, CC, RB, Bp;
    float inv_Bg;
    unsigned epoch;
    unsigned long long* prof;   // SBR_FLAG_PROFILE_REC: [workgroup][8] stamps of the 100 MHz clock at the phase boundaries (tools/head_prof.py), else NULL
};

#define HEAD_NT 5          // tiles per wave at most (CW <= 320)
#define HEAD_LOG2E 1.4426950408889634f

// NTW tiles of one wave (rows 64 i apart in the LDS image) through all of K, NTW independent accumulator chains
template <int HP, int NTW>
__device__ __forceinline__ void head_logits(const float* __restrict__ wr, const f32x4 (&hb)[HP / 16], f32x4 (&acc)[HEAD_NT]) {
    constexpr int LDW = HP + 4, KG = HP / 16;
#pragma unroll
    for (int g = 0; g < KG; ++g) {
        f32x4 wv[NTW];
#pragma unroll
        for (int i = 0; i < NTW; ++i) wv[i] = *(const f32x4*)(wr + (size_t)64 * i * LDW + 16 * g);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int i = 0; i < NTW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[i][m], hb[g][m], acc[i], 0, 0, 0);
    }
}

#define HEAD_WAIT_TICKS 6000ull      // 60 us of the 100 MHz clock: how long a chunk's statistics are waited for before they are recomputed

// rows [n_lo, n_lo + CW) of W_out^T -> the LDS image (rows beyond the catalogue: zeros).  Rounds of 16 pieces per thread in flight
// (C2's chunk of 240 rows x 128 floats is 30 pieces per thread; in rounds of 6 the fill was five dependent round trips to L2)
template <int HP>
__device__ __forceinline__ void head_fill(const HeadArgs& a, float* __restrict__ Wl, int n_lo, int tid) {
    constexpr int LDW = HP + 4, P = HP / 4, U = 16;                    // P: 16-byte pieces per row
    const int total = a.CW * P;
    for (int i0 = tid; i0 < total; i0 += 256 * U) {
        f32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 256 * u, r = i / P, c4 = i - r * P;
            v[u] = (i < total && n_lo + r < a.N) ? *(const f32x4*)(a.W + (size_t)(n_lo + r) * HP + 4 * c4) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + 256 * u, r = i / P, c4 = i - r * P;
            if (i < total) *(f32x4*)(Wl + r * LDW + 4 * c4) = v[u];
        }
    }
}

// this wave's tiles of the chunk whose rows are in the LDS image: logits (+ bias; -inf beyond the catalogue) into lg, the wave's
// row maximum and sum of exponentials into red_w[wave][row]
template <int HP>
__device__ __forceinline__ void head_wave_stats(const HeadArgs& a, const float* __restrict__ Wl, const f32x4 (&hb)[HP / 16], int n_lo, int ntiles,
                                                int wave, int j, int q, f32x4 (&lg)[HEAD_NT], float* __restrict__ red_w) {
    constexpr int LDW = HP + 4;
    float mx = -INFINITY;
    {
        f32x4 acc[HEAD_NT];
#pragma unroll
        for (int i = 0; i < HEAD_NT; ++i) acc[i] = f32x4{0, 0, 0, 0};
        const float* wr = Wl + (16 * wave + j) * LDW + 4 * q;          // tile wave + 4 i: + 64 i rows
        const int ntw = ntiles > wave ? (ntiles - wave + 3) >> 2 : 0;   // tiles of this wave (wave-uniform)
        // the tiles advance TOGETHER through k, one accumulator chain each; straight-line code per tile count (guards inside the
        // loops made 160 basic blocks of it: 4.7 -> 24.7 us, profiles/round5_variants.txt call d)
        switch (ntw) {
            case 5: head_logits<HP, 5>(wr, hb, acc); break;
            case 4: head_logits<HP, 4>(wr, hb, acc); break;
            case 3: head_logits<HP, 3>(wr, hb, acc); break;
            case 2: head_logits<HP, 2>(wr, hb, acc); break;
            case 1: head_logits<HP, 1>(wr, hb, acc); break;
            default: break;
        }
        asm volatile("s_nop 15");                                      // MFMA D -> VALU read (see sbr_gemm.hip)
#pragma unroll
        for (int i = 0; i < HEAD_NT; ++i) {
            const int t = wave + 4 * i;
            lg[i] = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (t < ntiles) {
                const int c0 = n_lo + 16 * t + 4 * q;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (c0 + r < a.N) ? acc[i][r] + a.b[c0 + r] : -INFINITY;
                    lg[i][r] = v; mx = fmaxf(mx, v);
                }
            }
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16)); mx = fmaxf(mx, __shfl_xor(mx, 32));      // the lanes of a row: q = 0 .. 3
    float se = 0.0f;
    if (mx > -INFINITY) {
#pragma unroll
        for (int i = 0; i < HEAD_NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) se += __builtin_amdgcn_exp2f((lg[i][r] - mx) * HEAD_LOG2E);      // (-inf -> 0)
    }
    se += __shfl_xor(se, 16); se += __shfl_xor(se, 32);
    if (q == 0) { red_w[(wave * 16 + j) * 2] = mx; red_w[(wave * 16 + j) * 2 + 1] = se; }
}
// the four waves' statistics of row r -> the chunk's (fixed order)
__device__ __forceinline__ void head_chunk_combine(const float* __restrict__ red_w, int r, float& m, float& s) {
    m = -INFINITY; s = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) m = fmaxf(m, red_w[(w * 16 + r) * 2]);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float mw = red_w[(w * 16 + r) * 2];
        if (mw > -INFINITY) s += red_w[(w * 16 + r) * 2 + 1] * __builtin_amdgcn_exp2f((mw - m) * HEAD_LOG2E);
    }
}

template <int HP>
__global__ void __launch_bounds__(256) head_cce_kernel(HeadArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int LDW = HP + 4, KG = HP / 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    // workgroups of one chunk share its W rows: keep them on one XCD (workgroup ids go round-robin over the 8 XCDs)
    int rb, cc;
    if ((a.CC & 7) == 0) { const int x = blockIdx.x & 7, li = blockIdx.x >> 3; cc = x * (a.CC >> 3) + li / a.RB; rb = li % a.RB; }
    else { cc = blockIdx.x / a.RB; rb = blockIdx.x % a.RB; }
    const int n_lo = cc * a.CW, ntiles = a.CW >> 4;
#define HEAD_STAMP(I) do { if (a.prof && tid == 0) a.prof[(size_t)blockIdx.x * 8 + (I)] = wall_clock64(); } while (0)
    HEAD_STAMP(0);
    float* Wl = lds;                                                   // [CW][LDW]  (later: the waves' partial dh [4][16][HP])
    const int wl_floats = max(a.CW * LDW, 64 * HP);
    float* red_w = lds + wl_floats;                                    // [4][16][2] the waves' statistics of the chunk in LDS
    float* cst = red_w + 128;                                          // [CC <= 16][16][2] every chunk's statistics of this row block
    int* miss = (int*)(cst + 512);                                     // [16] chunks whose workgroup did not publish in time
    // ---- 0. W chunk -> LDS (rows beyond the catalogue: zeros), h rows -> registers
    const int row = rb * 16 + j;
    f32x4 hb[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) hb[g] = *(const f32x4*)(a.h + (size_t)row * HP + 16 * g + 4 * q);
    const int y = a.tgt[row];
    const float scale = a.inv_Bg / a.pop[row];
    // (a target outside the catalogue hits no column: the row's cost is written here, or last step's value would be summed again)
    if (cc == 0 && wave == 0 && q == 0 && (unsigned)y >= (unsigned)a.N) a.rowcost[row] = 0.0f;
    if (tid < 16) miss[tid] = 0;
    head_fill<HP>(a, Wl, n_lo, tid);
    __syncthreads();
    HEAD_STAMP(1);
    // ---- 1. logits of this wave's tiles, 2. the waves' row statistics
    f32x4 lg[HEAD_NT];
    head_wave_stats<HP>(a, Wl, hb, n_lo, ntiles, wave, j, q, lg, red_w);
    HEAD_STAMP(2);
    __syncthreads();
    if (tid < 16) {                                                    // this chunk's (max, sum) of row tid: kept, and published
        float m, sx;
        head_chunk_combine(red_w, tid, m, sx);
        cst[(cc * 16 + tid) * 2] = m; cst[(cc * 16 + tid) * 2 + 1] = sx;
        unsigned* p = a.stats + ((size_t)(rb * a.CC + cc) * 16 + tid) * 4;
        const unsigned mb = __float_as_uint(m), sb = __float_as_uint(sx);
        __hip_atomic_store(p + 0, mb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, mb ^ a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 2, sb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 3, sb ^ a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    HEAD_STAMP(3);
    // The other chunks' statistics.  A workgroup that has not published within HEAD_WAIT_TICKS is not waited for: its chunk's
    // statistics are RECOMPUTED here (below) -- the grid is co-resident on an otherwise idle chip (one workgroup per CU), but two
    // processes sharing a GPU can interleave two such grids so that neither is ever complete: round 5's first form then spun for
    // 1.5 s and failed the step (tests/test_gpu_bench_contract.py::test_gpus_2_starts_its_own_ranks, two ranks on one GPU).  Now every
    // workgroup can always finish on its own.
    if (tid < a.CC * 16 && (tid >> 4) != cc) {
        const int c = tid >> 4, r = tid & 15;
        const unsigned* p = a.stats + ((size_t)(rb * a.CC + c) * 16 + r) * 4;
        const unsigned long long t0 = wall_clock64();
        unsigned d0, d1, d2, d3;
        bool ok = false;
        for (;;) {
            d0 = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d2 = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            d3 = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((d0 ^ d1) == a.epoch && (d2 ^ d3) == a.epoch) { ok = true; break; }
            if (wall_clock64() - t0 > a.wait_ticks) break;
            __builtin_amdgcn_s_sleep(2);
        }
        if (ok) { cst[(c * 16 + r) * 2] = __uint_as_float(d0); cst[(c * 16 + r) * 2 + 1] = __uint_as_float(d2); }
        else miss[c] = 1;
    }
    __syncthreads();
    {
        bool any = false;
        for (int c = 0; c < a.CC; ++c) {
            if (!miss[c]) continue;                                    // (LDS word: the same answer in every thread)
            any = true;
            __syncthreads();                                           // everyone has read miss[c] / finished with the image
            head_fill<HP>(a, Wl, c * a.CW, tid);
            __syncthreads();
            f32x4 lgx[HEAD_NT];
            head_wave_stats<HP>(a, Wl, hb, c * a.CW, ntiles, wave, j, q, lgx, red_w);
            __syncthreads();
            if (tid < 16) {
                float m, sx;
                head_chunk_combine(red_w, tid, m, sx);
                cst[(c * 16 + tid) * 2] = m; cst[(c * 16 + tid) * 2 + 1] = sx;
            }
        }
        if (any) {                                                     // this chunk's own rows again: phase 4 multiplies with them
            __syncthreads();
            head_fill<HP>(a, Wl, n_lo, tid);
            __syncthreads();
        }
    }
    float M = -INFINITY, S = 0.0f;
    for (int c = 0; c < a.CC; ++c) M = fmaxf(M, cst[(c * 16 + j) * 2]);
    for (int c = 0; c < a.CC; ++c) {                                   // chunk order: the same bits in every workgroup of the row block
        const float mc = cst[(c * 16 + j) * 2];
        if (mc > -INFINITY) S += cst[(c * 16 + j) * 2 + 1] * __builtin_amdgcn_exp2f((mc - M) * HEAD_LOG2E);
    }
    const float inv = 1.0f / S;
    HEAD_STAMP(4);
    // ---- 3. dlogits (in the registers of the tile), the row's cost
#pragma unroll
    for (int i = 0; i < HEAD_NT; ++i) {
        const int t = wave + 4 * i;
        if (t < ntiles) {
            const int c0 = n_lo + 16 * t + 4 * q;
            f32x4 d;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = lg[i][r];
                const float p = __builtin_amdgcn_exp2f((v - M) * HEAD_LOG2E) * inv;      // (column beyond the catalogue: 0)
                const bool hit = c0 + r == y;
                d[r] = (p - (hit ? 1.0f : 0.0f)) * scale;
                if (hit) a.rowcost[row] = (logf(S) + M - v) * scale;
            }
            lg[i] = d;
            if (c0 < a.N) *(f32x4*)(a.dlog + (size_t)row * a.Nl + c0) = d;
        } else lg[i] = f32x4{0, 0, 0, 0};
    }
    HEAD_STAMP(5);
    // ---- 4. dh partial of this chunk
    f32x4 da[KG];
#pragma unroll
    for (int kt = 0; kt < KG; ++kt) da[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < HEAD_NT; ++i) {
        const int t = wave + 4 * i;
        if (t < ntiles) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float* wr = Wl + (16 * t + 4 * q + e) * LDW + j;
#pragma unroll
                for (int kt = 0; kt < KG; ++kt) da[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * kt], lg[i][e], da[kt], 0, 0, 0);
            }
        }
    }
    asm volatile("s_nop 15");
    __syncthreads();                                                   // every wave has read its W rows: the image becomes the partials
    HEAD_STAMP(6);
    float* part = Wl + (size_t)wave * 16 * HP;
#pragma unroll
    for (int kt = 0; kt < KG; ++kt) *(f32x4*)(part + j * HP + 16 * kt + 4 * q) = da[kt];
    __syncthreads();
    float* slab = a.slabs + ((size_t)cc * a.Bp + (size_t)rb * 16) * HP;
    for (int i = tid; i < 16 * HP / 4; i += 256) {
        const f32x4 s0 = *(const f32x4*)(Wl + 4 * i), s1 = *(const f32x4*)(Wl + 16 * HP + 4 * i);
        const f32x4 s2 = *(const f32x4*)(Wl + 32 * HP + 4 * i), s3 = *(const f32x4*)(Wl + 48 * HP + 4 * i);
        *(f32x4*)(slab + 4 * i) = (s0 + s1) + (s2 + s3);
    }
    HEAD_STAMP(7);
#undef HEAD_STAMP
    // (Round 5 released the side stream from here -- every workgroup fenced and counted itself in, a one-lane gate kernel on the side
    // stream waited for the count -- to save the hipEventRecord between this kernel and the BPTT chain.  Round 6 measured the two forms
    // with the chain timed by its own stamps: 0.3348 ms with the gate against 0.3247 with the event -- 256 agent-scope fences at the
    // end of this launch cost the chain more than the record does.  Removed.)
}

// chunks / chunk width for this shape; false: not served
bool sbr_head_plan(int Bp, int N, int Hp, int* CC, int* CW, size_t* lds_bytes) {
    if (!(Hp == 32 || Hp == 64 || Hp == 128) || Bp < 16 || (Bp & 15) || Bp > 4096 || N < 1) return false;
    const int RB = Bp / 16;
    if (RB > 256) return false;
    const int cc = std::min(16, 256 / RB);
    if (cc < 1) return false;
    const int cw = ((N + cc - 1) / cc + 15) / 16 * 16;
    if (cw > 16 * 4 * HEAD_NT) return false;
    const size_t fl = (size_t)std::max(cw * (Hp + 4), 64 * Hp) + 128 + 512 + 32;      // image + wave statistics + chunk statistics + miss flags
    if (fl * 4 > 160 * 1024) return false;
    *CC = cc; *CW = cw; *lds_bytes = fl * 4;
    return true;
}

// slabs: CC * Bp * Hp floats; stats: (Bp / 16) * CC * 64 unsigned; false: shape not served, nothing launched
bool launch_head_cce(hipStream_t s, const float* h, const float* WoutT, const float* bout, const int* tgt, const float* pop, float* dlogits,
                     float* rowcost, float* slabs, size_t slab_floats, unsigned* stats, int* fault, int Bp, int N, int Nl, int Hp, int Bglobal,
                     unsigned epoch, int* n_slabs, hipError_t* err, unsigned long long* prof) {
    int CC = 0, CW = 0; size_t lds = 0;
    if (!sbr_head_plan(Bp, N, Hp, &CC, &CW, &lds) || (size_t)CC * Bp * Hp > slab_floats || epoch == 0) return false;
    HeadArgs a;
    a.h = h; a.W = WoutT; a.b = bout; a.tgt = tgt; a.pop = pop; a.dlog = dlogits; a.rowcost = rowcost; a.slabs = slabs; a.stats = stats;
    a.fault = fault; a.N = N; a.Nl = Nl; a.CW = CW; a.CC = CC; a.RB = Bp / 16; a.Bp = Bp; a.inv_Bg = 1.0f / (float)Bglobal; a.epoch = epoch;
    a.prof = prof;
    {   // read per launch: the tests flip it (0: nobody is waited for -- the recompute path serves every foreign chunk)
        const char* e = getenv("SBR_HEAD_WAIT_TICKS");
        a.wait_ticks = e ? strtoull(e, nullptr, 10) : HEAD_WAIT_TICKS;
    }
    const int grid = a.RB * CC;
    if (Hp == 128) { SBR_DYN_LDS(head_cce_kernel<128>, lds); head_cce_kernel<128><<<grid, 256, lds, s>>>(a); }
    else if (Hp == 64) { SBR_DYN_LDS(head_cce_kernel<64>, lds); head_cce_kernel<64><<<grid, 256, lds, s>>>(a); }
    else { SBR_DYN_LDS(head_cce_kernel<32>, lds); head_cce_kernel<32><<<grid, 256, lds, s>>>(a); }
    *n_slabs = CC;
    *err = hipGetLastError();
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// The SAMPLED head of one training step in one launch (round 6): activations of the C = Bg + S sampled cells, the sampled loss
// and its gradient, dh -- rnn_sampling.py:52-91 (Blackout / BPR / TOP1), rnn_cluster.py:158-175 (the cluster model's three).
// Before: four launches on the main stream between the two recurrent chains -- the activations GEMM 256 x 288 x Hp on twenty
// workgroups (36 us at C3, 58 at C5), sampled_loss_kernel (27 / 9), the dh GEMM (14 / 51), their gaps: 113 us of C3's 1.17 ms.
// Nothing in that work crosses batch rows, so a workgroup owns 16 rows and ALL cells (C <= 320: five 16-cell tiles per wave):
//   1. act[item][row] = sum_k Wc[item][k] h[row][k] on v_mfma_f32_16x16x4_f32 (exact f32 products), the gathered rows read straight
//      from memory: a lane's 16-byte piece holds the k slots of four instructions (head_logits above), one tile's sixteen (Hp = 256)
//      pieces per lane in flight; h rows in registers.  A lane ends with four consecutive cells of one batch row.
//   2. the loss row by row: what a row needs from its other cells (maximum, sums, the positive's activation) crosses the lanes of
//      a row by shuffles and the four waves through LDS, two or three rounds; the gradient replaces the activations in the
//      registers -- the layout the next phase multiplies with -- and is stored for the side stream's dWc GEMM.
//   3. dh[row][k] = sum_items dact[row][item] Wc[item][k]: the cell is the reduction index, and the same 16-byte pieces serve
//      again with the k of a tile PERMUTED (instruction 4 m + s of a lane's piece at 64 m + 4 j holds k = 64 m + 4 i + s for output
//      row i); the four waves' partial sums meet in LDS and leave as the finished dh rows.
// Bound by what one CU can pull: the gathered rows (0.3 MB at C3, 0.6 at C5) twice per workgroup.
// Served: Hp in {128, 256, 512}, full 16-row blocks, C <= 320, one direction.  Everything else keeps the four launches.
// ---------------------------------------------------------------------------------------------------------------------
struct SampArgs {
    const float* h; const float* Wc; const float* bc; const float* pop;
    float* act;            // [rows][C]  out: d cost / d act
    float* rowcost;        // [rows]
    float* dh;             // [rows][HP]
    int C, Bg, S, row_offset, loss;
    float inv_Bglobal;
    unsigned long long* prof;   // SBR_FLAG_PROFILE_REC: [workgroup][8] stamps of the 100 MHz clock (tools/head_prof.py c3), else NULL
};

// NW waves: eight where the partial dh rows of eight waves fit LDS (Hp <= 256) -- a wave's tiles are a serial chain of (sixteen loads ->
// 64 MFMAs), and sixteen workgroups are all the launch has: 73 us with four waves at C3 (profiles/round6_variants.txt, call s2)
template <int HP, int NW>
__global__ void __launch_bounds__(64 * NW) head_sampled_kernel(SampArgs a) {
    constexpr int KG = HP / 16, NT = (20 + NW - 1) / NW;
    extern __shared__ __attribute__((aligned(16))) float slds[];       // [NW][16][HP] partial dh | [3][NW][16] row reductions
    float* part = slds;
    float* red = slds + NW * 16 * HP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, q = lane >> 4;
    const int row = blockIdx.x * 16 + j, C = a.C, ntiles = (C + 15) >> 4;
#define SAMP_STAMP(I) do { if (a.prof && tid == 0) a.prof[(size_t)blockIdx.x * 8 + (I)] = wall_clock64(); } while (0)
    SAMP_STAMP(0);
    f32x4 hb[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g) hb[g] = *(const f32x4*)(a.h + (size_t)row * HP + 16 * g + 4 * q);
    // ---- 1. activations of this wave's tiles (tile t = wave + 4 i)
    f32x4 lg[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = wave + NW * i;
        f32x4 acc = f32x4{0, 0, 0, 0};
        if (t < ntiles) {                                              // (wave-uniform)
            const int item = 16 * t + j;
            const float* wr = a.Wc + (size_t)min(item, C - 1) * HP + 4 * q;
            f32x4 wv[KG];
#pragma unroll
            for (int g = 0; g < KG; ++g) wv[g] = *(const f32x4*)(wr + 16 * g);
            if (item >= C) {
#pragma unroll
                for (int g = 0; g < KG; ++g) wv[g] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int g = 0; g < KG; ++g)
#pragma unroll
                for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[g][m], hb[g][m], acc, 0, 0, 0);
        }
        lg[i] = acc;
    }
    asm volatile("s_nop 15");                                          // MFMA D -> VALU read
    SAMP_STAMP(1);
    // ---- 2. the loss of row j.  A lane holds cells c = 16 (wave + 4 i) + 4 q + r of its row.
    const int pos = a.row_offset + row;
    const float scale = a.inv_Bglobal / a.pop[row];
    bool valid[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 16 * (wave + NW * i) + 4 * q + r;
            valid[i][r] = c < C;
            if (valid[i][r] && a.bc) lg[i][r] += a.bc[c];
        }
    // sums / maxima over a row: its four lanes by shuffles, the four waves through LDS (slot k of `red`)
    auto row_sum = [&](float v, int k) -> float {
        v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
        if (q == 0) red[(k * NW + wave) * 16 + j] = v;
        __syncthreads();
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < NW; ++w) t += red[(k * NW + w) * 16 + j];
        return t;
    };
    auto row_max = [&](float v, int k) -> float {
        v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 32));
        if (q == 0) red[(k * NW + wave) * 16 + j] = v;
        __syncthreads();
        float t = -INFINITY;
#pragma unroll
        for (int w = 0; w < NW; ++w) t = fmaxf(t, red[(k * NW + w) * 16 + j]);
        return t;
    };
    float ap = 0.0f;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) if (valid[i][r] && 16 * (wave + NW * i) + 4 * q + r == pos) ap = lg[i][r];
    const float apos = row_sum(ap, 0);
    float L;
    const int loss = a.loss, Bg = a.Bg;
    if (loss == SBR_LOSS_BLACKOUT || loss == SBR_LOSS_SCCE) {           // rnn_sampling.py:68-72; SCCE: rnn_cluster.py:158-162
        const bool blackout = loss == SBR_LOSS_BLACKOUT;
        float mx = -INFINITY;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (valid[i][r]) mx = fmaxf(mx, lg[i][r]);
        mx = row_max(mx, 1);
        float se = 0.0f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (valid[i][r]) se += expf(lg[i][r] - mx);
        se = row_sum(se, 2);
        __syncthreads();                                               // (slot 0 is written again below)
        const float inv = 1.0f / se;
        const float ppos = expf(apos - mx) * inv;
        float dot = 0.0f, lneg = 0.0f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * (wave + NW * i) + 4 * q + r;
                if (valid[i][r] && c >= Bg && blackout) { const float p = expf(lg[i][r] - mx) * inv; dot += p / (1.0f - p); lneg -= logf(1.0f - p); }
            }
        dot = row_sum(dot, 0) - 1.0f;                                  // positive: (-1 / p_pos) * p_pos
        lneg = row_sum(lneg, 1);
        L = -logf(ppos) + lneg;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * (wave + NW * i) + 4 * q + r;
                float d = 0.0f;
                if (valid[i][r]) {
                    const float p = expf(lg[i][r] - mx) * inv;
                    float dldp = 0.0f;
                    if (c >= Bg && blackout) dldp = 1.0f / (1.0f - p);
                    if (c == pos) dldp += -1.0f / p;
                    d = p * (dldp - dot) * scale;
                }
                lg[i][r] = d;
            }
    } else {
        float lsum = 0.0f, dpos = 0.0f;
        const float fS = (float)a.S;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * (wave + NW * i) + 4 * q + r;
                float da = 0.0f;
                if (valid[i][r] && c != pos && c >= Bg) {
                    const float v = lg[i][r], diff = v - apos;
                    if (loss == SBR_LOSS_BPR) {                        // :80-84  -log(sigmoid(-diff)) = softplus(diff)
                        lsum += fmaxf(diff, 0.0f) + log1pf(expf(-fabsf(diff)));
                        const float dd = (1.0f / (1.0f + expf(-diff))) / fS;
                        da = dd; dpos += dd;
                    } else if (loss == SBR_LOSS_BPRELU) {              // rnn_cluster.py:173-175
                        const float yv = diff + 0.5f;
                        lsum += yv > 0.0f ? yv : 0.01f * yv;
                        const float dd = (yv > 0.0f ? 1.0f : 0.01f) / fS;
                        da = dd; dpos += dd;
                    } else if (loss == SBR_LOSS_LIN) {                 // rnn_cluster.py:164-167
                        lsum += v;
                        da = 1.0f;
                    } else {                                           // TOP1 :86-91
                        const float s1 = 1.0f / (1.0f + expf(-diff)), s2 = 1.0f / (1.0f + expf(-v * v));
                        lsum += s1 + s2;
                        const float d1 = s1 * (1.0f - s1) / fS;
                        da = d1 + s2 * (1.0f - s2) * 2.0f * v / fS; dpos += d1;
                    }
                }
                lg[i][r] = da * scale;
            }
        lsum = row_sum(lsum, 1);
        dpos = row_sum(dpos, 2);
        L = lsum / fS;
        if (loss == SBR_LOSS_LIN) { L = lsum - apos; dpos = 1.0f; }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) if (valid[i][r] && 16 * (wave + NW * i) + 4 * q + r == pos) lg[i][r] = -dpos * scale;
    }
    SAMP_STAMP(2);
    if (wave == 0 && q == 0) a.rowcost[row] = L * scale;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int c0 = 16 * (wave + NW * i) + 4 * q;
        if (c0 + 3 < C && (C & 3) == 0) *(f32x4*)(a.act + (size_t)row * C + c0) = lg[i];
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (c0 + r < C) a.act[(size_t)row * C + c0 + r] = lg[i][r];
        }
    }
    SAMP_STAMP(3);
    // ---- 3. dh: cells are the reduction index; lane (j, q) multiplies its own dact (cell 16 t + 4 q + e of row j)
    f32x4 da[KG];
#pragma unroll
    for (int kt = 0; kt < KG; ++kt) da[kt] = f32x4{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int t = wave + NW * i;
        if (t < ntiles) {
            f32x4 w4[4][KG / 4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int item = min(16 * t + 4 * q + e, C - 1);       // (a cell past C multiplies a zero gradient)
#pragma unroll
                for (int m = 0; m < KG / 4; ++m) w4[e][m] = *(const f32x4*)(a.Wc + (size_t)item * HP + 64 * m + 4 * j);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < KG / 4; ++m)
#pragma unroll
                    for (int sidx = 0; sidx < 4; ++sidx)
                        da[4 * m + sidx] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4[e][m][sidx], lg[i][e], da[4 * m + sidx], 0, 0, 0);
        }
    }
    asm volatile("s_nop 15");
    SAMP_STAMP(4);
    // instruction 4 m + s left k = 64 m + 4 (4 q + r) + s of row j in element r
    float* pw = part + (size_t)(wave * 16 + j) * HP;
#pragma unroll
    for (int kt = 0; kt < KG; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) pw[64 * (kt >> 2) + 16 * q + 4 * r + (kt & 3)] = da[kt][r];
    __syncthreads();
    SAMP_STAMP(5);
    float* out = a.dh + (size_t)blockIdx.x * 16 * HP;
    for (int i = tid; i < 16 * HP / 4; i += 64 * NW) {
        f32x4 t = *(const f32x4*)(part + 4 * i);
#pragma unroll
        for (int w = 1; w < NW; ++w) t += *(const f32x4*)(part + (size_t)w * 16 * HP + 4 * i);
        *(f32x4*)(out + 4 * i) = t;
    }
    SAMP_STAMP(6);
#undef SAMP_STAMP
}

// false: shape not served, nothing launched (the caller keeps the four launches)
bool launch_head_sampled(hipStream_t s, const float* h, const float* Wc, const float* bc, const float* pop, float* act, float* rowcost,
                         float* dh, int rows, int C, int Hp, int Bg, int S, int row_offset, int loss, int Bglobal, hipError_t* err,
                         unsigned long long* prof) {
    if (!(Hp == 128 || Hp == 256 || Hp == 512) || rows < 16 || (rows & 15) || C < 1 || C > 320) return false;
    if (loss == SBR_LOSS_CCE || SBR_LOSS_IS_MARGIN(loss)) return false;
    SampArgs a{h, Wc, bc, pop, act, rowcost, dh, C, Bg, S, row_offset, loss, 1.0f / (float)Bglobal, prof};
    const int nw = Hp == 512 ? 4 : 8;
    const size_t lds = ((size_t)nw * 16 * Hp + 3 * nw * 16) * sizeof(float);
    const int grid = rows / 16;
    if (Hp == 512) { SBR_DYN_LDS((head_sampled_kernel<512, 4>), lds); head_sampled_kernel<512, 4><<<grid, 256, lds, s>>>(a); }
    else if (Hp == 256) { SBR_DYN_LDS((head_sampled_kernel<256, 8>), lds); head_sampled_kernel<256, 8><<<grid, 512, lds, s>>>(a); }
    else { SBR_DYN_LDS((head_sampled_kernel<128, 8>), lds); head_sampled_kernel<128, 8><<<grid, 512, lds, s>>>(a); }
    *err = hipGetLastError();
    return true;
}
