// Internal declarations shared by the HIP translation units of libsbr_rnn.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "../../include/sbr_rnn.h"

#define SBR_BWD_CHUNKS 4       // BPTT launches per layer when T >= 64 (bf16x6 kernels)
#define SBR_ALIGN_FLOATS 64   // every carved buffer starts on a 256-byte boundary

static inline size_t sbr_align(size_t n_floats) { return (n_floats + SBR_ALIGN_FLOATS - 1) / SBR_ALIGN_FLOATS * SBR_ALIGN_FLOATS; }
static inline int sbr_gates(int cell) { return cell == SBR_CELL_LSTM ? 4 : (cell == SBR_CELL_GRU ? 3 : 1); }
// Hidden size padded for the MFMA tiling: 16/32/64/128 (W_hid register-resident single-workgroup kernels), 256 for
// anything in (128, 256] and 512 for (256, 512] (the cluster kernels of sbr_rec_cl.hip: a 150-wide layer padded to 256
// runs ~6x faster there than at 192 on the streamed f32 kernels), above that the next multiple of 64 (streamed-W kernels).
static inline int sbr_pad_hidden(int H) {
    if (H <= 16) return 16;
    if (H <= 32) return 32;
    if (H <= 64) return 64;
    if (H <= 128) return 128;
    if (H <= 256) return 256;
    if (H <= 512) return 512;
    return (H + 63) / 64 * 64;
}

// ---------------------------------------------------------------------------------------
// Parameter / activation layout in the device arena (all float32, offsets in floats)
// ---------------------------------------------------------------------------------------
struct LayerLayout {
    int H, Hp, G;
    int n_in;        // logical rows of W_in (input_size for layer 0, H of the layer below otherwise)
    int n_in_p;      // stored rows (== n_in for layer 0, Hp of the layer below otherwise)
    // parameter section offsets (the gradient section mirrors them exactly)
    size_t p_Win;    // [n_in_p][G*Hp]   item-major rows, gate-major inside a row
    size_t p_b;      // [G*Hp]
    size_t p_Whid;   // [Hp][G*Hp]
    size_t p_peep;   // [3][Hp]          LSTM only (i, f, o)
    size_t p_cinit;  // [Hp]             LSTM only
    size_t p_hinit;  // [Hp]
    // activation offsets (activation section)
    size_t a_xt;     // [T][Bp][G*Hp]    precomputed input (gather or dense GEMM)
    size_t a_hs;     // [T+1][Bp][Hp]    slot 0 = hid_init, slot t+1 = h_t
    size_t a_cs;     // [T+1][Bp][Hp]    LSTM cell states
    size_t a_xh, a_pring;   // wide layers (Hp = 256 / 512): exchange arrays of the 16-row cluster kernels, else 0
    size_t a_g[4];   // [T][Bp][Hp]      LSTM i,f,g,o / GRU r,u,c~,hid_c
    size_t a_dxt;    // [T][Bp][G*Hp]    grad wrt xt (= grad wrt gates for LSTM/Vanilla)
    size_t a_dhi;    // [T][Bp][Hp]      GRU only: candidate-gate slice of grad wrt hid_input (r,u slices equal dxt)
    size_t a_dhext;  // [T][Bp][Hp]      grad arriving from the layer above (layers below the top)
    size_t a_state;  // [2][Bp][Hp] dh, dc carried between BPTT chunk launches
    size_t a_part;   // [SBR_BWD_CHUNKS][Bp][G*Hp + 5*Hp] per-workgroup partial sums (up to one workgroup per row): bias, peepholes, inits
};

// A row-sparse parameter block: rows of up to two arrays that are touched together (the forward and backwards layer-0
// W_in under --r_bi; W_out^T and b_out of a sampled head) share one row index space and one `last` array.
struct SparseBlockLayout {
    int kind;            // 0: rows indexed by the batch's input ids (layer-0 W_in or the embedding table); 1: by the sampled cells
    int npairs;
    size_t off[2];       // offset of row 0 inside the parameter / gradient / state sections
    int width[2];        // floats of a row that are stored there
    int stride[2];       // floats between consecutive rows
    int n_rows;
    int W;               // packed row width (sum of the widths) in the exchange buffers
    int max_local;       // most rows one rank can touch per step (capacity of its exchange buffers)
    size_t a_last;       // [n_rows] ints: step through which the row is current
    size_t a_mark;       // [n_rows] ints: pack epoch (dedupe of the exchange)
    size_t a_cand;       // [cand_cap] ints: the step's candidate rows in a data-parallel step (ids of every rank)
    size_t a_count;      // device int: rows packed by the last sbr_sparse_pack
    int cand_cap;
};

struct Layout {
    sbr_config cfg;
    int n_sparse;                         // row-sparse blocks (0: every parameter takes the dense update)
    SparseBlockLayout sparse[2];
    size_t a_at; int n_at; int adam_early_exit;   // adam's a_t table (floats) for the lazy catch-up
    int L, G, T, B, Bp, N, F, Bg, S, C;   // C = Bg + S sampled columns
    int HLp;                              // padded width of one direction of the top layer
    int D, HLt;                           // directions per level (2 with --r_bi) and the output layer's input width D * HLp
    // --r_bi only (the backwards direction runs the ordinary kernels on per-row time-reversed copies of its input):
    size_t a_Xr;                          // [Bp][T][F] ints: item ids reversed inside each row's valid length
    size_t a_embr;                        // [T][Bp][F*Ep] reversed flattened embeddings (--r_emb)
    size_t a_cat[SBR_MAX_LAYERS], a_catr[SBR_MAX_LAYERS];   // [T][Bp][2*Hp_l] concatenated outputs of level l (forward time / reversed)
    size_t a_hcat;                        // [Bp][2*HLp] final states of both directions
    size_t a_dhl[2];                      // [Bp][HLp] halves of dh_last
    size_t a_dinp[2];                     // [T][Bp][max dense n_in_p] gradient wrt a dense level's input, per direction
    size_t a_s2cnt, a_s2off, a_s2cur, a_s2sid, a_s2pos;     // scatter sort workspace of the reversed ids
    int E, Ep;                            // --r_emb: embedding width and its stored width (multiple of 4); 0 = none
    size_t p_Emb;                         // [input_size][Ep]
    size_t a_emb, a_demb;                 // [T][Bp][F*Ep] flattened embeddings = dense input of layer 0, and its gradient
    LayerLayout layer[2 * SBR_MAX_LAYERS];   // [l * D + d]: d = 0 forward, 1 backwards (--r_bi)
    size_t p_WoutT, p_bout;               // [N][HLp], [N]
    size_t n_params;                      // floats in the parameter section
    size_t p_split;                       // == p_WoutT : output-layer part starts here
    size_t n_state_arrays;                // optimizer state arrays of n_params floats each (1 or 2)
    // arena sections (float offsets from the arena base)
    size_t s_params, s_grads, s_state, s_act, s_end;
    // misc activation-section buffers
    size_t a_logits;                      // [Bp][Nl], Nl = N rounded up to 4 floats in the training step (16-byte rows for the
                                          // bf16x6 GEMMs that read dlogits); [rows][N] in predict / top-k
    size_t a_dhlast;                      // [Bp][HLp]
    size_t a_rowcost;                     // [Bp]
    int NT;                               // RNNMargin: target columns per row (1 for the other heads)
    size_t a_dflt;                        // RNNMargin: [N] default target (zeros unless sbr_set_default_target)
    size_t a_Wc, a_bc, a_act, a_dWc, a_dbc; // sampled heads: [C][HLp], [C], [Bp][C], [C][HLp], [C]
    size_t a_ws; size_t ws_floats;        // split-K workspace (main stream)
    size_t a_ws2; size_t ws2_floats;      // split-K workspace of the side stream (output-layer + weight gradients)
    size_t a_ws3; size_t ws3_floats;      // split-K workspace of the output layer's dW_out GEMM when it runs on a stream of its own (SBR_TAIL_OUT_STREAM)
    size_t a_csum;                        // [16][max(N,C)] column-sum partials
    size_t a_prof;                        // [2][nblk][16][4] uint64 in-kernel cycle counters (fwd, bwd)
    size_t a_fault;                       // int: a cluster exchange wait timed out
    size_t a_clx;                         // cluster handshake slots (ints)
    size_t a_X, a_len, a_tgt, a_smp, a_cells, a_pop, a_topk; // batch buffers (ints stored in float slots)
    size_t a_X2, a_len2, a_tgt2, a_smp2, a_pop2;             // ... second set: sbr_build_batch fills the set the step in flight does not read
    size_t a_scnt, a_soff, a_scur, a_sid, a_spos;            // scatter counting-sort workspace (ints)
    size_t a_sP;                          // [input_size + 1] running cost of the ids (launch_scatter_lds_poll)
    size_t a_hstat;                       // [256][16][4] row statistics the chunks of the fused head exchange (sbr_head.hip)
    size_t a_srpart, a_srid; int sr_slots;   // scatter-add of wide rows: partial rows of the long segments' pieces, counters + records (0: not used)
    int tail_keys;                        // time chunks the sort's key space was sized for (1: no tail overlap possible)
    size_t a_prog;                        // [Bp / 4 * 8] progress words of the running BPTT chain (tail overlap)
    size_t a_done;                        // [SBR_DONE_COPIES * SBR_DONE_STRIDE] the monitor's word (the minimum over a_prog), replicated
};

int sbr_build_layout(const sbr_config& cfg, Layout& lay, std::string& err);

struct ParamDesc { std::string name; int layer; int kind; int gate; int64_t d0, d1; int ndim; };
// kind: 0 W_in, 1 W_hid, 2 b, 3 peephole(i,f,o by gate 0..2), 4 cell_init, 5 hid_init, 6 out.W, 7 out.b
void sbr_param_descs(const Layout& lay, std::vector<ParamDesc>& out);

// Time chunks of the time-chunked sort (overlapped step tail): chunk c holds the time steps [lo[c], lo[c + 1]), n chunks;
// n <= 1: plain keys.  The chunks need not be equal: the BPTT chain completes chunk 0 last, and whatever of the scatter-add
// belongs to it can only start at the chain's end -- so the chunks shrink geometrically towards t = 0.
#define SBR_TCHUNKS_MAX 9
struct SbrTChunks { int n; int lo[SBR_TCHUNKS_MAX + 1]; };
static inline SbrTChunks sbr_uniform_tchunks(int tch, int n) {
    SbrTChunks tc; tc.n = tch > 0 ? n : 0;
    for (int c = 0; c <= SBR_TCHUNKS_MAX; ++c) tc.lo[c] = tch * (c < n ? c : n);
    return tc;
}
struct sbr_handle {
    Layout lay;
    float* arena; bool own_arena;
    hipStream_t stream;
    hipStream_t side;            // batch-only preprocessing (scatter sort) overlapped with the chain
    hipEvent_t ev_fork, ev_join;
    hipEvent_t ev_sort, ev_lg, ev_fill, ev_og, ev_chunk[SBR_BWD_CHUNKS];
    bool in_train_step;  // phases called from sbr_train_step: the side stream joins only before the update
    bool side_pending;   // side-stream work issued and not yet joined by the main stream
    bool deferred_join;  // phases called one by one do not join the side stream (sbr_set_deferred_join)
    bool og_recorded;    // ev_og marks the output-layer gradients of this step complete
    bool fill_done;      // the cluster BPTT sentinel fill of this step was issued on the side stream (ev_fill)
    bool out_early;      // this step's output-layer parameters were stepped on the side stream beside the BPTT chain
    int sparse_out_early;    // SBR_SPARSE_OUT_EARLY (default 1): the sampled head's row-sparse block is caught up beside the forward
                             // chain and stepped beside the BPTT chain (side stream) instead of in front of / behind them
    bool cells_early;    // this step: cells built + their rows caught up on the side stream by sbr_forward (ev_cells)
    bool wout_early;     // this step: the sampled head's rows were stepped beside the BPTT chain
    hipEvent_t ev_cells;
    int head_fuse;       // SBR_HEAD_FUSE (default 1): the full-softmax head in one launch (sbr_head.hip)
    unsigned head_epoch;
    int row_aware;       // SBR_ROW_AWARE_UPDATE (default 1): the dense pass over a wide index-input block skips the gradient traffic of the rows the batch did not touch
    int out_fuse;        // SBR_OUT_FUSE (default 1): the dense head's gradient and step in one launch (launch_out_grad_step)
    bool out_stepped;    // this step: done, the output layer's range needs no update launch
    int dh_slabs_n;      // > 0: dh_last of this step sits in the main workspace as that many unreduced split-K slabs
    std::vector<ParamDesc> descs;
    int rpt;             // rows per workgroup for the bf16x6 recurrent kernels
    int bwd_chunks;      // BPTT launches per layer (1..SBR_BWD_CHUNKS)
    int wgrad_slices;    // K-slices of the weight-gradient kernel (total over the chunks)
    int cluster, cl_linear; // cluster recurrent kernels for wide layers (SBR_CLUSTER, SBR_CL_LINEAR)
    int cl_epoch;
    int x6_split, fuse_gather;
    int sp_exchanged[2]; // data-parallel step: rows of block b were packed / gathered (candidates = a_cand[0 .. sp_ncand[b]))
    int sp_ncand[2];
    int sp_epoch;        // pack epoch
    float* lag_host;     // pinned: [2] cost, [2] fault flag, [2] sequence number (sbr_train_step_lagged: lag_report_kernel)
    unsigned lag_seq[2], lag_counter;
    int lag_slot, lag_pending;
    int x6_pipe;         // per-k-block publish counters instead of a workgroup barrier per step (SBR_X6_PIPE, default 1; Hp = 128)
    int wgrad_x6;        // weight gradients through the bf16x6 GEMM instead of the dedicated f32 kernel (SBR_WGRAD_X6, default 1; the f32 kernel serves Hp < 96 and SBR_FLAG_F32_MFMA)
    // current batch: the arena's own buffers, or (device-resident inputs covering all Bp rows) the caller's
    const int *bX, *blen, *btgt, *bsmp; const float* bpop;
    // sbr_build_batch beside the step in flight (sbr_batch.hip): its own stream, the set the current batch sits in, and what
    // tells it that the set it is about to overwrite is no longer read
#ifndef SBR_BB_STREAM
#define SBR_BB_STREAM 0      // (probe builds: 1 = a stream of its own, 2 = ... at low priority)
#endif
    hipStream_t s_bb; hipEvent_t ev_bb, ev_bbw;
    int bb_set;                 // arena set of the current batch (0: also what sbr_set_batch fills)
    uint64_t batch_seq;         // sbr_forward calls so far
    uint64_t set_use[2];        // batch_seq of the last forward that read set i
    uint64_t lg_seq;            // batch_seq when ev_lg_rec was last recorded (a main-stream record in the middle of a training step)
    bool bb_unread;             // the current batch was built and no forward has read it yet
    bool train_fwd_open;        // a training forward whose step has not reached sbr_apply_update
    int bb_slow;                // builds left that wait for all of the engine's streams (a step was abandoned: its side streams were not joined)
    int n_rows;          // rows of the current batch (<= local_batch)
    int64_t step_count;  // adam t
    bool have_batch, fwd_done;
    bool grads_clean;    // the gradient section is all zero (fresh arena, or the update kernel cleared it)
    bool timing;
    unsigned timing_marks;   // which of the SBR_N_PHASES event marks a step records (sbr_enable_timing)
    int tail_overlap;    // SBR_TAIL_OVERLAP (default 1): weight-gradient GEMM and scatter-add of finished time chunks run beside the BPTT chain
    int tail_chunks_max; // SBR_TAIL_CHUNKS (default 8)
    int tail_pub_every;  // SBR_TAIL_PUBLISH_EVERY: time steps between two progress words of a chain wave (default 2)
    int tail_nc, tail_ch;   // this step: time chunks of the sort's keys / steps per chunk (0: plain keys)
    SbrTChunks tail_bounds; // ... and their bounds (tail_plan)
    unsigned long long* tail_trace = nullptr;    // SBR_TAIL_TRACE=1: SbrPoll.trace
    int tail_fence_kb;      // SBR_TAIL_FENCE_KB: LDS the chains claim while consumers / the sort run beside them (0: none)
    int tail_early_sort;    // SBR_TAIL_EARLY_SORT: the time-chunked sort beside the forward chain
    int tail_out_stream;    // SBR_TAIL_OUT_STREAM: output-layer gradients + update on a third side stream (single-call steps)
    bool tail_sorted, out3; // this step: the sort already ran (sbr_forward) / the output layer's work is on side3
    hipStream_t out3_stream;   // this step: the stream of the output layer's gradient kernels when they left the side stream (out3)
    hipStream_t side3; hipEvent_t ev_tail3;      // side3: the overlapped tail's monitor (tail_monitor_kernel)
    int tail_short_chunks;  // time chunks (from t = 0) whose scatter-add entries are cut into short pieces (SBR_TAIL_SHORT_CHUNKS)
    double tail_geom;       // SBR_TAIL_GEOM: growth of the small time chunks near t = 0 (<= 1: equal chunks)
    bool tail_cost_scanned = false;                                       // this step's sort was followed by launch_scatter_cost_scan
    int scnt_zero_n = 0;                                                  // leading counters of a_scnt known to be zero (launch_scatter_sort)
    int tail_mon_units;                                                   // SBR_TAIL_MONITOR_IN_UNITS: the monitor is a workgroup of the scatter-add launch
    int tail_first;                                                       // SBR_TAIL_FIRST: time steps of the last time chunk (LDS-row scatter-add)
    int tail_scatter_lds, tail_scatter_units;                             // SBR_TAIL_SCATTER_LDS / SBR_TAIL_SCATTER_UNITS
    int tail_gemm_groups;                                                 // SBR_TAIL_GEMM_GROUPS: persistent groups of the polling dW_hid GEMM
    int tail_fuse_slabs, tail_slab_max;                                   // SBR_TAIL_FUSE_SLABS / SBR_TAIL_SLAB_MAX (rows)
    double tail_slab_growth;                                              // SBR_TAIL_SLAB_GROWTH (k steps of a slab per time step of lead)
    std::vector<int> tail_slab_host; int* tail_slab_dev = nullptr; int tail_slab_key[2] = {0, 0};   // the table of the last plan
    bool fold_dh;           // SBR_FOLD_DH
    int wgrad_f16, wgrad_x6_wgs;                                          // SBR_WGRAD_F16, SBR_WGRAD_X6_WGS
    int prog_epoch;
    bool tail_updated;      // this step: the overlapped tail has applied the optimizer itself (single-call step)
    hipEvent_t ev_tail, ev_tail2;
    bool tail_join_pending; // overlapped tail of a phase-by-phase step: the main stream has not joined the consumer streams yet
    bool step_open;         // sbr_zero_grads has opened a training step (cleared by sbr_forward)
    hipEvent_t ev_lg_rec;   // this step: the main-stream record that released the side stream (sbr_loss_backward_output)
    hipStream_t side2;      // second consumer stream of the overlapped tail (scatter-add)
    bool swap_tail;      // SBR_SWAP_TAIL: after the BPTT chain the main stream keeps dW_hid, the side stream takes the scatter
    bool tail_swapped;   // ... done for this step (sbr_apply_update splits its ranges accordingly)
    unsigned marks_shared;   // marks of this step already recorded as a cross-stream event (record_shared)
    // ring of per-step event sets, read back after the timed region (no per-step sync)
    static const int kRing = 64;
    hipEvent_t ev[kRing][SBR_N_PHASES];
    int ring_used;       // train steps recorded since timing was enabled
    int ring_cur;        // set used by the step in flight
    // chain-only timing (sbr_chain_times): an event pair around every launch of a recurrent chain kernel, any layer, either direction
    static const int kChain = 256;
    hipEvent_t ev_ch[kChain][2];
    unsigned char ch_dir[kChain];
    int ch_n = 0;
    bool chain_timing = false;
    float* P(size_t off) const { return arena + lay.s_params + off; }
    float* Gd(size_t off) const { return arena + lay.s_grads + off; }
    float* St(int k, size_t off) const { return arena + lay.s_state + (size_t)k * lay.n_params + off; }
    float* A(size_t off) const { return arena + lay.s_act + off; }
    float* cost_ptr() const { return arena + lay.s_grads + lay.n_params; }
};

void sbr_set_error(const char* fmt, ...);
#define SBR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    sbr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return SBR_EHIP; } } while (0)

#define CHECK_ARG(cond, ...) do { if (!(cond)) { sbr_set_error(__VA_ARGS__); return SBR_EINVAL; } } while (0)
#define SBR_LAUNCH(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
    sbr_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return SBR_EHIP; } } while (0)

// Consumers of a RUNNING BPTT chain (overlapped step tail): the chain's waves publish (epoch << 12) | t in words[0 .. n) once
// all their time steps >= t are complete and written through (RecArgs.progress); ONE workgroup -- the MONITOR: workgroup 0 of
// the scatter-add launch, or tail_monitor_kernel -- folds them into `done` = (epoch << 12) | max t, which every consumer polls.  rows_per_step: K rows per time step.
// Slabs of a polling GEMM are K-ascending, slab z = rows [slab_lo[z], slab_lo[z + 1]) (a device table of multiples of 32, see
// sbr_tail_slab_table: one k step of 32 rows for the time steps the chain reaches last, growing with the time the chain still
// needs once a slab is released -- a slab must be DONE when the chain ends, not started); workgroups take them from the far end.
// The monitor's word exists in SBR_DONE_COPIES copies, SBR_DONE_STRIDE ints apart (4 KB + 256 B: other pages AND other channels):
// hundreds of waves poll it with agent-scope loads, which are served by the memory side, and each poller reads the copy its
// workgroup number selects.  (Built on the suspicion that ONE word makes every poll queue at one channel; the consumers' slow
// loads of profiles/round3_n_trace.txt turned out to have other causes -- DESIGN.md 3a -- and the copies cost nothing.)
// Dynamic LDS above the default limit needs the function attribute -- once per kernel and size, not per launch (a driver call
// each: six of them per training step before round 3c, on a path whose host side is as long as its device side).  One static
// per expansion site, i.e. per kernel; one process drives one GPU.
#define SBR_DYN_LDS(KERNEL, LDS) do { static int sbr_dyn_lds_set_ = -1; if ((int)(LDS) > sbr_dyn_lds_set_) { \
        (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS)); sbr_dyn_lds_set_ = (int)(LDS); } } while (0)
#define SBR_DONE_COPIES 32
#define SBR_DONE_STRIDE 1088
struct SbrPoll { const int* words; int n; int* done; int epoch; int rows_per_step; int* fault; int n_small, k_small;
                 unsigned long long* trace; const int* slab_lo; int n_slabs; };    // trace (SBR_TAIL_TRACE=1, tools/tail_trace.py): 100 MHz stamps of the consumers' waits and ends

// ---------------------------------------------------------------------------------------
// Kernel launchers (each returns hipGetLastError())
// ---------------------------------------------------------------------------------------
// K1: xt[t][b][:] = sum_f W_in[X[b][t][f]][:] + bias         (sparse_lstm.py:368,:755,:1111)
hipError_t launch_gather_xt(hipStream_t s, const float* Win, const float* bias, const int* X, float* xt,
                            int T, int Bp, int F, int GHp, int n_rows_in);
// K6: dWin[X[b][t][f]][:] += dxt[t][b][:] for t < len[b]     (AdvancedIncSubtensor grad [3P])
hipError_t launch_scatter_rows(hipStream_t s, float* dWin, const float* dxt, const int* X, const int* len,
                               int T, int Bp, int F, int GHp);
// counting sort of the valid (position, id) pairs by id (depends on the batch only), then one wave per
// chunk of 32 sorted entries reduces the dxt rows of equal id in registers
// concat != 0: entry (pos, f) addresses row pos*F + f of the gradient array (embedding layer: the F embeddings of a step are
// concatenated, not summed)
// tch > 0: time-chunked keys (t / tch) * n_ids + id over n_tchunks chunks (cnt / offs / cur then hold n_tchunks * n_ids + 1 ints)
hipError_t launch_scatter_sort(hipStream_t s, const int* X, const int* len, int T, int Bp, int F, int n_ids, int* cnt,
                               int* offs, int* cur, int* sid, int* spos, int concat = 0, int tch = 0, int n_tchunks = 1, const SbrTChunks* bounds = nullptr,
                               int* cnt_zero_n = nullptr);
int sbr_scatter_lds_ids();     // largest key space the LDS-histogram sort takes
// overlapped step tail: wait (bounded) until every progress word of the running BPTT chain is (epoch, <= target)
hipError_t launch_tail_gate(hipStream_t s, const int* progress, int n, int epoch, int target, int* fault);
// emb[t][b][f*Ep + e] = W_emb[X[b][t][f]][e]          (lasagne EmbeddingLayer + flatten(outdim=3), recurrent_layers.py:48)
hipError_t launch_gather_concat(hipStream_t s, const float* Wemb, const int* X, float* out, int T, int Bp, int F, int Ep);
// --r_bi helpers (sbr_misc.hip).  rev(t, len) = t < len ? len-1-t : t (padding stays in place).
hipError_t launch_rev_rows_int(hipStream_t s, const int* X, const int* len, int* Xr, int T, int Bp, int F);
hipError_t launch_rev_rows(hipStream_t s, const float* src, const int* len, float* dst, int T, int Bp, int W);   // dst[t][b] = src[rev(t)][b]
// cat[t][b] = [hs_f[t+1][b] | hs_b[rev(t)+1][b]]   (hs_*: [T+1][Bp][Hp], slot t+1 = state after step t of that scan)
hipError_t launch_cat_outputs(hipStream_t s, const float* hs_f, const float* hs_b, const int* len, float* cat, int T, int Bp, int Hp);
hipError_t launch_hcat(hipStream_t s, const float* hf, const float* hb, float* out, int Bp, int Hp);            // [hf | hb]
hipError_t launch_split_cols(hipStream_t s, const float* src, float* a, float* b, int rows, int Hp);             // src [rows][2Hp] -> a, b
// d[t][b] = f[t][b] + r[rev(t)][b] over W columns; then out_f[t][b] = d[t][b][0:Hp], out_b[t][b] = d[rev(t)][b][Hp:2Hp]
// (out_b == NULL: W arbitrary, only the sum d is written to out_f -- the embedding case)
hipError_t launch_uncat(hipStream_t s, const float* f, const float* r, const int* len, float* out_f, float* out_b, int T, int Bp,
                        int W, int Hp);
// wide rows (G*Hp >= 512), form 1: range scatter-add, two passes, no atomics (sbr_misc.hip); false: not served
bool launch_scatter_range(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs, int n_ids,
                          int GHp, float* part, int* part_id, int n_ranges, hipError_t* err);
#define SBR_SCAT_RANGES 512
// form 2: segment-parallel scatter-add, no atomics on rows; false: not served
bool launch_scatter_wide(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs, int n_ids,
                         int max_entries, int GHp, float* part, int* aux, int n_slots, hipError_t* err);
// arguments of the optimizer step of a block's rows (update_rows_aware_kernel, out_grad_step_kernel): p / s0 / s1 = parameter and
// optimizer-state rows (s1 NULL where the updater has one state array); the arithmetic is update_kernel's, element for element
struct SbrScatStep { float* p; float* s0; float* s1; int* last; int updater, t_to; float lr, rho, b1, b2, a_t; };
// partial-row slots launch_scatter_wide can claim for max_entries sorted entries (the caller's slab has at least as many)
static inline int sbr_scatter_wide_slots(size_t max_entries) { return (int)(max_entries / 64 + max_entries / 65 + 2); }
// key_lo / accumulate: the entries of the keys [key_lo, key_lo + n_ids) of a time-chunked sort, ADDED to dWin
hipError_t launch_scatter_reduce(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos,
                                 const int* offs, int n_ids, int max_entries, int GHp, int Bp, int key_lo = 0, bool accumulate = false, int acc_chunk = 32);
// all time chunks of a time-chunked sort in ONE launch beside the running chain: every wave waits for poll.done to reach the
// time chunk of its entries (tch steps per chunk), rows are added with float atomics (an id may occur in every chunk)
// the same without a global atomic per piece: `units` workgroups own id ranges of equal cost (P: n_ids + 1 ints of workspace, the
// running cost) and accumulate their rows in LDS (sbr_misc.hip).  false = shape not supported, nothing launched.
bool launch_scatter_cost_scan(hipStream_t s, const int* offs, int* P, int n_ids, int n_tchunks, int max_entries, int GHp, int units,
                              hipError_t* err);
hipError_t launch_tail_monitor(hipStream_t s, const SbrPoll& poll, int t_lo);
bool launch_scatter_lds_poll(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs, const int* P,
                             int n_ids, int n_tchunks, int max_entries, int GHp, const SbrPoll& poll, const SbrTChunks& bounds,
                             int units, hipError_t* err, bool monitor = false, int t_lo = 0);      // monitor: one more workgroup, the tail's monitor
hipError_t launch_scatter_reduce_poll(hipStream_t s, float* dWin, const float* dxt, const int* sid, const int* spos, const int* offs,
                                      int n_ids, int n_tchunks, int tch, int max_entries, int GHp, int Bp, const SbrPoll& poll, int first_key = 0, const SbrTChunks* bounds = nullptr, int short_chunks = 0, bool fence_on = false);

// Tile-blocked activation layout [t][row tile of 16][column tile of 16][row 16][col 16] (floats):
// the 16x16 tile one wave of the recurrent kernels owns is one contiguous KiB (8 full 128-B lines per
// wave-wide 16-B access instead of 16 half lines of 16 different rows).  Used for xt (layer 0), the
// saved gate activations, dxt and dhi; hs/cs stay row-major for the GEMM consumers.
__host__ __device__ __forceinline__ size_t sbr_blocked_index(int t, int row, int col, int Bp, int ncols) {
    return (((size_t)t * (Bp >> 4) + (row >> 4)) * (ncols >> 4) + (col >> 4)) * 256 + (row & 15) * 16 + (col & 15);
}

struct RecArgs {
    int cell, T, Bp, H, Hp, G;
    float clip;
    const int* len;         // [Bp]
    const float* xt;        // [T][Bp][G*Hp]
    const float* Whid;      // [Hp][G*Hp]
    const float* peep;      // [3][Hp]
    const float* cinit;     // [Hp]
    const float* hinit;     // [Hp]
    float* hs; float* cs; float* g[4];
    // backward only
    const float* dh_last;   // [Bp][Hp] grad wrt the final hidden state (top layer) or NULL
    const float* dh_slabs;  // rec_bwd_x6p only: dh_last as n_dh_slabs unreduced split-K slabs of [Bp][Hp] (the kernel's prologue
    int n_dh_slabs;         // adds them: one small reduction launch less in front of the BPTT chain), or NULL / 0
    const float* dh_ext;    // [T][Bp][Hp] grad wrt every hid_out[t] (lower layers) or NULL
    int dhe_on;             // rec_bwd_c16 only (set by its launcher): dh_ext is live (else it points at hs and is only requested)
    float* dxt; float* dhi; // dhi: GRU only, compact [T][Bp][Hp] = candidate-gate slice of grad wrt hid_input (r,u slices == dxt)
    // BPTT in time chunks (so the weight-gradient GEMM of finished chunks runs beside the chain): this launch
    // covers t in [t_lo, t_hi); dh/dc cross launches through `state` [2][Bp][Hp]; part block = chunk*nblocks + block
    int t_lo, t_hi, chunk;
    float* state;
    float* part;            // [chunks][nblk][G*Hp + 5*Hp]
    int rpt;                // live batch rows per workgroup of the bf16x6 kernels (16, 8, 4, 2, 1); part[] has Bp/rpt blocks
    int xt_blocked;         // xt is tile-blocked (layer 0: written by the gather) or row-major (GEMM output)
    int x6_split;           // 4-row tiles use the split-gate-math kernels (SBR_X6_SPLIT, default 1)
    int x6_pipe;            // Hp = 128: waves synchronise per k-block through LDS counters, no per-step barrier (SBR_X6_PIPE)
    // layer 0, one index per step: the embedding gather is fused into the forward kernel (xt is never written)
    const int* gX;          // [Bp][T] item ids, or NULL: read xt
    const float* gWin;      // [input_size][G*Hp]
    int n_in;               // rows of W_in (the pipelined kernels address them with 32-bit byte offsets)
    const float* gbias;     // [G*Hp]
    int f32_mfma;           // SBR_FLAG_F32_MFMA: exact-f32 v_mfma_f32_16x16x4_f32 kernels instead of bf16x6
    unsigned long long* prof; // SBR_FLAG_PROFILE_REC: [nblk][waves][4] cycle counters, else NULL
    // cluster kernels (sbr_rec_cl.hip): several workgroups per row tile for layers too wide for one CU
    int cluster;            // allowed (SBR_CLUSTER != 0)
    int cl_linear;          // (experiment, SBR_CL_LINEAR=1) cluster members on consecutive workgroup ids = different XCDs
    int* fault;             // set to 1 when a cluster exchange wait gave up (bounded spin)
    int sentinel_done;      // the backward exchange arrays were already filled with the sentinel (side stream)
    int* clx;               // [tiles][C] start-of-launch handshake: (epoch << 4) | XCC id of every member
    void* xh;               // 16-row cluster kernels: ring [4][Bp/16][2 planes][16][Hp] fp16, the pre-split h exchange
    float* pring;           // ... and the ring of partial-sum blocks of the backward (sbr_rec_c16_ring_floats)
    int epoch;              // unique per launch (clx is never cleared)
    int relu;               // Vanilla layers with dense input = stock lasagne RecurrentLayer: rectify instead of tanh (sbr_cell.h)
    // overlapped step tail (rec_bwd_x6p only): dxt / dhi are stored write-through and every wave publishes
    // (prog_epoch << 12) | t in progress[block * 8 + wave] once all its time steps >= t are complete: at launch (t = first
    // live step + 1 ...), whenever t is a multiple of prog_every, and t_lo at the end.  NULL: plain stores, no progress
    int* progress; int prog_every; int prog_epoch;
    int fence_kb;           // rec_*_x6p: the workgroup claims this much of its CU's LDS (KiB; 0: what it needs) so that kernels which
                            // run BESIDE the chain and use LDS themselves are placed on other CUs (overlapped step tail)
};
#define SBR_CL_ROWS 8       // batch rows per cluster tile
#define SBR_C16_RING 2      // rec_bwd_c16: slots of the ring of partial-sum blocks (blocks carry the lap's parity: sbr_rec_c16.hip)
#define SBR_X6P_FUSE_MAX_T 4096      // fused gather of rec_fwd_x6p: 4 rows x T row offsets in LDS
bool sbr_rec_cluster_ok(const RecArgs& a);
int sbr_rec_cluster_bwd_rows(const RecArgs& a);
hipError_t launch_rec_forward_cl(hipStream_t s, const RecArgs& a);
hipError_t launch_rec_backward_cl(hipStream_t s, const RecArgs& a);
// sbr_rec_p.hip: pipelined bf16x6 kernels for Hp = 128 on 4-row tiles
bool sbr_rec_x6p_ok(const RecArgs& a);
bool sbr_rec_x6p_fuse_ok(const RecArgs& a);   // ... and its forward kernel can gather the W_in rows itself (never part of sbr_rec_x6p_ok)
int sbr_rec_x6p_f16_terms();              // MFMAs per f32 product of the x6p kernels' fp16 forms (2: packed planes, sbr_rec_p.hip)
bool sbr_rec_x6p_tail_ok(const RecArgs& a);   // ... and its backward kernel can publish progress (RecArgs.progress)
hipError_t launch_rec_forward_x6p(hipStream_t s, const RecArgs& a);
hipError_t launch_rec_backward_x6p(hipStream_t s, const RecArgs& a);
// sbr_rec_q.hip: the same step loop for Hp = 32 / 64 (one wave per SIMD)
bool sbr_rec_x6q_ok(const RecArgs& a);
hipError_t launch_rec_forward_x6q(hipStream_t s, const RecArgs& a);
hipError_t launch_rec_backward_x6q(hipStream_t s, const RecArgs& a);
hipError_t sbr_rec_bwd_cl_fill(hipStream_t s, const RecArgs& a);
bool sbr_rec_c16_ok(const RecArgs& a);
size_t sbr_rec_c16_ring_floats(int Bp, int Hp);
hipError_t launch_rec_forward(hipStream_t s, const RecArgs& a, bool simple);
// true when the forward launch for these args can gather its input rows itself (RecArgs.gX/gWin/gbias)
bool sbr_rec_fwd_can_fuse_gather(const RecArgs& a, bool simple);
hipError_t launch_rec_backward(hipStream_t s, const RecArgs& a, bool simple);
// number of part[] blocks the backward launch for these args writes
int sbr_rec_bwd_blocks(const RecArgs& a, bool simple);
// true when the backward launch honours RecArgs.t_lo/t_hi/chunk (bf16x6 kernels)
bool sbr_rec_bwd_chunkable(const RecArgs& a, bool simple);
// sums the per-workgroup partials into bias / peephole / init gradients
hipError_t launch_rec_reduce_partials(hipStream_t s, const float* part, int nblk, int G, int Hp, int cell,
                                      float* db, float* dpeep, float* dcinit, float* dhinit);

// Generic f32 GEMM on v_mfma_f32_16x16x4_f32:  C[m][n] = sum_k A(m,k) * B(k,n) (+ bias[n])
// A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn], C row-major with leading dim ldc.
// ws: split-K workspace of ws_floats floats (may be NULL -> no split).
// a_blk_Bp / b_blk_Bp > 0: that operand is a tile-blocked activation [pos = t*Bp + row][col] (strides ignored;
// A: m = pos, k = col over K columns;  B: k = pos, n = col over N columns).
hipError_t launch_gemm(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn,
                       float* C, long ldc, int M, int N, int K, const float* bias, float* ws, size_t ws_floats,
                       bool simple, int a_blk_Bp = 0, int b_blk_Bp = 0, int* keep_slabs = nullptr);
void sbr_gemm_x6_no_wide(bool on);                       // this thread's next launches stay on the 128-wide tile (parity tests of gemm_x6w_kernel)
void sbr_gemm_hint(int planes, float sa, float sb);      // operand planes of the NEXT launch_gemm call (sbr_gemm.hip)
// keep_slabs != NULL: where the split-K form runs, its slabs stay in ws (slab z at ws + z * M * N, row stride N), the
// reduction is left to the consumer and *keep_slabs = their number; otherwise *keep_slabs = 0 and C holds the result

// bf16x6 GEMM (sbr_gemm_x6.hip): false = shape not supported, use the f32 kernel
bool launch_gemm_x6(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, float* C, long ldc,
                    int M, int N, int K, const float* bias, int nsplit, int kchunk, size_t slab_stride, hipError_t* err,
                    const float* B2 = nullptr, long sbk2 = 0, int n_split = 0, bool small = false, int planes = 3,
                    float sa = 1.0f, float sb = 1.0f,       // planes = 2: fp16 x3 with operand scales sa, sb (powers of two)
                    const SbrPoll* poll = nullptr);
void sbr_gemm_set_exact_f32(bool on);
// planes = 1: the next launch_gemm calls run on plain bf16 operands (one MFMA per block, no split-K), for any number of rows
void sbr_gemm_set_planes(int planes);

// slabs are [z][slab_stride] with row stride ws_ld: a GEMM may fill only a column range of wider slabs
hipError_t launch_gemm_slabs(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M,
                             int N, int K, float* ws, int nsplit, long ws_ld, size_t slab_stride);
// same through the bf16x6 kernel only, with the B columns >= n_split taken from B2 (row stride sbk2); false: not launched
bool launch_gemm_slabs_x6(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M, int N,
                          int K, float* ws, int nsplit, long ws_ld, size_t slab_stride, const float* B2, long sbk2, int n_split,
                          hipError_t* err, int planes = 3, float sa = 1.0f, float sb = 1.0f);
// the same as a consumer of the running chain: poll.n_slabs slabs of K, poll.slab_lo[0 .. n_slabs] (device) their first rows,
// shared by n_groups persistent groups of workgroups, each of which leaves ONE partial in ws (n_groups partials to reduce)
bool launch_gemm_slabs_x6_poll(hipStream_t s, const float* A, long sam, long sak, const float* B, long sbk, long sbn, int M, int N,
                               int K, float* ws, int n_groups, long ws_ld, size_t slab_stride, const float* B2, long sbk2,
                               int n_split, hipError_t* err, int planes, float sa, float sb, const SbrPoll& poll);
// host side of the table: lo[0] = 0 < lo[1] < ... < lo[n] = K, n <= cap.  A slab that starts at time step t has
// max(1, floor(growth * t - 1)) k steps of 32 rows, at most max_rows / 32 (sizes scaled up until n <= cap).
int sbr_tail_slab_table(int K, int rows_per_step, int cap, double growth, int max_rows, std::vector<int>& lo);
hipError_t launch_splitk_reduce(hipStream_t s, const float* ws, int nslabs, int M, int N, float* C, long ldc,
                                const float* bias);

// dedicated dW_hid kernel: nslices slabs [Hp][GHp]; dhc != NULL: GRU compact candidate-gate array for cols >= 2*Hp
bool launch_wgrad_slabs(hipStream_t s, const float* hs, const float* dxt, const float* dhc, float* slabs, int Hp, int GHp,
                        int npos, int nslices, hipError_t* err);

// sbr_head.hip: logits + softmax / CCE + dh of a full-softmax head in one launch (C1 / C2-class catalogues); false: not served
bool sbr_head_plan(int Bp, int N, int Hp, int* CC, int* CW, size_t* lds_bytes);
// the sampled head in one launch (sbr_head.hip: head_sampled_kernel); false: shape not served, nothing launched
bool launch_head_sampled(hipStream_t s, const float* h, const float* Wc, const float* bc, const float* pop, float* act, float* rowcost,
                         float* dh, int rows, int C, int Hp, int Bg, int S, int row_offset, int loss, int Bglobal, hipError_t* err,
                         unsigned long long* prof = nullptr);
bool launch_head_cce(hipStream_t s, const float* h, const float* WoutT, const float* bout, const int* tgt, const float* pop, float* dlogits,
                     float* rowcost, float* slabs, size_t slab_floats, unsigned* stats, int* fault, int Bp, int N, int Nl, int Hp, int Bglobal,
                     unsigned epoch, int* n_slabs, hipError_t* err, unsigned long long* prof = nullptr);
// full softmax + categorical cross-entropy (rnn_one_hot.py:65-77): logits (rows,N), row stride ld, in; dlogits out in place
hipError_t launch_softmax_cce(hipStream_t s, float* logits, const float* bout, const int* target, const float* pop,
                              float* rowcost, int rows, int N, long ld, int Bglobal);
hipError_t launch_softmax_rows(hipStream_t s, float* logits, const float* bout, int rows, int N, int do_softmax);
// RNNMargin (rnn_margin.py:62-69, :112-147): logits (rows, N) raw h.W_out in, d cost / d logits out in place; target [rows][NT]
// (-1 = none), X / len: the rows' input items (weight 0 and target 0 on them when unique)
hipError_t launch_margin_loss(hipStream_t s, float* logits, const float* bout, const int* target, int NT, const int* X,
                              const int* len, int T, int F, const float* dflt, float* rowcost, int rows, int N, long ld,
                              int Bglobal, int loss, float balance, int unique);
// db[n] = sum_rows d[r][n] + reg term ; cost += reg term
hipError_t launch_colsum_bias(hipStream_t s, const float* d, int rows, int N, long ld, float* db, const float* b,
                              float reg, float* cost, float* ws /* >= 16*N floats */);
hipError_t launch_sum_cost(hipStream_t s, const float* rowcost, int rows, float* cost);
// sampled heads (sparse_lstm.py:42-54, rnn_sampling.py:68-91,137)
hipError_t launch_build_cells(hipStream_t s, const int* target, const int* samples, int Bg, int S, int* cells);
hipError_t launch_gather_rows(hipStream_t s, const float* W, const float* b, const int* cells, int C, int Hp,
                              float* Wc, float* bc);
hipError_t launch_sampled_loss(hipStream_t s, float* act, const float* bc, const float* pop, float* rowcost, int rows,
                               int Bg, int S, int row_offset, int loss, int Bglobal);
hipError_t launch_scatter_cells(hipStream_t s, float* dW, float* db, const float* dWc, const float* dbc,
                                const int* cells, int C, int Hp);
// optimizers (lasagne.updates.* [3P], update_manager.py:24-82)
// n elements starting at p / g / s0 / s1, skipping gap_len elements after the first gap_at (two ranges, one launch)
// the output layer's gradient + its step in one launch (sbr_misc.hip); false: shape not served
bool launch_out_grad_step(hipStream_t s, const float* dlogits, const float* h_last, const float* rowcost, float* cost, int updater,
                          float* W, float* Ws0, float* Ws1, float* b, float* bs0, float* bs1, int R, int N, int Nl, int Hp, float lr,
                          float rho, float b1, float b2, long t, hipError_t* err);
hipError_t launch_update_rows_aware(hipStream_t s, int updater, float* p, float* g, float* s0, float* s1, int n_rows, int row_floats,
                                    const int* offs, float lr, float rho, float b1, float b2, long t);
hipError_t launch_update(hipStream_t s, int updater, float* p, float* g, float* s0, float* s1, size_t n,
                         float lr, float rho, float b1, float b2, long t, size_t gap_at = (size_t)-1, size_t gap_len = 0);
// ... of a block whose gradient is still split-K slabs [nslabs][n] (16-byte aligned, n % 4 == 0): reduction + step in one launch
hipError_t launch_update_from_slabs(hipStream_t s, int updater, const float* ws, int nslabs, float* p, float* s0, float* s1, size_t n,
                                    float lr, float rho, float b1, float b2, long t);
// sbr_sparse.hip: row-sparse optimizer steps (lazy catch-up) and gradient exchange for the large-catalogue blocks
struct SbrSparseRows { int npairs; size_t off[2]; int width[2]; int stride[2]; int n_rows; float *p, *g, *s0, *s1; int* last; };
struct SbrSparseUpd { int updater; float lr, rho, b1, b2; const float* at; int n_at; int early_exit; };
hipError_t launch_sparse_catch_up_batch(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, const int* X, const int* len, int T,
                                        int Bp, int F, int t_to);
hipError_t launch_sparse_catch_up_list(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, const int* list, const int* n_dev,
                                       int n_host, int n_max, int t_to);
hipError_t launch_sparse_flush(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, int t_to);
hipError_t launch_sparse_step_list(hipStream_t s, const SbrSparseRows& r, const SbrSparseUpd& c, const int* list, const int* n_dev,
                                   int n_host, int n_max, int t_to);
hipError_t launch_sparse_pack(hipStream_t s, const SbrSparseRows& r, const int* list, const int* n_dev, int n_host, int n_max, int* mark,
                              int epoch, int* ids_out, float* rows_out, int W, int* count);
hipError_t launch_sparse_unpack_add(hipStream_t s, const SbrSparseRows& r, const int* ids, const float* rows, int n, int W, int* cand);
hipError_t launch_sparse_unpack_add_dev(hipStream_t s, const SbrSparseRows& r, const int* ids, const float* rows, int cap, int W, int* cand);
// top-k (rnn_base.py:196-211)
// value: -inf (top_k_recommendations, rnn_base.py:154-155) or 0 (the compiled test function's scores * (1 - exclude), :201-202)
hipError_t launch_exclude_seen(hipStream_t s, float* scores, const int* X, const int* len, int rows, int T, int F,
                               int N, float value);
hipError_t launch_topk(hipStream_t s, float* scores, int rows, int N, int k, int* ids);
hipError_t launch_fill(hipStream_t s, float* p, float v, size_t n);
