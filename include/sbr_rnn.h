/* sbr_rnn.h -- C-ABI of libsbr_rnn.so: the MI355X (gfx950) engine for the
 * `train.py -m RNN` training hot path of rdevooght/sequence-based-recommendations.
 *
 * The reference has no FFI for this path: its seam is three Theano-compiled callables
 * plus a parameter list, all owned by RNNBase (neural_networks/rnn_base.py):
 *     train_function(*theano_inputs) -> cost        built :185, called :290
 *     test_function(theano_inputs, k) -> ids[k]     built :196-211, called :361
 *     predict_function(X, mask) -> scores (1,N)     built :188-194, called :151
 *     lasagne.layers.get/set_all_param_values       :476, :515
 * Each entry point below cites the reference interface it replaces.  Plain pointers and
 * sizes only; no torch types.  All functions return 0 on success or a negative
 * sbr_status; sbr_last_error() gives the message (thread-local).  Nothing throws across
 * the ABI.  A handle is not thread-safe (one stream, one host thread per rank), which
 * matches the reference's single-threaded synchronous calls (rnn_base.py:285-300).
 *
 * Memory: the caller may hand in a device arena (e.g. a torch tensor's data_ptr(), so
 * that torch.distributed/RCCL can all-reduce the gradient section in place); with
 * arena == NULL the library hipMallocs its own.
 */
#ifndef SBR_RNN_H
#define SBR_RNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBR_MAX_LAYERS 4
#define SBR_ABI_VERSION 10

typedef enum { SBR_OK = 0, SBR_EINVAL = -1, SBR_ENOMEM = -2, SBR_EHIP = -3, SBR_ESTATE = -4,
               SBR_EUNSUPPORTED = -5 } sbr_status;

/* --r_t (recurrent_layers.py:9) */
typedef enum { SBR_CELL_LSTM = 0, SBR_CELL_GRU = 1, SBR_CELL_VANILLA = 2 } sbr_cell;
/* --loss (command_parser.py:43, :116-121) */
typedef enum { SBR_LOSS_CCE = 0, SBR_LOSS_BLACKOUT = 1, SBR_LOSS_BPR = 2, SBR_LOSS_TOP1 = 3,
               /* RNNMargin (rnn_margin.py:62-69; command_parser.py:118-119): linear output layer, multi-target losses */
               SBR_LOSS_HINGE = 4, SBR_LOSS_LOGIT = 5, SBR_LOSS_LOGSIG = 6,
               /* RNNCluster's further sampled losses (rnn_cluster.py:158-175; `--clusters C` with --loss CCE | BPRelu | lin):
                * cross-entropy over the sampled columns, leaky hinge on the score differences, plain differences */
               SBR_LOSS_SCCE = 7, SBR_LOSS_BPRELU = 8, SBR_LOSS_LIN = 9 } sbr_loss;
#define SBR_LOSS_IS_MARGIN(l) ((l) >= SBR_LOSS_HINGE && (l) <= SBR_LOSS_LOGSIG)
/* --u_m (update_manager.py:4) */
typedef enum { SBR_UPD_ADAGRAD = 0, SBR_UPD_ADADELTA = 1, SBR_UPD_RMSPROP = 2, SBR_UPD_NESTEROV = 3,
               SBR_UPD_ADAM = 4 } sbr_updater;

/* Everything RNNBase.__init__/prepare_model fixes (rnn_base.py:61-109, rnn_one_hot.py:37-78,
 * rnn_sampling.py:93-137, recurrent_layers.py:18-26, update_manager.py:24-82). */
typedef struct sbr_config {
    int32_t abi_version;            /* SBR_ABI_VERSION */
    int32_t cell;                   /* sbr_cell */
    int32_t n_layers;               /* --r_l "100-50" -> 2 */
    int32_t layers[SBR_MAX_LAYERS]; /* hidden units per layer */
    int32_t n_items;                /* N: output units (data/stats, data_handling.py:94) */
    int32_t input_size;             /* rows of layer-0 W_in = N + n_optional_features (rnn_one_hot.py:48-49) */
    int32_t n_feat;                 /* F indices per step: 1, or 2 with --rf (rnn_base.py:615-642) */
    int32_t max_length;             /* T (--max_length) */
    int32_t batch_size;             /* global batch B: the mean in the cost (rnn_one_hot.py:71) */
    int32_t local_batch;            /* rows this rank holds per step (== batch_size on one GPU) */
    int32_t row_offset;             /* first global row of this rank (sampled heads: positive column) */
    int32_t loss;                   /* sbr_loss */
    int32_t n_samples;              /* S = effective_sampling (rnn_sampling.py:102-106); 0 for CCE */
    int32_t updater;                /* sbr_updater */
    float learning_rate;            /* --u_l */
    float rho;                      /* --u_rho (adadelta/rmsprop rho, nesterov momentum) */
    float beta1, beta2;             /* --u_b1 --u_b2 */
    float regularization;           /* -r: >0 L2, <0 L1, on b_out only (rnn_one_hot.py:73-77) */
    float grad_clip;                /* 100 (recurrent_layers.py:19); <=0 disables */
    int32_t flags;                  /* SBR_FLAG_* */
    int32_t embedding_size;         /* --r_emb E (recurrent_layers.py:46-50): EmbeddingLayer (input_size, E) + flatten in front of
                                     * DENSE recurrent layers (layer 0 input = n_feat * E); 0 = index-input layer 0 */
    int32_t bidirectional;          /* --r_bi (recurrent_layers.py:70-76): every level = a forward and a backwards layer over the same
                                     * input, concatenated on the feature axis (next level / output layer see 2*H features) */
    /* RNNMargin only (losses hinge / logit / logsig; rnn_margin.py:32-51, :112-147) */
    float balance;                  /* --balance: weight of a false positive = balance * n_pos / (N - n_pos - n_in) */
    int32_t n_targets;              /* --n_targets: positives per row at most (columns of `target` in sbr_set_batch, -1 = none) */
    int32_t unique;                 /* interactions_are_unique (not --repeated_interactions): a row's input items get weight 0, target 0 */
} sbr_config;

#define SBR_FLAG_SIMPLE_REC  1   /* triage: per-step VALU recurrent kernels instead of the MFMA persistent ones */
#define SBR_FLAG_SIMPLE_GEMM 2   /* triage: naive GEMM instead of the MFMA tiled one */
#define SBR_FLAG_F32_MFMA 16      /* recurrent GEMM on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) instead of the bf16x6 split */
#define SBR_FLAG_PROFILE_REC 8    /* recurrent kernels record s_memtime / s_memrealtime phase counters ("prof" debug buffer) */
#define SBR_FLAG_ATOMIC_SCATTER 4 /* triage: per-element float atomics instead of the sorted segment reduce */
/* Row-sparse optimizer (see "Row-sparse blocks" below): by default a block is stepped row by row when a step cannot touch
 * every row of it anyway (more rows than the batch has item ids / more items than sampled cells). */
#define SBR_FLAG_SPARSE_UPDATE 32 /* always, for every block that exists (tests) */
#define SBR_FLAG_DENSE_UPDATE 64  /* never: one dense elementwise pass over every parameter, as lasagne.updates.* does */
/* Output projection h . W_out (DenseLayer rnn_one_hot.py:65 / BlackoutLayer's deterministic branch sparse_lstm.py:37-40) on
 * plain bf16 operands with f32 accumulation (one v_mfma_f32_16x16x32_bf16 per block): the training forward of the full
 * softmax and predict / top-k of every head.  Scores then carry bf16 input rounding (~3e-3 of their spread) instead of
 * float32 rounding; gradients still come from the float32-class kernels. */
#define SBR_FLAG_BF16_PROJECTION 128
/* The dense GEMMs between stacked recurrent layers (recurrent_layers.py:94-104: layer l >= 2 reads the hidden states of layer
 * l - 1 through a dense W_in; its backward pair dW_in = h^T . dxt and dh = dxt . W_in^T) on plain bf16 operands with f32
 * accumulation, one MFMA per product (BASELINE configs[4]: "bf16 MFMA output projection (and bf16 layer-2 input GEMM)").  Default:
 * the f32-class two-plane fp16 split (three MFMAs).  Gradients then carry bf16 input rounding (~4e-3 relative per product). */
#define SBR_FLAG_BF16_LAYERS 256

typedef struct sbr_handle sbr_handle;

const char* sbr_last_error(void);
int sbr_abi_version(void);

/* Bytes of device arena a handle for this config needs (params + grads + optimizer state +
 * activations saved for BPTT + workspaces). */
int sbr_arena_bytes(const sbr_config* cfg, size_t* bytes);

/* Replaces RNNBase.prepare_model + _compile_*_function (rnn_base.py:106-109, :175-213).
 * arena: device pointer of >= sbr_arena_bytes() bytes, 256-B aligned, or NULL (library
 * allocates).  stream: hipStream_t every launch goes to (NULL = default stream).
 * Parameters start as zeros: call sbr_set_params. */
int sbr_create(const sbr_config* cfg, void* arena, size_t arena_bytes, void* stream, sbr_handle** out);
void sbr_destroy(sbr_handle* h);

/* lasagne.layers.get_all_param_values / set_all_param_values (rnn_base.py:476, :515):
 * n arrays in Lasagne order and shape (float32, C-contiguous, host memory); LSTM layer:
 * 12 gate arrays, 3 peepholes, cell_init, hid_init; GRU: 9 gate arrays, hid_init; then
 * out.W (H,N), out.b (N,)  (sparse_lstm.py:240-279, :660-676). */
int sbr_num_params(const sbr_handle* h);
int sbr_param_shape(const sbr_handle* h, int i, int64_t dims[2], int* ndim);
/* Name and shape of array i of that list for a configuration -- host only, no device or handle needed (tooling: which
 * arrays are weights / biases / initial states, for the init laws of the Lasagne layers).  "l0.W_in_to_ingate", "emb.W",
 * "out.W" ...; a Vanilla layer with dense input is the stock RecurrentLayer and lists hid_init, input_to_hidden.W,
 * input_to_hidden.b, hidden_to_hidden.W (recurrent_layers.py:94-104).  i past the end: SBR_EINVAL. */
int sbr_describe_param(const sbr_config* cfg, int i, char* name, size_t name_cap, int64_t dims[2], int* ndim);
int sbr_set_params(sbr_handle* h, int n, const float* const* host_arrays);
int sbr_get_params(sbr_handle* h, int n, float* const* host_arrays);
/* Same layout, gradients of the last forward/backward (parity tests). */
int sbr_get_grads(sbr_handle* h, int n, float* const* host_arrays);

/* Flat device sections (float32).  The gradient section carries one extra trailing float:
 * the batch cost, so that one all-reduce sums gradients and cost across ranks.
 * which: 0 params, 1 grads(+cost), 2 optimizer state.  split_floats (grads only): offset of
 * the output-layer gradients, which are complete after sbr_loss_backward_output and can be
 * all-reduced while sbr_backward_recurrent runs. */
int sbr_section(sbr_handle* h, int which, void** dev_ptr, size_t* n_floats, size_t* split_floats);

/* RNNMargin, --pb: the default target of every item (RNNMargin._default_target, rnn_margin.py:149-161: min(1 - p_i,
 * (1 - min_access) p_i / min_access)), N floats on the host; without a call (or with NULL) it is 0 everywhere. */
int sbr_set_default_target(sbr_handle* h, const float* default_target);

/* The per-call inputs of train_function (rnn_one_hot.py:61,106; rnn_sampling.py:128,194):
 * X int32 (B,T,F) left-aligned; lengths int32 (B,) = mask.sum(1) (masks are prefix masks,
 * rnn_one_hot.py:100-101); target int32 (Bg,) -- all GLOBAL rows for the sampled heads
 * (Blackout's softmax spans every target, rnn_sampling.py:68-72,137), local rows for CCE;
 * samples int32 (S,) or NULL; target_popularity float (B,) local rows (pop**db,
 * rnn_one_hot.py:103).  `exclude` (B,N) is never materialised (unused by train_function,
 * rnn_base.py:185 on_unused_input='ignore').  on_device != 0: pointers are device memory. */
int sbr_set_batch(sbr_handle* h, const int32_t* X, const int32_t* lengths, const int32_t* target,
                  const int32_t* samples, const float* target_popularity, int n_rows, int on_device);

/* train_function(*batch) -> cost (rnn_base.py:290): zero grads, forward, loss, backward,
 * update.  cost_host may be NULL (cost stays on device: sbr_read_cost). */
int sbr_train_step(sbr_handle* h, float* cost_host);
/* The training loop's form of the same call (rnn_base.py:285-300 reads the cost of every iteration): enqueues the step
 * and a copy of its cost, and hands back the cost of the PREVIOUS call once that is ready, so the host never waits for
 * the step it has just enqueued.  *have_prev = 0 on the first call.  sbr_lagged_flush waits for and returns the cost of
 * the last enqueued step (call it before reading parameters, validating, or leaving the loop). */
int sbr_train_step_lagged(sbr_handle* h, float* prev_cost, int* have_prev);
int sbr_lagged_flush(sbr_handle* h, float* cost, int* have);
/* The same in phases (data-parallel: all-reduce the gradient section between them). */
int sbr_zero_grads(sbr_handle* h);
int sbr_forward(sbr_handle* h);
int sbr_loss_backward_output(sbr_handle* h);
int sbr_backward_recurrent(sbr_handle* h);
int sbr_apply_update(sbr_handle* h);
int sbr_read_cost(sbr_handle* h, float* cost_host);
/* The phases put everything that only feeds the optimizer (output-layer weight/bias gradients, the cost, the
 * weight-gradient kernels) on an internal side stream.  Called one by one they join it at their end so that the
 * caller may read the gradients on `stream`.  A data-parallel driver that orders its collectives itself turns the
 * joins off: after sbr_loss_backward_output the output-layer slice is complete ON THE SIDE STREAM (sbr_query
 * "side_stream" returns it) -- all-reduce it from there while `stream` runs the BPTT chain -- and calls
 * sbr_join_side before touching the rest.  sbr_apply_update always joins. */
int sbr_set_deferred_join(sbr_handle* h, int on);
int sbr_join_side(sbr_handle* h);

/* ------------------------------------------------------------------------------------------------
 * Row-sparse blocks.  The reference's updates are dense over every parameter (update_manager.py:24-82, lasagne.updates.*);
 * only the rows a batch gathers (sparse_lstm.py:368) and the sampled cells (sparse_lstm.py:50-54) receive a non-zero
 * gradient.  Block kinds: the index-addressed rows of layer 0 (W_in of both directions, or the --r_emb table), and -- for
 * the sampled heads -- the rows of W_out / b_out.  A sparse block is stepped row by row over the touched rows only:
 * exact for adagrad (its zero-gradient step is a no-op); for rmsprop / adadelta / nesterov / adam the zero-gradient steps
 * a row missed are replayed when the row is next read or stepped ("lazy-exact": same results as the dense pass to
 * float32 rounding).  sbr_get_params, sbr_section(0 / 2), sbr_predict_scores and sbr_topk bring everything they read up to
 * date themselves; sbr_flush_lazy does it explicitly (e.g. before reading the parameter section through a pointer
 * obtained earlier). */
int sbr_flush_lazy(sbr_handle* h);
/* Data-parallel exchange of a sparse block b (0 .. sbr_query("sparse_blocks") - 1) instead of an all-reduce of the whole
 * block: after sbr_backward_recurrent,
 *   sbr_sparse_pack        moves this rank's touched gradient rows into ids_dev [max_local_rows] / rows_dev
 *                          [max_local_rows][row_floats] (device buffers of the caller, e.g. torch tensors handed to RCCL)
 *                          and returns their number (synchronises the stream);
 *   (the ranks all-gather ids and rows)
 *   sbr_sparse_unpack_add  adds ONE rank's rows into the gradient block; call it for every rank, own rows included, in
 *                          rank order on every rank, so that the replicas stay bit-identical;
 * sbr_apply_update then steps the union of the gathered rows.  sbr_dense_ranges lists the [lo, hi) float ranges of the
 * gradient section (trailing cost included) that still take the dense all-reduce. */
int sbr_sparse_info(sbr_handle* h, int b, int64_t* n_rows, int64_t* row_floats, int64_t* max_local_rows);
int sbr_sparse_pack(sbr_handle* h, int b, int32_t* ids_dev, float* rows_dev, int32_t* count_host);
int sbr_sparse_unpack_add(sbr_handle* h, int b, const int32_t* ids_dev, const float* rows_dev, int count);
int sbr_dense_ranges(sbr_handle* h, int cap, int64_t* lo, int64_t* hi, int* n);
/* The same exchange without a host round trip (ABI 7): the number of packed rows stays on the device and travels IN BAND.
 *   sbr_sparse_pack_device     ids_dev is [1 + max_local_rows] ints: ids_dev[0] = the row count, ids_dev[1 ..] the ids;
 *                              rows_dev [max_local_rows][row_floats].  Nothing is synchronised.
 *   (the ranks all-gather both buffers at their fixed capacity: ids_all [world][1 + max_local_rows], rows_all
 *    [world][max_local_rows][row_floats])
 *   sbr_sparse_unpack_add_all  adds every rank's rows in rank order (one launch per rank, its count read on the device).
 * A fixed-capacity all-gather moves max_local_rows rows per rank whatever the step touched: the caller chooses per block
 * between this form (small blocks: the host read is the cost) and the counted one above (large blocks: the bytes are). */
int sbr_sparse_pack_device(sbr_handle* h, int b, int32_t* ids_dev, float* rows_dev);
int sbr_sparse_unpack_add_all(sbr_handle* h, int b, const int32_t* ids_all, const float* rows_all, int world);

/* ------------------------------------------------------------------------------------------------
 * The cluster head of RNNCluster (`--clusters C`; rnn_cluster.py:237-256 the cluster loss, :275-300 its training and the hard
 * clusters, :327-352 the test function).  A second object beside the engine: it owns the cluster-selection weights Wc (H, C)
 * and the item / cluster repartition R (N, C), reads the user representation (the engine's final hidden state: device pointer,
 * row stride and the offset of the backwards half from sbr_debug_buffer("h_last")) and trains only its own two arrays, with its
 * own updater state and step count, as the reference's second `self.updater(...)` call does.  The recurrent network's sampled
 * head and loss are the engine's (losses SBR_LOSS_BLACKOUT .. SBR_LOSS_LIN).  Single rank (the head is not sharded). */
typedef enum { SBR_CLUSTER_MIX = 0, SBR_CLUSTER_SOFTMAX = 1, SBR_CLUSTER_SIGMOID = 2 } sbr_cluster_type;   /* --cluster_type */
typedef struct sbr_cluster_config {
    int32_t abi_version;            /* SBR_ABI_VERSION */
    int32_t n_items;                /* N */
    int32_t n_hidden;               /* H: features of the user representation (2 x the top layer's width with --r_bi) */
    int32_t hidden_split;           /* features [0, hidden_split) sit at columns [0, ..) of a row, the rest at off2 + (k - hidden_split) */
    int32_t n_clusters;             /* --clusters */
    int32_t cluster_type;           /* sbr_cluster_type */
    int32_t loss;                   /* sbr_loss: Blackout / BPR / TOP1 / SCCE (--loss CCE) / BPRelu / lin (rnn_cluster.py:88-101) */
    int32_t batch_size;             /* B rows per step: row b's positive is column b of the score matrix */
    int32_t max_samples;            /* most cluster samples per step (--c_sampling, or --sampling when unset) */
    int32_t updater;                /* sbr_updater */
    float learning_rate, rho, beta1, beta2;
    float scale;                    /* --init_scale (sbr_cluster_set_scale follows --scale_growing_rate) */
    float noise_std;                /* --csn: std of the gaussian noise on the selection activations in training (a counter-based
                                     * generator here, MRG_RandomStreams there: equal in law, not in stream) */
    uint64_t seed;
} sbr_cluster_config;
typedef struct sbr_cluster sbr_cluster;
sbr_cluster* sbr_cluster_create(const sbr_cluster_config* cfg, void* hip_stream);
void sbr_cluster_destroy(sbr_cluster* c);
int sbr_cluster_set_params(sbr_cluster* c, const float* R_host, const float* Wc_host);          /* [N][C], [H][C] */
int sbr_cluster_get_params(sbr_cluster* c, float* R_host, float* Wc_host);
int sbr_cluster_get_grads(sbr_cluster* c, float* dR_host, float* dWc_host);                      /* of the last forward_backward */
int sbr_cluster_set_scale(sbr_cluster* c, float scale);                                          /* T_scale.set_value, rnn_cluster.py:400 */
/* cost_clusters and its gradients for the B rows of h_dev (row stride ld_h floats) with targets_dev [B] and samples_dev
 * [n_samples]; cost_host may be NULL (nothing is synchronised then) */
int sbr_cluster_forward_backward(sbr_cluster* c, const float* h_dev, int ld_h, int off2, const int32_t* targets_dev,
                                 const int32_t* samples_dev, int n_samples, float* cost_host);
int sbr_cluster_apply_update(sbr_cluster* c);
/* test path: csel_dev[r] = argmax of the selection activations of row r (z_dev [rows][C] receives them when not NULL) */
int sbr_cluster_select(sbr_cluster* c, const float* h_dev, int ld_h, int off2, int rows, int32_t* csel_dev, float* z_dev);
/* scores_dev[r][n] *= hard[n][csel[r]] (NULL: skip), n_used_dev[r] = sum_n hard[n][csel[r]] (NULL: skip) */
int sbr_cluster_mask_scores(sbr_cluster* c, float* scores_dev, int ld, int rows, const int32_t* csel_dev, float* n_used_dev);
int sbr_cluster_hard(sbr_cluster* c, float* hard_host);                                          /* [N][C] = _get_hard_clusters() */

/* predict_function(X, mask) (rnn_base.py:188-194) on the current batch: scores (rows,N);
 * softmax probabilities for CCE (DenseLayer softmax, rnn_one_hot.py:65), raw activations
 * for the sampled heads (sparse_lstm.py:37-40).  probs != 0 forces softmax (test function of
 * the sampled heads, rnn_sampling.py:144).  out_host (rows*N floats) may be NULL. */
int sbr_predict_scores(sbr_handle* h, int probs, float* out_host);
/* test_function(theano_inputs, k) (rnn_base.py:196-211): ordered top-k ids per row of
 * softmax * (1 - exclude) where exclude = the row's own input items when exclude_seen
 * (interactions_are_unique, rnn_base.py:200-201).  Ties break to the lowest id.  A row with fewer than k rankable items
 * (more than N - k items excluded, or NaN scores) gets -1 in the places it cannot fill. */
int sbr_topk(sbr_handle* h, int k, int exclude_seen, int32_t* ids_host);

/* The dense GEMM of the hot path on caller-provided DEVICE buffers (parity tests of the kernels themselves):
 * C[m][n] = sum_k A[m*sam + k*sak] * B[k*sbk + n*sbn] (+ bias[n]); ws: split-K workspace (may be NULL).
 * exact_f32 = 1: v_mfma_f32_16x16x4_f32 kernel; 0: bf16x6 kernel where the shape allows it; 2: plain bf16 operands (the
 * SBR_FLAG_BF16_PROJECTION kernel); 3: the two-plane fp16 split, three MFMAs per product (operands within fp16's range: what the
 * step's logits / layer GEMMs take since ABI 8); 4 / 5: as 3 / 2 but kept on the 128-wide tile where the library would pick the
 * 256 x 128 one (gemm_x6w_kernel; the two give the same bits: tests/test_gpu_gemm.py). */
int sbr_debug_gemm(void* stream, const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                   float* C, int64_t ldc, int32_t M, int32_t N, int32_t K, const float* bias, float* ws, size_t ws_floats,
                   int32_t exact_f32);

/* Named device buffers for parity tests: "h_last" (rows,Hp), "logits", "xt0", "hs0" ... */
int sbr_debug_buffer(sbr_handle* h, const char* name, void** dev_ptr, size_t* n_floats);
/* Copy n_floats from a device pointer to host (tests without torch). */
int sbr_copy_to_host(sbr_handle* h, const void* dev_ptr, float* host, size_t n_floats);
int sbr_synchronize(sbr_handle* h);

/* Per-phase device time of the train steps since sbr_enable_timing, in microseconds (hipEvents on the
 * handle's stream): gather, rec_fwd, output, rec_bwd, wgrad, scatter, update, total.
 * on = 0: off; 1: every phase (eight event records per step: each costs the stream a few microseconds, ~40 us per
 * C2 step); 2 + p: only phase p (two records; the others read 0) -- what bench.py uses inside its timed region. */
#define SBR_N_PHASES 8
int sbr_enable_timing(sbr_handle* h, int on);
/* Which kernels this handle's shapes select (tooling: bench.py names the kernels it prices):
 * "fused_gather" (layer-0 input rows gathered inside the forward kernel), "rows_per_workgroup",
 * "cluster" (multi-workgroup recurrent kernels for the top layer), "rec_kernel" (kernel family of the top layer:
 * 0 triage, 1 cluster, 2 counter-synchronised 128-unit, 3 counter-synchronised 32/64-unit, 4 barrier / general),
 * "arena_bytes", "side_stream" (hipStream_t), "sparse_blocks" (row-sparse parameter blocks of this configuration),
 * "adam_table" (entries of the a_t table of the lazy Adam catch-up); what the top layer's recurrent kernels put on the
 * matrix pipe, for roofline reports: "rec_products_fwd" / "rec_products_bwd" (low-precision MFMA terms per f32 product:
 * 6 = bf16x6, 3 = fp16x3, 0 = exact-f32 MFMA kernels), "rec_rows_fwd" / "_bwd" (live batch rows among the 16 columns of an
 * MFMA tile), "rec_workgroups_fwd" / "_bwd" (workgroups of the launch = CUs it can occupy); ABI 9: "head_fused" (column chunks of
 * the one-launch full-softmax head a full batch of a training step takes, 0 = the three launches: rnn_one_hot.py:65-71 forward +
 * backward), "scatter_step" (does the scatter-add of a single-call step apply the optimizer to layer 0's index-input rows itself:
 * 0 no, 1 dense block, 2 row-sparse block -- SBR_SCAT_FUSE). */
int sbr_query(sbr_handle* h, const char* what, int64_t* value);
int sbr_phase_times(sbr_handle* h, float us[SBR_N_PHASES]);
/* Chain-only timing (ABI 8; tooling: bench.py prices the recurrent chain kernels of stacked layers apart from the dense GEMMs
 * between them).  on = 1: from now on every launch of a recurrent chain kernel -- the compiled scan of sparse_lstm.py:425 /
 * recurrent_layers.py:57-68 and its gradient, any layer, either direction -- is bracketed by a HIP-event pair (up to 256 launches);
 * on = 0: stop and report us[0] / us[1] = device time summed over the forward / backward chain launches since the start,
 * n[0] / n[1] = the number of launches (us, n may be NULL with on = 1).  A survey facility: the records delay the stream. */
int sbr_chain_times(sbr_handle* h, int on, float us[2], int n[2]);
/* The scatter-add of the embedding gradient ON ITS OWN (ABI 9; tooling: bench.py's `kernels.scatter_unfused`, north_star "achieved
 * HBM GB/s on the embedding gather/scatter"): the gradient of sparse_lstm.py:368's gather (AdvancedIncSubtensor), layer 0's
 * dW_in[X[b][t][f]][:] += dxt[t][b][:] over the current batch, as the step's stand-alone form for this shape runs it -- counting
 * sort of the ids once, then `reps` launches of the scatter-add on the engine's stream between two HIP events, nothing beside
 * them; *us = mean microseconds per launch, *entries = valid (t, b, f) positions, *rows = distinct ids (gradient rows written).
 * dxt holds whatever the last backward pass left; the gradient block is cleared again afterwards.  Index-input layer 0 only. */
int sbr_debug_scatter(sbr_handle* h, int reps, float* us, int64_t* entries, int64_t* rows);
/* A FOREIGN kernel that holds part of the chip (ABI 10; test hook): `workgroups` workgroups of 256 threads, each claiming `lds_kb`
 * KiB of LDS, spin for `milliseconds` on a stream of the library's own (not the engine's) -- asynchronous, returns at once.  What the
 * tests use to show that a cluster chain whose workgroups cannot all be resident (rnn_base.py has no such notion: one process, one
 * Theano function) ends in the fault code of sbr_read_cost / sbr_train_step -- never in a wrong gradient -- and that the one-launch
 * head recomputes what it cannot wait for.  sbr_synchronize(h) does NOT wait for it; sbr_debug_occupy(h, 0, 0, 0) does. */
int sbr_debug_occupy(sbr_handle* h, int workgroups, int lds_kb, int milliseconds);

/* ------------------------------------------------------------------------------------------------
 * Native batch builder: replaces SequenceGenerator + _gen_mini_batch + _prepare_input
 * (data_handling.py:126-174, rnn_base.py:373-420, rnn_one_hot.py:83-106, rnn_sampling.py:159-194) for
 * the default training options (one item index per step, next-item target, no sequence noise).
 * The training sequences live on the device in CSR form; a pass over the users is PLANNED on the host
 * exactly as the reference fills its batches (users in file or shuffled order; user u contributes
 * k = min(B - j, len(u) - 2) rows, a user that overflows the batch is truncated, rnn_base.py:400-415),
 * and each batch is then BUILT by two kernels: k distinct sorted split points l in [2, len) per user
 * (uniform without replacement = random.sample, :402), then per row the window items[max(0,l-T):l], its
 * length, the target items[l], pop[target]**db and (sampled heads) S negatives, uniform or by inverse CDF
 * of popularity**sampling_bias.  Same distribution as the reference, not the same random stream. */
typedef struct sbr_dataset sbr_dataset;

/* items: concatenated item ids of all training sequences; offsets: n_users+1 prefix offsets (host arrays). */
int sbr_dataset_create(const int32_t* items, const int64_t* offsets, int64_t n_users, int32_t n_items, void* stream,
                       sbr_dataset** out);
int sbr_dataset_destroy(sbr_dataset* d);
/* pop_db[n_items] = item_popularity ** diversity_bias (rnn_one_hot.py:103; NULL: all 1);
 * sample_cdf[n_items] = cumsum(item_popularity ** sampling_bias) (rnn_sampling.py:159-163; NULL: uniform). */
int sbr_dataset_set_tables(sbr_dataset* d, const float* pop_db, const double* sample_cdf);
/* Options that shape a batch beyond the defaults.  ratings[nnz] (parallel to `items`, NULL = none): with --rf a step feeds the
 * item index and n_items + the rating's one-hot index round(rating * 2) - 1 (rnn_base.py:590-642; model built with n_feat = 2,
 * input_size = n_items + 10).  shuffle_targets != 0: --shuffle_targets -- a row's targets are a uniform random subset of the
 * whole remaining sequence instead of its first items (target_selection.py:45-46).  The number of targets per row is the
 * model's (n_targets of the multi-target losses, 1 otherwise). */
int sbr_dataset_set_options(sbr_dataset* d, const float* ratings, int shuffle_targets);
/* Sequence noise for the pass planned NEXT (SequenceNoise.__call__, sequence_noise.py:52-94; call before sbr_dataset_plan_pass,
 * once per pass): per user, in the reference's order -- dropout of items (a user left with fewer than two items yields no rows
 * this pass), swaps of neighbours with probability n_swap (an item swaps at most once), swaps with the item int(N(0, 1) *
 * shuf_std) places away with probability n_shuf, rating +- 0.5 clamped to [1, 5] with probability n_ratings (only with ratings
 * attached).  The batches of the pass read the noised copy; rows carried over from the previous pass are re-drawn from it.
 * All probabilities 0: the pass reads the sequences as they are.  A law, not the reference's random stream. */
int sbr_dataset_noise_pass(sbr_dataset* d, float dropout, float swap, float shuf, float shuf_std, float ratings_perturb, uint64_t seed);
/* --target_bias (SelectTargets, target_selection.py:36-53): keep_prob[n_items] = (min(pop) / pop) ** bias; a candidate target
 * survives with that probability, a row none of whose remaining items survives is skipped and does not count towards its batch
 * (rnn_base.py:404-409).  Which rows a pass has is then a draw, so with a table set the rows of a pass -- user, split point,
 * target positions -- are planned on the host at sbr_dataset_plan_pass (sbr_plan_rows_host below: the reference's procedure
 * with splitmix64 draws seeded from `seed` and the pass number) and uploaded; sbr_build_batch only packs them.  n_targets must
 * be the model's.  With sequence noise on, the rows a pass carries over are dropped (they index the previous pass's copy).
 * keep_prob == NULL switches back to device-drawn rows. */
int sbr_dataset_set_target_bias(sbr_dataset* d, const float* keep_prob, int32_t n_targets, uint64_t seed);
/* The host row planner (no device needed; used by the CPU tests).  items / offsets: the CSR of sbr_dataset_create; lengths:
 * NULL or the current length of every user's sequence (<= its CSR extent); order: NULL or the user order of the pass;
 * pend_*: rows carried in (n_pend < batch_size of them) and out; row_*: the rows of the complete batches, batch-major,
 * row_tgt[n_rows][n_targets] = positions inside the remaining sequence (items[offsets[u] + split + pos]), -1 behind the last. */
int sbr_plan_rows_host(const int32_t* items, const int64_t* offsets, const int64_t* lengths, const int32_t* order, int64_t n_users,
                       int32_t batch_size, int32_t n_targets, int32_t shuffle, const float* keep_prob, uint64_t seed,
                       int32_t* pend_user, int32_t* pend_split, int32_t* pend_tgt, int32_t* n_pend, int64_t cap_rows,
                       int32_t* row_user, int32_t* row_split, int32_t* row_tgt, int64_t* n_rows, int64_t* n_batches);
/* (tests / tooling) the sequences the next planned pass reads: items[nnz] and rating_index[nnz] (may be NULL) in the CSR layout
 * of sbr_dataset_create -- user u's items[offsets[u] .. offsets[u] + lengths[u]) -- and lengths[n_users]; the noised copy behind
 * sbr_dataset_noise_pass, the sequences as uploaded otherwise. */
int sbr_dataset_current_sequences(sbr_dataset* d, int32_t* items, int32_t* rating_index, int32_t* lengths);
/* Plans one pass over the users in `order` (n_users ids, NULL = file order) for batches of batch_size rows.
 * A trailing partial batch is carried into the next planned pass, as the reference's endless generator does.
 * n_batches: complete batches now available (indices 0..n_batches-1 for sbr_build_batch). */
int sbr_dataset_plan_pass(sbr_dataset* d, const int32_t* order, int32_t batch_size, int64_t* n_batches);
/* The plan as host arrays, for tests and tooling: segment s = (user, k rows, first row, batch). */
int sbr_dataset_plan_segments(sbr_dataset* d, int64_t* n_segments, const int32_t** seg_user, const int32_t** seg_k,
                              const int32_t** seg_row0, const int32_t** seg_batch);
/* Host-only planner behind sbr_dataset_plan_pass (no device needed; used by the CPU tests).  lengths[n_users];
 * pend_*: the carried partial batch (in/out, capacity batch_size); seg_* capacity n_users + *n_pend. */
int sbr_plan_pass_host(const int64_t* lengths, const int32_t* order, int64_t n_users, int32_t batch_size,
                       int32_t* pend_user, int32_t* pend_k, int32_t* n_pend, int32_t* seg_user, int32_t* seg_k,
                       int32_t* seg_row0, int32_t* seg_batch, int64_t* n_segments, int64_t* n_batches);
/* Builds planned batch `batch` into the engine's own batch buffers (rows [row_offset, row_offset+local_batch)
 * of the global batch for X / lengths / pop; targets: local rows for CCE, all rows for the sampled heads) and
 * makes it the current batch, as sbr_set_batch does.  Entirely on the handle's stream; no host sync. */
int sbr_build_batch(sbr_handle* h, sbr_dataset* d, int64_t batch, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif /* SBR_RNN_H */
