"""ctypes binding of libsbr_rnn.so + `RNNEngine`, the object that stands where the
reference keeps its three Theano callables and the Lasagne parameter list
(neural_networks/rnn_base.py:185 train_function, :196-211 test_function, :188-194
predict_function, :476/:515 get/set_all_param_values).

There is NO CPU path here: if the HIP library is missing or no GPU is visible the
constructor raises.  PyTorch is used for plumbing only (device arena allocation, current
stream, torch.distributed all-reduce of the gradient section in data-parallel runs).
"""
import ctypes
import os

import numpy as np

# HIP multiplexes streams onto this many hardware queues (default 4); the engine's side stream, torch's streams and
# RCCL's must not pile onto one queue.  Read when the HIP runtime initialises, i.e. it only helps when this module is
# imported before the first device call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsbr_rnn.so")

SBR_MAX_LAYERS = 4
SBR_ABI_VERSION = 10
SBR_N_PHASES = 8
PHASE_NAMES = ("gather", "rec_fwd", "output", "rec_bwd", "wgrad", "scatter", "update", "total")

CELLS = {"LSTM": 0, "GRU": 1, "Vanilla": 2}                     # --r_t, recurrent_layers.py:9
LOSSES = {"CCE": 0, "Blackout": 1, "BPR": 2, "TOP1": 3,          # --loss, command_parser.py:43
          "hinge": 4, "logit": 5, "logsig": 6,                   # ... RNNMargin's multi-target losses (command_parser.py:118-119)
          "SCCE": 7, "BPRelu": 8, "lin": 9}                      # ... RNNCluster's further sampled losses (its "CCE" is SCCE here)
MARGIN_LOSSES = ("hinge", "logit", "logsig")
SAMPLED_LOSSES = ("Blackout", "BPR", "TOP1", "SCCE", "BPRelu", "lin")      # the last three: RNNCluster's (rnn_cluster.py:158-175)
UPDATERS = {"adagrad": 0, "adadelta": 1, "rmsprop": 2, "nesterov": 3, "adam": 4}   # --u_m
FLAG_SIMPLE_REC = 1
FLAG_SIMPLE_GEMM = 2
FLAG_ATOMIC_SCATTER = 4
FLAG_PROFILE_REC = 8
FLAG_F32_MFMA = 16
FLAG_SPARSE_UPDATE = 32      # row-sparse optimizer steps for every block that exists (default: where a step cannot touch every row)
FLAG_DENSE_UPDATE = 64       # never: lasagne's dense pass over every parameter
FLAG_BF16_PROJECTION = 128   # output projection on plain bf16 operands (one MFMA per block) -- training forward of CCE, predict, top-k
FLAG_BF16_LAYERS = 256       # the dense GEMMs between stacked layers (input projection of layer >= 2 and its backward pair) on plain bf16 operands


class SbrConfig(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("cell", ctypes.c_int32), ("n_layers", ctypes.c_int32),
                ("layers", ctypes.c_int32 * SBR_MAX_LAYERS), ("n_items", ctypes.c_int32),
                ("input_size", ctypes.c_int32), ("n_feat", ctypes.c_int32), ("max_length", ctypes.c_int32),
                ("batch_size", ctypes.c_int32), ("local_batch", ctypes.c_int32), ("row_offset", ctypes.c_int32),
                ("loss", ctypes.c_int32), ("n_samples", ctypes.c_int32), ("updater", ctypes.c_int32),
                ("learning_rate", ctypes.c_float), ("rho", ctypes.c_float), ("beta1", ctypes.c_float),
                ("beta2", ctypes.c_float), ("regularization", ctypes.c_float), ("grad_clip", ctypes.c_float),
                ("flags", ctypes.c_int32), ("embedding_size", ctypes.c_int32), ("bidirectional", ctypes.c_int32),
                ("balance", ctypes.c_float), ("n_targets", ctypes.c_int32), ("unique", ctypes.c_int32)]


# every symbol include/sbr_rnn.h declares (tests check the library exports all of them)
EXPORTS = ["sbr_last_error", "sbr_abi_version", "sbr_arena_bytes", "sbr_create", "sbr_destroy", "sbr_num_params",
           "sbr_param_shape", "sbr_describe_param", "sbr_set_params", "sbr_get_params", "sbr_get_grads", "sbr_section", "sbr_set_batch",
           "sbr_set_default_target",
           "sbr_train_step", "sbr_train_step_lagged", "sbr_lagged_flush", "sbr_zero_grads", "sbr_forward", "sbr_loss_backward_output", "sbr_backward_recurrent",
           "sbr_apply_update", "sbr_read_cost", "sbr_predict_scores", "sbr_topk", "sbr_debug_buffer",
           "sbr_copy_to_host", "sbr_synchronize", "sbr_enable_timing", "sbr_phase_times", "sbr_chain_times", "sbr_query",
           "sbr_set_deferred_join", "sbr_join_side", "sbr_debug_gemm", "sbr_debug_scatter", "sbr_debug_occupy", "sbr_flush_lazy", "sbr_sparse_info", "sbr_sparse_pack",
           "sbr_sparse_unpack_add", "sbr_dense_ranges", "sbr_sparse_pack_device", "sbr_sparse_unpack_add_all",
           "sbr_cluster_create", "sbr_cluster_destroy", "sbr_cluster_set_params", "sbr_cluster_get_params", "sbr_cluster_get_grads",
           "sbr_cluster_set_scale", "sbr_cluster_forward_backward", "sbr_cluster_apply_update", "sbr_cluster_select",
           "sbr_cluster_mask_scores", "sbr_cluster_hard",
           "sbr_dataset_create", "sbr_dataset_destroy", "sbr_dataset_set_tables", "sbr_dataset_set_options", "sbr_dataset_noise_pass", "sbr_dataset_current_sequences", "sbr_dataset_set_target_bias",
           "sbr_plan_rows_host", "sbr_dataset_plan_pass",
           "sbr_dataset_plan_segments", "sbr_plan_pass_host", "sbr_build_batch"]

_lib = None


def load_library(path=None):
    """dlopen libsbr_rnn.so (import torch first so one HIP runtime serves both)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("SBR_LIB") or LIB_PATH      # SBR_LIB: another build of the library (kernel experiments)
    try:        # torch bundles its own libamdhip64: it must be the one already loaded when the extension binds to HIP,
        import torch  # noqa: F401  otherwise two runtimes coexist and this library sees no device
    except ImportError:
        pass
    if not os.path.exists(path):
        raise RuntimeError("HIP extension %s is missing: build it with __graft_entry__.build() "
                           "(there is no CPU fallback)" % path)
    lib = ctypes.CDLL(path)
    lib.sbr_last_error.restype = ctypes.c_char_p
    vp, i32p, f32p = ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_float)
    lib.sbr_arena_bytes.argtypes = [ctypes.POINTER(SbrConfig), ctypes.POINTER(ctypes.c_size_t)]
    lib.sbr_create.argtypes = [ctypes.POINTER(SbrConfig), vp, ctypes.c_size_t, vp, ctypes.POINTER(vp)]
    lib.sbr_destroy.argtypes = [vp]
    lib.sbr_destroy.restype = None
    lib.sbr_num_params.argtypes = [vp]
    lib.sbr_param_shape.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    lib.sbr_describe_param.argtypes = [ctypes.POINTER(SbrConfig), ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t,
                                       ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int)]
    for fn in (lib.sbr_set_params, lib.sbr_get_params, lib.sbr_get_grads):
        fn.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp)]
    lib.sbr_section.argtypes = [vp, ctypes.c_int, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t),
                                ctypes.POINTER(ctypes.c_size_t)]
    lib.sbr_set_batch.argtypes = [vp, vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int]
    lib.sbr_set_default_target.argtypes = [vp, vp]
    lib.sbr_train_step.argtypes = [vp, f32p]
    lib.sbr_train_step_lagged.argtypes = [vp, f32p, i32p]
    lib.sbr_lagged_flush.argtypes = [vp, f32p, i32p]
    for fn in (lib.sbr_zero_grads, lib.sbr_forward, lib.sbr_loss_backward_output, lib.sbr_backward_recurrent,
               lib.sbr_apply_update, lib.sbr_synchronize):
        fn.argtypes = [vp]
    lib.sbr_read_cost.argtypes = [vp, f32p]
    lib.sbr_predict_scores.argtypes = [vp, ctypes.c_int, vp]
    lib.sbr_topk.argtypes = [vp, ctypes.c_int, ctypes.c_int, vp]
    lib.sbr_debug_buffer.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(vp), ctypes.POINTER(ctypes.c_size_t)]
    lib.sbr_copy_to_host.argtypes = [vp, vp, vp, ctypes.c_size_t]
    lib.sbr_enable_timing.argtypes = [vp, ctypes.c_int]
    lib.sbr_phase_times.argtypes = [vp, f32p]
    lib.sbr_chain_times.argtypes = [vp, ctypes.c_int, f32p, ctypes.POINTER(ctypes.c_int)]
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.sbr_query.argtypes = [vp, ctypes.c_char_p, i64p]
    lib.sbr_debug_scatter.argtypes = [vp, ctypes.c_int, f32p, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.sbr_debug_occupy.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.sbr_debug_gemm.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64,
                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, vp, ctypes.c_size_t, ctypes.c_int32]
    lib.sbr_flush_lazy.argtypes = [vp]
    lib.sbr_sparse_info.argtypes = [vp, ctypes.c_int, i64p, i64p, i64p]
    lib.sbr_sparse_pack.argtypes = [vp, ctypes.c_int, vp, vp, i32p]
    lib.sbr_sparse_unpack_add.argtypes = [vp, ctypes.c_int, vp, vp, ctypes.c_int]
    lib.sbr_dense_ranges.argtypes = [vp, ctypes.c_int, i64p, i64p, ctypes.POINTER(ctypes.c_int)]
    lib.sbr_set_deferred_join.argtypes = [vp, ctypes.c_int]
    lib.sbr_join_side.argtypes = [vp]
    lib.sbr_cluster_create.restype = vp
    lib.sbr_cluster_create.argtypes = [vp, vp]
    lib.sbr_cluster_destroy.restype = None
    lib.sbr_cluster_destroy.argtypes = [vp]
    for fn in (lib.sbr_cluster_set_params, lib.sbr_cluster_get_params, lib.sbr_cluster_get_grads):
        fn.argtypes = [vp, vp, vp]
    lib.sbr_cluster_set_scale.argtypes = [vp, ctypes.c_float]
    lib.sbr_cluster_forward_backward.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, f32p]
    lib.sbr_cluster_apply_update.argtypes = [vp]
    lib.sbr_cluster_select.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.sbr_cluster_mask_scores.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, vp, vp]
    lib.sbr_cluster_hard.argtypes = [vp, vp]
    lib.sbr_dataset_create.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.POINTER(vp)]
    lib.sbr_dataset_destroy.argtypes = [vp]
    lib.sbr_dataset_set_tables.argtypes = [vp, vp, vp]
    lib.sbr_dataset_set_options.argtypes = [vp, vp, ctypes.c_int]
    lib.sbr_dataset_noise_pass.argtypes = [vp, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_uint64]
    lib.sbr_dataset_current_sequences.argtypes = [vp, vp, vp, vp]
    lib.sbr_dataset_set_target_bias.argtypes = [vp, vp, ctypes.c_int32, ctypes.c_uint64]
    lib.sbr_plan_rows_host.argtypes = [vp, vp, vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, vp, ctypes.c_uint64,
                                       vp, vp, vp, i32p, ctypes.c_int64, vp, vp, vp, i64p, i64p]
    lib.sbr_dataset_plan_pass.argtypes = [vp, vp, ctypes.c_int32, i64p]
    lib.sbr_dataset_plan_segments.argtypes = [vp, i64p, ctypes.POINTER(i32p), ctypes.POINTER(i32p), ctypes.POINTER(i32p),
                                              ctypes.POINTER(i32p)]
    lib.sbr_plan_pass_host.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int32, vp, vp, i32p, vp, vp, vp, vp, i64p, i64p]
    lib.sbr_build_batch.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_uint64]
    if path == LIB_PATH:
        _lib = lib
    return lib


class SbrError(RuntimeError):
    pass


def mask_to_lengths(mask):
    """The reference's masks are prefix masks (rnn_one_hot.py:100-101: mask[i, :len] = 1);
    the engine takes lengths.  Anything else is rejected loudly."""
    mask = np.asarray(mask)
    lengths = (mask != 0).sum(axis=1).astype(np.int32)
    if not np.array_equal(mask != 0, np.arange(mask.shape[1])[None, :] < lengths[:, None]):
        raise ValueError("mask must be a left-aligned prefix mask (rows of ones followed by zeros)")
    return lengths


def plan_pass_host(lengths, order, batch_size, pending=None, lib=None):
    """The batch plan of one pass over the users (sbr_plan_pass_host; no GPU needed): which user contributes how many
    rows to which batch, exactly as the reference fills its batches (rnn_base.py:394-415).
    Returns (segments, n_batches, pending): segments = int array (n, 4) of (user, k, first_row, batch) for the complete
    batches; pending = [(user, k), ...] of the trailing partial batch, to be passed to the next pass."""
    lib = lib or load_library()
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    n = len(lengths)
    order_a = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
    pending = list(pending or [])
    pu, pk = np.zeros(batch_size, np.int32), np.zeros(batch_size, np.int32)
    for i, (u, k) in enumerate(pending):
        pu[i], pk[i] = u, k
    npend = ctypes.c_int32(len(pending))
    cap = n + len(pending) + 1
    su, sk, sr, sb = (np.zeros(cap, np.int32) for _ in range(4))
    ns, nb = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.sbr_plan_pass_host(lengths.ctypes.data, None if order_a is None else order_a.ctypes.data, n, batch_size,
                                pu.ctypes.data, pk.ctypes.data, ctypes.byref(npend), su.ctypes.data, sk.ctypes.data,
                                sr.ctypes.data, sb.ctypes.data, ctypes.byref(ns), ctypes.byref(nb))
    if rc != 0:
        raise ValueError(lib.sbr_last_error().decode("utf-8", "replace"))
    seg = np.stack([su[:ns.value], sk[:ns.value], sr[:ns.value], sb[:ns.value]], axis=1)
    return seg, nb.value, [(int(pu[i]), int(pk[i])) for i in range(npend.value)]


def plan_rows_host(items, offsets, order, batch_size, n_targets=1, shuffle=False, keep_prob=None, seed=0, pending=None, lengths=None,
                   lib=None):
    """The rows of one pass planned on the host (sbr_plan_rows_host; no GPU needed): the reference's _gen_mini_batch with a
    target selection that can come back empty (--target_bias).  Returns (rows, n_batches, pending): rows = int array
    (n, 2 + n_targets) of (user, split, target positions...), batch-major; pending = the carried rows in the same format."""
    lib = lib or load_library()
    items = np.ascontiguousarray(items, dtype=np.int32)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    lens = None if lengths is None else np.ascontiguousarray(lengths, dtype=np.int64)
    order_a = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
    kp = None if keep_prob is None else np.ascontiguousarray(keep_prob, dtype=np.float32)
    pend = np.zeros((0, 2 + n_targets), np.int32) if pending is None else np.asarray(pending, dtype=np.int32).reshape(-1, 2 + n_targets)
    pu, ps, pt = np.zeros(batch_size, np.int32), np.zeros(batch_size, np.int32), np.zeros(batch_size * n_targets, np.int32)
    pu[:len(pend)], ps[:len(pend)] = pend[:, 0], pend[:, 1]
    pt[:len(pend) * n_targets] = pend[:, 2:].reshape(-1)
    npend = ctypes.c_int32(len(pend))
    L = (offsets[1:] - offsets[:-1]) if lens is None else lens
    cap = int(batch_size + np.minimum(batch_size, np.maximum(0, L - 2)).sum())
    ru, rs, rt = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap * n_targets, np.int32)
    nr, nb = ctypes.c_int64(), ctypes.c_int64()
    rc = lib.sbr_plan_rows_host(items.ctypes.data, offsets.ctypes.data, None if lens is None else lens.ctypes.data,
                                None if order_a is None else order_a.ctypes.data, n, int(batch_size), int(n_targets), 1 if shuffle else 0,
                                None if kp is None else kp.ctypes.data, int(seed) & 0xFFFFFFFFFFFFFFFF, pu.ctypes.data, ps.ctypes.data,
                                pt.ctypes.data, ctypes.byref(npend), cap, ru.ctypes.data, rs.ctypes.data, rt.ctypes.data,
                                ctypes.byref(nr), ctypes.byref(nb))
    if rc != 0:
        raise ValueError(lib.sbr_last_error().decode("utf-8", "replace"))
    rows = np.concatenate([ru[:nr.value, None], rs[:nr.value, None], rt[:nr.value * n_targets].reshape(-1, n_targets)], axis=1)
    k = npend.value
    pending = np.concatenate([pu[:k, None], ps[:k, None], pt[:k * n_targets].reshape(-1, n_targets)], axis=1)
    return rows, nb.value, pending


class DeviceDataset(object):
    """The training sequences in HBM (CSR) + the per-pass batch plan (sbr_dataset_* in include/sbr_rnn.h)."""

    def __init__(self, engine, items, offsets, n_items):
        self.engine, self.lib = engine, engine.lib
        items = np.ascontiguousarray(items, dtype=np.int32)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        self.n_users, self.n_batches = len(offsets) - 1, 0
        d = ctypes.c_void_p()
        with engine.torch.cuda.device(engine.device):
            engine._check(self.lib.sbr_dataset_create(items.ctypes.data, offsets.ctypes.data, self.n_users, int(n_items),
                                                      ctypes.c_void_p(engine.stream.cuda_stream), ctypes.byref(d)))
        self.d = d

    def set_tables(self, pop_db=None, sample_cdf=None):
        a = None if pop_db is None else np.ascontiguousarray(pop_db, dtype=np.float32)
        c = None if sample_cdf is None else np.ascontiguousarray(sample_cdf, dtype=np.float64)
        self.engine._check(self.lib.sbr_dataset_set_tables(self.d, None if a is None else a.ctypes.data,
                                                           None if c is None else c.ctypes.data))

    def set_options(self, ratings=None, shuffle_targets=False):
        """ratings (nnz,) parallel to the items: --rf feeds item index + rating index (model: n_feat 2, input_size N + 10);
        shuffle_targets: --shuffle_targets (sbr_dataset_set_options)."""
        r = None if ratings is None else np.ascontiguousarray(ratings, dtype=np.float32)
        self.engine._check(self.lib.sbr_dataset_set_options(self.d, None if r is None else r.ctypes.data, 1 if shuffle_targets else 0))

    def set_target_bias(self, keep_prob=None, n_targets=1, seed=0):
        """--target_bias: keep_prob (n_items,) = (min(pop) / pop) ** bias; the rows of a pass are then planned on the host
        (sbr_dataset_set_target_bias).  None switches back to device-drawn rows."""
        kp = None if keep_prob is None else np.ascontiguousarray(keep_prob, dtype=np.float32)
        self.engine._check(self.lib.sbr_dataset_set_target_bias(self.d, None if kp is None else kp.ctypes.data, int(n_targets),
                                                                int(seed) & 0xFFFFFFFFFFFFFFFF))

    def noise_pass(self, dropout=0.0, swap=0.0, shuf=0.0, shuf_std=0.0, ratings_perturb=0.0, seed=0):
        """Sequence noise for the pass planned next (sbr_dataset_noise_pass; SequenceNoise.__call__)."""
        with self.engine.torch.cuda.device(self.engine.device):
            self.engine._check(self.lib.sbr_dataset_noise_pass(self.d, float(dropout), float(swap), float(shuf), float(shuf_std),
                                                               float(ratings_perturb), int(seed) & 0xFFFFFFFFFFFFFFFF))

    def current_sequences(self, nnz):
        """(items, rating_index, lengths) the next planned pass reads: the noised copy behind noise_pass (tests / tooling)."""
        items, rate = np.zeros(max(1, int(nnz)), np.int32), np.zeros(max(1, int(nnz)), np.int32)
        lens = np.zeros(self.n_users, np.int32)
        with self.engine.torch.cuda.device(self.engine.device):
            self.engine._check(self.lib.sbr_dataset_current_sequences(self.d, items.ctypes.data, rate.ctypes.data, lens.ctypes.data))
        return items[:int(nnz)], rate[:int(nnz)], lens

    def plan_pass(self, order, batch_size):
        o = None if order is None else np.ascontiguousarray(order, dtype=np.int32)
        nb = ctypes.c_int64()
        self.engine._check(self.lib.sbr_dataset_plan_pass(self.d, None if o is None else o.ctypes.data, int(batch_size),
                                                          ctypes.byref(nb)))
        self.n_batches = nb.value
        return nb.value

    def segments(self):
        n = ctypes.c_int64()
        ptrs = [ctypes.POINTER(ctypes.c_int32)() for _ in range(4)]
        self.engine._check(self.lib.sbr_dataset_plan_segments(self.d, ctypes.byref(n), *[ctypes.byref(p) for p in ptrs]))
        return np.stack([np.ctypeslib.as_array(p, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
                         for p in ptrs], axis=1)

    def close(self):
        if getattr(self, "d", None):
            self.lib.sbr_dataset_destroy(self.d)
            self.d = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_config(cell="GRU", layers=(50,), n_items=None, max_length=30, batch_size=16, loss="CCE", n_samples=0,
                updater="adam", learning_rate=0.001, rho=0.9, beta1=0.9, beta2=0.999, regularization=0.0, grad_clip=100.0,
                input_size=None, n_feat=1, local_batch=None, row_offset=0, flags=0, embedding_size=0, bidirectional=False,
                balance=1.0, n_targets=1, unique=True):
    """sbr_config for the options RNNBase.__init__ / prepare_model fix (include/sbr_rnn.h)."""
    if cell not in CELLS:
        raise ValueError("Unknown layer type")                      # recurrent_layers.py:90
    if loss not in LOSSES:
        raise ValueError("Unknown loss for the RNN model")          # command_parser.py:123
    if updater not in UPDATERS:
        raise ValueError("Unknown update option")                   # update_manager.py:22
    layers = [int(h) for h in layers]
    if len(layers) > SBR_MAX_LAYERS:
        raise ValueError("at most %d recurrent layers" % SBR_MAX_LAYERS)
    cfg = SbrConfig()
    cfg.abi_version = SBR_ABI_VERSION
    cfg.cell = CELLS[cell]
    cfg.n_layers = len(layers)
    for i, hsz in enumerate(layers):
        cfg.layers[i] = hsz
    cfg.n_items = int(n_items)
    cfg.input_size = int(input_size if input_size is not None else n_items)
    cfg.n_feat = int(n_feat)
    cfg.max_length = int(max_length)
    cfg.batch_size = int(batch_size)
    cfg.local_batch = int(local_batch if local_batch is not None else batch_size)
    cfg.row_offset = int(row_offset)
    cfg.loss = LOSSES[loss]
    cfg.n_samples = int(n_samples) if loss in SAMPLED_LOSSES else 0
    cfg.balance, cfg.n_targets, cfg.unique = float(balance), max(1, int(n_targets)) if loss in MARGIN_LOSSES else 1, 1 if unique else 0
    cfg.updater = UPDATERS[updater]
    cfg.learning_rate, cfg.rho, cfg.beta1, cfg.beta2 = learning_rate, rho, beta1, beta2
    cfg.regularization = regularization
    cfg.grad_clip = grad_clip
    cfg.flags = int(flags)
    sp = os.environ.get("SBR_SPARSE_UPDATE")             # 1 / 0 force the row-sparse optimizer on / off (include/sbr_rnn.h)
    if sp is not None and not cfg.flags & (FLAG_SPARSE_UPDATE | FLAG_DENSE_UPDATE):
        cfg.flags |= FLAG_SPARSE_UPDATE if sp not in ("0", "") else FLAG_DENSE_UPDATE
    cfg.embedding_size = max(0, int(embedding_size))     # --r_emb: a size < 1 means no embedding layer
    cfg.bidirectional = 1 if bidirectional else 0        # --r_bi
    return cfg


def describe_params(cfg, lib=None):
    """[(name, shape)] of lasagne.layers.get_all_param_values(l_out) for a configuration (sbr_describe_param: host only,
    no GPU needed)."""
    lib = lib or load_library()
    out, i = [], 0
    name = ctypes.create_string_buffer(96)
    while True:
        dims, nd = (ctypes.c_int64 * 2)(), ctypes.c_int()
        if lib.sbr_describe_param(ctypes.byref(cfg), i, name, 96, dims, ctypes.byref(nd)) != 0:
            if i == 0:
                raise ValueError(lib.sbr_last_error().decode("utf-8", "replace"))
            return out
        out.append((name.value.decode(), tuple(int(d) for d in dims[:nd.value])))
        i += 1


def initial_values(descs, rng, last_layer_init=1.0):
    """Random-init arrays in get_all_param_values order by the initialisers the reference's layers name [3P]: gate weights
    and peepholes of the LSTM / GRU / index-input Vanilla layers Normal(std 0.1) (sparse_lstm.py:143-171 Gate defaults),
    biases and initial states 0, stock RecurrentLayer weights (dense Vanilla layers) init.Uniform() = U(-0.01, 0.01),
    EmbeddingLayer W init.Normal() = std 0.01, output W GlorotUniform(gain) (rnn_one_hot.py:65, rnn_sampling.py:131),
    output b 0.  rng: numpy Generator or RandomState."""
    out = []
    for name, shp in descs:
        base = name.split(".", 1)[1]
        if name == "out.W":
            lim = last_layer_init * np.sqrt(6.0 / (shp[0] + shp[1]))
            a = rng.uniform(-lim, lim, size=shp)
        elif name == "emb.W":
            a = rng.normal(0.0, 0.01, size=shp)
        elif base in ("input_to_hidden.W", "hidden_to_hidden.W"):
            a = rng.uniform(-0.01, 0.01, size=shp)
        elif base.startswith("W_"):
            a = rng.normal(0.0, 0.1, size=shp)
        else:
            a = np.zeros(shp)
        out.append(np.asarray(a, dtype=np.float32))
    return out


class RNNEngine(object):
    """Device-resident model + optimizer state behind the reference's callables.

    train_function / test_function / predict_function take the same tuples the Theano
    functions take (rnn_one_hot.py:61,106; rnn_sampling.py:128,194); `exclude` is accepted
    and ignored in training exactly like the reference (rnn_base.py:185
    on_unused_input='ignore') and derived on the device for the test path.
    """

    def __init__(self, cell="GRU", layers=(50,), n_items=None, max_length=30, batch_size=16, loss="CCE",
                 n_samples=0, updater="adam", learning_rate=0.001, rho=0.9, beta1=0.9, beta2=0.999,
                 regularization=0.0, grad_clip=100.0, input_size=None, n_feat=1, local_batch=None, row_offset=0,
                 flags=0, device=None, embedding_size=0, bidirectional=False, balance=1.0, n_targets=1, unique=True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("RNNEngine needs a HIP device (MI355X); there is no CPU fallback")
        self.torch = torch
        self.lib = load_library()
        if cell not in CELLS:
            raise ValueError("Unknown layer type")                      # recurrent_layers.py:90
        if loss not in LOSSES:
            raise ValueError("Unknown loss for the RNN model")          # command_parser.py:123
        if updater not in UPDATERS:
            raise ValueError("Unknown update option")                   # update_manager.py:22
        layers = [int(h) for h in layers]
        cfg = make_config(cell=cell, layers=layers, n_items=n_items, max_length=max_length, batch_size=batch_size, loss=loss,
                          n_samples=n_samples, updater=updater, learning_rate=learning_rate, rho=rho, beta1=beta1, beta2=beta2,
                          regularization=regularization, grad_clip=grad_clip, input_size=input_size, n_feat=n_feat,
                          local_batch=local_batch, row_offset=row_offset, flags=flags, embedding_size=embedding_size,
                          bidirectional=bidirectional, balance=balance, n_targets=n_targets, unique=unique)
        self.cfg = cfg
        self.n_targets = cfg.n_targets
        self.cell, self.layers, self.loss = cell, layers, loss
        self.n_items, self.max_length = cfg.n_items, cfg.max_length
        self.batch_size, self.local_batch, self.n_feat = cfg.batch_size, cfg.local_batch, cfg.n_feat
        self.n_samples = cfg.n_samples

        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        nbytes = ctypes.c_size_t()
        self._check(self.lib.sbr_arena_bytes(ctypes.byref(cfg), ctypes.byref(nbytes)))
        self.arena_bytes = nbytes.value
        with torch.cuda.device(self.device):
            self.arena = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=self.device)
            self.stream = torch.cuda.current_stream(self.device)
            handle = ctypes.c_void_p()
            self._check(self.lib.sbr_create(ctypes.byref(cfg), ctypes.c_void_p(self.arena.data_ptr()),
                                            ctypes.c_size_t(self.arena.numel() * 4),
                                            ctypes.c_void_p(self.stream.cuda_stream), ctypes.byref(handle)))
        self.h = handle
        # data-parallel guard (parallel.DataParallel sets it with more than one rank): calls that bring lazily stepped rows of the
        # row-sparse blocks up to date must then be made by every rank at the same step, through DataParallel's collectives
        self.dp_guard = False
        self._dp_collective = False
        self.n_params = self.lib.sbr_num_params(self.h)
        self.param_descs = describe_params(cfg, self.lib)
        self.param_shapes = []
        for i in range(self.n_params):
            dims = (ctypes.c_int64 * 2)()
            nd = ctypes.c_int()
            self._check(self.lib.sbr_param_shape(self.h, i, dims, ctypes.byref(nd)))
            self.param_shapes.append(tuple(int(d) for d in dims[:nd.value]))
        self._sections = {}

    # ---------------------------------------------------------------- plumbing
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.sbr_last_error().decode("utf-8", "replace")
            if rc == -1:
                raise ValueError(msg)
            raise SbrError("libsbr_rnn error %d: %s" % (rc, msg))

    def close(self):
        if getattr(self, "h", None):
            self.lib.sbr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ptr_array(self, arrays):
        arr = (ctypes.c_void_p * len(arrays))()
        for i, a in enumerate(arrays):
            arr[i] = a.ctypes.data
        return arr

    def build_batch(self, dataset, batch, seed):
        """Native batch builder: planned batch `batch` of `dataset` becomes the current batch (no host arrays)."""
        self._check(self.lib.sbr_build_batch(self.h, dataset.d, int(batch), ctypes.c_uint64(int(seed) & (2 ** 64 - 1))))

    def current_batch(self):
        """Host copies of the current batch held in the engine's own buffers (tests/tooling):
        X (B,T), lengths (B,), target, pop (B,), samples (S,)."""
        out = {}
        for name, rows in (("X", self.local_batch * self.max_length * self.n_feat), ("lengths", self.local_batch),
                           ("target", self.batch_size if self.n_samples else self.local_batch * self.n_targets), ("pop", self.local_batch),
                           ("samples", self.n_samples)):
            if rows == 0:
                continue
            buf = self.debug_buffer("batch_" + name)[:rows]
            out[name] = buf if name == "pop" else buf.view(np.int32)
        out["X"] = out["X"].reshape(self.local_batch, self.max_length, self.n_feat)
        return out

    def section(self, which):
        """torch view of a flat arena section: 'params', 'grads' (last element = cost), 'state'.
        Returns (tensor, split) where split = first float of the output-layer part."""
        # (always through the library: for 'params' / 'state' the call also brings lazily updated rows up to date)
        idx = {"params": 0, "grads": 1, "state": 2}[which]
        ptr, n, split = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
        self._check(self.lib.sbr_section(self.h, idx, ctypes.byref(ptr), ctypes.byref(n), ctypes.byref(split)))
        if which not in self._sections:
            off = (ptr.value - self.arena.data_ptr()) // 4
            self._sections[which] = (self.arena[off:off + n.value], split.value)
        return self._sections[which]

    # ---------------------------------------------------------------- parameters
    def set_all_param_values(self, values):
        """lasagne.layers.set_all_param_values(l_out, values) (rnn_base.py:515)."""
        if len(values) != self.n_params:
            raise ValueError("mismatch: got %d values to set %d parameters" % (len(values), self.n_params))
        arrays = []
        for v, shp in zip(values, self.param_shapes):
            a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
            if a.shape != shp:
                raise ValueError("mismatch: parameter has shape %r but value to set has shape %r" % (shp, a.shape))
            arrays.append(a)
        self._check(self.lib.sbr_set_params(self.h, len(arrays), self._ptr_array(arrays)))

    def _get(self, fn):
        arrays = [np.empty(shp, dtype=np.float32) for shp in self.param_shapes]
        self._check(fn(self.h, len(arrays), self._ptr_array(arrays)))
        return arrays

    def _rank_local_flush(self, what):
        """predict / top-k / export / flush replay the zero-gradient steps lazily stepped rows have missed; WHERE a replay is split
        changes float32 roundings, so with several data-parallel ranks a rank-local call would fork the replicas."""
        if self.dp_guard and not self._dp_collective and self.query("sparse_blocks") > 0:
            raise RuntimeError("%s on one data-parallel rank only would bring lazily stepped rows up to date on this replica alone: "
                               "call DataParallel.%s on every rank at the same step (sequence-based-recommendations_amd/parallel.py)"
                               % (what, what))

    def get_all_param_values(self):
        """lasagne.layers.get_all_param_values(l_out) (rnn_base.py:476)."""
        self._rank_local_flush("get_all_param_values")
        return self._get(self.lib.sbr_get_params)

    def get_all_grad_values(self):
        return self._get(self.lib.sbr_get_grads)

    # ---------------------------------------------------------------- batches
    def set_batch(self, X, mask=None, target=None, samples=None, target_popularity=None, lengths=None):
        X = np.ascontiguousarray(np.asarray(X, dtype=np.int32))
        if X.ndim == 2:
            X = X[:, :, None]
        n_rows = X.shape[0]
        if X.shape[1] != self.max_length or X.shape[2] != self.n_feat:
            raise ValueError("X must have shape (rows, %d, %d), got %r" % (self.max_length, self.n_feat, X.shape))
        if lengths is None:
            lengths = mask_to_lengths(mask)
        lengths = np.ascontiguousarray(np.asarray(lengths, dtype=np.int32))
        tgt = None if target is None else np.ascontiguousarray(np.asarray(target, dtype=np.int32))
        smp = None if samples is None else np.ascontiguousarray(np.asarray(samples, dtype=np.int32))
        pop = None if target_popularity is None else np.ascontiguousarray(np.asarray(target_popularity, dtype=np.float32))
        if tgt is not None and self.loss in MARGIN_LOSSES:
            # RNNMargin: (rows, n_targets) positives per row, -1 where a row has fewer (rnn_margin.py:112-147)
            if tgt.ndim == 1:
                tgt = tgt[:, None]
            if tgt.shape[0] != n_rows or tgt.shape[1] > self.n_targets:
                raise ValueError("targets must have shape (%d, <= %d), got %r" % (n_rows, self.n_targets, tgt.shape))
            if tgt.shape[1] < self.n_targets:
                tgt = np.concatenate([tgt, -np.ones((n_rows, self.n_targets - tgt.shape[1]), dtype=np.int32)], axis=1)
            tgt = np.ascontiguousarray(tgt)
        elif tgt is not None:
            need = self.batch_size if self.loss in SAMPLED_LOSSES else n_rows
            if tgt.shape[0] != need:
                raise ValueError("target must have %d entries, got %d" % (need, tgt.shape[0]))
        if self.loss in SAMPLED_LOSSES and smp is not None and smp.shape[0] != self.n_samples:
            raise ValueError("samples must have %d entries" % self.n_samples)
        p = lambda a: None if a is None else ctypes.c_void_p(a.ctypes.data)
        self._check(self.lib.sbr_set_batch(self.h, p(X), p(lengths), p(tgt), p(smp), p(pop), n_rows, 0))
        return n_rows

    def set_default_target(self, default_target=None):
        """RNNMargin --pb: the default target of every item (RNNMargin._default_target, rnn_margin.py:149-161); None = zeros."""
        a = None if default_target is None else np.ascontiguousarray(np.asarray(default_target, dtype=np.float32))
        if a is not None and a.shape != (self.n_items,):
            raise ValueError("default_target must have %d entries" % self.n_items)
        self._check(self.lib.sbr_set_default_target(self.h, None if a is None else ctypes.c_void_p(a.ctypes.data)))

    def set_batch_device(self, X, lengths, target, samples, target_popularity, n_rows):
        """Same, inputs already resident in HBM (torch int32/float32 tensors on this device)."""
        dp = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        self._check(self.lib.sbr_set_batch(self.h, dp(X), dp(lengths), dp(target), dp(samples), dp(target_popularity),
                                           int(n_rows), 1))

    # ---------------------------------------------------------------- the reference's callables
    def train_function(self, X, mask, target, *rest):
        """cost = train_function(X, mask, target, [samples,] target_popularity, exclude)."""
        if self.loss in MARGIN_LOSSES:      # RNNMargin: target = the positives (rows, n_targets); no popularity weights, no samples
            pop = samples = None
        elif self.loss == "CCE":
            pop = rest[0] if len(rest) > 0 else None
            samples = None
        else:
            samples = rest[0]
            pop = rest[1] if len(rest) > 1 else None
        self.set_batch(X, mask, target, samples, pop)
        return self.train_step(sync=True)

    def train_step(self, sync=True):
        if sync:
            cost = ctypes.c_float()
            self._check(self.lib.sbr_train_step(self.h, ctypes.byref(cost)))
            return float(cost.value)
        self._check(self.lib.sbr_train_step(self.h, None))
        return None

    def train_step_lagged(self):
        """Enqueue a step; returns the cost of the PREVIOUS lagged call (None on the first): the training loop's
        form, the host never waits for the step it has just enqueued.  flush_lagged() returns the last one."""
        cost, have = ctypes.c_float(), ctypes.c_int32()
        self._check(self.lib.sbr_train_step_lagged(self.h, ctypes.byref(cost), ctypes.byref(have)))
        return float(cost.value) if have.value else None

    def flush_lagged(self):
        cost, have = ctypes.c_float(), ctypes.c_int32()
        self._check(self.lib.sbr_lagged_flush(self.h, ctypes.byref(cost), ctypes.byref(have)))
        return float(cost.value) if have.value else None

    def forward_backward(self):
        """Gradients without the update (parity tests, data-parallel)."""
        for fn in (self.lib.sbr_zero_grads, self.lib.sbr_forward, self.lib.sbr_loss_backward_output,
                   self.lib.sbr_backward_recurrent):
            self._check(fn(self.h))
        return self.read_cost()

    # phases of one step (data-parallel: all-reduce the gradient section between them)
    def zero_grads(self):
        self._check(self.lib.sbr_zero_grads(self.h))

    def forward(self):
        self._check(self.lib.sbr_forward(self.h))

    def loss_backward_output(self):
        self._check(self.lib.sbr_loss_backward_output(self.h))

    def backward_recurrent(self):
        self._check(self.lib.sbr_backward_recurrent(self.h))

    def apply_update(self):
        self._check(self.lib.sbr_apply_update(self.h))

    def read_cost(self):
        cost = ctypes.c_float()
        self._check(self.lib.sbr_read_cost(self.h, ctypes.byref(cost)))
        return float(cost.value)

    def predict_function(self, X, mask):
        """scores (rows, N): softmax probabilities for CCE, raw activations for sampled heads."""
        self._rank_local_flush("predict_function")
        n = self.set_batch(X, mask)
        out = np.empty((n, self.n_items), dtype=np.float32)
        self._check(self.lib.sbr_predict_scores(self.h, 0, ctypes.c_void_p(out.ctypes.data)))
        return out

    def test_probabilities(self, X, mask):
        n = self.set_batch(X, mask)
        out = np.empty((n, self.n_items), dtype=np.float32)
        self._check(self.lib.sbr_predict_scores(self.h, 1, ctypes.c_void_p(out.ctypes.data)))
        return out

    def test_function(self, theano_inputs, k=10, exclude_seen=True):
        """ids = test_function(theano_inputs, k) (rnn_base.py:205-209): ordered top-k of
        softmax * (1 - exclude), for every row (the reference feeds one row at a time)."""
        self._rank_local_flush("test_function")
        X, mask = theano_inputs[0], theano_inputs[1]
        n = self.set_batch(X, mask)
        ids = np.empty((n, k), dtype=np.int32)
        # exclude_seen: True / 1 = viewed items can never be ranked (top_k_recommendations, rnn_base.py:154-155); 2 = the compiled
        # test function's scores * (1 - exclude) (:201-202): the same ranking except for RNNMargin's raw outputs
        self._check(self.lib.sbr_topk(self.h, int(k), int(exclude_seen), ctypes.c_void_p(ids.ctypes.data)))
        return ids

    # ---------------------------------------------------------------- debug / timing
    def debug_buffer(self, name):
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self.lib.sbr_debug_buffer(self.h, name.encode(), ctypes.byref(ptr), ctypes.byref(n)))
        out = np.empty(n.value, dtype=np.float32)
        self._check(self.lib.sbr_copy_to_host(self.h, ptr, ctypes.c_void_p(out.ctypes.data), n.value))
        return out

    # ---------------------------------------------------------------- row-sparse blocks (include/sbr_rnn.h)
    def flush_lazy(self):
        """Every lazily stepped row current through the last applied step (sbr_flush_lazy)."""
        self._rank_local_flush("flush_lazy")
        self._check(self.lib.sbr_flush_lazy(self.h))

    def sparse_blocks(self):
        """[(n_rows, row_floats, max_local_rows)] of the row-sparse parameter blocks of this configuration."""
        out = []
        for b in range(self.query("sparse_blocks")):
            v = [ctypes.c_int64() for _ in range(3)]
            self._check(self.lib.sbr_sparse_info(self.h, b, *[ctypes.byref(x) for x in v]))
            out.append(tuple(int(x.value) for x in v))
        return out

    def sparse_pack(self, b):
        """This rank's touched gradient rows of block b: (ids int32 [cap], rows float32 [cap, row_floats], count); the
        first `count` entries are valid.  The tensors are the engine's exchange buffers (reused every step)."""
        if not hasattr(self, "_sp_buf"):
            self._sp_buf = {}
        if b not in self._sp_buf:
            n_rows, w, cap = self.sparse_blocks()[b]
            with self.torch.cuda.device(self.device):
                self._sp_buf[b] = (self.torch.empty(cap, dtype=self.torch.int32, device=self.device),
                                   self.torch.empty((cap, w), dtype=self.torch.float32, device=self.device))
        ids, rows = self._sp_buf[b]
        n = ctypes.c_int32()
        self._check(self.lib.sbr_sparse_pack(self.h, b, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(rows.data_ptr()),
                                             ctypes.byref(n)))
        return ids, rows, int(n.value)

    def sparse_unpack_add(self, b, ids, rows, count):
        """Adds ONE rank's packed rows into the gradient block (call for every rank in rank order, own rows included)."""
        if count:
            assert ids.is_contiguous() and rows.is_contiguous()
        self._check(self.lib.sbr_sparse_unpack_add(self.h, b, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(rows.data_ptr()),
                                                   int(count)))

    def sparse_pack_device(self, b):
        """The touched rows of block b with their count IN BAND and nothing synchronised: (ids int32 [1 + cap] with ids[0] =
        the count, rows float32 [cap, row_floats]) -- the engine's exchange buffers, gathered at their fixed capacity."""
        if not hasattr(self, "_sp_dev"):
            self._sp_dev = {}
        if b not in self._sp_dev:
            n_rows, w, cap = self.sparse_blocks()[b]
            with self.torch.cuda.device(self.device):
                self._sp_dev[b] = (self.torch.empty(cap + 1, dtype=self.torch.int32, device=self.device),
                                   self.torch.empty((cap, w), dtype=self.torch.float32, device=self.device))
        ids, rows = self._sp_dev[b]
        self._check(self.lib.sbr_sparse_pack_device(self.h, b, ctypes.c_void_p(ids.data_ptr()), ctypes.c_void_p(rows.data_ptr())))
        return ids, rows

    def sparse_unpack_add_all(self, b, ids_all, rows_all, world):
        """Adds every rank's gathered rows (ids_all [world, 1 + cap], rows_all [world, cap, row_floats]) in rank order."""
        assert ids_all.is_contiguous() and rows_all.is_contiguous()
        self._check(self.lib.sbr_sparse_unpack_add_all(self.h, b, ctypes.c_void_p(ids_all.data_ptr()),
                                                       ctypes.c_void_p(rows_all.data_ptr()), int(world)))

    def dense_ranges(self):
        """[(lo, hi)] float ranges of the gradient section (trailing cost included) outside the sparse blocks."""
        lo, hi, n = (ctypes.c_int64 * 16)(), (ctypes.c_int64 * 16)(), ctypes.c_int()
        self._check(self.lib.sbr_dense_ranges(self.h, 16, lo, hi, ctypes.byref(n)))
        return [(int(lo[i]), int(hi[i])) for i in range(n.value)]

    def set_deferred_join(self, on=True):
        self._check(self.lib.sbr_set_deferred_join(self.h, 1 if on else 0))

    def join_side(self):
        self._check(self.lib.sbr_join_side(self.h))

    def side_stream(self):
        """torch view of the engine's internal side stream (data-parallel: collectives ordered behind it)."""
        return self.torch.cuda.ExternalStream(self.query("side_stream"), device=self.device)

    def side_stream2(self):
        """... and of the second one (overlapped step tail: the embedding scatter-add runs there)."""
        return self.torch.cuda.ExternalStream(self.query("side_stream2"), device=self.device)

    def tail_ranges(self):
        """Overlapped step tail (query('tail_chunks') >= 2), phase-by-phase step: the float ranges of the gradient section
        that the two consumer streams produce: ((W_in lo, hi) on side_stream2, (W_hid lo, hi) on side_stream)."""
        q = self.query
        return (q("tail_win_lo"), q("tail_win_hi")), (q("tail_whid_lo"), q("tail_whid_hi"))

    def debug_scatter(self, reps=10):
        """the stand-alone scatter-add of the embedding gradient over the current batch (sbr_debug_scatter): (us per launch,
        valid entries, distinct rows)"""
        us, n, r = ctypes.c_float(), ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.sbr_debug_scatter(self.h, int(reps), ctypes.byref(us), ctypes.byref(n), ctypes.byref(r)))
        return float(us.value), int(n.value), int(r.value)

    def debug_occupy(self, workgroups, lds_kb, milliseconds):
        """a foreign kernel that holds `workgroups` x `lds_kb` KiB of the chip's LDS for `milliseconds` on a stream of the library's
        own (sbr_debug_occupy; asynchronous); debug_occupy(0, 0, 0) waits for it"""
        self._check(self.lib.sbr_debug_occupy(self.h, int(workgroups), int(lds_kb), int(milliseconds)))

    def query(self, what):
        v = ctypes.c_int64()
        self._check(self.lib.sbr_query(self.h, what.encode(), ctypes.byref(v)))
        return int(v.value)

    def synchronize(self):
        self._check(self.lib.sbr_synchronize(self.h))

    def enable_timing(self, on=True, only=None):
        """on: record the per-phase events of every train step; only="rec_bwd": just the two events around that phase
        (every event record costs the stream a few microseconds)."""
        mode = 0 if not on else 1 if only is None else 2 + PHASE_NAMES.index(only)
        self._check(self.lib.sbr_enable_timing(self.h, mode))

    def phase_times(self):
        us = (ctypes.c_float * SBR_N_PHASES)()
        self._check(self.lib.sbr_phase_times(self.h, us))
        return dict(zip(PHASE_NAMES, [float(v) for v in us]))

    def chain_timing(self, on=True):
        """on: bracket every launch of a recurrent chain kernel (any layer, either direction) with a HIP-event pair from now on;
        off: stop and return {"fwd_us", "bwd_us": device time summed over the launches, "fwd_launches", "bwd_launches"}."""
        if on:
            self._check(self.lib.sbr_chain_times(self.h, 1, None, None))
            return None
        us, n = (ctypes.c_float * 2)(), (ctypes.c_int * 2)()
        self._check(self.lib.sbr_chain_times(self.h, 0, us, n))
        return {"fwd_us": float(us[0]), "bwd_us": float(us[1]), "fwd_launches": int(n[0]), "bwd_launches": int(n[1])}


# ------------------------------------------------------------------------------------------------ RNNCluster's cluster head
CLUSTER_TYPES = {"mix": 0, "softmax": 1, "sigmoid": 2}            # --cluster_type (command_parser.py:77)


class SbrClusterConfig(ctypes.Structure):
    _fields_ = [("abi_version", ctypes.c_int32), ("n_items", ctypes.c_int32), ("n_hidden", ctypes.c_int32),
                ("hidden_split", ctypes.c_int32), ("n_clusters", ctypes.c_int32), ("cluster_type", ctypes.c_int32),
                ("loss", ctypes.c_int32), ("batch_size", ctypes.c_int32), ("max_samples", ctypes.c_int32), ("updater", ctypes.c_int32),
                ("learning_rate", ctypes.c_float), ("rho", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float),
                ("scale", ctypes.c_float), ("noise_std", ctypes.c_float), ("seed", ctypes.c_uint64)]


class ClusterHead(object):
    """The cluster head of RNNCluster beside an RNNEngine (include/sbr_rnn.h, csrc/sbr_cluster.hip): owns the repartition R (N, C)
    and the selection weights Wc (H, C), reads the engine's user representation on the device, trains with its own updater state
    (rnn_cluster.py:237-256, :282-285).  `loss` uses the engine's names (RNNCluster's "CCE" is "SCCE")."""

    def __init__(self, engine, n_clusters, cluster_type="mix", loss="SCCE", max_samples=32, updater="adam", learning_rate=0.01,
                 rho=0.9, beta1=0.9, beta2=0.999, scale=1.0, noise_std=0.0, seed=0):
        if cluster_type not in CLUSTER_TYPES:
            raise ValueError("Unknown cluster type")
        if loss not in SAMPLED_LOSSES:
            raise ValueError("Unknown cluster loss")                     # rnn_cluster.py:101
        self.engine, self.lib, self.torch = engine, engine.lib, engine.torch
        H = int(engine.cfg.layers[engine.cfg.n_layers - 1])
        self.bi = bool(engine.cfg.bidirectional)
        self.n_items, self.n_clusters, self.n_hidden = engine.n_items, int(n_clusters), H * (2 if self.bi else 1)
        self.batch_size, self.max_samples = engine.batch_size, int(max_samples)
        c = SbrClusterConfig()
        c.abi_version, c.n_items, c.n_hidden, c.hidden_split = SBR_ABI_VERSION, self.n_items, self.n_hidden, H
        c.n_clusters, c.cluster_type, c.loss = self.n_clusters, CLUSTER_TYPES[cluster_type], LOSSES[loss]
        c.batch_size, c.max_samples, c.updater = self.batch_size, self.max_samples, UPDATERS[updater]
        c.learning_rate, c.rho, c.beta1, c.beta2 = float(learning_rate), float(rho), float(beta1), float(beta2)
        c.scale, c.noise_std, c.seed = float(scale), float(noise_std), int(seed)
        self.h = self.lib.sbr_cluster_create(ctypes.byref(c), ctypes.c_void_p(engine.stream.cuda_stream))
        if not self.h:
            raise SbrError("sbr_cluster_create: %s" % self.lib.sbr_last_error().decode())
        dev = engine.device
        self._tgt = self.torch.empty(self.batch_size, dtype=self.torch.int32, device=dev)
        self._smp = self.torch.empty(self.max_samples, dtype=self.torch.int32, device=dev)
        self._csel = self.torch.empty(max(self.batch_size, 16), dtype=self.torch.int32, device=dev)
        self._hard = None

    def _check(self, rc):
        if rc != 0:
            raise SbrError("libsbr_rnn error %d: %s" % (rc, self.lib.sbr_last_error().decode()))

    def _h_last(self):
        """(device pointer, row stride, offset of the backwards half) of the engine's user representation"""
        ptr, n = ctypes.c_void_p(), ctypes.c_size_t()
        self.engine._check(self.lib.sbr_debug_buffer(self.engine.h, b"h_last", ctypes.byref(ptr), ctypes.byref(n)))
        Bp = (self.engine.batch_size + 15) // 16 * 16
        ld = n.value // Bp
        return ptr, ld, (ld // 2 if self.bi else 0)

    def close(self):
        if self.h:
            self.lib.sbr_cluster_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_params(self, R, Wc):
        R = np.ascontiguousarray(R, dtype=np.float32); Wc = np.ascontiguousarray(Wc, dtype=np.float32)
        if R.shape != (self.n_items, self.n_clusters) or Wc.shape != (self.n_hidden, self.n_clusters):
            raise ValueError("mismatch: cluster arrays have shapes %r, %r" % (R.shape, Wc.shape))
        self._check(self.lib.sbr_cluster_set_params(self.h, ctypes.c_void_p(R.ctypes.data), ctypes.c_void_p(Wc.ctypes.data)))
        self._hard = None

    def _get(self, fn):
        R = np.empty((self.n_items, self.n_clusters), dtype=np.float32); Wc = np.empty((self.n_hidden, self.n_clusters), dtype=np.float32)
        self._check(fn(self.h, ctypes.c_void_p(R.ctypes.data), ctypes.c_void_p(Wc.ctypes.data)))
        return R, Wc

    def get_params(self):
        return self._get(self.lib.sbr_cluster_get_params)

    def get_grads(self):
        return self._get(self.lib.sbr_cluster_get_grads)

    def set_scale(self, scale):
        self._check(self.lib.sbr_cluster_set_scale(self.h, float(scale)))

    def forward_backward(self, targets, cluster_samples, read_cost=True):
        """cost_clusters and its gradients on the engine's CURRENT user representations (call after engine.forward / a step)."""
        t = np.ascontiguousarray(targets, dtype=np.int32); sm = np.ascontiguousarray(cluster_samples, dtype=np.int32)
        if t.shape[0] != self.batch_size or not 1 <= sm.shape[0] <= self.max_samples:
            raise ValueError("cluster head: %d targets, %d samples (batch %d, at most %d samples)" % (t.shape[0], sm.shape[0], self.batch_size, self.max_samples))
        self._tgt.copy_(self.torch.from_numpy(t)); self._smp[:sm.shape[0]].copy_(self.torch.from_numpy(sm))
        ptr, ld, off2 = self._h_last()
        cost = ctypes.c_float()
        self._check(self.lib.sbr_cluster_forward_backward(self.h, ptr, ld, off2, ctypes.c_void_p(self._tgt.data_ptr()),
                                                          ctypes.c_void_p(self._smp.data_ptr()), int(sm.shape[0]),
                                                          ctypes.byref(cost) if read_cost else None))
        return float(cost.value) if read_cost else None

    def apply_update(self):
        self._check(self.lib.sbr_cluster_apply_update(self.h))
        self._hard = None

    def select(self, rows, with_activations=False):
        """argmax cluster of the first `rows` user representations (rnn_cluster.py:334) [, the selection activations]"""
        ptr, ld, off2 = self._h_last()
        z = self.torch.empty((rows, self.n_clusters), dtype=self.torch.float32, device=self.engine.device) if with_activations else None
        self._check(self.lib.sbr_cluster_select(self.h, ptr, ld, off2, int(rows), ctypes.c_void_p(self._csel.data_ptr()),
                                                ctypes.c_void_p(z.data_ptr()) if z is not None else None))
        csel = self._csel[:rows].cpu().numpy().astype(np.int64)
        return (csel, z.cpu().numpy()) if with_activations else csel

    def hard_clusters(self):
        """_get_hard_clusters() (rnn_cluster.py:293-300) as a host array (N, C); cached until the arrays change"""
        if self._hard is None:
            out = np.empty((self.n_items, self.n_clusters), dtype=np.float32)
            self._check(self.lib.sbr_cluster_hard(self.h, ctypes.c_void_p(out.ctypes.data)))
            self._hard = out
        return self._hard

    def mask_scores(self, scores_dev, csel, want_used=True):
        """device form: scores_dev (rows, >= N) float32 tensor *= hard[:, csel[row]]; returns the items per selected cluster"""
        rows = int(scores_dev.shape[0])
        self._csel[:rows].copy_(self.torch.from_numpy(np.ascontiguousarray(csel, dtype=np.int32)))
        used = self.torch.empty(rows, dtype=self.torch.float32, device=self.engine.device) if want_used else None
        self._check(self.lib.sbr_cluster_mask_scores(self.h, ctypes.c_void_p(scores_dev.data_ptr()), int(scores_dev.stride(0)), rows,
                                                     ctypes.c_void_p(self._csel.data_ptr()),
                                                     ctypes.c_void_p(used.data_ptr()) if used is not None else None))
        return used.cpu().numpy() if used is not None else None
