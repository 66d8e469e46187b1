"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol the header
declares (no compute calls without a GPU), the option surface / filenames follow the reference's
grammar, and the vectorised batch packing equals the oracle's literal loop restatement."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import rnn_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "sequence-based-recommendations_amd", "libsbr_rnn.so")
HEADER = os.path.join(ROOT, "include", "sbr_rnn.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build()
    import sbr_amd.engine as E
    return E.load_library()


def test_library_exports_every_declared_symbol(lib):
    import sbr_amd.engine as E
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(sbr_[a-z_]+)\s*\(", src))
    assert declared == set(E.EXPORTS), declared ^ set(E.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sbr_abi_version() == E.SBR_ABI_VERSION


def _cfg(E, **kw):
    c = E.SbrConfig()
    c.abi_version = E.SBR_ABI_VERSION
    c.cell, c.n_layers = E.CELLS["GRU"], 1
    c.layers[0] = 128
    c.n_items = c.input_size = 3706
    c.n_feat, c.max_length, c.batch_size, c.local_batch = 1, 200, 256, 256
    c.loss, c.updater, c.learning_rate = E.LOSSES["CCE"], E.UPDATERS["adam"], 1e-3
    c.rho, c.beta1, c.beta2, c.grad_clip = 0.9, 0.9, 0.999, 100.0
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def test_arena_size_and_config_validation(lib):
    import sbr_amd.engine as E
    n = ctypes.c_size_t()
    assert lib.sbr_arena_bytes(ctypes.byref(_cfg(E)), ctypes.byref(n)) == 0
    # params+grads+2 adam arrays (~4 x 7.8 MB) + saved activations (~0.5 GB at B=256, T=200)
    assert 300e6 < n.value < 1.5e9
    bad = _cfg(E, cell=7)
    assert lib.sbr_arena_bytes(ctypes.byref(bad), ctypes.byref(n)) == -1
    assert b"Unknown layer type" in lib.sbr_last_error()
    bad = _cfg(E, loss=E.LOSSES["BPR"], n_samples=0)
    assert lib.sbr_arena_bytes(ctypes.byref(bad), ctypes.byref(n)) == -1
    bad = _cfg(E, local_batch=300)
    assert lib.sbr_arena_bytes(ctypes.byref(bad), ctypes.byref(n)) == -1


def test_engine_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from sbr_amd.engine import RNNEngine
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        RNNEngine(cell="GRU", layers=[8], n_items=10)


def test_mask_to_lengths_rejects_non_prefix_masks():
    from sbr_amd.engine import mask_to_lengths
    m = np.array([[1, 1, 0, 0], [1, 1, 1, 1], [0, 0, 0, 0]], dtype=np.float32)
    assert list(mask_to_lengths(m)) == [2, 4, 0]
    with pytest.raises(ValueError):
        mask_to_lengths(np.array([[1, 0, 1, 0]]))


class FakeDataset(object):
    n_items = 50
    item_popularity = np.arange(1, 51, dtype=np.float64)


def _model(cls, **kw):
    from sbr_amd import options as Opt
    common = dict(use_ratings_features=False, use_movies_features=False, use_users_features=False,
                  recurrent_layer=Opt.RecurrentLayers("GRU", [128]), updater=Opt.Adam(), max_length=200, batch_size=256)
    common.update(kw)
    m = cls(**common)
    m.n_items = FakeDataset.n_items
    m.set_dataset(FakeDataset())
    return m


def test_checkpoint_filename_grammar():
    # SURVEY 8(a14) example: rnn_cce_db0.0_r0.0_ml200_bs256_ne1.234_GRU_gc100_h128_Ua_lr0.001_b10.9_b20.999_nt1_nf
    from sbr_amd.models import RNNOneHot, RNNSampling
    from sbr_amd import options as Opt
    m = _model(RNNOneHot)
    assert m._get_model_filename(1.234) == "rnn_cce_db0.0_r0.0_ml200_bs256_ne1.234_GRU_gc100_h128_Ua_lr0.001_b10.9_b20.999_nt1_nf"
    s = _model(RNNSampling, loss_function="BPR", sampling=32.0, recurrent_layer=Opt.RecurrentLayers("LSTM", [100, 50]),
               updater=Opt.Adagrad(0.1), use_ratings_features=True, interactions_are_unique=False)
    assert s._get_model_filename("*") == "rnn_sampling_BPR_s32.0_ini1.0_db0.0_ml200_bs256_ne*_gc100_h100-50_Ug_lr0.1_nt1_ri_rf"
    assert Opt.RecurrentLayers("LSTM", [20], bidirectional=True).name == "bLSTM_gc100_h20"
    assert Opt.SelectTargets(n_targets=3, shuffle=True, bias=0.5).name == "nt3_tb0.5_shufT"
    assert Opt.SequenceNoise(dropout=0.1, swap=0.2).name == "do0.1_sw0.2"
    assert Opt.NesterovMomentum(0.5, 0.8).name == "Un_lr0.5_m0.8" and Opt.RMSProp(1.0, 0.9).name == "Ur_lr1.0_rho0.9"


def test_cli_defaults_match_the_reference_parser():
    from sbr_amd import options as Opt
    a = Opt.command_parser(Opt.predictor_command_parser, Opt.training_command_parser, argv=[])
    # command_parser.py:37,65; recurrent_layers.py:9-10; update_manager.py:4-5; train.py:19,23
    assert (a.batch_size, a.max_length, a.recurrent_layer_type, a.r_l) == (16, 30, "GRU", "50")
    assert (a.update_manager, a.u_l, a.loss, a.sampling, a.save, a.progress) == ("adam", 0.001, "CCE", 32.0, "Best", "2.")
    assert Opt.num("2.") == 2.0 and isinstance(Opt.num("5"), int)


def test_prepare_input_matches_the_literal_restatement():
    from sbr_amd.models import RNNOneHot, RNNSampling
    rng = np.random.default_rng(0)
    seqs = []
    for b in range(6):
        n = int(rng.integers(1, 9))
        in_seq = [[int(rng.integers(0, 50)), float(rng.integers(1, 11)) / 2] for _ in range(n)]
        seqs.append(["u%d" % b, in_seq, [[int(rng.integers(0, 50)), 4.0]]])
    m = _model(RNNOneHot, diversity_bias=0.5, max_length=8, batch_size=6)
    X, mask, Y, pop, excl = m._prepare_input(seqs)
    oX, omask, oY, opop, oexcl = O.prepare_input_one_hot(seqs, 8, 50, FakeDataset.item_popularity, 0.5)
    assert np.array_equal(X, oX) and np.array_equal(mask, omask) and np.array_equal(Y, oY)
    assert np.allclose(pop, opop, rtol=1e-6) and excl is None
    s = _model(RNNSampling, loss_function="TOP1", sampling=7, max_length=8, batch_size=6)
    s.effective_sampling = 7
    X2, mask2, Y2, samples, pop2, _ = s._prepare_input(seqs)
    assert np.array_equal(X2, oX) and samples.shape == (7,) and samples.dtype == np.int32 and samples.max() < 50
    r = _model(RNNOneHot, use_ratings_features=True, max_length=8, batch_size=6)
    Xr = r._prepare_input(seqs)[0]
    assert Xr.shape == (6, 8, 2)
    b, t = 0, 0
    assert Xr[b, t, 1] == 50 + int(round(seqs[b][1][t][1] * 2)) - 1      # rnn_base.py:590-605


def test_gen_mini_batch_policy():
    # rnn_base.py:396-415: distinct sorted split points in [2, len), inputs truncated to max_length
    from sbr_amd.models import RNNOneHot
    m = _model(RNNOneHot, max_length=5, batch_size=7)

    def gen():
        while True:
            yield [[i % 50, 3.0] for i in range(12)], "7"
    X, mask, Y, pop, _ = next(m._gen_mini_batch(gen()))
    assert X.shape == (7, 5, 1)
    lens = mask.sum(1).astype(int)
    assert lens.min() >= 2 and lens.max() <= 5
    for b in range(7):                                       # target = the item right after the input window
        last = X[b, lens[b] - 1, 0]
        assert Y[b] == (last + 1) % 50
    (bi, goal) = next(m._gen_mini_batch(gen(), test=True))   # test mode: split in the middle, goal = the rest
    assert bi[0].shape == (1, 5, 1) and goal == [i % 50 for i in range(6, 12)]


def test_gen_mini_batch_ends_with_a_finite_source():
    # validation/test sets are streamed with epochs=1 (rnn_base.py:367): the batch generator must end with them
    from sbr_amd.models import RNNOneHot
    m = _model(RNNOneHot, max_length=5, batch_size=7)
    src = iter([([[i % 50, 3.0] for i in range(8)], str(u)) for u in range(3)])
    got = list(m._gen_mini_batch(src, test=True))
    assert len(got) == 3 and all(goal == [4, 5, 6, 7] for _, goal in got)


def test_save_load_roundtrip_layout(tmp_path):
    # checkpoints are plain pickled lists of arrays in Lasagne order, protocol 2 (rnn_base.py:470-479)
    import pickle
    from sbr_amd.models import RNNOneHot

    class FakeEngine(object):
        def __init__(self):
            self.vals = [np.full(s, i, dtype=np.float32) for i, (n, s) in enumerate(O.model_param_shapes("GRU", [4], 9))]

        def get_all_param_values(self):
            return self.vals

        def set_all_param_values(self, v):
            self.vals = v
    m = _model(RNNOneHot)
    m.engine = FakeEngine()
    fn = str(tmp_path / "models" / m._get_model_filename(0.5))
    m.save(fn)
    raw = pickle.load(open(fn, "rb"))
    assert isinstance(raw, list) and len(raw) == 12 and raw[0].shape == (9, 4) and raw[-2].shape == (4, 9)
    m.engine.vals = None
    assert m.load_last(str(tmp_path / "models") + "/") == 0.5
    assert len(m.engine.vals) == 12 and m.engine.vals[3][0, 0] == 3.0


def test_evaluator_metrics():
    from sbr_amd.data import Evaluator
    ev = Evaluator(FakeDataset(), k=3)
    ev.add_instance([5, 7, 9], [7, 1, 5, 9])
    ev.add_instance([2], [3, 4, 6])
    assert ev.sps() == 0.5 and ev.user_coverage() == 0.5 and ev.item_coverage() == 2
    assert abs(ev.average_recall() - (2 / 3) / 2) < 1e-12 and abs(ev.average_precision() - (2 / 3) / 2) < 1e-12
    dcg = 1 / np.log2(2) + 1 / np.log2(4); mx = 1 / np.log2(2) + 1 / np.log2(3) + 1 / np.log2(4)
    assert abs(ev.average_ndcg() - (dcg / mx) / 2) < 1e-12


def test_data_handler_reads_the_preprocess_format(tmp_path):
    from sbr_amd.data import DataHandler
    d = tmp_path / "ds" / "data"
    d.mkdir(parents=True)
    (d / "train_set_sequences").write_text("0 1 4.0 2 3.5 3 5.0\n1 2 1.0 0 2.0\n")
    (d / "val_set_sequences").write_text("2 3 4.0 1 4.0 0 1.0 2 2.0\n")
    (d / "test_set_sequences").write_text("3 0 4.0 1 4.0\n")
    (d / "train_set_triplets").write_text("0 1 4.0\n0 2 3.5\n0 3 5.0\n1 2 1.0\n1 0 2.0\n")
    (d / "stats").write_text("set n_users n_items n_interactions longest_sequence\nFull 4 4 11 4\nTrain 2 4 5 3\nVal 1 4 4 4\nTest 1 2 2 2\n")
    dh = DataHandler(str(tmp_path / "ds") + "/")
    assert (dh.n_users, dh.n_items, dh.training_set.n_interactions) == (4, 4, 5)
    assert list(dh.item_popularity) == [1, 1, 2, 1]
    seqs = list(dh.training_set(epochs=1))
    assert seqs[0] == ([[1, 4.0], [2, 3.5], [3, 5.0]], "0") and seqs[1][1] == "1"


def _host_segments(lengths, order, B, n_batches):
    """(user, k) per batch as the reference-style generator produces them: item ids encode (user, position)."""
    from sbr_amd.models import RNNOneHot
    m = _model(RNNOneHot, max_length=6, batch_size=B)
    m.n_items = 10 ** 7

    def gen():
        while True:
            for u in order:
                if lengths[u] >= 2:
                    yield [[u * 1000 + p, 3.0] for p in range(lengths[u])], str(u)
    m._prepare_input = lambda sequences: [(int(uid), tgt[0][0] % 1000) for uid, _, tgt in sequences]
    g = m._gen_mini_batch(gen())
    out = []
    for b in range(n_batches):
        rows = next(g)
        assert len(rows) == B
        segs = []
        for u, l in rows:
            if segs and segs[-1][0] == u:
                assert l > segs[-1][2]                       # sorted distinct split points within a user
                segs[-1] = (u, segs[-1][1] + 1, l)
            else:
                segs.append((u, 1, l))
        out.append([(u, k) for u, k, _ in segs])
    return out


@pytest.mark.parametrize("B,seed", [(7, 0), (16, 1), (5, 2)])
def test_native_batch_plan_equals_the_reference_fill_order(lib, B, seed):
    # rnn_base.py:394-415: users in order, k = min(B - j, len - 2), overflowing users truncated, len-2 users consumed
    # without rows; the partial batch at the end of a pass continues with the first users of the next pass
    from sbr_amd.engine import plan_pass_host
    rng = np.random.default_rng(seed)
    lengths = rng.integers(1, 14, size=23)
    lengths[3], lengths[11] = 2, 40                           # a 2-item user (no rows), one that fills several batches alone
    order = list(rng.permutation(len(lengths)))
    pending, got = [], []
    for _ in range(3):                                        # three passes with carry-over
        seg, nb, pending = plan_pass_host(lengths, order, B, pending, lib=lib)
        for b in range(nb):
            rows = seg[seg[:, 3] == b]
            assert rows[:, 1].sum() == B and list(rows[:, 2]) == list(np.cumsum(np.r_[0, rows[:-1, 1]]))
            got.append([(int(u), int(k)) for u, k in rows[:, :2]])
    want = _host_segments(lengths, order, B, len(got))
    assert got == want


def test_native_batch_plan_with_lengths_that_change_between_passes(lib):
    # sequence noise (sbr_dataset_noise_pass): every pass is planned on that pass's noised lengths; the rows a pass carries
    # over are cut from the NEXT pass's copy of their user, so their count is clamped to what that copy offers, and a user the
    # dropout left with fewer than two items (length 0) owns no rows at all
    from sbr_amd.engine import plan_pass_host
    B = 8
    seg, nb, pending = plan_pass_host([12, 6, 4], None, B, lib=lib)          # 8 rows (user 0, truncated) | 4 + 2 rows carried
    assert nb == 1 and pending == [(1, 4), (2, 2)]
    new_len = np.array([12, 3, 4])                                           # user 1 keeps 3 items this pass: one row at most
    seg2, nb2, pending2 = plan_pass_host(new_len, None, B, pending, lib=lib)
    first = seg2[seg2[:, 3] == 0]
    assert list(first[0, :2]) == [1, 1] and list(first[1, :2]) == [2, 2]     # carried rows first: clamped 4 -> 1, then 2 as counted
    assert list(first[2, :2]) == [0, 5]                                      # the new pass fills the batch: user 0, min(8 - 3, 10)
    for b in range(nb2):
        rows = seg2[seg2[:, 3] == b]
        assert rows[:, 1].sum() == B and list(rows[:, 2]) == list(np.cumsum(np.r_[0, rows[:-1, 1]]))
    assert np.all(seg2[:, 1] <= new_len[seg2[:, 0]] - 2)
    # a carried user that vanished this pass (fewer than two items left) is dropped, the batch is filled by the others
    seg3, nb3, _ = plan_pass_host([0, 9, 7], None, B, [(0, 5)], lib=lib)
    assert nb3 >= 1 and 0 not in seg3[:, 0]


def _reference_rows(lengths, order, B, n_batches, n_targets=1, shuffle=False, bias=-1.0, pop=None, seed=0):
    """Rows (user, split, [target positions]) per batch as the reference-style generator produces them (models._gen_mini_batch
    with options.SelectTargets: rnn_base.py:394-415, target_selection.py:36-53); item ids encode (user, position)."""
    import random
    from sbr_amd.models import RNNOneHot
    from sbr_amd.options import SelectTargets
    random.seed(seed); np.random.seed(seed)
    ts = SelectTargets(n_targets=n_targets, shuffle=shuffle, bias=bias)
    m = _model(RNNOneHot, max_length=6, batch_size=B, target_selection=ts)
    if bias >= 0:
        ts.keep_prob = _KeepByUserPos(pop)      # (set_dataset derived a table from the fake dataset's popularity)
    m.n_items = 10 ** 7

    def gen():
        while True:
            for u in order:
                if lengths[u] >= 2:
                    yield [[u * 1000 + p, 3.0] for p in range(lengths[u])], str(u)
    m._prepare_input = lambda sequences: [(int(uid), seq[-1][0] % 1000 + 1, [t[0] % 1000 for t in tgt]) for uid, seq, tgt in sequences]
    g = m._gen_mini_batch(gen())
    return [next(g) for _ in range(n_batches)]


class _KeepByUserPos(object):
    """keep_prob table indexed by the encoded item id u * 1000 + p: the popularity of the REAL item at that position."""
    def __init__(self, table):
        self.table = table
    def __getitem__(self, enc):
        return self.table[enc]


@pytest.mark.parametrize("shuffle,bias,NT", [(False, -1.0, 1), (False, 0.7, 1), (True, 0.7, 3), (False, 1.5, 2)])
def test_host_planned_rows_follow_the_reference_procedure(lib, shuffle, bias, NT):
    # sbr_plan_rows_host (the rows of a pass with --target_bias): structure exactly, draws as laws against the reference-style
    # generator run many times -- how many rows a user yields, how often a row is skipped, where the first target lies
    from sbr_amd.engine import plan_rows_host, plan_pass_host
    rng = np.random.default_rng(3)
    lengths = rng.integers(1, 30, size=40)
    lengths[5], lengths[9] = 2, 60
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    items = np.concatenate([u * 1000 + np.arange(L) for u, L in enumerate(lengths)]).astype(np.int32)     # encoded (user, position)
    n_items = int(items.max()) + 1
    keep = None
    if bias >= 0:
        pop = np.ones(n_items); pop[items] = 1 + (items % 7) ** 2                # some positions hold "popular" items
        keep = np.power(pop.min() / pop, bias).astype(np.float32)
    B, order = 16, list(rng.permutation(len(lengths)))
    # ---- structure
    pending, all_rows = None, []
    for p in range(3):
        rows, nb, pending = plan_rows_host(items, offsets, order, B, n_targets=NT, shuffle=shuffle, keep_prob=keep, seed=11 + p, pending=pending, lib=lib)
        assert len(rows) == nb * B and len(pending) < B
        for r in rows:
            u, l, tg = r[0], r[1], r[2:]
            n_rem = lengths[u] - l
            assert 2 <= l < lengths[u] and tg[0] >= 0
            got = tg[tg >= 0]
            assert np.all(got < n_rem) and len(set(got.tolist())) == len(got) and np.all(tg[len(got):] == -1)
            if not shuffle:
                assert np.all(np.diff(got) > 0)                                  # in sequence order
                if keep is None:
                    assert list(got) == list(range(min(NT, n_rem)))              # the next items
        for b in range(nb):                                                      # a user's rows of a batch: contiguous, split points ascending
            blk = rows[b * B:(b + 1) * B]
            for u in set(blk[:, 0].tolist()):
                idx = np.nonzero(blk[:, 0] == u)[0]
                assert np.all(np.diff(blk[idx, 1]) > 0) or len(idx) == 1 or (np.diff(idx) > 1).any()
        all_rows.append(rows)
    if keep is None and not shuffle:      # nothing is skipped: the fill order is sbr_plan_pass_host's
        seg, nb0, _ = plan_pass_host(lengths, order, B, lib=lib)
        rows, nb, _ = plan_rows_host(items, offsets, order, B, n_targets=NT, seed=5, lib=lib)
        assert nb == nb0
        counts = [(int(u), int((rows[b * B:(b + 1) * B, 0] == u).sum())) for b in range(nb) for u in dict.fromkeys(rows[b * B:(b + 1) * B, 0].tolist())]
        assert counts == [(int(u), int(k)) for u, k in seg[:, :2]]
        return
    # ---- laws: the same statistics from the reference-style generator and from the planner, over many passes
    def stats(row_iter):
        n = first_sum = ntg = 0
        per_user = np.zeros(len(lengths))
        for u, l, tg in row_iter:
            n += 1; per_user[u] += 1; ntg += len(tg)
            first_sum += (tg[0] / max(1, lengths[u] - l))                        # relative position of the first target
        return n, per_user / max(1, n), first_sum / max(1, n), ntg / max(1, n)
    ref_rows, dev_rows = [], []
    nbatch, nbs = 10, []                                                         # the first ten batches of a pass, either way
    for sd in range(150):
        rows, nb, _ = plan_rows_host(items, offsets, order, B, n_targets=NT, shuffle=shuffle, keep_prob=keep, seed=100 + sd, lib=lib)
        nbs.append(nb)
        assert nb >= nbatch
        dev_rows += [(int(r[0]), int(r[1]), [int(t) for t in r[2:] if t >= 0]) for r in rows[:nbatch * B]]
    for sd in range(150):
        for batch in _reference_rows(lengths, order, B, nbatch, n_targets=NT, shuffle=shuffle, bias=bias, pop=keep, seed=sd):
            ref_rows += [(u, l, [t - l for t in tg]) for u, l, tg in batch]
    nd, pud, fd, td = stats(dev_rows)
    nr_, pur, fr, tr = stats(ref_rows)
    assert nd == nr_ == 150 * nbatch * B
    assert np.abs(pud - pur).max() < 0.015, np.abs(pud - pur).max()               # share of the rows each user owns
    assert abs(fd - fr) < 0.02 and abs(td - tr) < 0.06, (fd, fr, td, tr)


def test_native_batch_plan_rejects_bad_arguments(lib):
    from sbr_amd.engine import plan_pass_host
    with pytest.raises(ValueError):
        plan_pass_host([5, 6], [0, 7], 4, lib=lib)            # user id out of range
    seg, nb, pend = plan_pass_host([1, 2, 2], None, 4, lib=lib)    # nothing long enough: no rows at all
    assert len(seg) == 0 and nb == 0 and pend == []


def test_bench_initial_parameters_follow_the_init_law():
    # bench.py draws its own random-init weights (the oracle is only its cpu_baseline leg): same zero / non-zero pattern and
    # scales as the oracle's Lasagne-law initialiser, for every cell and for stacked layers
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from sbr_amd.engine import make_config
    for cell, layers, emb in (("GRU", [128], 0), ("LSTM", [20], 0), ("Vanilla", [8], 0), ("LSTM", [512, 512], 0),
                              ("Vanilla", [40, 30], 0), ("Vanilla", [32], 16), ("GRU", [20], 6)):
        N = 50
        ref = O.init_params(cell, layers, N, np.random.default_rng(0), dtype=np.float32, embedding=emb)
        got = bench.initial_parameters(make_config(cell=cell, layers=layers, n_items=N, embedding_size=emb), np.random.default_rng(1))
        assert [g.shape for g in got] == [p.shape for p in ref] and all(g.dtype == np.float32 for g in got)
        for a, b in zip(got, ref):
            assert (np.abs(a).max() > 0) == (np.abs(b).max() > 0)
            if np.abs(b).max() > 0 and b.size > 500:
                assert 0.7 < a.std() / b.std() < 1.4


def test_documented_switches_exist_in_the_sources():
    # every SBR_* switch README.md names is read somewhere in the package sources, and every getenv("SBR_...") of the
    # library is documented (README.md, or DESIGN.md for the experiment-only ones)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    readme = open(os.path.join(root, "README.md")).read()
    design = open(os.path.join(root, "DESIGN.md")).read()
    pkg = os.path.join(root, "sequence-based-recommendations_amd")
    src = ""
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".hip", ".h", ".py")):
                src += open(os.path.join(d, f)).read()
    src += open(os.path.join(root, "include", "sbr_rnn.h")).read()
    named = set(re.findall(r"`(SBR_[A-Z0-9_]+)", readme))
    assert len(named) >= 15
    for n in sorted(named):
        assert n in src, n + " is documented but nothing reads it"
    read = set(re.findall(r'getenv\("(SBR_[A-Z0-9_]+)"\)', src)) | set(re.findall(r'environ\.get\("(SBR_[A-Z0-9_]+)"', src))
    for n in sorted(read):
        assert n in readme or n in design, n + " is read but not documented"
