"""GPU tests of the native batch builder (sbr_dataset_* / sbr_build_batch): every built batch is
checked against the reference's batch semantics (rnn_base.py:394-415, rnn_one_hot.py:83-106, rnn_sampling.py:159-194)
on data whose item ids encode (user, position), and the split-point / negative-sample distributions against the laws
the reference draws from (random.sample, np.random.choice, bisect over cumsum(pop**bias))."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def encoded_dataset(lengths):
    """items[u][p] = 1 + u * 64 + p : a row's target reveals (user, split point)."""
    items = [1 + u * 64 + np.arange(L) for u, L in enumerate(lengths)]
    offsets = np.concatenate([[0], np.cumsum(lengths)])
    return np.concatenate(items).astype(np.int32), offsets.astype(np.int64), 1 + 64 * len(lengths)


def make(lengths, B, T, loss="CCE", S=0, local_batch=None, row_offset=0):
    from sbr_amd.engine import RNNEngine, DeviceDataset
    items, offsets, n_items = encoded_dataset(lengths)
    eng = RNNEngine(cell="GRU", layers=[16], n_items=n_items, max_length=T, batch_size=B, loss=loss, n_samples=S,
                    local_batch=local_batch, row_offset=row_offset)
    return eng, DeviceDataset(eng, items, offsets, n_items), items, offsets


def check_batch(cur, seg_rows, items, offsets, lengths, T, pop_db=None, row_offset=0):
    """seg_rows: the plan's (user, k, row0) for this batch."""
    X, lens, tgt, pop = cur["X"][:, :, 0], cur["lengths"], cur["target"], cur["pop"]
    want_user = np.concatenate([[u] * k for u, k, _ in seg_rows])
    B = len(lens)
    for b in range(B):
        g = row_offset + b
        u, l = (tgt[b] - 1) // 64, (tgt[b] - 1) % 64
        assert u == want_user[g]
        assert 2 <= l < lengths[u]                                            # random.sample(range(2, len), k)
        start = max(0, l - T)
        seq = items[offsets[u]:offsets[u + 1]]
        assert lens[b] == l - start
        assert np.array_equal(X[b, :lens[b]], seq[start:l]) and not X[b, lens[b]:].any()
        assert pop[b] == (1.0 if pop_db is None else pop_db[tgt[b]])
    ls = (tgt - 1) % 64
    for u, k, r0 in seg_rows:                                                 # sorted, distinct within a user
        seg = ls[max(r0 - row_offset, 0):max(r0 + k - row_offset, 0)]
        assert np.all(np.diff(seg) > 0)


def test_built_batches_follow_the_reference_semantics():
    rng = np.random.default_rng(0)
    lengths = rng.integers(1, 40, size=60)
    lengths[5], lengths[17] = 2, 63
    B, T = 32, 10
    eng, ds, items, offsets = make(lengths, B, T)
    pop_db = rng.uniform(0.5, 2.0, size=1 + 64 * len(lengths)).astype(np.float32)
    ds.set_tables(pop_db, None)
    order = rng.permutation(len(lengths)).astype(np.int32)
    nb = ds.plan_pass(order, B)
    seg = ds.segments()
    assert nb >= 5
    for b in range(nb):
        eng.build_batch(ds, b, seed=1234 + b)
        rows = [(int(u), int(k), int(r0)) for u, k, r0, bb in seg if bb == b]
        check_batch(eng.current_batch(), rows, items, offsets, lengths, T, pop_db)
    # the same (batch, seed) rebuilds the same batch; another seed another one
    eng.build_batch(ds, 0, seed=7); a = eng.current_batch()["target"].copy()
    eng.build_batch(ds, 0, seed=7); assert np.array_equal(a, eng.current_batch()["target"])
    eng.build_batch(ds, 0, seed=8); assert not np.array_equal(a, eng.current_batch()["target"])
    with pytest.raises(ValueError):
        eng.build_batch(ds, nb, seed=0)                                       # outside the planned pass
    ds.close(); eng.close()


def test_split_points_are_uniform_without_replacement():
    # one user of 34 items fills the batch alone: k = 8 of the 32 candidates {2..33}; every candidate must come up
    # with probability k/n, pairs never repeat inside a batch (checked above), chi-square against the uniform law
    lengths = [34]
    B, T, n_draws = 8, 6, 1500
    eng, ds, items, offsets = make(lengths, B, T)
    ds.plan_pass(None, B)
    counts = np.zeros(34)
    for i in range(n_draws):
        eng.build_batch(ds, 0, seed=99 + i)
        counts[(eng.current_batch()["target"] - 1) % 64] += 1
    assert counts[:2].sum() == 0 and counts.sum() == n_draws * B
    exp = n_draws * B / 32.0
    chi2 = ((counts[2:] - exp) ** 2 / exp).sum()
    assert chi2 < 70.0, chi2                                                   # 31 dof: p(chi2 > 70) ~ 1e-4
    ds.close(); eng.close()


def test_negative_samples_follow_uniform_and_popularity_laws():
    lengths = [20] * 8
    B, T, S = 16, 6, 64
    eng, ds, items, offsets = make(lengths, B, T, loss="BPR", S=S)
    n_items = 1 + 64 * len(lengths)
    ds.plan_pass(None, B)
    got = []
    for i in range(300):
        eng.build_batch(ds, 0, seed=5 + i)
        cur = eng.current_batch()
        got.append(cur["samples"].copy())
        assert len(cur["target"]) == B                                         # sampled heads: all global rows
    got = np.concatenate(got)
    assert got.min() >= 0 and got.max() < n_items
    h = np.bincount(got, minlength=n_items)
    exp = len(got) / n_items
    assert ((h - exp) ** 2 / exp).sum() < n_items + 6 * np.sqrt(2 * n_items)   # uniform: np.random.choice(n_items, S)
    # popularity ** sampling_bias by inverse CDF (rnn_sampling.py:159-163): mass only where popularity > 0
    pop = np.zeros(n_items); pop[10:20] = np.arange(1, 11)
    ds.set_tables(None, np.cumsum(pop ** 0.5))
    got = []
    for i in range(300):
        eng.build_batch(ds, 0, seed=1000 + i)
        got.append(eng.current_batch()["samples"].copy())
    got = np.concatenate(got)
    assert got.min() >= 10 and got.max() < 20
    freq = np.bincount(got, minlength=n_items)[10:20] / len(got)
    want = pop[10:20] ** 0.5 / (pop[10:20] ** 0.5).sum()
    assert np.abs(freq - want).max() < 0.02
    ds.close(); eng.close()


def test_data_parallel_ranks_build_their_rows_of_the_same_global_batch():
    rng = np.random.default_rng(3)
    lengths = rng.integers(3, 30, size=40)
    B, T = 32, 8
    full, dsf, items, offsets = make(lengths, B, T)
    dsf.plan_pass(None, B)
    full.build_batch(dsf, 1, seed=42)
    whole = full.current_batch()
    for r in range(2):
        eng, ds, _, _ = make(lengths, B, T, local_batch=16, row_offset=16 * r)
        ds.plan_pass(None, B)
        eng.build_batch(ds, 1, seed=42)
        part = eng.current_batch()
        for k in ("X", "lengths", "target", "pop"):
            assert np.array_equal(part[k], whole[k][16 * r:16 * (r + 1)]), k
        ds.close(); eng.close()
    dsf.close(); full.close()


def test_training_from_native_batches_matches_host_batches_in_quality(tmp_path, monkeypatch):
    # same synthetic rule-following dataset as test_gpu_train_cli: both builders must learn it
    from test_gpu_train_cli import make_dataset
    from sbr_amd import train as Tr
    res = {}
    for native in ("1", "0"):
        monkeypatch.setenv("SBR_NATIVE_BATCHES", native)
        root = make_dataset(str(tmp_path / ("ds" + native)), n_users=120)
        metrics, _, _ = Tr.main(["-d", root, "-b", "16", "--max_length", "12", "--max_iter", "400", "--progress", "400",
                                 "--save", "None", "--r_t", "GRU", "--r_l", "32", "--u_l", "0.01"])
        res[native] = metrics["sps"]
    assert res["1"] > 0.2 and res["0"] > 0.2, res


def test_lagged_train_step_returns_the_same_costs_one_call_late():
    # the training loop's pipelined form (sbr_train_step_lagged): same steps, each cost handed back one call later
    rng = np.random.default_rng(3)
    lengths = rng.integers(3, 40, size=40)
    B, T, n = 16, 12, 9

    def run(lagged):
        eng, ds, _, _ = make(lengths, B, T)
        try:
            from oracle import rnn_oracle as O
            eng.set_all_param_values(O.init_params("GRU", [16], 1 + 64 * len(lengths), np.random.default_rng(7), dtype=np.float32))
            nb = ds.plan_pass(None, B)
            costs = []
            for i in range(n):
                eng.build_batch(ds, i % nb, seed=50 + i)
                c = eng.train_step_lagged() if lagged else eng.train_step()
                if c is not None:
                    costs.append(c)
            if lagged:
                assert len(costs) == n - 1
                costs.append(eng.flush_lagged())
                assert eng.flush_lagged() is None
            return np.array(costs), eng.get_all_param_values()
        finally:
            ds.close(); eng.close()

    c0, p0 = run(False)
    c1, p1 = run(True)
    # (not bitwise: the scatter adds the few segments that straddle chunk seams with float atomics)
    assert np.allclose(c0, c1, rtol=1e-5, atol=0)
    assert all(np.allclose(a, b, rtol=1e-4, atol=1e-6) for a, b in zip(p0, p1))
    assert c0[-1] < c0[0]


def test_rating_feature_multiple_targets_and_shuffled_targets():
    """The options the device builder covers beyond the defaults (sbr_dataset_set_options): --rf (second index = n_items +
    rating one-hot index, rnn_base.py:590-642), --n_targets of the multi-target losses (the first min(k, remaining) items
    after the split, -1 behind them; target_selection.py:53), --shuffle_targets (a uniform random subset of the whole
    remaining sequence, :45-46)."""
    from sbr_amd.engine import RNNEngine, DeviceDataset
    rng = np.random.default_rng(3)
    lengths = rng.integers(3, 30, size=50)
    items, offsets, n_items = encoded_dataset(lengths)
    ratings = (0.5 + 0.5 * ((items * 7) % 10)).astype(np.float32)              # 0.5 .. 5.0 in halves: every one-hot index occurs
    B, T, NT = 32, 8, 3
    eng = RNNEngine(cell="GRU", layers=[16], n_items=n_items, max_length=T, batch_size=B, loss="hinge", n_targets=NT,
                    n_feat=2, input_size=n_items + 10)
    ds = DeviceDataset(eng, items, offsets, n_items)
    ds.set_options(ratings, False)
    nb = ds.plan_pass(None, B)
    assert nb >= 3
    for b in range(nb):
        eng.build_batch(ds, b, seed=50 + b)
        cur = eng.current_batch()
        X, lens, tgt = cur["X"], cur["lengths"], cur["target"].reshape(B, NT)
        for r in range(B):
            u, l = (tgt[r, 0] - 1) // 64, (tgt[r, 0] - 1) % 64
            seq, rat = items[offsets[u]:offsets[u + 1]], ratings[offsets[u]:offsets[u + 1]]
            start = max(0, l - T)
            assert np.array_equal(X[r, :lens[r], 0], seq[start:l])
            want_idx = (np.floor(rat[start:l] * 2 + 0.5).astype(int) - 1) % 10
            assert np.array_equal(X[r, :lens[r], 1], n_items + want_idx)        # python-2 round(), one-hot on a scale of ten
            assert not X[r, lens[r]:].any()
            k = min(NT, lengths[u] - l)
            assert np.array_equal(tgt[r, :k], seq[l:l + k]) and np.all(tgt[r, k:] == -1)
    # shuffled targets: every target lies in the remaining sequence, the targets of a row are distinct positions, and over
    # many seeds the FIRST target is uniform over the remaining items
    ds.set_options(ratings, True)
    eng.build_batch(ds, 0, seed=1000)
    base = eng.current_batch()["target"].reshape(B, NT).copy()
    eng.build_batch(ds, 0, seed=1000)                                           # deterministic per seed
    assert np.array_equal(base, eng.current_batch()["target"].reshape(B, NT))
    hist, want = np.zeros(4), np.zeros(4)
    n_draws = 0
    for sd in range(400):
        eng.build_batch(ds, 0, seed=sd)
        cur = eng.current_batch()
        tgt, lens, X = cur["target"].reshape(B, NT), cur["lengths"], cur["X"]
        for r in range(B):
            u = (X[r, 0, 0] - 1) // 64
            l = (X[r, lens[r] - 1, 0] - 1) % 64 + 1                             # the split point: one past the last input item
            pos = (tgt[r][tgt[r] >= 0] - 1) % 64
            users = (tgt[r][tgt[r] >= 0] - 1) // 64
            assert np.all(users == u) and np.all(pos >= l) and np.all(pos < lengths[u])
            assert len(set(pos.tolist())) == len(pos) == min(NT, lengths[u] - l)
            n_rem = lengths[u] - l
            if n_rem >= 4:
                hist[int((pos[0] - l) * 4 // n_rem)] += 1                       # quartile of the remaining sequence
                want += np.bincount(np.arange(n_rem) * 4 // n_rem, minlength=4) / n_rem      # what a uniform draw gives
                n_draws += 1
    assert n_draws > 2000 and np.all(np.abs(hist - want) / n_draws < 0.04), (hist / n_draws, want / n_draws)
    assert hist[3] / n_draws > 0.15                                             # next-item targets would put everything in the first quartile
    ds.close(); eng.close()


def _host_noise(seqs, **kw):
    """The reference-style host implementation (options.SequenceNoise, sequence_noise.py:52-94) over a list of sequences of
    [item, rating] pairs; returns the noised sequences (None where the user is skipped)."""
    from sbr_amd.options import SequenceNoise
    nz = SequenceNoise(**kw)
    out = []
    for s in seqs:
        got = list(nz(iter([([list(p) for p in s], 0)])))      # (a skipped user yields nothing)
        out.append(got[0][0] if got else None)
    return out


def test_sequence_noise_laws():
    """sbr_dataset_noise_pass against the reference's procedure (sequence_noise.py:52-94), as laws over many passes: the device
    draws are hashed, not the reference's Mersenne stream."""
    from sbr_amd.engine import RNNEngine, DeviceDataset
    rng = np.random.default_rng(11)
    lengths = rng.integers(2, 60, size=60)
    items, offsets, n_items = encoded_dataset(lengths)
    nnz = len(items)
    ratings = (0.5 + 0.5 * ((items * 7) % 10)).astype(np.float32)
    eng = RNNEngine(cell="GRU", layers=[16], n_items=n_items, max_length=8, batch_size=32, loss="CCE", n_feat=2, input_size=n_items + 10)
    ds = DeviceDataset(eng, items, offsets, n_items)
    ds.set_options(ratings, False)
    idx0 = ((np.floor(ratings * 2 + 0.5).astype(int) - 1) % 10)
    seqs = [[[int(items[offsets[u] + i]), float(ratings[offsets[u] + i])] for i in range(lengths[u])] for u in range(len(lengths))]
    np.random.seed(5)

    def device(seed, **kw):
        ds.noise_pass(seed=seed, **kw)
        it, rt, ln = ds.current_sequences(nnz)
        return [(it[offsets[u]:offsets[u] + ln[u]], rt[offsets[u]:offsets[u] + ln[u]]) for u in range(len(lengths))], ln

    # ---- no noise: the sequences as uploaded
    got, ln = device(1)
    assert np.array_equal(ln, lengths) and all(np.array_equal(g[0], items[offsets[u]:offsets[u + 1]]) for u, g in enumerate(got))
    # ---- dropout: an order-preserving subsequence; every item kept with probability 1 - p; fewer than two left -> skipped
    kept = tot = skipped = short = 0
    for sd in range(40):
        got, ln = device(100 + sd, dropout=0.3)
        for u, (it, rt) in enumerate(got):
            pos = (it - 1) % 64
            assert np.all((it - 1) // 64 == u) and np.all(np.diff(pos) > 0)
            assert np.array_equal(rt, idx0[offsets[u] + pos])                   # the rating travels with its item
            if ln[u] == 0:
                skipped += 1
            else:
                assert ln[u] >= 2
                kept += ln[u]; tot += lengths[u]
    assert abs(kept / tot - 0.7) < 0.02, kept / tot                             # (skipped users are short ones: a small bias upward)
    assert skipped > 0                                                          # users of 2 - 3 items lose one often enough
    # ---- swap / shuffle: permutations of the user's own sequence; displacement statistics match the host procedure's
    def displacement_hist(noised_positions):
        h = np.zeros(5)
        for pos in noised_positions:
            d = np.abs(pos - np.arange(len(pos)))
            h += np.bincount(np.digitize(d, [1, 2, 4, 8]), minlength=5)
        return h / h.sum()

    for kw in (dict(swap=0.25), dict(shuf=0.2, shuf_std=5.0), dict(swap=0.1, shuf=0.1, shuf_std=3.0)):
        dev_pos, host_pos = [], []
        for sd in range(30):
            got, ln = device(200 + sd, **kw)
            assert np.array_equal(ln, lengths)
            for u, (it, rt) in enumerate(got):
                pos = (it - 1) % 64
                assert sorted(pos.tolist()) == list(range(lengths[u]))          # a permutation
                assert np.array_equal(rt, idx0[offsets[u] + pos])
                dev_pos.append(pos)
            for u, s in enumerate(_host_noise(seqs, **kw)):
                host_pos.append((np.array([p[0] for p in s]) - 1) % 64)
        hd, hh = displacement_hist(dev_pos), displacement_hist(host_pos)
        assert np.all(np.abs(hd - hh) < 0.012), (kw, hd, hh)
        assert hd[0] < 0.95                                                     # something moved
    if True:      # swap: an item moves at most one place
        got, _ = device(999, swap=0.4)
        assert all(np.all(np.abs((it - 1) % 64 - np.arange(len(it))) <= 1) for it, _ in got)
    # ---- rating perturbation: +- one half-star index, clamped to ratings 1 .. 5 (index 1 .. 9); items untouched
    up = down = same = 0
    exp_up = exp_down = 0.0
    for sd in range(30):
        got, ln = device(300 + sd, ratings_perturb=0.3)
        for u, (it, rt) in enumerate(got):
            assert np.array_equal(it, items[offsets[u]:offsets[u + 1]])
            d = rt - idx0[offsets[u]:offsets[u + 1]]
            assert np.all(np.abs(d) <= 1) and np.all(rt[d != 0] >= 1) and np.all(rt <= 9)
            up += int((d == 1).sum()); down += int((d == -1).sum()); same += int((d == 0).sum())
            i0 = idx0[offsets[u]:offsets[u + 1]]
            # a perturbed rating changes unless the clamp undoes it: up at index 9, down at index 1; down at index 0 (rating
            # 0.5) is max(1, 0) = rating 1: index 1, a step UP
            exp_up += 0.3 * (0.5 * (i0 < 9).sum() + 0.5 * (i0 == 0).sum())
            exp_down += 0.3 * 0.5 * (i0 >= 2).sum()
    n = up + down + same
    assert abs(up - exp_up) / n < 0.008 and abs(down - exp_down) / n < 0.008, (up / n, exp_up / n, down / n, exp_down / n)
    ds.close(); eng.close()


def test_batches_of_a_noised_pass_read_the_noised_copy():
    """With noise the pass is planned on the noised lengths and every row is cut from the noised copy; rows carried over from
    the previous pass are cut from the NEW copy (their count clamped to what it offers)."""
    from sbr_amd.engine import RNNEngine, DeviceDataset
    rng = np.random.default_rng(13)
    lengths = rng.integers(2, 40, size=70)
    items, offsets, n_items = encoded_dataset(lengths)
    B, T = 32, 8
    eng = RNNEngine(cell="GRU", layers=[16], n_items=n_items, max_length=T, batch_size=B, loss="CCE")
    ds = DeviceDataset(eng, items, offsets, n_items)
    rows_total = 0
    for p in range(3):
        ds.noise_pass(dropout=0.3, swap=0.2, seed=40 + p)
        it, _, ln = ds.current_sequences(len(items))
        nb = ds.plan_pass(None, B)
        seg = ds.segments()
        assert nb >= 2
        for u, k in zip(seg[:, 0], seg[:, 1]):
            assert 1 <= k <= ln[u] - 2                                          # never more rows than the noised sequence offers
        assert not np.isin(np.nonzero(ln == 0)[0], seg[:, 0]).any()             # skipped users own no rows
        for b in range(nb):
            eng.build_batch(ds, b, seed=70 + 10 * p + b)
            cur = eng.current_batch()
            X, lens, tgt = cur["X"], cur["lengths"], cur["target"]
            for r in range(B):
                u = (tgt[r] - 1) // 64
                seq = it[offsets[u]:offsets[u] + ln[u]]
                where = np.nonzero(seq == tgt[r])[0]
                assert len(where) == 1                                          # (items of a user are distinct in this dataset)
                l = int(where[0])
                assert l >= 2 and np.array_equal(X[r, :lens[r], 0], seq[max(0, l - T):l]) and not X[r, lens[r]:].any()
                rows_total += 1
    assert rows_total >= 3 * 2 * B
    # a training loop on noised passes runs and learns
    from oracle import rnn_oracle as O
    eng.set_all_param_values(O.init_params("GRU", [16], n_items, np.random.default_rng(7), dtype=np.float32))
    costs = []
    for p in range(6):
        ds.noise_pass(dropout=0.1, shuf=0.1, shuf_std=2.0, seed=90 + p)
        nb = ds.plan_pass(None, B)
        for b in range(nb):
            eng.build_batch(ds, b, seed=500 + 31 * p + b)
            costs.append(eng.train_step())
    assert np.all(np.isfinite(costs)) and np.mean(costs[-4:]) < np.mean(costs[:4])
    ds.close(); eng.close()


def test_target_bias_rows_planned_on_the_host_are_packed_as_planned():
    """--target_bias (sbr_dataset_set_target_bias): the rows of a pass come from the host planner (its laws against the
    reference's procedure: tests/test_host_cpu.py); the device packs exactly those rows -- inputs cut at the planned split
    point, the planned target positions, popularity weight of the first target."""
    from sbr_amd.engine import RNNEngine, DeviceDataset, plan_rows_host
    rng = np.random.default_rng(17)
    lengths = rng.integers(2, 40, size=60)
    items, offsets, n_items = encoded_dataset(lengths)
    pop = 1.0 + (np.arange(n_items) % 5) ** 2
    keep = np.power(pop.min() / pop, 0.8).astype(np.float32)
    pop_db = np.power(pop, 0.5).astype(np.float32)
    B, T, NT = 32, 8, 2
    eng = RNNEngine(cell="GRU", layers=[16], n_items=n_items, max_length=T, batch_size=B, loss="hinge", n_targets=NT)
    ds = DeviceDataset(eng, items, offsets, n_items)
    ds.set_tables(pop_db, None)
    for shuffle in (False, True):
        ds.set_options(None, shuffle)
        ds.set_target_bias(keep, NT, seed=77)
        pending = None
        for p in range(2):
            nb = ds.plan_pass(None, B)
            want, nb_w, pending = plan_rows_host(items, offsets, None, B, n_targets=NT, shuffle=shuffle, keep_prob=keep,
                                                 seed=77 + 0x9E37 * (p + 1), pending=pending)
            assert nb == nb_w and nb >= 3
            seg = ds.segments()
            assert seg[:, 1].sum() == nb * B
            for b in range(nb):
                eng.build_batch(ds, b, seed=5)
                cur = eng.current_batch()
                X, lens, tgt, w = cur["X"], cur["lengths"], cur["target"].reshape(B, NT), cur["pop"]
                for r in range(B):
                    u, l, tp = want[b * B + r, 0], want[b * B + r, 1], want[b * B + r, 2:]
                    seq = items[offsets[u]:offsets[u + 1]]
                    assert np.array_equal(X[r, :lens[r], 0], seq[max(0, l - T):l]) and lens[r] == min(T, l)
                    assert np.array_equal(tgt[r], np.where(tp >= 0, seq[np.minimum(l + np.maximum(tp, 0), len(seq) - 1)], -1))
                    assert abs(w[r] - pop_db[seq[l + tp[0]]]) < 1e-6
    # with the bias on, popular items are targets less often than their share of the candidates
    ds.set_options(None, False)
    ds.set_target_bias(keep, NT, seed=5)
    hits = np.zeros(5); cand = np.zeros(5)
    for p in range(6):
        nb = ds.plan_pass(None, B)
        for b in range(nb):
            eng.build_batch(ds, b, seed=9)
            t0 = eng.current_batch()["target"].reshape(B, NT)[:, 0]
            hits += np.bincount(t0 % 5, minlength=5)
    assert hits[0] / hits.sum() > 0.30 and hits[4] / hits.sum() < 0.16, hits / hits.sum()      # keep_prob 1 against 17 ** -0.8 = 0.10
    ds.set_target_bias(None)
    assert ds.plan_pass(None, B) >= 3                                           # back to device-drawn rows
    ds.close(); eng.close()
