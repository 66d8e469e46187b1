"""Shared by tools/make_cli_golden.py (which drives the reference) and tests/test_cli_reference_golden.py (which drives
this package): the command lines, the seeds, the deterministic stand-ins for the compiled train / test functions, and the
JSON form of a mini-batch.  Nothing here imports either implementation."""
import numpy as np

D = ["-d", "/tmp/x/"]

# command lines whose parsed options, predictor class and checkpoint names are recorded
CASES = [
    D,
    D + ["-b", "256", "--max_length", "200", "--r_t", "GRU", "--r_l", "128"],
    D + ["--r_t", "LSTM", "--r_l", "100-50", "--r_bi", "--u_m", "adagrad", "--u_l", "0.1"],
    D + ["--r_t", "Vanilla", "--r_l", "32", "--r_emb", "16", "--u_m", "rmsprop", "--u_l", "1.0", "--u_rho", "0.8"],
    D + ["--u_m", "adadelta", "--u_rho", "0.95", "-r", "0.01"],
    D + ["--u_m", "nesterov", "--u_l", "0.5", "--r_l", "64", "--db", "0.3"],
    D + ["--u_m", "adam", "--u_b1", "0.8", "--u_b2", "0.99", "--rf"],
    D + ["--loss", "BPR", "--sampling", "0.2", "--repeated_interactions"],
    D + ["--loss", "TOP1", "--sampling", "64", "--sampling_bias", "0.75", "--db", "0.5"],
    D + ["--loss", "Blackout", "--n_targets", "3", "--shuffle_targets", "--target_bias", "0.5"],
    D + ["--n_dropout", "0.1", "--n_swap", "0.2", "--n_shuf", "0.3", "--n_shuf_std", "2.0", "--n_ratings", "0.4"],
    D + ["--tshuffle", "--extended_set", "--save", "All", "--metrics", "sps,recall", "--progress", "500", "--mpi", "1000",
         "--max_iter", "2000", "--min_iter", "10", "--es_m", "StopAfterN", "--es_n", "4"],
    # RNNMargin (command_parser.py:118-119)
    D + ["--loss", "hinge", "--n_targets", "3", "--balance", "2.0"],
    D + ["--loss", "logit", "--pb", "--min_access", "0.1", "--repeated_interactions", "--r_t", "GRU"],
    D + ["--loss", "logsig", "--balance", "0.5", "--n_targets", "5", "--shuffle_targets"],
    # RNNCluster (command_parser.py:70-77, :114-115)
    D + ["--clusters", "10"],
    D + ["--clusters", "5", "--loss", "BPR", "--sampling", "20", "--c_sampling", "11", "--cluster_type", "softmax", "--init_scale", "2.0",
         "--scale_growing_rate", "1.1", "--max_scale", "20", "--csn", "0.1", "--sampling_bias", "0.5"],
    D + ["--clusters", "3", "--loss", "Blackout", "--cluster_type", "sigmoid", "--ignore_clusters", "--repeated_interactions"],
]

# (argv, seed, number of training batches): the mini-batch streams that are recorded
B = D + ["-b", "8", "--max_length", "6"]
BATCH_CASES = [
    (B, 3, 6),
    (B + ["--rf", "--db", "0.5"], 4, 4),
    (D + ["-b", "5", "--max_length", "30", "--tshuffle", "--extended_set"], 5, 30),         # > 1 epoch: reshuffles
    (B + ["--loss", "BPR", "--sampling", "7"], 6, 4),
    (B + ["--loss", "Blackout", "--sampling", "0.3", "--sampling_bias", "0.75", "--db", "0.2"], 7, 4),
    (B + ["--n_dropout", "0.2", "--n_swap", "0.2", "--n_ratings", "0.3", "--rf"], 8, 4),
    (B + ["--shuffle_targets", "--target_bias", "0.5", "--rand_test_target"], 9, 4),
    (B + ["--n_shuf", "0.3", "--n_shuf_std", "2.0", "--loss", "TOP1", "--sampling", "4"], 10, 4),
    (B + ["--loss", "hinge", "--n_targets", "3", "--balance", "2.0"], 19, 4),
    (B + ["--loss", "logsig", "--n_targets", "2", "--repeated_interactions", "--pb", "--min_access", "0.2"], 20, 4),
    (B + ["--clusters", "4", "--sampling", "5"], 21, 4),
    (B + ["--clusters", "4", "--loss", "TOP1", "--sampling", "6", "--c_sampling", "3", "--sampling_bias", "0.75"], 22, 4),
]

# training-loop runs with the fake functions: what is validated / saved / removed / returned
LOOP_CASES = [
    dict(argv=B + ["--max_iter", "40", "--progress", "5", "--save", "All"], seed=11),
    dict(argv=B + ["--max_iter", "60", "--progress", "1.5", "--save", "Best", "--metrics", "sps,recall"], seed=12),
    dict(argv=B + ["--max_iter", "80", "--progress", "2.", "--mpi", "4", "--save", "Best", "--es_m", "StopAfterN", "--es_n", "2"], seed=13),
    dict(argv=B + ["--max_iter", "30", "--progress", "4", "--min_iter", "10", "--save", "None"], seed=14),
    dict(argv=B + ["--loss", "BPR", "--sampling", "5", "--max_iter", "25", "--progress", "6", "--save", "Best",
                   "--metrics", "blockbuster_share,ndcg"], seed=15),
    dict(argv=B + ["--max_iter", "120", "--progress", "4", "--save", "Best", "--es_m", "WorstTimesX", "--es_x", "1.0",
                   "--es_min_wait", "0.5"], seed=16),
    dict(argv=B + ["--max_iter", "12", "--progress", "4", "--save", "All", "--load_last_model"], seed=17,
         pre=[0.5, 2.25, 10.0]),
    dict(argv=B + ["--max_iter", "9", "--progress", "3", "--save", "Best", "--load_last_model"], seed=18, pre=[]),
]


class FakeFunctions(object):
    """Deterministic stand-ins for the compiled functions: a cost that depends on the batch and the call count, a
    ranking that depends on the row and on how much training happened (so that the validation curves move)."""

    def __init__(self, n_items):
        self.n_items, self.n_train, self.costs = int(n_items), 0, []

    def train_function(self, *batch):
        X, Y = np.asarray(batch[0]), np.asarray(batch[2])
        self.n_train += 1
        cost = ((int(X[..., 0].sum()) * 31 + int(Y.sum()) * 17 + self.n_train) % 1000) / 1000.0 + 0.001
        self.costs.append(cost)
        return cost

    def rank_rows(self, X, mask, k=10):
        X, L = np.asarray(X), np.asarray(mask).sum(1).astype(int)
        n, phase = self.n_items, self.n_train // 3
        rows = []
        for b in range(len(X)):
            base = (int(X[b, L[b] - 1, 0]) * 7 + int(L[b]) + phase) % n
            rows.append([(base + 3 * j) % n for j in range(k)])
        return np.array(rows)


def batch_to_json(batch, p=None):
    """(X, mask, Y, [samples,] pop, exclude) -> plain lists.  `exclude` (B, n_items) is stored as the sorted item ids per
    row; this package hands None there (the engine derives it from X on the device): then it is derived from X / mask."""
    X, mask, Y = np.asarray(batch[0]), np.asarray(batch[1]), np.asarray(batch[2])
    if len(batch) == 5 and (batch[3] is None or np.asarray(batch[3]).ndim == 2):
        return margin_batch_to_json(batch, p)
    if len(batch) == 6 and np.asarray(batch[4]).dtype.kind == "i":      # RNNCluster: (X, mask, Y, samples, cluster_samples, exclude)
        samples, csm, exclude = np.asarray(batch[3]), np.asarray(batch[4]), batch[5]
        assert mask.dtype == np.float32 and Y.dtype == np.int32 and X.dtype == np.int32 and samples.dtype == np.int32 and csm.dtype == np.int32
        if exclude is None:
            ex = [sorted(set(int(i) for i in X[b, :int(mask[b].sum()), 0])) for b in range(len(X))]
        else:
            ex = [np.flatnonzero(np.asarray(exclude)[b]).tolist() for b in range(len(X))]
        return dict(X=X.tolist(), mask=mask.astype(int).tolist(), Y=Y.tolist(), samples=samples.tolist(), cluster_samples=csm.tolist(), exclude=ex)
    samples = np.asarray(batch[3]) if len(batch) == 6 else None
    pop, exclude = np.asarray(batch[-2]), batch[-1]
    assert mask.dtype == np.float32 and pop.dtype == np.float32 and Y.dtype == np.int32 and X.dtype == np.int32
    if exclude is None:
        ex = [sorted(set(int(i) for i in X[b, :int(mask[b].sum()), 0])) for b in range(len(X))]
    else:
        assert np.asarray(exclude).dtype == np.float32
        ex = [np.flatnonzero(np.asarray(exclude)[b]).tolist() for b in range(len(X))]
    out = dict(X=X.tolist(), mask=mask.astype(int).tolist(), Y=Y.tolist(), pop=[float(p) for p in pop], exclude=ex)
    if samples is not None:
        assert samples.dtype == np.int32
        out["samples"] = samples.tolist()
    return out


def margin_batch_to_json(batch, p=None):
    """RNNMargin: the reference hands (X, mask, Y (B,N), weight (B,N), exclude (B,N)); this package hands (X, mask, targets (B,NT),
    None, None) and its engine derives the dense pair on the device.  Common form: the dense target / weight rows reduced to
    what determines them (the non-zero targets, the ids with weight -1 and 0, the false-positive weight); for this package
    they are rebuilt here from the positives, the row's inputs and the predictor's options, following rnn_margin.py:112-147."""
    X, mask = np.asarray(batch[0]), np.asarray(batch[1])
    if batch[3] is None:
        N, tg = int(p.n_items), np.asarray(batch[2])
        dflt = np.asarray(p._default_target(), dtype=np.float64)
        Y, W = np.zeros((len(X), N)), np.zeros((len(X), N))
        for i in range(len(X)):
            in_seq = [int(v) for v in X[i, :int(mask[i].sum()), 0]]
            t = [int(v) for v in tg[i] if v >= 0]
            W[i, :] = p.balance * len(t) / float(N - len(t) - len(in_seq))
            W[i, t] = -1
            Y[i, :] = dflt
            Y[i, t] = 1
            if p.interactions_are_unique:
                W[i, in_seq] = 0
                Y[i, in_seq] = 0
    else:
        Y, W = np.asarray(batch[2], dtype=np.float64), np.asarray(batch[3], dtype=np.float64)
        assert np.asarray(batch[2]).dtype == np.float32 and np.asarray(batch[3]).dtype == np.float32
    return dict(X=X.tolist(), mask=mask.astype(int).tolist(),
                Y=[[[int(i), round(float(v), 5)] for i, v in enumerate(row) if v != 0.0] for row in Y],
                W=[dict(minus_one=np.flatnonzero(row == -1).tolist(), zero=np.flatnonzero(row == 0).tolist(),
                        fp=round(float(np.max(row)), 6)) for row in W])


# `test.py` runs: (argv after -d ROOT, epochs of the checkpoint files present in models/)
TEST_CASES = [
    (["-b", "8", "--max_length", "6", "--save"], [0.115, 0.23, 1.5]),
    (["-b", "4", "--max_length", "3", "--save", "-k", "5", "--metrics", "sps,ndcg,precision,recall"], [2.0, 0.5]),
    (["-b", "8", "--max_length", "30", "--save", "--save_rank", "--metrics", "sps,item_coverage,assr"], [0.75]),
    (["-b", "3", "--max_length", "8", "--save", "--repeated_interactions", "--loss", "BPR", "--sampling", "5",
      "--metrics", "sps,recall,item_coverage,user_coverage,blockbuster_share,ndcg"], [3.0, 1.0]),
    (["-b", "8", "--max_length", "6", "--save", "-i", "2", "--rf"], [2, 4]),
]


class FakeScores(object):
    """stand-in for the compiled predict function: distinct scores per item that depend on the last item fed, on the
    number of items fed and on which checkpoint was `loaded`"""

    def __init__(self, n_items):
        self.n_items, self.phase, self.loaded = int(n_items), 0, []

    def load(self, filename):
        import re
        self.loaded.append(filename.split("/")[-1])
        self.phase = int(round(float(re.search(r"_ne([0-9]+(\.[0-9]+)?)_", filename).group(1)) * 1000)) % 17

    def predict_function(self, X, mask):
        X, L = np.asarray(X), np.asarray(mask).sum(1).astype(int)
        n = self.n_items
        out = np.zeros((len(X), n), dtype=np.float32)
        for b in range(len(X)):
            last = int(X[b, L[b] - 1, 0]) if L[b] else 0
            out[b] = (np.arange(n) * 13 + last * 7 + int(L[b]) + self.phase) % n       # 13 and n coprime: a permutation
        return out
