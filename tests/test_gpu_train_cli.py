"""End-to-end on the GPU: the `train.py -m RNN` mirror (options -> models -> engine) on a tiny synthetic
dataset in the reference's on-disk format: trains, validates, writes reference-format checkpoints, resumes
with --load_last_model, and recommends."""
import glob
import os
import pickle

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_dataset(root, n_users=40, n_items=30, seed=0):
    rng = np.random.default_rng(seed)
    d = os.path.join(root, "data")
    os.makedirs(d)
    os.makedirs(os.path.join(root, "models"))

    def seqs(n):
        out = []
        for u in range(n):
            L = int(rng.integers(6, 15))
            start = int(rng.integers(0, n_items))
            items = [(start + 2 * k + int(rng.integers(0, 2))) % n_items for k in range(L)]   # learnable: mostly +2 steps
            out.append((u, items))
        return out
    sets = {"train": seqs(n_users), "val": seqs(8), "test": seqs(8)}
    trip = []
    for name, ss in sets.items():
        with open(os.path.join(d, name + "_set_sequences"), "w") as f:
            for u, items in ss:
                f.write(str(u) + " " + " ".join("%d %.1f" % (i, 4.0) for i in items) + "\n")
                if name == "train":
                    trip += ["%d %d 4.0" % (u, i) for i in items]
    open(os.path.join(d, "train_set_triplets"), "w").write("\n".join(trip) + "\n")
    with open(os.path.join(d, "stats"), "w") as f:
        f.write("set n_users n_items n_interactions longest_sequence\n")
        for name, ss in (("Full", sum(sets.values(), [])), ("Train", sets["train"]), ("Val", sets["val"]), ("Test", sets["test"])):
            f.write("%s %d %d %d %d\n" % (name, len(ss), n_items, sum(len(s[1]) for s in ss), max(len(s[1]) for s in ss)))
    return root + "/"


@pytest.mark.parametrize("extra", [["--loss", "CCE", "--r_t", "GRU", "--r_l", "16"],
                                   ["--loss", "CCE", "--r_t", "GRU", "--r_l", "16", "--r_emb", "8"],
                                   ["--loss", "CCE", "--r_t", "LSTM", "--r_l", "12-8", "--r_bi"],
                                   ["--loss", "BPR", "--r_t", "LSTM", "--r_l", "12", "--sampling", "8", "--u_m", "adagrad", "--u_l", "0.1"],
                                   # RNNMargin: multi-target hinge loss, popularity-based default target
                                   ["--loss", "hinge", "--r_t", "GRU", "--r_l", "16", "--n_targets", "3", "--balance", "2.0", "--pb",
                                    "--u_m", "adagrad", "--u_l", "0.1"],
                                   ["--loss", "logsig", "--r_t", "GRU", "--r_l", "16", "--n_targets", "2", "--repeated_interactions"]])
def test_train_cli_end_to_end(tmp_path, extra):
    from sbr_amd import train as T
    root = make_dataset(str(tmp_path / "ds"))
    argv = ["-d", root, "-b", "8", "--max_length", "10", "--max_iter", "60", "--progress", "20", "--save", "All"] + extra
    metrics, elapsed, best_file = T.main(argv)
    assert set(metrics) == {"recall", "sps", "user_coverage", "item_coverage", "ndcg", "blockbuster_share"}
    files = sorted(glob.glob(root + "models/*"))
    assert len(files) == 3 and best_file in files
    params = pickle.load(open(files[-1], "rb"))
    assert isinstance(params, list) and all(isinstance(p, np.ndarray) and p.dtype == np.float32 for p in params)
    assert np.all(np.isfinite(np.concatenate([p.ravel() for p in params])))
    # resume: --load_last_model picks the file with the largest epoch count and keeps training
    metrics2, _, _ = T.main(argv + ["--load_last_model", "--max_iter", "20"])
    assert len(glob.glob(root + "models/*")) >= 3
    # recommendations from a loaded checkpoint
    from sbr_amd import options as parse
    from sbr_amd.data import DataHandler
    args = parse.command_parser(parse.predictor_command_parser, parse.training_command_parser, T.early_stopping_command_parser, argv=argv)
    predictor = parse.get_predictor(args)
    dataset = DataHandler(dirname=root)
    predictor.prepare_model(dataset)
    predictor.load(files[-1])
    seq = [[3, 4.0], [5, 4.0], [7, 4.0]]
    rec = predictor.top_k_recommendations(seq, k=5)
    assert len(rec) == 5 and len(set(rec)) == 5
    if "--repeated_interactions" not in extra:
        assert not set(rec) & {3, 5, 7}
    predictor.engine.close()


def test_training_learns_the_synthetic_rule(tmp_path):
    from sbr_amd import train as T
    root = make_dataset(str(tmp_path / "ds"), n_users=120)
    metrics, _, _ = T.main(["-d", root, "-b", "16", "--max_length", "12", "--max_iter", "400", "--progress", "400",
                            "--save", "None", "--r_t", "GRU", "--r_l", "32", "--u_l", "0.01"])
    assert metrics["sps"] > 0.2          # items follow "+2 or +3": far above the 10/30 chance level of sps@10... with k=10


def test_batched_validation_ranks_exactly_like_one_row_calls(tmp_path):
    # the validation pass stacks batch_size users per test_function call; ordered top-k ids and all
    # six metrics must equal the reference-style one-user-at-a-time loop (rnn_base.py:358-371)
    from sbr_amd import options as parse, train as T
    from sbr_amd.data import DataHandler, Evaluator
    root = make_dataset(str(tmp_path / "ds"), n_users=60)
    argv = ["-d", root, "-b", "8", "--max_length", "10", "--r_t", "GRU", "--r_l", "16"]
    args = parse.command_parser(parse.predictor_command_parser, parse.training_command_parser, T.early_stopping_command_parser, argv=argv)
    predictor = parse.get_predictor(args)
    dataset = DataHandler(dirname=root)
    predictor.prepare_model(dataset)
    predictor.train(dataset, max_iter=30, progress=10 ** 9, autosave="None")
    one, ev1 = [], Evaluator(dataset, k=10)
    for batch_input, goal in predictor._gen_mini_batch(dataset.validation_set(epochs=1), test=True):
        ids = predictor.test_function(batch_input)
        one.append(list(ids)); ev1.add_instance(goal, ids)
    many, ev2 = [], Evaluator(dataset, k=10)
    for goals, ids in predictor.batched_test_predictions(dataset.validation_set(epochs=1), k=10):
        for goal, row in zip(goals, ids):
            many.append(list(row)); ev2.add_instance(goal, row)
    assert len(one) == 8 and one == many
    for m in ("average_recall", "sps", "average_ndcg", "user_coverage", "item_coverage", "blockbuster_share"):
        assert getattr(ev1, m)() == getattr(ev2, m)()
    predictor.engine.close()


def test_test_cli_scores_saved_models_like_the_one_by_one_loop(tmp_path):
    # test.py mirror: trains, saves two checkpoints, `python -m sbr_amd.test` finds them by the training options, ranks the
    # test users in batches and appends the results file; metrics equal the reference-style per-user loop
    from sbr_amd import test as Te, train as T, options as parse
    from sbr_amd.data import DataHandler, Evaluator
    root = make_dataset(str(tmp_path / "ds"), n_users=60)
    base = ["-d", root, "-b", "8", "--max_length", "6", "--r_t", "GRU", "--r_l", "16"]
    T.main(base + ["--max_iter", "40", "--progress", "20", "--save", "All"])
    res = Te.main(base + ["--save", "--metrics", "sps,recall,ndcg,item_coverage,user_coverage,blockbuster_share,precision"])
    assert len(res) == 2
    out = glob.glob(root + "results/*")
    assert len(out) == 1 and len(open(out[0]).read().strip().split("\n")) == 2
    assert Te.main(base + ["--save"]) == []                       # both checkpoints already in the results file
    # reference-style loop on the last checkpoint
    args = parse.command_parser(parse.predictor_command_parser, Te.test_command_parser, argv=base)
    predictor = parse.get_predictor(args)
    dataset = DataHandler(dirname=root)
    predictor.prepare_model(dataset)
    predictor.load(res[-1][0])
    ev = Evaluator(dataset, k=10)
    for sequence, user_id in dataset.test_set(epochs=1):
        nv = int(len(sequence) / 2)
        ev.add_instance([i[0] for i in sequence[nv:]], predictor.top_k_recommendations(sequence[:nv], user_id=user_id, k=10))
    for m, v in res[-1][1].items():
        assert ev.metrics[m]() == v, m
    predictor.engine.close()


def test_raw_file_to_trained_model_through_preprocess(tmp_path):
    # the whole user journey with this package only: raw interactions -> sbr_amd.preprocess -> train CLI -> test CLI
    from sbr_amd import preprocess as P, train as T, test as Te
    rng = np.random.default_rng(5)
    lines, t = [], 10 ** 9
    for u in range(70):
        start, L = int(rng.integers(0, 30)), int(rng.integers(8, 20))
        for k in range(L):
            t += int(rng.integers(1, 100))
            lines.append("%d::%d::%d::%d" % (1000 + u, 500 + (start + 2 * k + int(rng.integers(0, 2))) % 30, 4, t))
    rng.shuffle(lines)
    (tmp_path / "ratings.dat").write_text("\n".join(lines) + "\n")
    root = P.main(["-f", str(tmp_path / "ratings.dat"), "--columns", "uirt", "--sep", "::", "--yes", "--min_item_pop", "2"])
    argv = ["-d", root, "-b", "8", "--max_length", "10", "--r_t", "GRU", "--r_l", "16", "--max_iter", "150", "--progress", "75",
            "--save", "All"]
    metrics, _, best_file = T.main(argv)
    assert best_file is not None and 0.0 <= metrics["sps"] <= 1.0 and np.isfinite(metrics["ndcg"])
    res = Te.main(["-d", root, "-b", "8", "--max_length", "10", "--r_t", "GRU", "--r_l", "16", "--save", "--metrics", "sps,recall"])
    assert len(res) == 2 and len(glob.glob(root + "results/rnn_*")) == 1


@pytest.mark.parametrize("extra,native", [
    (["--rf"], True), (["--shuffle_targets"], True), (["--loss", "hinge", "--n_targets", "3"], True),
    (["--loss", "logit", "--n_targets", "2", "--shuffle_targets", "--rf"], True),
    (["--loss", "BPR", "--sampling", "8", "--sampling_bias", "0.5", "--db", "0.3", "--rf"], True),
    (["--n_dropout", "0.1"], True), (["--n_swap", "0.2", "--rf"], True),
    (["--n_shuf", "0.2", "--n_shuf_std", "3", "--n_ratings", "0.3", "--rf", "--n_dropout", "0.2"], True),
    (["--target_bias", "0.5"], True), (["--target_bias", "1.0", "--shuffle_targets", "--n_dropout", "0.1"], True),
    (["--loss", "hinge", "--n_targets", "3", "--target_bias", "0.5"], True)])
def test_which_options_train_on_device_built_batches(tmp_path, extra, native):
    # every batch option is served by the device batch builder (include/sbr_rnn.h: sbr_dataset_set_options,
    # sbr_dataset_noise_pass, sbr_dataset_set_target_bias)
    from sbr_amd import options as parse, train as T
    from sbr_amd.data import DataHandler
    root = make_dataset(str(tmp_path / "ds"))
    argv = ["-d", root, "-b", "8", "--max_length", "10", "--r_t", "GRU", "--r_l", "16"] + extra
    args = parse.command_parser(parse.predictor_command_parser, parse.training_command_parser, T.early_stopping_command_parser, argv=argv)
    predictor = parse.get_predictor(args)
    dataset = DataHandler(dirname=root)
    predictor.prepare_model(dataset)
    predictor.set_dataset(dataset)
    nb = predictor._native_batch_builder(dataset)
    assert (nb is not None) == native
    if nb is not None:
        nb.close()
    predictor.train(dataset, max_iter=12, progress=10 ** 9, autosave="None")      # twelve steps on those batches: finite costs or it raises
    predictor.engine.close()


@pytest.mark.parametrize("extra", [["--clusters", "4", "--sampling", "8"],                                   # RNNCluster, --loss CCE over the samples
                                   ["--clusters", "3", "--loss", "BPR", "--sampling", "8", "--c_sampling", "5", "--cluster_type", "softmax",
                                    "--init_scale", "2.0", "--scale_growing_rate", "1.2", "--u_m", "adagrad", "--u_l", "0.1"],
                                   ["--clusters", "5", "--loss", "Blackout", "--sampling", "10", "--cluster_type", "sigmoid", "--csn", "0.05",
                                    "--ignore_clusters"]])
def test_cluster_cli_end_to_end(tmp_path, extra):
    # `train.py -m RNN --clusters C` (command_parser.py:114-115): both models train, the nine cluster metrics come back, checkpoints
    # carry the two cluster arrays behind the network's list (rnn_cluster.py:510-537), recommendations come from inside the user's
    # cluster (or from the whole catalogue with --ignore_clusters)
    from sbr_amd import train as T
    root = make_dataset(str(tmp_path / "ds"))
    argv = ["-d", root, "-b", "8", "--max_length", "10", "--max_iter", "60", "--progress", "20", "--save", "All", "--r_t", "GRU",
            "--r_l", "16"] + extra
    metrics, elapsed, best_file = T.main(argv)
    assert set(metrics) == {"recall", "cluster_recall", "sps", "cluster_sps", "ignored_items", "assr", "cluster_use", "cluster_use_std",
                            "cluster_size"}
    C = int(extra[1])
    assert len(metrics["cluster_use"]) == C and int(np.sum(metrics["cluster_use"])) == 8      # eight validation users, one cluster each
    assert 1.0 <= metrics["assr"] <= 30.0 * C
    files = sorted(glob.glob(root + "models/*"))
    assert len(files) == 3 and best_file in files and os.path.basename(files[0]).startswith("rnn_clusters%d_sc" % C)
    params = pickle.load(open(files[-1], "rb"))
    R, (Wc,) = params[-2], params[-1]
    assert R.shape == (30, C) and Wc.shape == (16, C) and np.all(np.isfinite(R)) and np.all(np.isfinite(Wc))
    assert all(isinstance(p, np.ndarray) and p.dtype == np.float32 for p in params[:-2])
    from sbr_amd import options as parse
    from sbr_amd.data import DataHandler
    args = parse.command_parser(parse.predictor_command_parser, parse.training_command_parser, T.early_stopping_command_parser, argv=argv)
    predictor = parse.get_predictor(args)
    dataset = DataHandler(dirname=root)
    predictor.prepare_model(dataset)
    predictor.load(files[-1])
    assert np.array_equal(predictor.head.get_params()[0], R)
    assert set(np.concatenate(predictor.clusters).tolist()) == set(range(30))      # prepare_tests: every item lands in at least one cluster
    seq = [[3, 4.0], [5, 4.0], [7, 4.0]]
    rec, n_scored = predictor.top_k_recommendations(seq, k=5)
    assert 1 <= len(rec) <= 5 and len(set(rec)) == len(rec) and not set(rec) & {3, 5, 7}
    assert n_scored == (30 if "--ignore_clusters" in extra else len(predictor.clusters[int(predictor.head.select(1)[0])]))
    predictor.head.close(); predictor.engine.close()
