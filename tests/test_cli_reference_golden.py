"""The host side of `train.py -m RNN` against known answers recorded from the REFERENCE's own code
(tests/golden/cli_reference.json, written by tools/make_cli_golden.py from /root/reference): parsed options, the
predictor they build, checkpoint names, the seeded training / validation mini-batch streams for every option that shapes
them, and the control flow of train() -- when it validates, what it saves and deletes, when early stopping ends it, what
it returns -- with the compiled functions replaced by the same deterministic fakes on both sides (tests/cli_cases.py).
No GPU: the engine is a stand-in object here; the device side is covered by the -m gpu tests."""
import json
import os
import random
import shutil

import numpy as np
import pytest

from cli_cases import CASES, BATCH_CASES, LOOP_CASES, TEST_CASES, FakeFunctions, FakeScores, batch_to_json
from sbr_amd import options as parse, train as T
from sbr_amd.data import DataHandler

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "cli_reference.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def root(tmp_path_factory):
    d = tmp_path_factory.mktemp("ds")
    os.makedirs(d / "data")
    src = os.path.join(HERE, "golden", "preprocess")
    for n in os.listdir(src):
        if n not in ("ARGS", "ratings.dat"):
            shutil.copy(os.path.join(src, n), d / "data" / n)
    return str(d) + "/"


class FakeEngine(object):
    """what models.py asks of the engine in train(): the three callables of the boundary + the parameter list"""

    def __init__(self, fake):
        self.fake = fake

    def train_function(self, *batch):
        return self.fake.train_function(*batch)

    def test_function(self, inputs, k=10, exclude_seen=True):
        return self.fake.rank_rows(inputs[0], inputs[1], k)

    def get_all_param_values(self):
        return [np.zeros(1, np.float32)]


def build(argv, root=None):
    args = parse.command_parser(parse.predictor_command_parser, parse.training_command_parser,
                                T.early_stopping_command_parser, argv=list(argv))
    p = parse.get_predictor(args)
    dataset = None
    if root is not None:
        dataset = DataHandler(dirname=root, extended_training_set=args.extended_set, shuffle_training=args.tshuffle)
        p.n_items = dataset.n_items
        if hasattr(p, "sampling"):
            p.effective_sampling = int(p.sampling * p.n_items) if p.sampling < 1 else int(p.sampling)
        p.set_dataset(dataset)
    return args, p, dataset


@pytest.mark.parametrize("i", range(len(CASES)))
def test_options_predictor_and_checkpoint_names(gold, i):
    g = gold["cases"][i]
    assert g["argv"] == CASES[i]
    args, p, _ = build(CASES[i])
    mine = {k: (v if not isinstance(v, float) or np.isfinite(v) else repr(v)) for k, v in vars(args).items()}
    shared = set(mine) & set(g["args"])
    assert len(shared) >= 40                                   # every option of the RNN path (the rest belongs to other models)
    for k in sorted(shared):
        assert mine[k] == g["args"][k] and type(mine[k]) is type(g["args"][k]), k
    assert type(p).__name__ == g["cls"] and p.name == g["name"]
    assert p._get_model_filename(1.5) == g["file_1p5"]
    assert p._get_model_filename("*") == g["file_glob"]
    assert p._get_model_filename(round(12.34567, 3)) == g["file_round"]


@pytest.mark.parametrize("i", range(len(BATCH_CASES)))
def test_mini_batch_streams(gold, root, i):
    argv, seed, n_train = BATCH_CASES[i]
    g = gold["batches"][i]
    assert g["argv"] == argv and g["seed"] == seed
    args, p, dataset = build(argv, root)
    random.seed(seed); np.random.seed(seed)
    gen = p._gen_mini_batch(p.sequence_noise(dataset.training_set()))
    for k in range(n_train):
        assert batch_to_json(next(gen), p) == g["train"][k], "training batch %d" % k
    assert float(dataset.training_set.epochs) == g["epochs"]
    random.seed(seed + 1); np.random.seed(seed + 1)
    got = [[batch_to_json(b, p), [int(x) for x in goal]] for b, goal in p._gen_mini_batch(dataset.validation_set(epochs=1), test=True)]
    assert got == g["test"]


@pytest.mark.parametrize("i", range(len(LOOP_CASES)))
def test_training_loop_control_flow(gold, root, tmp_path, monkeypatch, i):
    case, g = LOOP_CASES[i], gold["loops"][i]
    assert g["argv"] == case["argv"] and g["seed"] == case["seed"]
    monkeypatch.setenv("SBR_NATIVE_BATCHES", "0")              # the reference-style host generator: same random stream
    work = str(tmp_path) + "/"
    args, p, dataset = build(case["argv"], root)
    for ne in case.get("pre", []):
        open(work + p._get_model_filename(ne), "w").close()
    fake = FakeFunctions(dataset.n_items)
    p.engine = FakeEngine(fake)
    log = []
    save = p.save
    p.save = lambda fn: (log.append(["save", os.path.basename(fn)]), save(fn))
    p.load = lambda fn: log.append(["load", os.path.basename(fn)])
    validate = p._compute_validation_metrics

    def logged_validation(metrics):
        log.append(["validate", fake.n_train])
        metrics = validate(metrics)
        log.append(["metrics", {k: float(v[-1]) for k, v in metrics.items()}])
        return metrics
    p._compute_validation_metrics = logged_validation
    remove = os.remove
    monkeypatch.setattr(os, "remove", lambda fn: (log.append(["remove", os.path.basename(fn)]), remove(fn)))
    random.seed(case["seed"]); np.random.seed(case["seed"])
    metrics, _, best = p.train(dataset, save_dir=work, time_based_progress=args.time_based_progress,
                               progress=parse.num(args.progress), autosave=args.save, max_progress_interval=args.mpi,
                               max_iter=args.max_iter, min_iterations=args.min_iter, max_time=args.max_time,
                               early_stopping=T.get_early_stopper(args), load_last_model=args.load_last_model,
                               validation_metrics=args.metrics.split(","))
    assert fake.n_train == g["train_calls"] and fake.costs[:50] == g["costs"]
    assert len(log) == len(g["log"])
    for mine, ref in zip(log, g["log"]):
        if mine[0] == "metrics":
            assert ref[0] == "metrics" and set(mine[1]) == set(ref[1])
            for k in ref[1]:
                assert mine[1][k] == pytest.approx(ref[1][k], rel=1e-12, abs=1e-15), k
        else:
            assert mine == ref
    assert sorted(os.listdir(work)) == g["left"]
    if "error" in g["ret"]:                # the reference dies on filename[best_run] when nothing was saved; here: no file
        assert best is None and set(metrics) == set(p.metrics)
    else:
        assert os.path.basename(best) == g["ret"]["best"]
        assert {k: pytest.approx(v, rel=1e-12, abs=1e-15) for k, v in g["ret"]["metrics"].items()} == metrics


def test_pareto_front(gold):
    _, p, _ = build(["-d", "/tmp/x/"])
    for names, want in gold["pareto"]["fronts"].items():
        assert [int(i) for i in p.get_pareto_front(gold["pareto"]["curves"], names.split(","))] == want


class FakeScoreEngine(object):
    """predict_function = the shared stand-in scores; test_function = what the device does with them: items of the input
    window to -inf when asked, ordered top-k"""

    def __init__(self, fake):
        self.fake = fake

    def predict_function(self, X, mask):
        return self.fake.predict_function(X, mask)

    def test_function(self, inputs, k=10, exclude_seen=True):
        X, mask = np.asarray(inputs[0]), np.asarray(inputs[1])
        scores = self.fake.predict_function(X, mask).astype(np.float64)
        if exclude_seen:
            for b in range(len(X)):
                scores[b, X[b, :int(mask[b].sum()), 0]] = -np.inf
        return np.argsort(-scores, axis=1, kind="stable")[:, :k]


@pytest.mark.parametrize("i", range(len(TEST_CASES)))
def test_test_cli_finds_scores_and_records_like_the_reference(gold, root, tmp_path, monkeypatch, i):
    from sbr_amd import test as Te
    argv, present = TEST_CASES[i]
    g = gold["tests"][i]
    assert g["argv"] == argv and g["present"] == present
    droot = str(tmp_path / "ds") + "/"
    shutil.copytree(root, droot)
    os.makedirs(droot + "models")
    real_get_predictor = parse.get_predictor
    state = {}

    def prepared_predictor(args):
        p = real_get_predictor(args)
        fake = FakeScores(0)

        def prepare_model(dataset):
            p.n_items = fake.n_items = dataset.n_items
            p.set_dataset(dataset)
            p.engine = FakeScoreEngine(fake)
        p.prepare_model = prepare_model
        p.load = fake.load
        state["fake"] = fake
        return p
    args = parse.command_parser(parse.predictor_command_parser, Te.test_command_parser, argv=["-d", droot] + argv)
    p0 = real_get_predictor(args)
    for ne in present:
        open(droot + "models/" + p0._get_model_filename(ne), "w").close()
    monkeypatch.setattr(parse, "get_predictor", prepared_predictor)
    Te.main(["-d", droot] + argv)
    assert state["fake"].loaded == g["loaded"]                 # same checkpoints, in the same (epoch) order
    files = {n: open(droot + "results/" + n).read() for n in sorted(os.listdir(droot + "results")) if n != "README"}
    assert sorted(files) == sorted(g["results"])
    for name, ref in g["results"].items():
        mine = files[name]
        if not name.endswith("_full_rank"):
            # the one deliberate difference: a tab between the epoch count and the first metric (the reference glues them,
            # test.py:92, and cannot read its own file back, :112-118)
            mine = "".join(line.replace("\t", "", 1) + "\n" for line in mine.splitlines())
            assert [l.count("\t") for l in mine.splitlines()] == [l.count("\t") for l in ref.splitlines()]
            for lm, lr in zip(mine.splitlines(), ref.splitlines()):
                assert lm == lr, name
        else:
            assert mine == ref
