"""The host logic either side of the hot path against known answers produced by the REFERENCE's own classes
(tests/golden/host_reference.json, written by tools/make_host_golden.py from /root/reference): evaluation metrics,
the training-sequence stream under every sub-sequence option, target selection, sequence noise, early stopping.
The random draws come from `random` / `np.random` in the same order as the reference's, so same seed => same stream."""
import json
import os
import random
import shutil

import numpy as np
import pytest

from sbr_amd import options, train
from sbr_amd.data import DataHandler, Evaluator

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "host_reference.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    d = tmp_path_factory.mktemp("ds")
    os.makedirs(d / "data")
    src = os.path.join(HERE, "golden", "preprocess")
    for n in os.listdir(src):
        if n not in ("ARGS", "ratings.dat"):
            shutil.copy(os.path.join(src, n), d / "data" / n)
    return DataHandler(dirname=str(d) + "/")


def test_dataset_stats_and_item_popularity(gold, dataset):
    assert dataset.n_users == gold["stats"]["n_users"] and dataset.n_items == gold["stats"]["n_items"]
    assert dataset.training_set.n_users == gold["stats"]["train_users"]
    assert dataset.training_set.n_interactions == gold["stats"]["train_interactions"]
    assert np.asarray(dataset.item_popularity).tolist() == gold["item_popularity"]


def test_evaluator_metrics(gold, dataset):
    g = gold["evaluator"]
    e = Evaluator(dataset, k=g["k"])
    for goal, pred in g["instances"]:
        e.add_instance(goal, pred)
    for name, want in g["metrics"].items():
        got = float(getattr(e, name)())
        assert got == pytest.approx(want, rel=1e-12, abs=1e-15), name


@pytest.mark.parametrize("key,kw", [("default", {}), ("max5_contiguous", dict(max_length=5)),
                                    ("max5_begining", dict(max_length=5, subsequence="begining")),
                                    ("max6_random_sub", dict(max_length=6, subsequence="random")),
                                    ("random_length", dict(max_length=8, length_choice="random"))])
def test_training_sequence_stream(gold, dataset, key, kw):
    random.seed(17); np.random.seed(17)
    got = [[[[int(i), float(r)] for i, r in seq], str(u)] for seq, u in dataset.training_set(epochs=1, **kw)]
    assert got == gold["sequence_streams"][key]


@pytest.mark.parametrize("key,kw", [("next1", dict(n_targets=1)), ("next3", dict(n_targets=3)),
                                    ("shuffle2", dict(n_targets=2, shuffle=True)), ("bias", dict(n_targets=2, bias=0.5))])
def test_select_targets(gold, dataset, key, kw):
    seqs = [s for s, _ in gold["sequence_streams"]["default"][:12]]
    g = gold["select_targets"][key]
    random.seed(23); np.random.seed(23)
    t = options.SelectTargets(**kw)
    t.set_dataset(dataset)
    assert t.name == g["name"]
    assert [t([list(x) for x in s[2:]]) for s in seqs] == g["train"]
    assert [t([list(x) for x in s[2:]], test=True) for s in seqs] == g["test"]


@pytest.mark.parametrize("key,kw", [("none", {}), ("dropout", dict(dropout=0.3)), ("swap", dict(swap=0.3)),
                                    ("ratings", dict(ratings_perturb=0.4)), ("shuf", dict(shuf=0.5, shuf_std=2.0))])
def test_sequence_noise(gold, key, kw):
    seqs = [s for s, _ in gold["sequence_streams"]["default"][:12]]
    g = gold["sequence_noise"][key]
    random.seed(29); np.random.seed(29)
    nz = options.SequenceNoise(**kw)
    assert nz.name == g["name"]
    gen = (([list(x) for x in s], "u%d" % k) for k, s in enumerate(seqs))
    assert [[[[int(i), float(r)] for i, r in s], u] for s, u in nz(gen)] == g["out"]


def test_early_stopping_rules(gold):
    g = gold["early_stopping"]
    epochs = lambda n: [0.5 * (k + 1) for k in range(n)]
    stoppers = dict(after3=lambda: train.StopAfterN(n=3), after2_lib=lambda: train.StopAfterN(n=2, higher_is_better=False),
                    worst2=lambda: train.WaitWorstCaseTimesX(x=2., min_wait=1.),
                    worst15_lib=lambda: train.WaitWorstCaseTimesX(x=1.5, min_wait=0.5, higher_is_better=False))
    assert set(stoppers) == set(g["decisions"])
    for k, mk in stoppers.items():
        got = [[bool(mk()(epochs(n + 1), c[:n + 1])) for n in range(len(c))] for c in g["curves"]]
        assert got == g["decisions"][k], k
    assert any(any(d) for d in g["decisions"]["after3"]) and any(any(d) for d in g["decisions"]["worst2"])


def test_early_stopping_options_reach_the_stoppers():
    args = options.command_parser(train.early_stopping_command_parser, argv=["--es_m", "WorstTimesX", "--es_x", "3", "--es_LiB"])
    s = train.get_early_stopper(args)
    assert isinstance(s, train.WaitWorstCaseTimesX) and s.x == 3.0 and s.min_wait == 1.0 and not s.higher_is_better
    args = options.command_parser(train.early_stopping_command_parser, argv=["--es_m", "StopAfterN"])
    s = train.get_early_stopper(args)
    assert isinstance(s, train.StopAfterN) and s.n == 5 and s.higher_is_better
    assert train.get_early_stopper(options.command_parser(train.early_stopping_command_parser, argv=[])) is None
