"""Known-answer vectors for the host logic around the hot path, produced by the REFERENCE's own code
(/root/reference, imported here; only runs in this container, the fixture is committed):

  helpers/evaluation.py  Evaluator            -> every metric test.py / train.py can ask for, on seeded instances
  helpers/data_handling.py DataHandler / SequenceGenerator -> the (sequence, user) stream of the training set under
                                                  each sub-sequence option, and the item popularity table
  neural_networks/target_selection.py SelectTargets, neural_networks/sequence_noise.py SequenceNoise,
  helpers/early_stopping.py StopAfterN

evaluation.py and data_handling.py import theano at module level without using it in these classes: an EMPTY stand-in
module named `theano` is put in sys.modules for the import, and `xrange` is aliased to `range` (Python 2 source run
under Python 3); nothing of the arithmetic is touched by either.
The dataset directory is tests/golden/preprocess (written by the reference's preprocess.py, see
tools/make_preprocess_golden.py).  Output: tests/golden/host_reference.json
    python tools/make_host_golden.py
"""
import importlib.util
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def load(name, path):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, path))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def dataset_dir():
    """tests/golden/preprocess/* laid out as <dir>/data/* (what DataHandler expects)."""
    d = tempfile.mkdtemp()
    os.makedirs(os.path.join(d, "data"))
    src = os.path.join(ROOT, "tests", "golden", "preprocess")
    for n in os.listdir(src):
        if n not in ("ARGS", "ratings.dat"):
            shutil.copy(os.path.join(src, n), os.path.join(d, "data", n))
    return d + "/"


def drain(gen):
    """a Python 2 generator that ends by letting StopIteration escape is a RuntimeError under PEP 479"""
    out = []
    try:
        for x in gen:
            out.append(x)
    except RuntimeError as e:
        assert "StopIteration" in str(e)
    return out


def instances(seed, n_items, n=40):
    rng = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        goal = rng.choice(n_items, size=rng.randint(1, 8), replace=False).tolist()
        pred = rng.choice(n_items, size=rng.choice([0, 3, 10, 10, 10]), replace=False).tolist()
        if rng.rand() < 0.5 and pred:
            pred[rng.randint(len(pred))] = goal[0]
            pred = list(dict.fromkeys(pred))
        out.append((goal, pred))
    return out


def main():
    import builtins
    builtins.xrange = range                                        # the reference is Python 2
    stub = types.ModuleType("theano"); stub.tensor = types.ModuleType("theano.tensor")
    sys.modules["theano"] = stub; sys.modules["theano.tensor"] = stub.tensor
    ev = load("ref_evaluation", "helpers/evaluation.py")
    dh = load("ref_data_handling", "helpers/data_handling.py")
    ts = load("ref_target_selection", "neural_networks/target_selection.py")
    sn = load("ref_sequence_noise", "neural_networks/sequence_noise.py")
    es = load("ref_early_stopping", "helpers/early_stopping.py")
    out = {}

    root = dataset_dir()
    dataset = dh.DataHandler(dirname=root)
    out["stats"] = dict(n_users=dataset.n_users, n_items=dataset.n_items, train_users=dataset.training_set.n_users,
                        train_interactions=dataset.training_set.n_interactions)
    out["item_popularity"] = np.asarray(dataset.item_popularity).tolist()

    # --- Evaluator
    inst = instances(5, dataset.n_items)
    e = ev.Evaluator(dataset, k=10)
    for g, p in inst:
        e.add_instance(g, p)
    metrics = {}
    for name in ("average_precision", "average_recall", "average_ndcg", "sps", "user_coverage", "item_coverage",
                 "blockbuster_share", "average_novelty"):
        metrics[name] = float(getattr(e, name)())
    out["evaluator"] = dict(instances=inst, k=10, metrics=metrics)

    # --- SequenceGenerator under each option (one pass over the training file)
    streams = {}
    for key, kw in (("default", {}), ("max5_contiguous", dict(max_length=5)),
                    ("max5_begining", dict(max_length=5, subsequence="begining")),
                    ("max6_random_sub", dict(max_length=6, subsequence="random")),
                    ("random_length", dict(max_length=8, length_choice="random"))):
        random.seed(17); np.random.seed(17)
        streams[key] = [[[[int(i), float(r)] for i, r in seq], str(u)]
                        for seq, u in dataset.training_set(epochs=1, **kw)]
    out["sequence_streams"] = streams

    # --- SelectTargets
    seqs = [s for s, _ in streams["default"][:12]]
    sel = {}
    for key, kw in (("next1", dict(n_targets=1)), ("next3", dict(n_targets=3)), ("shuffle2", dict(n_targets=2, shuffle=True)),
                    ("bias", dict(n_targets=2, bias=0.5))):
        random.seed(23); np.random.seed(23)
        t = ts.SelectTargets(**kw)
        t.set_dataset(dataset)
        sel[key] = dict(name=t.name, train=[t(s[2:]) for s in seqs], test=[t(s[2:], test=True) for s in seqs])
    out["select_targets"] = sel

    # --- SequenceNoise
    noise = {}
    for key, kw in (("none", {}), ("dropout", dict(dropout=0.3)), ("swap", dict(swap=0.3)),
                    ("ratings", dict(ratings_perturb=0.4)), ("shuf", dict(shuf=0.5, shuf_std=2.0))):
        random.seed(29); np.random.seed(29)
        nz = sn.SequenceNoise(**kw)
        gen = ((([list(x) for x in s]), "u%d" % k) for k, s in enumerate(seqs))
        noise[key] = dict(name=nz.name, out=[[[[int(i), float(r)] for i, r in s], u] for s, u in drain(nz(gen))])
    out["sequence_noise"] = noise

    # --- early stopping: decision after every validation pass of a few curves
    curves = [[0.1, 0.2, 0.15, 0.19, 0.18, 0.17, 0.3], [0.3, 0.2, 0.2, 0.1, 0.1, 0.05, 0.01, 0.0],
              [0.1, 0.1, 0.1, 0.1, 0.1, 0.1], [0.1, 0.2, 0.3, 0.25, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2, 0.2]]
    epochs = lambda n: [0.5 * (k + 1) for k in range(n)]
    stoppers = dict(after3=lambda: es.StopAfterN(n=3), after2_lib=lambda: es.StopAfterN(n=2, higher_is_better=False),
                    worst2=lambda: es.WaitWorstCaseTimesX(x=2., min_wait=1.),
                    worst15_lib=lambda: es.WaitWorstCaseTimesX(x=1.5, min_wait=0.5, higher_is_better=False))
    out["early_stopping"] = dict(curves=curves, decisions={
        k: [[bool(mk()(epochs(n + 1), c[:n + 1])) for n in range(len(c))] for c in curves] for k, mk in stoppers.items()})

    path = os.path.join(ROOT, "tests", "golden", "host_reference.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
