"""Known answers for the host side of `train.py -m RNN`, produced by driving the REFERENCE's own code
(/root/reference; runs only in this container, the fixture is committed):

  helpers/command_parser.py   command_parser + get_predictor  -> parsed options and the predictor they build
  neural_networks/rnn_base.py _get_model_filename             -> checkpoint names (what --load_last_model / test.py glob for)
                              _gen_mini_batch + _prepare_input (rnn_one_hot.py / rnn_sampling.py)
                                                              -> the training and validation mini-batches, seeded
                              train()                         -> the loop's control flow: when it validates, what it saves
                                                                 and deletes, when early stopping ends it, what it returns
                              get_pareto_front, load_last, top_k_recommendations
  test.py                     main(): which checkpoint files it finds and in what order it scores them, the viewed / goal
                              split, the exclusion of viewed items, the results/ files it appends

What is NOT the reference here: Theano, Lasagne and gensim are absent, so every `theano.*` / `lasagne.*` / `gensim.*`
import resolves to an inert stand-in (the network-building methods are never called: `n_items` /
`effective_sampling` are assigned as rnn_one_hot.py:40 / rnn_sampling.py:102-106 do, and the compiled
`train_function` / `test_function` are replaced by the deterministic fakes below, the same ones the test hands to this
package's train()).  Python 2 semantics the source relies on are restored for the run: `xrange`, list-returning `map`,
`dict.keys()[0]` (model.metrics), generators ended by an escaping StopIteration, `np.cast`, `cPickle`, `file`, `time.clock`.  None of it touches the logic being recorded.

Output: tests/golden/cli_reference.json        python tools/make_cli_golden.py
"""
import builtins
import importlib.abc
import importlib.machinery
import json
import os
import pickle
import random
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

# ------------------------------------------------------------------ cases (shared with tests/test_cli_reference_golden.py)
from cli_cases import (CASES, BATCH_CASES, LOOP_CASES, TEST_CASES, FakeFunctions, FakeScores,      # noqa: E402
                       batch_to_json)


def install_python2_and_stubs():
    builtins.xrange = range
    _map = map
    builtins.map = lambda *a: list(_map(*a))
    builtins.file = open
    import time
    time.clock = time.perf_counter

    class _Cast(dict):
        def __missing__(self, k):
            return lambda x: np.asarray(x, dtype=k)[()]
    if not hasattr(np, "cast"):
        np.cast = _Cast()
    sys.modules["cPickle"] = pickle

    class Meta(type):
        def __getattr__(cls, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return Inert

    class Inert(metaclass=Meta):
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return Inert

        def __call__(self, *a, **k):
            return Inert()

    class Stub(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return sys.modules.get(self.__name__ + "." + name, Inert)

    class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in ("theano", "lasagne", "gensim"):
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            m = Stub(spec.name)
            m.__path__ = []
            if spec.name == "theano":
                m.config = types.SimpleNamespace(floatX="float32")
            return m

        def exec_module(self, module):
            pass
    sys.meta_path.insert(0, Finder())
    sys.path[:0] = ["/root/reference"] + ["/root/reference/" + d for d in
                                          ("neural_networks", "helpers", "factorization", "lazy", "word2vec")]


def python2_generator(make):
    """Python 2: a StopIteration escaping a generator body just ends it (PEP 479 turned that into a RuntimeError)."""
    def wrapped(*a, **k):
        g = make(*a, **k)
        while True:
            try:
                item = next(g)
            except RuntimeError as e:
                if "StopIteration" not in str(e):
                    raise
                return
            yield item
    return wrapped


class ListKeys(dict):                      # Python 2: dict.keys() is a list
    def keys(self):
        return list(dict.keys(self))


def main():
    install_python2_and_stubs()
    import helpers.command_parser as cp
    import train as reftrain
    from helpers.data_handling import DataHandler
    import make_host_golden as mh

    def build(argv, root=None):
        sys.argv = ["train.py"] + list(argv)
        args = cp.command_parser(cp.predictor_command_parser, reftrain.training_command_parser, cp.early_stopping_command_parser)
        p = cp.get_predictor(args)
        dataset = None
        if root is not None:
            dataset = DataHandler(dirname=root, extended_training_set=args.extended_set, shuffle_training=args.tshuffle)   # train.py:43
            p.n_items = dataset.n_items                                   # rnn_one_hot.py:40, rnn_sampling.py:102
            if hasattr(p, "sampling"):
                p.effective_sampling = int(p.sampling * p.n_items) if p.sampling < 1 else int(p.sampling)   # :103-106
            p.set_dataset(dataset)
        return args, p, dataset

    out = {"cases": [], "batches": [], "loops": []}
    for argv in CASES:
        args, p, _ = build(argv)
        a = {k: (v if not isinstance(v, float) or np.isfinite(v) else repr(v)) for k, v in vars(args).items()}
        out["cases"].append(dict(argv=argv, args=a, cls=type(p).__name__, name=p.name,
                                 file_1p5=p._get_model_filename(1.5), file_glob=p._get_model_filename("*"),
                                 file_round=p._get_model_filename(round(12.34567, 3))))

    root = mh.dataset_dir()
    for argv, seed, n_train in BATCH_CASES:
        args, p, dataset = build(argv, root)
        random.seed(seed); np.random.seed(seed)
        gen = p._gen_mini_batch(p.sequence_noise(dataset.training_set()))
        train_batches = [batch_to_json(next(gen)) for _ in range(n_train)]
        random.seed(seed + 1); np.random.seed(seed + 1)
        test_batches = [[batch_to_json(b), [int(g) for g in goal]]
                        for b, goal in mh.drain(p._gen_mini_batch(dataset.validation_set(epochs=1), test=True))]
        out["batches"].append(dict(argv=argv, seed=seed, train=train_batches, test=test_batches,
                                   epochs=float(dataset.training_set.epochs)))

    for case in LOOP_CASES:
        argv, seed = case["argv"], case["seed"]
        work = tempfile.mkdtemp() + "/"
        args, p, dataset = build(argv, root)
        for ne in case.get("pre", []):
            open(work + p._get_model_filename(ne), "w").close()
        p.metrics = ListKeys(p.metrics)
        p._gen_mini_batch = python2_generator(p._gen_mini_batch)
        fake = FakeFunctions(dataset.n_items)
        p.train_function = fake.train_function
        p.test_function = lambda batch_input, k=10, fake=fake: fake.rank_rows(batch_input[0], batch_input[1], k)[0]
        log = []
        p.save = lambda fn, log=log: (log.append(["save", os.path.basename(fn)]), open(fn, "w").close())
        p.load = lambda fn, log=log: log.append(["load", os.path.basename(fn)])
        validate = p._compute_validation_metrics

        def logged_validation(metrics, log=log, fake=fake, validate=validate):
            log.append(["validate", fake.n_train])
            metrics = validate(metrics)
            log.append(["metrics", {k: float(v[-1]) for k, v in metrics.items()}])
            return metrics
        p._compute_validation_metrics = logged_validation
        _remove = os.remove
        os.remove = lambda fn, log=log: (log.append(["remove", os.path.basename(fn)]), _remove(fn))
        random.seed(seed); np.random.seed(seed)
        try:
            try:                                                           # the call of train.py:46-57
                metrics, _, best = p.train(dataset, save_dir=work, time_based_progress=args.time_based_progress,
                                           progress=reftrain.num(args.progress), autosave=args.save,
                                           max_progress_interval=args.mpi, max_iter=args.max_iter, min_iterations=args.min_iter,
                                           max_time=args.max_time, early_stopping=cp.get_early_stopper(args),
                                           load_last_model=args.load_last_model, validation_metrics=args.metrics.split(","))
                ret = dict(metrics={k: float(v) for k, v in metrics.items()}, best=os.path.basename(best))
            except KeyError:                # rnn_base.py:356 filename[best_run]: the best run was not saved (--save None)
                ret = dict(error="KeyError")
        finally:
            os.remove = _remove
        out["loops"].append(dict(argv=argv, seed=seed, pre=case.get("pre"), log=log, ret=ret, train_calls=fake.n_train,
                                 left=sorted(os.listdir(work)), costs=fake.costs[:50]))

    # --- test.py main() with a stand-in for the compiled predict function
    import test as reftest
    out["tests"] = []
    real_get_predictor = cp.get_predictor
    for argv, present in TEST_CASES:
        droot = mh.dataset_dir()
        os.makedirs(droot + "models")
        state = {}

        def prepared_predictor(args, state=state):
            p = real_get_predictor(args)
            fake = FakeScores(0)

            def prepare_model(dataset, p=p, fake=fake):                   # rnn_one_hot.py:40 / rnn_sampling.py:102-106
                p.n_items = fake.n_items = dataset.n_items
                p.dataset = dataset
            p.prepare_model = prepare_model
            p.load = fake.load
            p.predict_function = fake.predict_function
            state["fake"] = fake
            return p
        cp.get_predictor = prepared_predictor
        sys.argv = ["test.py", "-d", droot] + list(argv)
        args = cp.command_parser(cp.predictor_command_parser, reftest.test_command_parser)
        p0 = real_get_predictor(args)
        for ne in present:
            open(droot + "models/" + p0._get_model_filename(ne), "w").close()
        try:
            reftest.main()
        finally:
            cp.get_predictor = real_get_predictor
        files = {}
        for n in sorted(os.listdir(droot + "results")) if os.path.isdir(droot + "results") else []:
            files[n] = open(droot + "results/" + n).read()
        out["tests"].append(dict(argv=argv, present=present, loaded=state["fake"].loaded, results=files))

    # get_pareto_front on hand-made curves (rnn_base.py:436-468)
    args, p, _ = build(["-d", "/tmp/x/"])
    curves = dict(sps=[0.1, 0.3, 0.2, 0.3, 0.25, 0.4], recall=[0.5, 0.2, 0.6, 0.1, 0.7, 0.1],
                  blockbuster_share=[0.9, 0.8, 0.85, 0.7, 0.95, 0.99])
    out["pareto"] = dict(curves=curves, fronts={",".join(names): [int(i) for i in p.get_pareto_front(curves, list(names))]
                                                for names in (("sps",), ("sps", "recall"), ("sps", "blockbuster_share"),
                                                              ("recall", "blockbuster_share", "sps"))})
    path = os.path.join(ROOT, "tests", "golden", "cli_reference.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
