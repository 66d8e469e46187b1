"""Host-side mirror of the reference's RNN model classes, Python 3, with the Theano/Lasagne
graph replaced by the HIP engine (engine.RNNEngine).  Same class names, constructor arguments,
methods, filenames and checkpoint layout as neural_networks/rnn_base.py (RNNBase),
rnn_one_hot.py (RNNOneHot) and rnn_sampling.py (RNNSampling), so `train.py -m RNN` code that
holds one of these objects keeps working:

    prepare_model(dataset) / train(dataset, ...) / top_k_recommendations(sequence, ...) /
    save(filename) / load(filename) / load_last(save_dir)

Only what the RNN hot path needs is here, `--r_bi` and `--r_emb` included; `--mf/--uf`
(feature tables the reference never loads, rnn_base.py:27-29,572,607) raise NotImplementedError in prepare_model.
"""
import glob
import math
import os
import pickle
import random
import re
import sys
from bisect import bisect
from time import time

import numpy as np

from .options import Adagrad, RecurrentLayers, SelectTargets, SequenceNoise

MAX_LENGTH = 200      # rnn_base.py:24
BATCH_SIZE = 10       # rnn_base.py:32


class RNNBase(object):
    """rnn_base.py:58-642 minus everything Theano: the engine owns parameters and math."""

    def __init__(self, sequence_noise=None, recurrent_layer=None, updater=None, target_selection=None,
                 interactions_are_unique=True, other_features=None, use_ratings_features=True, movies_features=None,
                 use_movies_features=True, users_features=None, use_users_features=True, max_length=MAX_LENGTH,
                 batch_size=BATCH_SIZE):
        self.other_features, self.movies_features, self.users_features = other_features, movies_features, users_features
        self.use_ratings_features = use_ratings_features
        self.use_movies_features = use_movies_features
        self.use_users_features = use_users_features
        self.max_length, self.batch_size = max_length, batch_size
        self.sequence_noise = sequence_noise if sequence_noise is not None else SequenceNoise()
        self.recurrent_layer = recurrent_layer if recurrent_layer is not None else RecurrentLayers()
        self.updater = updater if updater is not None else Adagrad()
        self.target_selection = target_selection if target_selection is not None else SelectTargets()
        self.interactions_are_unique = interactions_are_unique
        self._input_type = "int32"
        self.name = "RNN base"
        self.metrics = {"recall": {"direction": 1}, "sps": {"direction": 1}, "user_coverage": {"direction": 1},
                        "item_coverage": {"direction": 1}, "ndcg": {"direction": 1},
                        "blockbuster_share": {"direction": -1}}
        self.engine = None
        self.dp = None          # parallel.DataParallel around self.engine in a multi-rank driver (attach_data_parallel)

    def attach_data_parallel(self, dp):
        """A multi-rank training driver hands over its parallel.DataParallel: from then on everything that reads parameters as a
        whole or ranks with them -- save, test_function / predict_function, the batched validation -- goes through its
        collectives, which every rank enters at the same step.  With lazily stepped row-sparse blocks the engine refuses those
        calls when they come from one rank alone (engine.dp_guard: a rank-local flush would fork the replicas by float32
        roundings); through `dp` they are legal, and every rank must then call save / the tests together (rank 0 may be the only
        one that keeps the file: see save(write=...))."""
        self.dp = dp

    def _rd(self):
        """who answers get_all_param_values / test_function / predict_function: the data-parallel wrapper when there is one"""
        return self.dp if self.dp is not None else self.engine

    # ------------------------------------------------------------------ model construction
    def _n_optional_features(self):
        return 10 if self.use_ratings_features else 0       # rating one-hot on a scale of ten (rnn_base.py:590-605)

    def _input_size(self):
        return 2 if self.use_ratings_features else 1        # indices per step (rnn_base.py:615-622)

    def _engine_kwargs(self):
        raise NotImplementedError

    def prepare_model(self, dataset):
        """Must be called before train, load or top_k_recommendations (rnn_base.py:106-109)."""
        from .engine import RNNEngine
        if self.use_movies_features or self.use_users_features:
            # the reference dereferences feature tables that are always None (rnn_base.py:27-29,572,607)
            raise NotImplementedError("movie/user features (--mf/--uf) are unusable in the reference and unsupported here")
        self.n_items = dataset.n_items
        kw = dict(cell=self.recurrent_layer.layer_type, layers=self.recurrent_layer.layers, n_items=self.n_items,
                  max_length=self.max_length, batch_size=self.batch_size, grad_clip=float(self.recurrent_layer.grad_clip),
                  input_size=self.n_items + self._n_optional_features(), n_feat=self._input_size(),
                  embedding_size=self.recurrent_layer.embedding_size, bidirectional=self.recurrent_layer.bidirectional)
        kw.update(self.updater.engine_kwargs())
        kw.update(self._engine_kwargs())
        self.engine = RNNEngine(**kw)
        self._init_parameters()

    def _init_parameters(self, seed=None):
        """The initialisers the reference's layers name [3P] (engine.initial_values: gate weights and peepholes
        Normal(std 0.1), biases and initial states 0, stock RecurrentLayer weights U(-0.01, 0.01), embedding Normal(0.01),
        output W GlorotUniform(gain), output b 0), by parameter name."""
        from .engine import initial_values
        rng = np.random.RandomState(seed)
        self.engine.set_all_param_values(initial_values(self.engine.param_descs, rng, getattr(self, "last_layer_init", 1.0)))

    # ------------------------------------------------------------------ filenames (rnn_base.py:111-130)
    def _common_filename(self, epochs):
        filename = ("ml" + str(self.max_length) + "_bs" + str(self.batch_size) + "_ne" + str(epochs) + "_" +
                    self.recurrent_layer.name + "_" + self.updater.name + "_" + self.target_selection.name)
        if self.sequence_noise.name != "":
            filename += "_" + self.sequence_noise.name
        if not self.interactions_are_unique:
            filename += "_ri"
        if not (self.use_ratings_features or self.use_movies_features or self.use_users_features):
            filename += "_nf"
        if self.use_ratings_features:
            filename += "_rf"
        if self.use_movies_features:
            filename += "_mf"
        if self.use_users_features:
            filename += "_uf"
        return filename

    def _get_model_filename(self, epochs):
        raise NotImplementedError

    # ------------------------------------------------------------------ features (rnn_base.py:590-642)
    def _get_features(self, item, user_id=None):
        """[item_id] (+ [n_items + rating bucket] with --rf): rating one-hot index round(r*2)-1."""
        item_id, rating = item
        if self.use_ratings_features:
            # Python 2's round() goes half away from zero (the reference's interpreter); Python 3's goes to even
            return [item_id, self.n_items + (int(math.floor(rating * 2 + 0.5)) - 1) % 10]
        return [item_id]

    def _pack(self, sequences):
        """Common part of _prepare_input: X left-aligned zero-padded (pad id 0 is a real item, the
        mask alone marks validity), mask, first target, popularity**db (rnn_one_hot.py:83-106)."""
        B, T, F = len(sequences), self.max_length, self._input_size()
        X = np.zeros((B, T, F), dtype=np.int32)
        mask = np.zeros((B, T), dtype=np.float32)
        Y = np.zeros((B,), dtype=np.int32)
        pop = np.zeros((B,), dtype=np.float32)
        for i, (user_id, in_seq, target) in enumerate(sequences):
            n = len(in_seq)
            if n:
                X[i, :n, :] = np.array([self._get_features(x, user_id) for x in in_seq], dtype=np.int32)
            mask[i, :n] = 1
            Y[i] = target[0][0]
            pop[i] = self.dataset.item_popularity[target[0][0]] ** self.diversity_bias
        return X, mask, Y, pop

    def set_dataset(self, dataset):
        self.dataset = dataset
        self.target_selection.set_dataset(dataset)

    # ------------------------------------------------------------------ batches (rnn_base.py:373-420)
    def _gen_mini_batch(self, sequence_generator, test=False, max_reuse_sequence=np.inf):
        while True:
            j = 0
            sequences = []
            batch_size = 1 if test else self.batch_size
            while j < batch_size:
                try:
                    sequence, user_id = next(sequence_generator)
                except StopIteration:       # the source ran dry (validation/test: epochs=1): end this generator too,
                    return                  # as the reference's python-2 generator does implicitly
                if not test:
                    k = int(min(batch_size - j, len(sequence) - 2, max_reuse_sequence))
                    seq_lengths = sorted(random.sample(range(2, len(sequence)), k))
                else:
                    seq_lengths = [int(len(sequence) / 2)]
                skipped = 0
                for l in seq_lengths:
                    target = self.target_selection(sequence[l:], test=test)
                    if len(target) == 0:
                        skipped += 1
                        continue
                    start = max(0, l - self.max_length)
                    sequences.append([user_id, sequence[start:l], target])
                j += len(seq_lengths) - skipped
            if test:
                yield self._prepare_input(sequences), [i[0] for i in sequence[seq_lengths[0]:]]
            else:
                yield self._prepare_input(sequences)

    # ------------------------------------------------------------------ the compiled-function seam
    def train_function(self, *batch):
        return self.engine.train_function(*batch)

    def _exclude_mode(self):
        """how the compiled test function treats viewed items (engine.test_function): 0 not at all, 1 never ranked"""
        return 1 if self.interactions_are_unique else 0

    def test_function(self, theano_inputs, k=10):
        """ordered top-k ids of the (single) row (rnn_base.py:205-209)."""
        return self._rd().test_function(theano_inputs, k=k, exclude_seen=self._exclude_mode())[0]

    def predict_function(self, X, mask):
        return self._rd().predict_function(X, mask)

    def top_k_recommendations(self, sequence, user_id=None, k=10, exclude=None):
        """rnn_base.py:132-159: last max_length items -> scores -> viewed/excluded to -inf -> top k."""
        if exclude is None:
            exclude = []
        seq = sequence[-min(self.max_length, len(sequence)):]
        X = np.zeros((1, self.max_length, self._input_size()), dtype=np.int32)
        X[0, :len(seq), :] = np.array([self._get_features(x, user_id) for x in seq], dtype=np.int32)
        mask = np.zeros((1, self.max_length), dtype=np.float32)
        mask[0, :len(seq)] = 1
        output = self.predict_function(X, mask)[0]
        if self.interactions_are_unique:
            output[[i[0] for i in sequence]] = -np.inf
        output[exclude] = -np.inf
        return list(np.argpartition(-output, range(k))[:k])

    # ------------------------------------------------------------------ training loop (rnn_base.py:215-356)
    def get_pareto_front(self, metrics, metrics_names):
        costs = np.zeros((len(metrics[metrics_names[0]]), len(metrics_names)))
        for i, m in enumerate(metrics_names):
            costs[:, i] = np.array(metrics[m]) * self.metrics[m]["direction"]
        is_efficient = np.ones(costs.shape[0], dtype=bool)
        for i, c in enumerate(costs):
            if is_efficient[i]:
                is_efficient[is_efficient] = np.any(costs[is_efficient] >= c, axis=1)
        return np.where(is_efficient)[0].tolist()

    def train(self, dataset, max_time=np.inf, progress=2.0, time_based_progress=False, autosave="All", save_dir="",
              min_iterations=0, max_iter=np.inf, max_progress_interval=np.inf, load_last_model=False,
              early_stopping=None, validation_metrics=("sps",)):
        self.set_dataset(dataset)
        validation_metrics = list(validation_metrics)
        if len(set(validation_metrics) & set(self.metrics.keys())) < len(validation_metrics):
            raise ValueError("Incorrect validation metrics. Metrics must be chosen among: " + ", ".join(self.metrics.keys()))
        iterations, epochs_offset = 0, 0
        if load_last_model:
            epochs_offset = self.load_last(save_dir)
        native = self._native_batch_builder(dataset)
        batch_generator = native if native is not None else self._gen_mini_batch(self.sequence_noise(dataset.training_set()))
        start_time = time()
        next_save = int(progress)
        train_costs, current_train_cost, epochs = [], [], []
        metrics = {name: [] for name in self.metrics.keys()}
        filename = {}
        def take(cost):                 # rnn_base.py:290-293; with device batches the cost arrives one iteration late
            if cost is not None:
                if np.isnan(cost):
                    raise ValueError("Cost is NaN")
                current_train_cost.append(cost)

        try:
            while time() - start_time < max_time and iterations < max_iter:
                try:
                    batch = next(batch_generator)
                    # device batches: the step is enqueued and the PREVIOUS step's cost comes back (no wait for the step
                    # just enqueued); the last one is collected before anything reads the costs or the parameters
                    take(self.engine.train_step_lagged() if native is not None else self.train_function(*batch))
                except StopIteration:
                    break
                iterations += 1
                progress_indicator = int(time() - start_time) if time_based_progress else iterations
                if progress_indicator >= next_save:
                    if native is not None:
                        take(self.engine.flush_lagged())
                    if progress_indicator >= min_iterations:
                        epochs.append(epochs_offset + dataset.training_set.epochs)
                        train_costs.append(np.mean(current_train_cost))
                        current_train_cost = []
                        metrics = self._compute_validation_metrics(metrics)
                        self._print_progress(iterations, epochs[-1], start_time, train_costs, metrics, validation_metrics)
                        run_nb = len(metrics[list(self.metrics.keys())[0]]) - 1
                        if autosave == "All":
                            filename[run_nb] = save_dir + self._get_model_filename(round(epochs[-1], 3))
                            self.save(filename[run_nb])
                        elif autosave == "Best":
                            pareto_runs = self.get_pareto_front(metrics, validation_metrics)
                            if run_nb in pareto_runs:
                                filename[run_nb] = save_dir + self._get_model_filename(round(epochs[-1], 3))
                                self.save(filename[run_nb])
                                for run in [r for r in filename if r not in pareto_runs]:
                                    try:
                                        os.remove(filename[run])
                                    except OSError:
                                        print("Warning : Previous model could not be deleted")
                                    del filename[run]
                        if early_stopping is not None:
                            if all([early_stopping(epochs, metrics[m]) for m in validation_metrics]):
                                break
                    if isinstance(progress, int):
                        next_save += min(progress, max_progress_interval)
                    else:
                        next_save += min(max_progress_interval, next_save * (progress - 1))
        except KeyboardInterrupt:
            print("Training interrupted")
        if native is not None:
            take(self.engine.flush_lagged())
        if not metrics[validation_metrics[0]]:
            return {}, time() - start_time, None
        best_run = int(np.argmax(np.array(metrics[validation_metrics[0]]) * self.metrics[validation_metrics[0]]["direction"]))
        # the reference raises KeyError here when the best run was not saved (--save None): return None instead
        return ({m: metrics[m][best_run] for m in self.metrics.keys()}, time() - start_time, filename.get(best_run))

    def batched_test_predictions(self, sequence_generator, k=10):
        """Batched form of the reference's per-user validation loop (rnn_base.py:358-371): the same
        test instances `_gen_mini_batch(..., test=True)` yields one by one (split in the middle, last max_length items
        in, the rest as goal) are stacked batch_size at a time and ranked by one test_function call -- rows are
        independent in every kernel, so the ordered top-k ids are identical to the one-row calls.
        Yields (goals, ids) per chunk in the generator's order."""
        Xs, masks, goals = [], [], []

        def flush():
            ids = self._rd().test_function((np.concatenate(Xs), np.concatenate(masks)), k=k, exclude_seen=self._exclude_mode())
            # a row with fewer than k rankable items carries -1 in the places it cannot fill (include/sbr_rnn.h): drop them
            out = (list(goals), [ids[i][ids[i] >= 0] for i in range(len(goals))])
            del Xs[:], masks[:], goals[:]
            return out
        for batch_input, goal in self._gen_mini_batch(sequence_generator, test=True):
            Xs.append(batch_input[0]); masks.append(batch_input[1]); goals.append(goal)
            if len(goals) == self.batch_size:
                yield flush()
        if goals:
            yield flush()

    def _native_batch_builder(self, dataset):
        """Device-side batch builder: item index, + rating index with --rf; next-item or shuffled targets, --n_targets of them
        for the multi-target losses; --db, --sampling_bias; the sequence noise; --target_bias (rows planned on the host by
        the library).  None -> the reference-style host generator: SBR_NATIVE_BATCHES=0, or a case the builder refuses."""
        if os.environ.get("SBR_NATIVE_BATCHES", "1") == "0":
            return None
        ts = self.target_selection
        multi = isinstance(self, RNNMargin)                  # only the multi-target losses look past the first target
        if self.sequence_noise.name != "":
            if not hasattr(dataset.training_set, "users"):
                dataset.training_set.load()
            if max([len(x) for x in dataset.training_set.items] + [0]) > 8192:
                return None  # (a sequence longer than the noise kernel's LDS staging)
        if ts.shuffle and multi and self._engine_targets() > 16 and ts.bias < 0.0:
            return None
        if ts.bias >= 0.0 and multi and self._engine_targets() > 1024:
            return None      # (the host row planner of --target_bias holds at most 1024 targets per row: sbr_dataset_set_target_bias)
        from .data import NativeBatchBuilder
        pop = np.asarray(dataset.item_popularity, dtype=np.float64)
        db = float(getattr(self, "diversity_bias", 0.0))
        sb = float(getattr(self, "sampling_bias", 0.0))
        cdf = np.cumsum(np.power(pop, sb)) if (sb > 0.0 and getattr(self, "effective_sampling", 0)) else None
        return NativeBatchBuilder(self.engine, dataset.training_set, self.n_items, self.batch_size,
                                  pop_db=np.power(pop, db).astype(np.float32), sample_cdf=cdf,
                                  ratings=self.use_ratings_features, shuffle_targets=ts.shuffle, noise=self.sequence_noise,
                                  keep_prob=(np.power(np.min(np.maximum(1, pop)) / np.maximum(1, pop), ts.bias).astype(np.float32)
                                             if ts.bias >= 0.0 else None),
                                  n_targets=self._engine_targets() if multi else 1)

    def _compute_validation_metrics(self, metrics):
        from .data import Evaluator
        ev = Evaluator(self.dataset, k=10)
        for goals, ids in self.batched_test_predictions(self.dataset.validation_set(epochs=1), k=10):
            for goal, row in zip(goals, ids):
                ev.add_instance(goal, row)
        metrics["recall"].append(ev.average_recall())
        metrics["sps"].append(ev.sps())
        metrics["ndcg"].append(ev.average_ndcg())
        metrics["user_coverage"].append(ev.user_coverage())
        metrics["item_coverage"].append(ev.item_coverage())
        metrics["blockbuster_share"].append(ev.blockbuster_share())
        return metrics

    def _print_progress(self, iterations, epochs, start_time, train_costs, metrics, validation_metrics):
        print(self.name, iterations, "batchs, ", epochs, " epochs in", time() - start_time, "s")
        print("Last train cost : ", train_costs[-1])
        for m in self.metrics:
            print(m, ": ", metrics[m][-1])
            if m in validation_metrics:
                d = self.metrics[m]["direction"]
                print("Best ", m, ": ", max(np.array(metrics[m]) * d) * d)
        print("-----------------")
        # machine-readable line on stderr (rnn_base.py:434)
        print(iterations, epochs, time() - start_time, train_costs[-1],
              " ".join(map(str, [metrics[m][-1] for m in self.metrics])), file=sys.stderr)

    # ------------------------------------------------------------------ checkpoints (rnn_base.py:470-515)
    def save(self, filename, write=True):
        """pickle of get_all_param_values(l_out): a plain list of float arrays in Lasagne order.
        Protocol 2 so that a Python-2 reference install can load it.
        write=False: take part in the export (a collective under data parallelism: every rank calls save at the same step)
        without keeping the file -- e.g. every rank but 0."""
        param = self._rd().get_all_param_values()
        if not write:
            return
        print("Save model in " + filename)
        d = os.path.dirname(filename)
        if d and not os.path.exists(d):
            os.makedirs(d)
        with open(filename, "wb") as f:
            pickle.dump(param, f, protocol=2)

    def load_last(self, save_dir):
        def extract_number_of_epochs(filename):
            m = re.search(r"_ne([0-9]+(\.[0-9]+)?)_", filename)
            return float(m.group(1))
        files = glob.glob(save_dir + self._get_model_filename("*"))
        if len(files) == 0:
            print("No previous model, starting from scratch")
            return 0
        last_batch = np.amax(np.array([extract_number_of_epochs(f) for f in files]))
        last_model = save_dir + self._get_model_filename(last_batch)
        print("Starting from model " + last_model)
        self.load(last_model)
        return last_batch

    def load(self, filename):
        """Reads reference checkpoints too (py2 cPickle protocol 2 -> encoding='latin1')."""
        with open(filename, "rb") as f:
            param = pickle.load(f, encoding="latin1")
        self.engine.set_all_param_values([np.asarray(i, dtype=np.float32) for i in param])


class RNNOneHot(RNNBase):
    """RNN + full softmax + categorical cross-entropy, `--loss CCE` (rnn_one_hot.py:13-106)."""

    def __init__(self, diversity_bias=0.0, regularization=0.0, **kwargs):
        super(RNNOneHot, self).__init__(**kwargs)
        self.diversity_bias = np.float64(diversity_bias)     # np.cast[floatX]: the filename prints it as a float
        self.regularization = regularization
        self.name = "RNN with categorical cross entropy"

    def _engine_kwargs(self):
        return dict(loss="CCE", regularization=float(self.regularization))

    def _get_model_filename(self, epochs):
        return "rnn_cce_db" + str(self.diversity_bias) + "_r" + str(self.regularization) + "_" + self._common_filename(epochs)

    def _prepare_input(self, sequences):
        """(X, mask, Y, pop, exclude) as rnn_one_hot.py:83-106 returns them, except that `exclude`
        (B,N) float -- built every batch and never used by train_function in the reference -- is
        None: the engine derives the exclusion from X on the device."""
        X, mask, Y, pop = self._pack(sequences)
        return (X, mask, Y, pop, None)


class RNNSampling(RNNBase):
    """RNN + sampled output losses Blackout / BPR / TOP1 (rnn_sampling.py:14-194)."""

    def __init__(self, loss_function="Blackout", sampling=32, last_layer_tanh=False, last_layer_init=1.0,
                 diversity_bias=0.0, sampling_bias=0.0, **kwargs):
        super(RNNSampling, self).__init__(**kwargs)
        if last_layer_tanh:
            raise NotImplementedError("last_layer_tanh is not reachable from the reference CLI (command_parser.py:120-121)")
        self.last_layer_init, self.last_layer_tanh = last_layer_init, last_layer_tanh
        self.diversity_bias, self.sampling, self.sampling_bias = diversity_bias, sampling, sampling_bias
        if loss_function is None:
            loss_function = "Blackout"
        if loss_function not in ("BPR", "TOP1", "Blackout"):
            raise ValueError("Unknown loss function")
        self.loss_function_name = loss_function
        self.name = "RNN with sampling loss"

    def prepare_model(self, dataset):
        n_items = dataset.n_items                            # rnn_sampling.py:102-106
        self.effective_sampling = int(self.sampling * n_items) if self.sampling < 1 else int(self.sampling)
        super(RNNSampling, self).prepare_model(dataset)

    def _engine_kwargs(self):
        return dict(loss=self.loss_function_name, n_samples=self.effective_sampling)

    def _get_model_filename(self, epochs):
        filename = "rnn_sampling_" + self.loss_function_name + "_"
        if self.sampling_bias > 0.0:
            filename += "p" + str(self.sampling_bias)
        filename += "s" + str(self.sampling) + "_ini" + str(self.last_layer_init) + "_db" + str(self.diversity_bias)
        return filename + "_" + self._common_filename(epochs)

    def _popularity_sample(self):
        if not hasattr(self, "_cumsum"):
            self._cumsum = np.cumsum(np.power(self.dataset.item_popularity, self.sampling_bias))
        return bisect(self._cumsum, random.uniform(0, self._cumsum[-1]))

    def _prepare_input(self, sequences):
        """(X, mask, Y, samples, pop, exclude) as rnn_sampling.py:165-194; the S negatives are shared
        by the batch, drawn with replacement and NOT filtered against the targets."""
        X, mask, Y, pop = self._pack(sequences)
        if self.sampling_bias > 0:
            samples = np.array([self._popularity_sample() for _ in range(self.effective_sampling)], dtype=np.int32)
        else:
            samples = np.random.choice(self.n_items, self.effective_sampling).astype(np.int32)
        return (X, mask, Y, samples, pop, None)


class RNNMargin(RNNBase):
    """RNN + linear output layer + multi-target losses hinge / logit / logsig, `--loss hinge|logit|logsig`
    (rnn_margin.py:13-161).  The reference packs dense (B, N) target and weight matrices on the host every batch; here a
    batch carries the positives of every row (up to --n_targets of them, -1 = none) and the engine's loss kernel derives
    target / weight from them, from the row's own input items and from the default target (csrc/sbr_misc.hip)."""

    def __init__(self, loss_function="hinge", balance=1., popularity_based=False, min_access=0.05, n_targets=1, **kwargs):
        super(RNNMargin, self).__init__(**kwargs)
        self.balance, self.popularity_based, self.min_access, self.n_targets = balance, popularity_based, min_access, n_targets
        if loss_function is None:
            loss_function = "hinge"
        if loss_function not in ("hinge", "logit", "logsig"):
            raise ValueError("Unknown loss function")                   # rnn_margin.py:49
        self.loss_function_name = loss_function
        self.name = "RNN multi-targets"

    MAX_TARGETS = 4096      # columns of the engine's target array (include/sbr_rnn.h)

    def _engine_targets(self):
        return int(min(max(1, self.target_selection.n_targets), self.MAX_TARGETS))

    def _engine_kwargs(self):
        return dict(loss=self.loss_function_name, balance=float(self.balance), n_targets=self._engine_targets(),
                    unique=bool(self.interactions_are_unique))

    def prepare_model(self, dataset):
        super(RNNMargin, self).prepare_model(dataset)
        if self.popularity_based:                                       # rnn_margin.py:149-161
            self.dataset = dataset
            self.engine.set_default_target(self._default_target())

    def _default_target(self):
        if not self.popularity_based:
            return np.zeros(self.n_items)
        view_prob = np.asarray(self.dataset.item_popularity, dtype=np.float64) / self.dataset.training_set.n_users
        return np.minimum(1 - view_prob, (1 - self.min_access) * view_prob / self.min_access)

    def _get_model_filename(self, epochs):
        filename = "rnn_multitarget_" + self.loss_function_name + "_b" + str(self.balance)
        if self.popularity_based:
            filename += "_pb_ma" + str(self.min_access)
        return filename + "_" + self._common_filename(epochs)

    def _prepare_input(self, sequences):
        """(X, mask, targets, None, None) where the reference returns (X, mask, Y, weight, exclude) (rnn_margin.py:112-147):
        targets (B, n_targets) int32, the positives of every row in SelectTargets' order, -1 behind the last one."""
        B, T, F, NT = len(sequences), self.max_length, self._input_size(), self._engine_targets()
        X = np.zeros((B, T, F), dtype=np.int32)
        mask = np.zeros((B, T), dtype=np.float32)
        targets = -np.ones((B, NT), dtype=np.int32)
        for i, (user_id, in_seq, target) in enumerate(sequences):
            n = len(in_seq)
            if n:
                X[i, :n, :] = np.array([self._get_features(x, user_id) for x in in_seq], dtype=np.int32)
            mask[i, :n] = 1
            if len(target) > NT:
                raise ValueError("a row has %d targets, the engine was built for %d (--n_targets)" % (len(target), NT))
            targets[i, :len(target)] = [t[0] for t in target]
        return (X, mask, targets, None, None)

    def _exclude_mode(self):
        """The generic compiled test function (rnn_base.py:196-209) on RAW outputs: viewed items are multiplied by 0, not removed."""
        return 2 if self.interactions_are_unique else 0



class RNNCluster(RNNBase):
    """RNN + sampled output loss + item clustering, `--clusters C` (rnn_cluster.py:19-539; command_parser.py:71-77, :114-115).
    Two models trained side by side from one forward pass: the recurrent network with its sampled head on `cost` (the engine:
    losses Blackout / CCE over the sampled columns / BPR / TOP1 / BPRelu / lin, rnn_cluster.py:151-180), and the cluster head
    -- selection weights and the item / cluster repartition -- on `cost_clusters` (engine.ClusterHead, csrc/sbr_cluster.hip),
    each with its own updater state (:277-285).  Host logic as in the reference: filename grammar, batch packing with the cluster
    samples and the growing scale, the per-user test function and its nine metrics, hard clusters for top_k_recommendations,
    checkpoints with the two cluster arrays appended."""

    ENGINE_LOSS = {"CCE": "SCCE", "Blackout": "Blackout", "BPR": "BPR", "TOP1": "TOP1", "BPRelu": "BPRelu", "lin": "lin"}

    def __init__(self, n_clusters=10, loss="Blackout", cluster_type="mix", sampling=100, cluster_sampling=-1, sampling_bias=0.0,
                 predict_with_clusters=True, cluster_selection_noise=0.0, init_scale=1.0, scale_growing_rate=1.0, max_scale=50, **kwargs):
        super(RNNCluster, self).__init__(**kwargs)
        self.n_clusters = n_clusters
        self.init_scale = np.float64(init_scale)                       # np.cast[floatX]: the filename prints them as floats
        self.effective_scale = np.float64(init_scale)
        self.scale_growing_rate = np.float64(scale_growing_rate)
        self.max_scale = np.float64(max_scale)
        self.cluster_type, self.sampling_bias, self.loss = cluster_type, sampling_bias, loss
        self.cluster_selection_noise = cluster_selection_noise
        self.predict_with_clusters = predict_with_clusters
        if loss not in self.ENGINE_LOSS:
            raise ValueError("Unknown cluster loss")                   # rnn_cluster.py:101
        self.n_samples, self.n_cluster_samples = int(sampling), int(cluster_sampling)
        self.diversity_bias = 0.0
        self.name = "RNN Cluster with categorical cross entropy"
        self.metrics = {"recall": {"direction": 1}, "cluster_recall": {"direction": 1}, "sps": {"direction": 1},
                        "cluster_sps": {"direction": 1}, "ignored_items": {"direction": -1}, "assr": {"direction": 1},
                        "cluster_use": {"direction": 1}, "cluster_use_std": {"direction": -1}, "cluster_size": {"direction": 1}}
        self.head = None

    # ------------------------------------------------------------------ construction
    def _engine_kwargs(self):
        return dict(loss=self.ENGINE_LOSS[self.loss], n_samples=self.n_samples)

    def prepare_model(self, dataset):
        from .engine import ClusterHead
        super(RNNCluster, self).prepare_model(dataset)
        kw = self.updater.engine_kwargs()
        self.head = ClusterHead(self.engine, self.n_clusters, self.cluster_type, loss=self.ENGINE_LOSS[self.loss],
                                max_samples=max(self.n_samples, self.n_cluster_samples, 1), scale=float(self.effective_scale),
                                noise_std=float(self.cluster_selection_noise), seed=np.random.randint(1, 2147462579), **kw)
        H = self.head.n_hidden
        # cluster_selection_layer: DenseLayer(b=None), W GlorotUniform [3P]; cluster_repartition: 0.1 * randn (rnn_cluster.py:182, :239)
        lim = np.sqrt(6.0 / (H + self.n_clusters))
        self.head.set_params(self._create_ini_clusters(), np.random.uniform(-lim, lim, size=(H, self.n_clusters)).astype(np.float32))

    def _create_ini_clusters(self):
        return (0.1 * np.random.randn(self.n_items, self.n_clusters)).astype(np.float32)

    def _get_model_filename(self, epochs):
        filename = "rnn_clusters" + str(self.n_clusters) + "_sc" + str(self.init_scale)
        if self.scale_growing_rate != 1.0:
            filename += "-" + str(self.scale_growing_rate) + "-" + str(self.max_scale)
        filename += "_"
        if self.sampling_bias > 0.0:
            filename += "p" + str(self.sampling_bias)
        filename += "s" + str(self.n_samples)
        if self.n_cluster_samples > 0:
            filename += "_"
            if self.sampling_bias > 0.0:
                filename += "p" + str(self.sampling_bias)
            filename += "cs" + str(self.n_cluster_samples)
        if self.cluster_type == "softmax":
            filename += "_softmax"
        elif self.cluster_type == "mix":
            filename += "_mix"
        if self.cluster_selection_noise > 0.0:
            filename += "_n" + str(self.cluster_selection_noise)
        filename += "_c" + self.loss
        return filename + "_" + self._common_filename(epochs)

    # ------------------------------------------------------------------ batches (rnn_cluster.py:354-407)
    def _popularity_sample(self):
        if not hasattr(self, "_cumsum"):
            self._cumsum = np.cumsum(np.power(self.dataset.item_popularity, self.sampling_bias))
        return bisect(self._cumsum, random.uniform(0, self._cumsum[-1]))

    def _prepare_input(self, sequences):
        """(X, mask, Y, samples, cluster_samples, exclude); `exclude` (B, N) -- unused by the train function, derived from X
        on the device for the test path -- is None.  The scale of the softmax / sigmoid grows with the epochs (:395-400)."""
        X, mask, Y, _ = self._pack(sequences)
        if self.sampling_bias > 0.0:
            samples = np.array([self._popularity_sample() for _ in range(self.n_samples)], dtype=np.int32)
            if self.n_cluster_samples > 0:
                cluster_samples = np.array([self._popularity_sample() for _ in range(self.n_cluster_samples)], dtype=np.int32)
            else:
                cluster_samples = samples
        else:
            samples = np.random.choice(self.n_items, self.n_samples).astype(np.int32)
            if self.n_cluster_samples > 0:
                cluster_samples = np.random.choice(self.n_items, self.n_cluster_samples).astype(np.int32)
            else:
                cluster_samples = samples
        if not hasattr(self, "_last_epoch"):
            self._last_epoch = self.dataset.training_set.epochs
        elif self.dataset.training_set.epochs > self._last_epoch + 1 and self.scale_growing_rate != 1.0:
            self.effective_scale *= self.scale_growing_rate ** int(self.dataset.training_set.epochs - self._last_epoch)
            self._last_epoch += int(self.dataset.training_set.epochs - self._last_epoch)
            print("New scale: ", self.effective_scale)
            self.head.set_scale(float(self.effective_scale))
        return (X, mask, Y, samples, cluster_samples, None)

    def _native_batch_builder(self, dataset):
        return None          # the cluster samples and the growing scale are host state: the reference-style generator feeds this model

    # ------------------------------------------------------------------ the compiled-function seam
    def train_function(self, X, mask, target, samples, cluster_samples, exclude=None):
        """cost = train_function(X, mask, target, samples, cluster_samples, exclude) (rnn_cluster.py:287): both models step, the
        recurrent network's cost comes back"""
        self.engine.set_batch(X, mask, target, samples, np.ones(len(target), dtype=np.float32))
        cost = self.engine.train_step(sync=True)
        self.head.forward_backward(target, cluster_samples, read_cost=False)      # on the user representations of that step's forward
        self.head.apply_update()
        self._drop_hard_clusters()               # the hard clusters / embeddings cached by prepare_tests belong to the old parameters
        return cost

    def _drop_hard_clusters(self):
        for a in ("clusters", "clusters_reverse_index", "clusters_embeddings", "clusters_bias"):
            if hasattr(self, a):
                delattr(self, a)

    def close(self):
        """device arrays of the cluster head (several N x C float arrays) and of the engine"""
        if getattr(self, "head", None) is not None:
            self.head.close()
            self.head = None
        sup = getattr(super(RNNCluster, self), "close", None)
        if sup is not None:
            sup()

    def __del__(self):
        try:
            if getattr(self, "head", None) is not None:
                self.head.close()
        except Exception:
            pass

    def _ranked(self, scores, k):
        return np.argpartition(-scores, range(k), axis=-1)[..., :k]

    def test_function(self, theano_inputs, k=10):
        """(ids without clusters, ids inside the selected cluster, the cluster, items in it) for the single row
        (rnn_cluster.py:327-352): softmax scores, times the hard membership of the row's cluster, viewed items zeroed"""
        X, mask = theano_inputs[0], theano_inputs[1]
        s1 = self.engine.test_probabilities(X, mask)
        rows = s1.shape[0]
        csel = self.head.select(rows)
        used = self.head.hard_clusters()[:, csel].T
        s2 = s1 * used
        if self.interactions_are_unique:
            for b in range(rows):
                seen = X[b, :int(mask[b].sum()), 0]
                s1[b, seen] = 0.0; s2[b, seen] = 0.0
        return self._ranked(s1, k)[0], self._ranked(s2, k)[0], int(csel[0]), float(used[0].sum())

    def _compute_validation_metrics(self, metrics):
        from .data import Evaluator
        clusters = np.zeros(self.n_clusters, dtype="int")
        used_items = []
        ev, ev_clusters = Evaluator(self.dataset, k=10), Evaluator(self.dataset, k=10)
        for batch, goal in self._gen_mini_batch(self.dataset.validation_set(epochs=1), test=True):
            pred1, pred2, cl, n_used = self.test_function(batch)
            ev.add_instance(goal, pred1)
            ev_clusters.add_instance(goal, pred2)
            clusters[cl] += 1
            used_items.append(n_used)
        R = self.head.get_params()[0]
        if self.cluster_type == "softmax":
            ignored_items = 0
            cluster_size = np.histogram(R.argmax(axis=1), bins=range(self.n_clusters + 1))[0].tolist()
        elif self.cluster_type == "mix":
            ignored_items = 0
            sig_clusters = R > 0.0
            sig_clusters[np.arange(self.n_items), R.argmax(axis=1)] = True
            cluster_size = sig_clusters.sum(axis=0)
        else:
            ignored_items = (R.max(axis=1) < 0.0).sum()
            cluster_size = (R > 0.0).sum(axis=0)
        metrics["recall"].append(ev.average_recall())
        metrics["cluster_recall"].append(ev_clusters.average_recall())
        metrics["sps"].append(ev.sps())
        metrics["cluster_sps"].append(ev_clusters.sps())
        metrics["assr"].append(self.n_items / np.mean(used_items))
        metrics["ignored_items"].append(ignored_items)
        metrics["cluster_use"].append(clusters)
        metrics["cluster_use_std"].append(np.std(clusters))
        metrics["cluster_size"].append(cluster_size)
        return metrics

    def _print_progress(self, iterations, epochs, start_time, train_costs, metrics, validation_metrics):
        print(self.name, iterations, "batchs, ", epochs, " epochs in", time() - start_time, "s")
        print("Last train cost : ", train_costs[-1])
        for m in self.metrics.keys():
            print(m, ": ", metrics[m][-1])
            if m in validation_metrics:
                print("Best ", m, ": ", max(np.array(metrics[m]) * self.metrics[m]["direction"]) * self.metrics[m]["direction"])
        print("-----------------")
        print(iterations, epochs, time() - start_time, train_costs[-1], metrics["sps"][-1], metrics["cluster_sps"][-1],
              metrics["recall"][-1], metrics["cluster_recall"][-1], metrics["assr"][-1], metrics["ignored_items"][-1],
              metrics["cluster_use_std"][-1], file=sys.stderr)

    # ------------------------------------------------------------------ recommendations (rnn_cluster.py:440-497)
    def prepare_tests(self):
        """Take the soft clustering and make actual clusters (rnn_cluster.py:440-466)."""
        membership = self.head.get_params()[0]
        params = self.engine.get_all_param_values()
        item_embeddings, item_bias = params[-2], params[-1]
        pos = membership > 0
        best = np.argmax(np.where(np.arange(self.n_clusters)[None, :] == 0, membership, np.where(pos, -np.inf, membership)), axis=1)
        # (:447-458: an item joins every cluster whose membership is positive; one with none joins the cluster of its largest
        # non-positive membership -- the scan starts from cluster 0's value and only a strictly larger one replaces it)
        self.clusters = []
        none = ~pos.any(axis=1)
        for j in range(self.n_clusters):
            self.clusters.append(np.where(pos[:, j] | (none & (best == j)))[0])
        self.clusters_reverse_index = [{c[j]: j for j in range(len(c))} for c in self.clusters]
        self.clusters_embeddings = [item_embeddings[:, c] for c in self.clusters]
        self.clusters_bias = [item_bias[c] for c in self.clusters]

    def predict_function(self, sequence, mask, k, exclude):
        """(top-k ids, number of items scored): inside the user's cluster, or over the whole catalogue with --ignore_clusters
        (rnn_cluster.py:302-325)"""
        self.engine.set_batch(sequence, mask)
        self.engine.forward()
        Bp = (self.engine.batch_size + 15) // 16 * 16
        hl = self.engine.debug_buffer("h_last").reshape(Bp, -1)
        H = self.recurrent_layer.layers[-1]
        u = np.concatenate([hl[0, :H], hl[0, hl.shape[1] // 2:hl.shape[1] // 2 + H]]) if self.recurrent_layer.bidirectional else hl[0, :H]
        if self.predict_with_clusters:
            if not hasattr(self, "clusters"):
                self.prepare_tests()
            c = int(self.head.select(1)[0])
            scores = u.dot(self.clusters_embeddings[c]) + self.clusters_bias[c]
            idx = [self.clusters_reverse_index[c][i] for i in exclude if i in self.clusters_reverse_index[c]]
            scores[idx] = -np.inf
            effective_k = min(k, len(self.clusters[c]))
            return list(self.clusters[c][np.argpartition(-scores, range(effective_k))[:effective_k]]), len(self.clusters[c])
        params = self.engine.get_all_param_values()
        scores = u.dot(params[-2]) + params[-1]
        scores[exclude] = -np.inf
        return list(np.argpartition(-scores, range(k))[:k]), self.n_items

    def top_k_recommendations(self, sequence, user_id=None, k=10, exclude=None):
        if exclude is None:
            exclude = []
        seq = sequence[-min(self.max_length, len(sequence)):]
        X = np.zeros((1, self.max_length, self._input_size()), dtype=np.int32)
        X[0, :len(seq), :] = np.array([self._get_features(x, user_id) for x in seq], dtype=np.int32)
        mask = np.zeros((1, self.max_length), dtype=np.float32)
        mask[0, :len(seq)] = 1
        should_exclude = [i[0] for i in sequence] if self.interactions_are_unique else []
        should_exclude.extend(exclude)
        return self.predict_function(X, mask, k, should_exclude)

    # ------------------------------------------------------------------ checkpoints (rnn_cluster.py:510-537)
    def save(self, filename):
        print("Save model in " + filename)
        d = os.path.dirname(filename)
        if d and not os.path.exists(d):
            os.makedirs(d)
        R, Wc = self.head.get_params()
        param = self.engine.get_all_param_values()
        param.append(R)
        param.append([Wc])
        with open(filename, "wb") as f:
            pickle.dump(param, f, protocol=2)

    def load(self, filename):
        with open(filename, "rb") as f:
            param = pickle.load(f, encoding="latin1")
        self.engine.set_all_param_values([np.asarray(i, dtype=np.float32) for i in param[:-2]])
        self.head.set_params(np.asarray(param[-2], dtype=np.float32), np.asarray(param[-1][0], dtype=np.float32))
        self.prepare_tests()
