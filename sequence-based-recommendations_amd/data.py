"""Dataset access + top-k metrics for the RNN path, Python 3.  Behaviour follows the reference's
helpers/data_handling.py:12-174 (on-disk format written by preprocess.py:153-214) and
helpers/evaluation.py:16-216; unlike the reference, sequence files are parsed ONCE into integer
arrays instead of re-splitting every text line every epoch (data_handling.py:142-145).
"""
import os
import random

import numpy as np

DEFAULT_DIR = "../../data/"      # data_handling.py:9


class SequenceGenerator(object):
    """Streams (sequence=[[item, rating], ...], user_id) from a `*_set_sequences` file whose lines
    are `user item rating item rating ...`; `epochs` is the fractional pass counter the training
    loop records (data_handling.py:139, rnn_base.py:312)."""

    def __init__(self, filename, shuffle=False):
        self.filename, self.shuffle, self.epochs = filename, shuffle, 0.0

    def load(self):
        self.users, self.items, self.ratings = [], [], []
        with open(self.filename, "r") as f:
            for line in f:
                tok = line.split()
                if not tok:
                    continue
                n = (len(tok) - 1) // 2
                self.users.append(tok[0])
                self.items.append(np.array(tok[1:1 + 2 * n:2], dtype=np.int64))
                self.ratings.append(np.array(tok[2:2 + 2 * n:2], dtype=np.float64))
        self.order = list(range(len(self.users)))

    def __call__(self, min_length=2, max_length=None, length_choice="max", subsequence="contiguous", epochs=np.inf):
        if not hasattr(self, "users"):
            self.load()
        counter = 0
        self.epochs = 0.0
        while counter < epochs:
            counter += 1
            print("Opening file ({})".format(counter))
            if self.shuffle:
                random.shuffle(self.order)
            for j, li in enumerate(self.order):
                self.epochs = counter - 1 + j / len(self.order)
                sequence = [[int(i), float(r)] for i, r in zip(self.items[li], self.ratings[li])]
                if len(sequence) < min_length:
                    continue
                this_max = len(sequence) if max_length is None else max_length
                if length_choice == "random":
                    length = np.random.randint(min_length, min(this_max, len(sequence)) + 1)
                elif length_choice == "max":
                    length = min(this_max, len(sequence))
                else:
                    raise ValueError('Unrecognised length_choice option. Authorised values are "random" and "max" ')
                if length < len(sequence):
                    if subsequence == "random":
                        sequence = [sequence[i] for i in sorted(random.sample(range(len(sequence)), length))]
                    elif subsequence == "contiguous":
                        start = np.random.randint(0, len(sequence) - length + 1)
                        sequence = sequence[start:start + length]
                    elif subsequence == "begining":
                        sequence = sequence[:length]
                    else:
                        raise ValueError('Unrecognised subsequence option. Authorised values are "random", '
                                         '"contiguous" and "begining".')
                yield sequence, self.users[li]


class NativeBatchBuilder(object):
    """Training batches built on the device (sbr_dataset_* / sbr_build_batch in include/sbr_rnn.h):
    the stand-in for `_gen_mini_batch(sequence_noise(dataset.training_set()))` + `_prepare_input`
    (rnn_base.py:373-420, rnn_one_hot.py:83-106, rnn_sampling.py:159-194, rnn_margin.py:112-147) for every option that leaves
    the batch plan of a pass computable on the host: --rf, --n_targets, --shuffle_targets, --db, --sampling_bias, and the
    sequence noise (--n_dropout, --n_swap, --n_shuf, --n_ratings: a device pass over the users before each pass is planned,
    sequence_noise.py:52-94), and --target_bias (the rows of a pass are then planned on the host by the library, the device
    packs them: sbr_dataset_set_target_bias).

    The training file is parsed once (SequenceGenerator.load) and uploaded as CSR.  Per pass the users are walked in
    file order, or reshuffled like data_handling.py:139-141 with --tshuffle; the walk prints the reference's
    "Opening file (n)" line and keeps `training_set.epochs` (the fractional pass counter the training loop records,
    rnn_base.py:312) up to date per batch.  `next()` makes the next batch the engine's current batch."""

    def __init__(self, engine, training_set, n_items, batch_size, pop_db=None, sample_cdf=None, seed=None, ratings=False,
                 shuffle_targets=False, noise=None, keep_prob=None, n_targets=1):
        from .engine import DeviceDataset
        if not hasattr(training_set, "users"):
            training_set.load()
        self.engine, self.ts, self.batch_size = engine, training_set, int(batch_size)
        lengths = np.array([len(x) for x in training_set.items], dtype=np.int64)
        offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
        items = np.concatenate(training_set.items).astype(np.int32) if len(lengths) else np.zeros(0, np.int32)
        self.ds = DeviceDataset(engine, items, offsets, n_items)
        self.ds.set_tables(pop_db, sample_cdf)
        if ratings or shuffle_targets:      # --rf: the rating index rides beside the item index; --shuffle_targets
            r = (np.concatenate(training_set.ratings) if len(lengths) else np.zeros(0)) if ratings else None
            self.ds.set_options(r, shuffle_targets)
        if keep_prob is not None:           # --target_bias: rows planned on the host (a row may run out of targets)
            self.ds.set_target_bias(keep_prob, n_targets, seed=int(seed if seed is not None else random.getrandbits(62)))
        self.n_users = len(lengths)
        self.noise = noise if (noise is not None and getattr(noise, "name", "") != "") else None      # a SequenceNoise
        self.seed = int(seed if seed is not None else random.getrandbits(63))
        self.passes, self.cursor, self.n_batches, self.built = 0, 0, 0, 0
        self._frac = np.zeros(0)

    def _plan(self):
        self.passes += 1
        print("Opening file ({})".format(self.passes))
        order = None
        if self.ts.shuffle:
            random.shuffle(self.ts.order)
            order = np.asarray(self.ts.order, dtype=np.int32)
        if self.noise is not None:      # this pass's noised copy of every sequence; the plan below sees its lengths
            nz = self.noise
            self.ds.noise_pass(nz.dropout, nz.swap, nz.shuf, nz.shuf_std, nz.ratings_perturb, seed=self.seed + 7919 * self.passes)
        self.n_batches = self.ds.plan_pass(order, self.batch_size)
        seg = self.ds.segments()
        # fraction of the pass consumed when batch b is complete: position of its last user in the walk
        pos = np.empty(self.n_users, dtype=np.int64)
        pos[np.arange(self.n_users) if order is None else order] = np.arange(self.n_users)
        self._frac = np.zeros(self.n_batches)
        if len(seg):
            last_user = np.zeros(self.n_batches, dtype=np.int64)
            last_user[seg[:, 3]] = seg[:, 0]               # later segments of a batch overwrite earlier ones
            self._frac = pos[last_user] / float(max(1, self.n_users))
        self.cursor = 0

    def __iter__(self):
        return self

    def __next__(self):
        guard = 0
        while self.cursor >= self.n_batches:
            self._plan()
            guard += 1
            if guard > 4 + self.batch_size:      # no pass can complete a batch: nothing to train on
                raise StopIteration
        b = self.cursor
        self.engine.build_batch(self.ds, b, self.seed + self.built)
        self.cursor += 1
        self.built += 1
        self.ts.epochs = self.passes - 1 + float(self._frac[b])
        return b

    next = __next__

    def close(self):
        self.ds.close()


class DataHandler(object):
    """Directory resolver + `stats` loader + item popularity cache (data_handling.py:12-102)."""

    def __init__(self, dirname, extended_training_set=False, shuffle_training=False):
        self.dirname = self._get_path(dirname)
        self.extended_training_set = extended_training_set
        train = "data/train_set_sequences+" if extended_training_set else "data/train_set_sequences"
        self.training_set = SequenceGenerator(self.dirname + train, shuffle=shuffle_training)
        self.validation_set = SequenceGenerator(self.dirname + "data/val_set_sequences")
        self.test_set = SequenceGenerator(self.dirname + "data/test_set_sequences")
        self._load_stats()

    @property
    def item_popularity(self):
        if not hasattr(self.training_set, "_item_pop"):
            cache = self.dirname + "data/training_set_item_popularity.npy"
            if os.path.isfile(cache):
                self.training_set._item_pop = np.load(cache)
            else:
                pop = np.zeros(self.n_items)
                with open(self.dirname + "data/train_set_triplets") as f:
                    for line in f:
                        pop[int(line.split()[1])] += 1
                self.training_set._item_pop = pop
                np.save(cache, pop)
        return self.training_set._item_pop

    def _get_path(self, dirname):
        a, b = os.path.exists(dirname), os.path.exists(DEFAULT_DIR + dirname + "/")
        if a and b:
            print('WARNING: ambiguous directory name, both "' + dirname + '" and "' + DEFAULT_DIR + dirname +
                  '" exist. "' + dirname + '" is used.')
        if a:
            return dirname
        if b:
            return DEFAULT_DIR + dirname + "/"
        raise ValueError("Dataset not found")

    def _load_stats(self):
        with open(self.dirname + "data/stats", "r") as f:
            f.readline()                                    # column titles
            for obj in (self, self.training_set, self.validation_set, self.test_set):
                obj.n_users, obj.n_items, obj.n_interactions, obj.longest_sequence = map(int, f.readline().split()[1:])
        if self.extended_training_set:
            self.training_set.n_users, self.training_set.n_items = self.n_users, self.n_items
            self.training_set.n_interactions += (self.validation_set.n_interactions + self.test_set.n_interactions) // 2


class Evaluator(object):
    """Top-k metrics over (goal, predictions) instances (evaluation.py:16-216)."""

    def __init__(self, dataset, k=10):
        self.instances, self.dataset, self.k = [], dataset, k
        # evaluation.py:24-35: the names test.py's --metrics accepts
        self.metrics = {"sps": self.sps, "recall": self.average_recall, "precision": self.average_precision,
                        "ndcg": self.average_ndcg, "item_coverage": self.item_coverage, "user_coverage": self.user_coverage,
                        "blockbuster_share": self.blockbuster_share, "novelty": self.average_novelty, "assr": self.assr}

    def add_instance(self, goal, predictions):
        self.instances.append([list(goal), list(predictions)])

    def _top(self, prediction):
        return prediction[:min(len(prediction), self.k)]

    def average_precision(self):
        s = sum(len(set(g) & set(self._top(p))) / min(len(p), self.k) for g, p in self.instances if len(p) > 0)
        return s / len(self.instances)

    def average_recall(self):
        s = sum(len(set(g) & set(self._top(p))) / len(g) for g, p in self.instances if len(g) > 0)
        return s / len(self.instances)

    def average_ndcg(self):
        ndcg = 0.0
        for goal, prediction in self.instances:
            if len(prediction) > 0:
                dcg = max_dcg = 0.0
                for i, p in enumerate(self._top(prediction)):
                    if i < len(goal):
                        max_dcg += 1.0 / np.log2(2 + i)
                    if p in goal:
                        dcg += 1.0 / np.log2(2 + i)
                ndcg += dcg / max_dcg
        return ndcg / len(self.instances)

    def sps(self):
        return sum(int(g[0] in self._top(p)) for g, p in self.instances) / len(self.instances)

    def user_coverage(self):
        return sum(int(len(set(g) & set(self._top(p))) > 0) for g, p in self.instances) / len(self.instances)

    def get_correct_predictions(self):
        out = []
        for g, p in self.instances:
            out.extend(list(set(g) & set(self._top(p))))
        return out

    def item_coverage(self):
        return len(set(self.get_correct_predictions()))

    def average_novelty(self):
        """evaluation.py:90-101 ("Auralist"): -mean log2(popularity share) of the recommended items."""
        total = float(np.sum(self.dataset.item_popularity))
        nov = 0.0
        for _, p in self.instances:
            if len(p) > 0:
                top = np.asarray(self._top(p), dtype=np.int64)
                nov += float(np.sum(np.log2(self.dataset.item_popularity[top] / total))) / min(len(p), self.k)
        return -nov / len(self.instances)

    def get_rank_comparison(self):
        """evaluation.py:196-205: (position in the goal list, position in the recommendation list) pairs; needs full
        recommendation lists (test.py --save_rank)."""
        out = []
        for goal, prediction in self.instances:
            pos = np.argsort(prediction)[goal]
            out.extend(list(enumerate(pos)))
        return out

    def assr(self):
        """evaluation.py:207-216: average search-space reduction (1 without the clustering head)."""
        nb = getattr(self, "nb_of_dp", 0)
        return self.dataset.n_items / nb if nb and nb > 0 else 1

    def blockbuster_share(self):
        correct = self.get_correct_predictions()
        nb_pop = self.dataset.n_items // 100
        pop_items = set(np.argpartition(-self.dataset.item_popularity, nb_pop)[:nb_pop].tolist())
        if len(correct) == 0:
            return 0
        return len([i for i in correct if i in pop_items]) / len(correct)
