"""The `train.py -m RNN` option surface and the small policy objects whose `.name` strings end
up in checkpoint filenames.  Python 3, written from the behaviour of the reference's
helpers/command_parser.py:34-125 (RNN options only), neural_networks/update_manager.py:3-82,
recurrent_layers.py:8-39, target_selection.py:5-53, sequence_noise.py:4-94, train.py:12-33.
"""
import argparse
import random

import numpy as np


# ----------------------------------------------------------------------------- updaters
class _Updater(object):
    kind = None

    def engine_kwargs(self):
        return dict(updater=self.kind, learning_rate=self.learning_rate, rho=getattr(self, "rho", 0.9),
                    beta1=getattr(self, "beta1", 0.9), beta2=getattr(self, "beta2", 0.999))


class Adagrad(_Updater):            # update_manager.py:24-33
    kind = "adagrad"

    def __init__(self, learning_rate=0.1):
        self.learning_rate = learning_rate
        self.name = "Ug_lr" + str(self.learning_rate)


class Adadelta(_Updater):           # update_manager.py:35-45
    kind = "adadelta"

    def __init__(self, learning_rate=1.0, rho=0.9):
        self.learning_rate, self.rho = learning_rate, rho
        self.name = "Ud_lr" + str(self.learning_rate) + "_rho" + str(self.rho)


class RMSProp(_Updater):            # update_manager.py:47-57
    kind = "rmsprop"

    def __init__(self, learning_rate=1.0, rho=0.9):
        self.learning_rate, self.rho = learning_rate, rho
        self.name = "Ur_lr" + str(self.learning_rate) + "_rho" + str(self.rho)


class NesterovMomentum(_Updater):   # update_manager.py:59-69 (momentum travels in rho)
    kind = "nesterov"

    def __init__(self, learning_rate=1.0, momentum=0.9):
        self.learning_rate, self.momentum, self.rho = learning_rate, momentum, momentum
        self.name = "Un_lr" + str(self.learning_rate) + "_m" + str(self.momentum)


class Adam(_Updater):               # update_manager.py:71-82
    kind = "adam"

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999):
        self.learning_rate, self.beta1, self.beta2 = learning_rate, beta1, beta2
        self.name = "Ua_lr" + str(self.learning_rate) + "_b1" + str(self.beta1) + "_b2" + str(self.beta2)


def update_manager_command_parser(parser):
    parser.add_argument("--u_m", dest="update_manager", choices=["adagrad", "adadelta", "rmsprop", "nesterov", "adam"],
                        help="Update mechanism", default="adam")
    parser.add_argument("--u_l", help="Learning rate", default=0.001, type=float)
    parser.add_argument("--u_rho", help="rho parameter for Adadelta and RMSProp (momentum for Nesterov momentum)",
                        default=0.9, type=float)
    parser.add_argument("--u_b1", help="Beta 1 parameter for Adam", default=0.9, type=float)
    parser.add_argument("--u_b2", help="Beta 2 parameter for Adam", default=0.999, type=float)


def get_update_manager(args):
    if args.update_manager == "adagrad":
        return Adagrad(learning_rate=args.u_l)
    if args.update_manager == "adadelta":
        return Adadelta(learning_rate=args.u_l, rho=args.u_rho)
    if args.update_manager == "rmsprop":
        return RMSProp(learning_rate=args.u_l, rho=args.u_rho)
    if args.update_manager == "nesterov":
        return NesterovMomentum(learning_rate=args.u_l, momentum=args.u_rho)
    if args.update_manager == "adam":
        return Adam(learning_rate=args.u_l, beta1=args.u_b1, beta2=args.u_b2)
    raise ValueError("Unknown update option")


# ----------------------------------------------------------------------------- recurrent stack
class RecurrentLayers(object):
    """Shape of the recurrent stack (recurrent_layers.py:18-39).  grad_clipping is always 100:
    the CLI's -g is parsed but never forwarded (command_parser.py:40, recurrent_layers.py:15)."""

    def __init__(self, layer_type="LSTM", layers=(32,), bidirectional=False, embedding_size=0, grad_clipping=100):
        if layer_type not in ("LSTM", "GRU", "Vanilla"):
            raise ValueError("Unknown layer type")
        self.layer_type = layer_type
        self.layers = [int(h) for h in layers]
        self.bidirectional = bidirectional
        self.embedding_size = embedding_size
        self.grad_clip = grad_clipping
        name = ""
        if self.bidirectional:
            name += "b" + self.layer_type + "_"
        elif self.layer_type != "LSTM":
            name += self.layer_type + "_"
        name += "gc" + str(self.grad_clip) + "_"
        if self.embedding_size > 0:
            name += "e" + str(self.embedding_size)
        name += "h" + "-".join(map(str, self.layers))
        self.name = name


def recurrent_layers_command_parser(parser):
    parser.add_argument("--r_t", dest="recurrent_layer_type", choices=["LSTM", "GRU", "Vanilla"],
                        help="Type of recurrent layer", default="GRU")
    parser.add_argument("--r_l", help="Layers' size, (eg: 100-50-50)", default="50", type=str)
    parser.add_argument("--r_bi", help="Bidirectional layers.", action="store_true")
    parser.add_argument("--r_emb", help="Add an embedding layer before the RNN. Takes the size of the embedding as "
                        "parameter, a size<1 means no embedding layer.", type=int, default=0)


def get_recurrent_layers(args):
    return RecurrentLayers(layer_type=args.recurrent_layer_type, layers=list(map(int, args.r_l.split("-"))),
                           bidirectional=args.r_bi, embedding_size=args.r_emb)


# ----------------------------------------------------------------------------- target selection
class SelectTargets(object):
    """Which of the items after the split point are the targets (target_selection.py:14-53)."""

    def __init__(self, n_targets=1, shuffle=False, bias=-1, determinist_test=True):
        self.n_targets, self.shuffle, self.bias, self.determinist_test = n_targets, shuffle, bias, determinist_test

    @property
    def name(self):
        name = "nt" + str(self.n_targets)
        if self.bias >= 0.0:
            name += "_tb" + str(self.bias)
        if self.shuffle:
            name += "_shufT"
        return name

    def set_dataset(self, dataset):
        if self.bias >= 0.0:
            pop = np.maximum(1, dataset.item_popularity)
            self.keep_prob = np.power(min(pop) / pop, self.bias)

    def __call__(self, remaining_sequence, test=False):
        if not (test and self.determinist_test):
            if self.shuffle:
                random.shuffle(remaining_sequence)
            if self.bias >= 0.0:
                remaining_sequence = [i for i in remaining_sequence if np.random.random() <= self.keep_prob[i[0]]]
        return remaining_sequence[:min(len(remaining_sequence), self.n_targets)]


def target_selection_command_parser(parser):
    parser.add_argument("--n_targets", help="Number of targets (Only for RNN with hinge, logit or logsig loss).",
                        default=1, type=int)
    parser.add_argument("--shuffle_targets", help="Instead of picking the next items in the sequence as the target(s), "
                        "the targets are picked randomly in the remaining sequence.", action="store_true")
    parser.add_argument("--rand_test_target", help="Use the exact same procedure for target selection during training "
                        "and testing. Otherwise shuffling and bias are used only during training.", action="store_true")
    parser.add_argument("--target_bias", help="Popular item are picked as item with a lower probability. Set negative "
                        "bias to avoid this procedure.", default=-1.0, type=float)


def get_target_selection(args):
    return SelectTargets(n_targets=args.n_targets, shuffle=args.shuffle_targets, bias=args.target_bias,
                         determinist_test=(not args.rand_test_target))


# ----------------------------------------------------------------------------- sequence noise
class SequenceNoise(object):
    """Optional perturbation of the training sequences (sequence_noise.py:15-94); the CLI
    default is the identity, whose name is the empty string."""

    def __init__(self, dropout=0.0, swap=0.0, ratings_perturb=0.0, shuf=0.0, shuf_std=0.0):
        self.dropout, self.swap, self.ratings_perturb, self.shuf, self.shuf_std = dropout, swap, ratings_perturb, shuf, shuf_std
        if self.dropout < 0.0 or self.dropout >= 1.0:
            raise ValueError("Dropout should be in [0,1)")
        if self.swap < 0.0 or self.swap >= 1.0:
            raise ValueError("Swapping probability should be in [0,1)")
        if self.ratings_perturb < 0.0 or self.ratings_perturb >= 1.0:
            raise ValueError("Rating perturbation probability should be in [0,1)")
        name = []
        if self.dropout > 0:
            name.append("do" + str(self.dropout))
        if self.swap > 0:
            name.append("sw" + str(self.swap))
        if self.ratings_perturb > 0:
            name.append("rp" + str(self.ratings_perturb))
        if self.shuf > 0:
            name.append("sh" + str(self.shuf) + "-" + str(self.shuf_std))
        self.name = "_".join(name)

    def __call__(self, sequence_generator):
        """sequence_noise.py:52-94: dropout (sequences left with < 2 items are skipped), swap of
        consecutive items (never twice the same item), swap with an item int(N(0,1)*shuf_std)
        away, then +-0.5 rating perturbation clamped to [1, 5]."""
        for sequence, user in sequence_generator:
            if self.dropout > 0.0:
                sequence = [i for i in sequence if np.random.random() >= self.dropout]
                if len(sequence) < 2:
                    continue
            if self.swap > 0.0:
                i = 0
                while i < len(sequence) - 1:
                    if np.random.random() < self.swap:
                        sequence[i], sequence[i + 1] = sequence[i + 1], sequence[i]
                        i += 1
                    i += 1
            if self.shuf > 0.0:
                for i in range(len(sequence)):
                    if np.random.random() < self.shuf:
                        other = max(0, min(len(sequence) - 1, int(np.random.randn() * self.shuf_std) + i))
                        sequence[i], sequence[other] = sequence[other], sequence[i]
            if self.ratings_perturb > 0:
                for i in range(len(sequence)):
                    if np.random.random() < self.ratings_perturb:
                        if np.random.random() < 0.5:
                            sequence[i][1] = min(5, sequence[i][1] + 0.5)
                        else:
                            sequence[i][1] = max(1, sequence[i][1] - 0.5)
            yield sequence, user


def sequence_noise_command_parser(parser):
    parser.add_argument("--n_dropout", help="Dropout probability", default=0.0, type=float)
    parser.add_argument("--n_swap", help="Probability of swapping two consecutive items", default=0.0, type=float)
    parser.add_argument("--n_shuf", help="Probability of swapping two random items", default=0.0, type=float)
    parser.add_argument("--n_shuf_std", help="The distance between the two items to be swapped is drawn from a normal "
                        "distribution whose std is defined by this parameter", default=5.0, type=float)
    parser.add_argument("--n_ratings", help="Probability of changing the rating.", default=0.0, type=float)


def get_sequence_noise(args):
    return SequenceNoise(dropout=args.n_dropout, swap=args.n_swap, ratings_perturb=args.n_ratings, shuf=args.n_shuf,
                         shuf_std=args.n_shuf_std)


# ----------------------------------------------------------------------------- CLI surface
def predictor_command_parser(parser):
    """The options of helpers/command_parser.py:34-77 that reach the RNN models (defaults kept,
    including the surprising ones: -b 16, --max_length 30)."""
    parser.add_argument("-m", dest="method", choices=["RNN"], help="Method", default="RNN")
    parser.add_argument("-b", dest="batch_size", help="Batch size", default=16, type=int)
    parser.add_argument("-r", dest="regularization", help="Regularization (positive for L2, negative for L1)",
                        default=0.0, type=float)
    parser.add_argument("-g", dest="gradient_clipping", help="Gradient clipping (parsed, never used: always 100)",
                        default=100, type=int)
    parser.add_argument("--loss", help="Loss function: TOP1, BPR, Blackout (sampling), hinge, logit, logsig (multi-targets) or CCE",
                        default="CCE", type=str)
    parser.add_argument("--pb", help="Popularity based (for RNNMargin).", action="store_true")
    parser.add_argument("--balance", help="Balance between false positive and false negative error (for RNNMargin).",
                        default=1., type=float)
    parser.add_argument("--min_access", help="Estimation of minimum access probability (for RNNMargin).", default=0.05, type=float)
    parser.add_argument("--sampling", help="Number of sample for the computation of the loss in RNNSampling",
                        default=32.0, type=float)
    parser.add_argument("--sampling_bias", help="0. means uniform sampling, 1. means proportional to the item frequency",
                        default=0.0, type=float)
    parser.add_argument("--db", dest="diversity_bias", help="Diversity bias", default=0.0, type=float)
    parser.add_argument("--rf", help="Use rating features.", action="store_true")
    parser.add_argument("--mf", help="Use movie features.", action="store_true")
    parser.add_argument("--uf", help="Use users features.", action="store_true")
    parser.add_argument("--max_length", help="Maximum length of sequences during training (for RNNs)", default=30, type=int)
    parser.add_argument("--repeated_interactions", help="The model can recommend items with which the user already "
                        "interacted", action="store_true")
    # RNNCluster (command_parser.py:70-77)
    parser.add_argument("--c_sampling", help="Number of sample for the clustering loss. If unset, the same samples are used for the "
                        "recommendation loss and for the clustering loss.", default=-1, type=int)
    parser.add_argument("--ignore_clusters", help="Don't use clusters during test. Useful to observe the influence of clustering",
                        action="store_true")
    parser.add_argument("--clusters", help="Number of clusters. If unset, no clustering is used", default=-1, type=int)
    parser.add_argument("--init_scale", help="Initial scale of the softmax and sigmoid in the clustering method.", default=1., type=float)
    parser.add_argument("--scale_growing_rate", help="Rate of the geometric growth of the sigmoid/softmax scale in the clustering method.",
                        default=1., type=float)
    parser.add_argument("--max_scale", help="Max scale of the softmax and sigmoid in the clustering method.", default=50, type=float)
    parser.add_argument("--csn", help="Cluster selection noise", default=0., type=float)
    parser.add_argument("--cluster_type", choices=["softmax", "mix", "sigmoid"], help="Type of clusters. Softmax puts every item in 1 "
                        "and only 1 cluster. Sigmoid allow puts items in 0 to n clusters. Mix puts items in 1 to n clusters.",
                        default="mix", type=str)
    update_manager_command_parser(parser)
    recurrent_layers_command_parser(parser)
    sequence_noise_command_parser(parser)
    target_selection_command_parser(parser)


def training_command_parser(parser):                   # train.py:12-27
    parser.add_argument("--tshuffle", help="Shuffle sequences during training.", action="store_true")
    parser.add_argument("--extended_set", help="Use extended training set.", action="store_true")
    parser.add_argument("-d", dest="dataset", help="Directory name of the dataset.", default="", type=str)
    parser.add_argument("--dir", help="Directory name to save model.", default="", type=str)
    parser.add_argument("--save", choices=["All", "Best", "None"], help="Policy for saving models.", default="Best")
    parser.add_argument("--metrics", help="Metrics for validation, comma separated", default="sps", type=str)
    parser.add_argument("--time_based_progress", help="Follow progress based on time rather than iterations.",
                        action="store_true")
    parser.add_argument("--load_last_model", help="Load Last model before starting training.", action="store_true")
    parser.add_argument("--progress", help="Progress intervals", default="2.", type=str)
    parser.add_argument("--mpi", help="Max progress intervals", default=np.inf, type=float)
    parser.add_argument("--max_iter", help="Max number of iterations", default=np.inf, type=float)
    parser.add_argument("--max_time", help="Max training time in seconds", default=np.inf, type=float)
    parser.add_argument("--min_iter", help="Min number of iterations before showing progress", default=0.0, type=float)


def num(s):                                            # train.py:29-33
    try:
        return int(s)
    except ValueError:
        return float(s)


def command_parser(*sub_command_parser, argv=None):    # helpers/command_parser.py:22-32
    parser = argparse.ArgumentParser()
    for scp in sub_command_parser:
        scp(parser)
    return parser.parse_args(argv)


def get_predictor(args):
    """helpers/command_parser.py:84-125, RNN branch (:113-123)."""
    from .models import RNNOneHot, RNNSampling, RNNMargin, RNNCluster
    if args.mf or args.uf:
        raise ValueError("--mf/--uf need feature tables the reference never loads (rnn_base.py:27-29): unsupported")
    common = dict(interactions_are_unique=(not args.repeated_interactions), max_length=args.max_length,
                  updater=get_update_manager(args), target_selection=get_target_selection(args),
                  sequence_noise=get_sequence_noise(args), recurrent_layer=get_recurrent_layers(args),
                  use_ratings_features=args.rf, use_movies_features=args.mf, use_users_features=args.uf,
                  batch_size=args.batch_size)
    if args.clusters > 0:                                               # command_parser.py:114-115
        return RNNCluster(cluster_selection_noise=args.csn, loss=args.loss, predict_with_clusters=(not args.ignore_clusters),
                          sampling_bias=args.sampling_bias, sampling=args.sampling, cluster_sampling=args.c_sampling,
                          init_scale=args.init_scale, scale_growing_rate=args.scale_growing_rate, max_scale=args.max_scale,
                          n_clusters=args.clusters, cluster_type=args.cluster_type, **common)
    if args.loss == "CCE":
        return RNNOneHot(diversity_bias=args.diversity_bias, regularization=args.regularization, **common)
    if args.loss in ("BPR", "TOP1", "Blackout"):
        return RNNSampling(loss_function=args.loss, diversity_bias=args.diversity_bias, sampling=args.sampling,
                           sampling_bias=args.sampling_bias, **common)
    if args.loss in ("hinge", "logit", "logsig"):                       # command_parser.py:118-119
        return RNNMargin(loss_function=args.loss, balance=args.balance, popularity_based=args.pb, min_access=args.min_access, **common)
    raise ValueError("Unknown loss for the RNN model")
