"""`python -m sbr_amd.test -d DIR -m RNN ...` -- the reference's test.py (test.py:22-160) for the RNN path: finds the
checkpoints the training options name, ranks every test user's continuation and prints / appends the metrics.

Same file discovery (`models/<dir><model filename with _ne*_>`), same metric names, same `results/` file.  Deliberate
differences: users are scored `batch_size` at a time through the engine (identical top-k ids to one-row calls, see
RNNBase.batched_test_predictions); the results line puts a tab between the epoch count and the first metric -- the
reference writes them glued together (test.py:92) and then cannot parse its own file again (:112-118)."""
import glob
import os
import re
import sys
import time

import numpy as np

from . import options as parse
from .data import DataHandler, Evaluator


def get_file_name(predictor, args):                                   # test.py:22-23
    return args.dir + re.sub("_ml" + str(args.max_length), "_ml" + str(args.training_max_length),
                             predictor._get_model_filename(args.number_of_batches))


def find_models(predictor, dataset, args):                            # test.py:25-34
    f = dataset.dirname + "models/" + get_file_name(predictor, args)
    print(f)
    return np.array(sorted(glob.glob(f))) if args.number_of_batches == "*" else f


def save_file_name(predictor, dataset, args):                         # test.py:36-41
    if not args.save:
        return None
    return re.sub(r"_ne\*_", "_", dataset.dirname + "results/" + get_file_name(predictor, args))


def run_tests(predictor, model_file, dataset, args, get_full_recommendation_list=False, k=10):
    """test.py:43-77: the first half of every test sequence is viewed, the rest is the goal;
    `top_k_recommendations` feeds the last max_length viewed items and excludes EVERY viewed item (rnn_base.py:132-159).
    Users whose viewed half fits the window (the engine derives the exclusion from its input) are ranked batch_size at a
    time; longer ones, and full rankings (--save_rank), go through top_k_recommendations one by one."""
    predictor.load(model_file)
    evaluator = Evaluator(dataset, k=k)
    if get_full_recommendation_list:
        k = dataset.n_items
    start = time.perf_counter()
    pending, results, order = [], {}, []

    def flush():
        if not pending:
            return
        X = np.zeros((len(pending), predictor.max_length, predictor._input_size()), dtype=np.int32)
        mask = np.zeros((len(pending), predictor.max_length), dtype=np.float32)
        for i, (_, viewed, user_id) in enumerate(pending):
            X[i, :len(viewed), :] = np.array([predictor._get_features(x, user_id) for x in viewed], dtype=np.int32)
            mask[i, :len(viewed)] = 1
        ids = predictor.engine.test_function((X, mask), k=k, exclude_seen=predictor.interactions_are_unique)
        for (n, _, _), row in zip(pending, ids):
            results[n] = list(row[row >= 0])      # -1 = a place the row had no rankable item for (include/sbr_rnn.h)
        del pending[:]
    for n, (sequence, user_id) in enumerate(dataset.test_set(epochs=1)):
        num_viewed = int(len(sequence) / 2)
        viewed, goal = sequence[:num_viewed], [i[0] for i in sequence[num_viewed:]]
        if len(goal) == 0:
            raise ValueError
        order.append((n, goal))
        if get_full_recommendation_list or k > 64 or len(viewed) > predictor.max_length or len(viewed) == 0:
            results[n] = list(predictor.top_k_recommendations(viewed, user_id=user_id, k=k))
        else:
            pending.append((n, viewed, user_id))
            if len(pending) == predictor.batch_size:
                flush()
    flush()
    for n, goal in order:
        evaluator.add_instance(goal, results[n])
    print("Timer: ", time.perf_counter() - start)
    evaluator.nb_of_dp = dataset.n_items
    return evaluator


def print_results(ev, metrics, file=None, n_batches=None, print_full_rank_comparison=False):   # test.py:79-103
    for m in metrics:
        if m not in ev.metrics:
            raise ValueError("Unkown metric: " + m)
        print(m + "@" + str(ev.k) + ": ", ev.metrics[m]())
    values = "\t".join(str(ev.metrics[m]()) for m in metrics)
    if file is not None:
        if not os.path.exists(os.path.dirname(file)):
            os.makedirs(os.path.dirname(file))
        with open(file, "a") as f:
            f.write(str(n_batches) + "\t" + values + "\n")
        if print_full_rank_comparison:
            with open(file + "_full_rank", "a") as f:
                for data in ev.get_rank_comparison():
                    f.write("\t".join(map(str, data)) + "\n")
    else:
        print("-\t" + values, file=sys.stderr)


def extract_number_of_epochs(filename):                               # test.py:105-107
    return float(re.search(r"_ne([0-9]+(\.[0-9]+)?)_", filename).group(1))


def get_last_tested_batch(filename):                                  # test.py:109-120
    if filename is not None and os.path.isfile(filename):
        line = None
        with open(filename) as f:
            for line in f:
                pass
        return float(line.split()[0]) if line else 0
    return 0


def test_command_parser(parser):                                      # test.py:122-130
    parser.add_argument("-d", dest="dataset", help="Directory name of the dataset.", default="", type=str)
    parser.add_argument("-i", dest="number_of_batches", help="Number of epochs, if not set it will compare all the available models",
                        default=-1, type=int)
    parser.add_argument("-k", dest="nb_of_predictions", help='Number of predictions to make. It is the "k" in "prec@k", "rec@k", etc.',
                        default=10, type=int)
    parser.add_argument("--metrics", help="List of metrics to compute, comma separated",
                        default="sps,recall,item_coverage,user_coverage,blockbuster_share", type=str)
    parser.add_argument("--save", help="Save results to a file", action="store_true")
    parser.add_argument("--dir", help="Model directory.", default="", type=str)
    parser.add_argument("--save_rank", help="Save the full comparison of goal and prediction ranking.", action="store_true")


def main(argv=None):                                                  # test.py:132-160
    args = parse.command_parser(parse.predictor_command_parser, test_command_parser, argv=argv)
    args.training_max_length = args.max_length
    if args.number_of_batches == -1:
        args.number_of_batches = "*"
    dataset = DataHandler(dirname=args.dataset)
    predictor = parse.get_predictor(args)
    predictor.prepare_model(dataset)
    files = find_models(predictor, dataset, args)
    metrics = args.metrics.split(",")
    results = []
    if args.number_of_batches == "*":
        output_file = save_file_name(predictor, dataset, args)
        last_tested_batch = get_last_tested_batch(output_file)
        batches = np.array([extract_number_of_epochs(f) for f in files])
        order = np.argsort(batches)
        for i, j in enumerate(order):
            if batches[j] > last_tested_batch:
                ev = run_tests(predictor, files[j], dataset, args, get_full_recommendation_list=args.save_rank, k=args.nb_of_predictions)
                print("-------------------")
                print("(", i + 1, "/", len(files), ") results on " + files[j])
                print_results(ev, metrics, file=output_file, n_batches=batches[j], print_full_rank_comparison=args.save_rank)
                results.append((files[j], {m: ev.metrics[m]() for m in metrics}))
    else:
        ev = run_tests(predictor, files, dataset, args, get_full_recommendation_list=args.save_rank, k=args.nb_of_predictions)
        print_results(ev, metrics, file=save_file_name(predictor, dataset, args), print_full_rank_comparison=args.save_rank)
        results.append((files, {m: ev.metrics[m]() for m in metrics}))
    return results


if __name__ == "__main__":
    main()
