"""`python -m sbr_amd.train -d DIR -m RNN ...` -- the reference's train.py (train.py:35-57) for the RNN
path, on the MI355X engine.  Same options, same progress output, same checkpoint files."""
from . import options as parse
from .data import DataHandler


class _EarlyStopper(object):
    """helpers/early_stopping.py:19-32: called with the epochs of the validation passes and the metric's values so far;
    a lower-is-better metric is negated so that the rules below always look for increases."""

    def __init__(self, higher_is_better=True):
        self.higher_is_better = higher_is_better

    def __call__(self, epochs, values):
        if not self.higher_is_better:
            values = [-v for v in values]
        return self.decide(epochs, values)


class StopAfterN(_EarlyStopper):
    """Stop once each of the last n validation values failed to improve on the one before it
    (helpers/early_stopping.py:34-52: n consecutive non-improvements, not "best is n evaluations old")."""

    def __init__(self, n=3, **kwargs):
        super(StopAfterN, self).__init__(**kwargs)
        self.n = n

    def decide(self, epochs, values):
        if len(values) <= self.n:
            return False
        return all(values[-1 - i] <= values[-2 - i] for i in range(self.n))


class WaitWorstCaseTimesX(_EarlyStopper):
    """Stop when the wait since the last best value exceeds x times the longest wait there has been between two
    successive bests, and min_wait (helpers/early_stopping.py:55-86); waits are measured in epochs."""

    def __init__(self, x=2.0, min_wait=1.0, **kwargs):
        super(WaitWorstCaseTimesX, self).__init__(**kwargs)
        self.x, self.min_wait = x, min_wait

    def decide(self, epochs, values):
        best, best_epoch, longest = values[0], epochs[0], 0
        for epoch, v in zip(epochs[1:], values[1:]):
            if v > best:
                longest = max(longest, epoch - best_epoch)
                best, best_epoch = v, epoch
        wait = epochs[-1] - best_epoch
        if longest == 0:
            return wait > self.min_wait
        print("current wait : ", round(wait, 3), " longest wait : ", round(longest, 3), " ratio : ", wait / longest, " / ", self.x)
        return wait > max(self.min_wait, longest * self.x)


def early_stopping_command_parser(parser):      # helpers/early_stopping.py:4-9
    parser.add_argument("--es_m", dest="early_stopping_method", choices=["WorstTimesX", "StopAfterN", "None"],
                        help="Early stopping method", default="None")
    parser.add_argument("--es_n", help="N parameter (for StopAfterN)", default=5, type=int)
    parser.add_argument("--es_x", help="X parameter (for WorstTimesX)", default=2.0, type=float)
    parser.add_argument("--es_min_wait", help="Mininum wait before stopping (for WorstTimesX)", default=1.0, type=float)
    parser.add_argument("--es_LiB", help="Lower is better for validation score.", action="store_true")


def get_early_stopper(args):                    # helpers/early_stopping.py:11-17
    if args.early_stopping_method == "StopAfterN":
        return StopAfterN(n=args.es_n, higher_is_better=(not args.es_LiB))
    if args.early_stopping_method == "WorstTimesX":
        return WaitWorstCaseTimesX(x=args.es_x, min_wait=args.es_min_wait, higher_is_better=(not args.es_LiB))
    return None


def main(argv=None):
    args = parse.command_parser(parse.predictor_command_parser, parse.training_command_parser,
                                early_stopping_command_parser, argv=argv)
    predictor = parse.get_predictor(args)
    dataset = DataHandler(dirname=args.dataset, extended_training_set=args.extended_set, shuffle_training=args.tshuffle)
    predictor.prepare_model(dataset)
    return predictor.train(dataset,
                           save_dir=dataset.dirname + "models/" + args.dir,
                           time_based_progress=args.time_based_progress,
                           progress=parse.num(args.progress),
                           autosave=args.save,
                           max_progress_interval=args.mpi,
                           max_iter=args.max_iter,
                           min_iterations=args.min_iter,
                           max_time=args.max_time,
                           early_stopping=get_early_stopper(args),
                           load_last_model=args.load_last_model,
                           validation_metrics=args.metrics.split(","))


if __name__ == "__main__":
    main()
