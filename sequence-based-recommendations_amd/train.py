"""`python -m sbr_amd.train -d DIR -m RNN ...` -- the reference's train.py (train.py:35-57) for the RNN
path, on the MI355X engine.  Same options, same progress output, same checkpoint files."""
from . import options as parse
from .data import DataHandler


class StopAfterN(object):
    """early stopping helpers/early_stopping.py:19-50: stop when the best value is n evaluations old."""

    def __init__(self, n=3, higher_is_better=True):
        self.n, self.direction = n, 1 if higher_is_better else -1

    def __call__(self, epochs, criterion):
        import numpy as np
        if len(criterion) <= self.n:
            return False
        c = np.array(criterion) * self.direction
        return int(np.argmax(c)) < len(c) - self.n


def early_stopping_command_parser(parser):      # helpers/early_stopping.py:4-17 (StopAfterN flavour)
    parser.add_argument("--es_m", dest="early_stopping_method", choices=["WorstTimesX", "StopAfterN", "None"],
                        help="Early stopping method", default="None")
    parser.add_argument("--es_n", help="N parameter (for StopAfterN)", default=5, type=int)


def get_early_stopper(args):
    if args.early_stopping_method == "StopAfterN":
        return StopAfterN(n=args.es_n)
    if args.early_stopping_method == "None":
        return None
    raise NotImplementedError("early stopping method " + args.early_stopping_method)


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
