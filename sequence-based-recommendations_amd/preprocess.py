"""`python -m sbr_amd.preprocess -f ratings.dat --columns uirt --sep ::` -- the on-disk formats the training path reads
(SURVEY 8f rank 4), written the way the reference's preprocess.py does (preprocess.py:45-214), Python 3.

A raw interaction file (one line per interaction; columns named by --columns: u user, i item, r rating, t timestamp,
anything else ignored) becomes, next to it:
    data/user_id_mapping, data/item_id_mapping      original id <-> consecutive id (ids = rank of the sorted originals)
    data/{train,val,test}_set_triplets              `user<TAB>item<TAB>rating`, chronological
    data/{train,val,test}_set_sequences             `user item rating item rating ...`, one line per user (what
                                                    data.py: SequenceGenerator / NativeBatchBuilder parse)
    data/train_set_sequences+                       training sequences + the first half of every val / test sequence
    data/stats                                      n_users, n_items, n_interactions, longest_sequence per set
    data/README, results/README, models/
Behaviour kept from the reference, quirks included: users below --min_user_activity are dropped, then items below
--min_item_pop, then users again (items may end up below the threshold, preprocess.py:63-83); the val / test users are
drawn with `np.random.choice(users, n)` -- WITH replacement, so a set can hold fewer than n users -- test first, then
val from the rest, from `np.random.seed(--seed)` (preprocess.py:137-145, 289); a user with a single interaction gets no
sequence line unless it is the last user of its set (preprocess.py:160-172); "half" keeps floor(n/2) interactions.
Different on purpose: `--yes` skips the are-you-sure prompt (raw_input in the reference, preprocess.py:26-32).
"""
import argparse
import os
import sys

import numpy as np
import pandas as pd


def command_parser(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("-f", dest="filename", help="Input file", required=True, type=str)
    p.add_argument("--columns", default="uit", type=str,
                   help='Order of the columns in the file (eg: "uirt"), u for user, i for item, t for timestamp, r for '
                        'rating. Without r every interaction gets rating 1; without t the file order is taken as '
                        'chronological. Extra columns are ignored. Default: uit')
    p.add_argument("--sep", default=r"\s+", type=str, help="Separator between the columns (regular expression). Default: whitespace")
    p.add_argument("--min_user_activity", default=2, type=int, help="Users with fewer interactions are removed. Default: 2")
    p.add_argument("--min_item_pop", default=5, type=int, help="Items with fewer interactions are removed. Default: 5")
    p.add_argument("--val_size", default=0.1, type=float,
                   help="Number of users in the validation set; a value in (0,1) is a fraction of the users. Default: 0.1")
    p.add_argument("--test_size", default=0.1, type=float, help="Idem for the test set. Default: 0.1")
    p.add_argument("--seed", default=1, type=int, help="Seed of the random train/val/test split")
    p.add_argument("--yes", action="store_true", help="Do not ask before creating files")
    args = p.parse_args(argv)
    args.dirname = os.path.dirname(os.path.abspath(args.filename)) + "/"
    return args


def load_data(filename, columns, separator):
    """DataFrame with columns u, i, r (+ t) in chronological order (preprocess.py:45-66)."""
    data = pd.read_csv(filename, sep=separator, names=list(columns), index_col=False, usecols=range(len(columns)),
                       engine="python", header=None)
    if "r" not in columns:
        data["r"] = 1
    if "t" in columns:
        if np.issubdtype(data["t"].dtype, np.integer):
            data["t"] = pd.to_datetime(data["t"], unit="s")
        else:
            data["t"] = pd.to_datetime(data["t"])
        data = data.sort_values("t")
    return data


def remove_rare_elements(data, min_user_activity, min_item_pop):
    """users, items, users again (preprocess.py:68-90)."""
    def keep(frame, col, n):
        counts = frame[col].value_counts()
        return frame[frame[col].isin(counts.index[counts >= n])]
    return keep(keep(keep(data, "u", min_user_activity), "i", min_item_pop), "u", min_user_activity)


def map_ids(data, dirname):
    """consecutive ids = position of the original id among the sorted originals (pandas categorical codes,
    preprocess.py:92-117); writes the two mapping files."""
    data = data.copy()
    for col, name in (("u", "user_id_mapping"), ("i", "item_id_mapping")):
        originals, codes = np.unique(data[col].to_numpy(), return_inverse=True)
        pd.DataFrame({"original_id": originals, "new_id": np.arange(len(originals))}).to_csv(
            dirname + "data/" + name, sep="\t", index=False)
        data[col] = codes
    return data


def split_data(data, nb_val_users, nb_test_users, dirname):
    """every user in exactly one set; test users drawn first, val users from the rest (preprocess.py:119-150)."""
    nb_users = data["u"].nunique()
    if nb_val_users < 1:
        nb_val_users = round(nb_val_users * nb_users)
    if nb_test_users < 1:
        nb_test_users = round(nb_test_users * nb_users)
    nb_val_users, nb_test_users = int(nb_val_users), int(nb_test_users)
    if nb_users <= nb_val_users + nb_test_users:
        raise ValueError("Not enough users in the dataset: choose less users for validation and test splits")

    def extract(frame, n):
        chosen = np.random.choice(frame["u"].unique(), n)          # with replacement, like the reference
        mask = frame["u"].isin(chosen)
        return frame[mask], frame[~mask]

    test_set, rest = extract(data, nb_test_users)
    val_set, train_set = extract(rest, nb_val_users)
    for name, frame in (("train", train_set), ("val", val_set), ("test", test_set)):
        frame.to_csv(dirname + "data/" + name + "_set_triplets", sep="\t", columns=["u", "i", "r"], index=False, header=False)
    return train_set, val_set, test_set


def gen_sequences(data, half=False):
    """[user, item, rating, item, rating, ...] per user in user order, interactions in time order
    (preprocess.py:152-172)."""
    data = data.sort_values("u", kind="mergesort")                # stable: keeps the time order inside a user
    users = data["u"].to_numpy()
    items, ratings = data["i"].tolist(), data["r"].tolist()
    bounds = np.flatnonzero(np.diff(users)) + 1
    starts = np.concatenate([[0], bounds]).astype(int) if len(users) else np.zeros(0, int)
    ends = np.concatenate([bounds, [len(users)]]).astype(int) if len(users) else np.zeros(0, int)
    for k, (lo, hi) in enumerate(zip(starts, ends)):
        n = hi - lo
        is_last = k == len(starts) - 1
        if n < 2 and not is_last:                                 # the reference's len(seq) > 3 test, skipped for the last user
            continue
        if half:
            n = n // 2
        seq = [int(users[lo])]
        for p in range(lo, lo + n):
            seq.extend([items[p], ratings[p]])
        yield seq
    if not len(users):
        yield []


def make_sequence_format(train_set, val_set, test_set, dirname):
    def write(path, frames, mode="w", half=False):
        with open(path, mode) as f:
            for frame in frames:
                for s in gen_sequences(frame, half=half):
                    f.write(" ".join(map(str, s)) + "\n")
    write(dirname + "data/train_set_sequences", [train_set])
    write(dirname + "data/val_set_sequences", [val_set])
    write(dirname + "data/test_set_sequences", [test_set])
    write(dirname + "data/train_set_sequences+", [train_set])
    write(dirname + "data/train_set_sequences+", [val_set, test_set], mode="a", half=True)


def save_data_stats(data, train_set, val_set, test_set, dirname):
    def stats(frame):
        longest = int(frame["u"].value_counts().max()) if len(frame) else 0
        return "\t".join(map(str, [frame["u"].nunique(), frame["i"].nunique(), len(frame.index), longest]))
    with open(dirname + "data/stats", "w") as f:
        f.write("set\tn_users\tn_items\tn_interactions\tlongest_sequence\n")
        for name, frame in (("Full", data), ("Train", train_set), ("Val", val_set), ("Test", test_set)):
            f.write(name + "\t" + stats(frame) + "\n")


def make_readme(dirname, val_set, test_set):
    with open(dirname + "data/README", "w") as f:
        f.write("Files written by sbr_amd.preprocess (formats of the reference's preprocess.py):\n"
                "user_id_mapping / item_id_mapping: original id and new consecutive id, tab separated.\n"
                "<set>_set_triplets: user<TAB>item<TAB>rating per interaction, chronological.\n"
                "<set>_set_sequences: one line per user: user item rating item rating ...\n"
                "train_set_sequences+: the training sequences plus the first half of every validation / test sequence.\n"
                "stats: users, items, interactions and longest sequence per set.\n"
                "Users are partitioned at random: %d validation users, %d test users, the rest for training.\n"
                % (val_set["u"].nunique(), test_set["u"].nunique()))
    with open(dirname + "results/README", "w") as f:
        f.write("One line per tested model (sbr_amd.test): number of epochs, then the requested metrics @10, tab separated.\n")


def main(argv=None):
    args = command_parser(argv)
    np.random.seed(seed=args.seed)
    if not args.yes:
        print("This program will create a lot of files and directories in " + args.dirname)
        if input("Are you sure that you want to do that ? [y/n]") != "y":
            sys.exit(0)
    for sub in ("data", "models", "results"):
        os.makedirs(args.dirname + sub, exist_ok=True)
    print("Load data...")
    data = load_data(args.filename, args.columns, args.sep)
    print("Remove inactive users and rare items...")
    data = remove_rare_elements(data, args.min_user_activity, args.min_item_pop)
    print("Map original users and items ids to consecutive numerical ids...")
    data = map_ids(data, args.dirname)
    print("Split data into training, validation and test sets...")
    train_set, val_set, test_set = split_data(data, args.val_size, args.test_size, args.dirname)
    print("Save the sets in the sequences format...")
    make_sequence_format(train_set, val_set, test_set, args.dirname)
    save_data_stats(data, train_set, val_set, test_set, args.dirname)
    make_readme(args.dirname, val_set, test_set)
    print("Data ready!")
    return args.dirname


if __name__ == "__main__":
    main()
