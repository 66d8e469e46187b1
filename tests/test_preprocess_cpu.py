"""CPU tests of the preprocess mirror (sbr_amd.preprocess; reference: preprocess.py:45-214): the files it writes are
the ones the training path parses (data.py), with the reference's filtering / split / sequence rules."""
import os

import numpy as np
import pytest

from sbr_amd import preprocess as P
from sbr_amd.data import DataHandler


def raw_file(tmp_path, n_users=60, n_items=40, seed=0, sep="::"):
    """ML-1M-style `user::item::rating::timestamp`, shuffled on disk; timestamps are unique and increase with the
    position inside a user's history, item ids are strings of the form i<k>."""
    rng = np.random.default_rng(seed)
    lines, truth = [], {}
    t = 1000000
    for u in range(n_users):
        n = int(rng.integers(1, 30))
        hist = []
        for _ in range(n):
            it = int(min(n_items - 1, rng.zipf(1.3) - 1))
            t += int(rng.integers(1, 50))
            hist.append((it, int(rng.integers(1, 6)), t))
        truth["u%03d" % u] = hist
        lines += ["u%03d%s%d%s%d%s%d" % (u, sep, it, sep, r, sep, ts) for it, r, ts in hist]
    rng.shuffle(lines)
    path = tmp_path / "ratings.dat"
    path.write_text("\n".join(lines) + "\n")
    return str(path), truth


def read_sequences(path):
    out = {}
    with open(path) as f:
        for line in f:
            tok = line.split()
            if tok:
                out[int(tok[0])] = (list(map(int, tok[1::2])), list(map(int, tok[2::2])))
    return out


def run(tmp_path, **kw):
    path, truth = raw_file(tmp_path)
    argv = ["-f", path, "--columns", "uirt", "--sep", "::", "--yes"]
    for k, v in kw.items():
        argv += ["--" + k, str(v)]
    return P.main(argv), truth


def test_filters_ids_and_chronology(tmp_path):
    root, truth = run(tmp_path, min_user_activity=3, min_item_pop=4)
    # expected filtering, restated on plain dicts: users, items, users again
    inter = [(u, it, r, ts) for u, h in truth.items() for it, r, ts in h]
    def count(idx):
        c = {}
        for x in inter:
            c[x[idx]] = c.get(x[idx], 0) + 1
        return c
    cu = count(0); inter = [x for x in inter if cu[x[0]] >= 3]
    ci = count(1); inter = [x for x in inter if ci[x[1]] >= 4]
    cu = count(0); inter = [x for x in inter if cu[x[0]] >= 3]
    users, items = sorted({x[0] for x in inter}), sorted({x[1] for x in inter})
    umap = {u: k for k, u in enumerate(users)}
    imap = {i: k for k, i in enumerate(items)}
    want = {}
    for u, it, r, ts in sorted(inter, key=lambda x: x[3]):
        want.setdefault(umap[u], []).append((imap[it], r))
    got = {}
    for name in ("train", "val", "test"):
        seqs = read_sequences(root + "data/%s_set_sequences" % name)
        assert not set(seqs) & set(got)                       # every user in exactly one set
        got.update(seqs)
    # users with one interaction left have no sequence line (unless last of their set)
    for u, (its, rs) in got.items():
        assert list(zip(its, rs)) == want[u]
    assert all(len(v) < 2 for u, v in want.items() if u not in got)
    # the mapping files
    rows = [l.split("\t") for l in open(root + "data/user_id_mapping").read().splitlines()[1:]]
    assert {r[0]: int(r[1]) for r in rows} == umap
    rows = [l.split("\t") for l in open(root + "data/item_id_mapping").read().splitlines()[1:]]
    assert {int(r[0]): int(r[1]) for r in rows} == imap


def test_stats_triplets_and_extended_training_file(tmp_path):
    root, _ = run(tmp_path, val_size=6, test_size=0.2)
    stats = {l.split()[0]: list(map(int, l.split()[1:])) for l in open(root + "data/stats").read().splitlines()[1:]}
    sets = {}
    for name, key in (("train", "Train"), ("val", "Val"), ("test", "Test")):
        trip = np.loadtxt(root + "data/%s_set_triplets" % name, dtype=np.int64, ndmin=2)
        sets[name] = trip
        users, counts = np.unique(trip[:, 0], return_counts=True)
        assert stats[key] == [len(users), len(np.unique(trip[:, 1])), len(trip), counts.max()]
    assert stats["Full"][2] == sum(len(t) for t in sets.values())
    assert stats["Val"][0] <= 6 and stats["Test"][0] <= round(0.2 * stats["Full"][0])      # drawn with replacement
    assert stats["Val"][0] >= 1 and stats["Test"][0] >= 1
    # train_set_sequences+ = training sequences + first half (floor) of every val / test sequence
    plus = open(root + "data/train_set_sequences+").read().splitlines()
    train = open(root + "data/train_set_sequences").read().splitlines()
    assert plus[:len(train)] == train
    halves = plus[len(train):]
    full = {**read_sequences(root + "data/val_set_sequences"), **read_sequences(root + "data/test_set_sequences")}
    seen = 0
    for line in halves:
        tok = line.split()
        u, its = int(tok[0]), list(map(int, tok[1::2]))
        assert its == full[u][0][:len(full[u][0]) // 2]
        seen += 1
    assert seen == len(full)


def test_split_is_seeded_and_loadable(tmp_path):
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir(); (tmp_path / "c").mkdir()
    ra, _ = run(tmp_path / "a", seed=7)
    rb, _ = run(tmp_path / "b", seed=7)
    rc, _ = run(tmp_path / "c", seed=8)
    read = lambda r: open(r + "data/test_set_sequences").read()
    assert read(ra) == read(rb) and read(ra) != read(rc)
    ds = DataHandler(dirname=ra)                              # the training path's own loader
    assert ds.n_items > 0 and ds.training_set.n_users > 0
    n = sum(1 for _ in ds.training_set(epochs=1))
    assert n == len(read_sequences(ra + "data/train_set_sequences"))
    assert os.path.isdir(ra + "models") and os.path.isdir(ra + "results")
    assert ds.item_popularity.sum() == np.loadtxt(ra + "data/train_set_triplets", ndmin=2).shape[0]


def test_not_enough_users(tmp_path):
    path, _ = raw_file(tmp_path, n_users=6)
    with pytest.raises(ValueError):
        P.main(["-f", path, "--columns", "uirt", "--sep", "::", "--yes", "--val_size", "4", "--test_size", "4"])


def test_matches_the_reference_outputs_byte_for_byte(tmp_path):
    # tests/golden/preprocess/: written by the reference's own preprocess.py (tools/make_preprocess_golden.py)
    import shutil
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "preprocess")
    shutil.copy(os.path.join(gold, "ratings.dat"), tmp_path)
    root = P.main(["-f", str(tmp_path / "ratings.dat"), "--yes"] + open(os.path.join(gold, "ARGS")).read().split())
    names = [n for n in sorted(os.listdir(gold)) if n not in ("ARGS", "ratings.dat")]
    assert len(names) == 10
    for name in names:
        assert open(root + "data/" + name).read() == open(os.path.join(gold, name)).read(), name
