"""End-to-end throughput of the training LOOP (batch building + step), native device builder vs the reference-style host
generator, on an ML-1M-shaped synthetic dataset written in the reference's on-disk format.
    python tools/bench_train_loop.py [--iters 1000] [--host-iters 8]
Prints one JSON line."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def write_dataset(root, n_users=6040, n_items=3706, seed=0):
    rng = np.random.default_rng(seed)
    d = os.path.join(root, "data")
    os.makedirs(d)
    os.makedirs(os.path.join(root, "models"))
    lengths = np.clip(rng.lognormal(4.6, 0.9, size=n_users).astype(int), 20, 2300)      # ML-1M: mean ~165, max 2314
    rank = np.arange(1, n_items + 1, dtype=np.float64)
    p = (1.0 / rank) / (1.0 / rank).sum()
    trip = []
    for name, users in (("train", range(n_users)), ("val", range(200)), ("test", range(200))):
        with open(os.path.join(d, name + "_set_sequences"), "w") as f:
            for u in users:
                items = rng.choice(n_items, size=lengths[u], p=p)
                f.write(str(u) + " " + " ".join("%d 4.0" % i for i in items) + "\n")
                if name == "train":
                    trip.append(np.bincount(items, minlength=n_items))
    np.save(os.path.join(d, "training_set_item_popularity.npy"), np.sum(trip, axis=0).astype(np.float64))
    with open(os.path.join(d, "stats"), "w") as f:
        f.write("set n_users n_items n_interactions longest_sequence\n")
        for name in ("Full", "Train", "Val", "Test"):
            f.write("%s %d %d %d %d\n" % (name, n_users, n_items, int(lengths.sum()), int(lengths.max())))
    return root + "/"


def run(root, native, iters, B, T):
    os.environ["SBR_NATIVE_BATCHES"] = "1" if native else "0"
    from sbr_amd import options as parse, train as Tr
    from sbr_amd.data import DataHandler
    argv = ["-d", root, "-b", str(B), "--max_length", str(T), "--r_t", "GRU", "--r_l", "128", "--max_iter", str(iters),
            "--progress", str(10 ** 9), "--save", "None"]
    args = parse.command_parser(parse.predictor_command_parser, parse.training_command_parser, Tr.early_stopping_command_parser, argv=argv)
    predictor = parse.get_predictor(args)
    dataset = DataHandler(dirname=root)
    predictor.prepare_model(dataset)
    predictor.train(dataset, max_iter=3, progress=10 ** 9, autosave="None")            # warm-up: parse file, first launches
    t0 = time.perf_counter()
    predictor.train(dataset, max_iter=iters, progress=10 ** 9, autosave="None")
    predictor.engine.synchronize()
    dt = time.perf_counter() - t0
    predictor.engine.close()
    return iters * B / dt, dt / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--host-iters", type=int, default=8)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--max_length", type=int, default=200)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        root = write_dataset(os.path.join(tmp, "ds"))
        nat, nat_s = run(root, True, a.iters, a.batch, a.max_length)
        host, host_s = run(root, False, a.host_iters, a.batch, a.max_length)
    print(json.dumps({"metric": "end-to-end training-loop user-sequences/s (batch building + step, cost read back every step)",
                      "workload": "GRU-128, N=3706, B=%d, T=%d, ML-1M-shaped synthetic file, 6040 users" % (a.batch, a.max_length),
                      "native_builder": round(nat, 1), "native_ms_per_iteration": round(nat_s * 1e3, 3),
                      "host_generator": round(host, 1), "host_ms_per_iteration": round(host_s * 1e3, 3),
                      "ratio": round(nat / host, 1)}))


if __name__ == "__main__":
    main()
