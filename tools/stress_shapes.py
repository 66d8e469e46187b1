"""Unusual shapes through the whole step (no oracle: finite cost, cost decreasing, no error): large batches (cluster
grids beyond the CU count), long sequences, ragged lengths, big catalogues."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import rnn_oracle as O
from sbr_amd.engine import RNNEngine

CASES = [("LSTM", [256], 5000, "CCE", 0, 512, 100), ("LSTM", [256], 5000, "CCE", 0, 1024, 60), ("GRU", [256], 3000, "BPR", 16, 768, 50),
         ("GRU", [100], 5000, "CCE", 0, 1000, 300), ("LSTM", [128], 2000, "CCE", 0, 2048, 40), ("Vanilla", [256, 128], 1000, "CCE", 0, 200, 30),
         ("GRU", [128], 3706, "CCE", 0, 13, 200)]
for cell, layers, N, loss, S, B, T in CASES:
    eng = RNNEngine(cell=cell, layers=layers, n_items=N, max_length=T, batch_size=B, loss=loss, n_samples=S, learning_rate=0.005)
    eng.set_all_param_values(O.init_params(cell, layers, N, np.random.default_rng(1), dtype=np.float32))
    hb = bench.synth_batches(1, B, T, N, S, "ml1m", 7)[0]
    eng.set_batch(hb["X"], None, hb["target"], hb["samples"] if loss != "CCE" else None, hb["pop"], lengths=hb["lengths"])
    t0 = time.time()
    costs = [eng.train_step(sync=True) for _ in range(6)]
    ok = all(np.isfinite(costs)) and costs[-1] < costs[0]
    print("%-8s %-10s N=%-5d %-5s B=%-5d T=%-4d cluster=%d rpt=%d  cost %.4f -> %.4f  %.1f ms/step  %s" % (
        cell, layers, N, loss, B, T, eng.query("cluster"), eng.query("rows_per_workgroup"), costs[0], costs[-1],
        (time.time() - t0) / 6 * 1e3, "OK" if ok else "BAD"))
    assert ok
    eng.close()
print("stress OK")
