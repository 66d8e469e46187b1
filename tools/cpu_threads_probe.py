"""How does the torch-CPU port of the step scale with threads on this host? (bench.py default)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import rnn_oracle as O, torch_ref as R
cell, layers, n_items, loss, ns = bench.CONFIGS["c2"]
hb = bench.synth_batches(1, 256, 200, n_items, ns, "full", 1235)[0]
params = O.init_params(cell, layers, n_items, np.random.default_rng(42), dtype=np.float32)
cb = dict(X=hb["X"], mask=hb["mask"], target=hb["target"], samples=hb["samples"], pop=hb["pop"])
print("host cores:", os.cpu_count(), flush=True)
for n in (8, 16, 32, 64):
    if n > (os.cpu_count() or 1):
        break
    torch.set_num_threads(n)
    tr = R.TorchTrainer(params, dict(cell=cell, layers=layers, loss=loss), O.recurrent_param_shapes)
    t0 = time.perf_counter(); tr.train_function(cb); t1 = time.perf_counter(); tr.train_function(cb); t2 = time.perf_counter()
    print("threads %3d: first %.2f s, second %.2f s" % (n, t1 - t0, t2 - t1), flush=True)
