"""Phase stamps of the one-launch SAMPLED head (csrc/sbr_head.hip: head_sampled_kernel, SBR_FLAG_PROFILE_REC): per workgroup the 100 MHz clock at
  0 start | 1 activations done | 2 loss done | 3 gradient stored | 4 dh MFMAs done | 5 partial rows in LDS | 6 dh stored
relative to the launch's first stamp, mean / min / max over the workgroups, in microseconds.      python tools/samp_prof.py [c3|c5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import rnn_oracle as O
from sbr_amd.engine import RNNEngine
cell, layers, n_items, loss, ns = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c3"]
B, T = 256, 200
eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=T, batch_size=B, loss=loss, n_samples=ns, flags=8)
hb = bench.synth_batches(1, B, T, n_items, ns, "full", 1235)[0]
eng.set_batch(hb["X"], None, hb["target"], hb["samples"], hb["pop"], lengths=hb["lengths"])
for _ in range(3):
    eng.train_step(sync=True)
raw = eng.debug_buffer("prof_head").view(np.uint64)[:(B // 16) * 8].reshape(B // 16, 8).astype(np.int64)
t0 = raw[:, 0].min()
for i, nm in enumerate(["start", "activations", "loss", "gradient stored", "dh MFMAs", "partials in LDS", "dh stored"]):
    v = (raw[:, i] - t0) / 100.0
    print("%-18s mean %6.2f  min %6.2f  max %6.2f us" % (nm, v.mean(), v.min(), v.max()))
eng.close()
