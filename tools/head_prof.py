"""Phase stamps of the one-launch head (csrc/sbr_head.hip, SBR_FLAG_PROFILE_REC): per workgroup the 100 MHz clock at
  0 start | 1 W chunk in LDS, h in registers | 2 logits done | 3 chunk statistics published | 4 all chunks' statistics read and
  combined | 5 dlogits stored | 6 dh MFMAs done | 7 slab stored
printed relative to the launch's first stamp, mean / min / max over the workgroups, in microseconds.
    python tools/head_prof.py [c2|c1]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import rnn_oracle as O
from sbr_amd.engine import RNNEngine
cell, layers, n_items, loss, ns = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
B, T = 256, 200
eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=T, batch_size=B, loss=loss, n_samples=ns, flags=8)
eng.set_all_param_values(O.init_params(cell, layers, n_items, np.random.default_rng(42), dtype=np.float32))
hb = bench.synth_batches(1, B, T, n_items, ns, "full", 1235)[0]
eng.set_batch(hb["X"], None, hb["target"], hb["samples"] if loss != "CCE" else None, hb["pop"], lengths=hb["lengths"])
cc = eng.query("head_fused")
print("head_fused:", cc)
for _ in range(4):
    eng.train_step(sync=True)
raw = eng.debug_buffer("prof_head").view(np.uint64)[:256 * 8].reshape(256, 8).astype(np.int64)
raw = raw[: (B // 16) * cc]
t0 = raw[:, 0].min()
names = ["start", "W chunk in LDS", "logits", "stats published", "stats combined", "dlogits stored", "dh MFMAs", "slab stored"]
for i, nm in enumerate(names):
    v = (raw[:, i] - t0) / 100.0
    print("%-18s mean %6.2f  min %6.2f  max %6.2f us" % (nm, v.mean(), v.min(), v.max()))
eng.close()
