"""In-kernel cycle counters of the fp16x3 BPTT chain of C2 (SBR_FLAG_PROFILE_REC) with the overlapped tail on, with its
kernels on one stream (no consumers beside the chain) and off:  python tools/tail_prof.py
Per step and wave of block 0: total | wait for the ring | loop top -> loads / stores issued | operands + pipe gate + MFMAs |
spin on the publish counters | wait at the pipe gate."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import rnn_oracle as O

cell, layers, n_items, loss, ns = bench.CONFIGS["c2"]
B, T = 256, 200
for mode in ("1", "2", "0"):
    os.environ["SBR_TAIL_OVERLAP"] = mode
    from sbr_amd.engine import RNNEngine
    eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=T, batch_size=B, loss=loss, n_samples=ns, flags=8)
    eng.set_all_param_values(O.init_params(cell, layers, n_items, np.random.default_rng(42), dtype=np.float32))
    hb = bench.synth_batches(1, B, T, n_items, ns, "full", 1235)[0]
    eng.set_batch(hb["X"], None, hb["target"], None, hb["pop"], lengths=hb["lengths"])
    for _ in range(5):
        eng.train_step(sync=True)
    raw = eng.debug_buffer("prof").view(np.uint64).reshape(2, B // 16, 16, 8)
    p = raw[1].astype(np.float64)[:, :8]
    tot, real = p[:, :, 0], p[:, :, 1]
    print("SBR_TAIL_OVERLAP=%s tail_chunks=%d: rec_bwd %.1f us, %.0f MHz, %.0f cycles per step" % (
        mode, eng.query("tail_chunks"), real.mean() / 100.0, (tot / real * 100.0).mean(), tot.mean() / T))
    for name, i in (("ring wait", 7), ("N phase (incl. ring wait)", 5), ("operands + gate + MFMA", 6), ("spin", 3), ("pipe gate", 4)):
        print("   %-28s by wave of block 0: %s   mean over blocks: %s" % (name, np.round(p[0, :, i] / T).astype(int).tolist(),
                                                                          np.round(p[:, :, i].mean(axis=0) / T).astype(int).tolist()))
    eng.close()
