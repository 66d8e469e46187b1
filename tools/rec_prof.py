"""In-kernel cycle counters of the recurrent kernels (SBR_FLAG_PROFILE_REC): effective shader clock
(s_memtime vs the 100 MHz s_memrealtime) and the split work / barrier-wait per wave.
The profiling instances follow the product form the launch would take (fp16x3 by default, bf16x6 with SBR_X6_F16=0;
profiles/round1_h_rec_phases_c2.txt is the bf16x6 forward of round 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from oracle import rnn_oracle as O
from sbr_amd.engine import RNNEngine
cell, layers, n_items, loss, ns = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
B, T = 256, 200
eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=T, batch_size=B, loss=loss, n_samples=ns, flags=8)
eng.set_all_param_values(O.init_params(cell, layers, n_items, np.random.default_rng(42), dtype=np.float32))
hb = bench.synth_batches(1, B, T, n_items, ns, "full", 1235)[0]
eng.set_batch(hb["X"], None, hb["target"], hb["samples"] if loss != "CCE" else None, hb["pop"], lengths=hb["lengths"])
for _ in range(3):
    eng.train_step(sync=True)
raw = eng.debug_buffer("prof").view(np.uint64).reshape(2, B // 16, 16, 8)
for k, name in enumerate(("rec_fwd", "rec_bwd")):
    p = raw[k].astype(np.float64)
    nw = min(8, int((p[0, :, 0] > 0).sum()))
    tot, real, work, bar = (p[:, :nw, i] for i in range(4))
    mhz = tot / real * 100.0
    print("%s: waves/block %d | kernel %.1f us (realtime) | shader clock %.0f MHz (min %.0f max %.0f)" % (
        name, nw, real.mean() / 100.0, mhz.mean(), mhz.min(), mhz.max()))
    if k == 0 and raw[0, 0, 8, 0] > 0:       # pipelined kernels: step-100 timeline of block 0, slots of "waves" 8..15
        tl = raw[0, 0, 8:16, :].astype(np.int64)
        t0 = tl[:, 0].min()
        print("   step-100 timeline of block 0 (cycles after the first wave's loop top): top, k-block 0..3 operands ready, MFMAs done, published, next top")
        for w in range(8):
            print("     wave %d: %s" % (w, (tl[w] - t0).tolist()))
        print("   waiting for the partner's pipe turn, per step: %s" % np.round(p[0, :8, 4] / T).astype(int).tolist())
        print("   loop top -> first MFMA %s | MFMA phase %s | gate math + publish %s" % tuple(
            np.round(p[0, :8, i] / T).astype(int).tolist() for i in (5, 6, 7)))
        p = p[:, :8]
        nw = 8
        tot, real, work, bar = (p[:, :nw, i] for i in range(4))
    if k == 1:
        print("   bwd split by wave of block 0: gate-math %s  store/LDS-write/prefetch issue %s  LDS-read+MFMA %s" % (
            np.round(p[0, :nw, 4] / T).astype(int).tolist(), np.round(p[0, :nw, 5] / T).astype(int).tolist(),
            np.round(p[0, :nw, 6] / T).astype(int).tolist()))
    print("   per step: total %.0f cyc = work %.0f + barrier-wait %.0f   (by wave of block 0: work %s  bar %s)" % (
        tot.mean() / T, work.mean() / T, bar.mean() / T, np.round(work[0] / T).astype(int).tolist(), np.round(bar[0] / T).astype(int).tolist()))
eng.close()
