"""In-kernel cycle counters of the CLUSTER recurrent kernels (SBR_FLAG_PROFILE_REC, sbr_rec_cl.hip):
per-step cycles by phase for the first 8 row tiles.   python tools/cl_prof.py [c4]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import rnn_oracle as O
from sbr_amd.engine import RNNEngine
cell, layers, n_items, loss, ns = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
B, T = (int(sys.argv[2]) if len(sys.argv) > 2 else 256), 200
C = 8
eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=T, batch_size=B, loss=loss, n_samples=ns, flags=8)
eng.set_all_param_values(O.init_params(cell, layers, n_items, np.random.default_rng(42), dtype=np.float32))
hb = bench.synth_batches(1, B, T, n_items, ns, "full", 1235)[0]
eng.set_batch(hb["X"], None, hb["target"], hb["samples"] if loss != "CCE" else None, hb["pop"], lengths=hb["lengths"])
for _ in range(3):
    eng.train_step(sync=True)
raw = eng.debug_buffer("prof").view(np.uint64).reshape(2, -1)
c16 = eng.query("rec_rows_fwd") == 16      # the 16-row kernels rec_*_c16 (one workgroup per 16 units)
if c16:
    # tick indices of sbr_rec_c16.hip (pc[i] -> column 3 + i); the "|" columns exist in counter mode only (an extra wait)
    names = (("rec_fwd_c16", ((0, "exchange wait"), (1, "requests+MFMA+partials"), (2, "reduce barrier"), (3, "gate math+publish")),
              (5, "everything since the poll acknowledged")),
             ("rec_bwd_c16", ((0, "gate math+A planes+barrier"), (1, "MFMA+block stores"), (2, "exchange wait"), (3, "sum+requests"),
                              (4, "reduce barrier"), (6, "reduce")), (5, "block stores acknowledged")))
    if max(layers) > 256 and os.environ.get("SBR_C16_TWO_LEVEL", "1") != "0":      # rec_bwd_c16t: two hand-offs per step
        names = (names[0], ("rec_bwd_c16t", ((0, "gate math+level-1 publish"), (1, "level-1 wait"), (2, "stage+barrier"), (3, "stores+MFMA+blocks"),
                                             (4, "level-2 wait"), (6, "sum+requests+barrier"), (7, "reduce")), (5, "everything since the level-1 poll acknowledged")))
    for k, (name, ph, extra) in enumerate(names):
        ne = min(32, B // 8)
        p = raw[k][:ne * 4 * 16].reshape(ne, 4, 16).astype(np.float64)          # [tile * C + member][wave][16]
        tot, real = p[..., 0], p[..., 1]
        print("%s: kernel %.1f us (realtime), shader clock %.0f MHz, %.0f cycles/step" % (
            name, real.mean() / 100.0, (tot / real * 100.0).mean(), tot.mean() / T))
        for w in range(4):
            print("   wave%d   polls/step %.2f  " % (w, p[:, w, 2].mean() / T) + "  ".join("%s %5.0f" % (nm, p[:, w, 3 + i].mean() / T) for i, nm in ph)
                  + "  | %s %5.0f" % (extra[1], p[:, w, 3 + extra[0]].mean() / T))
else:
  names = (("rec_fwd_cl", ("exchange wait", "publish+barrier", "LDS+MFMA", "reduce barrier", "gate math+stores")),
         ("rec_bwd_cl", ("gate math+stores", "exchange wait", "split+barrier", "LDS+MFMA", "reduce barrier")))
  for k, (name, ph) in enumerate(names):
    p = raw[k][:4 * C * 4 * 16].reshape(4, C, 4, 16).astype(np.float64)      # [tile][member][wave][8]
    tot, real = p[..., 0], p[..., 1]
    print("%s: kernel %.1f us (realtime), shader clock %.0f MHz, %.0f cycles/step" % (
        name, real.mean() / 100.0, (tot / real * 100.0).mean(), tot.mean() / T))
    for w, wn in enumerate(("wave0 (tile0,fin)", "wave1 (tile1,fin)", "wave2 (tile0,K-hi)", "wave3 (tile1,K-hi)")):
        print("   %-20s polls/step %.2f  " % (wn, p[:, :, w, 2].mean() / T) + "  ".join("%s %5.0f" % (ph[i], p[:, :, w, 3 + i].mean() / T) for i in range(5))
              + "  | fine " + " ".join("%.0f" % (p[:, :, w, 3 + i].mean() / T) for i in range(5, 10)))
eng.close()
