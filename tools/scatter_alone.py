"""The stand-alone scatter-add launch of a configuration's shape (sbr_debug_scatter: what bench.py reports as kernels.scatter_unfused):
   python tools/scatter_alone.py [c2 c1 ...]      (SBR_LIB=... for a same-box A/B)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from sbr_amd.engine import RNNEngine
for cfg in (sys.argv[1:] or ["c2"]):
    cell, layers, n_items, loss, ns = bench.CONFIGS[cfg]
    eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=200, batch_size=256, loss=loss, n_samples=ns)
    hb = bench.synth_batches(1, 256, 200, n_items, ns, "full", 1235)[0]
    eng.set_batch(hb["X"], None, hb["target"], hb["samples"] if loss != "CCE" else None, hb["pop"], lengths=hb["lengths"])
    eng.train_step()
    us, entries, rows = eng.debug_scatter(30)
    ghp = {"GRU": 3, "LSTM": 4, "Vanilla": 1}[cell] * ((layers[0] + 31) // 32 * 32)
    print("%s: %.1f us, %d entries, %d rows, %.2f TB/s" % (cfg, us, entries, rows, (entries + rows) * ghp * 4 / us / 1e6))
    eng.close()
