"""(diagnostic) the overlapped-tail parity cases with the tail switched on / off: python tools/dbg_tail.py"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import parity_util as PU

for ov in ("1", "0"):
    os.environ["SBR_TAIL_OVERLAP"] = ov
    for cell, sc in (("Vanilla", 0.05), ("GRU", 0.1)):
        for kw in (dict(N=300, B=64, T=131, full=True), dict(N=300, B=37, T=70, zipf=True)):
            r = PU.compare_step(cell, [128], "CCE", scale=sc, gap=1e-4, **kw)
            print("overlap", ov, cell, kw, {k: float("%.3g" % v) for k, v in r.items() if not k.startswith("grad:")}, flush=True)
