"""Debug helper: python tools/cmp_case.py CELL H[,H2] B T [seed [scale]] -- one compare_step against the oracle, key errors printed."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import parity_util as PU  # noqa: E402

cell, layers, B, T = sys.argv[1], [int(x) for x in sys.argv[2].split(",")], int(sys.argv[3]), int(sys.argv[4])
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 0
scale = float(sys.argv[6]) if len(sys.argv) > 6 else None
r = PU.compare_step(cell, layers, "CCE", N=41, B=B, T=T, seed=seed, scale=scale)
print({k: float(v) for k, v in r.items() if k in ("h_last", "grad_worst", "cost", "predict_scores", "params_after_2_steps", "topk_mismatch")})
