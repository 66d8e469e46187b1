import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import parity_util as PU
from oracle import rnn_oracle as O
cell, H, N, B, T = sys.argv[1], int(sys.argv[2]), 61, 37, 9
params, cfg, batch = PU.build_case(cell, [H], "CCE", N, B, T)
h_or, caches = O.network_forward(params, cell, [H], batch["X"], batch["mask"])
hs_or = caches[0]["hs"]                     # (T+1, B, H)
for trial in range(3):
    eng = PU.engine_for(cfg, N, B, T)
    eng.set_all_param_values(params)
    eng.set_batch(batch["X"], batch["mask"], batch["target"], None, batch["pop"])
    eng.forward()
    Bp = (B + 15) // 16 * 16
    hs = eng.debug_buffer("hs0").reshape(T + 1, Bp, -1)[:, :B, :H]
    xt = eng.debug_buffer("xt0").reshape(T, Bp, -1)[:, :B, :H]
    xt_or = caches[0]["xt"]
    err = np.abs(hs - hs_or).max(axis=2)      # (T+1, B)
    print("trial", trial, "xt err", np.abs(xt - xt_or).max(), "hs err by t:", np.round(err.max(axis=1), 5).tolist())
    bad = np.argwhere(err > 1e-3)
    print("   first bad (t,row):", bad[:8].tolist(), "lens of those rows:", [int(batch["mask"][b].sum()) for _, b in bad[:8]])
    eng.close()
