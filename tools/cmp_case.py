"""Debug helper: one compare_step against the oracle per argument, key errors printed.
    python tools/cmp_case.py CELL:H[,H2]:B:T[:key=value...] ...      keys: loss N S seed scale full zipf steps updater gf (grad_floor)
                                                                       env.NAME=value sets an environment variable for that case"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import parity_util as PU  # noqa: E402

KEEP = ("h_last", "grad_worst", "cost", "predict_scores", "topk_mismatch", "grad_worst_steps", "params_twin")
for spec in sys.argv[1:]:
    f = spec.split(":")
    cell, layers, B, T = f[0], [int(x) for x in f[1].split(",")], int(f[2]), int(f[3])
    kw = dict(loss="CCE", N=41, S=0, seed=0, scale=None, full=False, zipf=False, steps=1, updater="adam", gf=1e-12)
    env = {}
    for item in f[4:]:
        k, v = item.split("=", 1)
        if k.startswith("env."):
            env[k[4:]] = v
        else:
            kw[k] = type(kw[k])(v) if kw[k] is not None and not isinstance(kw[k], bool) else (float(v) if k == "scale" else v not in ("0", "False"))
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    t0 = time.time()
    try:
        r = PU.compare_step(cell, layers, kw["loss"], N=kw["N"], B=B, T=T, S=kw["S"], seed=kw["seed"], scale=kw["scale"], full=kw["full"],
                            zipf=kw["zipf"], steps=kw["steps"], updater=kw["updater"], grad_floor=kw["gf"])
        bad = {k: float("%.3g" % v) for k, v in r.items() if k.startswith("grad:") and v > 1e-5}
        print(spec, {k: float("%.3g" % v) for k, v in r.items() if k in KEEP}, "grads > 1e-5:", bad, "%.1fs" % (time.time() - t0), flush=True)
    except Exception as ex:
        print(spec, "ERROR", repr(ex)[:300], flush=True)
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
