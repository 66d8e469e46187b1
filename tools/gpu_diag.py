"""Runs every HIP-vs-oracle parity case and keeps going after failures; writes
gpurun_out/diag.json.  Usage on the GPU box: python tools/gpu_diag.py [quick]"""
import json
import os
import sys
import time
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402  (first: one HIP runtime for torch and libsbr_rnn.so)
import parity_util as PU  # noqa: E402

CASES = []
for flags, tag in ((3, "simple"), (2, "mfma-rec+naive-gemm"), (1, "simple-rec+mfma-gemm"), (0, "fast")):
    for cell in ("GRU", "LSTM", "Vanilla"):
        CASES.append(dict(tag=tag, flags=flags, cell=cell, layers=[4], loss="CCE", N=23, B=5, T=7))
for cell in ("GRU", "LSTM", "Vanilla"):
    for H in (20, 50, 128, 160):
        CASES.append(dict(tag="fast", flags=0, cell=cell, layers=[H], loss="CCE", N=61, B=37, T=9))
for loss in ("Blackout", "BPR", "TOP1"):
    CASES.append(dict(tag="fast", flags=0, cell="GRU", layers=[16], loss=loss, N=40, B=6, T=5, S=7))
    CASES.append(dict(tag="simple", flags=3, cell="GRU", layers=[16], loss=loss, N=40, B=6, T=5, S=7))
CASES.append(dict(tag="fast", flags=0, cell="LSTM", layers=[20, 12], loss="CCE", N=30, B=6, T=6))
CASES.append(dict(tag="simple", flags=3, cell="LSTM", layers=[20, 12], loss="CCE", N=30, B=6, T=6))
CASES.append(dict(tag="fast", flags=0, cell="GRU", layers=[20, 12], loss="BPR", N=30, B=6, T=6, S=5))
CASES.append(dict(tag="fast", flags=0, cell="LSTM", layers=[8], loss="CCE", N=19, B=4, T=5, F=2, n_opt=10))
for upd in ("adagrad", "adadelta", "rmsprop", "nesterov"):
    CASES.append(dict(tag="fast", flags=0, cell="GRU", layers=[8], loss="CCE", N=19, B=4, T=5, updater=upd))
CASES.append(dict(tag="fast", flags=0, cell="GRU", layers=[8], loss="CCE", N=19, B=4, T=5, reg=0.05))
CASES.append(dict(tag="fast", flags=0, cell="GRU", layers=[8], loss="CCE", N=19, B=4, T=5, reg=-0.05))
CASES.append(dict(tag="fast-clip", flags=0, cell="GRU", layers=[8], loss="CCE", N=19, B=4, T=5, popscale=1e-4))
CASES.append(dict(tag="fast-clip", flags=0, cell="LSTM", layers=[8], loss="CCE", N=19, B=4, T=5, popscale=1e-4))
# bench-shaped (config 2 of BASELINE.json) single step, smaller T to keep the oracle fast
CASES.append(dict(tag="fast", flags=0, cell="GRU", layers=[128], loss="CCE", N=3706, B=256, T=20, steps=1))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    os.makedirs("gpurun_out", exist_ok=True)
    print("device:", torch.cuda.get_device_name(0), flush=True)
    results = []
    for c in CASES[:12] if quick else CASES:
        kw = dict(c)
        tag = kw.pop("tag")
        t0 = time.time()
        try:
            r = PU.compare_step(**kw)
            summary = {k: v for k, v in r.items() if not k.startswith("grad:")}
            bad = [k for k, v in r.items() if k.startswith("grad:") and v > 1e-3]
            status = "ok" if (r["grad_worst"] < 1e-3 and r["h_last"] < 1e-4 and r["cost"] < 1e-4 and
                              r["topk_mismatch"] == 0 and list(r.values())[-3] < 1e-2) else "MISMATCH"
            results.append(dict(case=c, status=status, summary=summary, bad_grads={k: r[k] for k in bad}))
            print("%-9s %-22s %s  %.1fs" % (status, tag, json.dumps(c), time.time() - t0))
            print("          ", json.dumps({k: float("%.3g" % v) for k, v in summary.items()}), flush=True)
            if bad:
                print("           bad grads:", {k: float("%.3g" % r[k]) for k in bad}, flush=True)
        except Exception as e:  # keep going: one broken kernel must not hide the others
            results.append(dict(case=c, status="ERROR", error=repr(e), tb=traceback.format_exc()))
            print("ERROR     %-22s %s: %r" % (tag, json.dumps(c), e), flush=True)
        with open("gpurun_out/diag.json", "w") as f:
            json.dump(results, f, indent=1)
    n_ok = sum(r["status"] == "ok" for r in results)
    print("SUMMARY: %d/%d ok" % (n_ok, len(results)))


if __name__ == "__main__":
    main()
