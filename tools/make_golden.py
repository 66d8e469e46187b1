"""Generates tests/golden/*.npz: seeded inputs + float64 oracle outputs for the RNN hot path.

The reference cannot be run here (Python 2 + Theano/Lasagne, not installable: SURVEY.md 8c), so
these vectors come from oracle/rnn_oracle.py AFTER it was cross-checked against the independent
torch-autograd restatement and finite differences (tests/test_oracle.py).  They freeze the
oracle's behaviour (a later edit that changes results fails tests/test_golden.py) and give the
GPU tests known answers that do not depend on importing the oracle's code path at all.

    python tools/make_golden.py [names]   # default: the option fixtures; name the originals explicitly to rewrite them
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import rnn_oracle as O  # noqa: E402
import parity_util as PU  # noqa: E402

CASES = {
    # name: (cell, layers, loss, N, B, T, S, updater, F, n_opt)
    "gru16_cce_adam": ("GRU", [16], "CCE", 40, 7, 9, 0, "adam", 1, 0),
    "lstm20_cce_adagrad": ("LSTM", [20], "CCE", 45, 9, 8, 0, "adagrad", 1, 0),
    "vanilla8_cce_rmsprop": ("Vanilla", [8], "CCE", 30, 5, 6, 0, "rmsprop", 1, 0),
    "gru16_blackout_adam": ("GRU", [16], "Blackout", 40, 6, 7, 5, "adam", 1, 0),
    "lstm12_bpr_adagrad": ("LSTM", [12], "BPR", 40, 6, 7, 5, "adagrad", 1, 0),
    "gru12_top1_nesterov": ("GRU", [12], "TOP1", 40, 6, 7, 5, "nesterov", 1, 0),
    "lstm20x12_cce_adadelta": ("LSTM", [20, 12], "CCE", 35, 6, 6, 0, "adadelta", 1, 0),
    "lstm8_cce_rf_adam": ("LSTM", [8], "CCE", 25, 5, 6, 0, "adam", 2, 10),
}
# option fixtures added later: (..., embedding, bidirectional); the eight above are never regenerated
OPTION_CASES = {
    "gru12_cce_emb6_adam": ("GRU", [12], "CCE", 40, 7, 8, 0, "adam", 1, 0, 6, False),
    "lstm10x8_cce_bi_adagrad": ("LSTM", [10, 8], "CCE", 35, 7, 7, 0, "adagrad", 1, 0, 0, True),
    "gru8_bpr_bi_emb5_rf_adam": ("GRU", [8], "BPR", 30, 6, 6, 5, "adam", 2, 10, 5, True),
    # dense Vanilla layers = stock lasagne RecurrentLayer: rectify, parameters [hid_init, W_in, b, W_hid]
    "vanilla12x8_cce_adam": ("Vanilla", [12, 8], "CCE", 35, 7, 7, 0, "adam", 1, 0, 0, False),
    "vanilla10_blackout_emb6_adagrad": ("Vanilla", [10], "Blackout", 35, 7, 7, 5, "adagrad", 1, 0, 6, False),
}


def make(name):
    if name in CASES:
        cell, layers, loss, N, B, T, S, updater, F, n_opt = CASES[name]
        emb, bi = 0, False
    else:
        cell, layers, loss, N, B, T, S, updater, F, n_opt, emb, bi = OPTION_CASES[name]
    params, cfg, batch = PU.build_case(cell, layers, loss, N, B, T, S=S, seed=sum(map(ord, name)), F=F, n_opt=n_opt, emb=emb, bi=bi)
    ob = PU.oracle_batch(batch)
    cost, grads, aux = O.cost_and_grads(params, cfg, ob)
    upd = O.Updater(updater, 0.01, rho=0.9, beta1=0.9, beta2=0.999)
    p2 = [p.copy() for p in params]
    costs = [O.train_function(p2, cfg, upd, ob) for _ in range(3)]
    scores, logits = O.predict_scores(p2, cfg, batch["X"], batch["mask"])
    k = 5
    excl = [[int(i) for i in batch["X"][b, :int(batch["mask"][b].sum()), 0]] for b in range(B)]
    ids = O.test_function(p2, cfg, batch["X"], batch["mask"], excl, k=k)
    # tie-free guarantee for the bit-exact top-k check: gaps between ranked probabilities
    p = O.softmax_rows(logits)
    for b in range(B):
        row = p[b].copy(); row[excl[b]] = 0
        top = np.sort(row)[::-1][:k + 1]
        assert np.all(np.diff(top) < -1e-6 * top[0]), (name, b, top)
    out = dict(cell=cell, layers=np.array(layers), loss=loss, N=N, B=B, T=T, S=S, updater=updater, F=F, n_opt=n_opt,
               embedding=emb, bidirectional=int(bi),
               X=batch["X"], mask=batch["mask"], target=batch["target"], samples=batch["samples"], pop=batch["pop"],
               cost=cost, h_last=aux["h"], act=aux["act"], costs3=np.array(costs), scores=scores, topk=ids, n_params=len(params))
    for i, (p0, g, pn) in enumerate(zip(params, grads, p2)):
        out["p%d" % i] = p0.astype(np.float32); out["g%d" % i] = g; out["q%d" % i] = pn
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name + ".npz"), **out)
    print(name, "cost", cost, "bytes", os.path.getsize(os.path.join(ROOT, "tests", "golden", name + ".npz")))


if __name__ == "__main__":
    names = sys.argv[1:] or list(OPTION_CASES)       # default: only the option fixtures (the originals stay frozen)
    for n in names:
        make(n)
