"""The reference's OWN layer and cost code as the known answer: tests/golden/reference_layers/*.npz hold the cost, the
gradient of every parameter, the recurrent output and the deterministic scores computed by
/root/reference/neural_networks/{sparse_lstm,rnn_one_hot,rnn_sampling,recurrent_layers}.py themselves, executed through
an eager stand-in for the Theano / Lasagne calls they make (tools/theano_on_torch.py, tools/make_reference_layer_golden.py).
CPU: the oracle agrees with them to float64 round-off (so the oracle is pinned by the reference's code on this part of
the path).  GPU: the HIP engine, through the C-ABI, agrees within the parity bar."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(p for p in glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_layers", "*.npz"))
              if not os.path.basename(p).startswith("cl_"))          # (cl_*: RNNCluster, tests/test_reference_cluster.py)
IDS = [os.path.basename(p)[:-4] for p in GOLD]
TOL_LOGITS = 1e-3      # north_star: within 1e-3 relative on logits
TOL_GRADS = 1e-4
MARGIN = ("hinge", "logit", "logsig")


def load(path):
    z = np.load(path, allow_pickle=False)
    n = int(z["n_params"])
    cfg = dict(cell=str(z["cell"]), layers=[int(h) for h in z["layers"]], loss=str(z["loss"]),
               regularization=float(z["regularization"]), embedding=0, bidirectional=bool(int(z["bidirectional"])))
    batch = dict(X=z["X"], mask=z["mask"], target=z["target"], samples=z["samples"], pop=z["pop"])
    if cfg["loss"] in MARGIN:      # RNNMargin: Y / weight as the reference's own _prepare_input packed them, and the positives they came from
        batch.update(Y=z["Y"], weight=z["weight"], targets=z["targets"])
    return z, cfg, batch, [z["p%d" % i] for i in range(n)], [z["g%d" % i] for i in range(n)]


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-300))


def test_fixtures_cover_every_cell_and_head():
    seen = set()
    for path in GOLD:
        z = np.load(path)
        seen.add((str(z["cell"]), str(z["loss"])))
    assert {c for c, _ in seen} == {"GRU", "LSTM", "Vanilla"}
    assert {l for _, l in seen} == {"CCE", "Blackout", "BPR", "TOP1", "hinge", "logit", "logsig"}
    assert len(GOLD) >= 24
    assert sum(float(np.load(p)["clip_changes"]) > 0.5 for p in GOLD) >= 4        # cases where the gradient clip decides the result


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_oracle_agrees_with_the_reference_code(path):
    from oracle import rnn_oracle as O
    z, cfg, batch, p0, g = load(path)
    N, F, n_opt = int(z["N"]), int(z["F"]), int(z["n_opt"])
    names = [n for n, _ in O.model_param_shapes(cfg["cell"], cfg["layers"], N, N + n_opt, 0, F, cfg["bidirectional"])]
    ref_names = [str(n) for n in z["names"]]
    assert len(names) == len(ref_names)
    for mine, ref in zip(names, ref_names):        # same parameter, same place in the checkpoint list
        assert mine.split(".")[-1] == ref or mine.endswith(ref), (mine, ref)
    params = [p.astype(np.float64) for p in p0]
    ob = dict(batch); ob["pop"] = batch["pop"].astype(np.float64)
    if cfg["loss"] in MARGIN:
        # the oracle's restatement of RNNMargin._prepare_input reproduces the reference's own dense target / weight matrices
        dflt = O.margin_default_target(z["item_popularity"], int(z["n_users"]), float(z["min_access"])) if int(z["popularity_based"]) else None
        tg = [[int(t) for t in row if t >= 0] for row in z["targets"]]
        Y, W = O.margin_targets(batch["X"], batch["mask"], tg, N, balance=float(z["balance"]), unique=bool(int(z["unique"])),
                                default_target=dflt)
        assert np.array_equal(Y, z["Y"]) and np.allclose(W, z["weight"], rtol=1e-6, atol=0)      # (the reference packs float32)
    cost, grads, aux = O.cost_and_grads(params, cfg, ob)
    assert abs(cost - float(z["cost"])) <= 1e-12 * abs(float(z["cost"]))
    assert rel(aux["h"], z["h_last"]) <= 1e-12
    for n, a, b in zip(ref_names, grads, g):
        assert a.shape == b.shape, n
        assert np.abs(a - b).max() <= 1e-11 * max(np.abs(b).max(), 1e-3), n
    scores, logits = O.predict_scores(params, cfg, batch["X"], batch["mask"])
    assert rel(scores, z["scores"]) <= 1e-12                   # predict_function: probabilities (CCE) / raw scores (sampled heads, RNNMargin)
    # test function: softmax, viewed items zeroed (rnn_base.py:196-209, rnn_sampling.py:140-156) -> ordered top-k ids
    excl = [[int(i) for i in batch["X"][b, :int(batch["mask"][b].sum()), 0]] if int(z["unique"]) else []
            for b in range(len(batch["X"]))]
    k = 5
    ids = O.test_function(params, cfg, batch["X"], batch["mask"], excl, k=k)
    want = np.argsort(-z["test_scores"], axis=1, kind="stable")[:, :k]
    top = -np.sort(-z["test_scores"], axis=1)[:, :k + 1]
    assert np.all(np.diff(top, axis=1) < 0)                    # no ties among the ranked items: the ids are well defined
    assert np.array_equal(ids, want)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_engine_agrees_with_the_reference_code(path):
    from sbr_amd.engine import RNNEngine
    z, cfg, batch, p0, g = load(path)
    N, B, T, S, F, n_opt = (int(z[k]) for k in ("N", "B", "T", "S", "F", "n_opt"))
    H = cfg["layers"][-1]
    margin = cfg["loss"] in MARGIN
    eng = RNNEngine(cell=cfg["cell"], layers=cfg["layers"], n_items=N, max_length=T, batch_size=B, loss=cfg["loss"],
                    n_samples=S, updater="adam", learning_rate=0.01, regularization=cfg["regularization"],
                    input_size=N + n_opt, n_feat=F, bidirectional=cfg["bidirectional"],
                    balance=float(z["balance"]) if margin else 1.0, n_targets=S if margin else 1, unique=bool(int(z["unique"])))
    try:
        eng.set_all_param_values(p0)
        if margin:
            from oracle import rnn_oracle as O
            if int(z["popularity_based"]):
                eng.set_default_target(O.margin_default_target(z["item_popularity"], int(z["n_users"]), float(z["min_access"])))
            eng.set_batch(batch["X"], batch["mask"], batch["targets"])
        else:
            eng.set_batch(batch["X"], batch["mask"], batch["target"], batch["samples"] if cfg["loss"] != "CCE" else None, batch["pop"])
        cost = eng.forward_backward()
        assert abs(cost - float(z["cost"])) <= 1e-5 * abs(float(z["cost"]))
        Bp = (B + 15) // 16 * 16
        hl = eng.debug_buffer("h_last").reshape(Bp, -1)[:B]
        if cfg["bidirectional"]:
            half = hl.shape[1] // 2
            hl = np.concatenate([hl[:, :H], hl[:, half:half + H]], axis=1)
        else:
            hl = hl[:, :H]
        assert rel(hl, z["h_last"]) <= TOL_LOGITS
        for n, a, b in zip(z["names"], eng.get_all_grad_values(), g):
            assert np.abs(a - b).max() <= TOL_GRADS * max(np.abs(b).max(), 1e-3), str(n)
        assert rel(eng.predict_function(batch["X"], batch["mask"]), z["scores"]) <= TOL_LOGITS
        k = 5
        top = -np.sort(-z["test_scores"], axis=1)[:, :k + 1]
        # ids are bit-exact on every row whose ranked scores are well separated; a fixture without such a row would compare
        # nothing, so that is a failure of the fixture, not a pass
        sep = np.all(top[:, :-1] - top[:, 1:] > 1e-5 * np.abs(top[:, :1]), axis=1)
        if not sep.any():
            pytest.fail("no row of %s has its %d best scores separated: the top-k assertion would be vacuous" % (os.path.basename(path), k + 1))
        ids = eng.test_function((batch["X"], batch["mask"]), k=k, exclude_seen=(2 if margin else 1) if int(z["unique"]) else 0)
        assert np.array_equal(ids[sep], np.argsort(-z["test_scores"], axis=1, kind="stable")[sep, :k])
    finally:
        eng.close()
