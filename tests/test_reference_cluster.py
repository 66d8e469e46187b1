"""RNNCluster (rnn_cluster.py, `train.py -m RNN --clusters C`) against the reference's OWN code: tests/golden/reference_layers/cl_*.npz
hold what /root/reference/neural_networks/rnn_cluster.py's `_prepare_networks` / `_get_hard_clusters` / test-function lines
computed -- through the eager stand-in for the Theano / Lasagne calls they make (tools/theano_on_torch.py,
tools/make_reference_layer_golden.py) -- for seeded parameters and batches: both costs, d cost / d every network parameter,
d cost_clusters / d (selection weights, repartition), the selection activations, the hard clusters, both test scores.
CPU: the oracle's restatement agrees to float64 round-off.  GPU: the HIP engine, through the C-ABI, within the parity bar."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_layers", "cl_*.npz")))
IDS = [os.path.basename(p)[:-4] for p in GOLD]


def load(path):
    z = np.load(path, allow_pickle=False)
    n = int(z["n_params"])
    cfg = dict(cell=str(z["cell"]), layers=[int(h) for h in z["layers"]], loss=str(z["loss"]), regularization=0.0, embedding=0,
               bidirectional=False, clusters=dict(n=int(z["n_clusters"]), type=str(z["cluster_type"]), scale=float(z["scale"]),
                                                  c_sampling=int(z["c_sampling"])))
    batch = dict(X=z["X"], mask=z["mask"], target=z["target"], samples=z["samples"],
                 cluster_samples=z["cluster_samples"] if int(z["c_sampling"]) else None, pop=np.ones(len(z["X"])))
    return z, cfg, batch, [z["p%d" % i] for i in range(n)], [z["g%d" % i] for i in range(n)]


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-300))


def test_fixtures_cover_every_cluster_type_and_loss():
    seen = [(str(np.load(p)["cluster_type"]), str(np.load(p)["loss"]), int(np.load(p)["c_sampling"]) > 0) for p in GOLD]
    assert {t for t, _, _ in seen} == {"mix", "softmax", "sigmoid"}
    assert {l for _, l, _ in seen} == {"CCE", "Blackout", "BPR", "TOP1", "BPRelu", "lin"}          # rnn_cluster.py:88-101
    assert {c for _, _, c in seen} == {True, False}                                               # --c_sampling set / unset


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_oracle_agrees_with_the_reference_cluster_code(path):
    from oracle import rnn_oracle as O
    z, cfg, batch, p0, g = load(path)
    N = int(z["N"])
    names = [n for n, _ in O.model_param_shapes(cfg["cell"], cfg["layers"], N, N, 0, 1, False)] + ["cluster_repartition", "cluster_selection.W"]
    ref_names = [str(n) for n in z["names"]]
    assert len(names) == len(ref_names) == len(p0)
    for mine, ref in zip(names, ref_names):
        assert mine.split(".")[-1] == ref or mine.endswith(ref) or mine == ref, (mine, ref)
    params = [p.astype(np.float64) for p in p0]
    cost, grads, aux = O.cost_and_grads(params, cfg, batch)
    assert abs(cost - float(z["cost"])) <= 1e-12 * abs(float(z["cost"]))
    assert abs(aux["cost_clusters"] - float(z["cost_clusters"])) <= 1e-12 * abs(float(z["cost_clusters"]))
    assert rel(aux["h"], z["h_last"]) <= 1e-12
    for n, a, b in zip(ref_names, grads, g):
        assert a.shape == b.shape, n
        assert np.abs(a - b).max() <= 1e-11 * max(np.abs(b).max(), 1e-3), n
    assert rel(aux["h"] @ params[-1], z["selection"]) <= 1e-12
    assert rel(O.cluster_hard(params[-2], cfg["clusters"]["type"]), z["hard"]) <= 1e-12
    excl = [[int(i) for i in batch["X"][b, :int(batch["mask"][b].sum()), 0]] if int(z["unique"]) else [] for b in range(len(batch["X"]))]
    k = 5
    ids1, ids2, csel, n_used, (s1, s2) = O.cluster_test_rows(params, cfg, batch["X"], batch["mask"], excl, k=k)
    assert np.array_equal(csel, np.argmax(z["selection"], axis=1))
    assert np.allclose(n_used, z["hard"][:, csel].sum(axis=0), rtol=1e-12)
    want1 = np.argsort(-z["test_scores"], axis=1, kind="stable")[:, :k]
    assert np.array_equal(ids1, want1)
    # inside the cluster: where the selected cluster holds at least k unseen items with distinct scores the ids are well defined
    t2 = -np.sort(-z["test_scores_clusters"], axis=1)[:, :k + 1]
    rows = np.all(np.diff(t2, axis=1) < 0, axis=1)
    assert rows.any()
    assert np.array_equal(ids2[rows], np.argsort(-z["test_scores_clusters"], axis=1, kind="stable")[rows, :k])


ENGINE_LOSS = {"CCE": "SCCE", "Blackout": "Blackout", "BPR": "BPR", "TOP1": "TOP1", "BPRelu": "BPRelu", "lin": "lin"}


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_engine_and_cluster_head_agree_with_the_reference_cluster_code(path):
    import torch
    from sbr_amd.engine import RNNEngine, ClusterHead
    z, cfg, batch, p0, g = load(path)
    N, B, T, S = (int(z[k]) for k in ("N", "B", "T", "S"))
    cl = cfg["clusters"]
    H = cfg["layers"][-1]
    csm = batch["cluster_samples"] if cl["c_sampling"] else batch["samples"]
    eng = RNNEngine(cell=cfg["cell"], layers=cfg["layers"], n_items=N, max_length=T, batch_size=B, loss=ENGINE_LOSS[cfg["loss"]],
                    n_samples=S, updater="adam", learning_rate=0.01, unique=bool(int(z["unique"])))
    head = ClusterHead(eng, cl["n"], cl["type"], loss=ENGINE_LOSS[cfg["loss"]], max_samples=max(S, len(csm)), updater="adam",
                       learning_rate=0.01, scale=cl["scale"])
    try:
        eng.set_all_param_values(p0[:-2])
        head.set_params(p0[-2], p0[-1])
        eng.set_batch(batch["X"], batch["mask"], batch["target"], batch["samples"], np.ones(B, dtype=np.float32))
        cost = eng.forward_backward()
        assert abs(cost - float(z["cost"])) <= 1e-5 * abs(float(z["cost"]))
        Bp = (B + 15) // 16 * 16
        assert rel(eng.debug_buffer("h_last").reshape(Bp, -1)[:B, :H], z["h_last"]) <= 1e-3
        for n, a, b in zip(z["names"], eng.get_all_grad_values(), g[:-2]):
            assert np.abs(a - b).max() <= 1e-4 * max(np.abs(b).max(), 1e-3), str(n)
        # the cluster head on the same user representations
        ccost = head.forward_backward(batch["target"], csm)
        assert abs(ccost - float(z["cost_clusters"])) <= 1e-5 * abs(float(z["cost_clusters"]))
        dR, dWc = head.get_grads()
        assert np.abs(dR - g[-2]).max() <= 1e-4 * max(np.abs(g[-2]).max(), 1e-3)
        assert np.abs(dWc - g[-1]).max() <= 1e-4 * max(np.abs(g[-1]).max(), 1e-3)
        untouched = np.setdiff1d(np.arange(N), np.concatenate([batch["target"], csm]))
        assert not dR[untouched].any()                                   # rows no target / cluster sample names: exactly zero
        # test path: selection activations, selected cluster, hard clusters, scores inside the selected cluster
        csel, act = head.select(B, with_activations=True)
        assert rel(act, z["selection"]) <= 1e-3
        zs = np.sort(z["selection"], axis=1)
        clear = zs[:, -1] - zs[:, -2] > 1e-4                              # rows whose best two clusters are not a float32 tie
        assert clear.sum() >= B - 1 and np.array_equal(csel[clear], np.argmax(z["selection"], axis=1)[clear])
        hard = head.hard_clusters()
        assert np.abs(hard - z["hard"]).max() <= 2e-5
        probs = eng.test_probabilities(batch["X"], batch["mask"])
        sdev = torch.from_numpy(probs).to(eng.device)
        used = head.mask_scores(sdev, np.argmax(z["selection"], axis=1))
        assert np.allclose(used, z["hard"][:, np.argmax(z["selection"], axis=1)].sum(axis=0), rtol=1e-5)
        s2 = sdev.cpu().numpy()
        if int(z["unique"]):
            for b in range(B):
                s2[b, batch["X"][b, :int(batch["mask"][b].sum()), 0]] = 0.0
        assert np.abs(s2 - z["test_scores_clusters"]).max() <= 1e-3 * np.abs(z["test_scores_clusters"]).max() + 1e-7
    finally:
        head.close()
        eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("updater", ["adam", "adagrad", "nesterov"])
@pytest.mark.parametrize("ctype,loss,cs", [("mix", "CCE", 0), ("softmax", "Blackout", 9), ("sigmoid", "BPR", 0), ("mix", "TOP1", 5)])
def test_cluster_training_steps_follow_the_oracle(ctype, loss, cs, updater):
    # three steps of both models -- the recurrent network on `cost`, (Wc, R) on `cost_clusters`, each with its own updater state
    # (rnn_cluster.py:277-285) -- against the oracle's dense updates; duplicate cluster samples and a sample that is also a target
    import parity_util as PU
    from oracle import rnn_oracle as O
    from sbr_amd.engine import RNNEngine, ClusterHead
    N, B, T, S, C, H = 60, 9, 8, 6, 5, 20
    params, cfg, batch = PU.build_case("GRU", [H], loss, N, B, T, S=S, seed=77, clusters=dict(n=C, type=ctype, scale=1.3, c_sampling=cs))
    csm = batch["cluster_samples"] if cs else batch["samples"]
    csm[1] = csm[0]; csm[2] = batch["target"][0]
    eng = RNNEngine(cell="GRU", layers=[H], n_items=N, max_length=T, batch_size=B, loss=ENGINE_LOSS[loss], n_samples=S,
                    updater=updater, learning_rate=0.01, flags=64)
    head = ClusterHead(eng, C, ctype, loss=ENGINE_LOSS[loss], max_samples=max(S, len(csm)), updater=updater, learning_rate=0.01, scale=1.3)
    try:
        eng.set_all_param_values(params[:-2]); head.set_params(params[-2], params[-1])
        upd = O.Updater(updater, 0.01, rho=0.9, beta1=0.9, beta2=0.999)
        op = [p.copy() for p in params]
        for _ in range(3):
            ocost = O.train_function(op, cfg, upd, batch)
            eng.set_batch(batch["X"], batch["mask"], batch["target"], batch["samples"], batch["pop"])
            cost = eng.train_step(sync=True)
            head.forward_backward(batch["target"], csm, read_cost=False)
            head.apply_update()
            assert abs(cost - ocost) <= 1e-5 * abs(ocost)
        R, Wc = head.get_params()
        for a, b in zip(eng.get_all_param_values() + [R, Wc], op):
            assert PU.rel_err(a, b) <= 1e-3
        assert PU.rel_err(R, op[-2]) <= 2e-4 and PU.rel_err(Wc, op[-1]) <= 2e-4
    finally:
        head.close()
        eng.close()
