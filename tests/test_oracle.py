"""Pins the CPU oracle (oracle/rnn_oracle.py).  The reference has no tests or golden
vectors for this path, so the oracle is pinned by
(0) the reference's own layer / cost source executed through a Theano stand-in (tests/test_reference_layers.py),
(1) an independent torch-autograd restatement, (2) central finite differences,
(3) the committed golden fixtures (tests/test_golden.py).  (1)-(3) are this file and test_golden.py."""
import numpy as np
import pytest

from oracle import rnn_oracle as O
from oracle import torch_ref as R


def make_batch(rng, B, T, N, S, F=1, n_in0=None):
    n_in0 = n_in0 or N
    lens = rng.integers(1, T + 1, size=B)
    lens[0] = T          # a full-length row
    lens[1] = 1          # a length-1 row
    X = np.zeros((B, T, F), dtype=np.int32)
    mask = np.zeros((B, T))
    for b in range(B):
        X[b, :lens[b], 0] = rng.integers(0, N, size=lens[b])
        mask[b, :lens[b]] = 1
        if F > 1:
            X[b, :lens[b], 1] = rng.integers(N, n_in0, size=lens[b])
    X[2, :lens[2], 0] = 0    # pad id 0 is also a real item; duplicates must accumulate
    return dict(X=X, mask=mask, target=rng.integers(0, N, size=B).astype(np.int32),
                samples=rng.integers(0, N, size=S).astype(np.int32),
                pop=rng.uniform(0.5, 2.0, size=B))


def setup(cell, layers, loss, seed=0, popscale=1.0, F=1, n_opt=0):
    rng = np.random.default_rng(seed)
    B, T, N, S = 3, 5, 7, 4
    params = O.init_params(cell, layers, N, rng, n_in0=N + n_opt)
    for p in params:
        p += rng.normal(0, 0.3, size=p.shape)
    batch = make_batch(rng, B, T, N, S, F=F, n_in0=N + n_opt)
    batch["pop"] *= popscale
    cfg = dict(cell=cell, layers=layers, loss=loss, regularization=0.01)
    return params, cfg, batch


@pytest.mark.parametrize("cell", ["LSTM", "GRU", "Vanilla"])
@pytest.mark.parametrize("layers", [[4], [5, 3]])
@pytest.mark.parametrize("loss", ["CCE", "Blackout", "BPR", "TOP1"])
@pytest.mark.parametrize("popscale", [1.0, 1e-4])   # 1e-4 drives gate grads past +-100: clip active
def test_numpy_bptt_matches_torch_autograd(cell, layers, loss, popscale):
    params, cfg, batch = setup(cell, layers, loss, popscale=popscale)
    c1, g1, aux = O.cost_and_grads(params, cfg, batch)
    c2, g2, h2, a2 = R.cost_and_grads(params, cfg, batch, O.recurrent_param_shapes)
    assert abs(c1 - c2) <= 1e-10 * abs(c2)
    assert np.allclose(aux["h"], h2, rtol=0, atol=1e-12)
    for a, b in zip(g1, g2):
        assert a.shape == b.shape
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())


def test_clip_is_actually_active_in_the_scaled_case(monkeypatch):
    params, cfg, batch = setup("GRU", [4], "CCE", popscale=1e-4)
    _, g_clip, _ = O.cost_and_grads(params, cfg, batch)
    monkeypatch.setattr(O, "GRAD_CLIP", 1e30)
    _, g_free, _ = O.cost_and_grads(params, cfg, batch)
    assert max(np.abs(a - b).max() for a, b in zip(g_clip, g_free)) > 1.0


@pytest.mark.parametrize("cell", ["LSTM", "GRU", "Vanilla"])
@pytest.mark.parametrize("loss", ["CCE", "Blackout", "BPR", "TOP1"])
def test_finite_differences(cell, loss):
    params, cfg, batch = setup(cell, [4], loss, seed=3)
    _, grads, _ = O.cost_and_grads(params, cfg, batch)
    rng = np.random.default_rng(7)
    eps = 1e-6
    for pi, p in enumerate(params):
        for _ in range(4):
            idx = tuple(rng.integers(0, s) for s in p.shape)
            old = p[idx]
            p[idx] = old + eps
            cp, _, _ = O.cost_and_grads(params, cfg, batch)
            p[idx] = old - eps
            cm, _, _ = O.cost_and_grads(params, cfg, batch)
            p[idx] = old
            fd = (cp - cm) / (2 * eps)
            assert abs(fd - grads[pi][idx]) <= 1e-6 * max(1.0, abs(fd)), (pi, idx, fd, grads[pi][idx])


def test_multi_index_input_rating_feature():
    # --rf: F=2 (item id, N + rating bucket), input_size = N + 10 (rnn_base.py:615-642)
    params, cfg, batch = setup("LSTM", [4], "CCE", F=2, n_opt=10)
    c1, g1, _ = O.cost_and_grads(params, cfg, batch)
    c2, g2, _, _ = R.cost_and_grads(params, cfg, batch, O.recurrent_param_shapes)
    assert abs(c1 - c2) <= 1e-10 * abs(c2)
    for a, b in zip(g1, g2):
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())


@pytest.mark.parametrize("cell", ["LSTM", "GRU", "Vanilla"])
@pytest.mark.parametrize("F,n_opt", [(1, 0), (2, 10)])
def test_embedding_layer_option(cell, F, n_opt):
    # --r_emb E (recurrent_layers.py:46-50): EmbeddingLayer + flatten, dense layer 0; NumPy BPTT vs torch autograd vs FD
    rng = np.random.default_rng(5)
    B, T, N, S, E = 3, 5, 7, 4, 3
    layers = [4, 3]
    params = O.init_params(cell, layers, N, rng, n_in0=N + n_opt, embedding=E, n_feat=F)
    # (a dense Vanilla layer is the stock RecurrentLayer: hid_init comes first, then W_in)
    assert params[0].shape == (N + n_opt, E) and params[2 if cell == "Vanilla" else 1].shape[0] == F * E
    for p in params:
        p += rng.normal(0, 0.3, size=p.shape)
    batch = make_batch(rng, B, T, N, S, F=F, n_in0=N + n_opt)
    cfg = dict(cell=cell, layers=layers, loss="CCE", regularization=0.0, embedding=E)
    c1, g1, aux = O.cost_and_grads(params, cfg, batch)
    c2, g2, h2, _ = R.cost_and_grads(params, cfg, batch, O.recurrent_param_shapes)
    assert abs(c1 - c2) <= 1e-10 * abs(c2) and len(g1) == len(params)
    for a, b in zip(g1, g2):
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    eps = 1e-6
    for idx in [(0, 1), (int(batch["X"][0, 0, 0]), 2)]:          # rows hit by the batch (and maybe one that is not)
        old = params[0][idx]
        params[0][idx] = old + eps; cp, _, _ = O.cost_and_grads(params, cfg, batch)
        params[0][idx] = old - eps; cm, _, _ = O.cost_and_grads(params, cfg, batch)
        params[0][idx] = old
        assert abs((cp - cm) / (2 * eps) - g1[0][idx]) <= 1e-6 * max(1.0, abs(g1[0][idx]))


@pytest.mark.parametrize("cell", ["LSTM", "GRU", "Vanilla"])
@pytest.mark.parametrize("layers,emb", [([4], 0), ([4, 3], 0), ([3, 2], 3)])
def test_bidirectional_option(cell, layers, emb):
    # --r_bi (recurrent_layers.py:70-76): forward + backwards layer per level, concatenated; NumPy BPTT vs torch autograd vs FD
    rng = np.random.default_rng(11)
    B, T, N, S = 3, 5, 7, 4
    params = O.init_params(cell, layers, N, rng, embedding=emb, bidirectional=True)
    per = {"LSTM": 17, "GRU": 10, "Vanilla": 4}[cell]
    assert len(params) == (1 if emb else 0) + 2 * per * len(layers) + 2 and params[-2].shape == (2 * layers[-1], N)
    for p in params:
        p += rng.normal(0, 0.3, size=p.shape)
    batch = make_batch(rng, B, T, N, S)
    cfg = dict(cell=cell, layers=layers, loss="CCE", regularization=0.0, embedding=emb, bidirectional=True)
    c1, g1, aux = O.cost_and_grads(params, cfg, batch)
    c2, g2, h2, _ = R.cost_and_grads(params, cfg, batch, O.recurrent_param_shapes)
    assert abs(c1 - c2) <= 1e-10 * abs(c2) and np.allclose(aux["h"], h2, rtol=0, atol=1e-12)
    for a, b in zip(g1, g2):
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max())
    eps = 1e-6
    for pi in (0, len(params) // 2, len(params) - 3):
        p = params[pi]
        idx = tuple(rng.integers(0, s) for s in p.shape)
        old = p[idx]
        p[idx] = old + eps; cp, _, _ = O.cost_and_grads(params, cfg, batch)
        p[idx] = old - eps; cm, _, _ = O.cost_and_grads(params, cfg, batch)
        p[idx] = old
        assert abs((cp - cm) / (2 * eps) - g1[pi][idx]) <= 1e-6 * max(1.0, abs(g1[pi][idx]))


def test_param_order_and_shapes_match_lasagne_layout():
    # LSTM-OHE 1 layer = 19 arrays, GRU-OHE = 12 arrays
    assert len(O.model_param_shapes("LSTM", [20], 3706)) == 19
    shp = O.model_param_shapes("GRU", [128], 3706)
    assert len(shp) == 12
    assert shp[0] == ("l0.W_in_to_updategate", (3706, 128))
    assert shp[3][0] == "l0.W_in_to_resetgate" and shp[6][0] == "l0.W_in_to_hidden_update"
    assert shp[-2] == ("out.W", (128, 3706)) and shp[-1] == ("out.b", (3706,))
    l = O.model_param_shapes("LSTM", [8, 4], 11)
    assert l[0][1] == (11, 8) and l[17][1] == (8, 4)   # layer-1 W_in is (H0, H1)


@pytest.mark.parametrize("name", ["adagrad", "adam", "rmsprop", "adadelta", "nesterov"])
def test_updaters_against_scalar_formulas(name):
    # lasagne.updates.* restated one scalar at a time [3P]
    rng = np.random.default_rng(1)
    p = [rng.normal(size=(3, 2))]
    ref = p[0].copy()
    upd = O.Updater(name, 0.05, rho=0.8, beta1=0.7, beta2=0.9)
    s0 = np.zeros_like(ref); s1 = np.zeros_like(ref)
    for t in range(1, 4):
        g = rng.normal(size=(3, 2))
        upd.apply(p, [g])
        if name == "adagrad":
            s0 += g * g; ref -= 0.05 * g / np.sqrt(s0 + 1e-6)
        elif name == "rmsprop":
            s0 = 0.8 * s0 + 0.2 * g * g; ref -= 0.05 * g / np.sqrt(s0 + 1e-6)
        elif name == "adadelta":
            s0 = 0.8 * s0 + 0.2 * g * g
            u = g * np.sqrt(s1 + 1e-6) / np.sqrt(s0 + 1e-6)
            ref -= 0.05 * u; s1 = 0.8 * s1 + 0.2 * u * u
        elif name == "nesterov":
            s0 = 0.8 * s0 - 0.05 * g; ref += 0.8 * s0 - 0.05 * g
        else:
            a = 0.05 * np.sqrt(1 - 0.9 ** t) / (1 - 0.7 ** t)
            s0 = 0.7 * s0 + 0.3 * g; s1 = 0.9 * s1 + 0.1 * g * g
            ref -= a * s0 / (np.sqrt(s1) + 1e-8)
        assert np.allclose(p[0], ref, rtol=1e-13, atol=0)


def test_topk_semantics_ordered_descending():
    rng = np.random.default_rng(0)
    s = rng.permutation(50).astype(float)
    ids = O.topk_ordered(s, 10)
    assert list(ids) == list(np.argsort(-s)[:10])


def test_test_function_excludes_seen_items():
    params, cfg, batch = setup("GRU", [4], "CCE")
    excl = [[int(i) for i in batch["X"][b, :int(batch["mask"][b].sum()), 0]] for b in range(3)]
    ids = O.test_function(params, cfg, batch["X"], batch["mask"], excl, k=3)
    for b in range(3):
        assert not set(ids[b]) & set(excl[b]) or len(set(range(7)) - set(excl[b])) < 3


@pytest.mark.parametrize("loss", ["hinge", "logit", "logsig"])
def test_margin_head_gradients_against_finite_differences(loss):
    """RNNMargin's losses (rnn_margin.py:62-69): analytic gradients of the whole model against central differences, with
    rows whose positives repeat / also occur in the input and a popularity-based default target (third party: none --
    pinned by the reference's own code in tests/test_reference_layers.py; this is the independent check)."""
    import parity_util as PU
    rng = np.random.default_rng(3)
    N, B, T = 23, 5, 6
    params, cfg, batch = PU.build_case("GRU", [7], loss, N, B, T, S=3, seed=11)
    dflt = O.margin_default_target(rng.integers(1, 40, size=N), 50, 0.1)
    ob = PU.margin_oracle_batch(batch, N, balance=1.5, unique=True, default_target=dflt)
    cost, grads, _ = O.cost_and_grads(params, cfg, ob)
    eps = 1e-6
    for pi in (0, len(params) - 2, len(params) - 1):
        p = params[pi]
        for idx in [tuple(rng.integers(0, s) for s in p.shape) for _ in range(4)]:
            old = p[idx]
            p[idx] = old + eps; cp = O.cost_and_grads(params, cfg, ob)[0]
            p[idx] = old - eps; cm = O.cost_and_grads(params, cfg, ob)[0]
            p[idx] = old
            num = (cp - cm) / (2 * eps)
            if loss == "hinge" and abs(num - grads[pi][idx]) > 1e-6:      # a kink of the hinge inside the step: skip
                continue
            assert abs(num - grads[pi][idx]) <= 1e-6 * max(1.0, abs(num)), (pi, idx, num, grads[pi][idx])


@pytest.mark.parametrize("cell,H", [("LSTM", 20), ("GRU", 32)])
def test_scripted_cpu_port_equals_the_eager_port(cell, H):
    """bench.py's cpu_baseline times the torch port with its T-step scan under torch.jit.script (oracle/torch_ref.py,
    layer_forward_scripted: unbind / chunk instead of per-step selects).  Same arithmetic: costs of three Adam steps on ragged rows
    equal the eager port's to float32 rounding (no gradient is near the clip the scripted form leaves out)."""
    from oracle import torch_ref as R
    rng = np.random.default_rng(5)
    N, B, T = 50, 16, 12
    params = O.init_params(cell, [H], N, rng, dtype=np.float32)
    mask = np.ones((B, T), np.float32)
    mask[3, 5:] = 0
    mask[7, 1:] = 0
    batch = dict(X=rng.integers(0, N, size=(B, T, 1)).astype(np.int32), mask=mask, target=rng.integers(0, N, size=B).astype(np.int32),
                 pop=np.ones(B, np.float32))
    cfg = dict(cell=cell, layers=[H], loss="CCE", regularization=0.0)
    a = R.TorchTrainer(params, cfg, O.recurrent_param_shapes)
    b = R.TorchTrainer(params, cfg, O.recurrent_param_shapes, scripted=True)
    for _ in range(3):
        ca, cb = a.train_function(batch), b.train_function(batch)
        assert abs(ca - cb) <= 1e-6 * abs(ca), (ca, cb)
