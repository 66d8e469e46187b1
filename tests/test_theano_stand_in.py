"""tools/theano_on_torch.py is what lets the reference's own layer code run here (tests/golden/reference_layers/): these
tests hold the stand-in itself to the documented Theano / Lasagne behaviour of every call that code makes, on cases small
enough to check by hand or against numpy.  (The stand-in is a fixture-generation tool; nothing in the package uses it.)"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import theano_on_torch as E  # noqa: E402


@pytest.fixture(scope="module")
def mods():
    saved_meta, saved_mods = list(sys.meta_path), dict(sys.modules)
    theano, T, layers = E.install()
    yield theano, T, layers
    sys.meta_path[:] = saved_meta
    for k in list(sys.modules):
        if k.split(".")[0] in ("theano", "lasagne", "gensim") and k not in saved_mods:
            del sys.modules[k]


def test_tensor_methods_follow_theano(mods):
    a = np.arange(24.0).reshape(2, 3, 4)
    x = E.tt(a)
    assert np.array_equal(x.dimshuffle(1, 0, 2).numpy(), a.transpose(1, 0, 2))
    assert x.dimshuffle(1, 0, 2).dimshuffle(0, 1, "x", 2).shape == (3, 2, 1, 4)
    m = E.tt(np.ones((5, 7)))
    assert m.dimshuffle(1, 0, "x").shape == (7, 5, 1) and E.tt(np.ones(4)).dimshuffle("x", 0).shape == (1, 4)
    assert E.tt(np.ones(3)).dimshuffle([0, "x"]).shape == (3, 1)                      # list form (rnn_sampling.py:76)
    assert x.flatten(2).shape == (2, 12) and x.flatten(3).shape == (2, 3, 4) and x.flatten().shape == (24,)
    assert np.array_equal(x[:, ::-1].numpy(), a[:, ::-1])
    idx = E.tt(np.array([[2, 0], [1, 1]], dtype=np.int32))
    W = E.tt(np.arange(12.0).reshape(3, 4))
    assert np.array_equal(W[idx, :].numpy(), np.arange(12.0).reshape(3, 4)[[[2, 0], [1, 1]], :])     # (2, 2, 4): rows gathered
    assert np.array_equal(W[idx, :].sum(axis=-2).numpy(), np.arange(12.0).reshape(3, 4)[[[2, 0], [1, 1]], :].sum(-2))
    y = x
    y += 1                                           # symbolic += rebinds, the original is untouched
    assert float(x[0, 0, 0]) == 0.0 and float(y[0, 0, 0]) == 1.0
    assert x.astype("int32").dtype == torch.int64 and x.astype("float32").dtype == torch.float64


def test_ops(mods):
    theano, T, _ = mods
    a, b = np.random.default_rng(0).normal(size=(4, 3)), np.random.default_rng(1).normal(size=(3, 5))
    assert np.allclose(T.dot(E.tt(a), E.tt(b)).numpy(), a @ b)
    assert T.concatenate([E.tt(a), E.tt(a)], axis=1).shape == (4, 6)
    c = T.switch(E.tt(np.array([[1.0], [0.0], [1.0], [0.0]])), E.tt(a), E.tt(-a))
    assert np.array_equal(c.numpy(), np.where(np.array([[1], [0], [1], [0]]) != 0, a, -a))
    p = T.nnet.softmax(E.tt(a))
    assert np.allclose(p.numpy().sum(1), 1) and np.allclose(p.numpy(), np.exp(a) / np.exp(a).sum(1, keepdims=True))
    t = np.array([2, 0, 1, 1])
    assert np.allclose(T.nnet.categorical_crossentropy(p, E.tt(t)).numpy(), -np.log(p.numpy()[np.arange(4), t]))
    assert np.allclose(T.nnet.categorical_crossentropy(p, t).numpy(), -np.log(p.numpy()[np.arange(4), t]))
    sq = E.tt(np.arange(12.0).reshape(3, 4))
    assert np.array_equal(T.diag(sq).numpy(), [0.0, 5.0, 10.0])          # main diagonal of a non-square matrix
    assert np.allclose(T.nnet.sigmoid(E.tt(a)).numpy(), 1 / (1 + np.exp(-a))) and np.allclose(T.sqr(E.tt(a)).numpy(), a * a)
    assert T.ones((3, 1)).shape == (3, 1)


def test_grad_clip_is_identity_forward_and_clamps_backward(mods):
    theano, T, _ = mods
    x = torch.tensor([0.5, -2.0, 3.0], dtype=torch.float64, requires_grad=True)
    y = theano.gradient.grad_clip(E.tt(x), -1.0, 1.0)
    assert np.array_equal(y.detach().numpy(), [0.5, -2.0, 3.0])
    (y * torch.tensor([0.3, 5.0, -7.0], dtype=torch.float64)).sum().backward()
    assert np.array_equal(x.grad.numpy(), [0.3, 1.0, -1.0])


def test_scan_order_and_outputs(mods):
    theano, T, _ = mods
    seq = E.tt(np.arange(5.0).reshape(5, 1))
    w = E.tt(np.array([2.0]))

    def step(x_t, acc, w_):
        return acc * w_ + x_t
    out, upd = theano.scan(fn=step, sequences=[seq], outputs_info=[E.tt(np.zeros(1))], non_sequences=[w], strict=True)
    ref, acc = [], 0.0
    for t in range(5):
        acc = acc * 2 + t
        ref.append(acc)
    assert np.allclose(out.numpy()[:, 0], ref) and len(upd) == 0
    out_b, _ = theano.scan(fn=step, sequences=[seq], outputs_info=[E.tt(np.zeros(1))], non_sequences=[w], go_backwards=True)
    ref, acc = [], 0.0
    for t in reversed(range(5)):
        acc = acc * 2 + t
        ref.append(acc)
    assert np.allclose(out_b.numpy()[:, 0], ref)           # iteration order: the caller flips it back (sparse_lstm.py:489)

    def two(x_t, a, b):
        return [a + x_t, b * 2 + x_t]
    (oa, ob), _ = theano.scan(fn=two, sequences=seq, outputs_info=[E.tt(np.zeros(1)), E.tt(np.ones(1))])
    assert oa.shape == (5, 1) and np.allclose(oa.numpy()[:, 0], np.cumsum(np.arange(5.0)))


def test_layer_bookkeeping_follows_lasagne(mods):
    theano, T, L = mods
    E.new_network(dict(inputs=[np.zeros((2, 3, 1), np.int32), np.ones((2, 3))]))
    l_in, l_mask = L.InputLayer((2, 3, 1)), L.InputLayer((2, 3))
    assert l_in.input_var.dtype == torch.int64 and l_mask.input_var.shape == (2, 3)

    class Two(L.MergeLayer):
        def __init__(self, incomings):
            super(Two, self).__init__(incomings)
            self.a = self.add_param(E.Constant(1.0), (4,), name="a")
            self.frozen = self.add_param(E.Constant(2.0), (1, 4), name="frozen", trainable=False, regularizable=False)
            self.b = self.add_param(E.Constant(3.0), (4,), name="b", regularizable=False)

        def get_output_shape_for(self, shapes):
            return (shapes[0][0], 4)

        def get_output_for(self, inputs, **kw):
            return (self.a + self.b).dimshuffle("x", 0) * inputs[1].sum(axis=1).dimshuffle(0, "x")
    rec = Two([l_in, l_mask])
    out = L.DenseLayer(rec, num_units=5, nonlinearity=None)
    names = [p.pname for p in L.get_all_params(out)]
    assert names == ["a", "frozen", "b", "W", "b"]                                     # creation order, bottom layer first
    assert [p.pname for p in L.get_all_params(out, trainable=True)] == ["a", "b", "W", "b"]
    assert [p.pname for p in L.get_all_params(out, regularizable=True)] == ["a", "W"]
    assert out.W.shape == (4, 5) and out.output_shape == (2, 5)
    y = L.get_output(out)
    g = theano.grad(y.sum(), L.get_all_params(out, trainable=True))
    assert y.shape == (2, 5) and [tuple(x.shape) for x in g] == [(4,), (4,), (4, 5), (5,)]
    assert np.allclose(g[3].numpy(), 2.0)                                              # d sum / d bias = batch size
    vals = L.get_all_param_values(out)
    assert [v.shape for v in vals] == [(4,), (1, 4), (4,), (4, 5), (5,)]
    # seeded values go in by creation order
    E.new_network(dict(inputs=[np.zeros((2, 3, 1), np.int32), np.ones((2, 3))]), [np.full((4,), 7.0), np.zeros((1, 4)), np.zeros(4)])
    rec2 = Two([L.InputLayer((2, 3, 1)), L.InputLayer((2, 3))])
    assert float(rec2.a[0].detach()) == 7.0 and E.leftovers() == 0
