"""A small EAGER stand-in for the parts of the Theano and Lasagne APIs that the reference's RNN path touches, on
torch float64 tensors with autograd -- so that the reference's OWN layer and cost code can be executed here:

    neural_networks/sparse_lstm.py   LSTMLayerOHEInput / GRULayerOHEInput / VanillaLayerOHEInput.get_output_for,
                                     BlackoutLayer.get_output_for
    neural_networks/rnn_one_hot.py   _prepare_networks (softmax head, CCE cost / popularity, bias regularisation)
    neural_networks/rnn_sampling.py  _prepare_networks, _blackout_loss / _BPR_loss / _TOP1_loss
    neural_networks/recurrent_layers.py   which layer classes are built, in what order, with which arguments

Used only by tools/make_reference_layer_golden.py, in this container (it needs /root/reference); what it produces is
committed under tests/golden/reference_layers/.  Nothing under tests/, bench.py or the package imports this file.

What is restated here is LIBRARY behaviour, from the published Theano 0.8 / Lasagne 0.2.dev1 documentation -- not the
reference's code:
  theano.tensor   dot, concatenate, switch, tanh, log, sqr, ones, flatten, diag, nnet.sigmoid / softmax / relu /
                  categorical_crossentropy; tensor methods dimshuffle / astype / flatten(outdim) / negative-step slices /
                  integer-array indexing; `x += y` rebinds (symbolic variables are immutable)
  theano.scan     a Python loop over the leading axis (reversed for go_backwards), outputs stacked in iteration order
  theano.gradient.grad_clip   identity forward, gradient clipped elementwise on the way back
  theano.grad     torch.autograd.grad
  lasagne.layers  Layer / MergeLayer / InputLayer / DenseLayer / ConcatLayer / Gate, add_param tags, get_output in
                  topological order, get_all_params / get_all_param_values order
"Eager" means there are no symbolic placeholders: input variables take their value from FEED when they are created
(`T.ivector('target_output')` returns FEED['target_output']; the n-th InputLayer returns FEED['inputs'][n]), so a
network has to be rebuilt for every batch -- fine for a fixture generator.  Parameters take their initial values from
PARAM_VALUES (creation order) when that list is set."""
import importlib.abc
import importlib.machinery
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

FEED = {}
PARAM_VALUES = None          # list of numpy arrays consumed by add_param in creation order, or None: use the init spec
_input_layers_made = [0]


# --------------------------------------------------------------------------------------------- tensors
class TT(torch.Tensor):
    """torch tensor with the Theano variable methods the reference calls"""

    def dimshuffle(self, *pattern):
        if len(pattern) == 1 and isinstance(pattern[0], (list, tuple)):
            pattern = tuple(pattern[0])
        t = self.permute(*[p for p in pattern if p != "x"])
        for i, p in enumerate(pattern):
            if p == "x":
                t = t.unsqueeze(i)
        return t

    def astype(self, dtype):
        if "int" in str(dtype):
            return self.to(torch.int64)
        return self.to(torch.float64)                # floatX: everything floating stays float64 here

    def flatten(self, outdim=1):                     # Theano: keep the first outdim-1 axes, merge the rest
        return self.reshape(tuple(self.shape[:outdim - 1]) + (-1,))

    def __getitem__(self, idx):
        t = self
        items = list(idx) if isinstance(idx, tuple) else [idx]
        for d, it in enumerate(items):
            if isinstance(it, slice) and it.step is not None and it.step < 0:
                assert it.step == -1 and it.start is None and it.stop is None
                t = torch.Tensor.flip(t, [d])
                items[d] = slice(None)
            elif isinstance(it, torch.Tensor) and not it.dtype.is_floating_point:
                items[d] = it.to(torch.int64)
            elif isinstance(it, (np.ndarray, list)):
                items[d] = torch.as_tensor(np.asarray(it), dtype=torch.int64)
        return torch.Tensor.__getitem__(t, tuple(items))

    def dot(self, other):                            # Theano's x.dot(y) is a matrix product for 2-D operands
        return torch.matmul(self, other).as_subclass(TT)

    def __iadd__(self, other):
        return self + other

    def __isub__(self, other):
        return self - other

    def __imul__(self, other):
        return self * other


def tt(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x.as_subclass(TT)
    a = np.asarray(x)
    if dtype is None:
        dtype = torch.int64 if a.dtype.kind in "iu" else torch.float64
    return torch.as_tensor(a, dtype=dtype).as_subclass(TT)


class _GradClip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lo, hi):
        ctx.lo, ctx.hi = lo, hi
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.clamp(ctx.lo, ctx.hi), None, None


def scan(fn, sequences=None, outputs_info=None, non_sequences=None, go_backwards=False, truncate_gradient=-1,
         strict=False, n_steps=None):
    assert truncate_gradient == -1
    seqs = sequences if isinstance(sequences, (list, tuple)) else [sequences]
    prev = list(outputs_info)
    non = list(non_sequences or [])
    steps = range(seqs[0].shape[0])
    collected = [[] for _ in prev]
    single = False
    for t in (reversed(steps) if go_backwards else steps):
        out = fn(*([s[t] for s in seqs] + prev + non))
        if not isinstance(out, (list, tuple)):
            out, single = [out], True
        prev = list(out)
        for c, o in zip(collected, out):
            c.append(o)
    stacked = [torch.stack(c).as_subclass(TT) for c in collected]
    return (stacked[0] if single else stacked), OrderedDict()


# --------------------------------------------------------------------------------------------- module objects
class _Meta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Inert


class Inert(metaclass=_Meta):
    """whatever else gets imported and never used on this path"""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return Inert

    def __call__(self, *a, **k):
        return Inert()


class _Module(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return sys.modules.get(self.__name__ + "." + name, Inert)


def _mod(name, **attrs):
    m = _Module(name)
    m.__path__ = []
    m.__dict__.update(attrs)
    sys.modules[name] = m
    parent, _, leaf = name.rpartition(".")
    if parent:
        setattr(sys.modules[parent], leaf, m)
    return m


def _feed_var(name=None):
    return tt(FEED[name])


def _softmax(x):
    return torch.softmax(x, dim=-1).as_subclass(TT)


def _categorical_crossentropy(pred, targets):
    idx = torch.as_tensor(np.asarray(targets.detach() if isinstance(targets, torch.Tensor) else targets), dtype=torch.int64)
    return -torch.log(pred[torch.arange(pred.shape[0]), idx])


# --------------------------------------------------------------------------------------------- lasagne.layers
class Layer(object):
    def __init__(self, incoming, name=None):
        if isinstance(incoming, tuple):
            self.input_shape, self.input_layer = incoming, None
        else:
            self.input_shape, self.input_layer = incoming.output_shape, incoming
        self.name = name
        self.params = OrderedDict()

    @property
    def output_shape(self):
        return self.get_output_shape_for(self.input_shape)

    def get_output_shape_for(self, input_shape):
        return input_shape

    def get_params(self, **tags):
        result = list(self.params.keys())
        only = set(t for t, v in tags.items() if v)
        if only:
            result = [p for p in result if not (only - self.params[p])]
        exclude = set(t for t, v in tags.items() if not v)
        if exclude:
            result = [p for p in result if not (self.params[p] & exclude)]
        return result

    def add_param(self, spec, shape, name=None, **tags):
        global PARAM_VALUES
        if PARAM_VALUES is not None:
            value = np.asarray(PARAM_VALUES.pop(0), dtype=np.float64)
            assert value.shape == tuple(shape), (name, value.shape, shape)
        elif callable(spec):
            value = np.asarray(spec(shape), dtype=np.float64)
        else:
            value = np.asarray(spec, dtype=np.float64).reshape(shape)
        leaf = torch.tensor(value, dtype=torch.float64, requires_grad=True)
        p = leaf.as_subclass(TT)
        p.leaf, p.pname = leaf, name
        tags.setdefault("trainable", True)
        tags.setdefault("regularizable", True)
        self.params[p] = set(t for t, v in tags.items() if v)
        return p


class MergeLayer(Layer):
    def __init__(self, incomings, name=None):
        self.input_shapes = [i if isinstance(i, tuple) else i.output_shape for i in incomings]
        self.input_layers = [None if isinstance(i, tuple) else i for i in incomings]
        self.name = name
        self.params = OrderedDict()

    @property
    def output_shape(self):
        return self.get_output_shape_for(self.input_shapes)


class InputLayer(Layer):
    def __init__(self, shape, input_var=None, name=None, **kwargs):
        self.shape, self.name, self.params = shape, name, OrderedDict()
        if input_var is None:
            input_var = tt(FEED["inputs"][_input_layers_made[0]])
            _input_layers_made[0] += 1
        self.input_var = input_var

    @property
    def output_shape(self):
        return self.shape


def _identity(x):
    return x


_DEFAULT = object()


class DenseLayer(Layer):
    def __init__(self, incoming, num_units, W=None, b=_DEFAULT, nonlinearity=Inert, **kwargs):
        super(DenseLayer, self).__init__(incoming, **kwargs)
        self.nonlinearity = _identity if nonlinearity is None else nonlinearity
        self.num_units = num_units
        num_inputs = int(np.prod(self.input_shape[1:]))
        self.W = self.add_param(W if W is not None else GlorotUniform(), (num_inputs, num_units), name="W")
        # Lasagne: b omitted = Constant(0.), b=None = a layer without bias (rnn_cluster.py:239)
        self.b = None if b is None else self.add_param(Constant(0.0) if b is _DEFAULT else b, (num_units,), name="b", regularizable=False)

    def get_output_shape_for(self, input_shape):
        return (input_shape[0], self.num_units)

    def get_output_for(self, input, **kwargs):
        if input.ndim > 2:
            input = input.flatten(2)
        activation = torch.matmul(input, self.W)
        if self.b is not None:
            activation = activation + self.b.dimshuffle("x", 0)
        return self.nonlinearity(activation)


class ConcatLayer(MergeLayer):
    def __init__(self, incomings, axis=1, **kwargs):
        super(ConcatLayer, self).__init__(incomings, **kwargs)
        self.axis = axis

    def get_output_shape_for(self, input_shapes):
        out = list(input_shapes[0])
        out[self.axis] = sum(s[self.axis] for s in input_shapes)
        return tuple(out)

    def get_output_for(self, inputs, **kwargs):
        return torch.cat(list(inputs), dim=self.axis).as_subclass(TT)


class Gate(object):
    def __init__(self, W_in=None, W_hid=None, W_cell=Inert, b=None, nonlinearity=Inert):
        self.W_in = W_in if W_in is not None else Normal(0.1)
        self.W_hid = W_hid if W_hid is not None else Normal(0.1)
        if W_cell is not None:
            self.W_cell = Normal(0.1) if W_cell is Inert else W_cell
        self.b = b if b is not None else Constant(0.0)
        self.nonlinearity = _identity if nonlinearity is None else (sigmoid if nonlinearity is Inert else nonlinearity)


def get_all_layers(layer):
    seen, order = set(), []

    def visit(l):
        if l is None or id(l) in seen:
            return
        seen.add(id(l))
        for parent in (l.input_layers if hasattr(l, "input_layers") else [getattr(l, "input_layer", None)]):
            visit(parent)
        order.append(l)
    for l in (layer if isinstance(layer, (list, tuple)) else [layer]):
        visit(l)
    return order


def get_output(layer_or_layers, inputs=None, **kwargs):
    assert inputs is None
    done = {}
    for l in get_all_layers(layer_or_layers):
        if isinstance(l, InputLayer):
            done[id(l)] = l.input_var
        elif hasattr(l, "input_layers"):
            done[id(l)] = l.get_output_for([done[id(p)] for p in l.input_layers], **kwargs)
        else:
            done[id(l)] = l.get_output_for(done[id(l.input_layer)], **kwargs)
    if isinstance(layer_or_layers, (list, tuple)):
        return [done[id(l)] for l in layer_or_layers]
    return done[id(layer_or_layers)]


def get_all_params(layer, **tags):
    out = []
    for l in get_all_layers(layer):
        for p in l.get_params(**tags):
            if not any(p is q for q in out):
                out.append(p)
    return out


def get_all_param_values(layer, **tags):
    return [p.detach().numpy().copy() for p in get_all_params(layer, **tags)]


# --------------------------------------------------------------------------------------------- lasagne.init / nonlinearities
class _Init(object):
    def __call__(self, shape):
        return self.sample(shape)


class Normal(_Init):
    def __init__(self, std=0.01, mean=0.0):
        self.std, self.mean = std, mean

    def sample(self, shape):
        return np.random.normal(self.mean, self.std, size=shape)


class Constant(_Init):
    def __init__(self, val=0.0):
        self.val = val

    def sample(self, shape):
        return np.full(shape, self.val, dtype=np.float64)


class GlorotUniform(_Init):
    def __init__(self, gain=1.0, c01b=False):
        self.gain = gain

    def sample(self, shape):
        a = self.gain * np.sqrt(6.0 / (shape[0] + shape[1]))
        return np.random.uniform(-a, a, size=shape)


def sigmoid(x):
    return torch.sigmoid(x).as_subclass(TT)


def tanh(x):
    return torch.tanh(x).as_subclass(TT)


def grad(cost, wrt):
    gs = torch.autograd.grad(cost, list(wrt), allow_unused=True, retain_graph=True)
    return [torch.zeros_like(w) if g is None else g for g, w in zip(gs, wrt)]


def install(floatX="float64"):
    """registers the modules; everything else under theano.* / lasagne.* / gensim.* resolves to Inert"""
    class Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        def find_spec(self, name, path, target=None):
            if name.split(".")[0] in ("theano", "lasagne", "gensim") and name not in sys.modules:
                return importlib.machinery.ModuleSpec(name, self, is_package=True)

        def create_module(self, spec):
            m = _Module(spec.name)
            m.__path__ = []
            return m

        def exec_module(self, module):
            pass
    sys.meta_path.insert(0, Finder())

    def shared(value, **k):                          # a shared variable: a leaf that theano.grad can differentiate with respect to
        a = np.asarray(value)
        if a.dtype.kind != "f":
            return tt(a)
        leaf = torch.tensor(a.astype(np.float64), dtype=torch.float64, requires_grad=True)
        v = leaf.as_subclass(TT)
        v.leaf, v.pname = leaf, k.get("name")
        return v
    theano = _mod("theano", config=types.SimpleNamespace(floatX=floatX), scan=scan, grad=grad, shared=shared)
    _mod("theano.gradient", grad_clip=lambda x, lo, hi: _GradClip.apply(x, lo, hi).as_subclass(TT), grad=grad)
    T = _mod("theano.tensor",
             dot=lambda a, b: torch.matmul(a, b).as_subclass(TT),
             concatenate=lambda xs, axis=0: torch.cat([tt(x) for x in xs], dim=axis).as_subclass(TT),
             switch=lambda c, a, b: torch.where(c != 0, a, b).as_subclass(TT),
             tanh=tanh, log=lambda x: torch.log(x).as_subclass(TT), sqr=lambda x: (x * x),
             ones=lambda shape: tt(np.ones([int(s) for s in shape])),
             flatten=lambda x, outdim=1: x.flatten(outdim), diag=lambda x: torch.diagonal(x).as_subclass(TT),
             ivector=_feed_var, fvector=_feed_var, fmatrix=_feed_var, imatrix=_feed_var)
    _mod("theano.tensor.nnet", sigmoid=sigmoid, softmax=_softmax, categorical_crossentropy=_categorical_crossentropy,
         relu=lambda x: torch.relu(x).as_subclass(TT))

    class RandomStreams(object):
        def __init__(self, seed=None):
            pass
    _mod("theano.tensor.shared_randomstreams", RandomStreams=RandomStreams)

    _mod("lasagne")
    _mod("lasagne.nonlinearities", sigmoid=sigmoid, tanh=tanh, identity=_identity, softmax=_softmax, linear=_identity,
         leaky_rectify=lambda x: torch.where(x > 0, x, 0.01 * x).as_subclass(TT))      # LeakyRectify(leakiness=0.01) [3P]
    _mod("lasagne.init", Normal=Normal, Constant=Constant, GlorotUniform=GlorotUniform)
    _mod("lasagne.random", get_rng=lambda: np.random.RandomState(1))
    _mod("lasagne.utils", unroll_scan=Inert)
    _mod("lasagne.regularization", l2=lambda x: (x * x).sum(), l1=lambda x: x.abs().sum())
    layers = _mod("lasagne.layers", Layer=Layer, MergeLayer=MergeLayer, InputLayer=InputLayer, DenseLayer=DenseLayer,
                  ConcatLayer=ConcatLayer, Gate=Gate, get_output=get_output, get_all_params=get_all_params,
                  get_all_layers=get_all_layers, get_all_param_values=get_all_param_values)
    _mod("lasagne.layers.base", Layer=Layer, MergeLayer=MergeLayer)
    _mod("lasagne.layers.input", InputLayer=InputLayer)
    _mod("lasagne.layers.dense", DenseLayer=DenseLayer)
    _mod("lasagne.layers.recurrent", Gate=Gate)
    _mod("lasagne.layers.helper", get_output=get_output, get_all_params=get_all_params, get_all_layers=get_all_layers)
    return theano, T, layers


def new_network(feed, param_values=None):
    """call before every construction of a predictor's network"""
    global PARAM_VALUES
    FEED.clear()
    FEED.update(feed)
    _input_layers_made[0] = 0
    PARAM_VALUES = None if param_values is None else [np.asarray(p) for p in param_values]


def leftovers():
    return 0 if PARAM_VALUES is None else len(PARAM_VALUES)
