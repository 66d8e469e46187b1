"""Independent torch-autograd restatement of the `train.py -m RNN` hot path
--  TEST INFRASTRUCTURE ONLY (see oracle/rnn_oracle.py header: PARITY UNPINNED).

Written forward-only, straight from the reference's symbolic graph; the backward comes
from torch autograd with a custom GradClip function standing in for
theano.gradient.grad_clip [3P].  It exists to cross-check the hand-derived BPTT of
rnn_oracle.py (tests/test_oracle.py), and -- in float32 with all host threads -- as the
timed "port" CPU baseline of bench.py (Theano/Lasagne cannot be installed here).
"""
import torch

GRAD_CLIP = 100.0


class GradClip(torch.autograd.Function):
    """theano.gradient.grad_clip(x, -c, c): identity forward; backward clamps [3P]."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.clamp(-GRAD_CLIP, GRAD_CLIP)


def grad_clip(x):
    return GradClip.apply(x)


def _stack(layer, cell):
    # stacking order: sparse_lstm.py:348-360 (LSTM i,f,c,o), :737-749 (GRU r,u,c)
    order = {"LSTM": ("ingate", "forgetgate", "cell", "outgate"),
             "GRU": ("resetgate", "updategate", "hidden_update"),
             "Vanilla": ("hidden_update",)}[cell]
    W_in = torch.cat([layer["W_in_to_" + g] for g in order], dim=1)
    W_hid = torch.cat([layer["W_hid_to_" + g] for g in order], dim=1)
    b = torch.cat([layer["b_" + g] for g in order], dim=0)
    return W_in, W_hid, b


def layer_forward(layer, cell, inp, mask, index_input):
    """One recurrent layer; returns hid_out (T,B,H).  sparse_lstm.py:293-495 (LSTM),
    :690-864 (GRU), :1052-1211 (Vanilla); dense layers = Lasagne equivalents [3P]."""
    W_in, W_hid, b = _stack(layer, cell)
    H = W_hid.shape[0]
    if index_input:
        x = W_in[inp.long(), :].sum(dim=-2) + b
    else:
        x = inp @ W_in + b
    x = x.transpose(0, 1)                              # (T,B,GH)
    T, B = x.shape[0], x.shape[1]
    m = mask.transpose(0, 1).bool().unsqueeze(-1)      # (T,B,1)
    h = layer["hid_init"].expand(B, H)
    outs = []
    if cell == "LSTM":
        c = layer["cell_init"].expand(B, H)
        p_i, p_f, p_o = layer["W_cell_to_ingate"], layer["W_cell_to_forgetgate"], layer["W_cell_to_outgate"]
        for t in range(T):
            gates = grad_clip(x[t] + h @ W_hid)
            i = torch.sigmoid(gates[:, 0:H] + c * p_i)
            f = torch.sigmoid(gates[:, H:2 * H] + c * p_f)
            g = torch.tanh(gates[:, 2 * H:3 * H])
            c_new = f * c + i * g
            o = torch.sigmoid(gates[:, 3 * H:] + c_new * p_o)
            h_new = o * torch.tanh(c_new)
            c = torch.where(m[t], c_new, c)
            h = torch.where(m[t], h_new, h)
            outs.append(h)
    elif cell == "GRU":
        for t in range(T):
            hi = grad_clip(h @ W_hid)
            xi = grad_clip(x[t])
            r = torch.sigmoid(hi[:, 0:H] + xi[:, 0:H])
            u = torch.sigmoid(hi[:, H:2 * H] + xi[:, H:2 * H])
            q = grad_clip(xi[:, 2 * H:] + r * hi[:, 2 * H:])
            h_new = (1 - u) * h + u * torch.tanh(q)
            h = torch.where(m[t], h_new, h)
            outs.append(h)
    elif index_input:
        for t in range(T):
            hi = grad_clip(h @ W_hid)
            xi = grad_clip(x[t])
            h_new = torch.tanh(grad_clip(xi + hi))
            h = torch.where(m[t], h_new, h)
            outs.append(h)
    else:       # stock lasagne RecurrentLayer (recurrent_layers.py:94-104): hid_pre = hid.W_hid + input_n, one grad_clip,
        for t in range(T):                              # default nonlinearity rectify [3P]
            h_new = torch.relu(grad_clip(h @ W_hid + x[t]))
            h = torch.where(m[t], h_new, h)
            outs.append(h)
    return torch.stack(outs, dim=0)


# ---------------------------------------------------------------------------------------
# The same scans with the time loop under torch.jit.script (bench.py's second CPU baseline, "port_scripted"): no Python per step,
# element-wise chains fused by the profiling executor -- closer to what Theano's compiled scan does than the eager loop above.
# Gradient clipping (grad_clip nodes at +-100, inactive at the bench's magnitudes) is NOT in it: TorchScript has no custom
# autograd.Function.  A timing baseline only; parity never runs through it (tests/test_oracle.py holds it to the eager port's cost).
# ---------------------------------------------------------------------------------------
@torch.jit.script
def _lstm_scan(x, m, W_hid, h0, c0, p_i, p_f, p_o):
    # (unbind / chunk instead of x[t] / gates[:, a:b]: the backward of a select is a zero-filled tensor of the WHOLE input per step --
    # 200 x 78 MB at C2 --, which is what the eager loop above spends most of its time on)
    xs = x.unbind(0)
    ms = m.unbind(0)
    h, c = h0, c0
    outs = []
    for t in range(len(xs)):
        gi, gf, gg, go = (xs[t] + torch.mm(h, W_hid)).chunk(4, 1)
        i = torch.sigmoid(gi + c * p_i)
        f = torch.sigmoid(gf + c * p_f)
        g = torch.tanh(gg)
        c_new = f * c + i * g
        o = torch.sigmoid(go + c_new * p_o)
        h_new = o * torch.tanh(c_new)
        c = torch.where(ms[t], c_new, c)
        h = torch.where(ms[t], h_new, h)
        outs.append(h)
    return torch.stack(outs, dim=0)


@torch.jit.script
def _gru_scan(x, m, W_hid, h0):
    xs = x.unbind(0)
    ms = m.unbind(0)
    h = h0
    outs = []
    for t in range(len(xs)):
        hr, hu, hc = torch.mm(h, W_hid).chunk(3, 1)
        xr, xu, xc = xs[t].chunk(3, 1)
        r = torch.sigmoid(hr + xr)
        u = torch.sigmoid(hu + xu)
        h_new = (1 - u) * h + u * torch.tanh(xc + r * hc)
        h = torch.where(ms[t], h_new, h)
        outs.append(h)
    return torch.stack(outs, dim=0)


def layer_forward_scripted(layer, cell, inp, mask, index_input):
    """layer_forward with the scan under TorchScript (LSTM / GRU; everything else falls back to the eager loop)."""
    if cell not in ("LSTM", "GRU"):
        return layer_forward(layer, cell, inp, mask, index_input)
    W_in, W_hid, b = _stack(layer, cell)
    H = W_hid.shape[0]
    x = (W_in[inp.long(), :].sum(dim=-2) + b) if index_input else (inp @ W_in + b)
    x = x.transpose(0, 1).contiguous()
    m = mask.transpose(0, 1).bool().unsqueeze(-1)
    B = x.shape[1]
    h0 = layer["hid_init"].expand(B, H)
    if cell == "LSTM":
        return _lstm_scan(x, m, W_hid, h0, layer["cell_init"].expand(B, H), layer["W_cell_to_ingate"], layer["W_cell_to_forgetgate"],
                          layer["W_cell_to_outgate"])
    return _gru_scan(x, m, W_hid, h0)


def split_params(params, cell, layers, names_fn, embedding=0, bidirectional=False):
    per, pos = [], (1 if embedding else 0)
    for li, H in enumerate(layers):
        for _ in range(2 if bidirectional else 1):
            names = [n for n, _ in names_fn(cell, 1, H, dense=(li > 0 or embedding > 0))]
            per.append(dict(zip(names, params[pos:pos + len(names)])))
            pos += len(names)
    return per, params[pos], params[pos + 1]


def network_cost(params, cfg, batch, names_fn, scripted=False):
    """cost tensor of the whole network (rnn_one_hot.py:37-78 / rnn_sampling.py:93-137)."""
    lf = layer_forward_scripted if scripted else layer_forward
    cell, layers, emb, bi = cfg["cell"], cfg["layers"], cfg.get("embedding", 0), cfg.get("bidirectional", False)
    per, W_out, b_out = split_params(params, cell, layers, names_fn, emb, bi)
    inp = batch["X"]
    if emb:                                             # EmbeddingLayer + flatten(outdim=3) (recurrent_layers.py:46-50)
        X = batch["X"].long()
        inp = params[0][X].reshape(X.shape[0], X.shape[1], -1)
    D = 2 if bi else 1
    mask = batch["mask"]
    for li in range(len(layers)):
        outs, finals = [], []
        for d in range(D):          # --r_bi: second scan over the time-flipped input and mask, outputs flipped back
            src, m = (inp, mask) if d == 0 else (torch.flip(inp, dims=[1]), torch.flip(mask, dims=[1]))
            hid = lf(per[li * D + d], cell, src, m, index_input=(li == 0 and not emb))
            finals.append(hid[-1])
            outs.append(hid if d == 0 else torch.flip(hid, dims=[0]))
        inp = torch.cat(outs, dim=2).transpose(0, 1)
    h = torch.cat(finals, dim=1)
    pop = batch["pop"]
    B = h.shape[0]
    if cfg["loss"] == "CCE":
        logits = h @ W_out + b_out
        logp = torch.log_softmax(logits, dim=1)
        cost = (-logp[torch.arange(B), batch["target"].long()] / pop).mean()
        reg = cfg.get("regularization", 0.0)
        if reg > 0:
            cost = cost + reg * (b_out ** 2).sum()
        elif reg < 0:
            cost = cost - reg * b_out.abs().sum()
        return cost, h, logits
    cells = torch.cat([batch["target"].long(), batch["samples"].long()])
    a = h @ W_out[:, cells] + b_out[cells]
    rows = torch.arange(B)
    if cfg["loss"] == "Blackout":
        p = torch.softmax(a, dim=1)
        L = -torch.log(p[rows, rows]) - torch.log(1 - p[:, B:]).sum(dim=1)
    elif cfg["loss"] == "BPR":
        diff = (a - torch.diag(a).unsqueeze(1))[:, B:]
        L = -(torch.log(torch.sigmoid(-diff))).mean(dim=1)
    elif cfg["loss"] == "TOP1":
        diff = (a - torch.diag(a).unsqueeze(1))[:, B:]
        L = (torch.sigmoid(diff) + torch.sigmoid(a[:, B:] ** 2)).mean(dim=1)
    else:
        raise ValueError("Unknown loss function")
    return (L / pop).mean(), h, a


def cost_and_grads(np_params, cfg, np_batch, names_fn, dtype=torch.float64):
    params = [torch.tensor(p, dtype=dtype, requires_grad=True) for p in np_params]
    batch = {}
    for k, v in np_batch.items():
        t = torch.as_tensor(v)
        batch[k] = t.to(dtype) if t.is_floating_point() else t
    cost, h, act = network_cost(params, cfg, batch, names_fn)
    grads = torch.autograd.grad(cost, params, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
    return cost.item(), [g.numpy() for g in grads], h.detach().numpy(), act.detach().numpy()


class TorchTrainer(object):
    """float32 multi-threaded port used ONLY as bench.py's timed cpu_baseline ("port")."""

    def __init__(self, np_params, cfg, names_fn, updater="adam", lr=1e-3, b1=0.9, b2=0.999, rho=0.9, scripted=False):
        self.scripted = scripted
        self.params = [torch.tensor(p, dtype=torch.float32, requires_grad=True) for p in np_params]
        self.cfg, self.names_fn = cfg, names_fn
        self.updater, self.lr, self.b1, self.b2, self.rho = updater, lr, b1, b2, rho
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def train_function(self, np_batch):
        batch = {}
        for k, v in np_batch.items():
            t = torch.as_tensor(v)
            batch[k] = t.to(torch.float32) if t.is_floating_point() else t
        cost, _, _ = network_cost(self.params, self.cfg, batch, self.names_fn, scripted=self.scripted)
        grads = torch.autograd.grad(cost, self.params, allow_unused=True)
        with torch.no_grad():
            if self.updater == "adam":
                self.t += 1
                a_t = self.lr * (1 - self.b2 ** self.t) ** 0.5 / (1 - self.b1 ** self.t)
            for p, g, m, v in zip(self.params, grads, self.m, self.v):
                if g is None:
                    g = torch.zeros_like(p)
                if self.updater == "adam":
                    m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                    v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                    p.sub_(a_t * m / (v.sqrt() + 1e-8))
                else:  # adagrad
                    m.addcmul_(g, g)
                    p.sub_(self.lr * g / (m + 1e-6).sqrt())
        return float(cost.detach())
