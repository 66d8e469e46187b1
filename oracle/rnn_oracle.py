"""CPU oracle for the `train.py -m RNN` hot path  --  TEST INFRASTRUCTURE ONLY.

PINNING: the reference (rdevooght/sequence-based-recommendations) ships no tests, golden
vectors or fixtures for this path, and Theano/Lasagne cannot be installed here (no
network, python2-only code), so the reference cannot be RUN as a whole.  What pins this
file instead:
  * the reference's OWN layer and cost source (sparse_lstm.py get_output_for of the three
    index-input cells, BlackoutLayer, rnn_one_hot.py / rnn_sampling.py _prepare_networks
    and loss functions, recurrent_layers.py wiring) executed through an eager stand-in for
    the Theano / Lasagne calls it makes (tools/theano_on_torch.py): cost, every parameter
    gradient (incl. cases where grad_clip decides the result), recurrent output, scores
    and the parameter order agree with this file to 1e-12
    (tests/golden/reference_layers/, tests/test_reference_layers.py);
  * an independent torch-autograd restatement (oracle/torch_ref.py) and central finite
    differences (tests/test_oracle.py) -- these also cover what the above cannot:
STILL UNPINNED ("parity unpinned" for these parts): the library code the reference calls
but does not contain -- Lasagne's stock LSTMLayer / GRULayer / RecurrentLayer (stacked
layers, layers after --r_emb), EmbeddingLayer, lasagne.updates.* -- restated here from
their published formulas.
This file is a float64 NumPy restatement of the algorithm with hand-derived BPTT.  Nothing
in the product path may import it; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg do.

Every function cites the reference file:line it follows (paths relative to the
reference checkout).  "[3P]" marks semantics that live in Theano/Lasagne (not vendored
in the reference): Lasagne `master` (>=0.2.dev1, no commit pinned by the reference,
Dockerfile:12-13) and Theano >=0.8.2 (requirements.txt:1); those are restated from
their published formulas and anchored on the reference's call sites.

Conventions
-----------
B batch, T max_length, F indices per step, H hidden units, G gates (LSTM 4, GRU 3,
Vanilla 1), N items.  Parameter lists are in Lasagne `get_all_param_values` order
(neural_networks/rnn_base.py:470-479, sparse_lstm.py:240-279, :660-676).
"""
import numpy as np

GRAD_CLIP = 100.0  # recurrent_layers.py:19 (`-g` is never forwarded, command_parser.py:40)

CELL_GATES = {"LSTM": 4, "GRU": 3, "Vanilla": 1}


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _clip(x):
    # theano.gradient.grad_clip: identity forward, clamps the incoming gradient [3P]
    return np.clip(x, -GRAD_CLIP, GRAD_CLIP)


# --------------------------------------------------------------------------------------
# Parameter creation (Lasagne init laws [3P]; order = get_all_param_values)
# --------------------------------------------------------------------------------------
def recurrent_param_shapes(cell, n_in, H, dense=False):
    """Per-layer parameter (name, shape) list in Lasagne creation order.

    LSTM: sparse_lstm.py:240-279 (ingate, forgetgate, cell, outgate triples, then the
    three peepholes, cell_init, hid_init).  GRU: sparse_lstm.py:660-676 (updategate,
    resetgate, hidden_update triples, hid_init).  Vanilla: sparse_lstm.py:1030-1042.
    dense=True: the layer is a stock Lasagne layer over a dense input (recurrent_layers.py:94-104: every layer above
    layer 0, and layer 0 behind --r_emb).  LSTMLayer / GRULayer create their parameters in the order above [3P];
    RecurrentLayer (= CustomRecurrentLayer over two DenseLayers) lists its own parameter first, then the children's:
    hid_init, input_to_hidden.W, input_to_hidden.b, hidden_to_hidden.W [3P] (names kept in this file's spelling).
    """
    if cell == "Vanilla" and dense:
        return [("hid_init", (1, H)), ("W_in_to_hidden_update", (n_in, H)), ("b_hidden_update", (H,)),
                ("W_hid_to_hidden_update", (H, H))]
    if cell == "LSTM":
        names = []
        for g in ("ingate", "forgetgate", "cell", "outgate"):
            names += [("W_in_to_" + g, (n_in, H)), ("W_hid_to_" + g, (H, H)), ("b_" + g, (H,))]
        names += [("W_cell_to_ingate", (H,)), ("W_cell_to_forgetgate", (H,)),
                  ("W_cell_to_outgate", (H,)), ("cell_init", (1, H)), ("hid_init", (1, H))]
        return names
    if cell == "GRU":
        names = []
        for g in ("updategate", "resetgate", "hidden_update"):
            names += [("W_in_to_" + g, (n_in, H)), ("W_hid_to_" + g, (H, H)), ("b_" + g, (H,))]
        names += [("hid_init", (1, H))]
        return names
    if cell == "Vanilla":
        return [("W_in_to_hidden_update", (n_in, H)), ("W_hid_to_hidden_update", (H, H)),
                ("b_hidden_update", (H,)), ("hid_init", (1, H))]
    raise ValueError("Unknown layer type")  # recurrent_layers.py:90


def model_param_shapes(cell, layers, n_items, n_in0=None, embedding=0, n_feat=1, bidirectional=False):
    """Whole-model list: layer 0 (index input, input_size = n_items + n_optional,
    rnn_one_hot.py:48-49), dense layers >= 1 (recurrent_layers.py:94-104), then the
    output DenseLayer/BlackoutLayer W (H_last, N), b (N,) (rnn_one_hot.py:65,
    rnn_sampling.py:131).  --r_emb E (recurrent_layers.py:46-50): an EmbeddingLayer W (input_size, E) comes first and
    layer 0 becomes a dense Lasagne layer over the n_feat * E flattened embedding."""
    if n_in0 is None:
        n_in0 = n_items
    shapes = []
    n_in = n_in0
    if embedding > 0:
        shapes.append(("emb.W", (n_in0, embedding)))
        n_in = n_feat * embedding
    D = 2 if bidirectional else 1
    for li, H in enumerate(layers):
        # --r_bi (recurrent_layers.py:70-76): a forward and a backwards layer over the same input, ConcatLayer on the
        # feature axis; parameters of the forward layer come first
        for d in range(D):
            tag = "l%d." % li if D == 1 else "l%d%s." % (li, "fb"[d])
            for name, shp in recurrent_param_shapes(cell, n_in, H, dense=(li > 0 or embedding > 0)):
                shapes.append((tag + name, shp))
        n_in = D * H
    shapes += [("out.W", (D * layers[-1], n_items)), ("out.b", (n_items,))]
    return shapes


def init_params(cell, layers, n_items, rng, n_in0=None, last_layer_init=1.0, dtype=np.float64, embedding=0, n_feat=1,
                bidirectional=False):
    """Lasagne default initialisers [3P]: Gate W_in/W_hid/W_cell Normal(std=0.1),
    b Constant(0), cell_init/hid_init Constant(0) (sparse_lstm.py:156-162); output W
    GlorotUniform(gain) = U(+-gain*sqrt(6/(fan_in+fan_out))), b 0 (rnn_sampling.py:131); the stock RecurrentLayer
    (dense Vanilla layers) draws W_in_to_hid / W_hid_to_hid from init.Uniform() = U(-0.01, 0.01) [3P]."""
    out = []
    for name, shp in model_param_shapes(cell, layers, n_items, n_in0, embedding, n_feat, bidirectional):
        base = name.split(".")[1]
        level = 0 if name in ("emb.W", "out.W", "out.b") else int(name.split(".")[0][1:].rstrip("fb"))
        if cell == "Vanilla" and base.startswith("W_") and name[0] == "l" and (level > 0 or embedding > 0):
            a = rng.uniform(-0.01, 0.01, size=shp)
        elif name == "emb.W":
            a = rng.normal(0.0, 0.01, size=shp)          # lasagne EmbeddingLayer default W=init.Normal() (std 0.01) [3P]
        elif name == "out.W":
            lim = last_layer_init * np.sqrt(6.0 / (shp[0] + shp[1]))
            a = rng.uniform(-lim, lim, size=shp)
        elif base.startswith("W_"):
            a = rng.normal(0.0, 0.1, size=shp)
        else:
            a = np.zeros(shp)
        out.append(a.astype(dtype))
    return out


def split_params(params, cell, layers, embedding=0, bidirectional=False):
    """Split the flat Lasagne-ordered list into per-layer dicts + output (W, b); with --r_emb the embedding table is the
    extra FIRST entry (params[0]); with --r_bi every layer contributes two dicts (forward, backwards)."""
    per = []
    pos = 1 if embedding else 0
    for li, H in enumerate(layers):
        for _ in range(2 if bidirectional else 1):
            names = [n for n, _ in recurrent_param_shapes(cell, 1, H, dense=(li > 0 or embedding > 0))]
            per.append(dict(zip(names, params[pos:pos + len(names)])))
            pos += len(names)
    W_out, b_out = params[pos], params[pos + 1]
    assert pos + 2 == len(params)
    return per, W_out, b_out


def stacked(layer, cell):
    """Stacked matrices exactly as the reference stacks them.
    LSTM [i,f,c,o] (sparse_lstm.py:348-360); GRU **[r,u,c]** although parameters are
    created update-first (sparse_lstm.py:737-749); Vanilla single (sparse_lstm.py:1099-1105)."""
    if cell == "LSTM":
        order = ("ingate", "forgetgate", "cell", "outgate")
    elif cell == "GRU":
        order = ("resetgate", "updategate", "hidden_update")
    else:
        order = ("hidden_update",)
    W_in = np.concatenate([layer["W_in_to_" + g] for g in order], axis=1)
    W_hid = np.concatenate([layer["W_hid_to_" + g] for g in order], axis=1)
    b = np.concatenate([layer["b_" + g] for g in order], axis=0)
    return W_in, W_hid, b, order


# --------------------------------------------------------------------------------------
# Recurrent layers: forward (with cache) and hand-derived backward
# --------------------------------------------------------------------------------------
def input_projection(layer, cell, inp, index_input):
    """x~ = W_in_stacked[idx].sum(-2) + b for index input (sparse_lstm.py:368, :755,
    :1111) or dot(x, W_in_stacked) + b for dense layers (Lasagne LSTMLayer/GRULayer
    precompute_input [3P]).  inp: (B,T,F) int or (B,T,D) float.  Returns (T,B,G*H)."""
    W_in, _, b, _ = stacked(layer, cell)
    if index_input:
        x = W_in[inp, :].sum(axis=-2) + b          # (B,T,GH)
    else:
        x = inp @ W_in + b
    return np.transpose(x, (1, 0, 2))              # dimshuffle(1,0,2): sparse_lstm.py:343


def recurrent_forward(layer, cell, xt, mask, relu=False):
    """Scan over T (sparse_lstm.py:474-481 LSTM, :843-850 GRU, :1190-1197 Vanilla).
    relu (dense Vanilla layers only): the stock lasagne RecurrentLayer's default nonlinearity is rectify, where the
    reference's own VanillaLayerOHEInput uses tanh (sparse_lstm.py:1015); its step clips the gradient of the summed
    pre-activation once [3P] -- the same gradient as the three clips of the index-input layer (clip is idempotent).

    xt (T,B,G*H) precomputed input; mask (B,T).  Returns hid_out (T,B,H) and a cache.
    Masked steps copy the previous state (sparse_lstm.py:417-425, :798-805, :1145-1152).
    """
    T, B, _ = xt.shape
    _, W_hid, _, _ = stacked(layer, cell)
    H = W_hid.shape[0]
    m = np.transpose(mask, (1, 0)).astype(bool)    # (T,B)
    h = np.repeat(layer["hid_init"], B, axis=0)    # T.dot(ones, hid_init): :438-445
    hs = np.zeros((T + 1, B, H)); hs[0] = h
    cache = {"cell": cell, "xt": xt, "m": m, "hs": hs, "relu": relu}
    if cell == "LSTM":
        c = np.repeat(layer["cell_init"], B, axis=0)
        cs = np.zeros((T + 1, B, H)); cs[0] = c
        gi = np.zeros((T, B, H)); gf = np.zeros((T, B, H)); gg = np.zeros((T, B, H))
        go = np.zeros((T, B, H)); cn = np.zeros((T, B, H))
        p_i, p_f, p_o = layer["W_cell_to_ingate"], layer["W_cell_to_forgetgate"], layer["W_cell_to_outgate"]
        for t in range(T):
            a = xt[t] + h @ W_hid                                    # :383
            i = sigmoid(a[:, 0 * H:1 * H] + c * p_i)                 # :397-402
            f = sigmoid(a[:, 1 * H:2 * H] + c * p_f)
            g = np.tanh(a[:, 2 * H:3 * H])
            c_new = f * c + i * g                                    # :407
            o = sigmoid(a[:, 3 * H:4 * H] + c_new * p_o)             # :409-411
            h_new = o * np.tanh(c_new)                               # :414
            gi[t], gf[t], gg[t], go[t], cn[t] = i, f, g, o, c_new
            mt = m[t][:, None]
            c = np.where(mt, c_new, c); h = np.where(mt, h_new, h)   # :422-423
            cs[t + 1] = c; hs[t + 1] = h
        cache.update(cs=cs, gi=gi, gf=gf, gg=gg, go=go, cn=cn)
    elif cell == "GRU":
        gr = np.zeros((T, B, H)); gu = np.zeros((T, B, H)); gc = np.zeros((T, B, H)); hic = np.zeros((T, B, H))
        for t in range(T):
            hi = h @ W_hid                                           # :766
            r = sigmoid(hi[:, 0:H] + xt[t][:, 0:H])                  # :780-783
            u = sigmoid(hi[:, H:2 * H] + xt[t][:, H:2 * H])
            q = xt[t][:, 2 * H:] + r * hi[:, 2 * H:]                 # :786-788
            cc = np.tanh(q)
            h_new = (1 - u) * h + u * cc                             # :795
            gr[t], gu[t], gc[t], hic[t] = r, u, cc, hi[:, 2 * H:]
            h = np.where(m[t][:, None], h_new, h)                    # :803
            hs[t + 1] = h
        cache.update(gr=gr, gu=gu, gc=gc, hic=hic)
    else:  # Vanilla
        hn = np.zeros((T, B, H))
        for t in range(T):
            pre = xt[t] + h @ W_hid                                  # :1122-1143
            h_new = np.maximum(pre, 0.0) if relu else np.tanh(pre)
            hn[t] = h_new
            h = np.where(m[t][:, None], h_new, h)                    # :1150
            hs[t + 1] = h
        cache.update(hn=hn)
    return hs[1:], cache


def recurrent_backward(layer, cell, cache, dhid_out):
    """Hand-derived BPTT of recurrent_forward (the reference gets it from theano.grad
    through scan [3P]); grad_clip nodes clamp the gradient at: LSTM gates
    (sparse_lstm.py:386-388); GRU input_n, hid_input, hidden_update (:768-772, :789-791);
    Vanilla input_n, hid_input, hidden_update (:1125-1128, :1140).

    dhid_out (T,B,H): gradient wrt every hid_out[t] (only [-1] is non-zero for the last
    layer, only_return_final: sparse_lstm.py:485-486).
    Returns dict: dxt (T,B,G*H), dW_hid, peepholes / inits grads.
    """
    xt, m, hs = cache["xt"], cache["m"], cache["hs"]
    T, B, GH = xt.shape
    _, W_hid, _, _ = stacked(layer, cell)
    H = W_hid.shape[0]
    dW_hid = np.zeros_like(W_hid)
    dxt = np.zeros_like(xt)
    dh = np.zeros((B, H))
    out = {}
    if cell == "LSTM":
        cs = cache["cs"]
        p_i, p_f, p_o = layer["W_cell_to_ingate"], layer["W_cell_to_forgetgate"], layer["W_cell_to_outgate"]
        dp_i = np.zeros(H); dp_f = np.zeros(H); dp_o = np.zeros(H)
        dc = np.zeros((B, H))
        for t in range(T - 1, -1, -1):
            dh = dh + dhid_out[t]
            mt = m[t][:, None]
            i, f, g, o, c_new = cache["gi"][t], cache["gf"][t], cache["gg"][t], cache["go"][t], cache["cn"][t]
            c_prev, h_prev = cs[t], hs[t]
            dh_new = np.where(mt, dh, 0.0); dh_pass = np.where(mt, 0.0, dh)
            dc_new = np.where(mt, dc, 0.0); dc_pass = np.where(mt, 0.0, dc)
            tc = np.tanh(c_new)
            do = dh_new * tc
            dz_o = do * o * (1 - o)
            dc_new = dc_new + dh_new * o * (1 - tc * tc) + dz_o * p_o
            di = dc_new * g; df = dc_new * c_prev; dg = dc_new * i
            dz_i = di * i * (1 - i); dz_f = df * f * (1 - f); da_c = dg * (1 - g * g)
            dp_o += (dz_o * c_new).sum(0); dp_i += (dz_i * c_prev).sum(0); dp_f += (dz_f * c_prev).sum(0)
            da = _clip(np.concatenate([dz_i, dz_f, da_c, dz_o], axis=1))
            dW_hid += h_prev.T @ da
            dxt[t] = da
            dc = dc_pass + dc_new * f + dz_i * p_i + dz_f * p_f
            dh = dh_pass + da @ W_hid.T
        out.update(dp_i=dp_i, dp_f=dp_f, dp_o=dp_o, dcell_init=dc.sum(0, keepdims=True))
    elif cell == "GRU":
        for t in range(T - 1, -1, -1):
            dh = dh + dhid_out[t]
            mt = m[t][:, None]
            r, u, cc, hic = cache["gr"][t], cache["gu"][t], cache["gc"][t], cache["hic"][t]
            h_prev = hs[t]
            dh_new = np.where(mt, dh, 0.0); dh_pass = np.where(mt, 0.0, dh)
            du = dh_new * (cc - h_prev)
            dcc = dh_new * u
            dq = _clip(dcc * (1 - cc * cc))
            dr = dq * hic
            dz_r = dr * r * (1 - r); dz_u = du * u * (1 - u)
            dxi = _clip(np.concatenate([dz_r, dz_u, dq], axis=1))
            dhi = _clip(np.concatenate([dz_r, dz_u, dq * r], axis=1))
            dW_hid += h_prev.T @ dhi
            dxt[t] = dxi
            dh = dh_pass + dh_new * (1 - u) + dhi @ W_hid.T
    else:
        for t in range(T - 1, -1, -1):
            dh = dh + dhid_out[t]
            mt = m[t][:, None]
            hn = cache["hn"][t]
            dh_new = np.where(mt, dh, 0.0); dh_pass = np.where(mt, 0.0, dh)
            dq = _clip(dh_new * ((hn > 0) if cache.get("relu") else (1 - hn * hn)))
            dxi = _clip(dq); dhi = _clip(dq)
            dW_hid += hs[t].T @ dhi
            dxt[t] = dxi
            dh = dh_pass + dhi @ W_hid.T
    out.update(dxt=dxt, dW_hid=dW_hid, dhid_init=dh.sum(0, keepdims=True))
    return out


def layer_grads_to_list(layer, cell, inp, index_input, bw):
    """Map stacked gradients back to the Lasagne per-gate parameter order; returns
    (list of grads for this layer, d_input (B,T,D) or None for index input).
    Index-input gradient = AdvancedIncSubtensor: duplicates accumulate [3P]."""
    W_in, W_hid, b, order = stacked(layer, cell)
    H = W_hid.shape[0]
    dxt = bw["dxt"]                                   # (T,B,GH)
    dx_bt = np.transpose(dxt, (1, 0, 2))              # (B,T,GH)
    if index_input:
        dW_in = np.zeros_like(W_in)
        dflat = dx_bt.reshape(-1, dx_bt.shape[-1])
        for f in range(inp.shape[-1]):               # sum(axis=-2) fans the grad out to every f
            np.add.at(dW_in, inp[..., f].reshape(-1), dflat)
        d_inp = None
    else:
        flat_in = inp.reshape(-1, inp.shape[-1])
        dW_in = flat_in.T @ dx_bt.reshape(-1, dx_bt.shape[-1])
        d_inp = dx_bt @ W_in.T
    db = dxt.sum(axis=(0, 1))
    g = {}
    for k, name in enumerate(order):
        g["W_in_to_" + name] = dW_in[:, k * H:(k + 1) * H]
        g["W_hid_to_" + name] = bw["dW_hid"][:, k * H:(k + 1) * H]
        g["b_" + name] = db[k * H:(k + 1) * H]
    if cell == "LSTM":
        g["W_cell_to_ingate"] = bw["dp_i"]; g["W_cell_to_forgetgate"] = bw["dp_f"]
        g["W_cell_to_outgate"] = bw["dp_o"]; g["cell_init"] = bw["dcell_init"]
    g["hid_init"] = bw["dhid_init"]
    names = [n for n, _ in recurrent_param_shapes(cell, 1, H, dense=not index_input)]
    return [g[n] for n in names], d_inp


# --------------------------------------------------------------------------------------
# Whole-network forward / backward
# --------------------------------------------------------------------------------------
def network_forward(params, cell, layers, X, mask, embedding=0, bidirectional=False):
    """recurrent_layers.py:57-68: layer 0 index-input, later layers dense; only the
    last layer returns its final step (sparse_lstm.py:485-486: hid_out[-1], valid
    because X is left-aligned and masked steps copy state).  Returns h_last (B,H), caches.
    embedding > 0 (--r_emb, recurrent_layers.py:46-50): X -> W_emb[X] (B,T,F,E) flattened to (B,T,F*E), every layer dense.
    bidirectional (--r_bi, :70-76): per layer a second scan with go_backwards=True over the same input (time axis flipped,
    mask included: the padded steps come FIRST and copy hid_init); sequence outputs are flipped back (sparse_lstm.py:492-493)
    and concatenated on the feature axis; with only_return_final both directions contribute their LAST scan output
    (:485-486), i.e. the backwards one its state after consuming x_0."""
    D = 2 if bidirectional else 1
    per, W_out, b_out = split_params(params, cell, layers, embedding, bidirectional)
    caches = []
    inp = X
    if embedding:
        inp = params[0][X, :].reshape(X.shape[0], X.shape[1], -1)
    finals = []
    for li in range(len(layers)):
        index_input = (li == 0 and not embedding)
        outs, finals = [], []
        for d in range(D):
            layer = per[li * D + d]
            src, m = (inp, mask) if d == 0 else (inp[:, ::-1], mask[:, ::-1])
            xt = input_projection(layer, cell, src, index_input=index_input)
            hid, cache = recurrent_forward(layer, cell, xt, m, relu=(cell == "Vanilla" and not index_input))
            cache["inp"] = src
            caches.append(cache)
            finals.append(cache["hs"][-1])
            outs.append(hid if d == 0 else hid[::-1])
        inp = np.transpose(np.concatenate(outs, axis=2), (1, 0, 2))     # (B,T,D*H) dimshuffle back: :489
    return np.concatenate(finals, axis=1), caches


def network_backward(params, cell, layers, caches, dh_last, embedding=0, X=None, bidirectional=False):
    D = 2 if bidirectional else 1
    per, _, _ = split_params(params, cell, layers, embedding, bidirectional)
    grads = [None] * len(per)
    T, B, _ = caches[-1]["xt"].shape
    HL = layers[-1]
    # gradient wrt every step of each direction's scan output, in that scan's own time order
    dhid = []
    for d in range(D):
        g = np.zeros((T, B, HL)); g[-1] = dh_last[:, d * HL:(d + 1) * HL]
        dhid.append(g)
    d_inp = None
    for li in range(len(layers) - 1, -1, -1):
        index_input = (li == 0 and not embedding)
        d_inp = None
        for d in range(D):
            k = li * D + d
            bw = recurrent_backward(per[k], cell, caches[k], dhid[d])
            gl, di = layer_grads_to_list(per[k], cell, caches[k]["inp"], index_input, bw)
            grads[k] = gl
            if di is not None:
                di = di if d == 0 else di[:, ::-1]            # back to forward time
                d_inp = di if d_inp is None else d_inp + di
        if li > 0:
            H = layers[li - 1]
            dseq = np.transpose(d_inp, (1, 0, 2))             # (T,B,D*H) wrt the concatenated output of layer li-1
            dhid = [dseq[:, :, :H]] + ([dseq[::-1, :, H:2 * H]] if D == 2 else [])
    flat = []
    if embedding:      # EmbeddingLayer gradient = AdvancedIncSubtensor over the indices: duplicates accumulate [3P]
        dE = np.zeros_like(params[0])
        np.add.at(dE, X.reshape(-1), d_inp.reshape(-1, embedding))
        flat.append(dE)
    for gl in grads:
        flat += gl
    return flat


# --------------------------------------------------------------------------------------
# Output layers + costs
# --------------------------------------------------------------------------------------
def softmax_rows(a):
    e = np.exp(a - a.max(axis=1, keepdims=True))
    return e / e.sum(axis=1, keepdims=True)


def cce_cost_and_grads(h, W_out, b_out, target, target_popularity, regularization=0.0, Bglobal=None):
    """rnn_one_hot.py:65-77: DenseLayer(N, softmax) [3P]; cost = mean(CCE / pop);
    +reg*sum(b^2) if reg>0, +|reg|*sum|b| if reg<0 (bias only).
    Returns cost, logits, (dh, dW_out, db_out).  Bglobal (data-parallel restatement): h holds
    R of the Bglobal rows; the returned cost/grads are this shard's share (shares sum to the
    full-batch values, including the regulariser's)."""
    R = h.shape[0]
    B = Bglobal or R
    logits = h @ W_out + b_out
    p = softmax_rows(logits)
    nll = -np.log(p[np.arange(R), target])
    cost = (nll / target_popularity).sum() / B
    dlog = p.copy(); dlog[np.arange(R), target] -= 1.0
    dlog /= (target_popularity[:, None] * B)
    db = dlog.sum(0)
    regularization = regularization * R / B
    if regularization > 0.0:
        cost += regularization * (b_out ** 2).sum()
        db = db + 2.0 * regularization * b_out
    elif regularization < 0.0:
        cost -= regularization * np.abs(b_out).sum()
        db = db - regularization * np.sign(b_out)
    return cost, logits, (dlog @ W_out.T, h.T @ dlog, db)


def sampled_activation(h, W_out, b_out, target, samples):
    """BlackoutLayer.get_output_for, non-deterministic branch (sparse_lstm.py:42-54):
    output_cells = concat(targets, samples); a = h . W[:, cells] + b[cells]."""
    cells = np.concatenate([target, samples])
    return h @ W_out[:, cells] + b_out[cells], cells


def sampled_loss_rows(a, B, loss, row_offset=0):
    """Per-row loss and d loss / d a for the sampled heads; row b's positive is column b
    (targets = np.arange(batch_size), rnn_sampling.py:137), negatives are columns >= B.
    Blackout rnn_sampling.py:68-72; BPR :80-84; TOP1 :86-91 (last_layer_tanh is False
    from the CLI, command_parser.py:120-121).
    Data-parallel restatement: `a` may hold only the rows [row_offset, row_offset+R) of the
    global batch of B rows (all B+S columns); the formulas are unchanged."""
    R = a.shape[0]
    cols = row_offset + np.arange(R)          # positive column of each local row
    rows = np.arange(R)
    if loss == "Blackout":
        p = softmax_rows(a)
        L = -np.log(p[rows, cols]) - np.log(1 - p[:, B:]).sum(axis=1)
        dLdp = np.zeros_like(p)
        dLdp[rows, cols] = -1.0 / p[rows, cols]
        dLdp[:, B:] += 1.0 / (1 - p[:, B:])
        da = p * (dLdp - (dLdp * p).sum(axis=1, keepdims=True))
    elif loss == "BPR":
        diff = a[:, B:] - a[rows, cols][:, None]
        S = diff.shape[1]
        L = -np.log(sigmoid(-diff)).mean(axis=1)          # softplus(diff)
        dd = sigmoid(diff) / S
        da = np.zeros_like(a)
        da[:, B:] = dd
        da[rows, cols] -= dd.sum(axis=1)
    elif loss == "TOP1":
        neg = a[:, B:]
        diff = neg - a[rows, cols][:, None]
        S = diff.shape[1]
        s1 = sigmoid(diff); s2 = sigmoid(neg ** 2)
        L = (s1 + s2).mean(axis=1)
        d1 = s1 * (1 - s1) / S
        da = np.zeros_like(a)
        da[:, B:] = d1 + s2 * (1 - s2) * 2 * neg / S
        da[rows, cols] -= d1.sum(axis=1)
    elif loss == "SCCE":                                   # RNNCluster._cce_loss (rnn_cluster.py:158-162): CCE over the sampled columns
        p = softmax_rows(a)
        L = -np.log(p[rows, cols])
        da = p.copy()
        da[rows, cols] -= 1.0
    elif loss == "lin":                                    # rnn_cluster.py:164-167
        L = a[:, B:].sum(axis=1) - a[rows, cols]
        da = np.zeros_like(a)
        da[:, B:] = 1.0
        da[rows, cols] -= 1.0
    elif loss == "BPRelu":                                 # rnn_cluster.py:173-175: leaky_rectify(diff + 0.5), leakiness 0.01 [3P]
        diff = a[:, B:] - a[rows, cols][:, None]
        S = diff.shape[1]
        y = diff + 0.5
        L = np.where(y > 0, y, 0.01 * y).mean(axis=1)
        dd = np.where(y > 0, 1.0, 0.01) / S
        da = np.zeros_like(a)
        da[:, B:] = dd
        da[rows, cols] -= dd.sum(axis=1)
    else:
        raise ValueError("Unknown loss function")         # rnn_sampling.py:54
    return L, da


# --------------------------------------------------------------------------------------
# RNNCluster (rnn_cluster.py): the sampled head above (its own loss set, no popularity division) + a cluster head that reads the
# user representation h and trains ONLY its own two arrays -- the cluster-selection weights Wc (H, C) and the item / cluster
# repartition R (N, C) (rnn_cluster.py:283-285: updater(cost_clusters, params_clusters); nothing of it reaches the recurrent net)
# --------------------------------------------------------------------------------------
CLUSTER_LOSSES = {"CCE": "SCCE", "Blackout": "Blackout", "BPR": "BPR", "TOP1": "TOP1", "BPRelu": "BPRelu", "lin": "lin"}


def cluster_membership(r, scale, cluster_type):
    """target_and_samples_clusters (rnn_cluster.py:246-253) for rows r of R; also the pieces the backward needs"""
    sm = softmax_rows(scale * r)
    sg = sigmoid(scale * r)
    if cluster_type == "softmax":
        return sm, sm, None
    if cluster_type == "mix":
        return sm + sg, sm, sg
    return sg, None, sg


def cluster_cost_and_grads(h, Wc, R, target, cluster_samples, loss, scale, cluster_type, noise=None):
    """cost_clusters (rnn_cluster.py:237-256) and its gradients wrt (Wc, R).  noise: the (B, C) draw of cluster_selection_noise
    added to the selection activations in training (:241-242), None = off."""
    B = target.shape[0]
    z = h @ Wc
    if noise is not None:
        z = z + noise
    p = softmax_rows(scale * z)
    ids = np.concatenate([target, cluster_samples])
    M, sm, sg = cluster_membership(R[ids], scale, cluster_type)
    score = p @ M.T
    L, ds = sampled_loss_rows(score, B, CLUSTER_LOSSES[loss])
    cost = L.sum() / B
    ds = ds / B
    dp = ds @ M
    dM = ds.T @ p
    dz = scale * p * (dp - (dp * p).sum(axis=1, keepdims=True))
    dr = np.zeros_like(dM)
    if sm is not None:
        dr += scale * sm * (dM - (dM * sm).sum(axis=1, keepdims=True))
    if sg is not None:
        dr += scale * sg * (1.0 - sg) * dM
    dR = np.zeros_like(R)
    np.add.at(dR, ids, dr)
    return cost, score, (h.T @ dz, dR)


def cluster_hard(R, cluster_type):
    """_get_hard_clusters (rnn_cluster.py:293-300)"""
    if cluster_type == "softmax":
        return softmax_rows(100.0 * R)
    if cluster_type == "mix":
        return np.clip(softmax_rows(100.0 * R) + sigmoid(100.0 * R), 0.0, 1.0)
    return sigmoid(100.0 * R)


def cluster_test_rows(params, cfg, X, mask, exclude_ids, k=10):
    """RNNCluster's test function (rnn_cluster.py:327-352) for every row: (ids without clusters, ids inside the selected
    cluster, selected cluster, items in that cluster).  Scores = softmax over the full catalogue; the cluster version multiplies
    them by the hard membership column of the row's cluster (argmax of the selection activations); seen items score 0."""
    cell, layers, emb, bi = cfg["cell"], cfg["layers"], cfg.get("embedding", 0), cfg.get("bidirectional", False)
    cl = cfg["clusters"]
    R, Wc = params[-2], params[-1]
    _, W_out, b_out = split_params(params[:-2], cell, layers, emb, bi)
    h, _ = network_forward(params[:-2], cell, layers, X, mask, emb, bi)
    s1 = softmax_rows(h @ W_out + b_out)
    csel = np.argmax(h @ Wc, axis=1)
    used = cluster_hard(R, cl["type"])[:, csel].T            # (B, N)
    s2 = s1 * used
    out1, out2 = [], []
    for b in range(h.shape[0]):
        a1, a2 = s1[b].copy(), s2[b].copy()
        if exclude_ids is not None:
            a1[np.asarray(exclude_ids[b], dtype=np.int64)] = 0.0
            a2[np.asarray(exclude_ids[b], dtype=np.int64)] = 0.0
        out1.append(topk_ordered(a1, k)); out2.append(topk_ordered(a2, k))
    return np.array(out1), np.array(out2), csel, used.sum(axis=1), (s1, s2)


def sampled_cost_and_grads(h, W_out, b_out, target, samples, target_popularity, loss, row_offset=0):
    """rnn_sampling.py:131-137: cost = mean(loss_rows / target_popularity).  The
    gradient wrt W[:, cells] is an AdvancedIncSubtensor: duplicate cells accumulate [3P].
    `target` always lists the targets of ALL B rows of the global batch; h may hold only the
    rows [row_offset, row_offset+R) (data-parallel shard: its share of cost and grads)."""
    B = target.shape[0]
    a, cells = sampled_activation(h, W_out, b_out, target, samples)
    L, da = sampled_loss_rows(a, B, loss, row_offset)
    cost = (L / target_popularity).sum() / B
    da = da / (target_popularity[:, None] * B)
    dW = np.zeros_like(W_out); db = np.zeros_like(b_out)
    np.add.at(dW.T, cells, (h.T @ da).T)
    np.add.at(db, cells, da.sum(0))
    dh = da @ W_out[:, cells].T
    return cost, a, (dh, dW, db)


MARGIN_LOSSES = ("hinge", "logit", "logsig")


def margin_targets(X, mask, targets, n_items, balance=1.0, unique=True, default_target=None):
    """The dense (Y, weight) pair of RNNMargin._prepare_input (rnn_margin.py:112-147) for rows given as index arrays:
    X (B,T,F) / mask (B,T): the input items (feature 0), targets: list of per-row lists of positive item ids
    (SelectTargets' output, up to --n_targets of them).  weight = balance * len(target) / (N - len(target) - len(in_seq))
    everywhere, -1 on the targets, and -- with unique interactions -- 0 on the input items (applied LAST, so an input item
    that is also a target ends at 0); Y = the default target (zeros, or the popularity law of :149-161), 1 on the targets,
    0 on the input items."""
    B = len(targets)
    Y = np.zeros((B, n_items)); W = np.zeros((B, n_items))
    dflt = np.zeros(n_items) if default_target is None else np.asarray(default_target, dtype=np.float64)
    for i in range(B):
        n_in = int(np.asarray(mask[i]).sum())
        in_seq = [int(v) for v in np.asarray(X[i])[:n_in, 0]]
        tg = [int(t) for t in targets[i]]
        w = balance * len(tg) / float(n_items - len(tg) - len(in_seq))
        W[i, :] = w
        W[i, tg] = -1
        if unique:
            W[i, in_seq] = 0
        Y[i, :] = dflt
        Y[i, tg] = 1
        if unique:
            Y[i, in_seq] = 0
    return Y, W


def margin_default_target(item_popularity, n_users, min_access=0.05):
    """RNNMargin._default_target, popularity based (rnn_margin.py:155-159)."""
    view_prob = np.asarray(item_popularity, dtype=np.float64) / n_users
    return np.minimum(1 - view_prob, (1 - min_access) * view_prob / min_access)


def margin_cost_and_grads(h, W_out, b_out, Y, Wt, loss, Bglobal=None):
    """rnn_margin.py:62-69, :104-110: DenseLayer(N, linear) [3P]; cost = mean over rows of
    hinge relu((p - y) w).sum() | logit (sigmoid(p - y) w).sum() | logsig -log(sigmoid((y - p) w)).sum().
    Returns cost, p, (dh, dW_out, db_out); Bglobal as in cce_cost_and_grads."""
    R = h.shape[0]
    B = Bglobal or R
    p = h @ W_out + b_out
    if loss == "hinge":
        z = (p - Y) * Wt
        rows = np.maximum(z, 0.0).sum(1)
        dp = np.where(z > 0.0, Wt, 0.0)
    elif loss == "logit":
        s = sigmoid(p - Y)
        rows = (s * Wt).sum(1)
        dp = s * (1.0 - s) * Wt
    elif loss == "logsig":
        z = (Y - p) * Wt
        rows = -np.log(sigmoid(z)).sum(1)
        dp = (1.0 - sigmoid(z)) * Wt
    else:
        raise ValueError("Unknown loss function")          # rnn_margin.py:49
    cost = rows.sum() / B
    dp = dp / B
    return cost, p, (dp @ W_out.T, h.T @ dp, dp.sum(0))


def cost_and_grads(params, cfg, batch):
    """cost + gradient list (Lasagne parameter order) for one batch = the symbolic part
    of RNNBase._compile_train_function (rnn_base.py:175-186) before the updates."""
    cell, layers, emb, bi = cfg["cell"], cfg["layers"], cfg.get("embedding", 0), cfg.get("bidirectional", False)
    if cfg.get("clusters"):
        # RNNCluster: parameter list = RNNSampling's + [R, Wc] (the order RNNCluster.save appends them, rnn_cluster.py:517-521);
        # the train function returns `cost` only, `aux["cost_clusters"]` is the second one
        cl = cfg["clusters"]
        net, R, Wc = params[:-2], params[-2], params[-1]
        per, W_out, b_out = split_params(net, cell, layers, emb, bi)
        h, caches = network_forward(net, cell, layers, batch["X"], batch["mask"], emb, bi)
        B = batch["target"].shape[0]
        cost, act, (dh, dW, db) = sampled_cost_and_grads(h, W_out, b_out, batch["target"], batch["samples"], np.ones(B),
                                                         CLUSTER_LOSSES[cfg["loss"]])
        csm = batch["cluster_samples"] if batch.get("cluster_samples") is not None else batch["samples"]
        ccost, cscore, (dWc, dR) = cluster_cost_and_grads(h, Wc, R, batch["target"], csm, cfg["loss"], cl["scale"], cl["type"],
                                                          batch.get("cluster_noise"))
        grads = network_backward(net, cell, layers, caches, dh, emb, batch["X"], bi) + [dW, db, dR, dWc]
        return cost, grads, {"h": h, "act": act, "cost_clusters": ccost, "cluster_score": cscore}
    per, W_out, b_out = split_params(params, cell, layers, emb, bi)
    h, caches = network_forward(params, cell, layers, batch["X"], batch["mask"], emb, bi)
    if cfg["loss"] == "CCE":
        cost, act, (dh, dW, db) = cce_cost_and_grads(h, W_out, b_out, batch["target"], batch["pop"],
                                                     cfg.get("regularization", 0.0), batch.get("Bglobal"))
    elif cfg["loss"] in MARGIN_LOSSES:      # batch["Y"], batch["weight"]: the dense pair of RNNMargin._prepare_input
        cost, act, (dh, dW, db) = margin_cost_and_grads(h, W_out, b_out, batch["Y"], batch["weight"], cfg["loss"],
                                                        batch.get("Bglobal"))
    else:
        cost, act, (dh, dW, db) = sampled_cost_and_grads(h, W_out, b_out, batch["target"], batch["samples"],
                                                         batch["pop"], cfg["loss"], batch.get("row_offset", 0))
    grads = network_backward(params, cell, layers, caches, dh, emb, batch["X"], bi) + [dW, db]
    return cost, grads, {"h": h, "act": act}


# --------------------------------------------------------------------------------------
# Updaters (update_manager.py:24-82 -> lasagne.updates.* [3P], dense over every param)
# --------------------------------------------------------------------------------------
class Updater(object):
    def __init__(self, name, learning_rate, rho=0.9, beta1=0.9, beta2=0.999):
        self.name, self.lr, self.rho, self.b1, self.b2 = name, learning_rate, rho, beta1, beta2
        self.state = None
        self.t = 0

    def apply(self, params, grads):
        n = self.name
        if self.state is None:
            self.state = [[np.zeros_like(p), np.zeros_like(p)] for p in params]
        if n == "adam":                                  # lasagne.updates.adam, eps 1e-8
            self.t += 1
            a_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for p, g, st in zip(params, grads, self.state):
            if n == "adagrad":                           # eps 1e-6
                st[0] += g * g
                p -= self.lr * g / np.sqrt(st[0] + 1e-6)
            elif n == "rmsprop":                         # eps 1e-6
                st[0][...] = self.rho * st[0] + (1 - self.rho) * g * g
                p -= self.lr * g / np.sqrt(st[0] + 1e-6)
            elif n == "adadelta":                        # eps 1e-6
                st[0][...] = self.rho * st[0] + (1 - self.rho) * g * g
                upd = g * np.sqrt(st[1] + 1e-6) / np.sqrt(st[0] + 1e-6)
                p -= self.lr * upd
                st[1][...] = self.rho * st[1] + (1 - self.rho) * upd * upd
            elif n == "nesterov":                        # sgd + apply_nesterov_momentum
                st[0][...] = self.rho * st[0] - self.lr * g
                p += self.rho * st[0] - self.lr * g
            elif n == "adam":
                st[0][...] = self.b1 * st[0] + (1 - self.b1) * g
                st[1][...] = self.b2 * st[1] + (1 - self.b2) * g * g
                p -= a_t * st[0] / (np.sqrt(st[1]) + 1e-8)
            else:
                raise ValueError("Unknown update option")  # update_manager.py:22


def train_function(params, cfg, updater, batch):
    """One call of the compiled train function (rnn_base.py:185,290): cost of the
    batch *before* the update, parameters mutated in place."""
    cost, grads, _ = cost_and_grads(params, cfg, batch)
    updater.apply(params, grads)
    return cost


# --------------------------------------------------------------------------------------
# Test / predict path (rnn_base.py:132-159, :188-213; rnn_sampling.py:140-157)
# --------------------------------------------------------------------------------------
def predict_scores(params, cfg, X, mask):
    """predict_function output (rnn_base.py:188-194): one-hot head = softmax
    probabilities (DenseLayer nonlinearity, rnn_one_hot.py:65); sampling head = raw
    full activations (BlackoutLayer deterministic branch, sparse_lstm.py:37-40)."""
    cell, layers, emb, bi = cfg["cell"], cfg["layers"], cfg.get("embedding", 0), cfg.get("bidirectional", False)
    _, W_out, b_out = split_params(params, cell, layers, emb, bi)
    h, _ = network_forward(params, cell, layers, X, mask, emb, bi)
    logits = h @ W_out + b_out
    return (softmax_rows(logits) if cfg["loss"] == "CCE" else logits), logits      # (RNNMargin: linear DenseLayer, raw outputs)


def topk_ordered(scores_row, k):
    """np.argpartition(-out, range(k))[:k] (rnn_base.py:159,207): the k best ids in
    descending score order (ties: unspecified in the reference; fixtures avoid them)."""
    return np.argpartition(-scores_row, range(k))[:k]


def test_function(params, cfg, X, mask, exclude_ids, k=10):
    """test_function (rnn_base.py:196-211 / rnn_sampling.py:140-157): softmax
    probabilities * (1 - exclude) then ordered top-k, one row at a time."""
    cell, layers = cfg["cell"], cfg["layers"]
    _, logits = predict_scores(params, cfg, X, mask)
    # RNNMargin's output layer is linear: the generic test function (rnn_base.py:196-209) ranks its RAW outputs times
    # (1 - exclude), so a viewed item scores 0 there and outranks every item with a negative output
    p = logits if cfg["loss"] in MARGIN_LOSSES else softmax_rows(logits)
    out = []
    for b in range(p.shape[0]):
        row = p[b].copy()
        row[np.asarray(exclude_ids[b], dtype=np.int64)] *= 0.0
        out.append(topk_ordered(row, k))
    return np.array(out)


# --------------------------------------------------------------------------------------
# Host-side batch packing restated literally (loop form) for checking the host mirror
# --------------------------------------------------------------------------------------
def prepare_input_one_hot(sequences, max_length, n_items, item_popularity, diversity_bias):
    """RNNOneHot._prepare_input (rnn_one_hot.py:83-106), no optional features (F=1):
    X left-aligned zero-padded, mask, Y first target, pop = popularity[y]**db,
    exclude one-hot of the seen items (unused by train_function)."""
    B = len(sequences)
    X = np.zeros((B, max_length, 1), dtype=np.int32)
    mask = np.zeros((B, max_length))
    Y = np.zeros((B,), dtype=np.int32)
    pop = np.zeros((B,))
    exclude = np.zeros((B, n_items), dtype=np.float32)
    for i, (user_id, in_seq, target) in enumerate(sequences):
        X[i, :len(in_seq), 0] = [it[0] for it in in_seq]
        mask[i, :len(in_seq)] = 1
        Y[i] = target[0][0]
        pop[i] = item_popularity[target[0][0]] ** diversity_bias
        exclude[i, [j[0] for j in in_seq]] = 1
    return X, mask.astype(np.float32), Y, pop.astype(np.float32), exclude
