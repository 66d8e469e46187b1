"""Forward values and gradients of the REFERENCE's own layer and cost code (/root/reference/neural_networks), executed
through tools/theano_on_torch.py (an eager stand-in for the Theano / Lasagne calls that code makes, on torch float64
with autograd).  Runs only in this container; the fixtures are committed under tests/golden/reference_layers/.

For every case the reference builds its predictor from a command line (helpers/command_parser.py), `_prepare_networks`
runs (rnn_one_hot.py:36-78 / rnn_sampling.py:96-138: recurrent_layers.py picks the layer classes, sparse_lstm.py's
get_output_for computes the recurrence, the cost lines compute the cost), and we record
    parameter names / shapes in lasagne.layers.get_all_params order      (pins the checkpoint layout)
    cost, d cost / d every parameter (theano.grad -> autograd)           (pins the forward AND where grad_clip sits)
    the recurrent stack's output, the deterministic output (rnn_base.py:192 / rnn_sampling.py:144)
for seeded parameters and batches in the format of the other fixtures (tests/golden/*.npz).
What this does NOT pin: Theano's and Lasagne's own library code (dot, scan, the stock LSTMLayer / GRULayer used for
stacked layers and after an embedding, lasagne.updates.*) -- those are restated in the stand-in / the oracle.

    python tools/make_reference_layer_golden.py
"""
import builtins
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import parity_util as PU                   # noqa: E402   (seeded parameters and batches, same helper as the other fixtures)
import theano_on_torch as E                # noqa: E402

# name: (cell, hidden, loss, N items, B, T, S samples, F indices per step, n_opt, bidirectional, extra argv)
CASES = {
    "gru16_cce": ("GRU", 16, "CCE", 40, 7, 9, 0, 1, 0, False, []),
    "lstm12_cce": ("LSTM", 12, "CCE", 45, 9, 8, 0, 1, 0, False, []),
    "vanilla8_cce": ("Vanilla", 8, "CCE", 30, 5, 6, 0, 1, 0, False, []),
    "gru12_blackout": ("GRU", 12, "Blackout", 40, 6, 7, 5, 1, 0, False, []),
    "lstm10_bpr": ("LSTM", 10, "BPR", 40, 6, 7, 5, 1, 0, False, []),
    "gru10_top1": ("GRU", 10, "TOP1", 40, 6, 7, 5, 1, 0, False, []),
    "lstm8_cce_rf": ("LSTM", 8, "CCE", 25, 5, 6, 0, 2, 10, False, ["--rf"]),
    "gru8_cce_bi": ("GRU", 8, "CCE", 30, 6, 7, 0, 1, 0, True, ["--r_bi"]),
    "lstm6_blackout_bi": ("LSTM", 6, "Blackout", 30, 5, 6, 4, 1, 0, True, ["--r_bi"]),
    "gru12_cce_reg": ("GRU", 12, "CCE", 35, 6, 7, 0, 1, 0, False, ["-r", "0.05"]),
    "vanilla10_cce_l1": ("Vanilla", 10, "CCE", 35, 6, 7, 0, 1, 0, False, ["-r", "-0.05"]),          # negative -r: L1 on the bias
    "lstm8_top1_ri": ("LSTM", 8, "TOP1", 30, 5, 6, 4, 1, 0, False, ["--repeated_interactions"]),     # nothing excluded at test time
    # the widths of the benchmark kernels (rec_*_x6p with fp16 / bf16 split products, rec_*_x6s): scale 0.1 keeps 128-wide
    # layers well conditioned
    "gru128_cce": ("GRU", 128, "CCE", 30, 6, 8, 0, 1, 0, False, []),
    "vanilla128_blackout": ("Vanilla", 128, "Blackout", 30, 5, 7, 4, 1, 0, False, []),
    "lstm128_cce": ("LSTM", 128, "CCE", 30, 5, 6, 0, 1, 0, False, []),
    # targets with a tiny popularity weight -> gate gradients far beyond the clip at 100 (recurrent_layers.py:18): it bites
    "gru12_cce_clip": ("GRU", 12, "CCE", 35, 6, 8, 0, 1, 0, False, []),
    "lstm8_cce_clip": ("LSTM", 8, "CCE", 35, 6, 8, 0, 1, 0, False, []),
    "vanilla8_cce_clip": ("Vanilla", 8, "CCE", 35, 6, 8, 0, 1, 0, False, []),
    "gru8_bpr_clip_bi": ("GRU", 8, "BPR", 35, 6, 8, 4, 1, 0, True, ["--r_bi"]),
    # RNNMargin (rnn_margin.py): --loss hinge | logit | logsig, multi-target; S here = --n_targets.  Y / weight come from the
    # reference's own _prepare_input
    "gru12_hinge": ("GRU", 12, "hinge", 40, 6, 7, 3, 1, 0, False, ["--balance", "1.5"]),
    "lstm10_logit": ("LSTM", 10, "logit", 40, 6, 7, 2, 1, 0, False, []),
    "gru10_logsig_ri": ("GRU", 10, "logsig", 40, 6, 7, 1, 1, 0, False, ["--repeated_interactions", "--balance", "0.5"]),
    "vanilla8_hinge_pb": ("Vanilla", 8, "hinge", 30, 5, 6, 2, 1, 0, False, ["--pb", "--min_access", "0.1"]),
    "gru128_logsig": ("GRU", 128, "logsig", 30, 6, 8, 2, 1, 0, False, []),
}
# RNNCluster (rnn_cluster.py: --clusters C): name: (cell, hidden, loss, N, B, T, S, cluster dict, extra argv).  Recorded: both
# costs, d cost / d every net parameter, d cost_clusters / d (Wc, R), the selection activations, the hard clusters, both test scores
CLUSTER_CASES = {
    "cl_gru12_cce_mix": ("GRU", 12, "CCE", 40, 6, 7, 5, dict(n=4, type="mix", scale=1.0), []),
    "cl_lstm10_blackout_softmax": ("LSTM", 10, "Blackout", 40, 6, 7, 5, dict(n=3, type="softmax", scale=2.5), []),
    "cl_gru10_bpr_sigmoid_cs": ("GRU", 10, "BPR", 40, 6, 7, 5, dict(n=5, type="sigmoid", scale=1.5, c_sampling=7), []),
    "cl_vanilla8_top1_mix": ("Vanilla", 8, "TOP1", 30, 5, 6, 4, dict(n=3, type="mix", scale=0.7), ["--repeated_interactions"]),
    "cl_gru8_bprelu_softmax": ("GRU", 8, "BPRelu", 30, 5, 6, 4, dict(n=4, type="softmax", scale=1.0), []),
    "cl_lstm8_lin_mix_cs": ("LSTM", 8, "lin", 30, 5, 6, 4, dict(n=2, type="mix", scale=1.2, c_sampling=3), []),
    "cl_gru128_cce_mix": ("GRU", 128, "CCE", 30, 6, 8, 4, dict(n=6, type="mix", scale=1.0), []),
}
MARGIN = ("hinge", "logit", "logsig")
POPSCALE = {"gru12_cce_clip": 3e-5, "lstm8_cce_clip": 3e-5, "vanilla8_cce_clip": 3e-5, "gru8_bpr_clip_bi": 1e-5}


def main():
    builtins.xrange = range
    sys.modules["cPickle"] = pickle
    _map = map
    builtins.map = lambda *a: list(_map(*a))

    class _Cast(dict):
        def __missing__(self, k):
            return lambda x: np.asarray(x, dtype=k)[()]
    if not hasattr(np, "cast"):
        np.cast = _Cast()
    E.install()
    sys.path[:0] = ["/root/reference"] + ["/root/reference/" + d for d in
                                          ("neural_networks", "helpers", "factorization", "lazy", "word2vec")]
    import helpers.command_parser as cp
    import train as reftrain
    import lasagne
    import theano

    outdir = os.path.join(ROOT, "tests", "golden", "reference_layers")
    os.makedirs(outdir, exist_ok=True)
    for name, (cell, H, loss, N, B, T, S, F, n_opt, bi, extra) in CASES.items():
        seed = sum(map(ord, name))
        params, cfg, batch = PU.build_case(cell, [H], loss, N, B, T, S=S, seed=seed, F=F, n_opt=n_opt, bi=bi,
                                           popscale=POPSCALE.get(name, 1.0), scale=0.1 if H >= 128 else None)
        if loss != "CCE" and loss not in MARGIN:
            assert len(batch["samples"]) == S
        def predictor():
            sys.argv = ["train.py", "-d", "/tmp/x/", "-b", str(B), "--max_length", str(T), "--r_t", cell, "--r_l", str(H),
                        "--loss", loss] + (["--n_targets", str(S)] if loss in MARGIN else ["--sampling", str(S or 32)]) + extra
            args = cp.command_parser(cp.predictor_command_parser, reftrain.training_command_parser, cp.early_stopping_command_parser)
            return args, cp.get_predictor(args)
        args, p = predictor()
        assert p._input_size() == F
        exclude = np.zeros((B, N))
        for b in range(B):
            exclude[b, batch["X"][b, :int(batch["mask"][b].sum()), 0]] = 1
        feed = dict(inputs=[batch["X"], batch["mask"].astype(np.float64)], target_output=batch["target"],
                    target_popularity=batch["pop"].astype(np.float64), samples=batch["samples"], excluded_items=exclude)
        margin = {}
        if loss in MARGIN:
            # (user, input sequence, targets) rows as _gen_mini_batch hands them over; 1 .. n_targets positives per row, some of
            # them repeated, one that also occurs in the row's input
            import types
            rng = np.random.RandomState(seed)
            item_pop = rng.randint(1, 50, size=N)
            p.n_items = N
            p.dataset = types.SimpleNamespace(training_set=types.SimpleNamespace(n_users=60), item_popularity=item_pop)
            seqs, tg = [], -np.ones((B, S), dtype=np.int32)
            for b in range(B):
                n_in = int(batch["mask"][b].sum())
                k = 1 + (b % S)
                t_b = [int(v) for v in rng.randint(0, N, size=k)]
                if b == 1:
                    t_b[0] = int(batch["X"][b, 0, 0])
                if b == 2 and k > 1:
                    t_b[1] = t_b[0]
                tg[b, :k] = t_b
                seqs.append([b, [(int(i), 1.0) for i in batch["X"][b, :n_in, 0]], [(t, 1.0) for t in t_b]])
            Xr, mr, Yr, Wr, er = p._prepare_input(seqs)                 # the reference's own packing (rnn_margin.py:112-147)
            assert np.array_equal(Xr, batch["X"]) and np.array_equal(mr, batch["mask"]) and np.array_equal(er, exclude)
            feed["multiple_target_output"] = Yr
            feed["target_weight"] = Wr
            margin = dict(targets=tg, Y=Yr, weight=Wr, balance=float(args.balance), item_popularity=item_pop, n_users=60,
                          popularity_based=int(bool(args.pb)), min_access=float(args.min_access))
        # pass 1: the reference's own initialisers -> names and shapes in get_all_params order
        E.new_network(feed)
        p._prepare_networks(N)
        layout = [(q.pname, tuple(q.shape)) for q in lasagne.layers.get_all_params(p.l_out)]
        assert [s for _, s in layout] == [q.shape for q in params], (layout, [q.shape for q in params])
        assert len(lasagne.layers.get_all_params(p.l_out, trainable=True)) == len(layout)      # learn_init=True: all trained
        # pass 2: our seeded values (creation order == get_all_params order, checked by the shapes above and the names below)
        E.new_network(feed, params)
        p._prepare_networks(N)
        assert E.leftovers() == 0
        all_params = lasagne.layers.get_all_params(p.l_out, trainable=True)        # rnn_base.py:182
        assert [(q.pname, tuple(q.shape)) for q in all_params] == layout
        cost = p.cost
        grads = theano.grad(cost, all_params)
        clip_changes = 0.0
        if name in POPSCALE:                             # the same network without the clip: it has to make a difference
            real_clip = theano.gradient.grad_clip
            theano.gradient.grad_clip = lambda x, lo, hi: x
            try:
                E.new_network(feed, params)
                _, p2 = predictor()
                p2._prepare_networks(N)
                free = theano.grad(p2.cost, lasagne.layers.get_all_params(p2.l_out, trainable=True))
            finally:
                theano.gradient.grad_clip = real_clip
            assert abs(float(p2.cost) - float(cost)) < 1e-12 * abs(float(cost))
            clip_changes = max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(grads, free))
            assert clip_changes > 0.05, clip_changes
        h_last = lasagne.layers.get_output(p.l_out.input_layer)
        det = lasagne.layers.get_output(p.l_out, deterministic=True)               # predict_function, rnn_base.py:192
        test_scores = det if (loss == "CCE" or loss in MARGIN) else theano.tensor.nnet.softmax(det)    # test function, rnn_base.py:200 / rnn_sampling.py:144
        if p.interactions_are_unique:
            test_scores = test_scores * (1 - theano.tensor.fmatrix("excluded_items"))   # rnn_base.py:201-202
        out = dict(cell=cell, layers=np.array([H]), loss=loss, N=N, B=B, T=T, S=S, F=F, n_opt=n_opt, bidirectional=int(bi),
                   regularization=float(args.regularization), grad_clip=float(args.gradient_clipping) if hasattr(args, "gradient_clipping") else 100.0,
                   X=batch["X"], mask=batch["mask"], target=batch["target"], samples=batch["samples"], pop=batch["pop"],
                   cost=float(cost), h_last=h_last.detach().numpy(), scores=det.detach().numpy(),
                   test_scores=test_scores.detach().numpy(), n_params=len(params), clip_changes=clip_changes,
                   unique=int(p.interactions_are_unique),
                   names=np.array([n for n, _ in layout]), model_file=p._get_model_filename(1.0), **margin)
        for i, (q, g) in enumerate(zip(params, grads)):
            out["p%d" % i] = q.astype(np.float32)
            out["g%d" % i] = g.detach().numpy()
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
        print("%-20s cost %.6f  |g|max %.3e  %d params  clip changes grads by %.2f" % (
            name, float(cost), max(float(g.abs().max()) for g in grads), len(params), clip_changes))

    for name, (cell, H, loss, N, B, T, S, cl, extra) in CLUSTER_CASES.items():
        seed = sum(map(ord, name))
        params, cfg, batch = PU.build_case(cell, [H], loss, N, B, T, S=S, seed=seed, clusters=cl, scale=0.1 if H >= 128 else None)
        ncs = cfg["clusters"]["c_sampling"]
        sys.argv = ["train.py", "-d", "/tmp/x/", "-b", str(B), "--max_length", str(T), "--r_t", cell, "--r_l", str(H), "--loss", loss,
                    "--sampling", str(S), "--clusters", str(cl["n"]), "--cluster_type", cl["type"], "--init_scale", str(cl["scale"])] + \
                   (["--c_sampling", str(ncs)] if ncs else []) + extra
        args = cp.command_parser(cp.predictor_command_parser, reftrain.training_command_parser, cp.early_stopping_command_parser)
        p = cp.get_predictor(args)
        assert type(p).__name__ == "RNNCluster"
        exclude = np.zeros((B, N))
        for b in range(B):
            exclude[b, batch["X"][b, :int(batch["mask"][b].sum()), 0]] = 1
        csm = batch["cluster_samples"] if ncs else batch["samples"]
        feed = dict(inputs=[batch["X"], batch["mask"].astype(np.float64)], target_output=batch["target"], samples=batch["samples"],
                    cluster_samples=csm, excluded_items=exclude)
        net, R0, Wc0 = params[:-2], params[-2], params[-1]
        E.new_network(feed, net + [Wc0])                 # creation order: the net, out.W, out.b, then the selection layer's W
        p._create_ini_clusters = lambda: R0               # (the reference draws 0.1 * randn here, rnn_cluster.py:182)
        p._prepare_networks(N)
        assert E.leftovers() == 0
        all_params = lasagne.layers.get_all_params(p.l_out, trainable=True)        # rnn_cluster.py:278
        sel = p.cluster_selection_layer.get_params(trainable=True)                 # :282-283
        assert len(sel) == 1 and tuple(sel[0].shape) == Wc0.shape                   # b=None: a selection layer without bias
        assert [tuple(q.shape) for q in all_params] == [q.shape for q in net]
        grads = theano.grad(p.cost, all_params)
        gWc, gR = theano.grad(p.cost_clusters, sel + [p.cluster_repartition])
        h_last = lasagne.layers.get_output(p.user_representation_layer, deterministic=True)
        z = lasagne.layers.get_output(p.cluster_selection_layer, deterministic=True)
        hard = p._get_hard_clusters()
        s1 = theano.tensor.nnet.softmax(lasagne.layers.get_output(p.l_out, deterministic=True))      # :332
        used = hard[:, z.argmax(dim=1)].T                                                           # :334-336, for every row
        s2 = s1 * used
        if p.interactions_are_unique:
            s1 = s1 * (1 - theano.tensor.fmatrix("excluded_items")); s2 = s2 * (1 - theano.tensor.fmatrix("excluded_items"))
        out = dict(cell=cell, layers=np.array([H]), loss=loss, N=N, B=B, T=T, S=S, n_clusters=cl["n"], cluster_type=cl["type"],
                   scale=float(cl["scale"]), c_sampling=ncs, X=batch["X"], mask=batch["mask"], target=batch["target"],
                   samples=batch["samples"], cluster_samples=csm, cost=float(p.cost), cost_clusters=float(p.cost_clusters),
                   h_last=h_last.detach().numpy(), selection=z.detach().numpy(), hard=hard.detach().numpy(),
                   test_scores=s1.detach().numpy(), test_scores_clusters=s2.detach().numpy(), unique=int(p.interactions_are_unique),
                   n_params=len(params), names=np.array([q.pname for q in all_params] + ["cluster_repartition", "cluster_selection.W"]),
                   model_file=p._get_model_filename(1.0))
        for i, (q, g) in enumerate(zip(params, list(grads) + [gR, gWc])):
            out["p%d" % i] = q.astype(np.float32)
            out["g%d" % i] = g.detach().numpy()
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **out)
        print("%-28s cost %.6f  cost_clusters %.6f  |g|max %.3e  %d params" % (
            name, float(p.cost), float(p.cost_clusters), max(float(g.abs().max()) for g in grads), len(params)))


if __name__ == "__main__":
    main()
