"""Row-sparse optimizer steps (csrc/sbr_sparse.hip) against the DENSE float64 oracle (lasagne.updates.* semantics,
update_manager.py:24-82): runs of >= 6 steps whose batches touch different item rows, so that rows sit out steps and
are caught up later -- adagrad exactly, rmsprop / adadelta / nesterov / adam "lazy-exact".  Parameters are compared
after the run (sbr_get_params flushes), costs at every step (a stale row read by a forward pass would show there), and
predict / top-k in the middle of a run (they flush what they read).  The same runs with the dense kernel
(SBR_FLAG_DENSE_UPDATE) must agree with the sparse ones far inside the oracle tolerance."""
import numpy as np
import pytest

import parity_util as PU
from oracle import rnn_oracle as O

pytestmark = pytest.mark.gpu

SPARSE, DENSE = 32, 64
UPDATERS = ["adagrad", "adadelta", "rmsprop", "nesterov", "adam"]


def run_sequence(cell, layers, loss, N, B, T, S, updater, plan, flags, emb=0, bi=False, F=1, n_opt=0, seed=0, oracle=True,
                 probe_at=None, want_sections=False):
    """plan: list of batch seeds, one per step (equal seeds = the same batch again).  Returns engine costs / params /
    probe scores (+ the oracle's)."""
    params, cfg, _ = PU.build_case(cell, layers, loss, N, B, T, S=S, seed=seed, F=F, n_opt=n_opt, emb=emb, bi=bi)
    batches = {}
    for sd in set(plan):
        rng = np.random.default_rng(1000 + sd)
        bt = PU.make_batch(rng, B, T, N, S=S, F=F, n_in0=N + n_opt)
        if sd % 2:      # odd seeds draw from the upper half of the catalogue only: disjoint row sets
            bt["X"][:, :, 0] = np.where(bt["mask"] > 0, N // 2 + bt["X"][:, :, 0] % (N - N // 2), 0)
            bt["target"] = (N // 2 + bt["target"] % (N - N // 2)).astype(np.int32)
            bt["samples"] = (N // 2 + bt["samples"] % (N - N // 2)).astype(np.int32)
        else:
            bt["X"][:, :, 0] = np.where(bt["mask"] > 0, bt["X"][:, :, 0] % (N // 2), 0)
            bt["target"] = (bt["target"] % (N // 2)).astype(np.int32)
            bt["samples"] = (bt["samples"] % (N // 2)).astype(np.int32)
        batches[sd] = bt
    eng = PU.engine_for(cfg, N, B, T, S=S, F=F, n_opt=n_opt, updater=updater, flags=flags)
    out = {}
    try:
        out["sparse_blocks"] = eng.query("sparse_blocks")
        eng.set_all_param_values(params)
        costs, probe = [], None
        for i, sd in enumerate(plan):
            bt = batches[sd]
            eng.set_batch(bt["X"], bt["mask"], bt["target"], bt["samples"] if loss != "CCE" else None, bt["pop"])
            costs.append(eng.train_step(sync=True))
            if probe_at is not None and i == probe_at:
                pb = batches[plan[0]]
                probe = (eng.predict_function(pb["X"], pb["mask"]), eng.test_function((pb["X"], pb["mask"]), k=3))
        if want_sections:      # the flat arena sections in the engine's own layout (sbr_section brings lazy rows up to date)
            out["section_state"] = eng.section("state")[0].cpu().numpy().copy()
            out["section_params"] = eng.section("params")[0].cpu().numpy().copy()
        out.update(costs=np.array(costs), params=eng.get_all_param_values(), probe=probe)
    finally:
        eng.close()
    if oracle:
        upd = O.Updater(updater, 0.01, rho=0.9, beta1=0.9, beta2=0.999)
        op = [p.copy() for p in params]
        ocosts, oprobe = [], None
        for i, sd in enumerate(plan):
            ocosts.append(O.train_function(op, cfg, upd, PU.oracle_batch(batches[sd])))
            if probe_at is not None and i == probe_at:
                pb = batches[plan[0]]
                oprobe = O.predict_scores(op, cfg, pb["X"], pb["mask"])[0]
        out.update(ocosts=np.array(ocosts), oparams=op, oprobe=oprobe)
    return out


def assert_matches_oracle(r, tol_p=2e-4, tol_c=2e-5, tol_probe=1e-3):
    assert np.all(np.abs(r["costs"] - r["ocosts"]) <= tol_c * np.abs(r["ocosts"])), (r["costs"], r["ocosts"])
    worst = max(PU.rel_err(a, b) for a, b in zip(r["params"], r["oparams"]))
    assert worst <= tol_p, worst
    if r["probe"] is not None:
        assert PU.rel_err(r["probe"][0], r["oprobe"]) <= tol_probe


PLAN = [0, 1, 0, 0, 1, 2, 1, 0]        # rows of the lower / upper half of the catalogue alternate; 8 steps


@pytest.mark.parametrize("updater", UPDATERS)
@pytest.mark.parametrize("loss", ["CCE", "BPR"])
def test_sparse_steps_match_the_dense_oracle(updater, loss):
    r = run_sequence("GRU", [16], loss, N=60, B=6, T=7, S=5, updater=updater, plan=PLAN, flags=SPARSE, probe_at=4)
    assert r["sparse_blocks"] == (1 if loss == "CCE" else 2)
    assert_matches_oracle(r)


@pytest.mark.parametrize("updater", UPDATERS)
def test_sparse_and_dense_kernels_agree(updater):
    a = run_sequence("LSTM", [20], "Blackout", N=60, B=6, T=7, S=5, updater=updater, plan=PLAN, flags=SPARSE, oracle=False)
    b = run_sequence("LSTM", [20], "Blackout", N=60, B=6, T=7, S=5, updater=updater, plan=PLAN, flags=DENSE, oracle=False)
    assert a["sparse_blocks"] == 2 and b["sparse_blocks"] == 0
    assert np.allclose(a["costs"], b["costs"], rtol=5e-6)
    worst = max(PU.rel_err(x, y) for x, y in zip(a["params"], b["params"]))
    assert worst <= 5e-6, worst


@pytest.mark.parametrize("updater", ["adadelta", "rmsprop", "nesterov", "adam"])
def test_long_gaps_take_the_closed_forms(updater):
    # rows of the lower half sit out 44 steps (> 32: pow / geometric-sum forms; adam: early exit of the replay) and return.
    # 48 float32 steps drift from the float64 oracle by themselves (rmsprop divides by sqrt of a decayed accumulator), so
    # the sharp comparison is with the DENSE float32 kernel on the same run; the oracle bounds both
    plan = [0, 0] + [1] * 44 + [0, 1]
    kw = dict(N=40, B=4, T=5, S=4, updater=updater, plan=plan)
    if updater == "rmsprop":
        # rmsprop turns every gradient into a step of ~lr whatever its size, so one-ulp differences between the sparse and the
        # dense kernels' arithmetic (powf against 44 multiplications) flip noise-level gradients and the two runs drift apart
        # by 1e-2 over 48 steps (two executions of the SAME kernels differ in the fourth digit through the order of their float
        # atomics).  The closed form itself is therefore checked where that sensitivity cannot reach: on the rows that sat
        # out the gap -- their gradient is exactly zero meanwhile, so parameter and accumulator depend on the first two steps
        # only -- read right after the gap (sbr_section flushes).  The whole run is then only bounded loosely.
        gap = dict(kw, plan=plan[:-2])
        r = run_sequence("GRU", [8], "TOP1", flags=SPARSE, oracle=False, want_sections=True, **gap)
        d = run_sequence("GRU", [8], "TOP1", flags=DENSE, oracle=False, want_sections=True, **gap)
        n = (kw["N"] // 2) * 3 * 16                        # rows [0, N/2) of layer 0's W_in: item-major, 3 gates x Hp = 16
        for name in ("section_params", "section_state"):
            a, b = r[name][:n], d[name][:n]
            assert np.abs(b).max() > 0
            assert np.allclose(a, b, rtol=1e-4, atol=1e-12), (name, np.abs(a - b).max(), np.abs(b).max())
        r = run_sequence("GRU", [8], "TOP1", flags=SPARSE, probe_at=30, **kw)
        assert_matches_oracle(r, tol_p=5e-2, tol_c=5e-2, tol_probe=5e-2)      # (the scores in the middle of the run drift with it)
        return
    r = run_sequence("GRU", [8], "TOP1", flags=SPARSE, probe_at=30, **kw)
    d = run_sequence("GRU", [8], "TOP1", flags=DENSE, oracle=False, **kw)
    assert np.allclose(r["costs"], d["costs"], rtol=1e-4), np.abs(r["costs"] / d["costs"] - 1).max()
    worst = max(PU.rel_err(x, y) for x, y in zip(r["params"], d["params"]))
    assert worst <= 2e-4, worst
    assert_matches_oracle(r, tol_p=2e-3, tol_c=2e-3)


def test_sparse_steps_with_embedding_bidirectional_and_two_indices():
    r = run_sequence("GRU", [12], "CCE", N=50, B=5, T=6, S=0, updater="adam", plan=PLAN, flags=SPARSE, emb=6, probe_at=3)
    assert r["sparse_blocks"] == 1
    assert_matches_oracle(r)
    r = run_sequence("LSTM", [10], "BPR", N=50, B=5, T=6, S=4, updater="nesterov", plan=PLAN, flags=SPARSE, bi=True)
    assert_matches_oracle(r)
    r = run_sequence("GRU", [10], "CCE", N=40, B=5, T=6, S=0, updater="rmsprop", plan=PLAN, flags=SPARSE, F=2, n_opt=10)
    assert_matches_oracle(r)


def test_wide_rows_and_many_items():
    # 256-wide LSTM rows (1024 floats: four passes of a wave) over 5000 items, Zipf ids: C3's kernels at a size the oracle
    # steps six times in seconds; the default selection takes the sparse path here without the flag
    N, B, T, S = 5000, 16, 12, 8
    params, cfg, _ = PU.build_case("LSTM", [256], "Blackout", N, B, T, S=S, seed=5, scale=0.03)
    eng = PU.engine_for(cfg, N, B, T, S=S, updater="adam")
    upd = O.Updater("adam", 0.01, rho=0.9, beta1=0.9, beta2=0.999)
    op = [p.copy() for p in params]
    try:
        assert eng.query("sparse_blocks") == 2 and eng.query("adam_table") > 1000
        eng.set_all_param_values(params)
        for i in range(6):
            bt = PU.make_batch(np.random.default_rng(50 + i % 3), B, T, N, S=S, zipf=True)
            eng.set_batch(bt["X"], bt["mask"], bt["target"], bt["samples"], bt["pop"])
            c = eng.train_step(sync=True)
            oc = O.train_function(op, cfg, upd, PU.oracle_batch(bt))
            assert abs(c - oc) <= 2e-5 * abs(oc), (i, c, oc)
        worst = max(PU.rel_err(a, b) for a, b in zip(eng.get_all_param_values(), op))
        assert worst <= 1e-3, worst          # the bar of every multi-step comparison (test_gpu_parity.check)
    finally:
        eng.close()


def test_default_selection_by_shape():
    from sbr_amd.engine import RNNEngine
    cases = [(dict(cell="GRU", layers=[128], n_items=3706, loss="CCE"), 0),                              # C2: every row can be touched
             (dict(cell="LSTM", layers=[256], n_items=100000, loss="Blackout", n_samples=32), 2),        # C3
             (dict(cell="LSTM", layers=[256], n_items=26744, loss="CCE"), 0),                            # C4 on one GPU
             (dict(cell="LSTM", layers=[256], n_items=26744, loss="CCE", local_batch=32), 1)]            # C4's per-GPU share of 8
    for kw, want in cases:
        eng = RNNEngine(max_length=200, batch_size=256, **kw)
        try:
            assert eng.query("sparse_blocks") == want, kw
            rs = eng.dense_ranges()
            assert rs[-1][1] == eng.section("grads")[0].numel() and all(a[1] <= b[0] for a, b in zip(rs, rs[1:]))
        finally:
            eng.close()
