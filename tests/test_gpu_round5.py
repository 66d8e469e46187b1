"""Round 5 kernels, each against the float64 oracle AND against the form it replaces (the switches involved are read once per
engine, in sbr_create, so a pair is two engines of this process built under different environments):
  * the one-launch full-softmax head (csrc/sbr_head.hip: logits + softmax / CCE + dh, rnn_one_hot.py:65-71) -- chunk counts
    that leave whole chunks empty, catalogues that are no multiple of 16, every layer width it serves, one and sixteen row blocks;
  * the scatter-add that steps the rows it completes (launch_scatter_wide_step, sparse_lstm.py:368 + update_manager.py:24-82)
    -- row-sparse blocks with every updater over runs in which rows sit out steps, and a dense block with the three placements
    of the zero-gradient pass over the untouched rows;
  * the sampled head's row-sparse block caught up beside the forward chain and stepped beside the BPTT chain;
  *  one kernel family for the forward and the backward launch of a step when the catalogue is too large for the
    fused gather's 32-bit row offsets."""
import os

import numpy as np
import pytest

import parity_util as PU
from oracle import rnn_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bars(r, tol=1e-5):
    assert r["h_last"] <= tol and r["cost"] <= tol and r["grad_worst"] <= tol, r
    assert r["params_twin"] <= 2e-5 and r["topk_mismatch"] == 0, r
    PU.params_ok(r, steps=2, tol_g=tol)


# ----------------------------------------------------------------------------------------------------------------
# the one-launch head
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cell,H,N,B,T", [("GRU", 128, 3706, 256, 6),      # C2's head: 16 row blocks x 16 chunks of 240 columns
                                          ("GRU", 128, 50, 16, 5),         # 16 chunks of 16 columns, 12 of them empty
                                          ("LSTM", 20, 333, 32, 5),        # Hp = 32, a catalogue that is no multiple of 16
                                          ("GRU", 50, 1001, 64, 5),        # Hp = 64
                                          ("Vanilla", 128, 4850, 256, 4),  # the widest chunk that still fits (304 columns: 160 KB of LDS)
                                          ("GRU", 100, 77, 48, 5)])        # three row blocks, Hp = 128 with padded units
def test_one_launch_head_against_the_oracle(cell, H, N, B, T):
    r = PU.compare_step(cell, [H], "CCE", N=N, B=B, T=T, steps=2, seed=71, zipf=N > 1000, scale=0.05 if cell == "Vanilla" else None,
                        queries=("head_fused",))
    assert r["q:head_fused"] > 0, r
    bars(r)


def test_one_launch_head_is_not_taken_where_it_does_not_fit():
    from sbr_amd.engine import RNNEngine
    for kw, want in ((dict(cell="GRU", layers=[128], n_items=3706, batch_size=256), True),
                     (dict(cell="GRU", layers=[128], n_items=4900, batch_size=256), False),      # chunks of 320 columns: 169 KB > LDS
                     (dict(cell="GRU", layers=[128], n_items=3706, batch_size=250), False),      # padded rows
                     (dict(cell="LSTM", layers=[256], n_items=3706, batch_size=256), False),     # Hp = 256
                     (dict(cell="GRU", layers=[128], n_items=3706, batch_size=256, loss="BPR", n_samples=8), False)):
        eng = RNNEngine(max_length=8, **kw)
        try:
            assert (eng.query("head_fused") > 0) == want, kw
        finally:
            eng.close()


_CACHE = {}


def variant(env, cell, H, loss, N, B, T, S=0, upd="adam", steps=4, cache_key=None):
    """One form of the step: the switches of `env` are set while its engines are built (each of them is read per engine in sbr_create:
    SBR_HEAD_FUSE, SBR_SPARSE_OUT_EARLY), then (a) parity_util.compare_step against the oracle and (b) a
    run of `steps` training steps on two alternating batches; returns (compare_step's errors + what sbr_query says the engine
    selected, the parameters and costs of the run as one vector)."""
    if cache_key is not None and cache_key in _CACHE:
        return _CACHE[cache_key]
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        r = PU.compare_step(cell, [H], loss, N=N, B=B, T=T, S=S, zipf=True, steps=2, scale=0.03, seed=61, updater=upd,
                            queries=("head_fused", "sparse_blocks"))
        params, cfg, batch = PU.build_case(cell, [H], loss, N, B, T, S=S, seed=61, scale=0.03, zipf=True)
        eng = PU.engine_for(cfg, N, B, T, S=S, updater=upd)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        eng.set_all_param_values(params)
        costs = []
        for i in range(steps):
            bt = PU.make_batch(np.random.default_rng(300 + i % 2), B, T, N, S=S, zipf=True) if i else batch
            eng.set_batch(bt["X"], bt["mask"], bt["target"], bt["samples"] if loss != "CCE" else None, bt["pop"])
            costs.append(eng.train_step(sync=True))
        vec = np.concatenate([p.ravel() for p in eng.get_all_param_values()] + [np.array(costs, dtype=np.float32)])
    finally:
        eng.close()
    out = ({k: float(v) for k, v in r.items() if not k.startswith(("grad:", "pstep:"))}, vec)
    if cache_key is not None:
        _CACHE[cache_key] = out
    return out


def close(p0, p1, tol=5e-5):
    # Adam moves an element whose gradient is ~0 by ~lr whatever the gradient's size: summation-order roundings of the two forms
    # show at 5e-5 of the largest parameter (the bar of tests/test_gpu_wide_scatter_forms.py)
    assert np.abs(p0 - p1).max() <= tol * np.abs(p0).max(), np.abs(p0 - p1).max() / np.abs(p0).max()


def test_one_launch_head_against_the_three_launches():
    r0, p0 = variant({"SBR_HEAD_FUSE": "0"}, "GRU", 128, "CCE", 3706, 256, 12)
    r1, p1 = variant({"SBR_HEAD_FUSE": "1"}, "GRU", 128, "CCE", 3706, 256, 12)
    assert r0["q:head_fused"] == 0 and r1["q:head_fused"] == 16
    bars(r0); bars(r1)
    close(p0, p1)


@pytest.mark.parametrize("N,B,CC", [(3706, 256, 16), (1500, 512, 8)], ids=["16_chunks", "8_chunks"])
def test_one_launch_head_recompute_path(N, B, CC, monkeypatch):
    """the timeout / recompute path is what keeps the one-launch head correct when its grid is not co-resident (two
    processes on one GPU, a busy side stream).  SBR_HEAD_WAIT_TICKS=0: nobody is waited for, EVERY foreign chunk's statistics are
    recomputed by whoever misses them -- held to the oracle and to the three-launch form, with 16 chunks and (32 row blocks) with 8."""
    monkeypatch.setenv("SBR_HEAD_WAIT_TICKS", "0")           # read per launch: set for the whole test
    r0, p0 = variant({"SBR_HEAD_FUSE": "0"}, "GRU", 128, "CCE", N, B, 12)
    r1, p1 = variant({"SBR_HEAD_FUSE": "1"}, "GRU", 128, "CCE", N, B, 12)
    assert r0["q:head_fused"] == 0 and r1["q:head_fused"] == CC
    bars(r0); bars(r1)
    close(p0, p1)


# ----------------------------------------------------------------------------------------------------------------
# row-sparse blocks of a sampled head: caught up beside the forward chain, stepped beside the BPTT chain
# (round 5 also built a scatter-add that stepped the rows it completed -- correct, slower, removed in round 6: DESIGN.md section 3d)
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("updater", ["adagrad", "adadelta", "rmsprop", "nesterov", "adam"])
def test_sampled_head_rows_stepped_beside_the_chains(updater):
    # LSTM-128 rows (512 floats) over 4000 items with a sampled head: both blocks row-sparse; the head's block caught up beside the
    # forward chain and stepped beside the BPTT chain (the default) against the placement of rounds 2 - 4
    r0, p0 = variant({"SBR_SPARSE_OUT_EARLY": "0"}, "LSTM", 128, "Blackout", 4000, 16, 12, S=8, upd=updater, steps=6)
    r1, p1 = variant({"SBR_SPARSE_OUT_EARLY": "1"}, "LSTM", 128, "Blackout", 4000, 16, 12, S=8, upd=updater, steps=6)
    assert r0["q:sparse_blocks"] == 2 and r1["q:sparse_blocks"] == 2
    bars(r0); bars(r1)
    # rmsprop turns every gradient into a step of ~lr: roundings flip noise-level elements (tests/test_gpu_sparse_update.py)
    close(p0, p1, tol=2e-3 if updater == "rmsprop" else 5e-5)


def test_row_sparse_run_of_two_wide_layers_against_the_dense_oracle():
    # two stacked LSTM-512 layers (C5's kernels: the two-level backward exchange, rows of 2048 floats) over 3000 items, six steps
    # whose batches come from three alternating seeds: rows sit steps out and return.  Costs at every step against the dense oracle;
    # parameters after six Adam steps at the bar of every multi-step comparison
    N, B, T, S = 3000, 16, 10, 8
    params, cfg, _ = PU.build_case("LSTM", [512, 512], "Blackout", N, B, T, S=S, seed=9, scale=0.02)
    upd = O.Updater("adam", 0.01, rho=0.9, beta1=0.9, beta2=0.999)
    op = [p.copy() for p in params]
    ocosts = []
    for i in range(6):
        bt = PU.make_batch(np.random.default_rng(70 + i % 3), B, T, N, S=S, zipf=True)
        ocosts.append(O.train_function(op, cfg, upd, PU.oracle_batch(bt)))
    eng = PU.engine_for(cfg, N, B, T, S=S, updater="adam", flags=32)       # SBR_FLAG_SPARSE_UPDATE
    try:
        assert eng.query("sparse_blocks") == 2
        eng.set_all_param_values(params)
        for i in range(6):
            bt = PU.make_batch(np.random.default_rng(70 + i % 3), B, T, N, S=S, zipf=True)
            eng.set_batch(bt["X"], bt["mask"], bt["target"], bt["samples"], bt["pop"])
            c = eng.train_step(sync=True)
            assert abs(c - ocosts[i]) <= 5e-5 * abs(ocosts[i]), (i, c, ocosts[i])
        got = eng.get_all_param_values()
    finally:
        eng.close()
    worst = max(PU.rel_err(a, b) for a, b in zip(got, op))
    # (measured in round 5: 1.3e-3 against the oracle: six Adam steps on 2 x 512 units amplify summation-order roundings)
    assert worst <= 2e-3, worst


# ----------------------------------------------------------------------------------------------------------------
# forward and backward launch of a step take the same kernel family
# ----------------------------------------------------------------------------------------------------------------
def test_large_catalogue_keeps_one_kernel_family_for_both_directions():
    """LSTM-128 over 2.1 M items: W_in rows lie beyond 2^32 bytes, so rec_fwd_x6p cannot gather them itself (32-bit per-lane
    offsets).  Round 4 then sent the FORWARD launch to the barrier kernels (blocked saved-gate layout) and the backward launch to
    rec_bwd_x6p (16-byte elements): silently wrong gradients.  Now the gather runs as its own kernel and both launches are
    x6p's; the gradients of a step are compared with the same model at a catalogue where the fused gather runs (the ids of the
    batch planted at the far end of the large one)."""
    from sbr_amd.engine import RNNEngine
    NBIG, n, B, T = 2100000, 600, 16, 9
    params, cfg, batch = PU.build_case("LSTM", [128], "BPR", n, B, T, S=8, seed=13, scale=0.05)
    names = [nm for nm, _ in O.model_param_shapes("LSTM", [128], n, n, 0, 1, False)]
    off = NBIG - n                                             # compact id c <-> catalogue id off + c: byte offsets > 2^32

    def widen(nm, p):
        p32 = p.astype(np.float32)
        if nm.startswith("l0.W_in"):
            a = np.zeros((NBIG, p.shape[1]), dtype=np.float32); a[off:] = p32; return a
        if nm == "out.W":
            a = np.zeros((p.shape[0], NBIG), dtype=np.float32); a[:, off:] = p32; return a
        if nm == "out.b":
            a = np.zeros(NBIG, dtype=np.float32); a[off:] = p32; return a
        return p32

    small = PU.engine_for(cfg, n, B, T, S=8, updater="adagrad")
    try:
        assert small.query("fused_gather") == 1 and small.query("rec_kernel") == 2
        small.set_all_param_values(params)
        small.set_batch(batch["X"], batch["mask"], batch["target"], batch["samples"], batch["pop"])
        c0 = small.forward_backward()
        g0 = small.get_all_grad_values()
    finally:
        small.close()
    big = RNNEngine(cell="LSTM", layers=[128], n_items=NBIG, max_length=T, batch_size=B, loss="BPR", n_samples=8, updater="adagrad",
                    learning_rate=0.01)
    try:
        assert big.query("fused_gather") == 0 and big.query("rec_kernel") == 2
        big.set_all_param_values([widen(nm, p) for nm, p in zip(names, params)])
        big.set_batch((batch["X"] + off).astype(np.int32), batch["mask"], (batch["target"] + off).astype(np.int32),
                      (batch["samples"] + off).astype(np.int32), batch["pop"])
        c1 = big.forward_backward()
        g1 = big.get_all_grad_values()
    finally:
        big.close()
    assert abs(c0 - c1) <= 1e-6 * abs(c0)
    for nm, a, b in zip(names, g0, g1):
        if nm.startswith("l0.W_in") or nm == "out.b":
            sub, b[off:] = b[off:].copy(), 0.0
        elif nm == "out.W":
            sub, b[:, off:] = b[:, off:].copy(), 0.0
        else:
            sub = b
        assert PU.rel_err(sub, a, 1e-12) <= 2e-6, (nm, PU.rel_err(sub, a, 1e-12))
        if sub is not b:
            assert not b.any(), nm
