"""Degenerate shapes against the oracle: T = 1 and 2, a single row, one hidden unit, two items, every kernel family
(single-workgroup bf16x6, cluster 256 / 512), with the embedding and bidirectional options."""
import pytest

import parity_util as PU

pytestmark = pytest.mark.gpu

CASES = [("GRU", [8], 5, 3, 1, {}), ("LSTM", [8], 5, 3, 2, {}), ("GRU", [1], 4, 2, 3, {}), ("LSTM", [3, 2], 2, 1, 2, {}),
         ("Vanilla", [128], 9, 1, 1, {}), ("GRU", [256], 7, 2, 1, {"scale": 0.05}), ("LSTM", [128], 6, 33, 1, {"bi": True}),
         ("GRU", [16], 6, 5, 2, {"emb": 1}), ("LSTM", [512], 5, 3, 2, {"scale": 0.04})]


@pytest.mark.parametrize("case", CASES, ids=["%s%s_N%d_B%d_T%d" % (c[0], "x".join(map(str, c[1])), c[2], c[3], c[4]) for c in CASES])
def test_degenerate_shapes(case):
    cell, layers, N, B, T, kw = case
    r = PU.compare_step(cell, layers, "CCE", N=N, B=B, T=T, **kw)
    assert r["param_roundtrip"] == 0 and r["h_last"] < 2e-4 and r["grad_worst"] < 2e-4 and r["topk_mismatch"] == 0, r


def test_phase_timing_modes():
    # sbr_enable_timing: every phase, or only the two events around one phase (what bench.py keeps in its timed region)
    import numpy as np
    params, cfg, batch = PU.build_case("GRU", [16], "CCE", 30, 6, 8)
    eng = PU.engine_for(cfg, 30, 6, 8)
    try:
        eng.set_all_param_values(params)
        eng.set_batch(batch["X"], batch["mask"], batch["target"], None, batch["pop"])
        eng.enable_timing(True)
        for _ in range(3):
            eng.train_step(sync=True)
        full = eng.phase_times()
        assert all(full[k] > 0 for k in ("rec_fwd", "output", "rec_bwd", "update"))
        assert full["total"] == pytest.approx(sum(v for k, v in full.items() if k != "total"), rel=1e-5)
        eng.enable_timing(True, only="rec_bwd")
        for _ in range(3):
            eng.train_step(sync=True)
        one = eng.phase_times()
        assert one["rec_bwd"] > 0 and one["total"] == pytest.approx(one["rec_bwd"], rel=1e-5)
        assert all(one[k] == 0 for k in ("gather", "rec_fwd", "output", "wgrad", "scatter", "update"))
        assert 0.3 < one["rec_bwd"] / full["rec_bwd"] < 3.0
        eng.enable_timing(False)
        eng.train_step(sync=True)
        with pytest.raises(RuntimeError):
            eng.phase_times()
    finally:
        eng.close()


def test_topk_with_fewer_rankable_items_than_k():
    # a user who has seen more than N - k items (exclude_seen): the places that cannot be filled carry -1, never an
    # excluded or repeated id; NaN scores rank nowhere
    import numpy as np
    N, B, T = 12, 3, 10
    params, cfg, batch = PU.build_case("GRU", [8], "CCE", N, B, T, seed=2)
    eng = PU.engine_for(cfg, N, B, T)
    try:
        eng.set_all_param_values(params)
        X = np.zeros((B, T, 1), np.int32); mask = np.zeros((B, T), np.float32)
        X[0, :10, 0] = np.arange(10); mask[0, :10] = 1          # 2 unseen items
        X[1, :3, 0] = [4, 4, 5]; mask[1, :3] = 1                 # 10 unseen
        X[2, :1, 0] = [0]; mask[2, :1] = 1
        ids = eng.test_function((X, mask), k=5)
        assert sorted(ids[0][:2]) == [10, 11] and list(ids[0][2:]) == [-1, -1, -1]
        for b in (1, 2):
            seen = set(X[b, :int(mask[b].sum()), 0])
            assert len(set(ids[b])) == 5 and ids[b].min() >= 0 and not set(ids[b]) & seen
        params[-1][3] = np.nan                                   # a NaN bias -> NaN score of item 3 in every row
        eng.set_all_param_values(params)
        ids = eng.test_function((X, mask), k=5, exclude_seen=False)
        assert not (ids == 3).any() and ids.min() >= 0
    finally:
        eng.close()
