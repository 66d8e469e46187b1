"""Round 6: the rebuilt 16-row cluster chains (csrc/sbr_rec_c16.hip) beside a FOREIGN kernel that holds part of the chip
(sbr_debug_occupy), and the one-launch head in the same situation.  VERDICT round 5, items 7 / 14: a grid that is not co-resident must
end in the fault code (cluster chains: their members wait for each other inside the kernel) or in the recompute path (head) -- never
in a wrong gradient.  The parity of the kernels themselves is every wide-layer test of tests/test_gpu_parity.py,
tests/test_gpu_config_parity.py (C3 / C4 / C5 at full length) and tests/test_reference_layers.py."""
import numpy as np
import pytest

import parity_util as PU
from oracle import rnn_oracle as O

pytestmark = pytest.mark.gpu


def _engine(cell, H, N, B, T, seed):
    from sbr_amd.engine import RNNEngine, SbrError
    eng = RNNEngine(cell=cell, layers=[H], n_items=N, max_length=T, batch_size=B, loss="CCE")
    params = O.init_params(cell, [H], N, np.random.default_rng(seed), dtype=np.float32)
    batch = PU.make_batch(np.random.default_rng(seed + 1), B, T, N)
    return eng, params, batch, SbrError


def test_cluster_chain_that_cannot_be_resident_fails_loudly():
    """LSTM-512 at 256 rows: 512 workgroups of 32-member clusters, two per CU.  A foreign kernel holds the LDS of 192 of the 256 CUs
    for three seconds, so only 128 workgroups -- HALF of the members of the first eight clusters -- get onto the chip; they poll for
    members that cannot start.  The bounded polls give up (~0.5 s), the step returns the library's error, and after the foreign
    kernel has gone the same engine computes the same cost as before."""
    eng, params, batch, SbrError = _engine("LSTM", 512, 500, 256, 12, 5)
    try:
        eng.set_all_param_values(params)
        eng.set_batch(batch["X"], batch["mask"], batch["target"], None, batch["pop"])
        c0 = eng.forward_backward()
        assert eng.query("rec_rows_fwd") == 16 and np.isfinite(c0)
        eng.debug_occupy(192, 160, 3000)
        with pytest.raises(SbrError, match="bounded wait"):
            eng.forward_backward()
        eng.debug_occupy(0, 0, 0)                              # wait for the foreign kernel
        eng.set_all_param_values(params)
        c1 = eng.forward_backward()
        assert abs(c1 - c0) <= 1e-6 * abs(c0), (c0, c1)
    finally:
        eng.debug_occupy(0, 0, 0)
        eng.close()


def test_one_launch_head_beside_a_foreign_kernel_recomputes():
    """C2's head: 256 workgroups of 127 KB LDS, one per CU, that exchange chunk statistics inside the launch.  With 64 CUs held by a
    foreign kernel a quarter of them start only when others have left: the resident ones stop waiting after 60 us and recompute the
    missing chunks' statistics themselves.  Same cost, same gradients as on the idle chip."""
    eng, params, batch, _ = _engine("GRU", 128, 3706, 256, 16, 9)
    try:
        eng.set_all_param_values(params)
        eng.set_batch(batch["X"], batch["mask"], batch["target"], None, batch["pop"])
        assert eng.query("head_fused") == 16
        c0 = eng.forward_backward()
        g0 = [g.copy() for g in eng.get_all_grad_values()]
        eng.debug_occupy(64, 160, 300)
        c1 = eng.forward_backward()
        g1 = eng.get_all_grad_values()
        eng.debug_occupy(0, 0, 0)
        assert abs(c1 - c0) <= 1e-6 * abs(c0), (c0, c1)
        for a, b in zip(g0, g1):
            assert PU.rel_err(b, a) <= 1e-5
    finally:
        eng.debug_occupy(0, 0, 0)
        eng.close()


def test_batches_built_beside_the_step_in_flight_are_the_batches_the_steps_read():
    """sbr_build_batch packs batch i+1 on a stream of its own, into the second batch set, while step i runs.  Engine A queues
    build + lagged step back to back (nothing synchronises between them), with a ranking through sbr_set_batch thrown in every
    few steps and a build that no step reads; engine B builds each batch, copies it to the host, hands it back through
    sbr_set_batch and steps synchronously.  Same costs, same parameters: a batch overwritten while a step still read it, or read
    before it was complete, shows up as a different cost at once (the rows of two batches share nothing)."""
    from sbr_amd.engine import RNNEngine, DeviceDataset
    rng = np.random.default_rng(11)
    n_users, N, B, T, H, n = 900, 3706, 256, 30, 128, 14
    lengths = rng.integers(3, 70, size=n_users)
    offsets = np.concatenate([[0], np.cumsum(lengths)]).astype(np.int64)
    items = rng.integers(1, N, size=int(offsets[-1])).astype(np.int32)
    params = O.init_params("GRU", [H], N, np.random.default_rng(2), dtype=np.float32)
    probe = PU.make_batch(np.random.default_rng(3), B, T, N)

    def run(overlapped):
        eng = RNNEngine(cell="GRU", layers=[H], n_items=N, max_length=T, batch_size=B, loss="CCE")
        ds = DeviceDataset(eng, items, offsets, N)
        try:
            eng.set_all_param_values(params)
            nb = ds.plan_pass(None, B)
            assert nb >= 4
            costs, ranks = [], []
            for i in range(n):
                if i % 5 == 3:      # evaluation between two steps: another batch through the first set, no training step
                    if overlapped and i == 8:
                        eng.build_batch(ds, (i + 2) % nb, seed=999)      # built and never stepped
                    ranks.append(eng.predict_function(probe["X"], probe["mask"])[:4, :16].copy())
                eng.build_batch(ds, i % nb, seed=70 + i)
                if overlapped:
                    c = eng.train_step_lagged()
                    if c is not None:
                        costs.append(c)
                else:
                    cur = eng.current_batch()
                    eng.set_batch(cur["X"], None, cur["target"], None, cur["pop"], lengths=cur["lengths"])
                    costs.append(eng.train_step())
            if overlapped:
                costs.append(eng.flush_lagged())
            return np.array(costs), np.array(ranks), eng.get_all_param_values()
        finally:
            ds.close(); eng.close()

    c0, r0, p0 = run(False)
    c1, r1, p1 = run(True)
    assert len(c0) == len(c1) == n
    assert np.allclose(c0, c1, rtol=1e-5, atol=0), (c0, c1)
    assert np.allclose(r0, r1, rtol=1e-4, atol=1e-6)
    assert all(np.allclose(a, b, rtol=1e-4, atol=1e-6) for a, b in zip(p0, p1))
    assert np.ptp(c0) > 1e-3                   # (the steps did something)


@pytest.mark.parametrize("loss", ["Blackout", "BPR", "TOP1"])
@pytest.mark.parametrize("H", [128, 256])
def test_one_launch_sampled_head_against_the_oracle_and_the_four_launches(loss, H, monkeypatch):
    """head_sampled_kernel (activations, sampled loss, its gradient and dh in one launch: full 16-row blocks, Hp in {128, 256, 512},
    at most 320 cells) against the float64 oracle through the usual step comparison, and against the same engine on the four launches
    (SBR_HEAD_FUSE=0): costs and every gradient agree to f32 rounding -- both are f32-class, the one launch on exact-f32 products."""
    from sbr_amd.engine import RNNEngine
    r = PU.compare_step("LSTM", [H], loss, N=900, B=32, T=9, S=24, scale=0.08)
    assert r["cost"] <= 2e-6 and r["grad_worst"] <= 5e-6, r
    params = O.init_params("LSTM", [H], 900, np.random.default_rng(4), dtype=np.float32)
    batch = PU.make_batch(np.random.default_rng(5), 32, 9, 900)
    samples = np.random.default_rng(6).integers(1, 900, size=24).astype(np.int32)
    out = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("SBR_HEAD_FUSE", fuse)
        eng = RNNEngine(cell="LSTM", layers=[H], n_items=900, max_length=9, batch_size=32, loss=loss, n_samples=24)
        try:
            eng.set_all_param_values(params)
            eng.set_batch(batch["X"], batch["mask"], batch["target"], samples, batch["pop"])
            c = eng.forward_backward()
            out.append((c, [g.copy() for g in eng.get_all_grad_values()]))
        finally:
            eng.close()
    (c1, g1), (c0, g0) = out
    assert abs(c1 - c0) <= 2e-6 * abs(c0), (c1, c0)
    for a, b in zip(g1, g0):
        assert np.abs(a - b).max() <= 5e-6 * max(1e-30, np.abs(b).max()), (np.abs(a - b).max(), np.abs(b).max())
