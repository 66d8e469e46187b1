"""The data-parallel step with the REAL engine and world_size 2: two processes share the one GPU of the test box,
each holds half of the global batch in its own RNNEngine (local_batch = B/2, row_offset), gradients travel through
`parallel.DataParallel` (gloo on device tensors: RCCL refuses two ranks on one device) with the deferred-join /
side-stream choreography the 8-GPU run uses.  After two steps both replicas must hold the parameters of the
single-engine step on the whole batch, and the same costs."""
import os
import socket

import numpy as np
import pytest

import parity_util as PU

pytestmark = pytest.mark.gpu

CASES = {
    # name: (cell, layers, loss, N, B, T, S, updater)
    "gru128_cce_adam": ("GRU", [128], "CCE", 300, 64, 40, 0, "adam"),
    # T >= 64 on one 128-unit layer: the overlapped step tail -- one collective behind each of the engine's three streams
    "gru128_cce_adam_overlapped_tail": ("GRU", [128], "CCE", 300, 64, 70, 0, "adam"),
    "lstm20_blackout_adagrad": ("LSTM", [20], "Blackout", 200, 32, 12, 8, "adagrad"),
    "lstm256_bpr_adam": ("LSTM", [256], "BPR", 500, 32, 10, 8, "adam"),
    # row-sparse blocks (forced on these small shapes): all-gather of (ids, rows) instead of the all-reduce of W_in / W_out
    "sparse_gru128_cce_adam": ("GRU", [128], "CCE", 300, 64, 40, 0, "adam"),
    "sparse_lstm20_blackout_adagrad": ("LSTM", [20], "Blackout", 200, 32, 12, 8, "adagrad"),
    "sparse_lstm256_bpr_nesterov": ("LSTM", [256], "BPR", 500, 32, 10, 8, "nesterov"),
    # ... the counted form of the exchange (one read-back per block; what blocks above DataParallel.INBAND_BYTES take): the
    # other sparse cases travel with their row counts in band, nothing synchronised
    "sparse_counted_gru128_cce_adam": ("GRU", [128], "CCE", 300, 64, 40, 0, "adam"),
    # the init-time check of the collectives' stream placement says "not ordered" (forced): every collective on the main stream
    # behind a full join -- same parameters
    "gru128_cce_adam_join_fallback": ("GRU", [128], "CCE", 300, 64, 70, 0, "adam"),
}
SPARSE_FLAG = 32


def _worker(rank, world, port, name, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sbr_amd.parallel import DataParallel
    cell, layers, loss, N, B, T, S, updater = CASES[name]
    params, cfg, batch = PU.build_case(cell, layers, loss, N, B, T, S=S, seed=31, scale=0.05)
    lo, hi = DataParallel.shard(B, world, rank)
    flags = SPARSE_FLAG if name.startswith("sparse") else 0
    eng = PU.engine_for(cfg, N, B, T, S=S, updater=updater, local_batch=hi - lo, row_offset=lo, flags=flags)
    try:
        eng.set_all_param_values(params)
        if "fallback" in name:
            DataParallel._side_collectives_are_ordered = lambda self: False
        dp = DataParallel(eng, dist)
        if "counted" in name:
            dp.INBAND_BYTES = 0
        if "fallback" in name:
            assert dp.side is None and dp.stream_check == "fallback"
        else:
            assert dp.side is not None                      # the stream-level path, not the stand-in one
            assert dp.stream_check == "ordered", dp.stream_check      # ... verified with data on the real streams
            assert (dp.tail is not None) == ("overlapped_tail" in name)
        if flags:
            assert len(eng.sparse_blocks()) == (1 if loss == "CCE" else 2)
        smp = batch["samples"] if loss != "CCE" else None
        if loss != "CCE":      # every rank needs all B targets (Blackout's softmax spans them): all-gather of the local ones
            tgt = dp.gather_targets(torch.from_numpy(batch["target"][lo:hi]).cuda()).cpu().numpy()
            assert np.array_equal(tgt, batch["target"])
        else:
            tgt = batch["target"][lo:hi]
        costs = []
        for _ in range(2):
            eng.set_batch(batch["X"][lo:hi], batch["mask"][lo:hi], tgt, smp, batch["pop"][lo:hi])
            dp.train_step()
            costs.append(eng.read_cost())
        if flags:      # which form of the sparse exchange ran
            assert len(getattr(dp, "_sp_all", {})) == (0 if "counted" in name else len(eng.sparse_blocks()))
        # (the export brings lazily stepped rows up to date: through the collective, on both ranks at once -- see parallel.py)
        np.savez(out % rank, costs=np.array(costs), **{"p%d" % i: p for i, p in enumerate(dp.get_all_param_values())})
    finally:
        eng.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("name", sorted(CASES))
def test_two_ranks_on_one_gpu_equal_the_single_engine_step(tmp_path, name):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, port, name, out), nprocs=2, join=True)
    cell, layers, loss, N, B, T, S, updater = CASES[name]
    params, cfg, batch = PU.build_case(cell, layers, loss, N, B, T, S=S, seed=31, scale=0.05)
    eng = PU.engine_for(cfg, N, B, T, S=S, updater=updater, flags=64 if name.startswith("sparse") else 0)   # reference: the dense step
    try:
        eng.set_all_param_values(params)
        smp = batch["samples"] if loss != "CCE" else None
        eng.set_batch(batch["X"], batch["mask"], batch["target"], smp, batch["pop"])
        ref_costs = [eng.train_step(sync=True) for _ in range(2)]
        ref = eng.get_all_param_values()
    finally:
        eng.close()
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert np.allclose(r0["costs"], ref_costs, rtol=2e-5), (r0["costs"], ref_costs)
    assert np.array_equal(r0["costs"], r1["costs"])
    for i, p in enumerate(ref):
        assert np.array_equal(r0["p%d" % i], r1["p%d" % i]), i          # replicas stay bit-identical
        assert PU.rel_err(r0["p%d" % i], p) <= 2e-5, (i, PU.rel_err(r0["p%d" % i], p))



def _worker_midrun(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sbr_amd.parallel import DataParallel
    cell, layers, loss, N, B, T, S, updater = "GRU", [128], "CCE", 300, 64, 40, 0, "adam"
    params, cfg, _ = PU.build_case(cell, layers, loss, N, B, T, S=S, seed=31, scale=0.05)
    lo, hi = DataParallel.shard(B, world, rank)
    eng = PU.engine_for(cfg, N, B, T, S=S, updater=updater, local_batch=hi - lo, row_offset=lo, flags=SPARSE_FLAG)
    try:
        eng.set_all_param_values(params)
        dp = DataParallel(eng, dist)
        assert eng.dp_guard
        # six steps on batches that touch different rows (so rows go untouched for several steps: lazy Adam replays), a ranking
        # in the middle: through the collective on both ranks
        raised = False
        for step in range(6):
            _, _, batch = PU.build_case(cell, layers, loss, N, B, T, S=S, seed=100 + step, scale=0.05)
            eng.set_batch(batch["X"][lo:hi], batch["mask"][lo:hi], batch["target"][lo:hi], None, batch["pop"][lo:hi])
            dp.train_step()
            if step == 2:
                if rank == 0:
                    try:
                        eng.test_function((batch["X"][lo:hi], batch["mask"][lo:hi]), k=5)      # alone: refused
                    except RuntimeError:
                        raised = True
                ids = dp.test_function((batch["X"][lo:hi], batch["mask"][lo:hi]), k=5)          # together
                assert ids.shape == (hi - lo, 5)
        assert raised or rank != 0
        np.savez(out % rank, **{"p%d" % i: p for i, p in enumerate(dp.get_all_param_values())})
    finally:
        eng.close()
        dist.destroy_process_group()


def test_midrun_ranking_through_the_collective_keeps_lazily_stepped_replicas_bit_identical(tmp_path):
    """Row-sparse Adam (lazy-exact): a top-k in the middle of a run replays the missed zero-gradient steps of every row.  Made on
    one rank only it would split that rank's replays differently from the others' (float32 roundings: the replicas fork); the
    guarded engine refuses the rank-local call, DataParallel.test_function makes it on both ranks, and after three more steps the
    replicas are bit-identical."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "mid%d.npz")
    mp.spawn(_worker_midrun, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert len(r0.files) > 3
    for k in r0.files:
        assert np.array_equal(r0[k], r1[k]), k
