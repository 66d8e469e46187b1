"""world_size-2 data-parallel step on CPU (gloo): parallel.DataParallel drives an oracle-backed
stand-in engine per rank; after the all-reduce both replicas must hold exactly the parameters of
the unsharded single-process step (cost mean over the GLOBAL batch, sampled heads see every rank's
targets, the bias regulariser is added once)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import rnn_oracle as O
import parity_util as PU


class OracleEngine(object):
    """RNNEngine's phase interface on top of the CPU oracle (flat float64 gradient section with the
    cost as trailing element and the output-layer part last, like the arena of libsbr_rnn.so)."""

    def __init__(self, params, cfg, updater, Bglobal, row_offset):
        self.params, self.cfg, self.Bglobal, self.row_offset = params, cfg, Bglobal, row_offset
        self.upd = O.Updater(updater, 0.01, rho=0.9, beta1=0.9, beta2=0.999)
        self.sizes = [p.size for p in params]
        self.flat = torch.zeros(sum(self.sizes) + 1, dtype=torch.float64)
        self.split = sum(self.sizes[:-2])

    def section(self, which):
        assert which == "grads"
        return self.flat, self.split

    def set_batch(self, batch):
        self.batch = dict(batch, Bglobal=self.Bglobal, row_offset=self.row_offset)

    def zero_grads(self):
        self.flat.zero_()

    def forward(self):
        self.cost, self.g, _ = O.cost_and_grads(self.params, self.cfg, self.batch)

    def loss_backward_output(self):
        o = self.split
        for g in self.g[-2:]:
            self.flat[o:o + g.size] = torch.from_numpy(g.reshape(-1)); o += g.size
        self.flat[-1] = self.cost

    def backward_recurrent(self):
        o = 0
        for g in self.g[:-2]:
            self.flat[o:o + g.size] = torch.from_numpy(np.ascontiguousarray(g).reshape(-1)); o += g.size

    def apply_update(self):
        grads, o = [], 0
        for p in self.params:
            grads.append(self.flat[o:o + p.size].numpy().reshape(p.shape).copy()); o += p.size
        self.upd.apply(self.params, grads)

    def read_cost(self):
        return float(self.flat[-1])


class SparseOracleEngine(OracleEngine):
    """The same stand-in with the engine's row-sparse exchange interface (sparse_blocks / sparse_pack /
    sparse_unpack_add / dense_ranges): block 0 = the rows of the layer-0 W_in arrays (one (N, H) array per gate, packed
    side by side), block 1 (sampled heads) = the columns of out.W with out.b.  DataParallel must then all-gather
    (ids, rows) for those and all-reduce only the rest."""

    def __init__(self, params, cfg, updater, Bglobal, row_offset, cap):
        OracleEngine.__init__(self, params, cfg, updater, Bglobal, row_offset)
        offs = np.cumsum([0] + self.sizes)
        G = O.CELL_GATES[cfg["cell"]]
        N, H = params[0].shape
        self.blocks = [[(int(offs[3 * g]), N, H, H, 1) for g in range(G)]]       # (offset, rows, width, row stride, col stride)
        if cfg["loss"] != "CCE":
            HL, No = params[-2].shape
            self.blocks.append([(int(offs[-3]), No, HL, 1, No), (int(offs[-2]), No, 1, 1, 1)])   # out.W is (H, N): a "row" is a column
        self.cap = cap

    def _view(self, blk, ids):
        cols = []
        for off, rows, w, rs, cs in blk:
            idx = off + np.asarray(ids)[:, None] * rs + np.arange(w)[None, :] * cs
            cols.append(idx)
        return np.concatenate(cols, axis=1)

    def sparse_blocks(self):
        return [(blk[0][1], sum(b[2] for b in blk), self.cap) for blk in self.blocks]

    def sparse_pack(self, b):
        blk = self.blocks[b]
        allidx = self._view(blk, np.arange(blk[0][1]))
        g = self.flat.numpy()
        ids = np.nonzero(np.abs(g[allidx]).sum(axis=1) > 0)[0]
        n = len(ids)
        out_ids = torch.zeros(self.cap, dtype=torch.int32); out_rows = torch.zeros((self.cap, allidx.shape[1]), dtype=torch.float64)
        out_ids[:n] = torch.from_numpy(ids.astype(np.int32)); out_rows[:n] = torch.from_numpy(g[allidx[ids]])
        g[allidx[ids].reshape(-1)] = 0.0
        return out_ids, out_rows, n

    def sparse_unpack_add(self, b, ids, rows, count):
        if count:
            idx = self._view(self.blocks[b], ids[:count].numpy())
            self.flat.numpy()[idx] += rows[:count].numpy()

    def dense_ranges(self):
        taken = np.zeros(self.flat.numel(), dtype=bool)
        for blk in self.blocks:
            taken[self._view(blk, np.arange(blk[0][1])).reshape(-1)] = True
        edges = np.flatnonzero(np.diff(np.concatenate([[True], taken, [True]]).astype(np.int8)))
        rs = [(int(a), int(b)) for a, b in zip(edges[::2], edges[1::2])]
        out = []
        for a, b in rs:       # never across the output-layer split
            out += [(a, self.split), (self.split, b)] if a < self.split < b else [(a, b)]
        return out


class InbandSparseOracleEngine(SparseOracleEngine):
    """... and with the ABI-7 form of the exchange: the row count travels in band (ids[0]), buffers at fixed capacity, one
    unpack call for all ranks -- the path DataParallel takes for blocks below its INBAND_BYTES."""

    def sparse_pack_device(self, b):
        ids, rows, n = self.sparse_pack(b)
        return torch.cat([torch.tensor([n], dtype=torch.int32), ids]), rows

    def sparse_unpack_add_all(self, b, ids_all, rows_all, world):
        assert ids_all.shape == (world, self.cap + 1) and rows_all.shape[:2] == (world, self.cap)
        for r in range(world):
            self.sparse_unpack_add(b, ids_all[r, 1:], rows_all[r], int(ids_all[r, 0]))


def _worker(rank, world, port, loss, out, sparse=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sbr_amd.parallel import DataParallel
    B, T, N, S = 6, 5, 17, 4
    params, cfg, batch = PU.build_case("GRU", [6], loss, N, B, T, S=S, seed=11)
    cfg["regularization"] = 0.03 if loss == "CCE" else 0.0
    lo, hi = DataParallel.shard(B, world, rank)
    if sparse == "inband":
        eng = InbandSparseOracleEngine([p.copy() for p in params], cfg, "adam", B, lo, cap=N)
    elif sparse:
        eng = SparseOracleEngine([p.copy() for p in params], cfg, "adam", B, lo, cap=N)
    else:
        eng = OracleEngine([p.copy() for p in params], cfg, "adam", B, lo)
    dp = DataParallel(eng, dist)
    ob = PU.oracle_batch(batch)
    local_t = torch.from_numpy(ob["target"][lo:hi])
    tgt = dp.gather_targets(local_t).numpy() if loss != "CCE" else ob["target"][lo:hi]
    costs = []
    for _ in range(2):
        eng.set_batch(dict(X=ob["X"][lo:hi], mask=ob["mask"][lo:hi], target=tgt, samples=ob["samples"], pop=ob["pop"][lo:hi]))
        dp.train_step()
        costs.append(eng.read_cost())
    np.savez(out % rank, costs=np.array(costs), **{"p%d" % i: p for i, p in enumerate(eng.params)})
    dist.destroy_process_group()


@pytest.mark.parametrize("sparse", [False, True, "inband"], ids=["allreduce", "sparse_exchange", "sparse_exchange_inband_counts"])
@pytest.mark.parametrize("loss", ["CCE", "Blackout", "BPR"])
def test_two_rank_step_equals_single_process(tmp_path, loss, sparse):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank%d.npz")
    mp.spawn(_worker, args=(2, port, loss, out, sparse), nprocs=2, join=True)
    B, T, N, S = 6, 5, 17, 4
    params, cfg, batch = PU.build_case("GRU", [6], loss, N, B, T, S=S, seed=11)
    cfg["regularization"] = 0.03 if loss == "CCE" else 0.0
    upd = O.Updater("adam", 0.01, rho=0.9, beta1=0.9, beta2=0.999)
    ref_costs = [O.train_function(params, cfg, upd, PU.oracle_batch(batch)) for _ in range(2)]
    r0, r1 = np.load(out % 0), np.load(out % 1)
    assert np.allclose(r0["costs"], ref_costs, rtol=1e-12) and np.allclose(r1["costs"], ref_costs, rtol=1e-12)
    for i, p in enumerate(params):
        assert np.allclose(r0["p%d" % i], p, rtol=1e-10, atol=1e-14)
        assert np.array_equal(r0["p%d" % i], r1["p%d" % i])          # replicas stay identical


def test_shard_covers_the_batch():
    from sbr_amd.parallel import DataParallel
    for B, W in ((256, 8), (10, 4), (7, 2)):
        spans = [DataParallel.shard(B, W, r) for r in range(W)]
        assert spans[0][0] == 0 and spans[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))



def _worker_caps(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sbr_amd.parallel import DataParallel
    B, T, N, S = 7, 5, 17, 4                      # 7 rows over 2 ranks: shards of 4 and 3
    params, cfg, batch = PU.build_case("GRU", [6], "BPR", N, B, T, S=S, seed=12)
    lo, hi = DataParallel.shard(B, world, rank)
    # capacities as the engine derives them from the local shard (rows rounded up): they differ between the ranks
    eng = InbandSparseOracleEngine([p.copy() for p in params], cfg, "adam", B, lo, cap=N + 3 * rank)
    dp = DataParallel(eng, dist)
    ob = PU.oracle_batch(batch)
    tgt = ob["target"]                              # (unequal shards: the test hands every rank all targets)
    for _ in range(2):
        eng.set_batch(dict(X=ob["X"][lo:hi], mask=ob["mask"][lo:hi], target=tgt, samples=ob["samples"], pop=ob["pop"][lo:hi]))
        dp.train_step()
    assert dp._sp_same_cap == [False, False] and not getattr(dp, "_sp_all", {})     # both blocks took the counted form
    np.savez(out % rank, **{"p%d" % i: p for i, p in enumerate(eng.params)})
    dist.destroy_process_group()


def test_unequal_shard_capacities_take_the_counted_exchange(tmp_path):
    """the in-band sparse exchange all-gathers fixed-capacity buffers and strides every rank's slice by the LOCAL
    capacity; shards that differ by a row can have different capacities.  The ranks agree once on whether their capacities match;
    if not, the counted form (which gathers max(counts) rows) serves the block -- no mismatched all_gather, same result."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "cap%d.npz")
    mp.spawn(_worker_caps, args=(2, port, out), nprocs=2, join=True)
    B, T, N, S = 7, 5, 17, 4
    params, cfg, batch = PU.build_case("GRU", [6], "BPR", N, B, T, S=S, seed=12)
    upd = O.Updater("adam", 0.01, rho=0.9, beta1=0.9, beta2=0.999)
    for _ in range(2):
        O.train_function(params, cfg, upd, PU.oracle_batch(batch))
    r0, r1 = np.load(out % 0), np.load(out % 1)
    for i, p in enumerate(params):
        assert np.allclose(r0["p%d" % i], p, rtol=1e-10, atol=1e-14)
        assert np.array_equal(r0["p%d" % i], r1["p%d" % i])


class _GuardedStub(object):
    """the engine side of the data-parallel guard, without a GPU: RNNEngine's own method, a stand-in for the rest"""
    from sbr_amd.engine import RNNEngine as _E
    _rank_local_flush = _E._rank_local_flush

    def __init__(self):
        self.dp_guard, self._dp_collective, self.calls = False, False, []

    def query(self, what):
        return 2

    def section(self, name):
        return torch.zeros(4, dtype=torch.float64), 2

    def test_function(self, x, k=3):
        self._rank_local_flush("test_function")
        self.calls.append(("test_function", k))
        return k

    def flush_lazy(self):
        self._rank_local_flush("flush_lazy")
        self.calls.append(("flush_lazy",))


def _worker_guard(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sbr_amd.parallel import DataParallel
    eng = _GuardedStub()
    dp = DataParallel(eng, dist)
    assert eng.dp_guard
    with pytest.raises(RuntimeError, match="DataParallel.test_function"):
        eng.test_function(None)                      # rank-local: refused
    assert dp.test_function(None, k=7) == 7           # the collective: both ranks enter it
    dp.flush_lazy()
    assert eng.calls == [("test_function", 7), ("flush_lazy",)] and not eng._dp_collective
    dist.destroy_process_group()


def test_rank_local_flush_is_refused_and_the_collective_form_works():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker_guard, args=(2, port), nprocs=2, join=True)


def test_models_save_and_rank_through_the_data_parallel_collectives(tmp_path):
    """with dp_guard set every models.py path that called engine.get_all_param_values / test_function /
    predict_function directly raised, so a multi-rank driver built on models.py could neither checkpoint nor evaluate.  A model
    that was handed the DataParallel (attach_data_parallel) routes those three through its collectives (rnn_base.py:476 save,
    :185-211 the compiled functions)."""
    import pickle
    from sbr_amd.models import RNNOneHot

    class GuardedEngine(object):                      # what engine.RNNEngine does under dp_guard with sparse blocks
        def _no(self, *a, **k):
            raise RuntimeError("rank-local call: go through DataParallel")
        get_all_param_values = test_function = predict_function = _no

    class FakeDP(object):
        def __init__(self):
            self.calls = []

        def get_all_param_values(self):
            self.calls.append("get"); return [np.arange(6, dtype=np.float32).reshape(2, 3), np.ones(4, dtype=np.float32)]

        def test_function(self, inputs, k=10, exclude_seen=True):
            self.calls.append(("test", k)); return [np.zeros((1, k), dtype=np.int32)]

        def predict_function(self, X, mask):
            self.calls.append("predict"); return np.zeros((1, 5), dtype=np.float32)

    m = RNNOneHot(max_length=8, batch_size=4)
    m.engine = GuardedEngine()
    with pytest.raises(RuntimeError):
        m.save(str(tmp_path / "alone.pkl"))
    dp = FakeDP()
    m.attach_data_parallel(dp)
    f = str(tmp_path / "ckpt" / "model.pkl")
    m.save(f)
    m.save(str(tmp_path / "not_kept.pkl"), write=False)      # every other rank: takes part, keeps nothing
    assert not os.path.exists(str(tmp_path / "not_kept.pkl"))
    with open(f, "rb") as fh:
        got = pickle.load(fh)
    assert len(got) == 2 and np.array_equal(got[0], np.arange(6, dtype=np.float32).reshape(2, 3))
    m.test_function((None, None), k=7)
    m.predict_function(None, None)
    assert dp.calls == ["get", "get", ("test", 7), "predict"]


def test_an_unverifiable_stream_placement_falls_back_and_still_enters_the_agreement():
    """VERDICT round 5 / ADVICE: a stream check that cannot run (no spin kernel in this torch, a backend error on one rank) must
    (a) select the fully joined path -- nothing unverified is used -- and (b) still enter the MIN all-reduce of the verdict, so
    that no other rank waits in it alone.  CPU: torch.cuda.synchronize() raises here, which is exactly the `except` branch."""
    from sbr_amd.parallel import DataParallel

    class FakeDist:
        class ReduceOp:
            MIN = "min"
        calls = []

        def all_reduce(self, t, op=None, group=None):
            self.calls.append((tuple(t.shape), op))

    dp = DataParallel.__new__(DataParallel)
    dp.grads = torch.zeros(4)
    dp.side = None
    dp.world = 2
    dp.group = None
    dp.dist = FakeDist()
    dp.stream_check = None
    assert dp._side_collectives_are_ordered() is False
    assert dp.stream_check.startswith("fallback (skipped")
    assert [c[1] for c in dp.dist.calls] == ["min"]            # the agreement was entered exactly once
