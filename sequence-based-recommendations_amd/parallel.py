"""Data-parallel training step: the batch rows (independent user sub-sequences, the cost is their
mean: rnn_one_hot.py:71) are sharded over ranks, one process per GPU; every rank computes its
share of cost and gradients (already scaled by 1/B_global inside the engine), the flat gradient
section is summed with an all-reduce (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the
CPU tests), and every rank applies the identical optimizer step to its replica.

Overlap: the output-layer gradients (W_out, b_out + the cost scalar that rides at the end of the
section) are final after `loss_backward_output`, so their all-reduce is issued right then and runs
while the BPTT chain of `backward_recurrent` executes; the recurrent part follows.  With the HIP
engine every collective is a sync op issued under the engine's SIDE stream (where those gradients are
produced): torch enqueues a sync collective on the current stream, so the main stream goes straight on
to the BPTT chain instead of waiting (`sbr_set_deferred_join`, include/sbr_rnn.h), and no stream of
the process group's own is involved -- a round trip through one cost the step 0.45 ms (train_step).  The sampled heads need the targets of ALL rows on every rank (Blackout's softmax
spans every target column, rnn_sampling.py:68-72,137): `gather_targets` all-gathers B int32.

Row-sparse blocks with the lazy-exact updaters (rmsprop / adadelta / nesterov / adam): a row's missed zero-gradient steps are
replayed when the row is next read, and WHERE that happens splits the replay (a short loop below 32 steps, closed forms above):
calls that bring rows up to date on one rank only -- get_params / save, predict / top-k, flush_lazy -- would let the replicas
drift apart by float32 roundings.  So with more than one rank the engine is GUARDED (`engine.dp_guard`): those calls raise when
made on the engine directly and are offered here as collectives -- `flush_lazy`, `test_function`, `predict_function`,
`get_all_param_values` -- which every rank must enter at the same step (a barrier in front: a rank that enters alone waits for the
others instead of silently forking the replicas; tests/test_dp_gloo.py, tests/test_gpu_dp_two_ranks.py::*_midrun_*).

Where the output layer's all-reduce is issued: small buckets (C2: 1.9 MB) right behind its gradient kernels, on the side stream,
beside the BPTT chain; buckets above `OUT_EARLY_BYTES` (a 26 744 x 256 W_out is 27 MB: ~0.2 ms on a ring over xGMI) would sit in
that stream's queue IN FRONT of the weight-gradient kernels `backward_recurrent` enqueues there and hold them back for as long, so
they are issued behind that work instead, back to back with the recurrent bucket.  The placement relies on a SYNC collective being
ordered on the stream it is issued under (this torch, 2.10, enqueues it there): `_side_collectives_are_ordered` verifies that with
data when a DataParallel with more than one rank is built and otherwise falls back to the fully joined path (`stream_check`); no
>1-GPU box was available to the builder to measure either placement.

`engine` is anything with the RNNEngine phase methods -- the CPU tests pass an oracle-backed
stand-in, production passes engine.RNNEngine.
"""


class DataParallel(object):
    def __init__(self, engine, dist=None, group=None):
        if dist is None:
            import torch.distributed as dist
        self.engine, self.dist, self.group = engine, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.grads, self.split = engine.section("grads")
        # stream-level overlap needs the engine's side stream and device tensors (RCCL); the gloo CPU tests use a
        # stand-in engine without either
        if self.world > 1 and hasattr(engine, "dp_guard"):
            engine.dp_guard = True                       # rank-local flushes of lazily stepped rows now raise (module docstring)
        self.side = self.side2 = None
        self.tail = None
        self.stream_check = None      # "ordered" / "fallback" / "skipped: ..." once the side-stream placement has been verified
        if hasattr(engine, "side_stream") and getattr(self.grads, "is_cuda", False):
            self.side = engine.side_stream()
            if self.world > 1 and not self._side_collectives_are_ordered():
                # this torch / backend does not order a sync collective behind the stream it is issued under: every collective
                # then runs on the engine's main stream behind a full join (the stand-in path below; correct, nothing overlapped)
                self.side = None
                if not str(self.stream_check or "").startswith("fallback"):
                    self.stream_check = "fallback"
                return
            engine.set_deferred_join(True)
            # overlapped step tail (engine.query("tail_chunks") >= 2): dW_in is finished by the engine's second side stream,
            # dW_hid by the first, everything else of the recurrent part by the main stream -- one collective behind each
            # (query "tail_streams": with SBR_TAIL_OVERLAP=2 the same kernels run on the MAIN stream -- nothing to order behind)
            if hasattr(engine, "side_stream2") and engine.query("tail_streams") == 1:
                self.side2 = engine.side_stream2()
                self.tail = engine.tail_ranges()

    OUT_EARLY_BYTES = 8 << 20      # output-layer buckets up to this size are reduced beside the BPTT chain (module docstring)

    def _side_collectives_are_ordered(self):
        """The step's placement of its collectives rests on ONE property of torch.distributed: a sync collective issued under
        `torch.cuda.stream(side)` consumes what `side` has produced so far and is complete for whatever `side` runs next (torch
        2.10 enqueues it on the current stream; an implementation with a stream of its own must order it by events to the same
        effect).  Checked once per DataParallel, on the real streams, with data: the side stream is held back by a ~1 ms spin,
        then fills a buffer, all-reduces it and copies it -- every call returns to the host long before the spin ends, so a
        collective that did not wait for the side stream would reduce the zeros the buffer still holds.  False -> the caller
        falls back to the fully joined path."""
        try:
            import torch
            t = torch.zeros(8192, device=self.grads.device)
            torch.cuda.synchronize()
            with torch.cuda.stream(self.side):
                torch.cuda._sleep(2000000)
                t.fill_(1.0)
                self.dist.all_reduce(t, group=self.group)
                probe = t.clone()
            torch.cuda.current_stream().wait_stream(self.side)
            torch.cuda.synchronize()
            ok = bool((probe == float(self.world)).all().item())
            note = None
        except Exception as ex:      # (no such private spin kernel in another torch, a backend error on this rank ...): NOTHING was
            ok = False               # verified, so the unverified placement is not used -- and this rank still enters the agreement
            note = "skipped: %r" % (ex,)      # below, so that no rank waits in it alone
        # every rank must take the same path: one that falls back alone would issue its collectives in another order
        import torch
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.grads.device)
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        ok = bool(int(flag.item()))
        self.stream_check = "ordered" if ok else ("fallback" if note is None else "fallback (%s)" % note)
        return ok

    # ---- collectives that bring lazily stepped rows up to date: every rank enters them at the same step
    def _collective(self, name, *a, **kw):
        e = self.engine
        if self.world > 1:
            self.dist.barrier(group=self.group)
        e._dp_collective = True
        try:
            return getattr(e, name)(*a, **kw)
        finally:
            e._dp_collective = False

    def flush_lazy(self):
        return self._collective("flush_lazy")

    def test_function(self, *a, **kw):
        return self._collective("test_function", *a, **kw)

    def predict_function(self, *a, **kw):
        return self._collective("predict_function", *a, **kw)

    def get_all_param_values(self):
        return self._collective("get_all_param_values")

    @staticmethod
    def shard(batch_size, world, rank):
        """rows [lo, hi) of the global batch owned by `rank` (contiguous, sizes differ by <= 1)."""
        base, extra = divmod(batch_size, world)
        lo = rank * base + min(rank, extra)
        return lo, lo + base + (1 if rank < extra else 0)

    def gather_targets(self, local_target):
        """all ranks' targets in global row order (equal shard sizes)."""
        if self.world == 1:
            return local_target
        import torch
        parts = [torch.empty_like(local_target) for _ in range(self.world)]
        self.dist.all_gather(parts, local_target, group=self.group)
        return torch.cat(parts)

    def _ranges(self):
        """(output-layer ranges, recurrent ranges) of the gradient section that take the dense all-reduce."""
        if not hasattr(self, "_rng"):
            e = self.engine
            n = self.grads.numel()
            ranges = e.dense_ranges() if hasattr(e, "dense_ranges") else [(0, self.split), (self.split, n)]
            self._rng = ([r for r in ranges if r[0] >= self.split], [r for r in ranges if r[0] < self.split])
            self._nsparse = len(e.sparse_blocks()) if hasattr(e, "sparse_blocks") else 0
        return self._rng

    # Row-sparse blocks whose fixed-capacity exchange buffers stay below this many bytes per rank travel with their row count
    # IN BAND (no host read, nothing synchronised: engine.sparse_pack_device / sparse_unpack_add_all); larger ones keep the
    # counted form -- one 4-byte read-back per block and step against an all-gather of only the rows the step touched (at a
    # 1 M-item catalogue the capacity is T*B_local rows of 8 KB each: there the bytes are the cost, not the read).
    INBAND_BYTES = 32 << 20

    def _exchange_sparse(self, b):
        """Row-sparse block b: all-gather of (row ids, gradient rows) instead of an all-reduce of the whole block
        (8.2 GB of W_in at 1 M items against <= T*B_local rows of it per rank).  Every rank then adds all ranks'
        rows in rank order, so the replicas stay bit-identical."""
        import torch
        e, dist = self.engine, self.dist
        if not hasattr(self, "_sp_info"):
            self._sp_info, self._sp_all = e.sparse_blocks(), {}
        n_rows, width, cap = self._sp_info[b]
        if not hasattr(self, "_sp_same_cap"):
            # the in-band form all-gathers fixed-capacity buffers and strides every rank's slice by THIS rank's capacity: shards
            # that differ by a row can round to different capacities (Bp = local rows rounded up to 16), so the ranks agree once,
            # here, whether all of them hold the same capacity for every block; if not, the counted form serves the block
            caps = torch.tensor([c for _, _, c in self._sp_info], dtype=torch.int64, device=self.grads.device)
            lo, hi = caps.clone(), caps.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
            self._sp_same_cap = [bool(x) for x in (lo == hi).tolist()]
        if hasattr(e, "sparse_pack_device") and self._sp_same_cap[b] and cap * width * 4 <= self.INBAND_BYTES:
            ids, rows = e.sparse_pack_device(b)          # ids[0] = the count; enqueued, not waited for
            if b not in self._sp_all:                     # persistent gather targets: [world][1 + cap], [world][cap][width]
                self._sp_all[b] = (torch.empty((self.world,) + tuple(ids.shape), dtype=ids.dtype, device=ids.device),
                                   torch.empty((self.world,) + tuple(rows.shape), dtype=rows.dtype, device=rows.device))
            ids_all, rows_all = self._sp_all[b]
            dist.all_gather(list(ids_all.unbind(0)), ids, group=self.group)      # (sync ops: on the current stream, see train_step)
            dist.all_gather(list(rows_all.unbind(0)), rows, group=self.group)
            e.sparse_unpack_add_all(b, ids_all, rows_all, self.world)      # one call: rank order, counts read on the device
            return
        ids, rows, n = e.sparse_pack(b)
        cnt = torch.tensor([n], dtype=torch.int32, device=ids.device)
        cnts = [torch.empty_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        counts = [int(c) for c in torch.cat(cnts).tolist()]      # (one read-back for all ranks' counts)
        m = max(counts)
        if m == 0:
            return
        ids_m, rows_m = ids[:m].contiguous(), rows[:m].contiguous()
        gids = [torch.empty_like(ids_m) for _ in range(self.world)]
        grows = [torch.empty_like(rows_m) for _ in range(self.world)]
        dist.all_gather(gids, ids_m, group=self.group)
        dist.all_gather(grows, rows_m, group=self.group)
        for r in range(self.world):
            e.sparse_unpack_add(b, gids[r], grows[r], counts[r])

    def _views(self):
        """the slices of the flat gradient section that travel, made once (a slice per step costs ~2 us of host time each)"""
        if not hasattr(self, "_vw"):
            out_r, rec_r = self._ranges()
            g = self.grads
            self._vw = dict(out=[g[lo:hi] for lo, hi in out_r], rec=[g[lo:hi] for lo, hi in rec_r])
        return self._vw

    def train_step(self, exposed=None):
        """One step on the batch already set on the engine; returns nothing (cost: read_cost()).
        exposed: a dict -- the step then brackets the wait for every collective with events on the engine's stream and adds
        the microseconds the stream really stood still for each bucket ("rec": the join behind the side stream's collectives,
        "sparse": the row-sparse exchange; the output layer's collective runs on the side stream and is never waited for alone); a survey mode
        (events cost the stream a few microseconds each: bench.py uses it outside its timed regions)."""
        e = self.engine
        if self.world == 1 and not self.dist.is_initialized():
            e.zero_grads(); e.forward(); e.loss_backward_output(); e.backward_recurrent(); e.apply_update()   # joins inside
            return
        vw = self._views()
        # Collectives are SYNC ops issued under the stream that has to carry them: this torch enqueues a sync collective on the
        # current stream, with no stream of the process group's own and no events.  The async form (the group's stream, an event
        # each way) cost the step +0.45 ms of DEVICE time on one rank -- a record / wait round trip through a foreign stream that
        # shares a hardware queue with one of the engine's (tools/dp_sync_probe.py, profiles/round3_T_dp_sync_probe.txt: through a
        # normal-priority torch stream 0.79 ms, through the engine's own side stream 0.46, the sync op on the engine's stream 0.45
        # against 0.44 without any collective).  One communicator must not run two collectives at once, so all of a step's
        # collectives go to ONE stream: the side stream where there is one (behind the output layer's gradient kernels first,
        # beside the BPTT chain; then behind everything the recurrent part needs), the current stream otherwise.
        red = lambda t: self.dist.all_reduce(t, group=self.group)
        import contextlib
        torch = None
        if self.side is not None:
            import torch
        on_side = (lambda: torch.cuda.stream(self.side)) if self.side is not None else contextlib.nullcontext

        def wait(name, fn):                              # the engine's stream waits: bracketed in survey mode
            if exposed is None or torch is None:
                fn()
            else:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                exposed.setdefault("_ev", []).append((name, e0, e1))
        out_early = sum(t.numel() for t in vw["out"]) * 4 <= self.OUT_EARLY_BYTES
        e.zero_grads()
        e.forward()
        e.loss_backward_output()
        if out_early:
            with on_side():                              # output layer: final now; runs beside the BPTT chain on the side stream
                for t in vw["out"]:
                    red(t)
        e.backward_recurrent()
        if self.side is not None:
            cur = torch.cuda.current_stream()
            if self.tail is None:
                e.join_side()                            # weight-gradient kernels of the recurrent part: the current stream has them
                self.side.wait_stream(cur)
            else:
                # overlapped tail: W_in is finished by the second side stream, W_hid by the first, biases and initial states (the
                # chain's partial sums) by the main stream -- the side stream waits for the other two (sbr_apply_update joins)
                self.side.wait_stream(self.side2)
                self.side.wait_stream(cur)
            with on_side():
                for t in ([] if out_early else vw["out"]) + vw["rec"]:
                    red(t)
            wait("rec", lambda: cur.wait_stream(self.side))
            # row-sparse blocks: packed and unpacked by the engine on ITS stream, so their all-gathers run there too (behind the
            # wait above: one collective of the communicator at a time)
            for b in range(self._nsparse):
                wait("sparse", lambda b=b: self._exchange_sparse(b))
        else:
            for t in ([] if out_early else vw["out"]) + vw["rec"]:
                red(t)
            for b in range(self._nsparse):
                self._exchange_sparse(b)
        e.apply_update()

    @staticmethod
    def exposed_us(exposed):
        """folds the event pairs train_step(exposed=...) recorded into {bucket: total microseconds}; synchronises"""
        import torch
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1 in exposed.pop("_ev", []):
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1) * 1e3
        for k, v in out.items():
            exposed[k] = exposed.get(k, 0.0) + v
        return exposed
