"""Data-parallel training step: the batch rows (independent user sub-sequences, the cost is their
mean: rnn_one_hot.py:71) are sharded over ranks, one process per GPU; every rank computes its
share of cost and gradients (already scaled by 1/B_global inside the engine), the flat gradient
section is summed with an all-reduce (backend "nccl" = RCCL over xGMI on MI355X, "gloo" in the
CPU tests), and every rank applies the identical optimizer step to its replica.

Overlap: the output-layer gradients (W_out, b_out + the cost scalar that rides at the end of the
section) are final after `loss_backward_output`, so their all-reduce is launched asynchronously
and runs on RCCL's stream while the BPTT chain of `backward_recurrent` executes; the recurrent
part follows.  With the HIP engine the first collective is ordered behind the engine's SIDE stream
(where those gradients are produced), so the main stream goes straight on to the BPTT chain instead
of waiting for them (`sbr_set_deferred_join`, include/sbr_rnn.h).  The sampled heads need the targets of ALL rows on every rank (Blackout's softmax
spans every target column, rnn_sampling.py:68-72,137): `gather_targets` all-gathers B int32.

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
        self.side = self.side2 = None
        self.tail = None
        if hasattr(engine, "side_stream") and getattr(self.grads, "is_cuda", False):
            self.side = engine.side_stream()
            engine.set_deferred_join(True)
            # overlapped step tail (engine.query("tail_chunks") >= 2): dW_in is finished by the engine's second side stream,
            # dW_hid by the first, everything else of the recurrent part by the main stream -- one collective behind each
            if hasattr(engine, "side_stream2") and engine.query("tail_chunks") >= 2:
                self.side2 = engine.side_stream2()
                self.tail = engine.tail_ranges()

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

    def _exchange_sparse(self, b):
        """Row-sparse block b: all-gather of (row ids, gradient rows) instead of an all-reduce of the whole block
        (8.2 GB of W_in at 1 M items against <= T*B_local rows of it per rank).  Every rank then adds all ranks'
        rows in rank order, so the replicas stay bit-identical."""
        import torch
        e, dist = self.engine, self.dist
        ids, rows, n = e.sparse_pack(b)
        cnt = torch.tensor([n], dtype=torch.int32, device=ids.device)
        cnts = [torch.empty_like(cnt) for _ in range(self.world)]
        dist.all_gather(cnts, cnt, group=self.group)
        counts = [int(c.item()) for c in cnts]
        m = max(counts)
        if m == 0:
            return
        ids_m, rows_m = ids[:m].contiguous(), rows[:m].contiguous()
        gids = [torch.empty_like(ids_m) for _ in range(self.world)]
        grows = [torch.empty_like(rows_m) for _ in range(self.world)]
        w1 = dist.all_gather(gids, ids_m, group=self.group, async_op=True)
        w2 = dist.all_gather(grows, rows_m, group=self.group, async_op=True)
        w1.wait(); w2.wait()
        for r in range(self.world):
            e.sparse_unpack_add(b, gids[r], grows[r], counts[r])

    def train_step(self):
        """One step on the batch already set on the engine; returns nothing (cost: read_cost())."""
        e = self.engine
        if self.world == 1 and not self.dist.is_initialized():
            e.zero_grads(); e.forward(); e.loss_backward_output(); e.backward_recurrent(); e.apply_update()   # joins inside
            return
        out_r, rec_r = self._ranges()
        e.zero_grads()
        e.forward()
        e.loss_backward_output()
        red = lambda lo, hi: self.dist.all_reduce(self.grads[lo:hi], group=self.group, async_op=True)
        if self.side is not None:
            import torch
            with torch.cuda.stream(self.side):       # RCCL waits for the side stream only; the main stream runs the chain
                works = [red(lo, hi) for lo, hi in out_r]
            e.backward_recurrent()
            if self.tail is None:
                e.join_side()                        # weight-gradient kernels of the recurrent part
            # (overlapped tail: no join here -- the collectives below follow the producing streams, sbr_apply_update joins)
        else:
            works = [red(lo, hi) for lo, hi in out_r]
            e.backward_recurrent()
        if self.tail is not None and len(rec_r) == 1:
            # two buckets, each behind the stream that finishes it: W_in behind the scatter-add (second side stream), and the
            # contiguous rest -- biases and initial states (the chain's partial sums, main stream) around W_hid (slab
            # reduction, side stream) -- behind the side stream once it has also waited for the main stream's share
            import torch
            (wi_lo, wi_hi), (wh_lo, wh_hi) = self.tail
            lo, hi = rec_r[0]
            assert lo == wi_lo and wi_hi <= wh_lo and wh_hi <= hi
            with torch.cuda.stream(self.side2):
                works.append(red(wi_lo, wi_hi))
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                works.append(red(wi_hi, hi))
        else:
            works += [red(lo, hi) for lo, hi in rec_r]
        for b in range(self._nsparse):
            self._exchange_sparse(b)
        for w in works:
            w.wait()
        e.apply_update()
