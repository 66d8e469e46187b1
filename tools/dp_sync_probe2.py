#!/usr/bin/env python3
"""Follow-up of tools/dp_sync_probe.py: the process group's collective as a SYNC op (async_op=False: enqueued on the CURRENT stream by
this torch version, no internal stream, no events) on the engine's streams, against the async form through the group's own stream --
with the process group created FIRST, the order in which the async form costs the step +0.45 ms.  GPU only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import torch.distributed as dist
import bench
from sbr_amd.engine import RNNEngine


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29521")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    B, T, N = 256, 200, 3706
    eng = RNNEngine(cell="GRU", layers=[128], n_items=N, max_length=T, batch_size=B, loss="CCE", updater="adam", learning_rate=1e-3)
    eng.set_all_param_values(bench.initial_parameters(eng.cfg, np.random.default_rng(42)))
    hb = bench.synth_batches(1, B, T, N, 0, "full", seed=1235)[0]
    dev = eng.device
    X, L, Y, P = (torch.from_numpy(hb[k]).to(dev) for k in ("X", "lengths", "target", "pop"))
    small = torch.zeros(1024, device=dev)
    big = torch.zeros(7_300_000, device=dev)

    def timed(name, fn, n=300):
        try:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("%-76s %.4f ms/step (host %.4f)" % (name, (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3), flush=True)
        except Exception as ex:
            print("%-76s FAILED %r" % (name, ex), flush=True)

    def phases():
        eng.set_batch_device(X, L, Y, None, P, B)
        eng.zero_grads(); eng.forward(); eng.loss_backward_output(); eng.backward_recurrent(); eng.apply_update()

    cur = torch.cuda.current_stream()
    side = eng.side_stream()

    def async_wait():
        phases(); dist.all_reduce(small, async_op=True).wait()

    def sync_current(t):
        def f():
            phases(); dist.all_reduce(t)
        return f

    def sync_on_side(t):
        def f():
            phases()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                dist.all_reduce(t)
            cur.wait_stream(side)
        return f

    timed("five phase calls", phases)
    timed("+ all_reduce(4 KB), async_op=True + wait (the group's own stream)", async_wait)
    timed("+ all_reduce(4 KB), sync op on the current (engine) stream", sync_current(small))
    timed("+ all_reduce(29 MB), sync op on the current (engine) stream", sync_current(big))
    timed("+ all_reduce(4 KB), sync op under the engine's side stream + joins", sync_on_side(small))
    timed("+ all_reduce(29 MB), sync op under the engine's side stream + joins", sync_on_side(big))
    timed("+ all_reduce(4 KB), async_op=True + wait (again)", async_wait)
    from sbr_amd.parallel import DataParallel
    dp = DataParallel(eng, dist)

    def dpstep():
        eng.set_batch_device(X, L, Y, None, P, B); dp.train_step()
    timed("parallel.py step", dpstep)
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
