#!/usr/bin/env python3
"""Where the data-parallel step's time goes on ONE rank: the phase-by-phase step alone, with a process group alive, with the
collectives of parallel.py.  GPU only:  python tools/dp_phase_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from sbr_amd.engine import RNNEngine


def main():
    if os.environ.get("PROBE_OWN_STREAM", "0") != "0":      # the engine on a stream of its own instead of torch's default (null) stream
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            return body()
    return body()


def body():
    B, T, N = 256, 200, 3706
    eng = RNNEngine(cell="GRU", layers=[128], n_items=N, max_length=T, batch_size=B, loss="CCE", updater="adam", learning_rate=1e-3)
    eng.set_all_param_values(bench.initial_parameters(eng.cfg, np.random.default_rng(42)))
    hb = bench.synth_batches(1, B, T, N, 0, "full", seed=1235)[0]
    dev = eng.device
    X, L, Y, P = (torch.from_numpy(hb[k]).to(dev) for k in ("X", "lengths", "target", "pop"))

    def timed(name, fn, n=300):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()          # host time to ENQUEUE n steps
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("%-58s %.4f ms/step (host enqueue %.4f)" % (name, (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3), flush=True)

    def single():
        eng.set_batch_device(X, L, Y, None, P, B); eng.train_step(sync=False)

    def phases():
        eng.set_batch_device(X, L, Y, None, P, B)
        eng.zero_grads(); eng.forward(); eng.loss_backward_output(); eng.backward_recurrent(); eng.apply_update()

    timed("single call", single)
    timed("five phase calls", phases)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29517")
    if os.environ.get("PROBE_DEVICE_ID", "1") != "0":
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("nccl", rank=0, world_size=1)
    t_big = torch.zeros(7_300_000, device=dev)
    t_small = torch.zeros(1024, device=dev)
    timed("all_reduce of 29 MB alone (async + wait)", lambda: dist.all_reduce(t_big, async_op=True).wait())
    timed("all_reduce of 4 KB alone (async + wait)", lambda: dist.all_reduce(t_small, async_op=True).wait())

    def phases_then_small():
        phases(); dist.all_reduce(t_small, async_op=True).wait()
    timed("five phase calls + all_reduce of 4 KB on the current stream", phases_then_small)
    if os.environ.get("PROBE_SHORT", "0") != "0":
        dist.destroy_process_group(); eng.close(); return
    timed("five phase calls, process group alive", phases)
    timed("single call, process group alive", single)
    from sbr_amd.parallel import DataParallel
    dp = DataParallel(eng, dist)

    def dpstep():
        eng.set_batch_device(X, L, Y, None, P, B); dp.train_step()
    timed("parallel.py step (deferred join, %d views)" % (len(dp._views())), dpstep)
    eng.set_deferred_join(False)
    timed("five phase calls after the DataParallel object", phases)
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
