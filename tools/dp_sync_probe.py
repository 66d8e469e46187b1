#!/usr/bin/env python3
"""What exactly in a torch.distributed collective doubles the step's device time (tools/dp_phase_probe.py)?  The phase-by-phase step
followed by: an event record alone; a record / wait round trip through a second torch stream (normal and high priority; through the
engine's own side stream); an unrelated kernel on a second stream; the process group's all_reduce; RCCL's all_reduce called directly
on the engine's stream (no second stream at all).  GPU only:  python tools/dp_sync_probe.py"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
from sbr_amd.engine import RNNEngine


def main():
    B, T, N = 256, 200, 3706
    eng = RNNEngine(cell="GRU", layers=[128], n_items=N, max_length=T, batch_size=B, loss="CCE", updater="adam", learning_rate=1e-3)
    eng.set_all_param_values(bench.initial_parameters(eng.cfg, np.random.default_rng(42)))
    hb = bench.synth_batches(1, B, T, N, 0, "full", seed=1235)[0]
    dev = eng.device
    X, L, Y, P = (torch.from_numpy(hb[k]).to(dev) for k in ("X", "lengths", "target", "pop"))
    small = torch.zeros(1024, device=dev)

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
            print("%-72s %.4f ms/step (host %.4f)" % (name, (t2 - t0) / n * 1e3, (t1 - t0) / n * 1e3), flush=True)
        except Exception as ex:
            print("%-72s FAILED %r" % (name, ex), flush=True)

    def phases():
        eng.set_batch_device(X, L, Y, None, P, B)
        eng.zero_grads(); eng.forward(); eng.loss_backward_output(); eng.backward_recurrent(); eng.apply_update()

    cur = torch.cuda.current_stream()
    s_norm, s_high = torch.cuda.Stream(), torch.cuda.Stream(priority=-1)
    s_side = eng.side_stream()

    def round_trip(other):
        def f():
            phases()
            e1 = torch.cuda.Event(); e1.record(cur)
            other.wait_event(e1)
            e2 = torch.cuda.Event(); e2.record(other)
            cur.wait_event(e2)
        return f

    def record_only():
        phases(); e1 = torch.cuda.Event(); e1.record(cur)

    def unrelated_kernel():
        phases()
        with torch.cuda.stream(s_norm):
            small.add_(1.0)

    timed("five phase calls", phases)
    timed("+ one event record on the engine's stream", record_only)
    timed("+ an unrelated small kernel on a second (normal) stream", unrelated_kernel)
    timed("+ record / wait round trip through a second stream, normal priority", round_trip(s_norm))
    timed("+ record / wait round trip through a second stream, high priority", round_trip(s_high))
    timed("+ record / wait round trip through the engine's own side stream", round_trip(s_side))
    timed("five phase calls (again)", phases)
    # RCCL directly, on the engine's stream: no second stream, no events
    try:
        rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))
        comm = ctypes.c_void_p()
        devs = (ctypes.c_int * 1)(torch.cuda.current_device())
        rc = rccl.ncclCommInitAll(ctypes.byref(comm), 1, devs)
        assert rc == 0, rc
        big = torch.zeros(7_300_000, device=dev)
        rccl.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]

        def direct(t):
            def f():
                phases()
                r = rccl.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), 7, 0, comm, cur.cuda_stream)
                assert r == 0, r
            return f
        timed("+ ncclAllReduce(4 KB) called directly on the engine's stream", direct(small))
        timed("+ ncclAllReduce(29 MB) called directly on the engine's stream", direct(big))
        out = torch.zeros_like(big)

        def direct_oop():
            phases()
            r = rccl.ncclAllReduce(big.data_ptr(), out.data_ptr(), big.numel(), 7, 0, comm, cur.cuda_stream)
            assert r == 0, r
        timed("+ ncclAllReduce(29 MB, out of place) directly on the engine's stream", direct_oop)
        rccl.ncclCommDestroy(comm)
    except Exception as ex:
        print("direct RCCL part FAILED %r" % (ex,), flush=True)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29519")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))

    def pg():
        phases(); dist.all_reduce(small, async_op=True).wait()
    timed("+ process group all_reduce(4 KB), async + wait", pg)
    timed("five phase calls (process group alive)", phases)
    dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
