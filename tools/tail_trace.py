#!/usr/bin/env python3
"""Where the overlapped step tail's consumers spend the time behind the BPTT chain.

SBR_TAIL_TRACE=1 makes the library keep 100 MHz wall-clock stamps (s_memrealtime: one clock for the whole chip) of
  - the chain launch's first and last instruction (block 0, wave 0),
  - every slab of the polling dW_hid GEMM (tile 0 of its group): the end of the wait, the end of its k loop,
  - every wave of the polling scatter-add, per piece: the end of its wait, its last flush.
This prints them relative to the chain's start for the last of a few C2 steps.  GPU only:  python tools/tail_trace.py [steps]
"""
import os
import sys

os.environ["SBR_TAIL_TRACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np                     # noqa: E402
import torch                           # noqa: E402
import bench                           # noqa: E402
from sbr_amd.engine import RNNEngine   # noqa: E402


def slab_table(K, rps, cap, growth, max_rows):
    """sbr_gemm.hip sbr_tail_slab_table"""
    import math
    max_n, scale = max(1, max_rows // 32), 1.0
    while True:
        lo, r = [0], 0
        while r < K:
            n = int(math.floor((growth * (r / rps) - 1.0) * scale))
            n = max(int(scale), min(max_n, n))
            r += min(32 * n, K - r)
            lo.append(r)
        if len(lo) - 1 <= cap or scale > 1e6:
            return lo
        scale *= 1.5


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    B, T, N = 256, 200, 3706
    eng = RNNEngine(cell="GRU", layers=[128], n_items=N, max_length=T, batch_size=B, loss="CCE", updater="adam", learning_rate=1e-3)
    eng.set_all_param_values(bench.initial_parameters(eng.cfg, np.random.default_rng(42)))
    hb = bench.synth_batches(1, B, T, N, 0, "full", seed=1235)[0]
    dev = eng.device
    eng.set_batch_device(torch.from_numpy(hb["X"]).to(dev), torch.from_numpy(hb["lengths"]).to(dev),
                         torch.from_numpy(hb["target"]).to(dev), None, torch.from_numpy(hb["pop"]).to(dev), B)
    for _ in range(steps):
        eng.train_step(sync=False)
    eng.synchronize()
    tr = eng.debug_buffer("tail_trace").view(np.uint64).astype(np.int64)
    cc = eng.debug_buffer("tail_chain_clock").view(np.uint64).astype(np.int64)
    t0, t1 = int(cc[2]), int(cc[3])
    us = lambda x: (x - t0) / 100.0
    print("chain: %.1f us (%.3f us / step), shader clock %.0f MHz" % (us(t1), us(t1) / T, cc[0] / max(1, cc[1]) * 100.0))
    g = tr[16:16 + 2 * 1024].reshape(-1, 2)
    g = g[g[:, 1] > 0]
    print("GEMM: %d slabs; first release %.1f, last release %.1f, last end %.1f (chain end %.1f)" %
          (len(g), us(g[:, 0].min()), us(g[:, 0].max()), us(g[:, 1].max()), us(t1)))
    d = (g[:, 1] - g[:, 0]) / 100.0
    print("  release -> end: median %.1f us, p90 %.1f, max %.1f" % (np.median(d), np.percentile(d, 90), d.max()))
    late = g[g[:, 1] > t1]
    order = np.argsort(late[:, 1])
    print("  %d workgroups end behind the chain; the last 12 (release, end, both relative to the chain's END):" % len(late))
    for i in order[-12:]:
        print("    %+7.1f  %+7.1f" % ((late[i, 0] - t1) / 100.0, (late[i, 1] - t1) / 100.0))
    # per slab: blockIdx.z -> slab zz (sbr_gemm_x6.hip: the last time steps first, the monitor owns slab 0)
    growth, max_rows = float(os.environ.get("SBR_TAIL_SLAB_GROWTH", 0.35)), int(os.environ.get("SBR_TAIL_SLAB_MAX", 2048))
    lo = slab_table(T * B, B, 255, growth, max_rows)
    nz = len(lo) - 1
    per_step = us(t1) / T
    print("  slab: rows, first step, step complete at, released, ended  (us relative to the chain's END)")
    full = tr[16:16 + 2 * 1024].reshape(-1, 2)
    for zz in list(range(0, min(nz, 48), 2)) + list(range(48, nz, 6)):
        z = zz
        kb, kk = lo[zz], lo[zz + 1] - lo[zz]
        t_need = kb // B
        ideal = (T - t_need) * per_step - us(t1)
        print("   %3d: %4d rows  t=%3d  %+7.1f  %+7.1f  %+7.1f" % (zz, kk, t_need, ideal, (full[z, 0] - t1) / 100.0, (full[z, 1] - t1) / 100.0))
    pr = lambda x: " ".join("%6.1f" % ((v - t1) / 100.0) for v in np.percentile(x, [0, 10, 50, 90, 100]))
    if os.environ.get("SBR_TAIL_SCATTER_LDS", "1") != "0":
        # LDS-accumulating scatter-add: per unit (workgroup) and time chunk: wait over, chunk done; slot 15: rows stored
        s = tr[8192:8192 + 2 * 16 * 256].reshape(256, 16, 2)
        nu = int((s[:, 15, 0] > 0).sum())
        print("scatter-add (LDS rows): %d units; us relative to the chain's END (min, p10, median, p90, max)" % nu)
        for c in range(14, -1, -1):
            u = s[:, c, 1] > 0
            if u.any():
                print("   chunk %d (%3d units): wait over %s | done %s" % (c, u.sum(), pr(s[:, c, 0][u]), pr(s[:, c, 1][u])))
        e = s[:, 15, 0][s[:, 15, 0] > 0]
        print("   rows stored: %s" % pr(e))
        ends = [("GEMM", g[:, 1]), ("scatter", e)]
    else:
        s = tr[8192:8192 + 2 * 4 * 1024].reshape(-1, 4, 2)
        used = s[:, :, 1] > 0
        rel, end = s[:, :, 0][used], s[:, :, 1][used]
        print("scatter-add: %d pieces on %d waves; last release %.1f, last end %.1f" % (used.sum(), used.any(axis=1).sum(), us(rel.max()), us(end.max())))
        print("  a wave's k-th piece: wait over | last flush, us relative to the chain's END (min, p10, median, p90, max)")
        for k in range(4):
            u = used[:, k]
            if u.any():
                print("   k=%d (%4d): %s | %s" % (k, u.sum(), pr(s[:, k, 0][u]), pr(s[:, k, 1][u])))
        ends = [("GEMM", g[:, 1]), ("scatter", end)]
    for name, e in ends:
        h = np.histogram((e[e > t1] - t1) / 100.0, bins=np.arange(0, 80, 5))[0]
        print("  %s ends per 5 us behind the chain: %s" % (name, " ".join(str(x) for x in h)))
    eng.close()


if __name__ == "__main__":
    main()
