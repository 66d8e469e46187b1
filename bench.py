"""bench.py -- user-sequences/sec of training on MovieLens-1M-shaped synthetic data.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1], SURVEY.md section 8d "C2"): `train.py -m RNN --r_t GRU --r_l 128
--max_length 200 -b 256 --loss CCE --u_m adam`: 1-layer GRU-128, full softmax over N=3706 items,
batch 256 sequences of length 200 PER GPU (weak scaling: the global batch is 256*N, every rank holds
256 rows, gradients are all-reduced over RCCL and every rank applies the identical Adam step).
A "step" = one full train_function call (gather -> GRU -> softmax/CCE -> BPTT -> scatter -> Adam)
on one resident batch.  Inputs live in HBM before the timed region starts.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel of the step, measured live with HIP events (ring of per-step
                  event sets inside libsbr_rnn.so, read back after the timed region)
  cpu_baseline -- the torch-CPU float32 port of the same step (oracle/torch_ref.py; Theano/Lasagne
                  cannot run here) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (cell, layers, n_items, loss, n_samples)
    "c2": ("GRU", [128], 3706, "CCE", 0),        # BASELINE.json configs[1] -- the metric's config
    "c1": ("LSTM", [20], 3706, "CCE", 0),        # configs[0]: the reference's own CPU-runnable case
    "c4": ("LSTM", [256], 26744, "CCE", 0),      # configs[3] shape (per-GPU part)
    "c3": ("LSTM", [256], 100000, "Blackout", 32),
    "l128": ("LSTM", [128], 3706, "CCE", 0),     # C2's shape with the other gated cell (the pipelined 128-unit kernels' LSTM form)
    "c5": ("LSTM", [512, 512], 1000000, "Blackout", 32),   # configs[4] shape (per-GPU part): 41 GB arena, dense Adam over 2.6 G parameters
}
F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, exact f32
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 / fp16 MFMA peak: what the split products actually occupy
HBM_PEAK_GBS = 8000.0             # HBM3E spec
N_CUS = 256


def zipf_items(rng, n_items, size, perm):
    """item ids ~ Zipf(alpha=1.0) over N through a fixed permutation."""
    w = 1.0 / np.arange(1, n_items + 1)
    cdf = np.cumsum(w / w.sum())
    return perm[np.minimum(np.searchsorted(cdf, rng.random(size)), n_items - 1)].astype(np.int32)


def synth_batches(n_batches, B, T, n_items, n_samples, lengths_mode, seed):
    rng = np.random.default_rng(seed)
    perm = np.random.default_rng(1234).permutation(n_items)
    out = []
    for _ in range(n_batches):
        if lengths_mode == "full":
            lens = np.full(B, T, dtype=np.int32)
        else:   # "ml1m": L ~ clip(lognormal(4.6, 0.9), 2, T)
            lens = np.clip(rng.lognormal(4.6, 0.9, size=B), 2, T).astype(np.int32)
        X = np.zeros((B, T, 1), dtype=np.int32)
        ids = zipf_items(rng, n_items, B * T, perm).reshape(B, T)
        m = np.arange(T)[None, :] < lens[:, None]
        X[:, :, 0] = np.where(m, ids, 0)
        out.append(dict(X=X, lengths=lens, mask=m.astype(np.float32), target=zipf_items(rng, n_items, B, perm),
                        samples=rng.integers(0, n_items, size=max(n_samples, 1)).astype(np.int32),
                        pop=np.ones(B, dtype=np.float32)))
    return out


def initial_parameters(cfg, rng):
    """Random-init weights of the architecture, in the engine's parameter order (= Lasagne's get_all_param_values order), by
    the initialisers the reference's layers name (engine.initial_values; sparse_lstm.py:143-171, rnn_one_hot.py:65)."""
    from sbr_amd.engine import describe_params, initial_values
    return initial_values(describe_params(cfg), rng)


def arithmetic_note(eng):
    """what the kernels this configuration selects do with an f32 product (from sbr_query, not a fixed string)"""
    names = {0: "exact-f32 MFMA (v_mfma_f32_16x16x4_f32)", 1: "the two-plane fp16 split, planes packed into the rows of a 4-row tile, ONE 2:4-sparse "
             "matrix instruction per product (v_smfmac_f32_16x16x64_f16: three plane products in its four accumulator rows)",
             2: "the two-plane fp16 split, planes packed into the tile rows: two MFMAs per product", 3: "the two-plane fp16 split: three "
             "MFMAs per product", 6: "the exact three-plane bf16 split: six MFMAs per product"}
    try:
        pf, pb = eng.query("rec_products_fwd"), eng.query("rec_products_bwd")
        fam = {0: "triage", 1: "cluster (rec_*_c16 / _cl)", 2: "128-unit pipelined (rec_*_x6p)", 3: "32/64-unit (rec_*_x6q)", 4: "general"}.get(eng.query("rec_kernel"), "?")
    except Exception:
        return "f32 tensors and accumulation; f32-class split products on the fp16 / bf16 matrix pipe (DESIGN.md section 3)"
    chain = names.get(pf, "%d terms" % pf) if pf == pb else "forward: %s; backward: %s" % (names.get(pf, pf), names.get(pb, pb))
    return ("f32 tensors and accumulation; matrix products of f32 operands as splits on the fp16 / bf16 matrix pipe with f32-class error. "
            "Recurrent chains of the top layer [%s kernels]: %s. Weight-gradient GEMM of the chain: fp16 split, three MFMAs; the other "
            "GEMMs: bf16x6 (DESIGN.md section 3)" % (fam, chain))


def counter_passes(argv_child, log):
    """HBM bytes per launch and kernel from two rocprofv3 --pmc passes of THIS command (child: --quick, few steps), FETCH_SIZE and
    WRITE_SIZE each in its own run with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3: the two do not fit one pass; on
    gfx950 FETCH_SIZE reports half of the bytes of 16-byte-per-lane streams -> doubled; WRITE_SIZE as reported).  Counter passes
    serialise kernels, so the overlapped tail's consumers cannot wait beside the chain there: SBR_TAIL_OVERLAP=2 (the same
    kernels on one stream).  Returns {kernel name: {"fetch_kb", "write_kb", "hbm_bytes", "launches"}} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        log("counter passes skipped: no rocprofv3")
        return None
    out = {}
    tmp = tempfile.mkdtemp(prefix="sbr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", SBR_TAIL_OVERLAP="2")
    try:
        for ctr, key in (("FETCH_SIZE", "fetch_kb"), ("WRITE_SIZE", "write_kb")):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "c", "--",
                   sys.executable, os.path.abspath(__file__)] + argv_child
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                log("counter pass %s failed (rc %d): %s" % (ctr, r.returncode, r.stderr[-300:]))
                return None
            acc = {}
            for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                with open(fn) as f:
                    for row in csv.DictReader(f):
                        if row["Counter_Name"] == ctr:
                            acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            for k, v in acc.items():
                m = sum(v[1:]) / (len(v) - 1) if len(v) > 1 else v[0]      # (the first launch of a kernel includes its code fetch)
                out.setdefault(k, {})[key] = m
                out[k]["launches"] = len(v)
    except Exception as ex:
        log("counter passes skipped:", ex)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k, e in out.items():
        e["hbm_bytes"] = int(2 * e.get("fetch_kb", 0.0) * 1024 + e.get("write_kb", 0.0) * 1024)
    return out


def mfma_counter_pass(argv_child, log):
    """Matrix-pipe utilisation from the SQ counters (north_star: "MFMA utilisation ... from rocprof counters"): one rocprofv3 --pmc
    pass of THIS command (child: --quick) with SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES, --kernel-trace only.  MFMA_BUSY counts
    matrix-pipe cycles summed over the chip's 1024 SIMDs (16 per v_mfma_f32_16x16x32 / v_smfmac_f32_16x16x64); SQ_BUSY_CYCLES sums
    the 32 SQ instances (8 XCDs x 4 shader engines), so a kernel's length in shader cycles is SQ_BUSY / 32 and
    utilisation of the whole chip = MFMA_BUSY / (1024 x SQ_BUSY / 32).  Returns {kernel: {...}} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="sbr_sq_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", SBR_TAIL_OVERLAP="2")
    acc = {}
    try:
        cmd = [exe, "--pmc", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "c", "--",
               sys.executable, os.path.abspath(__file__)] + argv_child
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            log("MFMA counter pass failed (rc %d): %s" % (r.returncode, r.stderr[-300:]))
            return None
        for fn in glob.glob(tmp + "/**/*counter_collection.csv", recursive=True):
            with open(fn) as f:
                for row in csv.DictReader(f):
                    acc.setdefault(row["Kernel_Name"], {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    except Exception as ex:
        log("MFMA counter pass skipped:", ex)
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k, c in acc.items():
        mf, sq = c.get("SQ_VALU_MFMA_BUSY_CYCLES", []), c.get("SQ_BUSY_CYCLES", [])
        if not mf or not sq or max(mf) <= 0:
            continue
        m, b = (sum(x[1:]) / (len(x) - 1) if len(x) > 1 else x[0] for x in (mf, sq))
        out[k] = {"mfma_busy_cycles": int(m), "sq_busy_cycles": int(b), "launches": len(mf),
                  "kernel_shader_cycles": int(b / 32.0), "mfma_util_of_chip": round(m / (32.0 * b), 5) if b > 0 else None}
    return out


def strong_scaling_pieces(log, budget_s=150.0):
    """Single-rank steps at the per-GPU shares of a FIXED global batch of 256 (the reference's -b: SURVEY 8e strong scaling) -- what a
    rank of a 2 / 4 / 8-GPU job computes between its collectives: C2 and C4 at B_local = 128 / 64 / 32, child runs of this file
    (--quick: timed regions only).  The chains are 2 T dependent steps whatever the rows, so these do not shrink like 1 / N."""
    import subprocess
    out, t_all = {}, time.perf_counter()
    # (a rank of a launcher's job starts these: its children are plain one-GPU runs, not members of that job)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE",
                        "MASTER_ADDR", "MASTER_PORT", "ROLE_NAME", "OMP_NUM_THREADS") and not k.startswith(("TORCHELASTIC_", "PET_"))}
    for name in ("c2", "c4"):
        for bl in (128, 64, 32):
            key = "%s_b%d" % (name, bl)
            if budget_s - (time.perf_counter() - t_all) < 15:
                out[key] = {"skipped": "time budget spent"}
                continue
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", name, "--quick", "--batch", str(bl), "--steps", "20",
                                    "--warmup", "5", "--repeats", "3"], capture_output=True, text=True, timeout=60, env=env)
                d = json.loads(r.stdout.strip().splitlines()[-1])
                out[key] = {"ranks_of_global_256": 256 // bl, "rows_per_rank": bl, "ms_per_step": d["ms_per_step"],
                            "sequences_per_s_per_rank": d["value"]}
                log("strong-scaling piece %s: %.4f ms/step" % (key, d["ms_per_step"]))
            except Exception as ex:
                out[key] = {"skipped": repr(ex)[:200]}
    out["note"] = ("measured on ONE GPU (no collective): the compute of one rank of an N-rank job at global batch 256; an N-GPU step is this "
                   "plus the exposed part of its gradient exchange (DESIGN.md section 6: modelled, never measured on more than one GPU)")
    return out


def other_config_lines(log, names=("c1", "c4", "c3", "c5", "c5_bf16"), budget_s=520.0):
    """The other BASELINE configurations next to the headline one, each as a child run of this file (--brief: phase survey, chain-only
    timing, three timed regions): {name: {ms_per_step, value, roofline_frac, ...}}.  C5 builds a 47 GB arena and draws 2.6e9 initial
    values on the host first (~45 s); a configuration that does not fit the time budget is reported as skipped, with the reason."""
    import subprocess
    out, t_all = {}, time.perf_counter()
    for key, limit in ((n, {"c1": 150, "c4": 210, "c3": 150, "c5": 300, "c5_bf16": 300}[n]) for n in names):
        # "c5_bf16": BASELINE configs[4] on the arithmetic it names -- bf16 MFMA output projection and bf16 layer GEMMs (--flags 384)
        name, extra = (key, []) if key != "c5_bf16" else ("c5", ["--flags", "384"])
        left = budget_s - (time.perf_counter() - t_all)
        if left < 30:
            out[key] = {"skipped": "time budget of the default run spent (%d s)" % budget_s}
            continue
        t0 = time.perf_counter()
        try:
            if key in ("c1", "c4"):
                extra = extra + ["--brief-cpu-seconds", "25"]
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", name, "--brief", "--steps", "20", "--warmup", "5",
                                "--repeats", "3"] + extra, capture_output=True, text=True, timeout=min(limit, left))
            d = json.loads(r.stdout.strip().splitlines()[-1])
            rf, ch = d.get("roofline") or {}, d.get("chains") or {}
            out[key] = {"workload": d["config"]["workload"], "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "dtype": d.get("dtype"),
                         "roofline": {k: rf.get(k) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "launch_us")},
                         "chains_us": {"rec_fwd": ch.get("rec_fwd_us"), "rec_bwd": ch.get("rec_bwd_us")},
                         "outside_chains_us": ch.get("outside_chains_us"),
                         "phases_us": {k: v for k, v in (d.get("phases_us") or {}).items() if k != "note"},
                         "arithmetic": d["config"].get("arithmetic"), "wall_s": round(time.perf_counter() - t0, 1)}
            if d.get("cpu_baseline"):
                out[key]["cpu_baseline"] = {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample")}
            log("other config %s: %.4f ms/step (%.0f s)" % (key, d["ms_per_step"], time.perf_counter() - t0))
        except subprocess.TimeoutExpired:
            out[key] = {"skipped": "child run exceeded %d s" % min(limit, left)}
        except Exception as ex:
            out[key] = {"skipped": repr(ex)[:200]}
    return out


def self_launch(n):
    """`python bench.py --gpus N ...` started directly (no torch.distributed.run around it): start the N ranks here -- one
    process per GPU, rendezvous on 127.0.0.1 at a free port -- with this very command line; rank 0 prints the JSON line, the
    children's stdout / stderr pass through, the exit code is theirs."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes fails without it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")                 # (torch.distributed.run would set 1 and say so on stderr)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without a launcher: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--lengths", default="full", choices=["full", "ml1m"])
    ap.add_argument("--batch", type=int, default=256, help="sequences per GPU per step")
    ap.add_argument("--max_length", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dp", action="store_true", help="(debug) take the data-parallel phase path even with one rank")
    ap.add_argument("--cpu-steps", type=int, default=3, help="max timed CPU-baseline steps")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="time budget of the CPU-baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(host cores, 16): the port scales to ~16 threads on the GPU box")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the line reports the median region and all of them")
    ap.add_argument("--pmc-json", default=None, help="HBM counter bytes per launch for roofline.traffic, from a separate rocprofv3 --pmc pass "
                    "of this same command (tools/pmc_summary.py output); without it traffic is null")
    ap.add_argument("--no-pmc", action="store_true", help="skip the counter passes (two rocprofv3 --pmc child runs of this command, "
                    "FETCH_SIZE and WRITE_SIZE, ~1 min) that fill roofline.traffic")
    ap.add_argument("--quick", action="store_true", help="timed regions only: no survey extras, no sustained / train-loop / counter / "
                    "CPU legs (what the counter passes run as their child)")
    ap.add_argument("--sustained-seconds", type=float, default=2.0, help="length of the one long region behind the timed ones (0: skip)")
    ap.add_argument("--loop-iters", type=int, default=1000, help="iterations of the end-to-end training-loop leg (0: skip)")
    ap.add_argument("--flags", type=int, default=int(os.environ.get("SBR_BENCH_FLAGS", "0")), help="SBR_FLAG_* bits for the engine (include/sbr_rnn.h), e.g. 384 = bf16 output "
                    "projection + bf16 layer GEMMs (BASELINE configs[4]); the line's dtype then says so")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --batch rows PER GPU (the global batch grows with N; the default and the driver's curve); strong: --batch "
                         "is the GLOBAL batch, as the reference's -b is (rnn_one_hot.py:71: the cost is a mean over the batch), every "
                         "rank holds --batch / N rows")
    ap.add_argument("--brief", action="store_true", help="phase survey + timed regions + roofline only (what the default run starts "
                    "for the other BASELINE configurations): no counter / sustained / train-loop / CPU / side-kernel legs")
    ap.add_argument("--strong-pieces", action="store_true", help="with --gpus N > 1: rank 0 also measures strong_scaling_model (C2 / C4 at 128 / "
                    "64 / 32 rows on one GPU) behind the timed regions; the default single-GPU run always carries it")
    ap.add_argument("--brief-cpu-seconds", type=float, default=0.0,
                    help="with --brief: keep a bounded cpu_baseline leg (scripted port, one thread count) of about this many seconds "
                         "(what the default run asks of its C1 and C4 children: BASELINE.md section 2 puts a CPU number beside each)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the child runs of BASELINE configs c1 / c3 / c4 / c5")
    ap.add_argument("--other-configs", default="c1,c4,c3,c5,c5_bf16", help="which of them the default run starts (comma-separated, in this order; "
                    "c5_bf16 = C5 with --flags 384: bf16 output projection + bf16 layer GEMMs, the arithmetic BASELINE configs[4] names)")
    ap.add_argument("--dp-backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend of the data-parallel step: nccl = RCCL over xGMI (one rank per GPU); gloo lets several "
                         "ranks share one device (RCCL refuses that) -- how the one-GPU test box exercises --gpus 2")
    args = ap.parse_args()
    if args.brief:
        args.no_pmc = args.no_other_configs = True
        args.no_cpu_baseline = args.brief_cpu_seconds <= 0
        if args.brief_cpu_seconds > 0:
            args.cpu_seconds, args.cpu_steps = args.brief_cpu_seconds, min(args.cpu_steps, 3)
        args.sustained_seconds, args.loop_iters = 0.0, 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))

    # stdout carries ONE line, the JSON: whatever the libraries underneath print there (RCCL's version banner, gloo's connection
    # notes, the data handler's "Opening file") goes to stderr -- file descriptor 1 points at stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def log(*a):
        print("[bench %6.1fs]" % (time.perf_counter() - T0), *a, file=sys.stderr, flush=True)
    T0 = time.perf_counter()
    import torch
    import torch.distributed as dist
    from sbr_amd.engine import RNNEngine
    from sbr_amd.parallel import DataParallel
    log("imports done")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's --nproc-per-node must equal --gpus" % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py measures the HIP engine: no GPU visible")
    if world > n_dev and args.dp_backend == "nccl":
        raise SystemExit("--gpus %d but %d visible device(s): RCCL needs one GPU per rank (--dp-backend gloo shares devices)" % (world, n_dev))
    device_index = local_rank % n_dev
    torch.cuda.set_device(device_index)
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if args.dp_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    cell, layers, n_items, loss, n_samples = CONFIGS[args.config]
    B, T = args.batch, args.max_length
    if args.scaling == "strong":      # the reference's -b is the global batch: split it over the ranks
        if args.batch % world:
            raise SystemExit("--scaling strong: --batch %d is not a multiple of %d ranks" % (args.batch, world))
        B = args.batch // world
    Bg = B * world
    eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=T, batch_size=Bg, local_batch=B,
                    row_offset=rank * B, loss=loss, n_samples=n_samples, updater="adam", learning_rate=1e-3, flags=args.flags)
    params = initial_parameters(eng.cfg, np.random.default_rng(42))
    eng.set_all_param_values(params)

    # synthetic batches, resident in HBM before the timed region
    nb = 8
    host_batches = synth_batches(nb, B, T, n_items, n_samples, args.lengths, seed=1235 + rank)
    dev = eng.device
    dev_batches = []
    for hb in host_batches:
        d = dict(X=torch.from_numpy(hb["X"]).to(dev), lengths=torch.from_numpy(hb["lengths"]).to(dev),
                 target=torch.from_numpy(hb["target"]).to(dev), samples=torch.from_numpy(hb["samples"]).to(dev),
                 pop=torch.from_numpy(hb["pop"]).to(dev))
        dev_batches.append(d)
    dp = DataParallel(eng, dist)
    log("engine ready, arena %.1f MB" % (eng.arena_bytes / 1e6))

    def step(i, exposed=None):
        d = dev_batches[i % nb]
        tgt = d["target"]
        if world > 1 and loss != "CCE":      # sampled heads need every rank's targets (rnn_sampling.py:137)
            tgt = dp.gather_targets(tgt)
        eng.set_batch_device(d["X"], d["lengths"], tgt, d["samples"] if loss != "CCE" else None, d["pop"], B)
        if world == 1 and not args.force_dp:
            eng.train_step(sync=False)       # one C call: zero grads, fwd, loss, BPTT, scatter, Adam
        else:
            dp.train_step(exposed=exposed)   # same phases with the RCCL all-reduces in between

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if not np.isfinite(eng.read_cost()):       # (also raises on a device-side fault flag: a broken kernel build ends here, not after the regions)
        raise ValueError("Cost is NaN")
    log("warmup done")
    # survey pass (untimed, after the warm-up): every phase bracketed by HIP events -> the per-phase table and the name of
    # the dominant kernel.  Eight event records per step cost the stream ~40 us at C2, so the timed region below brackets
    # only that kernel (two records).
    survey, dom_phase = None, "rec_bwd"
    try:
        if args.quick:
            raise RuntimeError("--quick")
        eng.enable_timing(True)
        for i in range(min(args.steps, 20)):
            step(args.warmup + i)
        torch.cuda.synchronize()
        survey = eng.phase_times()
        cand = [k for k in ("rec_fwd", "rec_bwd", "scatter") + (() if eng.query("fused_gather") else ("gather",))]
        dom_phase = max(cand, key=lambda k: survey[k])
    except Exception as ex:
        log("phase survey skipped:", ex)
    # chain-only survey: event pairs around every launch of a recurrent chain kernel (all layers): what the roofline of stacked
    # configurations is priced on, and `outside_chains_us`
    chain = None
    if survey is not None:
        try:
            n_ch = 6
            eng.enable_timing(False)
            eng.chain_timing(True)
            for i in range(n_ch):
                step(args.warmup + i)
            ct = eng.chain_timing(False)
            chain = {"rec_fwd": ct["fwd_us"] / n_ch, "rec_bwd": ct["bwd_us"] / n_ch,
                     "launches_per_step": (ct["fwd_launches"] + ct["bwd_launches"]) / float(n_ch)}
        except Exception as ex:
            log("chain survey skipped:", ex)
    # data parallel: how long the engine's stream really stands still for each collective (events around every wait; a survey
    # pass like the one above, outside the timed regions)
    dp_info = None
    if world > 1 or args.force_dp:
        n_sv = min(args.steps, 10)
        eng.enable_timing(False)
        ex = {}
        for i in range(n_sv):
            step(args.warmup + i, exposed=ex)
        DataParallel.exposed_us(ex)
        dp_info = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                   "collectives_per_step": {"dense_buckets": len(dp._views()["out"]) + len(dp._views()["rec"]),
                                            "sparse_blocks": dp._nsparse},
                   "exposed_us_per_step": {k: round(v / n_sv, 2) for k, v in sorted(ex.items())},
                   "note": "exposed = time the engine's stream waited for the side stream's collectives (HIP events around the wait, "
                           "survey pass of %d steps); the collectives are sync ops enqueued on the engine's side stream: out = output "
                           "layer (behind its gradient kernels, beside the BPTT chain), rec = the recurrent part once scatter-add, "
                           "slab reduction and the chain's partial sums are in; sparse = row-sparse blocks (engine stream)" % n_sv}
    eng.enable_timing(True, only=dom_phase)

    def timed_region(first, n=None):
        n = args.steps if n is None else n
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for i in range(n):
            step(first + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt
    # --repeats regions of EXACTLY --steps steps each, every one bracketed by barrier + synchronize; the headline is the
    # median region (a DVFS blip in one region cannot move it), all regions are reported
    region_s = [timed_region(args.warmup + r * args.steps) for r in range(max(1, args.repeats))]
    dt = float(np.median(region_s))
    log("timed regions done: %s ms/step" % ", ".join("%.4f" % (x / args.steps * 1e3) for x in region_s))
    cost = eng.read_cost()
    if not np.isfinite(cost):
        raise ValueError("Cost is NaN")            # rnn_base.py:291-292
    # one long region behind the short ones: the timed regions are ~20 ms each, which the chip runs at its boost clock; this one
    # lasts --sustained-seconds (same steps, same barriers), with the chain's own clock read from its in-kernel counters
    sustained = None
    if args.sustained_seconds > 0 and not args.quick:
        try:
            timed_dom = eng.phase_times()              # (the dominant kernel's HIP-event time belongs to the timed regions above)
        except Exception:
            timed_dom = None
        n_sus = max(args.steps, int(args.sustained_seconds / (dt / args.steps)))
        eng.enable_timing(False)
        sdt = timed_region(args.warmup + max(1, args.repeats) * args.steps, n_sus)
        sustained = {"steps": n_sus, "seconds": round(sdt, 3), "ms_per_step": round(sdt / n_sus * 1e3, 4),
                     "value": round(Bg * n_sus / sdt, 1), "unit": "user-sequences/s"}
        try:
            if eng.query("tail_chunks") >= 2:
                cyc, ticks = eng.query("tail_chain_cycles"), eng.query("tail_chain_ticks")
                if ticks > 0:
                    sustained["rec_bwd_shader_mhz"] = round(cyc / ticks * 100.0, 1)
        except Exception:
            pass
        log("sustained region: %d steps, %.4f ms/step" % (n_sus, sdt / n_sus * 1e3))

    try:        # HIP-event time of the dominant kernel over the timed region (data-parallel: all-reduce waits sit inside it)
        timed = timed_dom if (sustained is not None and timed_dom is not None) else eng.phase_times()
        phases = dict(survey) if survey is not None else None
        if phases is not None:
            phases[dom_phase] = timed[dom_phase]
    except Exception:
        phases = None
    ms_per_step = dt / args.steps * 1e3
    value = Bg * args.steps / dt

    result = {
        "metric": "user-sequences/sec training (ML-1M shape, seq200 b256) at 1/2/4/8 GPUs",
        "value": round(value, 1), "unit": "user-sequences/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling,
        "repeats": {"n": len(region_s), "ms_per_step": [round(x / args.steps * 1e3, 4) for x in region_s],
                    "stddev_ms": round(float(np.std([x / args.steps * 1e3 for x in region_s])), 5),
                    "headline": "median region of --steps steps"},
        "vs_baseline": None, "dtype": "f32" if not (args.flags & 384) else "f32 with bf16-input GEMMs (flags %d)" % args.flags, "data": "synthetic",
        "config": {"workload": "%s: train.py -m RNN --r_t %s --r_l %s --max_length %d -b %d --loss %s --u_m adam, "
                               "N=%d items, Zipf(1.0) ids, lengths=%s, %d rows per GPU"
                               % (args.config, cell, "-".join(map(str, layers)), T, B, loss, n_items, args.lengths, B),
                   "global_batch": Bg, "seq_len": T, "parallelism": "dp%d" % world, "last_cost": round(cost, 5),
                   "arithmetic": arithmetic_note(eng)},
    }

    if sustained is not None:
        result["sustained"] = sustained
    if dp_info is not None:
        result["data_parallel"] = dp_info
    if rank == 0 and phases is not None:
        G = {"LSTM": 4, "GRU": 3, "Vanilla": 1}[cell]
        H = layers[0]
        Ltot = float(np.mean([hb["lengths"].sum() for hb in host_batches]))     # valid (t,row) positions per step
        rec_flops = sum(2.0 * Ltot * Hl * G * Hl for Hl in layers)                # the chain kernels of ALL layers, per direction
        row_bytes = G * H * 4.0
        kernels = {
            "gather": {"bound": "hbm", "alg": Ltot * (row_bytes * 2 + 4), "unit": "GB/s"},       # read row + write xt
            "rec_fwd": {"bound": "mfma", "alg": rec_flops, "unit": "TFLOP/s"},
            "rec_bwd": {"bound": "mfma", "alg": rec_flops, "unit": "TFLOP/s"},
            "scatter": {"bound": "hbm", "alg": Ltot * (row_bytes * 2 + 4), "unit": "GB/s"},      # read dxt + add row
        }
        if eng.query("fused_gather"):      # the rows are gathered inside rec_fwd: this phase is only the gradient memset
            del kernels["gather"]
        try:
            tail_chunks = eng.query("tail_chunks")
        except Exception:
            tail_chunks = 0
        if tail_chunks >= 2:
            try:      # effective shader clock of the BPTT chain while its consumers run beside it (in-kernel counters of the last launch)
                cyc, ticks = eng.query("tail_chain_cycles"), eng.query("tail_chain_ticks")
                if ticks > 0:
                    kernels["rec_bwd"]["shader_mhz_beside_consumers"] = round(cyc / ticks * 100.0, 1)
            except Exception:
                pass
            # overlapped tail: the scatter phase of the step is the LAST of tail_chunks time chunks (the others ran beside the BPTT
            # chain on the side stream); its entries are 1 / tail_chunks of the batch, and it re-reads the rows it adds to
            try:
                last = eng.query("tail_last_steps")
            except Exception:
                last = 0
            frac = (last / float(T)) if last > 0 else 1.0 / tail_chunks
            kernels["scatter"]["alg"] *= frac
            kernels["scatter"]["chunks"] = tail_chunks
            kernels["scatter"]["note"] = ("overlapped tail: the phase is what is left of the scatter-add at the chain's end -- time chunk 0 "
                                          "of %d (%s of %d time steps; the others run beside rec_bwd); bytes = that chunk's dxt rows + "
                                          "the rows it adds to" % (tail_chunks, last if last > 0 else "1 / %d" % tail_chunks, T))
        for k, v in kernels.items():
            us = phases[k]
            if chain is not None and k in ("rec_fwd", "rec_bwd") and (len(layers) > 1 or k != dom_phase):
                # stacked layers: the phase also holds the dense GEMMs between the layers; the roofline prices the chain KERNELS
                # (sum over the layers' launches, HIP-event pairs around each: sbr_chain_times), the phase time is kept beside it
                v["phase_us"] = round(us, 2)
                us = chain[k]
            peak = HBM_PEAK_GBS if v["bound"] == "hbm" else F32_MFMA_PEAK_TFLOPS
            ach = (v["alg"] / (us * 1e-6)) / (1e9 if v["bound"] == "hbm" else 1e12) if us > 0 else 0.0
            v.update(us=round(us, 2), achieved=round(ach, 3), peak=peak, frac=round(ach / peak, 5))
            del v["alg"]
        dom = dom_phase if dom_phase in kernels else max(kernels, key=lambda k: phases[k])
        d = kernels[dom]
        # what the recurrent kernels put on the matrix pipe: every f32 product is `products` bf16 / fp16 MFMA terms, on 16-column
        # tiles of which `rows` columns are live batch rows -- against the dense bf16 peak of the WHOLE chip; and the CUs the
        # launch can occupy at all (one workgroup per CU)
        for k, sfx in (("rec_fwd", "fwd"), ("rec_bwd", "bwd")):
            try:
                prod, rows, wgs = (eng.query("rec_%s_%s" % (q, sfx)) for q in ("products", "rows", "workgroups"))
            except Exception:
                continue
            v = kernels[k]
            v["active_cus"] = min(N_CUS, wgs)
            if prod > 0 and v["us"] > 0:
                issued = rec_flops * prod * 16.0 / rows
                v["matrix_pipe"] = {"issued_tflops": round(issued / (v["us"] * 1e-6) / 1e12, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                                    "frac": round(issued / (v["us"] * 1e-6) / 1e12 / BF16_MFMA_PEAK_TFLOPS, 5),
                                    "frac_of_active_cus": round(issued / (v["us"] * 1e-6) / 1e12 / (BF16_MFMA_PEAK_TFLOPS * min(N_CUS, wgs) / N_CUS), 5),
                                    "terms_per_f32_product": prod, "live_rows_of_16": rows}
        # HBM bytes per launch of the dominant kernel: only from a counter pass of this same command (rocprofv3 --pmc in its own
        # run, MI355X_MICROARCH.md; tools/pmc_summary.py writes the file) -- never from a stored number
        traffic, traffic_src = None, None
        if args.pmc_json:
            try:
                pmc = json.load(open(args.pmc_json))["kernels"]
                key = {"rec_fwd": "rec_fwd_", "rec_bwd": "rec_bwd_", "gather": "gather_xt_kernel", "scatter": "scat_"}[dom]
                traffic = next(v["hbm_bytes_per_launch"] for k, v in pmc.items() if key in k)
                traffic_src = os.path.relpath(args.pmc_json, ROOT)
            except Exception as ex:
                log("pmc json unusable:", ex)
        elif world == 1 and not args.quick and not args.no_pmc:
            # ... which bench.py runs itself: two rocprofv3 --pmc child passes of this very workload (counter_passes above)
            n_child = 6 + 2
            child = ["--steps", "6", "--warmup", "2", "--repeats", "1", "--quick", "--config", args.config, "--lengths", args.lengths,
                     "--batch", str(B), "--max_length", str(T)]
            pmc = counter_passes(child, log)
            if pmc:
                key = {"rec_fwd": "rec_fwd_", "rec_bwd": "rec_bwd_", "gather": "gather_xt_kernel", "scatter": "scat_"}[dom]
                hit = [v for k, v in pmc.items() if key in k]
                if hit:
                    traffic = hit[0]["hbm_bytes"]
                    traffic_src = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of this command (--quick --steps 6, "
                                   "SBR_TAIL_OVERLAP=2: counter passes serialise kernels), mean over the launches after the first; "
                                   "2 x FETCH_SIZE (gfx950) + WRITE_SIZE")
                # the whole step: every kernel's bytes x its launches per step, against the step's algorithmic bytes
                per_step = sum(v["hbm_bytes"] * v["launches"] / float(n_child) for v in pmc.values())
                n_par = float(sum(int(np.prod(p.shape)) for p in params))
                alg = Ltot * row_bytes + Ltot * row_bytes + 2 * 4 * Ltot + 8 * 4 * n_par
                top = sorted(pmc.items(), key=lambda kv: -kv[1]["hbm_bytes"] * kv[1]["launches"])[:6]
                result["hbm_traffic"] = {
                    "bytes_per_step": int(per_step), "algorithmic_bytes_per_step": int(alg), "ratio": round(per_step / alg, 3),
                    "achieved_GBps_over_the_step": round(per_step / (ms_per_step * 1e-3) / 1e9, 1), "peak_GBps": HBM_PEAK_GBS,
                    "largest": {k.split("(")[0][-60:]: {"MB_per_launch": round(v["hbm_bytes"] / 1e6, 2),
                                                        "launches_per_step": round(v["launches"] / float(n_child), 2)} for k, v in top},
                    "note": "algorithmic = W_in rows gathered (L x G*H*4) + dxt rows into the scatter-add (the same) + ids + the dense "
                            "optimizer pass (8 arrays of P floats: p, g, m, v read; p, m, v written, g cleared); counters: same child "
                            "passes as roofline.traffic"}
        result["roofline"] = {"kernel": dom, "bound": d["bound"], "achieved": d["achieved"], "peak": d["peak"],
                              "unit": d["unit"], "frac": d["frac"], "traffic": traffic, "traffic_source": traffic_src,
                              "launch_us": d["us"], "matrix_pipe": d.get("matrix_pipe"), "active_cus": d.get("active_cus"),
                              "note": "achieved = algorithmic f32 FLOPs 2*L*H*G*H of the chain / HIP-event time of the kernel over "
                                      "the timed regions, vs the f32 MFMA peak; matrix_pipe = the bf16 / fp16 MFMA flops the kernel "
                                      "really issues vs the 2.5 PF dense peak; the chain is 2*T dependent steps on active_cus CUs "
                                      "(DESIGN.md section 3)"}
        if traffic is not None and dom in ("rec_fwd", "rec_bwd"):
            # the chains' algorithmic bytes: forward = W_in rows in (fused gather) + hs and the saved gate values out; backward = those
            # back in + dxt (and GRU's compact candidate slice of dhi) out
            nsave = {"LSTM": 5, "GRU": 4, "Vanilla": 0}[cell] + 1
            alg_k = Ltot * (row_bytes + nsave * H * 4.0) if dom == "rec_fwd" else Ltot * (nsave * H * 4.0 + row_bytes + (H * 4.0 if cell == "GRU" else 0.0))
            result["roofline"]["traffic_over_algorithmic"] = round(traffic / alg_k, 3)
            result["roofline"]["algorithmic_bytes"] = int(alg_k)
        if chain is not None:
            result["chains"] = {"rec_fwd_us": round(chain["rec_fwd"], 2), "rec_bwd_us": round(chain["rec_bwd"], 2),
                                "launches_per_step": chain["launches_per_step"],
                                "outside_chains_us": round(ms_per_step * 1e3 - chain["rec_fwd"] - chain["rec_bwd"], 1),
                                "note": "device time of the recurrent chain kernels alone (HIP-event pair around every launch, all layers; "
                                        "survey pass), and what is left of ms_per_step outside them"}
        if world == 1 and not args.quick and not args.no_pmc:
            mc = mfma_counter_pass(["--steps", "6", "--warmup", "2", "--repeats", "1", "--quick", "--config", args.config, "--lengths", args.lengths,
                                    "--batch", str(B), "--max_length", str(T)], log)
            if mc:
                keep = {}
                for k, v in mc.items():
                    if "rec_fwd" in k or "rec_bwd" in k or "gemm_x6_kernel" in k or "gemm_kernel" in k or "head_cce" in k or "out_grad_step" in k:
                        name = k.split("(")[0].replace("void ", "")[-70:]
                        if "rec_" in k:
                            try:
                                wgs = eng.query("rec_workgroups_fwd" if "rec_fwd" in k else "rec_workgroups_bwd")
                                v["mfma_util_of_active_cus"] = round(v["mfma_util_of_chip"] * N_CUS / min(N_CUS, wgs), 5)
                            except Exception:
                                pass
                        keep[name] = v
                result["mfma_counters"] = {"kernels": keep,
                                           "note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES child pass of this command (--quick --steps 6, "
                                                   "SBR_TAIL_OVERLAP=2), mean over the launches after the first; mfma_util_of_chip = MFMA_BUSY / (1024 SIMDs x "
                                                   "SQ_BUSY / 32 SQ instances); the output projection: where the one-launch head runs (C1 / C2) logits and dh are "
                                                   "inside head_cce_kernel and dW_out inside out_grad_step_kernel, both on the exact-f32 matrix instruction; "
                                                   "elsewhere the logits GEMM is the gemm_x6_kernel<4, false, 4, false, ...> launch, its backward pair the "
                                                   "<.., true, ..> ones (at C2 the remaining gemm_x6_kernel launch is the polling dW_hid GEMM)"}
        result["phases_us"] = {k: round(v, 2) for k, v in phases.items() if k != "total"}
        result["phases_us"]["note"] = ("%s: HIP events over the timed region; the other phases: survey pass of %d steps with "
                                       "every phase bracketed (those event records lengthen a step, so the phases do not add "
                                       "up to ms_per_step)" % (dom_phase, min(args.steps, 20)))
        # the embedding gather (north_star: achieved HBM GB/s on gather AND scatter).  In the step the rows are read inside
        # rec_fwd (no xt array): its algorithmic bytes over the forward kernel's time is a LOWER bound of the rate the gather runs
        # at; the stand-alone gather_xt_kernel (the step with SBR_FUSE_GATHER=0) is timed on the same shape beside it.
        if "gather" not in kernels and not args.quick:
            gb = Ltot * (row_bytes + 4)
            kernels["gather_fused"] = {"bound": "hbm", "unit": "GB/s", "us": kernels["rec_fwd"]["us"],
                                       "achieved": round(gb / (kernels["rec_fwd"]["us"] * 1e-6) / 1e9, 3), "peak": HBM_PEAK_GBS,
                                       "frac": round(gb / (kernels["rec_fwd"]["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                       "note": "rows read inside rec_fwd, hidden under the chain: bytes / rec_fwd time"}
            if world == 1:
                try:
                    os.environ["SBR_FUSE_GATHER"] = "0"
                    e2 = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=T, batch_size=B, loss=loss,
                                   n_samples=n_samples, updater="adam", learning_rate=1e-3)
                    try:
                        e2.set_all_param_values(params)
                        d0 = dev_batches[0]
                        e2.set_batch_device(d0["X"], d0["lengths"], d0["target"], d0["samples"] if loss != "CCE" else None, d0["pop"], B)
                        for _ in range(3):
                            e2.train_step(sync=False)
                        e2.enable_timing(True, only="gather")
                        for _ in range(10):
                            e2.train_step(sync=False)
                        us = e2.phase_times()["gather"]
                    finally:
                        e2.close()
                        del os.environ["SBR_FUSE_GATHER"]
                    gb2 = Ltot * (row_bytes * 2 + 4)       # read the row, write xt
                    kernels["gather_unfused"] = {"bound": "hbm", "unit": "GB/s", "us": round(us, 2),
                                                 "achieved": round(gb2 / (us * 1e-6) / 1e9, 3), "peak": HBM_PEAK_GBS,
                                                 "frac": round(gb2 / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                                 "note": "gather_xt_kernel alone (SBR_FUSE_GATHER=0 engine, same batch): rows in, xt out"}
                except Exception as ex:
                    log("unfused gather timing skipped:", ex)
                # ... and the scatter-add of the embedding gradient on its own (sbr_debug_scatter: the step's stand-alone form for this
                # shape over the same batch, nothing beside it): bytes = the dxt rows read + the ids + the gradient rows written
                try:
                    us_s, n_ent, n_rows_w = eng.debug_scatter(20)
                    gb_s = n_ent * (row_bytes + 4.0) + n_rows_w * row_bytes
                    kernels["scatter_unfused"] = {"bound": "hbm", "unit": "GB/s", "us": round(us_s, 2), "entries": n_ent, "rows_written": n_rows_w,
                                                  "achieved": round(gb_s / (us_s * 1e-6) / 1e9, 3), "peak": HBM_PEAK_GBS,
                                                  "frac": round(gb_s / (us_s * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                                                  "note": "the stand-alone scatter-add launch(es) of this shape alone on the chip (sort excluded): dxt rows "
                                                          "+ ids in, one gradient row per distinct id out; the step itself runs it beside the BPTT chain "
                                                          "(kernels.scatter = what is left of it behind the chain)"}
                except Exception as ex:
                    log("stand-alone scatter-add timing skipped:", ex)
        # the dense output projection on its own (logits = h . W_out^T, rnn_one_hot.py:65): the same kernel the step
        # runs, timed with HIP events on this shape (north_star: MFMA utilisation of the output projection)
        try:
            if args.quick:
                raise RuntimeError("--quick")
            import ctypes
            Hl = layers[-1]
            # operands like the step's: hidden states in [-1, 1], weights; the kernel the step's logits GEMM takes (the two-plane fp16
            # split, three MFMAs per product, unless SBR_GEMM_F16=0 or a rectified Vanilla layer keeps bf16x6: six)
            A, Bm = torch.tanh(torch.randn(B, Hl, device=dev)), 0.1 * torch.randn(n_items, Hl, device=dev)
            C = torch.empty(B, n_items, device=dev)
            proj_f16 = os.environ.get("SBR_GEMM_F16", "1") != "0" and cell != "Vanilla"
            proj_mode, proj_terms = (3, 3) if proj_f16 else (0, 6)

            def proj():
                rc = eng.lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), Hl, 1,
                                            Bm.data_ptr(), 1, Hl, C.data_ptr(), n_items, B, n_items, Hl, None, None, 0, proj_mode)
                assert rc == 0
            for _ in range(3):
                proj()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                proj()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            tf = 2.0 * B * n_items * Hl / us / 1e6
            kernels["output_projection"] = {"bound": "mfma", "unit": "TFLOP/s", "us": round(us, 2), "achieved": round(tf, 3),
                                            "peak": F32_MFMA_PEAK_TFLOPS, "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 5),
                                            "shape": "M=%d N=%d K=%d" % (B, n_items, Hl),
                                            "on_step_path": eng.query("head_fused") == 0,      # (False: the step's head is the one-launch
                                            # head_cce_kernel -- logits + softmax + dh on the exact-f32 matrix instruction, phases_us.output)
                                            "matrix_pipe": {"issued_tflops": round(proj_terms * tf, 2), "peak": BF16_MFMA_PEAK_TFLOPS,
                                                            "frac": round(proj_terms * tf / BF16_MFMA_PEAK_TFLOPS, 5),
                                                            "terms_per_f32_product": proj_terms}}

            def proj_bf16():      # SBR_FLAG_BF16_PROJECTION's kernel: plain bf16 operands, one MFMA per block
                rc = eng.lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), Hl, 1,
                                            Bm.data_ptr(), 1, Hl, C.data_ptr(), n_items, B, n_items, Hl, None, None, 0, 2)
                assert rc == 0
            for _ in range(3):
                proj_bf16()
            e0.record()
            for _ in range(20):
                proj_bf16()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / 20
            tf = 2.0 * B * n_items * Hl / us / 1e6
            hbm = (n_items * Hl + B * Hl + B * n_items) * 4.0 / (us * 1e-6) / 1e9
            kernels["output_projection_bf16"] = {"bound": "mfma", "unit": "TFLOP/s", "us": round(us, 2), "achieved": round(tf, 3),
                                                 "peak": BF16_MFMA_PEAK_TFLOPS, "frac": round(tf / BF16_MFMA_PEAK_TFLOPS, 5),
                                                 "hbm_gbs": round(hbm, 1), "hbm_frac": round(hbm / HBM_PEAK_GBS, 5),
                                                 "shape": "M=%d N=%d K=%d" % (B, n_items, Hl),
                                                 "note": "f32 operands converted on the way to LDS: W_out in + logits out at f32 width "
                                                         "bound it by HBM long before the matrix pipe"}
        except Exception as ex:      # never let the side measurement break the bench line
            log("output projection timing skipped:", ex)
        result["kernels"] = kernels

    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.quick:
        from oracle import rnn_oracle as O          # the oracle: only this leg of the bench touches it
        from oracle import torch_ref as R
        ncores = os.cpu_count() or 1
        nthr = args.cpu_threads or min(ncores, 16)
        cfg = dict(cell=cell, layers=layers, loss=loss, regularization=0.0)
        hb = host_batches[0]
        cb = dict(X=hb["X"], mask=hb["mask"], target=hb["target"], samples=hb["samples"], pop=hb["pop"])
        brief_cpu = bool(args.brief)
        budget = args.cpu_seconds

        def timed(tr, counts, warmups, t_all):
            """best (threads, s/step, steps) over `counts`: `warmups` untimed steps at the first count, then timed steps while the budget lasts"""
            torch.set_num_threads(counts[0])
            t0 = time.perf_counter()
            for _ in range(warmups):
                tr.train_function(cb)
            warm = (time.perf_counter() - t0) / max(1, warmups)
            sweep, skipped, est = [], [], warm
            for c in counts:
                if sweep and (time.perf_counter() - t_all) + est * (sweep[-1][0] / float(c)) * 0.7 > budget:
                    skipped.append(c)
                    continue
                torch.set_num_threads(c)
                n, t0 = 0, time.perf_counter()
                while n < args.cpu_steps and (n == 0 or (time.perf_counter() - t_all) + (time.perf_counter() - t0) / n < budget):
                    tr.train_function(cb)
                    n += 1
                    if c != counts[0]:
                        break                      # one step for the narrower counts
                est = (time.perf_counter() - t0) / n
                sweep.append((c, est, n))
            return sweep, skipped, warm

        counts = [nthr] if (args.cpu_threads or brief_cpu) else [c for c in (16, 8, 4, 1) if c <= nthr] or [nthr]
        t_all = time.perf_counter()
        # (1) the time loop under torch.jit.script, selects / slices replaced by unbind / chunk (oracle/torch_ref.py): the figure quoted
        sweep_s, skipped_s, warm_s = timed(R.TorchTrainer(params, cfg, O.recurrent_param_shapes, updater="adam", lr=1e-3, scripted=True),
                                           counts, 2, t_all)
        nthr_s, cdt_s, n_s = min(sweep_s, key=lambda e: e[1])
        log("cpu baseline (scripted scan): " + ", ".join("%d threads %.2f s/step" % (c, t) for c, t, _ in sweep_s))
        base = {"value": round(B / cdt_s, 1), "unit": "user-sequences/s", "cores": nthr_s, "kind": "port",
                "sample": "%d train step(s) of the same %s workload (B=%d, T=%d) after two warm-up steps: torch-CPU float32 port of the "
                          "reference path with the T-step scan under torch.jit.script (oracle/torch_ref.py: layer_forward_scripted; the "
                          "reference's grad_clip nodes, inactive at these magnitudes, are not in it) -- Theano / Lasagne / python2 are not "
                          "installable here; %.2f s/step at the best thread count, %d threads of %d host cores"
                          % (n_s, args.config, B, T, cdt_s, nthr_s, ncores),
                "threads_sweep": {"s_per_step": {str(c): round(t, 3) for c, t, _ in sweep_s}, "not_run_within_budget": skipped_s,
                                  "budget_s": budget}}
        if not brief_cpu:
            # (2) the eager autograd loop the parity tests use as the independent restatement: a LOWER bound (its per-step selects
            # make the backward fill a whole-input tensor per time step), kept beside the figure above since rounds 1 - 5 quoted it
            t_all = time.perf_counter()
            sweep, skipped, warm = timed(R.TorchTrainer(params, cfg, O.recurrent_param_shapes, updater="adam", lr=1e-3), counts[:2], 1, t_all)
            nthr_e, cdt, n = min(sweep, key=lambda e: e[1])
            log("cpu baseline (eager loop): " + ", ".join("%d threads %.2f s/step" % (c, t) for c, t, _ in sweep))
            base["eager_loop"] = {"value": round(B / cdt, 1), "unit": "user-sequences/s", "cores": nthr_e, "s_per_step": round(cdt, 3),
                                  "note": "the same port with a Python loop over the T steps (what rounds 1 - 5 reported as cpu_baseline)"}
            # the reference pays its Python batch packing every iteration (rnn_one_hot.py:83-106: B*T list appends + a (B, N)
            # exclude matrix): restated literally in oracle.prepare_input_one_hot, timed on the same batch, single-threaded as there
            if loss == "CCE":
                seqs = [(0, [(int(i), 1.0) for i in hb["X"][b, :hb["lengths"][b], 0]], [(int(hb["target"][b]), 1.0)]) for b in range(B)]
                pop_table = np.ones(n_items)
                t0 = time.perf_counter()
                npk = 0
                while npk < 3 and (npk == 0 or time.perf_counter() - t0 < 5.0):
                    O.prepare_input_one_hot(seqs, T, n_items, pop_table, 0.0)
                    npk += 1
                pack = (time.perf_counter() - t0) / npk
                base["end_to_end"] = {"value": round(B / (cdt_s + pack), 1), "unit": "user-sequences/s", "packing_s_per_batch": round(pack, 4),
                                      "note": "compute step + reference-style _prepare_input packing of the batch "
                                              "(rnn_one_hot.py:83-106 restated in oracle.prepare_input_one_hot), 1 thread"}
        result["cpu_baseline"] = base
    eng.close()
    if world > 1 or args.force_dp:
        dist.destroy_process_group()
    if (rank == 0 and world == 1 and not args.quick and not args.force_dp and args.loop_iters > 0 and args.config == "c2"
            and (B, T, args.lengths) == (256, 200, "full")):
        # the training LOOP the reference runs around the step (rnn_base.py:285-300): batches built per iteration (native device
        # builder), every cost read back (one iteration late: sbr_train_step_lagged), on an ML-1M-shaped synthetic file in the
        # reference's on-disk format -- through the train.py mirror, >= 1000 iterations
        try:
            import importlib.util
            import tempfile
            spec = importlib.util.spec_from_file_location("bench_train_loop", os.path.join(ROOT, "tools", "bench_train_loop.py"))
            btl = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(btl)
            with tempfile.TemporaryDirectory() as tmp:
                root = btl.write_dataset(os.path.join(tmp, "ds"))
                seqs, per_it = btl.run(root, True, args.loop_iters, B, T)
            result["train_loop"] = {"value": round(seqs, 1), "unit": "user-sequences/s", "ms_per_iteration": round(per_it * 1e3, 4),
                                    "iterations": args.loop_iters,
                                    "note": "python -m sbr_amd.train mirror: device batch builder + step + cost of every iteration read "
                                            "back one iteration late; ML-1M-shaped synthetic file (6040 users, ragged sequences)"}
            log("train loop: %.4f ms/iteration" % (per_it * 1e3))
        except Exception as ex:
            log("train loop leg skipped:", ex)
            result["train_loop_error"] = repr(ex)[:400]
    if (rank == 0 and world == 1 and not args.quick and not args.force_dp and not args.no_other_configs and args.config == "c2"
            and (B, T, args.lengths) == (256, 200, "full")):
        try:
            torch.cuda.empty_cache()
            result["other_configs"] = other_config_lines(log, [n for n in args.other_configs.split(",") if n in ("c1", "c3", "c4", "c5", "c5_bf16")])
        except Exception as ex:
            result["other_configs"] = {"error": repr(ex)[:300]}
        try:
            result["strong_scaling_model"] = strong_scaling_pieces(log)
        except Exception as ex:
            result["strong_scaling_model"] = {"error": repr(ex)[:300]}
    elif rank == 0 and world > 1 and args.strong_pieces and not args.quick:
        # a data-parallel line that carries the single-rank pieces of its own strong-scaling model (the process group is gone by now:
        # the children are plain one-GPU runs on rank 0's device)
        try:
            result["strong_scaling_model"] = strong_scaling_pieces(log)
        except Exception as ex:
            result["strong_scaling_model"] = {"error": repr(ex)[:300]}
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)          # (C stdio buffers of the libraries: out through the redirected descriptor)
    except Exception:
        pass
    sys.stdout.flush()
    os.dup2(json_fd, 1)
    os.close(json_fd)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
