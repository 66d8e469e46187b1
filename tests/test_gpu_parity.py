"""GPU parity tests proper: the HIP path, called through the C-ABI, against the CPU oracle on the
same seeded inputs (sizes the oracle finishes in seconds), plus size-independent properties at
BASELINE.json's full sizes.  Tolerances: logits / hidden state 1e-3 relative (north_star), gradients
and updated parameters 1e-4 relative, top-k ids exact."""
import numpy as np
import pytest

import parity_util as PU

pytestmark = pytest.mark.gpu


def check(r, steps=2, tol_h=1e-4, tol_g=1e-4):
    assert r["param_roundtrip"] == 0.0
    assert r["h_last"] <= tol_h, r
    assert r["cost"] <= 1e-5, r
    assert r["grad_worst"] <= tol_g, sorted(((v, k) for k, v in r.items() if k.startswith("grad:")), reverse=True)[:4]
    PU.params_ok(r, steps, bar=1e-3, tol_g=tol_g)
    assert r["predict_scores"] <= 1e-3, r
    assert r["topk_mismatch"] == 0, r


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
@pytest.mark.parametrize("H", [4, 20, 50, 128])          # Hp = 16, 32, 64, 128: register-resident W_hid kernels
def test_one_layer_cce(cell, H):
    check(PU.compare_step(cell, [H], "CCE", N=61, B=37, T=9))


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
@pytest.mark.parametrize("H", [20, 50, 128])
def test_exact_f32_mfma_kernels(cell, H):                  # SBR_FLAG_F32_MFMA: v_mfma_f32_16x16x4_f32 path
    check(PU.compare_step(cell, [H], "CCE", N=61, B=37, T=9, flags=16))


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_streamed_whid_kernels(cell):                      # Hp = 576: W_hid fragments streamed from L2 (f32 MFMA kernels)
    check(PU.compare_step(cell, [520], "CCE", N=61, B=21, T=6, scale=0.03), tol_h=2e-4)


@pytest.mark.parametrize("linear", ["0", "1"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_cluster_kernels_512_wide(cell, linear, monkeypatch):
    # Hp = 512 (config C5's width): clusters of 32 workgroups (one unit tile each, K in four parts) on 8-row tiles (the
    # backward with bf16x6 products: 4-row tiles); linear=1 spreads a cluster over all XCDs
    monkeypatch.setenv("SBR_CL_LINEAR", linear)
    check(PU.compare_step(cell, [512], "CCE", N=61, B=37, T=7, scale=0.04), tol_h=2e-4)


def test_cluster_kernels_512_padded_two_layers_ragged():
    check(PU.compare_step("LSTM", [300, 300], "CCE", N=41, B=19, T=20, seed=3, scale=0.04), tol_h=2e-4)


def test_width_between_128_and_256_takes_the_cluster_kernels():
    check(PU.compare_step("GRU", [160], "CCE", N=61, B=21, T=8), tol_h=2e-4)


@pytest.mark.parametrize("rpt", ["1", "2", "8", "16"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_rows_per_workgroup_variants(cell, rpt, monkeypatch):
    # default: 4-row tiles with the gate math split over the duplicate MFMA columns (rec_*_x6s, every other test);
    # the general kernels (rec_*_x6) serve the other tile heights
    monkeypatch.setenv("SBR_RPT", rpt)
    check(PU.compare_step(cell, [50], "CCE", N=61, B=37, T=9))


def test_unfused_gather_agrees(monkeypatch):
    # default: layer 0 with one index per step gathers W_in rows inside the forward kernel; the separate gather
    # kernel (also used for --rf / F > 1, the f32 and the cluster kernels) must give the same step
    monkeypatch.setenv("SBR_FUSE_GATHER", "0")
    check(PU.compare_step("GRU", [128], "CCE", N=61, B=37, T=9))
    check(PU.compare_step("LSTM", [50], "CCE", N=61, B=37, T=9))


def test_benchmark_configs_take_their_fast_kernels():
    # a silent fallback to the barrier kernels would keep every parity test green and lose 20-25 % of the step
    from sbr_amd.engine import RNNEngine
    for cell, layers, n_items, want in (("GRU", [128], 3706, 2), ("LSTM", [20], 3706, 3), ("GRU", [50], 3706, 3),
                                        ("LSTM", [256], 26744, 1), ("LSTM", [128], 3706, 2)):
        eng = RNNEngine(cell=cell, layers=layers, n_items=n_items, max_length=200, batch_size=256, loss="CCE")
        try:
            assert eng.query("rec_kernel") == want, (cell, layers)
            assert eng.query("fused_gather") == 1
        finally:
            eng.close()


@pytest.mark.parametrize("mode", ["0", "1"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_pipelined_kernel_modes(cell, mode, monkeypatch):
    # default (2): rec_*_x6p at 128 units: LDS counters instead of a per-step barrier + the matrix-pipe
    # gate between the two waves of a SIMD (every other 128-wide test); 1 = without the gate, 0 = the barrier kernels
    monkeypatch.setenv("SBR_X6_PIPE", mode)
    check(PU.compare_step(cell, [128], "CCE", N=61, B=37, T=9))
    check(PU.compare_step(cell, [128, 128], "CCE", N=61, B=9, T=12))


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])      # (an LSTM leaves the pipelined kernels with either switch:
def test_forward_chain_with_bf16x6_products(cell, monkeypatch):   #  four gates fit the register file as fp16 planes only)
    # default (every other 128-wide GRU / Vanilla test): the forward chain's products as a 2-way fp16 split, 3 MFMAs
    # (rec_fwd_x6p<.., F16>); SBR_X6_F16=0: the 3-way bf16 split, 6 MFMAs, as the backward chain and the GEMMs use
    monkeypatch.setenv("SBR_X6_F16", "0")
    check(PU.compare_step(cell, [128], "CCE", N=61, B=37, T=9))
    check(PU.compare_step(cell, [128, 128], "CCE", N=61, B=9, T=12, scale=0.05))


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_backward_chain_with_bf16x6_products(cell, monkeypatch):
    # default (every other 128-wide GRU / Vanilla test): the BPTT chain's products as the fp16 split too, the gradient operand
    # scaled by 2^9 (it is bounded by the clip at 100); SBR_X6_F16_BWD=0: bf16x6
    monkeypatch.setenv("SBR_X6_F16_BWD", "0")
    check(PU.compare_step(cell, [128], "CCE", N=61, B=37, T=9))
    check(PU.compare_step(cell, [128, 128], "CCE", N=61, B=9, T=12, scale=0.05))


def test_fp16_backward_products_at_the_clip_boundary_and_with_tiny_gradients():
    # gate gradients driven beyond +-100 (clip active: the scaled operand reaches 51200 of fp16's 65504), and a batch whose
    # gradients are ~1e-9 (popularity weights of 1e4: the absolute floor of the scaled split is 3e-14)
    check(PU.compare_step("GRU", [128], "CCE", N=61, B=37, T=9, popscale=1e-4))
    check(PU.compare_step("Vanilla", [128], "CCE", N=61, B=37, T=9, popscale=1e-4))
    check(PU.compare_step("GRU", [128], "CCE", N=61, B=37, T=40, popscale=1e4, scale=0.1), tol_g=2e-4)
    check(PU.compare_step("GRU", [128, 128], "BPR", N=61, B=9, T=12, S=8, scale=0.05))      # dense lower layer: dh_ext every step
    check(PU.compare_step("LSTM", [128], "CCE", N=61, B=37, T=9, popscale=1e-4))
    check(PU.compare_step("LSTM", [128], "CCE", N=61, B=37, T=40, popscale=1e4, scale=0.1), tol_g=2e-4)
    check(PU.compare_step("LSTM", [128, 128], "BPR", N=61, B=9, T=12, S=8, scale=0.05))


def test_fp16_forward_products_keep_f32_accuracy_over_long_chains():
    # the forward state after 200 dependent steps: as close to the float64 oracle as the bf16x6 / f32 kernels get
    # (tolerances of the other tests), with weights up to |w| ~ 1.5 and initial states beyond 1
    r = PU.compare_step("GRU", [128], "CCE", N=300, B=8, T=200, scale=0.2)
    assert r["h_last"] < 5e-5 and r["grad_worst"] < 2e-4 and r["topk_mismatch"] == 0, r
    r = PU.compare_step("GRU", [64, 128], "CCE", N=61, B=9, T=1, scale=0.3)      # the case that exposed a double rounding
    assert r["h_last"] < 2e-6 and r["grad_worst"] < 3e-6, r


def _tail_chunks(cell, T, N=300, B=37, flags=0, loss="CCE", S=0, H=128):
    from sbr_amd.engine import RNNEngine
    eng = RNNEngine(cell=cell, layers=[H], n_items=N, max_length=T, batch_size=B, loss=loss, n_samples=S, flags=flags)
    try:
        return eng.query("tail_chunks")
    finally:
        eng.close()


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_overlapped_step_tail(cell):
    # One index-input layer of 128 units and T >= 64: the BPTT chain stores dxt / dhi write-through and publishes its progress;
    # dW_hid GEMM and scatter-add of every finished chunk of time steps run beside it (sbr_backward_recurrent).  Ragged rows
    # (whole tiles masked for most of the time axis), Zipf ids (long runs of equal ids across the time chunks: the scatter
    # of a chunk ADDS to rows earlier chunks wrote), T not a multiple of the chunk length.
    assert _tail_chunks(cell, 70) == 4
    # (gap: the ranked ids are compared on the rows whose oracle logits are further apart than 1e-4 -- most of them; a tanh-only
    # layer of 128 units is ill-conditioned over 70+ steps at larger weights, see parity_util.build_case)
    sc = 0.05 if cell == "Vanilla" else 0.1
    check(PU.compare_step(cell, [128], "CCE", N=300, B=37, T=70, scale=sc, zipf=True, gap=1e-4), tol_g=2e-4)
    if cell != "Vanilla":      # (a tanh-only layer over 131 steps + Adam is chaotic with the tail switched off as well: tools/dbg_tail.py)
        # LSTM: 131 steps from the loss the initial states' gradients are 5e-10 / 8e-10 (every other array: 6e-4 .. 4e-2), sums of
        # per-step terms of ~1e-11 -- where the fp16 split of the chain's gradient operand has reached its absolute floor of
        # 6e-14 per element (rec_bwd_x6p).  They come out 1.4e-12 off (2e-3 of themselves, the same with the tail switched
        # off; 1e-6 with SBR_X6_PIPE=0), so arrays that small are held to tol_g x 2e-8 = 4e-12 absolute here.
        # (GRU: 1.5e-4 of itself at the first step, 5e-4 at the second of the same run -- the same floor: both cells take it)
        check(PU.compare_step(cell, [128], "CCE", N=300, B=64, T=131, scale=sc, full=True, gap=1e-4, grad_floor=2e-8), tol_g=2e-4)
    else:                  # ... and Adam's normalised steps amplify it already at 80: momentum steps for this one
        check(PU.compare_step(cell, [128], "CCE", N=300, B=64, T=80, scale=sc, full=True, gap=1e-4, updater="nesterov"), tol_g=2e-4)
    check(PU.compare_step(cell, [128], "CCE", N=40, B=5, T=64, scale=sc), tol_g=2e-4)          # one row tile, few ids, many duplicates


def test_overlapped_step_tail_with_sampled_head_and_other_updaters():
    assert _tail_chunks("GRU", 70, flags=64, loss="BPR", S=8) == 4       # SBR_FLAG_DENSE_UPDATE: no row-sparse blocks
    check(PU.compare_step("GRU", [128], "BPR", N=300, B=37, T=70, S=8, scale=0.1, flags=64, gap=1e-4), tol_g=2e-4)
    check(PU.compare_step("GRU", [128], "CCE", N=300, B=37, T=70, scale=0.1, updater="adagrad", gap=1e-4), tol_g=2e-4)
    check(PU.compare_step("Vanilla", [128], "CCE", N=300, B=37, T=70, scale=0.05, updater="nesterov", gap=1e-4), tol_g=2e-4)


def test_overlapped_step_tail_switched_off_agrees(monkeypatch):
    monkeypatch.setenv("SBR_TAIL_OVERLAP", "0")
    assert _tail_chunks("GRU", 70) == 0
    check(PU.compare_step("GRU", [128], "CCE", N=300, B=37, T=70, scale=0.1, zipf=True, gap=1e-4), tol_g=2e-4)


def test_overlapped_step_tail_kernels_on_one_stream(monkeypatch):
    monkeypatch.setenv("SBR_TAIL_OVERLAP", "2")      # what the counter passes of tools/profile_round.sh run
    assert _tail_chunks("GRU", 70) == 4
    check(PU.compare_step("GRU", [128], "CCE", N=300, B=37, T=70, scale=0.1, zipf=True, gap=1e-4), tol_g=2e-4)


@pytest.mark.parametrize("env", [
    {"SBR_TAIL_SCATTER_LDS": "0"},                                  # the scatter-add with one global atomic per piece and row (rounds 2 / 3a)
], ids=lambda e: ",".join("%s=%s" % (k[9:], v) for k, v in e.items()))
def test_overlapped_step_tail_consumer_shapes(env, monkeypatch):
    # the other scatter-add of the overlapped tail (the form a step falls back to when the LDS rows of the default one run out):
    # same gradients, same updated parameters  (the consumers' shape parameters -- units, groups, slab table -- were environment
    # switches while they were tuned, rounds 3 - 5; they are constants now)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert _tail_chunks("GRU", 70) == 4
    check(PU.compare_step("GRU", [128], "CCE", N=300, B=37, T=70, scale=0.1, zipf=True, gap=1e-4), tol_g=2e-4)
    check(PU.compare_step("Vanilla", [128], "CCE", N=40, B=5, T=64, scale=0.05), tol_g=2e-4)     # 40 ids, 320 entries: every id is hot


def test_overlapped_step_tail_is_not_taken_where_it_does_not_apply():
    assert _tail_chunks("GRU", 40) == 0                                   # short sequences
    assert _tail_chunks("LSTM", 70, H=50) == 0                            # the 128-unit kernels only
    assert _tail_chunks("GRU", 70, flags=32) == 0                         # row-sparse blocks forced
    assert _tail_chunks("GRU", 70, N=20000) == 0                          # key space beyond the LDS histogram: chunks < 2


@pytest.mark.parametrize("H", [20, 50])
@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_small_layer_barrier_kernels_agree(cell, H, monkeypatch):
    # default for Hp = 32 / 64: rec_*_x6q (LDS counter instead of a barrier, every other small-layer test);
    # SBR_X6_PIPE=0 keeps the barrier kernels rec_*_x6s
    monkeypatch.setenv("SBR_X6_PIPE", "0")
    check(PU.compare_step(cell, [H], "CCE", N=61, B=37, T=9))


@pytest.mark.parametrize("which", ["SBR_X6_F16", "SBR_X6_F16_BWD"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_small_layer_kernels_with_bf16x6_products(cell, which, monkeypatch):
    # default (every other small-layer test): both chains of rec_*_x6q on the 2-way fp16 split, three MFMAs per product;
    # the switches bring back the three bf16 planes
    monkeypatch.setenv(which, "0")
    check(PU.compare_step(cell, [20], "CCE", N=61, B=37, T=9))
    check(PU.compare_step(cell, [50, 20], "CCE", N=61, B=9, T=12, scale=0.1))


def test_small_layer_fp16_products_at_the_clip_boundary_and_with_tiny_gradients():
    check(PU.compare_step("LSTM", [20], "CCE", N=61, B=37, T=9, popscale=1e-4))
    check(PU.compare_step("GRU", [50], "CCE", N=61, B=37, T=40, popscale=1e4, scale=0.1), tol_g=2e-4)
    check(PU.compare_step("LSTM", [50, 20], "BPR", N=61, B=9, T=12, S=8, scale=0.1))


def test_small_layer_kernels_two_layers_ragged_and_chunks(monkeypatch):
    monkeypatch.setenv("SBR_BWD_CHUNKS", "3")
    check(PU.compare_step("LSTM", [50, 20], "CCE", N=61, B=7, T=70, scale=0.05))
    check(PU.compare_step("GRU", [20, 50], "Blackout", N=61, B=21, T=33, S=8))


def test_pipelined_kernels_long_ragged_rows_and_chunks(monkeypatch):
    # rows of every length in one tile (masked tail, carried state), BPTT in time chunks, dense second layer (dh_ext)
    monkeypatch.setenv("SBR_BWD_CHUNKS", "3")
    check(PU.compare_step("GRU", [128, 128], "CCE", N=61, B=7, T=70, scale=0.05))
    check(PU.compare_step("Vanilla", [128], "Blackout", N=61, B=21, T=33, S=8))
    check(PU.compare_step("LSTM", [128, 128], "CCE", N=61, B=7, T=70, scale=0.05))
    check(PU.compare_step("LSTM", [128], "Blackout", N=61, B=21, T=33, S=8))


@pytest.mark.parametrize("linear", ["0", "1"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_cluster_kernels_wide_layers(cell, linear, monkeypatch):
    # Hp = 256 (configs C3/C4): 8 workgroups per 8-row tile exchange h_t / dhi_t through hs / dxt with
    # sentinel polling.  linear=1 deliberately puts the members of a cluster on different XCDs (different L2s):
    # the agent-scope exchange must stay coherent there too.
    monkeypatch.setenv("SBR_CL_LINEAR", linear)
    check(PU.compare_step(cell, [256], "CCE", N=61, B=37, T=9), tol_h=2e-4)


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_cluster_kernels_on_eight_row_tiles(cell, monkeypatch):
    # default (every other wide-layer test): rec_*_c16 -- 16-row tiles, 16 units per workgroup, h exchanged pre-split, the
    # backward exchanging partial sums through a ring of sentinel blocks; SBR_CL16=0: the 8-row kernels with fp16 products
    from sbr_amd.engine import RNNEngine
    for env, rows in (("1", 16), ("0", 8)):
        monkeypatch.setenv("SBR_CL16", env)
        eng = RNNEngine(cell=cell, layers=[256], n_items=61, max_length=9, batch_size=37, loss="CCE")
        try:
            assert eng.query("rec_rows_fwd") == rows and eng.query("rec_rows_bwd") == rows
        finally:
            eng.close()
    check(PU.compare_step(cell, [256], "CCE", N=61, B=37, T=9), tol_h=2e-4)
    # (rectifying upper layer of a Vanilla stack: Adam's normalised steps amplify rounding beyond any fixed tolerance; momentum steps)
    check(PU.compare_step(cell, [512, 300], "CCE", N=41, B=19, T=12, scale=0.04, updater="nesterov" if cell == "Vanilla" else "adam"),
          tol_h=2e-4)


@pytest.mark.parametrize("which", ["SBR_X6_F16", "SBR_X6_F16_BWD"])
def test_cluster_kernels_with_bf16x6_products(which, monkeypatch):
    # either switch off: the 8-row cluster kernels (the 16-row ones exist on fp16 planes only), that chain with three bf16
    # planes (W_hid plane 3 in LDS; Hp = 512 backward on 4-row tiles), the other with the 2-way fp16 split (rec_*_cl<.., F16>)
    monkeypatch.setenv(which, "0")
    check(PU.compare_step("LSTM", [256], "CCE", N=61, B=37, T=9), tol_h=2e-4)
    check(PU.compare_step("GRU", [512], "CCE", N=61, B=37, T=7, scale=0.04), tol_h=2e-4)


def test_cluster_kernels_fp16_products_with_active_clip_and_tiny_gradients():
    check(PU.compare_step("LSTM", [256], "CCE", N=61, B=21, T=9, popscale=1e-4), tol_h=2e-4)
    check(PU.compare_step("GRU", [256], "CCE", N=61, B=21, T=30, popscale=1e4, scale=0.05), tol_h=2e-4, tol_g=2e-4)
    # (gradients 1e4 times the usual: Adam's eps no longer damps the elements whose gradient is noise beside the largest of
    # their array, and their sign decides a whole step -- the updated parameters are not compared here)
    r = PU.compare_step("LSTM", [512], "CCE", N=61, B=13, T=7, popscale=1e-4, scale=0.04)
    assert r["h_last"] <= 2e-4 and r["cost"] <= 1e-5 and r["grad_worst"] <= 1e-4 and r["topk_mismatch"] == 0, r


def test_cluster_kernels_padded_width_and_long_ragged_rows():
    # H = 200 pads to 256; T = 40 with ragged lengths (rows of a tile finish at different steps, tiles of
    # padding rows finish at once); 2-layer stack: the lower layer receives dh_ext every step.
    # scale 0.05: with N(0, 0.3) recurrent weights a 256-wide layer has spectral radius ~5 and 40 steps amplify
    # float32 rounding beyond any fixed tolerance (in every kernel variant alike)
    check(PU.compare_step("LSTM", [200], "CCE", N=41, B=19, T=40, seed=3, scale=0.05), tol_h=2e-4)
    check(PU.compare_step("GRU", [256, 256], "CCE", N=41, B=9, T=12, seed=4, scale=0.05), tol_h=2e-4)


def test_cluster_and_streamed_kernels_agree(monkeypatch):
    # same wide layer through the single-workgroup streamed f32 kernels (SBR_CLUSTER=0)
    monkeypatch.setenv("SBR_CLUSTER", "0")
    check(PU.compare_step("LSTM", [256], "CCE", N=61, B=37, T=9), tol_h=2e-4)


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_triage_kernels_agree(cell):                       # SBR_FLAG_SIMPLE_REC | SBR_FLAG_SIMPLE_GEMM
    check(PU.compare_step(cell, [12], "CCE", N=23, B=5, T=7, flags=3))


@pytest.mark.parametrize("loss", ["Blackout", "BPR", "TOP1"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_sampled_heads(cell, loss):
    check(PU.compare_step(cell, [16], loss, N=40, B=6, T=5, S=7))


@pytest.mark.parametrize("loss", ["hinge", "logit", "logsig"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_margin_heads(cell, loss):
    # RNNMargin (rnn_margin.py): linear output layer, dense multi-target losses; S = --n_targets; rows with one to three
    # positives, one of them repeated, one that also occurs in the row's input
    check(PU.compare_step(cell, [20], loss, N=60, B=9, T=8, S=3, balance=1.5))
    check(PU.compare_step(cell, [20], loss, N=60, B=9, T=8, S=2, unique=False, balance=0.5))     # --repeated_interactions


def test_margin_head_with_popularity_based_targets_and_other_shapes():
    from oracle import rnn_oracle as O
    rng = np.random.default_rng(5)
    dflt = O.margin_default_target(rng.integers(1, 80, size=300), 100, 0.1)                        # --pb --min_access 0.1
    check(PU.compare_step("GRU", [50], "hinge", N=300, B=37, T=12, S=4, default_target=dflt))
    check(PU.compare_step("Vanilla", [128], "logsig", N=300, B=21, T=9, S=1, scale=0.05))
    check(PU.compare_step("LSTM", [50, 20], "logit", N=120, B=16, T=10, S=2, updater="adagrad"))
    # the benchmark's catalogue and width, the overlapped tail's shape (T >= 64)
    check(PU.compare_step("GRU", [128], "hinge", N=3706, B=64, T=70, S=3, scale=0.1, zipf=True, gap=1e-4), tol_g=2e-4)


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_two_layer_stack(cell):                            # recurrent_layers.py:57-68: dense layers above layer 0
    check(PU.compare_step(cell, [20, 12], "CCE", N=30, B=6, T=6))


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_embedding_layer_option(cell):
    # --r_emb E (recurrent_layers.py:46-50): EmbeddingLayer + flatten in front of dense layers; E = 6 pads to 8 columns,
    # F = 2 concatenates the two embeddings of a step; the table's gradient is a scatter-add with duplicates
    check(PU.compare_step(cell, [20], "CCE", N=61, B=37, T=9, emb=6))
    check(PU.compare_step(cell, [20, 12], "CCE", N=41, B=19, T=7, F=2, n_opt=10, emb=5))


@pytest.mark.parametrize("cell", ["GRU", "LSTM", "Vanilla"])
def test_bidirectional_option(cell):
    # --r_bi (recurrent_layers.py:70-76): a forward and a backwards layer per level, concatenated.  The backwards one
    # runs the ordinary kernels on per-row time-reversed copies of its input; ragged lengths make the reversal non-trivial
    check(PU.compare_step(cell, [20], "CCE", N=61, B=37, T=9, bi=True))
    check(PU.compare_step(cell, [20, 12], "CCE", N=41, B=19, T=7, bi=True, seed=2))


def test_bidirectional_with_embedding_sampled_head_and_two_indices():
    check(PU.compare_step("GRU", [16, 16], "BPR", N=41, B=19, T=7, S=8, bi=True, emb=6, F=2, n_opt=10, seed=3))
    check(PU.compare_step("LSTM", [128], "CCE", N=61, B=21, T=8, bi=True, seed=4))


def test_embedding_layer_with_sampled_head_and_wide_layer():
    check(PU.compare_step("GRU", [128], "BPR", N=61, B=37, T=9, S=8, emb=16))


def test_rating_feature_two_indices_per_step():            # --rf: F=2, input_size = N + 10
    check(PU.compare_step("LSTM", [8], "CCE", N=19, B=4, T=5, F=2, n_opt=10))


@pytest.mark.parametrize("updater", ["adagrad", "adadelta", "rmsprop", "nesterov", "adam"])
def test_updaters(updater):
    check(PU.compare_step("GRU", [8], "CCE", N=19, B=4, T=5, updater=updater, steps=3), steps=3)


@pytest.mark.parametrize("reg", [0.05, -0.05])
def test_bias_regularisation(reg):                         # rnn_one_hot.py:73-77
    check(PU.compare_step("GRU", [8], "CCE", N=19, B=4, T=5, reg=reg))


@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_gradient_clip_active(cell):                       # tiny popularity -> gate gradients beyond +-100
    check(PU.compare_step(cell, [8], "CCE", N=19, B=4, T=5, popscale=1e-4))


def test_atomic_scatter_fallback_agrees():               # SBR_FLAG_ATOMIC_SCATTER vs the sorted segment reduce
    check(PU.compare_step("GRU", [16], "CCE", N=33, B=17, T=11, seed=5, flags=4))


def test_scatter_with_heavy_duplicates():
    # few items, long rows: segments of the sorted scatter span many 32-entry chunks (atomic seams)
    check(PU.compare_step("GRU", [16], "CCE", N=12, B=40, T=30, seed=9))
    check(PU.compare_step("LSTM", [20], "BPR", N=12, B=24, T=20, S=4, seed=9))


@pytest.mark.parametrize("chunks", ["1", "2", "4"])
@pytest.mark.parametrize("cell", ["GRU", "LSTM"])
def test_long_sequences_chunked_bptt(cell, chunks, monkeypatch):
    # T >= 64 switches the bf16x6 BPTT to time chunks (state carried between launches, weight-gradient GEMM of
    # finished chunks on the side stream); ragged lengths put chunk borders inside and outside the valid range
    monkeypatch.setenv("SBR_BWD_CHUNKS", chunks)
    check(PU.compare_step(cell, [20], "CCE", N=41, B=9, T=70, seed=3), tol_h=2e-4)


@pytest.mark.parametrize("cell,layers,B,T", [("LSTM", [256], 1, 1), ("GRU", [256], 3, 2), ("LSTM", [512], 17, 1), ("Vanilla", [256], 16, 3),
                                             ("LSTM", [128], 1, 1), ("LSTM", [128], 5, 2), ("GRU", [256, 256], 2, 1), ("LSTM", [300], 33, 5),
                                             ("LSTM", [20], 1, 1)])
def test_smallest_shapes_on_the_cluster_and_pipelined_kernels(cell, layers, B, T):
    # one row, one step, a tile with one live row, a stack whose lower layer gets dh_ext at its only step
    r = PU.compare_step(cell, layers, "CCE", N=30, B=B, T=T, scale=0.05, k=1)
    assert r["h_last"] <= 2e-4 and r["cost"] <= 1e-5 and r["grad_worst"] <= 2e-4 and PU.params_ok(r, 2, bar=1e-3, tol_g=2e-4), r
    assert r["topk_mismatch"] == 0


def test_ragged_and_edge_lengths():
    # rows of length 1 and T, a row whose items are all id 0 (== the pad id), B not a multiple of 16
    check(PU.compare_step("GRU", [16], "CCE", N=33, B=17, T=11, seed=5))
    check(PU.compare_step("LSTM", [16], "CCE", N=33, B=1, T=3, seed=6))


def test_bench_shape_one_step():                           # BASELINE configs[1] with a shorter T for the oracle
    check(PU.compare_step("GRU", [128], "CCE", N=3706, B=256, T=20, steps=1), steps=1)


def test_full_size_properties():
    """Config 2 at full size (GRU-128, N=3706, B=256, T=200): properties that need no oracle."""
    from sbr_amd.engine import RNNEngine
    rng = np.random.default_rng(0)
    N, B, T, H = 3706, 256, 200, 128
    eng = RNNEngine(cell="GRU", layers=[H], n_items=N, max_length=T, batch_size=B, loss="CCE", updater="adam")
    try:
        params, cfg, batch = PU.build_case("GRU", [H], "CCE", N, B, T, seed=1, scale=0.0)
        batch["pop"][:] = 1.0
        eng.set_all_param_values(params)
        eng.set_batch(batch["X"], batch["mask"], batch["target"], None, batch["pop"])
        c1 = eng.forward_backward()
        g1 = eng.get_all_grad_values()
        # (1) softmax gradient rows sum to zero -> d cost / d b_out sums to ~0
        assert abs(g1[-1].sum()) <= 1e-5
        # (2) W_in rows of items that never occur in a valid position get exactly zero gradient
        seen = np.zeros(N, dtype=bool)
        m = batch["mask"].astype(bool)
        seen[np.unique(batch["X"][:, :, 0][m])] = True
        for gi in (0, 3, 6):
            assert np.all(g1[gi][~seen] == 0.0) and np.abs(g1[gi][seen]).max() > 0
        # (3) padding content is ignored: garbage ids behind the mask change nothing
        X2 = batch["X"].copy()
        X2[:, :, 0][~m] = rng.integers(0, N, size=(~m).sum())
        eng.set_batch(X2, batch["mask"], batch["target"], None, batch["pop"])
        c2 = eng.forward_backward()
        g2 = eng.get_all_grad_values()
        assert abs(c1 - c2) <= 1e-6 * abs(c1)
        assert all(np.allclose(a, b, rtol=1e-4, atol=1e-7) for a, b in zip(g1, g2))
        # (4) row permutation invariance of the cost (mean over rows)
        perm = rng.permutation(B)
        eng.set_batch(batch["X"][perm], batch["mask"][perm], batch["target"][perm], None, batch["pop"][perm])
        c3 = eng.forward_backward()
        assert abs(c1 - c3) <= 1e-5 * abs(c1)
        # (5) untrained model: cost ~ log(N) (uniform softmax), and a few Adam steps reduce it
        costs = [eng.train_step(sync=True) for _ in range(5)]
        assert abs(costs[0] - c3) <= 1e-5 * abs(c3) and costs[-1] < costs[0]
        assert abs(c1 - np.log(N)) < 0.5
        # (6) top-k ids are valid, distinct and never a seen item
        ids = eng.test_function((batch["X"], batch["mask"]), k=10)
        for b in range(B):
            assert len(set(ids[b])) == 10 and ids[b].min() >= 0 and ids[b].max() < N
            assert not set(ids[b]) & set(batch["X"][b, :int(batch["mask"][b].sum()), 0])
    finally:
        eng.close()


def test_error_behaviour():
    from sbr_amd.engine import RNNEngine
    eng = RNNEngine(cell="GRU", layers=[8], n_items=10, max_length=4, batch_size=2)
    try:
        with pytest.raises(ValueError, match="mismatch"):
            eng.set_all_param_values([np.zeros(3)])
        X = np.zeros((2, 4, 1), dtype=np.int32); X[0, 0, 0] = 10            # id out of range
        with pytest.raises(ValueError, match="out of range"):
            eng.set_batch(X, np.ones((2, 4), np.float32), np.zeros(2, np.int32), None, np.ones(2, np.float32))
        with pytest.raises(ValueError, match="prefix"):
            eng.set_batch(np.zeros((2, 4, 1), np.int32), np.array([[0, 1, 1, 1], [1, 1, 1, 1]], np.float32))
    finally:
        eng.close()


@pytest.mark.parametrize("loss", ["CCE", "Blackout"])
def test_virtual_ranks_sum_to_the_unsharded_step(loss):
    """Data-parallel by construction: two engines holding rows [0,B/2) and [B/2,B) of the same global
    batch produce gradient sections whose SUM equals the single-engine gradients ("N
    virtual ranks on one GPU"); cost shares add up; identical updates follow."""
    import torch
    N, B, T, S = 50, 32, 9, 6
    params, cfg, batch = PU.build_case("GRU", [16], loss, N, B, T, S=S, seed=4)
    smp = batch["samples"] if loss != "CCE" else None
    full = PU.engine_for(cfg, N, B, T, S=S)
    halves = [PU.engine_for(cfg, N, B, T, S=S, local_batch=B // 2, row_offset=r * (B // 2)) for r in range(2)]
    try:
        for e in [full] + halves:
            e.set_all_param_values(params)
        full.set_batch(batch["X"], batch["mask"], batch["target"], smp, batch["pop"])
        c_full = full.forward_backward()
        g_full = full.section("grads")[0].clone()
        total = torch.zeros_like(g_full)
        for r, e in enumerate(halves):
            sl = slice(r * (B // 2), (r + 1) * (B // 2))
            tgt = batch["target"] if loss != "CCE" else batch["target"][sl]
            e.set_batch(batch["X"][sl], batch["mask"][sl], tgt, smp, batch["pop"][sl])
            e.forward_backward()
            total += e.section("grads")[0]
        assert abs(float(total[-1]) - c_full) <= 1e-5 * abs(c_full)
        err = float((total - g_full).abs().max() / g_full.abs().max())
        assert err <= 1e-5, err
    finally:
        for e in [full] + halves:
            e.close()
