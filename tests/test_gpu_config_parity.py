"""Parity at the BASELINE.json configs' OWN shapes: the HIP engine, through the C-ABI, against the float64 oracle.

C1 / C2 run exactly as benched (B=256, T=200, N=3706; rows of full length as in bench.py's default, and ragged);
C3 / C4 / C5 keep their catalogue, width, head and batch and shorten T so that the dense float64 oracle (which
materialises the N x G*H gradient the reference's AdvancedIncSubtensor builds) finishes in seconds.  Every case
checks: cost, final hidden state, every parameter gradient, parameters after Adam / Adagrad steps, predict scores,
and the ordered top-10 ids on the rows whose oracle logits are further apart than the stated gap (a tie-free
fixture by assertion: at least 80 % of the rows must qualify).

Bars (north_star): logits / scores 1e-3 relative, gradients 1e-4 relative to the largest entry of the array, top-k
ids exact.  The tolerances asserted here are 10x tighter than those bars (measured on MI355X: hidden state and
gradients ~1e-6, scores ~1e-6, profiles/round2_config_parity.jsonl); parameters after optimizer steps keep the 1e-3
bar of the other parity tests (Adam turns a 1e-6 gradient difference on a near-zero gradient into a full-size step)."""
import numpy as np
import pytest

import parity_util as PU

pytestmark = pytest.mark.gpu

GAP = 2e-5        # logit units (the logits of these models span ~0.5); float32 scores agree with the oracle to ~1e-6


def check(r, steps, tol_g=1e-5, tol_h=1e-5, rows=None):
    grads = {k: v for k, v in r.items() if k.startswith("grad")}
    assert r["param_roundtrip"] == 0.0
    assert r["h_last"] <= tol_h, r
    assert r["cost"] <= 1e-5, r
    assert r["grad_worst"] <= tol_g, grads
    PU.params_ok(r, steps, bar=1e-3, tol_g=tol_g)
    assert r["predict_scores"] <= 1e-4, r
    assert r["topk_mismatch"] == 0, r
    if rows is not None:
        assert r["topk_rows_compared"] >= 0.8 * rows, r


@pytest.mark.parametrize("full", [True, False], ids=["full_length", "ragged"])
def test_c2_as_benched(full):
    # BASELINE configs[1]: GRU-128, N=3706, B=256, T=200, CCE, Adam -- the shape bench.py times, Zipf item ids
    r = PU.compare_step("GRU", [128], "CCE", N=3706, B=256, T=200, full=full, zipf=True, steps=2, k=10, gap=GAP,
                        scale=0.05, seed=21)
    check(r, 2, rows=256)


@pytest.mark.parametrize("full", [True, False], ids=["full_length", "ragged"])
def test_c1_reference_cpu_config(full):
    # BASELINE configs[0]: rnn_one_hot 1-layer LSTM-20 on the ML-1M shape, B=256, T=200
    r = PU.compare_step("LSTM", [20], "CCE", N=3706, B=256, T=200, full=full, zipf=True, steps=2, k=10, gap=GAP,
                        scale=0.05, seed=22)
    check(r, 2, rows=256)


def _plant_duplicate_cells(batch):
    # sampled negatives are drawn with replacement and are not filtered against the targets (rnn_sampling.py:188-191,
    # sparse_lstm.py:49): repeat a negative, make a negative equal a target, repeat a target
    s, t = batch["samples"], batch["target"]
    s[1] = s[0]
    s[2] = t[0]
    t[5] = t[4]


@pytest.mark.parametrize("loss", ["Blackout", "BPR", "TOP1"])
def test_c3_sampled_heads_100k_items(loss):
    # BASELINE configs[2]: rnn_sampling, LSTM-256, N=100 000, S=32 shared negatives, B=256; T=16 for the oracle
    r = PU.compare_step("LSTM", [256], loss, N=100000, B=256, T=16, S=32, zipf=True, steps=1, k=10, gap=GAP,
                        scale=0.03, seed=23, updater="adagrad", tweak=_plant_duplicate_cells)
    check(r, 1, rows=256)


def test_c4_ml20m_catalogue_full_softmax():
    # BASELINE configs[3] shape: LSTM-256, N=26 744, CCE; the per-GPU share of the 8-GPU run is 32 rows, the single-GPU
    # bench line of this config holds 256: both
    r = PU.compare_step("LSTM", [256], "CCE", N=26744, B=256, T=16, zipf=True, steps=1, k=10, gap=GAP, scale=0.03, seed=24)
    check(r, 1, rows=256)
    r = PU.compare_step("LSTM", [256], "CCE", N=26744, B=32, T=24, zipf=True, steps=2, k=10, gap=GAP, scale=0.03, seed=25)
    check(r, 2, rows=32)


def test_c5_shape_two_layers_512_sampled():
    # BASELINE configs[4] shape: 2 x LSTM-512 (recurrent_layers stack), sampled softmax, N as large as the float64 oracle
    # fits comfortably (100 000 items: W_in alone is 205 M parameters)
    r = PU.compare_step("LSTM", [512, 512], "Blackout", N=100000, B=64, T=12, S=32, zipf=True, steps=1, k=10, gap=GAP,
                        scale=0.02, seed=26, updater="adagrad", tweak=_plant_duplicate_cells)
    check(r, 1, rows=64)
