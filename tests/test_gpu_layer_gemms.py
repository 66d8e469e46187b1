"""The dense GEMMs between stacked recurrent layers (recurrent_layers.py:94-104: xt_l = h_{l-1} . W_in; backward pair dW_in = h^T . dxt,
dh_{l-1} = dxt . W_in^T) and the logits GEMM of the full-softmax head.

Default (round 4): the two-plane fp16 split, three MFMAs per product -- f32-class, held to the same 1e-5 bars as the bf16x6 form it
replaces (SBR_GEMM_F16=0), including a batch whose gate gradients reach the clip and one whose gradients are ~1e-9.
SBR_FLAG_BF16_LAYERS: plain bf16 operands, one MFMA per product (BASELINE configs[4]) -- bf16-input class: asserted against the
float64 oracle at 1e-2 of the largest entry, and asserted to differ from the f32-class result (the flag really changes the kernels)."""
import numpy as np
import pytest

import parity_util as PU
from oracle import rnn_oracle as O

pytestmark = pytest.mark.gpu
BF16_LAYERS = 256


@pytest.mark.parametrize("popscale", [1.0, 1e-4, 1e4], ids=["plain", "clip_active", "tiny_gradients"])
def test_fp16_split_layer_gemms_keep_the_f32_class_bars(popscale):
    r = PU.compare_step("LSTM", [256, 256], "CCE", N=3000, B=128, T=24, zipf=True, steps=1, scale=0.03, seed=51, popscale=popscale)
    assert r["h_last"] <= 1e-5 and r["cost"] <= 1e-5, r
    assert r["grad_worst"] <= (1e-5 if popscale == 1.0 else 2e-4), {k: v for k, v in r.items() if k.startswith("grad") and v > 1e-5}
    r = PU.compare_step("GRU", [128, 128], "BPR", N=2000, B=96, T=16, S=8, zipf=True, steps=1, scale=0.05, seed=52, popscale=popscale)
    assert r["h_last"] <= 1e-5 and r["grad_worst"] <= (1e-5 if popscale == 1.0 else 2e-4), r


def test_the_fp16_split_is_what_runs_and_the_switch_restores_bf16x6(monkeypatch):
    params, cfg, batch = PU.build_case("LSTM", [256, 256], "CCE", 3000, 128, 16, seed=53, scale=0.03, zipf=True)

    def grads():
        eng = PU.engine_for(cfg, 3000, 128, 16)
        try:
            eng.set_all_param_values(params)
            eng.set_batch(batch["X"], batch["mask"], batch["target"], None, batch["pop"])
            eng.forward_backward()
            return eng.get_all_grad_values()
        finally:
            eng.close()
    g16 = grads()
    monkeypatch.setenv("SBR_GEMM_F16", "0")
    import subprocess, sys, os, json     # the switch is read once per process: the bf16x6 run happens in a child
    code = ("import sys, json, numpy as np; sys.path[:0] = %r; import parity_util as PU\n"
            "params, cfg, batch = PU.build_case('LSTM', [256, 256], 'CCE', 3000, 128, 16, seed=53, scale=0.03, zipf=True)\n"
            "eng = PU.engine_for(cfg, 3000, 128, 16); eng.set_all_param_values(params)\n"
            "eng.set_batch(batch['X'], batch['mask'], batch['target'], None, batch['pop']); eng.forward_backward()\n"
            "np.save(sys.argv[1], np.concatenate([g.ravel() for g in eng.get_all_grad_values()]))\n") % ([os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__))],)
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "sbr_x6_grads.npy")
    subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, SBR_GEMM_F16="0"))
    gx6 = np.load(out)
    g16f = np.concatenate([g.ravel() for g in g16])
    d = np.abs(g16f - gx6).max() / np.abs(gx6).max()
    assert 0.0 < d <= 2e-6, d            # different kernels (not bit-identical), same f32 class


def test_bf16_layer_gemms_flag():
    cell, layers, loss, N, B, T, S = "LSTM", [512, 512], "Blackout", 20000, 64, 12, 32
    params, cfg, batch = PU.build_case(cell, layers, loss, N, B, T, S=S, seed=54, scale=0.02, zipf=True)
    ocost, ograds, aux = O.cost_and_grads(params, cfg, PU.oracle_batch(batch))
    res = {}
    for flags in (0, BF16_LAYERS):
        eng = PU.engine_for(cfg, N, B, T, S=S, flags=flags)
        try:
            eng.set_all_param_values(params)
            eng.set_batch(batch["X"], batch["mask"], batch["target"], batch["samples"], batch["pop"])
            cost = eng.forward_backward()
            Hp = eng.debug_buffer("h_last").size // (((B + 15) // 16) * 16)
            h = eng.debug_buffer("h_last").reshape(-1, Hp)[:B, :layers[-1]].copy()
            res[flags] = (cost, h, eng.get_all_grad_values())
        finally:
            eng.close()
    c0, h0, g0 = res[0]
    c1, h1, g1 = res[BF16_LAYERS]
    assert PU.rel_err(h0, aux["h"]) <= 1e-5 and max(PU.rel_err(a, b) for a, b in zip(g0, ograds)) <= 1e-5
    eh, eg = PU.rel_err(h1, aux["h"]), max(PU.rel_err(a, b) for a, b in zip(g1, ograds))
    assert abs(c1 - ocost) <= 1e-3 * abs(ocost) and eh <= 1e-2 and eg <= 2e-2, (c1, ocost, eh, eg)
    assert eh > 1e-5                      # bf16-input class: the flag really changes the kernels
