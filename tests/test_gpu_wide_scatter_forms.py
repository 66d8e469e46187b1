"""The non-default forms of the step tail of a wide index-input layer (G*Hp >= 512 floats per row: LSTM-128 and wider) -- each a switch
that is read once per process, so each runs in a child: the atomic scatter-add of rounds 1 - 3 (SBR_SCAT_RANGE=0) -- what rows wider
than 1024 floats take -- and the segment-parallel form (2), and the plain dense pass over W_in (SBR_ROW_AWARE_UPDATE=0).
Same bars as the default form (the range scatter-add, every other wide-layer test): cost, hidden state and gradients against the float64
oracle, parameters after two Adam steps; and after three steps the same parameters as the default form up to what Adam makes of
summation-order roundings (an element whose gradient is ~0 moves by ~lr whatever the gradient's size: 5e-5 of the largest parameter)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, json, numpy as np
sys.path[:0] = [%r, %r]
import parity_util as PU
N, B, T = (int(x) for x in sys.argv[2:5])
r = PU.compare_step("LSTM", [256], "CCE", N=N, B=B, T=T, zipf=True, steps=2, scale=0.03, seed=61)
params, cfg, batch = PU.build_case("LSTM", [256], "CCE", N, B, T, seed=61, scale=0.03, zipf=True)
eng = PU.engine_for(cfg, N, B, T)
eng.set_all_param_values(params)
eng.set_batch(batch["X"], batch["mask"], batch["target"], None, batch["pop"])
for _ in range(3):
    eng.train_step(sync=True)
np.save(sys.argv[1], np.concatenate([p.ravel() for p in eng.get_all_param_values()]))
eng.close()
print(json.dumps({k: float(v) for k, v in r.items() if not k.startswith(("grad:", "pstep:"))}))
""" % (ROOT, os.path.join(ROOT, "tests"))


def child(tmp_path, name, env, N, B, T):
    out = str(tmp_path / (name + ".npy"))
    p = subprocess.run([sys.executable, "-c", CHILD, out, str(N), str(B), str(T)], env=dict(os.environ, **env), capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-1500:]
    import json
    return json.loads(p.stdout.strip().splitlines()[-1]), np.load(out)


def bars(r):
    assert r["h_last"] <= 1e-5 and r["cost"] <= 1e-5 and r["grad_worst"] <= 1e-5, r
    assert r["params_twin"] <= 2e-5 and r["topk_mismatch"] == 0, r


@pytest.mark.parametrize("env", [{"SBR_SCAT_RANGE": "0"}, {"SBR_SCAT_RANGE": "2"}, {"SBR_ROW_AWARE_UPDATE": "0"}],
                         ids=["atomic", "segment_parallel", "plain_dense_pass"])
def test_scatter_add_forms_of_wide_rows(tmp_path, env):
    r0, p0 = child(tmp_path, "default", {}, 3000, 64, 24)
    r1, p1 = child(tmp_path, "form", env, 3000, 64, 24)
    bars(r0); bars(r1)
    # the atomic-free forms sum in a fixed order; against each other and against the atomic kernel they differ by roundings only
    assert np.abs(p0 - p1).max() <= 5e-5 * np.abs(p0).max()
