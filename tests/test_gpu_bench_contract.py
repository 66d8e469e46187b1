"""bench.py prints ONE JSON line with the fields the driver reads (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload) plus `roofline` for the dominant
kernel, measured with HIP events inside the timed region, and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2"] + list(extra),
                         cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def check_common(d, steps=6, warmup=2):
    assert d["metric"].startswith("user-sequences/sec") and d["unit"] == "user-sequences/s"
    assert (d["n_gpus"], d["steps"], d["warmup"]) == (1, steps, warmup)
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] == pytest.approx(d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3), rel=2e-3)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-2) and 0 < r["frac"] < 1
    assert r["launch_us"] > 0 and r["launch_us"] < d["ms_per_step"] * 1e3
    assert d["phases_us"][r["kernel"]] == r["launch_us"]


def test_default_config_line_without_cpu_leg():
    d = run("--no-cpu-baseline")
    check_common(d)
    assert "cpu_baseline" not in d and d["config"]["workload"].startswith("c2:")
    assert d["roofline"]["kernel"] == "rec_bwd" and isinstance(d["roofline"]["traffic"], int)


def test_cpu_baseline_leg_is_bounded_and_reported():
    d = run("--config", "c1", "--cpu-steps", "1", "--cpu-seconds", "5")
    check_common(d)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "user-sequences/s" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    assert d["value"] > 100 * c["value"]
