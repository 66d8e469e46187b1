"""Runs the REFERENCE's preprocess.py (imported from /root/reference, Python 2 source that happens to run under Python 3
once `raw_input` exists) on a small seeded synthetic interaction file and stores the raw file + everything it wrote
under tests/golden/preprocess/ -- the known-answer vectors tests/test_preprocess_cpu.py holds sbr_amd.preprocess to.
Only runs where /root/reference exists (this container); the fixtures are committed.
    python tools/make_preprocess_golden.py
"""
import builtins
import importlib.util
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARGS = ["--columns", "uirt", "--sep", "::", "--min_item_pop", "3", "--min_user_activity", "2", "--seed", "3"]


def raw_lines():
    rng = np.random.default_rng(11)
    lines, t = [], 10 ** 9
    for u in range(80):
        for _ in range(int(rng.integers(1, 25))):
            t += int(rng.integers(1, 100))
            lines.append("%d::%d::%d::%d" % (2000 + u, 300 + int(min(59, rng.zipf(1.4) - 1)), int(rng.integers(1, 6)), t))
    rng.shuffle(lines)
    return lines


def main():
    work = tempfile.mkdtemp()
    raw = os.path.join(work, "ratings.dat")
    with open(raw, "w") as f:
        f.write("\n".join(raw_lines()) + "\n")
    builtins.raw_input = lambda prompt="": "y"
    sys.argv = ["preprocess.py", "-f", raw] + ARGS
    spec = importlib.util.spec_from_file_location("ref_preprocess", "/root/reference/preprocess.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    out = os.path.join(ROOT, "tests", "golden", "preprocess")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out)
    shutil.copy(raw, out)
    for name in sorted(os.listdir(os.path.join(work, "data"))):
        if name != "README":
            shutil.copy(os.path.join(work, "data", name), out)
    with open(os.path.join(out, "ARGS"), "w") as f:
        f.write(" ".join(ARGS) + "\n")
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
