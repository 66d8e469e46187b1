#!/usr/bin/env python3
"""Instruction mix of the step loops of the 128-unit chains, from the gfx950 code hipcc generates (no GPU needed):
    python tools/isa_stats.py [cell]        cell: 0 = LSTM, 1 = GRU (default: C2), 2 = Vanilla
For each of the two role loops (waves 0-3 / 4-7) of rec_fwd_x6p<cell, fused gather, fp16 planes> and rec_bwd_x6p<cell, write-through>:
instructions per iteration by class.  One iteration = one time step of one wave; a SIMD holds two such waves (one per role)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sequence-based-recommendations_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_isa_lint import loops_of          # noqa: E402  (the same loop finder the lint uses)

CLASSES = [("mfma", r"^v_s?mfmac?"), ("valu", r"^v_(?!mfma|smfmac)"), ("salu", r"^s_(?!waitcnt|cbranch|branch|nop|sleep|barrier|endpgm)"),
           ("lds read", r"^ds_read"), ("lds write / other", r"^ds_(?!read)"), ("lds-dma load", r"^global_load_lds"),
           ("global load", r"^global_load_(?!lds)"), ("global store", r"^global_store|^global_atomic"),
           ("waitcnt", r"^s_waitcnt"), ("branch", r"^s_cbranch|^s_branch"), ("nop / sleep", r"^s_nop|^s_sleep")]


def main():
    cell = sys.argv[1] if len(sys.argv) > 1 else "1"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rec_p.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-slp-vectorize",
                               "-S", "--cuda-device-only", "-o", out, os.path.join(CSRC, "sbr_rec_p.hip")], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    for title, pat in (("rec_fwd_x6p (fused gather, fp16 planes)", r"^_Z11rec_fwd_x6pILi%sELb1ELb0ELb1EEv7RecArgs:" % cell),
                       ("rec_bwd_x6p (write-through, LDS ring)", r"^_Z11rec_bwd_x6pILi%sELb0ELb0ELb1ELi1EEv7RecArgs:" % cell)):
        st = next(i for i, ln in enumerate(text) if re.match(pat, ln))
        end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
        body = text[st:end + 1]
        meta = {}
        for k in ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "group_segment_fixed_size"):
            m = [re.search(r"\.%s:\s+(\d+)" % k, ln) for ln in text]
        print("== %s, cell %s: %d lines of code" % (title, cell, len([l for l in body if l.startswith("\t") and not l.strip().startswith((";", "."))])))
        for n, (lo, hi) in enumerate(loops_of(body)):
            code = [ln.strip() for ln in body[lo:hi] if ln.startswith("\t") and not ln.strip().startswith((";", "."))]
            counts, rest = {}, 0
            for c in code:
                for name, rx in CLASSES:
                    if re.match(rx, c):
                        counts[name] = counts.get(name, 0) + 1
                        break
                else:
                    rest += 1
            # fast path: without the blocks of the bounded-spin loops (depth-2 loops: taken only while a producer is late) and the
            # blocks that raise the fault flag -- what a step issues when nobody waits
            fast, skip = 0, False
            for ln in body[lo:hi]:
                if re.match(r"^\.LBB\d+_\d+:|^; %bb\.", ln):
                    skip = "Depth=2" in ln
                st = ln.strip()
                if not ln.startswith("\t") or st.startswith((";", ".")):
                    continue
                if "global_atomic_or" in st:
                    skip = True
                if not skip and not re.match(r"^v_s?mfmac?", st):
                    fast += 1
            total = len(code)
            non = total - counts.get("mfma", 0)
            print("  role loop %d: %d instructions in the loop's range, %d of them not MFMA; %d not MFMA outside the bounded-spin loops and fault blocks"
                  % (n, total, non, fast))
            print("    " + ", ".join("%s %d" % (k, counts[k]) for k, _ in CLASSES if k in counts) + (", other %d" % rest if rest else ""))
            mf = counts.get("mfma", 0)
            print("    issue estimate: %d MFMAs x 16 cycles = %d cycles of matrix pipe; %d other instructions x >= 4 cycles = >= %d cycles of issue"
                  % (mf, 16 * mf, non, 4 * non))


if __name__ == "__main__":
    main()
