"""Lint of the generated gfx950 code of rec_bwd_x6p<.., WT> (csrc/sbr_rec_p.hip), runs on CPU (hipcc cross-compiles).

That kernel loads the saved activations of the next time step with hand-written `global_load_dword` instructions the
compiler's vmcnt bookkeeping cannot see, and makes them valid with a hand-written `s_waitcnt vmcnt(n)` at the top of the next
iteration (so that the write-through stores issued behind the loads stay in flight).  Between the two the compiler believes
the destination registers hold their final values: any instruction it places there that READS one of them (a phi copy at the
loop header, a spill, a hoisted use) would read a register whose load is still in flight.  The source is written so that
this does not happen; this test checks the code the compiler actually produced."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sequence-based-recommendations_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _reads(line, regs):
    """registers of `regs` (ints) the instruction on `line` mentions: v12, v[10:13]."""
    hit = set()
    for m in re.finditer(r"\bv(\d+)\b", line):
        if int(m.group(1)) in regs:
            hit.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        for r in range(int(m.group(1)), int(m.group(2)) + 1):
            if r in regs:
                hit.add(r)
    return hit


def lint_kernel(lines):
    """-> list of (line number, text, registers) violations.  Checked per depth-1 loop (the two role loops of the kernel):
    from a hand-written load to the end of the loop in text order, and from the loop's header to its hand-written wait
    (the back edge).  Behind the loop the source waits with vmcnt(0) before anything else (publish_progress)."""
    bad = []
    headers = [(i, re.search(r"^\.(LBB\d+_\d+):", ln).group(1)) for i, ln in enumerate(lines)
               if "Loop Header: Depth=1" in ln and re.search(r"^\.(LBB\d+_\d+):", ln)]
    for h, name in headers:
        members = [i for i, ln in enumerate(lines) if re.search(r"Header=%s\b" % name[1:], ln)]
        last = max(members) if members else h
        end = next((i for i in range(last + 1, len(lines)) if re.match(r"^\.LBB\d+_\d+:", lines[i])), len(lines))
        in_asm, inflight, loaded = False, set(), set()
        for i in range(h, end):
            s = lines[i].strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not s or s.startswith(";") or s.startswith("."):
                continue
            if in_asm:
                m = re.match(r"global_load_dword v(\d+),", s)
                if m:
                    inflight.add(int(m.group(1))); loaded.add(int(m.group(1)))
                    continue
                if s.startswith("s_waitcnt vmcnt"):
                    inflight.clear()
                    continue
            hit = _reads(s, inflight)
            if hit:
                bad.append((i, s, sorted(hit)))
        in_asm = False
        for i in range(h, end):                  # the back edge: header -> first hand-written wait
            s = lines[i].strip()
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if in_asm and s.startswith("s_waitcnt vmcnt"):
                break
            if not s or s.startswith(";") or s.startswith("."):
                continue
            hit = _reads(s, loaded)
            if hit:
                bad.append((i, s, sorted(hit)))
    # the prologue (straight-line code in front of the first loop): first hand-written load -> the wait that follows it
    first = headers[0][0] if headers else len(lines)
    in_asm, inflight = False, set()
    for i in range(0, first):
        s = lines[i].strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if in_asm:
            m = re.match(r"global_load_dword v(\d+),", s)
            if m:
                inflight.add(int(m.group(1)))
                continue
            if s.startswith("s_waitcnt vmcnt"):
                inflight.clear()
                continue
        hit = _reads(s, inflight)
        if hit:
            bad.append((i, s, sorted(hit)))
    return bad


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_no_instruction_reads_a_register_whose_hand_written_load_is_in_flight():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "rec_p.s")
        subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-munsafe-fp-atomics", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out,
                               os.path.join(CSRC, "sbr_rec_p.hip")], stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    starts = [i for i, ln in enumerate(text) if re.match(r"^_Z11rec_bwd_x6pILi\dELb0ELb0ELb1ELb1EEv7RecArgs:", ln)]
    assert len(starts) == 2, "expected the GRU and the Vanilla instance of rec_bwd_x6p<.., WT>"
    for st in starts:
        end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
        body = text[st:end + 1]
        loads = [ln for ln in body if re.match(r"\s*global_load_dword v\d+, v\d+, s\[", ln)]
        assert len(loads) >= 2, "the hand-written loads are gone: update this lint"
        bad = lint_kernel(body)
        assert not bad, "\n".join("%s: line %d: %s reads in-flight %s" % (text[st].split(":")[0], i, s, r) for i, s, r in bad[:10])
        # both role loops wait with a non-zero count: the write-through stores stay in flight
        waits = [ln.strip() for ln in body if ln.strip().startswith("s_waitcnt vmcnt(") and "vmcnt(0)" not in ln]
        assert len(waits) >= 2, waits
