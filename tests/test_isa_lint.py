"""Lint of the generated gfx950 code of rec_bwd_x6p<.., WT> (csrc/sbr_rec_p.hip), runs on CPU (hipcc cross-compiles).

That kernel brings the saved activations of later time steps into an LDS ring with hand-written LDS-DMA loads
(`global_load_lds_dword`) that the compiler's vmcnt bookkeeping cannot see, and makes them valid with a hand-written
`s_waitcnt vmcnt(n)` in front of the LDS reads of the step that uses them; n > 0 keeps the write-through stores issued behind
the loads in flight.  What the source relies on, checked here on the code the compiler actually produced:

  * both role loops (waves 0-3 / 4-7) of all instances (LSTM, GRU, Vanilla) contain the LDS-DMA loads and a hand-written wait with
    the count the pipeline depth implies, and NO wait for vmcnt(0) -- compiler-inserted or otherwise -- inside the loop (it
    would drain the stores: the thing the scheme exists to avoid);
  * inside a loop, the LDS reads of the ring come after that wait (the asm's memory clobber pins them; a read hoisted above
    it would see a stage whose load may still be in flight);
  * the compiler itself never touches M0 in these kernels (the DMA's LDS address travels there)."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sequence-based-recommendations_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# (template argument CELL: 0 = LSTM, 1 = GRU, 2 = Vanilla)
# backward, PD = 4 stages; per step 2 load pieces + 4 stores (GRU, LSTM) / 1 + 1 (Vanilla): NST + (PD - 1) * (NLI + NST)
EXPECTED_WAIT = {"0": 22, "1": 22, "2": 7}
# forward with the fused gather (rec_fwd_x6p<CELL, FUSE, .., F16>), XPD = 4: (XPD - 1) * (stores of a step + 1 load piece);
# stores of a step: h + ONE 16-byte element of the four saved gate values (GRU), + c (LSTM), h alone (Vanilla)
EXPECTED_WAIT_FWD = {"0": 12, "1": 9, "2": 6}


_ASM = {}


def rec_p_asm():
    """gfx950 assembly of csrc/sbr_rec_p.hip (one compile for all tests of this file)"""
    if "text" not in _ASM:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "rec_p.s")
            subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-munsafe-fp-atomics", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out,
                                   os.path.join(CSRC, "sbr_rec_p.hip")], stderr=subprocess.DEVNULL)
            _ASM["text"] = open(out).read().splitlines()
    return _ASM["text"]


def loops_of(lines):
    """[(first, last) line index] of the depth-1 loops that carry matrix instructions (the two role loops; the prologue's fill
    loops carry none): from the loop header's label to the last branch back into the loop (the blocks the compiler's loop comments
    name, extended over every later branch that targets a label inside the range found so far)."""
    labels = {}
    for i, ln in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = i
    branches = []
    for j, ln in enumerate(lines):
        m = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)\s*$", ln)
        if m and m.group(1) in labels:
            branches.append((j, labels[m.group(1)]))
    heads = [i for i, ln in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:.*Loop Header: Depth=1", ln)]
    out = []
    for n, i in enumerate(heads):
        name = re.match(r"^\.L(BB\d+_\d+):", lines[i]).group(1)
        members = [j for j, l2 in enumerate(lines) if re.search(r"Header=%s\b" % name, l2)]
        first = min(members + [i])                                # (a rotated loop keeps blocks in front of its header label)
        end = max(members + [i]) + 1
        limit = min([h for h in heads if h > i] + [len(lines)])
        grown = True
        while grown:
            grown = False
            for j, tgt in branches:
                if end <= j < limit and first <= tgt < end:
                    end, grown = j + 1, True
        if any("v_mfma" in l2 or "v_smfmac" in l2 for l2 in lines[first:end]):
            out.append((first, end))
    return out


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_lds_ring_prefetch_of_the_write_through_backward_kernel():
    text = rec_p_asm()
    starts = [(i, re.match(r"^_Z11rec_bwd_x6pILi(\d)ELb0ELb0ELb1ELi1EEv7RecArgs:", ln).group(1)) for i, ln in enumerate(text)
              if re.match(r"^_Z11rec_bwd_x6pILi\dELb0ELb0ELb1ELi1EEv7RecArgs:", ln)]
    assert len(starts) == 3, "expected the LSTM, the GRU and the Vanilla instance of rec_bwd_x6p<.., WT>"
    for st, cell in starts:
        end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
        body = text[st:end + 1]
        m0 = [ln.strip() for ln in body if re.search(r"\bm0\b", ln) and not ln.strip().startswith("s_mov_b32 m0,")
              and not ln.strip().startswith(";")]
        assert not m0, m0
        loops = loops_of(body)
        assert len(loops) == 2, "two role loops expected, found %d" % len(loops)
        want = "s_waitcnt vmcnt(%d)" % EXPECTED_WAIT[cell]
        for lo, hi in loops:
            code = [ln.strip() for ln in body[lo:hi]]
            assert any(c.startswith("global_load_lds_dwordx4") for c in code), "no LDS-DMA load in the role loop"
            assert want in code, (want, [c for c in code if c.startswith("s_waitcnt vmcnt")])
            assert "s_waitcnt vmcnt(0)" not in code, "a full wait inside the role loop drains the write-through stores"
            # ring reads: pinned directly behind the wait by its memory clobber -- the row-major value (ds_read_b32) and, for the gated
            # cells, the 16-byte element of the four saved gate values (ds_read_b128); a read hoisted above the wait would be missing here
            w = code.index(want)
            behind = [c for c in code[w + 1:w + 8] if c.startswith("ds_read")]
            assert any(c.startswith("ds_read_b32") for c in behind), "the ring's reads do not follow the hand-written wait"
            assert cell == "2" or any(c.startswith("ds_read_b128") for c in behind), "the saved gate values are not one 16-byte read behind the wait"


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_lds_ring_of_the_fused_gather_in_the_forward_kernel():
    """rec_fwd_x6p<CELL, FUSE = true, PROF = false, F16 = true>: the W_in rows of later steps arrive in an LDS ring by LDS-DMA; the
    same invariants as for the backward kernel above."""
    text = rec_p_asm()
    starts = [(i, re.match(r"^_Z11rec_fwd_x6pILi(\d)ELb1ELb0ELb1EEv7RecArgs:", ln).group(1)) for i, ln in enumerate(text)
              if re.match(r"^_Z11rec_fwd_x6pILi\dELb1ELb0ELb1EEv7RecArgs:", ln)]
    assert len(starts) == 3
    for st, cell in starts:
        end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
        body = text[st:end + 1]
        m0 = [ln.strip() for ln in body if re.search(r"\bm0\b", ln) and not ln.strip().startswith("s_mov_b32 m0,")
              and not ln.strip().startswith(";")]
        assert not m0, m0
        loops = loops_of(body)
        assert len(loops) == 2, "two role loops expected, found %d" % len(loops)
        want = "s_waitcnt vmcnt(%d)" % EXPECTED_WAIT_FWD[cell]
        for lo, hi in loops:
            code = [ln.strip() for ln in body[lo:hi]]
            assert any(c.startswith("global_load_lds_dword") for c in code)
            assert want in code, (want, [c for c in code if c.startswith("s_waitcnt vmcnt")])
            assert "s_waitcnt vmcnt(0)" not in code
            assert not any(c.startswith("global_load_dword") for c in code), "a register-destination load is back in the step loop"


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
def test_one_sparse_matrix_instruction_per_product_in_the_step_loops_of_the_c2_chains():
    """Sparse planes: a product of the 128-unit chains is ONE v_smfmac_f32_16x16x64_f16, so a GRU step issues
    3 gates x 4 k-blocks = 12 in each forward role loop and 12 k-blocks (K = 384) = 12 in each backward one; the packed dense form
    (round 3) had 24 / 24 v_mfma_f32_16x16x32_f16, three per product 36 / 36.  No dense MFMA is left in the loops.
    (tools/isa_stats.py prints the whole mix.)"""
    text = rec_p_asm()
    for pat, want in ((r"^_Z11rec_fwd_x6pILi1ELb1ELb0ELb1EEv7RecArgs:", 12), (r"^_Z11rec_bwd_x6pILi1ELb0ELb0ELb1ELi1EEv7RecArgs:", 12)):
        st = next(i for i, ln in enumerate(text) if re.match(pat, ln))
        end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
        body = text[st:end + 1]
        loops = loops_of(body)
        assert len(loops) == 2
        for lo, hi in loops:
            n = sum(1 for ln in body[lo:hi] if ln.strip().startswith("v_smfmac_f32_16x16x64_f16"))
            dense = sum(1 for ln in body[lo:hi] if ln.strip().startswith("v_mfma_"))
            assert n == want and dense == 0, (pat, n, want, dense)


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the 16-row cluster chains (csrc/sbr_rec_c16.hip)
# ---------------------------------------------------------------------------------------------------------------------------------
def rec_c16_asm():
    if "c16" not in _ASM:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "rec_c16.s")
            subprocess.check_call([HIPCC if os.path.exists(HIPCC) else "hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                                   "-munsafe-fp-atomics", "-fno-slp-vectorize", "-S", "--cuda-device-only", "-o", out,
                                   os.path.join(CSRC, "sbr_rec_c16.hip")], stderr=subprocess.DEVNULL)
            _ASM["c16"] = open(out).read().splitlines()
    return _ASM["c16"]


def _kernel(text, mangled):
    st = next(i for i, ln in enumerate(text) if ln.startswith(mangled + ":"))
    end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
    return text[st:end + 1]


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which("hipcc")), reason="hipcc not available")
@pytest.mark.parametrize("kernel,polls", [("_Z11rec_fwd_c16ILi0ELi256EEv7RecArgs", 1), ("_Z11rec_bwd_c16ILi0ELi256EEv7RecArgs", 1),
                                          ("_Z11rec_fwd_c16ILi0ELi512EEv7RecArgs", 1), ("_Z12rec_bwd_c16tILi0EEv7RecArgs", 2),
                                          ("_Z11rec_fwd_c16ILi1ELi256EEv7RecArgs", 1), ("_Z11rec_bwd_c16ILi1ELi256EEv7RecArgs", 1)])
def test_cluster_chain_step_loops_wait_only_behind_their_polls(kernel, polls):
    """What the rebuilt chains rest on (csrc/sbr_rec_c16.hip header), checked on the code hipcc produced for the
    same-XCC body of every chain (the first step loop of a kernel: its polls are the 16-byte `sc1` asm loads):

      * the only waits for vmcnt(0) in the step loop are the hand-written ones -- one behind each poll's loads, one in the counter-mode
        block -- i.e. the compiler placed none of its own (it did, in three ways, while the kernels were written: at the loop header
        for loop-carried prefetch registers, between the poll's four loads when the cross-XCC form shared their loop, and at the
        header for the prologue's pending loads);
      * the poll's loads are issued back to back: no wait of any kind between the first and the last of them;
      * no scratch traffic in the step loop (the 512-unit kernels run at the register limit; what spills, spills in the cross-XCC body)."""
    body = _kernel(rec_c16_asm(), kernel)
    loops = [(lo, hi) for lo, hi in loops_of(body) if any("dwordx4" in l and "sc1" in l for l in body[lo:hi])]
    assert loops, "no step loop with sc1 polls found"
    lo, hi = loops[0]
    code = body[lo:hi]
    asm_depth, hand, own = 0, 0, 0
    poll_runs, in_run = [], None
    for ln in code:
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            asm_depth += 1
        elif s.startswith(";;#ASMEND"):
            asm_depth -= 1
        elif re.match(r"s_waitcnt\s+vmcnt\(0\)\s*$", s) or re.match(r"s_waitcnt\s+vmcnt\(0\)\s", s):
            if asm_depth > 0:
                hand += 1
            else:
                own += 1
        if "global_load_dwordx4" in s and "sc1" in s:
            in_run = [] if in_run is None else in_run
        elif in_run is not None:
            if s.startswith("s_waitcnt"):
                in_run.append(s)
                if asm_depth > 0:                                 # the poll's own wait ends the run
                    poll_runs.append(in_run[:-1]); in_run = None
        assert "scratch_" not in s, s
    assert own == 0, "%d compiler-placed vmcnt(0) in the step loop of %s" % (own, kernel)
    assert hand == polls + 1, (hand, polls)                       # + the counter-mode block's
    assert len(poll_runs) == polls and all(r == [] for r in poll_runs), poll_runs
