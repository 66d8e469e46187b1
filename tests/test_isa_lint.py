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
# stores of a step: h + four saved gate values (GRU), + c (LSTM), h alone (Vanilla)
EXPECTED_WAIT_FWD = {"0": 21, "1": 18, "2": 6}


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
    """[(first, last) line index] of the depth-1 loops (the two role loops; the prologue's fill loops carry no MFMAs)."""
    out = []
    for i, ln in enumerate(lines):
        m = re.match(r"^\.(LBB\d+_\d+):.*Loop Header: Depth=1", ln)
        if not m:
            continue
        name = m.group(1)[1:]
        members = [j for j, l2 in enumerate(lines) if re.search(r"Header=%s\b" % name, l2)]
        last = max(members) if members else i
        end = next((j for j in range(last + 1, len(lines)) if re.match(r"^\.LBB\d+_\d+:", lines[j])), len(lines))
        if any("v_mfma" in l2 or "v_smfmac" in l2 for l2 in lines[i:end]):
            out.append((i, end))
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
            # ring reads: the first ds_read of the loop body is behind the wait (the ring is read at the top of an iteration,
            # the operand planes of the MFMA phase later)
            first_read = next(i for i, c in enumerate(code) if c.startswith("ds_read"))
            assert code.index(want) < first_read, "an LDS read sits in front of the hand-written wait"


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
def test_two_mfmas_per_product_in_the_step_loops_of_the_c2_chains():
    """Packed fp16 planes: a product of the 128-unit chains is TWO v_mfma_f32_16x16x32_f16, so a GRU step issues
    3 gates x 4 k-blocks x 2 = 24 in the forward role loops and 12 in the backward ones (a wave's share of the 384-deep
    product of the gate gradients with W_hid^T: six k-blocks).  Three MFMAs per product would be
    36 / 18.  (tools/isa_stats.py prints the whole mix.)"""
    text = rec_p_asm()
    for pat, want in ((r"^_Z11rec_fwd_x6pILi1ELb1ELb0ELb1EEv7RecArgs:", 24), (r"^_Z11rec_bwd_x6pILi1ELb0ELb0ELb1ELi1EEv7RecArgs:", 12)):
        st = next(i for i, ln in enumerate(text) if re.match(pat, ln))
        end = next(i for i in range(st, len(text)) if text[i].strip().startswith("s_endpgm"))
        body = text[st:end + 1]
        loops = loops_of(body)
        assert len(loops) == 2
        for lo, hi in loops:
            n = sum(1 for ln in body[lo:hi] if ln.strip().startswith("v_mfma_f32_16x16x32_f16"))
            assert n == want, (pat, n, want)
