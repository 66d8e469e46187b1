"""Discrete-event model of one SIMD of the 4-row-tile recurrent kernel (2 waves share a matrix pipe) to compare
synchronisation schemes: full barrier per step vs per-k-block dataflow flags.  All 4 SIMDs behave alike, so the
"A" wave stands for waves 0-3 (units 0-63 = k-blocks 0,1) and "B" for waves 4-7 (k-blocks 2,3)."""
import sys

MF = 16          # cycles per MFMA on the pipe
PER_KB = 18      # MFMAs per wave per k-block (6 products x 3 gates)
KB = 4

def run(c=950, lds=130, steps=60, scheme="flags", order=None, prio="late", bar=100):
    # state per wave: t, position in its kb order, time it becomes free
    owner = {0: "A", 1: "A", 2: "B", 3: "B"}
    order = order or {"A": [0, 1, 2, 3], "B": [0, 1, 2, 3]}
    pub = {"A": {0: 0.0}, "B": {0: 0.0}}          # pub[w][t] = time h_w(t) is visible
    wave = {w: dict(t=0, i=0, ready=0.0, left=0, cell_end=None) for w in "AB"}
    pipe_free = 0.0
    now = 0.0
    done_t = {"A": [], "B": []}
    while min(wave["A"]["t"], wave["B"]["t"]) < steps:
        # candidates: waves that can issue an MFMA now
        cand = []
        for w, s in wave.items():
            if s["t"] >= steps: continue
            if s["left"] == 0:
                if s["i"] == KB: continue   # in cell
                kb = order[w][s["i"]]
                src = owner[kb]
                if scheme == "barrier":
                    tneed = max(pub["A"].get(s["t"], 1e18), pub["B"].get(s["t"], 1e18)) + bar
                else:
                    tneed = pub[src].get(s["t"], 1e18)
                start = max(s["ready"], tneed + (lds if s["i"] == 0 or scheme != "barrier" else 0))
                cand.append((w, start))
            else:
                cand.append((w, s["ready"]))
        if not cand:
            # everyone in cell or blocked: advance to next event handled below
            pass
        # pick the wave to issue next MFMA at time max(pipe_free, start)
        best = None
        for w, st in cand:
            t_issue = max(pipe_free, st)
            s = wave[w]
            if prio == "late":      key = (t_issue, s["t"], s["i"])            # earliest; tie: behind in progress
            elif prio == "near":    key = (t_issue, -s["i"], s["t"])           # tie: nearer to its cell
            elif prio == "A":       key = (t_issue, w)
            else:                   key = (t_issue, )
            if best is None or key < best[0]: best = (key, w, t_issue)
        if best is None or best[2] >= 1e17:
            raise RuntimeError("deadlock")
        _, w, ti = best
        s = wave[w]
        if s["left"] == 0: s["left"] = PER_KB
        s["left"] -= 1
        pipe_free = ti + MF
        s["ready"] = ti + 4          # issue slots
        if s["left"] == 0:
            s["i"] += 1
            if s["i"] == KB:
                end = pipe_free + c
                pub[w][s["t"] + 1] = end
                done_t[w].append(end)
                s["t"] += 1; s["i"] = 0; s["ready"] = end
    a = done_t["B"]
    return (a[-1] - a[len(a) // 2]) / (len(a) - 1 - len(a) // 2)

if __name__ == "__main__":
    for c in (500, 700, 950):
        print("c=%d" % c)
        print("  barrier            %.0f" % run(c=c, scheme="barrier"))
        for prio in ("late", "near", "A", "fifo"):
            for oa, ob in (([0,1,2,3],[0,1,2,3]), ([0,1,2,3],[2,3,0,1]), ([2,3,0,1],[0,1,2,3]), ([0,1,2,3],[0,1,3,2])):
                print("  flags prio=%-5s A%s B%s  %.0f" % (prio, oa, ob, run(c=c, order={"A": oa, "B": ob}, prio=prio)))
