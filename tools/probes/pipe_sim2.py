"""pipe_sim with explicit s_setprio tables: prio[wave][kb position] (higher wins), ties alternate (round robin)."""
import itertools
MF, PER_KB, KB = 16, 18, 4
def run(c=600, lds=200, steps=80, prio=None, order=None):
    owner = {0: "A", 1: "A", 2: "B", 3: "B"}
    order = order or {"A": [0, 1, 2, 3], "B": [0, 1, 2, 3]}
    pub = {"A": {0: 0.0}, "B": {0: 0.0}}
    wave = {w: dict(t=0, i=0, ready=0.0, left=0) for w in "AB"}
    pipe_free, last = 0.0, "B"
    done = []
    while min(wave["A"]["t"], wave["B"]["t"]) < steps:
        cand = []
        for w, s in wave.items():
            if s["t"] >= steps: continue
            if s["left"] == 0:
                kb = order[w][s["i"]]
                tneed = pub[owner[kb]].get(s["t"], 1e18)
                start = max(s["ready"], tneed + lds if s["i"] == 0 else max(tneed + lds * 0.5, s["ready"]))
            else:
                start = s["ready"]
            cand.append((w, start))
        tmin = min(max(pipe_free, st) for _, st in cand)
        if tmin > 1e17: raise RuntimeError("deadlock")
        ready = [w for w, st in cand if max(pipe_free, st) <= tmin + 1e-9]
        if len(ready) > 1:
            pa, pb = prio["A"][wave["A"]["i"]], prio["B"][wave["B"]["i"]]
            w = "A" if pa > pb else ("B" if pb > pa else ("A" if last == "B" else "B"))
        else:
            w = ready[0]
        last = w
        s = wave[w]
        if s["left"] == 0: s["left"] = PER_KB
        s["left"] -= 1
        pipe_free = tmin + MF
        s["ready"] = tmin + 4
        if s["left"] == 0:
            s["i"] += 1
            if s["i"] == KB:
                end = pipe_free + c
                pub[w][s["t"] + 1] = end
                if w == "B": done.append(end)
                s["t"] += 1; s["i"] = 0; s["ready"] = end
    h = len(done) // 2
    return (done[-1] - done[h]) / (len(done) - 1 - h)
if __name__ == "__main__":
    for c in (450, 600, 800):
        res = []
        for pa in itertools.product(range(4), repeat=4):
            if list(pa) != sorted(pa): continue
            for pb in itertools.product(range(4), repeat=4):
                if list(pb) != sorted(pb): continue
                res.append((run(c=c, prio={"A": pa, "B": pb}), pa, pb))
        res.sort()
        print("c=%d  equal prio: %.0f   best:" % (c, run(c=c, prio={"A": (0,0,0,0), "B": (0,0,0,0)})))
        for r in res[:6]: print("   %.0f A%s B%s" % r)
        print("   worst %.0f" % res[-1][0])
