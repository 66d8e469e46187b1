"""Fold rocprofv3 counter CSVs (one pass per counter) + the kernel-stats CSV into one JSON summary.

usage: pmc_summary.py <fetch_dir> <write_dir> [<stats_dir>]
Per kernel: mean FETCH_SIZE / WRITE_SIZE (KB, as rocprofv3 reports them) over the launches after the first,
the corrected HBM bytes per launch, and the average duration from --stats.  gfx950 correction
(MI355X_MICROARCH.md, HBM/rocprofv3 section; calibrated in round 1 on gather_xt, whose writes are exactly
T*B*G*H*4 bytes, and rec_fwd, whose reads are exactly xt): FETCH_SIZE under-counts 16-B/lane streams by 2x
-> hbm_read = 2 * FETCH_SIZE * 1024; WRITE_SIZE * 1024 as is."""
import csv
import glob
import json
import sys
from collections import defaultdict


def counter(dirname, name):
    acc = defaultdict(list)
    for fn in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == name:
                    acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fetch, write = counter(sys.argv[1], "FETCH_SIZE"), counter(sys.argv[2], "WRITE_SIZE")
    dur = {}
    if len(sys.argv) > 3:
        for fn in glob.glob(sys.argv[3] + "/**/*kernel_stats.csv", recursive=True):
            with open(fn) as f:
                for row in csv.DictReader(f):
                    dur[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]))
    out = {"_note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (+ --kernel-trace only); mean over "
                    "launches after the first; KB as reported.  hbm_bytes_per_launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                    "(gfx950 FETCH_SIZE correction, see tools/pmc_summary.py).  avg_ns from the --stats pass.",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write), key=lambda k: -(dur.get(k, (0, 0))[0] * dur.get(k, (0, 0))[1])):
        fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
        fm = sum(fv[1:]) / max(1, len(fv) - 1) if len(fv) > 1 else fv[0]
        wm = sum(wv[1:]) / max(1, len(wv) - 1) if len(wv) > 1 else wv[0]
        e = {"launches": len(fv), "FETCH_SIZE_KB": round(fm, 1), "WRITE_SIZE_KB": round(wm, 1),
             "hbm_bytes_per_launch": int(2 * fm * 1024 + wm * 1024)}
        if k in dur:
            e["calls_in_stats"], e["avg_ns"] = dur[k][0], round(dur[k][1], 1)
            e["hbm_GBps"] = round(e["hbm_bytes_per_launch"] / dur[k][1], 1)
        out["kernels"][k] = e
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
